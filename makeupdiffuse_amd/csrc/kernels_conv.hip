// 3x3 convolution (stride 1, pad 1) with LDS-staged input tiles on the gfx950 matrix cores.
// Replaces Conv2d(C, C', 3, padding=1) of every cldm ResBlock / the hint block, reached
// from diffmk/makeup_diffuse.py:164-168.
//
// The generic implicit GEMM (kernels_gemm.hip) gathers each A row per (tap, channel) K-step, so every input
// pixel travels L2 -> LDS nine times per output-channel tile and the kernel sits on the L2 -> LDS ceiling.
// Here a workgroup owns a SPATIAL tile (TH x TW pixels of IMGS images = TM rows):
//   * per 64-channel chunk the haloed patch [(TH+2) x (TW+2)] x 64 ch is loaded ONCE (global_load_lds, zero page
//     for the padding ring) into a 2-slot LDS buffer;
//   * the 9 taps are 9 MFMA K-steps whose A fragments are read from the patch at shifted pixel offsets
//     (ds_read_b128, same XOR swizzle as the GEMM), so only the weights are streamed per tap:
//     [TN x 64] tiles through a 3-stage ring;
//   * one raw s_barrier per K-step and counted s_waitcnt vmcnt(N): N = W tiles issued after the one being
//     waited for (+ the patch pieces when a patch prefetch was issued inside that window) - exact, because
//     every wave issues a fixed number of loads per tile / per patch;
//   * output/epilogue as in the GEMM (transposed product, 4 consecutive channels per lane, fused bias /
//     time-embedding row-bias / residual / split-K over channel chunks).
// Staged bytes per output drop 2-5x versus the gather and 256-pixel tiles become affordable.
#include "mkd_common.h"
#include "gemm_device.h"

namespace {

using namespace mkdk;

// Taps per K-step (and barrier).  One tap per step: barrier, then every wave's 16 ds_read_b128 queue up at the LDS before its 32 MFMAs -
// the kernel takes 78 % of its time with NO global loads at all (profiles/exp_r4_conv_ablate.txt).  THREE taps per step (a W stage
// holds three taps, two stages): a third of the barriers, and the reads of tap t + 1 overlap the MFMAs of tap t inside a step.  Taken
// where two patch slots + two three-tap W stages fit the 160 KB of LDS (every tile but the 256 x 128 ones).
#ifndef MKD_CONV_TPS
#define MKD_CONV_TPS 3
#endif
constexpr int conv_tps(int TN, int NW, int PP) {
    // only where the one-tap form already holds a CU alone (> 80 KB of LDS: its co-residency does not change) and with eight waves or more
    // (four-wave tiles get slower: six weight pieces per wave and step)
    return (MKD_CONV_TPS == 3 && NW >= 8 && 2 * PP * NW * 1024 + 3 * TN * 128 > 80 * 1024 && 2 * PP * NW * 1024 + 2 * 3 * TN * 128 + 4096 <= 160 * 1024) ? 3 : 1;
}
constexpr int conv_stages(int TN, int NW, int PP, int STAGES) { return conv_tps(TN, NW, PP) == 3 ? 2 : STAGES; }

// GNS = 1: the epilogue also accumulates the GroupNorm statistics of the output (gemm_device.h); separate instantiation so that
// the plain kernel keeps its registers and occupancy
template <int TM, int TN, int WM, int WN, int STAGES_, int PP, int GNS = 0>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_patch_kernel(const GemmArgs2 pg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = (int)blockIdx.z >= pg.g[0].gz ? 1 : 0;      // grouped launch: the second problem owns the upper half of grid z
    const GemmArgs& p = pg.g[grp];
    constexpr int NW = WM * WN;
    constexpr int WP = TN / 8 / NW;          // W pieces (1 KiB = 8 rows x 128 B) per wave per tile
    constexpr int NI = TN / WN / 16;
    constexpr int MI = TM / WM / 16;
    constexpr int PSLOT = PP * NW * 1024;    // bytes of one patch slot (PP pieces per wave)
    constexpr int TPS = conv_tps(TN, NW, PP);             // taps per step
    constexpr int STAGES = conv_stages(TN, NW, PP, STAGES_);
    constexpr int NTG = 9 / TPS;                           // steps per channel chunk
    constexpr int WTB = TN * 128;                          // bytes of one tap's W tile
    constexpr int WSB = TPS * WTB;                         // bytes of one W stage
    static_assert(WP >= 1 && NI >= 1 && MI >= 1 && (NW == 4 || NW == 8 || NW == 16) && STAGES >= 2 && STAGES <= 4, "layout");

    const bf16_t* const gA = p.A; const bf16_t* const gW = p.W; const bf16_t* const gZ = p.zero;
    const int lda = p.lda, ldw = p.ldw, M = p.M, N = p.N;
    const int H = p.Hin, Wd = p.Win, Cin = p.Cin;
    const int TH = p.tile_h, TW = p.tile_w, IMGS = p.tile_imgs;
    const int splitk = p.splitk, per = p.ksteps_per_split;      // `per` = channel chunks per split here
    float* const ws = p.ws;
    const Epilogue epi = make_epilogue(p);
    const int batch = M / (H * Wd);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid / 64;
    const int wm = w / WN, wn = w % WN;
    const int PH = TH + 2, PW = TW + 2, PPIX = PH * PW, NP = IMGS * PPIX;
    const int tiles_x = Wd / TW, tiles_y = H / TH;
    int bid, by, bz;
    xcd_tile_order(p.xcd_mode, p.gz, grp, bid, by, bz);
    const int tx = bid % tiles_x;
    const int ty = (bid / tiles_x) % tiles_y;
    const int b0 = (bid / (tiles_x * tiles_y)) * IMGS;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = by * TN;
    const int nch_total = Cin / BK;
    const int c_begin = bz * per;
    const int c_end = min(nch_total, c_begin + per);

    char* const patch0 = smem;
    char* const wring = smem + 2 * PSLOT;

    // ---- staging geometry (same lane -> (row, chunk) map and XOR key as the GEMM) --------------------------
    const int lrow = lane >> 3;
    const int key = (4 * (w & 1) + (lane >> 4)) & 7;
    const int sc = (lane & 7) ^ key;

    size_t poff[PP];
    bool pok[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int pp = 8 * (w + NW * i) + lrow;           // patch pixel held by this lane's row of piece i
        const int ppc = pp < NP ? pp : 0;
        const int img = ppc / PPIX;
        const int r = ppc - img * PPIX;
        const int ppy = r / PW, ppx = r - ppy * PW;
        const int iy = y0 - 1 + ppy, ix = x0 - 1 + ppx, b = b0 + img;
        pok[i] = pp < NP && b < batch && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        poff[i] = ((size_t)(b * H + iy) * Wd + ix) * lda + sc * 8;
    }
    size_t woff[WP];
    bool wok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int n = n0 + 8 * (w + NW * i) + lrow;
        wok[i] = n < N;
        woff[i] = (size_t)n * ldw + sc * 8;
    }

    auto issue_patch = [&](int slot, int c) {
        char* dst = patch0 + slot * PSLOT;
#pragma unroll
        for (int i = 0; i < PP; ++i)
            glds16(select_src(gA + poff[i] + c * BK, gZ, pok[i]), dst + (w + NW * i) * 1024);
    };
    auto issue_w = [&](int slot, int c, int tg) {          // the TPS taps of step (c, tg)
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            char* dst = wring + slot * WSB + t * WTB;
            const int k = (tg * TPS + t) * Cin + c * BK;
#pragma unroll
            for (int i = 0; i < WP; ++i)
                glds16(select_src(gW + woff[i] + k, gZ, wok[i]), dst + (w + NW * i) * 1024);
        }
    };

    // ---- MFMA fragment geometry -----------------------------------------------------------------------------
    const int frow = lane & 15;
    const int fq = lane >> 4;
    int pid0[MI];        // patch pixel of tap (0,0) for this lane's row of fragment mi
    int mrow[MI];        // global output row (pixel index) or -1
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pr = wm * (TM / WM) + mi * 16 + frow;
        const int img = pr / (TH * TW);
        const int r = pr - img * (TH * TW);
        const int py = r / TW, px = r - py * TW;
        pid0[mi] = img * PPIX + py * PW + px;
        const int b = b0 + img;
        mrow[mi] = b < batch ? (b * H + y0 + py) * Wd + x0 + px : -1;
    }

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int wslot, int pslot, int tg) {
        const char* ps = patch0 + pslot * PSLOT;
#pragma unroll
      for (int t = 0; t < TPS; ++t) {
        const int tap = tg * TPS + t;
        const char* wsm = wring + wslot * WSB + t * WTB;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int dpix = ky * PW + kx;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = 4 * kk + fq;
            bf16x8 xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int pidx = pid0[mi] + dpix;
                xf[mi] = *(const bf16x8*)(ps + pidx * 128 + ((chunk ^ ((pidx >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * (TN / WN) + ni * 16 + frow;
                wf[ni] = *(const bf16x8*)(wsm + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }
    };

    // GNS: (image, group) accumulator of this tile behind the W ring (GACCB bytes, see the launcher), zeroed here
    constexpr int GACCB = 4096;
    long long* const gacc = (long long*)(wring + STAGES * WSB);
    int g_nseg = 1, g_first = 0, g_ngl = 1;
    bool gfast = false;
    if (GNS) {
        const int vc = min(TN, N - n0);
        g_nseg = min(IMGS, batch - b0);
        g_first = (p.gn_coff + n0) / p.gn_cg;
        g_ngl = (p.gn_coff + n0 + vc - 1) / p.gn_cg - g_first + 1;
        gfast = ((TH * TW) % (TM / WM)) == 0 && g_nseg * g_ngl <= GACCB / 16;        // every wave's rows lie in one image of the tile
        if (gfast)
            for (int i = tid; i < g_nseg * g_ngl * 2; i += 64 * NW) gacc[i] = 0;
    }

    // ---- K loop over (chunk, tap) ------------------------------------------------------------------------------
    const int nch = c_end - c_begin;
    const int nsteps = nch * NTG;
    // bias / residual fragments first: older than every tile load (in-order vmcnt: the counted waits are unaffected), so their
    // latency hides under the K loop instead of being paid after it
    const bool pre = splitk == 1;
    f32x4 pbias[NI];
    U16x4 pres[NI][MI];
    if (pre) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
            pbias[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi.bias && n < N) pbias[ni] = *(const f32x4*)(epi.bias + n);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int j = 0; j < 4; ++j) pres[ni][mi].v[j] = 0;
                if (epi.R && n < N && mrow[mi] >= 0) pres[ni][mi] = *(const U16x4*)(epi.R + (size_t)mrow[mi] * epi.ldr + n);
            }
        }
    }
    if (nsteps > 0) {
        issue_patch(0, c_begin);
        {
            int c = 0, tap = 0;
#pragma unroll
            for (int s = 0; s < STAGES - 1; ++s) {
                if (s < nsteps) issue_w(s, c_begin + c, tap);
                if (++tap == NTG) { tap = 0; ++c; }
            }
        }
        int c = 0, tap = 0;                 // step i = (c, tap group)
        int pc = 0, ptap = STAGES - 1;      // step i + STAGES - 1
        while (ptap >= NTG) { ptap -= NTG; ++pc; }
        int wslot = 0, nslot = STAGES - 1;
        int last_patch = -1000;
        for (int i = 0; i < nsteps; ++i) {
            const int after_w = min(STAGES - 2, nsteps - 1 - i);
            const bool after_p = last_patch >= i - (STAGES - 1);      // a patch prefetch was issued after W(i)
            switch (after_w * 2 + (after_p ? 1 : 0)) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<PP>(); break;
                case 2: wait_vmcnt<TPS * WP>(); break;
                case 3: wait_vmcnt<TPS * WP + PP>(); break;
                case 4: wait_vmcnt<2 * TPS * WP>(); break;
                default: wait_vmcnt<2 * TPS * WP + PP>(); break;
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + STAGES - 1 < nsteps) issue_w(nslot, c_begin + pc, ptap);
            if (tap == 0 && c + 1 < nch) { issue_patch((c + 1) & 1, c_begin + c + 1); last_patch = i; }
            compute(wslot, c & 1, tap);
            if (++tap == NTG) { tap = 0; ++c; }
            if (++ptap == NTG) { ptap = 0; ++pc; }
            wslot = (wslot + 1 == STAGES) ? 0 : wslot + 1;
            nslot = (nslot + 1 == STAGES) ? 0 : nslot + 1;
        }
    }

    if constexpr (GNS != 0) {
        // ---- epilogue with GroupNorm statistics (as in gemm_kernel; tile row = local pixel index, one segment per image) ----
        constexpr int GTS = TN + 4;
        constexpr int GTILE = (TM * GTS * 2 + 15) & ~15;
        constexpr int LDS_MIN = 2 * PP * NW * 1024 + STAGES * WSB + GACCB;
        static_assert(GTILE + 64 * 16 <= LDS_MIN, "GroupNorm statistics tile must fit in the staging buffers");
        uint16_t* const gtile = (uint16_t*)smem;
        if (!gfast) __syncthreads();
        const int wseg = (wm * (TM / WM)) / (TH * TW);               // image of the tile this wave's rows belong to (fast path)
        const int vcols = min(TN, N - n0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
            float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = mrow[mi];
                    if (m < 0) continue;
                    const U16x4 o = epilogue_write_bits(epi, m, n, epilogue_value_pre(epi, m, n, acc[ni][mi], pbias[ni], pres[ni][mi]));
                    if (gfast) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float f = bf16_to_f32(o.v[j]); cs[j] += f; cq[j] += f * f; }
                    } else {
                        *(U16x4*)(gtile + (wm * (TM / WM) + mi * 16 + frow) * GTS + wn * (TN / WN) + ni * 16 + 4 * fq) = o;
                    }
                }
            }
            if (gfast && wseg < g_nseg) gn_wave_stats(cs, cq, frow, fq, wn * (TN / WN) + ni * 16, vcols, p.gn_cg, p.gn_coff + n0, wseg, g_ngl, gacc);
        }
        if (gfast) {
            __syncthreads();
            gn_acc_flush(gacc, g_nseg, g_ngl, b0, g_first, tid, 64 * NW, p.gn_stat);
        } else {
            gn_tile_stats(gtile, GTS, TM, TN, 64 * NW, tid, min(TM, (batch - b0) * TH * TW), vcols, TH * TW, 0, b0, p.gn_cg, p.gn_coff + n0,
                          (long long*)(smem + GTILE), (LDS_MIN - GTILE) / 16, p.gn_stat);
        }
        return;
    }
    // ---- epilogue -------------------------------------------------------------------------------------------------
    // fast form (gemm_device.h: epilogue_fast_store): bias / row bias / scale / residual -> bf16 without per-fragment branches; the
    // rows of a wave are pixels of ONE image whenever TH * TW is a multiple of the wave's rows (wave-uniform test)
    if (pre && epi.act == 0 && !epi.out_f32) {
        int rb_b = -1;
        bool ok = true;
        if (epi.rowbias) {
            const int pr0 = wm * (TM / WM);
            const int img0 = pr0 / (TH * TW), img1 = (pr0 + TM / WM - 1) / (TH * TW);
            rb_b = ((b0 + img0) * H * Wd) / epi.rpb;
            ok = img0 == img1 && epi.rpb == H * Wd && b0 + img0 < batch;
        }
        if (ok) {
            f32x4 prb[NI];
            if (rb_b >= 0) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
                    prb[ni] = n < N ? *(const f32x4*)(epi.rowbias + (size_t)rb_b * epi.ldrb + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = mrow[mi];
                if (m < 0) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
                    if (n >= N) continue;
                    if (rb_b >= 0) epilogue_fast_store<true>(epi, m, n, acc[ni][mi], pbias[ni], prb[ni], pres[ni][mi]);
                    else epilogue_fast_store<false>(epi, m, n, acc[ni][mi], pbias[ni], pbias[ni], pres[ni][mi]);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mrow[mi];
        if (m < 0) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
            if (n >= N) continue;
            if (splitk > 1) *(f32x4*)(ws + ((size_t)bz * M + m) * N + n) = acc[ni][mi];
            else epilogue_write(epi, m, n, epilogue_value_pre(epi, m, n, acc[ni][mi], pbias[ni], pres[ni][mi]));
        }
    }
}

template <int TM, int TN, int WM, int WN, int PP>
int launch_patch(const GemmArgs& a, dim3 grid, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, (int)grid.z);
    if (second) grid.z *= 2;
    constexpr int STAGES = 3;
    constexpr int NW = WM * WN;
    const bool gns = a.gn_stat != nullptr && a.splitk == 1;
    const size_t lds = (size_t)2 * PP * NW * 1024 + (size_t)conv_stages(TN, NW, PP, STAGES) * conv_tps(TN, NW, PP) * TN * 128 + (gns ? 4096 : 0);      // (+ GroupNorm statistics accumulator)
    if (lds > 160 * 1024) return mkd_fail(-4, "conv3x3_patch: LDS budget exceeded");
    static bool attr_set[2] = {false, false};
    if (!attr_set[gns]) {
        hipError_t e = gns ? hipFuncSetAttribute((const void*)conv3x3_patch_kernel<TM, TN, WM, WN, STAGES, PP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                           : hipFuncSetAttribute((const void*)conv3x3_patch_kernel<TM, TN, WM, WN, STAGES, PP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(conv patch LDS): ") + hipGetErrorString(e));
        attr_set[gns] = true;
    }
    if (gns) hipLaunchKernelGGL((conv3x3_patch_kernel<TM, TN, WM, WN, STAGES, PP, 1>), grid, dim3(64 * NW), lds, stream, ag);
    else     hipLaunchKernelGGL((conv3x3_patch_kernel<TM, TN, WM, WN, STAGES, PP, 0>), grid, dim3(64 * NW), lds, stream, ag);
    return 0;
}

template <int TM, int TN, int WM, int WN>
int launch_patch_pp(const GemmArgs& a, int pp, dim3 grid, hipStream_t stream, const GemmArgs* second) {
    if constexpr (WM * WN == 16) {                      // 16 waves: 2 or 3 pieces per wave hold a 256- / 128-pixel patch
        if (pp <= 2) return launch_patch<TM, TN, WM, WN, 2>(a, grid, stream, second);
        if (pp == 3) return launch_patch<TM, TN, WM, WN, 3>(a, grid, stream, second);
        return mkd_fail(-4, "conv3x3_patch: patch too large for the 16-wave tile");
    } else {
        if constexpr (WM * WN == 8 && TM <= 128) {      // 8 waves on a 64- / 128-pixel tile: 2 or 3 pieces per wave hold the patch
            if (pp <= 2) return launch_patch<TM, TN, WM, WN, 2>(a, grid, stream, second);
            if (pp == 3) return launch_patch<TM, TN, WM, WN, 3>(a, grid, stream, second);
        }
        if (pp <= 4) return launch_patch<TM, TN, WM, WN, 4>(a, grid, stream, second);
        if (pp == 5) return launch_patch<TM, TN, WM, WN, 5>(a, grid, stream, second);
        if (pp == 6) return launch_patch<TM, TN, WM, WN, 6>(a, grid, stream, second);
        if (pp == 7) return launch_patch<TM, TN, WM, WN, 7>(a, grid, stream, second);
        if (pp <= 9) return launch_patch<TM, TN, WM, WN, 9>(a, grid, stream, second);
        return mkd_fail(-4, "conv3x3_patch: patch too large");
    }
}

}  // namespace

// spatial tiling of a TM-row block: TW = min(W, 16), TH = min(H, TM / TW), IMGS images per block
bool conv_patch_geometry(int tm, int batch, int H, int W, int* th, int* tw, int* imgs, int* pieces_per_wave, int nwaves) {
    const int TW = W < 16 ? W : 16;
    if (TW < 4 || W % TW) return false;
    int TH = tm / TW;
    if (TH > H) TH = H;
    if (TH < 1 || H % TH) return false;
    const int IM = tm / (TH * TW);
    if (IM < 1 || IM * TH * TW != tm) return false;
    if (IM > 1 && IM > 2 * batch) return false;          // mostly-empty tiles
    const int np = IM * (TH + 2) * (TW + 2);
    int pp = ((np + 7) / 8 + nwaves - 1) / nwaves;
    if (pp > 9) return false;
    const int pp_min = ((nwaves == 8 && tm <= 128) || nwaves == 16) ? 2 : 4;
    if (pp < pp_min) pp = pp_min;
    if (nwaves == 16 && pp > 3) return false;
    if (pp == 8) pp = 9;
    *th = TH; *tw = TW; *imgs = IM; *pieces_per_wave = pp;
    return true;
}

// cfg: 6: 256x128 (8 waves), 7: 256x64 (8 waves), 8: 128x128, 9: 128x64, 10: 64x128, 11: 64x64 (4 waves);
// 38: 128x64, 39: 64x128, 40: 128x128 with EIGHT waves (32x32 / 32x32 / 32x64 per wave): a wave's LDS-DMA transfers complete one
// after the other (~1 KiB per 200-300 cycles, tools/micro/stream_rate3.hip), and with four waves each tap asks 2.7 pieces of every
// wave for 16 MFMAs - the tap waits for the transfers, not for the matrix cores; eight waves halve the pieces per wave
static bool patch_cfg_shape(int cfg, int* tm, int* tn, int* nw) {
    static const int tms[11] = {256, 256, 128, 128, 64, 64, 128, 64, 128, 256, 128};
    static const int tns[11] = {128, 64, 128, 64, 128, 64, 64, 128, 128, 128, 128};
    static const int nws[11] = {8, 8, 4, 4, 4, 4, 8, 8, 8, 16, 16};
    int i;
    if (cfg >= 6 && cfg <= 11) i = cfg - 6;
    else if (cfg >= 38 && cfg <= 40) i = cfg - 38 + 6;
    else if (cfg >= 42 && cfg <= 43) i = cfg - 42 + 9;          // 42: 256x128, 43: 128x128 with SIXTEEN waves (32x64 / 32x32 per wave)
    else return false;
    *tm = tms[i]; *tn = tns[i]; *nw = nws[i];
    return true;
}

bool conv_patch_supported(const GemmArgs& a, int cfg) {
    if (!a.conv || a.stride != 1 || a.up != 0 || a.Cin % 64 || a.Hin != a.Hout || a.Win != a.Wout) return false;
    int tm, tn, nw;
    if (!patch_cfg_shape(cfg, &tm, &tn, &nw)) return false;
    int th, tw, im, pp;
    if (!conv_patch_geometry(tm, a.M / (a.Hin * a.Win), a.Hin, a.Win, &th, &tw, &im, &pp, nw)) return false;
    const size_t lds = (size_t)2 * pp * nw * 1024 + (size_t)conv_stages(tn, nw, pp, 3) * conv_tps(tn, nw, pp) * tn * 128 + (a.gn_stat ? 4096 : 0);
    return lds <= 160 * 1024;
}

int launch_conv_patch(GemmArgs a, int cfg, int splitk, hipStream_t stream, const GemmArgs* second) {
    if (!conv_patch_supported(a, cfg)) return mkd_fail(-4, "conv3x3_patch: unsupported shape for this tile");
    if (second && !gemm_same_geometry(a, *second)) return mkd_fail(-1, "conv3x3_patch: a grouped launch needs two problems of identical geometry");
    int tm, tn, nw;
    patch_cfg_shape(cfg, &tm, &tn, &nw);
    const int batch = a.M / (a.Hin * a.Win);
    int pp;
    conv_patch_geometry(tm, batch, a.Hin, a.Win, &a.tile_h, &a.tile_w, &a.tile_imgs, &pp, nw);
    const int nch = a.Cin / 64;
    int s = splitk < 1 ? 1 : splitk;
    if (s > nch) s = nch;
    const int per = (nch + s - 1) / s;
    s = (nch + per - 1) / per;
    if (a.defer_epilogue && (s < 2 || (a.expect_splitk > 0 && s != a.expect_splitk)))
        return mkd_fail(-3, "conv3x3_patch: deferred split-K epilogue planned for " + std::to_string(a.expect_splitk) + " slabs, the launch resolves to " +
                                std::to_string(s) + " (tile / split-K settings changed after mkd_prepare: prepare again)");
    if (s > 1 && !a.ws) return mkd_fail(-1, "conv3x3_patch: split-K needs a workspace");
    if (s > 1 && gemm_ws_bytes(a.M, a.N, s) > a.ws_bytes) return mkd_fail(-1, "conv3x3_patch: split-K workspace too small");
    if (s > 1 && second && (!second->ws || second->ws == a.ws || gemm_ws_bytes(a.M, a.N, s) > second->ws_bytes))
        return mkd_fail(-1, "conv3x3_patch: split-K needs a workspace per problem");
    a.splitk = s;
    a.ksteps_per_split = per;
    GemmArgs b;
    if (second) {
        b = *second; b.zero = a.zero; b.splitk = s; b.ksteps_per_split = per; b.xcd_mode = a.xcd_mode;
        b.tile_h = a.tile_h; b.tile_w = a.tile_w; b.tile_imgs = a.tile_imgs;
    }
    const GemmArgs* const sp = second ? &b : nullptr;
    const int groups = (batch + a.tile_imgs - 1) / a.tile_imgs;
    dim3 grid(groups * (a.Hin / a.tile_h) * (a.Win / a.tile_w), (a.N + tn - 1) / tn, s);
    int rc;
    switch (cfg) {
        case 6: rc = launch_patch_pp<256, 128, 4, 2>(a, pp, grid, stream, sp); break;
        case 7: rc = launch_patch_pp<256, 64, 4, 2>(a, pp, grid, stream, sp); break;
        case 8: rc = launch_patch_pp<128, 128, 2, 2>(a, pp, grid, stream, sp); break;
        case 9: rc = launch_patch_pp<128, 64, 2, 2>(a, pp, grid, stream, sp); break;
        case 10: rc = launch_patch_pp<64, 128, 2, 2>(a, pp, grid, stream, sp); break;
        case 38: rc = launch_patch_pp<128, 64, 4, 2>(a, pp, grid, stream, sp); break;
        case 39: rc = launch_patch_pp<64, 128, 2, 4>(a, pp, grid, stream, sp); break;
        case 40: rc = launch_patch_pp<128, 128, 4, 2>(a, pp, grid, stream, sp); break;
        case 42: rc = launch_patch_pp<256, 128, 8, 2>(a, pp, grid, stream, sp); break;
        case 43: rc = launch_patch_pp<128, 128, 4, 4>(a, pp, grid, stream, sp); break;
        default: rc = launch_patch_pp<64, 64, 2, 2>(a, pp, grid, stream, sp); break;
    }
    if (rc) return rc;
    MKD_LAUNCH_CHECK("conv3x3_patch_kernel");
    if (s > 1 && !a.defer_epilogue) return launch_splitk_epilogue(a, stream, sp);
    return 0;
}
