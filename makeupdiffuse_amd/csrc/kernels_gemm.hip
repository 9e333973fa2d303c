// Implicit-GEMM on the gfx950 matrix cores: the conv3x3 / 1x1 / linear contraction of the
// UNet + ControlNet (SURVEY.md §2 "Build kernel" table rows 1-3, 6; reference call sites
// diffmk/makeup_diffuse.py:164-168 via cldm ResBlock / SpatialTransformer / zero_convs).
//
//   C[m, n] = act((sum_k X[m, k] * W[n, k] + bias[n] + rowbias[m / rpb][n]) * scale + R[m, n])
//
// X rows are either plain rows (1x1 conv / linear) or gathered on the fly from an NHWC bf16
// image (3x3, pad 1, stride 1|2, optional nearest x2 upsample of the input), K ordered (ky,kx,ci).
//
// Design (CDNA4):
//   * block tile 128 (m) x TN (n, 64|128) x 64 (k); 4 waves as 2x2, each wave owns 64 x TN/2.
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B/lane, 1 KiB per
//     wave-instruction, no VGPR round trip).  The LDS image is lane-linear, so the bank-conflict
//     XOR swizzle is applied to the per-lane SOURCE chunk and again on the ds_read_b128 side;
//     padding / out-of-range rows / conv halo read from a zero page instead of branching.
//   * 3-4 stage LDS ring, one raw s_barrier per K-step, counted s_waitcnt vmcnt(N): 2-3 K-steps of
//     global_load_lds stay in flight across every barrier (the small-M layers are latency-bound).
//   * the product is computed transposed (D = W_tile . X_tile^T with v_mfma_f32_16x16x32_bf16) so
//     each lane ends with 4 CONSECUTIVE n for one m: 8-byte bf16x4 stores, float4 bias loads.
//   * small-M layers (4x4 / 8x8 latents) are weight-bandwidth bound: split-K over blockIdx.z with
//     fp32 partial slabs + a fused reduce/epilogue kernel fills the 256 CUs.
#include "mkd_common.h"
#include "gemm_device.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <tuple>

namespace {

using namespace mkdk;

// GNS = 1: the epilogue also accumulates the GroupNorm statistics of the output (gemm_device.h); a separate instantiation, so
// that the plain kernels keep their register budget (the statistics code costs 10-36 VGPRs and a wave of occupancy)
// KW > 1: K is ALSO split inside the workgroup: KW groups of WM x WN waves, each with its own LDS ring, take K-steps kg, kg + KW, ...
// and the partial accumulators are summed through LDS at the end.  The small-M layers have fewer tiles than CUs and their K loop is
// a chain of (load latency + barrier) steps with one wave per SIMD: KW groups put KW times the bytes in flight on the CU and cut the
// chain by KW without the second launch and the fp32 slab traffic of split-K over blocks.
template <int TM, int TN, int WM, int WN, int CONV, int STAGES, int LN = 0, int GNS = 0, int KW = 1>
__global__ __launch_bounds__(64 * WM * WN * KW) void gemm_kernel(const GemmArgs2 pg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = (int)blockIdx.z >= pg.g[0].gz ? 1 : 0;      // grouped launch: the second problem owns the upper half of grid z
    const GemmArgs& p = pg.g[grp];
    constexpr int XS = TM * 128;          // bytes of one X stage (TM rows x 64 bf16)
    constexpr int WSB = TN * 128;         // bytes of one W stage
    constexpr int STAGE = XS + WSB;
    constexpr int NW = WM * WN;           // waves per block (WM along m, WN along n)
    constexpr int WP = TN / 8 / NW;       // W pieces (1 KiB = 8 rows) per wave
    constexpr int XP = TM / 8 / NW;       // X pieces per wave
    constexpr int NI = TN / WN / 16;      // W fragments (16 rows) per wave
    constexpr int MI = TM / WM / 16;      // X fragments (16 rows) per wave
    static_assert(WP >= 1 && XP >= 1 && NI >= 1 && MI >= 1 && (NW == 4 || NW == 8 || NW == 16), "tile/wave layout");
    static_assert(STAGES >= 2 && STAGES <= 6 && 4 * (XP + WP) + (LN > 0 ? MI * LN : 0) < 64, "vmcnt immediates");
    static_assert(!(LN && GNS), "fused LayerNorm and GroupNorm statistics are separate kernels");
    // LN < 0: LayerNorm of the A rows "on the fly": the GEMM runs on the RAW rows with gamma-folded weights (as LN > 0) and takes
    // the row statistics itself, from the A fragments it already holds, with two extra MFMAs per row fragment and k-step -
    // ones . X^T gives the row sums, the diagonal of X . X^T the sums of squares - so no LayerNorm kernel and no producer-side
    // statistics are needed.  These layers are latency-bound (MFMA pipe < 15 % busy): the extra MFMAs are hidden.
    constexpr bool LNF = LN < 0;
    static_assert(KW == 1 || (LN <= 0 && !GNS), "in-block K split: plain epilogue (or on-the-fly LayerNorm) only");
    static_assert(!LNF || !CONV, "LayerNorm fusion is for linear GEMMs");
    static_assert(KW >= 1 && KW <= 4 && (KW - 1) * TM * TN * 4 <= KW * STAGES * STAGE, "in-block K split: partial tiles are summed in the rings");
    constexpr int LPT = XP + WP;          // global_load_lds per wave per K-tile (exact)

    // hoist the argument block into registers (keeps it out of scratch)
    const bf16_t* const gA = p.A; const bf16_t* const gW = p.W; const bf16_t* const gZ = p.zero;
    const int lda = p.lda, ldw = p.ldw, M = p.M, N = p.N, K = p.K;
    const int Hin = p.Hin, Win = p.Win, Cin = p.Cin, Hout = p.Hout, Wout = p.Wout, cstride = p.stride, up = p.up;
    const bf16_t* const gA2 = p.A2; const int lda2 = p.lda2; const int Kc = K - p.K2;      // CONV: columns [Kc, K) contract with the second input (folded 1x1)
    const int splitk = p.splitk, per = p.ksteps_per_split;
    float* const ws = p.ws;
    const Epilogue epi = make_epilogue(p);
    const float ln_eps = p.ln_eps;
    const float* const ln_s = p.ln_s;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kg = KW > 1 ? (tid >> 6) / NW : 0;              // K group of this wave
    const int w = KW > 1 ? (tid >> 6) % NW : tid >> 6;        // wave inside its group
    char* const smg = smem + kg * (STAGES * STAGE);           // this group's ring
    const int wm = w / WN, wn = w % WN;
    int bx, by, bz;
    xcd_tile_order(p.xcd_mode, p.gz, grp, bx, by, bz);
    const int m0 = bx * TM;
    const int n0 = by * TN;
    const int nk_total = (K + BK - 1) / BK;
    const int kt_begin = bz * per;
    const int kt_end = min(nk_total, kt_begin + per);

    // ---- per-lane staging geometry -----------------------------------------------------------
    const int lrow = lane >> 3;                               // row inside a 1 KiB piece
    const int key = (4 * (w & 1) + (lane >> 4)) & 7;          // == ((tile_row >> 1) & 7) for every piece of this wave (NW even)
    const int sc = (lane & 7) ^ key;                          // source 16-B chunk inside the 128-B K row

    size_t xoff[XP];                   // LINEAR: element offset of the row;  CONV: pixel-index base of the sample
    int uy0[XP], ux0[XP];              // CONV: top-left input coordinate of the 3x3 window (in upsampled space)
    bool xok[XP];
    const int Hup = Hin << up, Wup = Win << up;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int m = m0 + 8 * (w + NW * i) + lrow;
        xok[i] = m < M;
        if (CONV) {
            const int hw = Hout * Wout;
            const int mm = xok[i] ? m : 0;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / Wout;
            const int ox = rem - oy * Wout;
            xoff[i] = (size_t)b * Hin * Win;
            uy0[i] = oy * cstride - 1;
            ux0[i] = ox * cstride - 1;
        } else {
            xoff[i] = (size_t)m * lda;
            uy0[i] = ux0[i] = 0;
        }
    }
    size_t woff[WP];
    bool wok[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int n = n0 + 8 * (w + NW * i) + lrow;
        wok[i] = n < N;
        woff[i] = (size_t)n * ldw;
    }

    auto stage = [&](int buf, int kt) {
        char* xs = smg + buf * STAGE;
        char* wsm = xs + XS;
        const int k = kt * BK + sc * 8;
        const bool kok = k < K && kt < kt_end;                // (KW > 1: the last round may have no tile for this group)
        int ci = 0, ky = 0, kx = 0;
        const bool second = CONV && k >= Kc;          // (uniform per K-step: Kc is a multiple of BK)
        if (CONV) {
            const int tap = k / Cin;
            ci = k - tap * Cin;
            ky = (tap * 11) >> 5;          // tap / 3 for tap in [0, 9)
            kx = tap - 3 * ky;
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const bf16_t* real;
            bool ok;
            if (CONV && second) {          // folded 1x1: the output pixel's own row of the second input (stride 1: input pixel (uy0 + 1, ux0 + 1))
                ok = kok && xok[i];
                real = gA2 + ((xoff[i] + (size_t)((uy0[i] + 1) * Win + ux0[i] + 1)) * lda2 + (k - Kc));
            } else if (CONV) {
                const int uy = uy0[i] + ky, ux = ux0[i] + kx;
                ok = kok && xok[i] && (unsigned)uy < (unsigned)Hup && (unsigned)ux < (unsigned)Wup;
                real = gA + ((xoff[i] + (size_t)((uy >> up) * Win + (ux >> up))) * lda + ci);
            } else {
                ok = kok && xok[i];
                real = gA + xoff[i] + k;
            }
            glds16(select_src(real, gZ, ok), xs + (w + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i)
            glds16(select_src(gW + woff[i] + k, gZ, kok && wok[i]), wsm + (w + NW * i) * 1024);
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;
    f32x4 ssum[LNF ? MI : 1], ssq[LNF ? MI : 1];          // LNF: row sums (every register) / X . X^T (diagonal = sums of squares)
#pragma unroll
    for (int mi = 0; mi < (LNF ? MI : 1); ++mi) { ssum[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    auto compute = [&](int buf) {
        const char* xs = smg + buf * STAGE;
        const char* wsm = xs + XS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = 4 * kk + fq;
            bf16x8 xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (TM / WM) + mi * 16 + frow;
                xf[mi] = *(const bf16x8*)(xs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
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
            if (LNF) {          // the WN waves that share these rows split the row fragments between them (shared through LDS at the end)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (MI >= WN && (mi % WN) != wn) continue;          // (wave-uniform)
                    ssum[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, xf[mi], ssum[mi], 0, 0, 0);
                    ssq[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mi], xf[mi], ssq[mi], 0, 0, 0);
                }
            }
        }
    };

    const float* const stat_in = p.stat_in; const int stat_in_slots = p.stat_in_slots;
    float* const stat_out = p.stat_out;
    constexpr int SL = LN > 0 ? LN : 1;          // LN = 0: off; 1 / 3 / 5: stat loads per lane per row
    float2 lnt[LN > 0 ? MI : 1][SL];

    // ---- K loop: STAGES-deep LDS ring, STAGES-1 tiles of global_load_lds in flight across each barrier ----
    // Every wave issues exactly LPT loads per tile, in order, so "tile i landed" == "at most LPT * (tiles
    // issued after i) of my loads are still outstanding": a counted s_waitcnt, never a full drain.
    const int ntile = (kt_end - kt_begin + KW - 1) / KW;      // rounds: round i stages K-step kt_begin + i * KW + kg
    // Epilogue operands (bias, residual) are fetched FIRST: older than every tile load, they retire before tile 0 is waited on
    // (vmcnt is in-order, so the counted waits below are unaffected) and their latency hides under the K loop instead of being
    // paid after it - these layers are latency-bound and K is often 5 steps.
    const bool pre = splitk == 1 && !LN && kg == 0 && NW < 16;          // (16 waves: 128 registers per wave, no room for the prefetched operands)
    // GNS: (sample, group) accumulator of this tile behind the ring, zeroed here, long before the epilogue (K-loop barriers between)
    constexpr int GTAIL = 4096;
    long long* const gacc = (long long*)(smem + STAGES * STAGE);
    int g_b_first = 0, g_nseg = 1, g_first = 0, g_ngl = 1;
    bool gfast = false;
    if (GNS) {
        const int vr = min(TM, M - m0), vc = min(TN, N - n0);
        g_b_first = m0 / p.gn_hw;
        g_nseg = (m0 + vr - 1) / p.gn_hw - g_b_first + 1;
        g_first = (p.gn_coff + n0) / p.gn_cg;
        g_ngl = (p.gn_coff + n0 + vc - 1) / p.gn_cg - g_first + 1;
        gfast = (p.gn_hw % (TM / WM)) == 0 && g_nseg * g_ngl <= GTAIL / 16;       // every wave's rows lie in one sample
        if (gfast)
            for (int i = tid; i < g_nseg * g_ngl * 2; i += 64 * NW) gacc[i] = 0;
    }
    f32x4 pbias[NI];
    U16x4 pres[NI][MI];
    if (pre) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * (lane >> 4);
            pbias[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi.bias && n < N) pbias[ni] = *(const f32x4*)(epi.bias + n);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + wm * (TM / WM) + mi * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) pres[ni][mi].v[j] = 0;
                if (epi.R && n < N && m < M) pres[ni][mi] = *(const U16x4*)(epi.R + (size_t)m * epi.ldr + n);
            }
        }
    }
    if (ntile > 0) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < ntile) stage(s, kt_begin + s * KW + kg);
        // fused LayerNorm: row statistics are issued AFTER the prologue tiles and consumed after the loop.  Loads
        // retire in order, so issued earlier they would hold up the first tile wait of every block (their lines
        // were just written by the producer GEMM, possibly through another XCD's L2).  Being younger than tiles
        // 0..STAGES-2 they add a constant NST to the counted waits of the first STAGES-1 iterations only.
        constexpr int NST = LN > 0 ? MI * SL : 0;
        if (LN > 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                ln_stats_issue<SL>(stat_in, stat_in_slots, M, m0 + wm * (TM / WM) + mi * 16 + frow, fq, lnt[mi]);
        }
        int buf = 0;                               // ring slot of tile i
        int nxt = STAGES - 1;                      // ring slot the next prefetch goes to
        for (int i = 0; i < ntile; ++i) {
            const int after = min(STAGES - 2, ntile - 1 - i);      // tiles issued after tile i
            const int nst = (LN > 0 && i < STAGES - 1) ? NST : 0;
            // counted wait: N = LPT * after (+ NST); immediates only, hence the switch
            if (nst) {
                switch (after) {
                    case 0: wait_vmcnt<NST>(); break;
                    case 1: wait_vmcnt<LPT + NST>(); break;
                    case 2: wait_vmcnt<2 * LPT + NST>(); break;
                    case 3: wait_vmcnt<(3 * LPT + NST) & 63>(); break;
                    default: wait_vmcnt<(4 * LPT + NST) & 63>(); break;
                }
            } else {
                switch (after) {
                    case 0: wait_vmcnt<0>(); break;
                    case 1: wait_vmcnt<LPT>(); break;
                    case 2: wait_vmcnt<2 * LPT>(); break;
                    case 3: wait_vmcnt<3 * LPT>(); break;
                    default: wait_vmcnt<4 * LPT>(); break;
                }
            }
            __builtin_amdgcn_s_barrier();          // tile i visible to all waves; slot of tile i-1 is free
            asm volatile("" ::: "memory");
            if (i + STAGES - 1 < ntile) stage(nxt, kt_begin + (i + STAGES - 1) * KW + kg);
            compute(buf);
            buf = (buf + 1 == STAGES) ? 0 : buf + 1;
            nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        }
    }

    if constexpr (KW > 1) {
        // ---- sum the K groups' partial tiles: [group - 1][fragment][thread] float4 images in the (now idle) rings ----
        __syncthreads();
        f32x4* const red = (f32x4*)smem;
        const int slot = w * 64 + lane;
        constexpr int NF = NI * MI + (LNF ? 2 * MI : 0);          // fragments per thread (+ the LayerNorm statistics tiles)
        static_assert((KW - 1) * NF * NW * 64 * 16 <= KW * STAGES * STAGE, "partial tiles must fit in the rings");
        if (kg > 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) red[((kg - 1) * NF + ni * MI + mi) * (NW * 64) + slot] = acc[ni][mi];
            if (LNF) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    red[((kg - 1) * NF + NI * MI + mi) * (NW * 64) + slot] = ssum[mi];
                    red[((kg - 1) * NF + NI * MI + MI + mi) * (NW * 64) + slot] = ssq[mi];
                }
            }
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KW; ++g) {          // fixed order: deterministic
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] += red[((g - 1) * NF + ni * MI + mi) * (NW * 64) + slot];
            if (LNF) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    ssum[mi] += red[((g - 1) * NF + NI * MI + mi) * (NW * 64) + slot];
                    ssq[mi] += red[((g - 1) * NF + NI * MI + MI + mi) * (NW * 64) + slot];
                }
            }
        }
    }
    // LNF: per-row (sum, sum of squares) of the whole tile in LDS: each wave publishes the row fragments it owns, all read theirs.
    // KW == 1: the region behind the ring (other waves may still be reading the last tile); KW > 1: every wave is past its K loop.
    float* const lnrow = (float*)(KW > 1 ? smem + KW * STAGES * STAGE - TM * 8 : smem + STAGES * STAGE);
    if constexpr (LNF) {
        static_assert(KW == 1 || (KW - 1) * (NI * MI + 2 * MI) * NW * 64 * 16 + TM * 8 <= KW * STAGES * STAGE, "LayerNorm row statistics behind the partial tiles");
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (MI >= WN && (mi % WN) != wn) continue;
            // this lane's column m = frow: its row sum sits in every register of ssum; its sum of squares is the diagonal element of
            // X . X^T, held by the lane of the same column whose row block is fq' = frow >> 2, in register frow & 3
            const int r = frow & 3;
            const float d = r == 0 ? ssq[mi][0] : (r == 1 ? ssq[mi][1] : (r == 2 ? ssq[mi][2] : ssq[mi][3]));
            if (fq == (frow >> 2)) {
                const int lr = wm * (TM / WM) + mi * 16 + frow;
                lnrow[lr * 2] = ssum[mi][0]; lnrow[lr * 2 + 1] = d;
            }
        }
        __syncthreads();
    }
    if constexpr (GNS != 0) {
        // ---- epilogue with GroupNorm statistics (split-K launches use the plain kernel + splitk_epilogue_gn_kernel) ----
        // Fast path: column-fragment major; per fragment the lane's 4 channels are summed over the wave's row fragments in
        // registers (8 values), reduced over the 16 lanes of a DPP row and added as fixed-point integers to the tile accumulator.
        // General path (a wave's rows span several samples): the stored bf16 tile is staged in LDS (the ring is free once every
        // wave has left the K loop) and reduced by gn_tile_stats.
        constexpr int GTS = TN + 4;
        constexpr int GTILE = (TM * GTS * 2 + 15) & ~15;
        static_assert(GTILE + 64 * 16 <= STAGES * STAGE + GTAIL, "GroupNorm statistics tile must fit in the ring");
        uint16_t* const gtile = (uint16_t*)smem;
        long long* const gn_stat = p.gn_stat;
        const int gn_cg = p.gn_cg, gn_coff = p.gn_coff, gn_hw = p.gn_hw;
        if (!gfast) __syncthreads();
        const int wrow0 = m0 + wm * (TM / WM);
        const int wseg = wrow0 / gn_hw - g_b_first;
        const int vcols = min(TN, N - n0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
            float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = m0 + wm * (TM / WM) + mi * 16 + frow;
                    if (m >= M) continue;
                    const U16x4 o = epilogue_write_bits(epi, m, n, epilogue_value_pre(epi, m, n, acc[ni][mi], pbias[ni], pres[ni][mi]));
                    if (gfast) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float f = bf16_to_f32(o.v[j]); cs[j] += f; cq[j] += f * f; }
                    } else {
                        *(U16x4*)(gtile + (wm * (TM / WM) + mi * 16 + frow) * GTS + wn * (TN / WN) + ni * 16 + 4 * fq) = o;
                    }
                }
            }
            if (gfast && wrow0 < M) gn_wave_stats(cs, cq, frow, fq, wn * (TN / WN) + ni * 16, vcols, gn_cg, gn_coff + n0, wseg, g_ngl, gacc);
        }
        if (gfast) {
            __syncthreads();
            gn_acc_flush(gacc, g_nseg, g_ngl, g_b_first, g_first, tid, 64 * NW, gn_stat);
        } else {
            gn_tile_stats(gtile, GTS, TM, TN, 64 * NW, tid, min(TM, M - m0), vcols, gn_hw, m0 % gn_hw, m0 / gn_hw, gn_cg, gn_coff + n0,
                          (long long*)(smem + GTILE), (STAGES * STAGE + GTAIL - GTILE) / 16, gn_stat);
        }
        return;
    }
    // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of every (ni, mi) fragment ---------------
    // Fast form (gemm_device.h: epilogue_fast_store) whenever the launch asks for nothing but bias / row bias / scale / residual
    // -> bf16: wave-uniform decision (the row bias needs all rows of the wave in one sample), same arithmetic as the general form.
    if (pre && !stat_out && epi.act == 0 && !epi.out_f32) {
        const int wr0 = m0 + wm * (TM / WM);
        if (wr0 >= M) return;                      // this wave owns no row (ragged last tile): nothing to store, no row bias to fetch
        int rb_b = -1;
        bool ok = true;
        if (epi.rowbias) {
            const int wr1 = min(wr0 + TM / WM, M) - 1;
            rb_b = wr0 / epi.rpb;
            ok = rb_b == wr1 / epi.rpb;
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
                const int m = wr0 + mi * 16 + frow;
                if (m >= M) continue;
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
    float* const statlds = (float*)(smem + STAGES * STAGE);        // [WN][TM][2] scratch behind the ring
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * (TM / WM) + mi * 16 + frow;
        const bool mok = m < M;
        float mu = 0.f, rstd = 1.f;
        if (LN > 0) ln_stats_finish<SL>(lnt[mi], stat_in_slots, fq, K, ln_eps, mu, rstd);
        if (LNF) {
            const int lr = wm * (TM / WM) + mi * 16 + frow;
            const float q = lnrow[lr * 2 + 1];
            const float mean = lnrow[lr * 2] / (float)K;
            float var = q / (float)K - mean * mean;
            var = var < 0.f ? 0.f : var;
            mu = mean; rstd = rsqrtf(var + ln_eps);
        }
        float ps = 0.f, pq = 0.f;          // partial row sum / sum of squares of this lane's output columns
        if (mok) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = n0 + wn * (TN / WN) + ni * 16 + 4 * fq;
                if (n >= N) continue;
                if (splitk > 1) {
                    *(f32x4*)(ws + ((size_t)bz * M + m) * N + n) = acc[ni][mi];
                } else {
                    const f32x4 v0 = LN ? ln_correct(ln_s, n, acc[ni][mi], mu, rstd) : acc[ni][mi];
                    const f32x4 r = epilogue_write(epi, m, n, pre ? epilogue_value_pre(epi, m, n, v0, pbias[ni], pres[ni][mi]) : epilogue_value(epi, m, n, v0));
                    ps += (r[0] + r[1]) + (r[2] + r[3]);
                    pq += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                }
            }
        }
        if (stat_out && splitk == 1) {     // (uniform branch) per-row partial of this wave's columns -> LDS
            ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
            pq += __shfl_xor(pq, 16, 64); pq += __shfl_xor(pq, 32, 64);
            const int lr = wm * (TM / WM) + mi * 16 + frow;
            if (fq == 0) { statlds[(wn * TM + lr) * 2] = ps; statlds[(wn * TM + lr) * 2 + 1] = pq; }
        }
    }
    if (stat_out && splitk == 1) {         // one slot per column tile: wave columns summed in a fixed order
        __syncthreads();
        for (int lr = tid; lr < TM; lr += 64 * NW) {
            const int m = m0 + lr;
            if (m >= M) continue;
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int c = 0; c < WN; ++c) { a += statlds[(c * TM + lr) * 2]; q += statlds[(c * TM + lr) * 2 + 1]; }
            *(float2*)(stat_out + ((size_t)by * M + m) * 2) = float2{a, q};
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// "Register-A" implicit GEMM (round 3): the ACTIVATION operand never touches LDS, and nothing goes through LDS-DMA.
//
// gemm_kernel stages both operand tiles through LDS with global_load_lds, and a wave's LDS-DMA transfers complete one after the
// other (~140-370 cycles per 1 KiB piece, tools/micro/stream_rate3.hip): a K-step of a 128x128 tile asks 8 pieces of
// every wave - ~1500 cycles for 512 cycles of MFMA work.  Here the NW waves of a workgroup split the ROWS of the tile only (wave w
// owns rows [w TM/NW, +TM/NW) and all TN columns), so an activation fragment belongs to exactly one wave: it is loaded straight
// from L2 / Infinity Cache into the registers the MFMA reads, in fragment layout (lane (row r, quarter q) of a 16 x 32 fragment
// holds 8 consecutive k of row r: one global_load_dwordx4, 64 B contiguous per row and instruction), STAGES - 1 K-steps ahead, with
// plain loads that pipeline.  The WEIGHT tile, shared by all waves (TN x 64 bf16 = 8-20 KiB per K-step, TN/8/NW 1 KiB pieces per
// wave), is loaded the same way - registers first, coalesced 128 B rows - and written to a double-buffered LDS tile one K-step ahead
// of its use (ds_write_b128 under the previous step's MFMAs), XOR-swizzled as in gemm_kernel.
// All loads are ORDINARY loads that the compiler sees: its own counted s_waitcnt vmcnt (exact for in-order VGPR loads) guards every
// use of a prefetched register - including the register COPIES it places at control-flow joins, which is what broke two earlier
// forms of this kernel that issued the loads as inline asm with hand-counted waits (a copy of a register whose load is still in
// flight reads stale bits, and nothing tells the compiler that the load is pending: sporadic NaNs, tools/dbg_ra.py).
// One s_barrier per K-step.  Epilogue as gemm_kernel's plain forms (bias, per-sample row bias, scale, residual, SiLU / quick-GELU /
// GEGLU, fp32 output, split-K slabs); fused LayerNorm / statistics variants stay on gemm_kernel (kTileBase).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// (explicitly GLOBAL: a pointer that went through select_src's opaque integer select would otherwise be loaded with flat_load, which
// counts in lgkmcnt as well and turns every wait into a drain)
__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
    dst = *(const __attribute__((address_space(1))) u32x4*)(unsigned long long)p;
}
template <int TM, int TN, int NW, int CONV, int STAGES>
__global__ __launch_bounds__(64 * NW) void gemm_ra_kernel(const GemmArgs2 pg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x [TN][64] bf16: W tile of the current and of the next K-step
    const int grp = (int)blockIdx.z >= pg.g[0].gz ? 1 : 0;
    const GemmArgs& p = pg.g[grp];
    constexpr int WSB = TN * 128;         // bytes of one W tile (TN rows x 64 bf16)
    constexpr int WP = TN / 8 / NW;       // W pieces (8 rows x 128 B) per wave and K-step
    constexpr int RW = TM / NW;           // rows of a wave
    constexpr int MI = RW / 16;           // A fragments of a wave
    constexpr int NI = TN / 16;           // W fragments (every wave reads all of them)
    constexpr int LPT = WP + 2 * MI;      // loads per wave per K-step (exact)
    static_assert((WP == 1 || WP == 2 || WP == 4 || WP == 5) && (MI == 2 || MI == 4) && NI >= 1 && (NW == 4 || NW == 8) && STAGES >= 3 && STAGES <= 4, "tile / wave layout");
    static_assert((STAGES - 1) * LPT < 64, "vmcnt: every prefetched load must be countable");

    const bf16_t* const gA = p.A; const bf16_t* const gW = p.W; const bf16_t* const gZ = p.zero;
    const int lda = p.lda, ldw = p.ldw, M = p.M, N = p.N, K = p.K;
    const int Hin = p.Hin, Win = p.Win, Cin = p.Cin, Hout = p.Hout, Wout = p.Wout, cstride = p.stride, up = p.up;
    const int splitk = p.splitk, per = p.ksteps_per_split;
    float* const ws = p.ws;
    const Epilogue epi = make_epilogue(p);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bx, by, bz;
    xcd_tile_order(p.xcd_mode, p.gz, grp, bx, by, bz);
    const int m0 = bx * TM, n0 = by * TN;
    const int nk_total = (K + BK - 1) / BK;
    const int kt_begin = bz * per;
    const int kt_end = min(nk_total, kt_begin + per);

    // ---- W pieces: lane (lrow = lane >> 3, chunk c = lane & 7) of piece i loads 16 B of tile row 8 (w + NW i) + lrow: 128 B per row,
    //      coalesced; it lands in LDS at row * 128 + ((c ^ key) << 4), key = (row >> 1) & 7 (the swizzle gemm_kernel reads with) ----
    const int lrow = lane >> 3, lc = lane & 7;
    const int key = (4 * (w & 1) + (lane >> 4)) & 7;          // == ((tile_row >> 1) & 7) for every piece of this wave (NW even)
    size_t woff[WP];
    bool wok[WP];
    int wlds[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = 8 * (w + NW * i) + lrow;
        const int n = n0 + r;
        wok[i] = n < N;
        woff[i] = (size_t)n * ldw + lc * 8;
        wlds[i] = r * 128 + ((lc ^ key) << 4);
    }
    // ---- A fragments: lane (frow, fq) of fragment mi holds row m0 + w RW + 16 mi + frow, k = 64 kt + 32 kk + 8 fq .. +8 ----
    const int frow = lane & 15, fq = lane >> 4;
    size_t aoff[MI];               // LINEAR: element offset of the row (+ the lane's k quarter);  CONV: pixel-index base of the sample
    int uy0[MI], ux0[MI];
    bool aok[MI];
    const int Hup = Hin << up, Wup = Win << up;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + w * RW + mi * 16 + frow;
        aok[mi] = m < M;
        if (CONV) {
            const int hw = Hout * Wout;
            const int mm = aok[mi] ? m : 0;
            const int b = mm / hw;
            const int rem = mm - b * hw;
            const int oy = rem / Wout;
            const int ox = rem - oy * Wout;
            aoff[mi] = (size_t)b * Hin * Win;
            uy0[mi] = oy * cstride - 1;
            ux0[mi] = ox * cstride - 1;
        } else {
            aoff[mi] = (size_t)m * lda + fq * 8;
            uy0[mi] = ux0[mi] = 0;
        }
    }

    u32x4 wr[STAGES][WP];          // W pieces of up to STAGES K-steps in flight (slot indices are compile-time after unrolling)
    u32x4 xr[STAGES][MI][2];       // A fragments of the same K-steps
#pragma unroll
    for (int s = 0; s < STAGES; ++s) {
#pragma unroll
        for (int i = 0; i < WP; ++i) wr[s][i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { xr[s][mi][0] = u32x4{0u, 0u, 0u, 0u}; xr[s][mi][1] = u32x4{0u, 0u, 0u, 0u}; }
    }
    // ALWAYS exactly LPT loads (W pieces first), also for K-steps past this block's range (zero page): the number of loads in flight
    // behind any K-step is then the same on every path, which is what lets the compiler's s_waitcnt vmcnt be a counted one
    // instead of a drain (a wait after a join must hold for the path with the FEWEST younger loads)
    auto issue = [&](int slot, int kt) {
        const int kw = kt < kt_end ? kt * BK : K;          // (K: every chunk out of range -> zero page)
#pragma unroll
        for (int i = 0; i < WP; ++i) gload16(wr[slot][i], select_src(gW + woff[i] + kw, gZ, wok[i] && kw + lc * 8 < K));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = kw + 32 * kk + 8 * fq;
            int ci = 0, ky = 0, kx = 0;
            if (CONV) {
                const int tap = k / Cin;
                ci = k - tap * Cin;
                ky = (tap * 11) >> 5;
                kx = tap - 3 * ky;
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const bf16_t* real;
                bool ok;
                if (CONV) {
                    const int uy = uy0[mi] + ky, ux = ux0[mi] + kx;
                    ok = k < K && aok[mi] && (unsigned)uy < (unsigned)Hup && (unsigned)ux < (unsigned)Wup;
                    real = gA + ((aoff[mi] + (size_t)((uy >> up) * Win + (ux >> up))) * lda + ci);
                } else {
                    ok = k < K && aok[mi];
                    real = gA + aoff[mi] + kw + 32 * kk;
                }
                gload16(xr[slot][mi][kk], select_src(real, gZ, ok));
            }
        }
    };
    // the W pieces of the K-step in `slot` into an LDS tile (the compiler's own counted s_waitcnt vmcnt precedes the first use of wr)
    auto land = [&](int slot, char* wbuf) {
#pragma unroll
        for (int i = 0; i < WP; ++i) *(u32x4*)(wbuf + wlds[i]) = wr[slot][i];
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int slot, const char* wsm) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = 4 * kk + fq;
            bf16x8 wf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = ni * 16 + frow;
                wf[ni] = *(const bf16x8*)(wsm + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], __builtin_bit_cast(bf16x8, xr[slot][mi][kk]), acc[ni][mi], 0, 0, 0);
        }
    };

    const int ntile = kt_end - kt_begin;
    if (ntile > 0) {
        // K-steps 0 .. STAGES-2 in flight, K-step 0 landed and in LDS buffer 0.  The loop body is STAGES K-steps without a branch
        // (register slots are compile-time); K-steps past the end multiply zeros (at most STAGES - 1 of them per workgroup).
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) issue(s, kt_begin + s);
        asm volatile("" ::: "memory");
        land(0, smem);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int i0 = 0; i0 < ntile; i0 += STAGES) {
#pragma unroll
            for (int j = 0; j < STAGES; ++j) {
                const int i = i0 + j;
                char* const cur = smem + (i & 1) * WSB;
                char* const nxt = smem + ((i + 1) & 1) * WSB;
                issue((j + STAGES - 1) % STAGES, kt_begin + i + STAGES - 1);      // (slot of K-step i - 1: consumed)
                asm volatile("" ::: "memory");                 // (the prefetch stays ahead of this step's LDS traffic and MFMAs)
                // K-step i + 1 into the other LDS buffer (read last during K-step i - 1: every wave is past that barrier), under this step's MFMAs
                land((j + 1) % STAGES, nxt);
                compute(j, cur);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // next W tile visible to all waves; all reads of the current one done
                asm volatile("" ::: "memory");
            }
        }
    }

    // epilogue operands (after the K loop: ordinary compiler-managed loads must not sit among the counted ones)
    const bool pre = splitk == 1;
    f32x4 pbias[NI];
    U16x4 pres[NI][MI];
    if (pre) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 16 + 4 * fq;
            pbias[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi.bias && n < N) pbias[ni] = *(const f32x4*)(epi.bias + n);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + w * RW + mi * 16 + frow;
#pragma unroll
                for (int j = 0; j < 4; ++j) pres[ni][mi].v[j] = 0;
                if (epi.R && n < N && m < M) pres[ni][mi] = *(const U16x4*)(epi.R + (size_t)m * epi.ldr + n);
            }
        }
    }

    // ---- epilogue: lane holds D[n = 4 fq + r][m = frow] of every (ni, mi) fragment (as gemm_kernel) ----
    const int wr0 = m0 + w * RW;
    if (pre && epi.act == 0 && !epi.out_f32) {
        if (wr0 >= M) return;
        int rb_b = -1;
        bool ok = true;
        if (epi.rowbias) {
            const int wr1 = min(wr0 + RW, M) - 1;
            rb_b = wr0 / epi.rpb;
            ok = rb_b == wr1 / epi.rpb;
        }
        if (ok) {
            f32x4 prb[NI];
            if (rb_b >= 0) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + ni * 16 + 4 * fq;
                    prb[ni] = n < N ? *(const f32x4*)(epi.rowbias + (size_t)rb_b * epi.ldrb + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = wr0 + mi * 16 + frow;
                if (m >= M) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + ni * 16 + 4 * fq;
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
        const int m = wr0 + mi * 16 + frow;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 16 + 4 * fq;
            if (n >= N) continue;
            if (splitk > 1) *(f32x4*)(ws + ((size_t)bz * M + m) * N + n) = acc[ni][mi];
            else epilogue_write(epi, m, n, epilogue_value_pre(epi, m, n, acc[ni][mi], pbias[ni], pres[ni][mi]));
        }
    }
}

// split-K reduce + epilogue: thread = (row m, 4 columns); slabs summed 4 at a time with independent loads.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs2 pg) {
    const GemmArgs& p = pg.g[blockIdx.z];      // grouped launch: z selects the problem
    const int n = (blockIdx.x * 64 + (threadIdx.x & 63)) << 2;
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool ok = n < p.N && m < p.M;
    float ps = 0.f, pq = 0.f;
    if (ok) {
        const Epilogue e = make_epilogue(p);
        // bias / residual first, so that their latency overlaps the slab loads instead of following them
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        U16x4 r4;
#pragma unroll
        for (int j = 0; j < 4; ++j) r4.v[j] = 0;
        if (e.bias) b4 = *(const f32x4*)(e.bias + n);
        if (e.R) r4 = *(const U16x4*)(e.R + (size_t)m * e.ldr + n);
        const size_t slab = (size_t)p.M * p.N;
        const float* src = p.ws + (size_t)m * p.N + n;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= p.splitk; z += 4) {
            const f32x4 a = *(const f32x4*)(src + (size_t)z * slab);
            const f32x4 b = *(const f32x4*)(src + (size_t)(z + 1) * slab);
            const f32x4 c = *(const f32x4*)(src + (size_t)(z + 2) * slab);
            const f32x4 d = *(const f32x4*)(src + (size_t)(z + 3) * slab);
            v += (a + b) + (c + d);
        }
        for (; z < p.splitk; ++z) v += *(const f32x4*)(src + (size_t)z * slab);
        const f32x4 r = epilogue_write(e, m, n, epilogue_value_pre(e, m, n, v, b4, r4));
        ps = (r[0] + r[1]) + (r[2] + r[3]);
        pq = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
    }
    if (p.stat_out) {                      // one slot per 256-column chunk (= blockIdx.x); the wave owns one row
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ps += __shfl_xor(ps, o, 64); pq += __shfl_xor(pq, o, 64); }
        if ((threadIdx.x & 63) == 0 && m < p.M) *(float2*)(p.stat_out + ((size_t)blockIdx.x * p.M + m) * 2) = float2{ps, pq};
    }
}

// split-K reduce + epilogue that also emits the GroupNorm statistics of its output: same mapping as splitk_epilogue_kernel
// (4 rows x 256 columns per block, a wave owns one row, a lane 4 columns); the lane's 8 (sum, sumsq) values go to the block's
// (sample, group) accumulator in LDS as fixed-point integers (order-free), then one device-scope atomic per entry.
__global__ __launch_bounds__(256) void splitk_epilogue_gn_kernel(const GemmArgs2 pg) {
    const GemmArgs& p = pg.g[blockIdx.z];
    constexpr int CAP = 512;
    __shared__ long long acc[CAP * 2];
    const int tid = threadIdx.x;
    const int n = (blockIdx.x * 64 + (tid & 63)) << 2;
    const int mb = blockIdx.y * 4;
    const int m = mb + (tid >> 6);
    const int hw = p.gn_hw, cg = p.gn_cg;
    const int rows = min(4, p.M - mb);
    const int b_first = mb / hw;
    const int nseg = (mb + rows - 1) / hw - b_first + 1;
    const int c0 = p.gn_coff + blockIdx.x * 256;                      // consumer-tensor column of this block's first column
    const int ncol = min(256, p.N - blockIdx.x * 256);
    const int g_first = c0 / cg;
    const int ngl = (c0 + ncol - 1) / cg - g_first + 1;
    const bool use_lds = nseg * ngl <= CAP;
    if (use_lds)
        for (int i = tid; i < nseg * ngl * 2; i += 256) acc[i] = 0;
    __syncthreads();
    if (n < p.N && m < p.M) {
        const Epilogue e = make_epilogue(p);
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        U16x4 r4;
#pragma unroll
        for (int j = 0; j < 4; ++j) r4.v[j] = 0;
        if (e.bias) b4 = *(const f32x4*)(e.bias + n);
        if (e.R) r4 = *(const U16x4*)(e.R + (size_t)m * e.ldr + n);
        const size_t slab = (size_t)p.M * p.N;
        const float* src = p.ws + (size_t)m * p.N + n;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= p.splitk; z += 4) {
            const f32x4 a = *(const f32x4*)(src + (size_t)z * slab);
            const f32x4 b = *(const f32x4*)(src + (size_t)(z + 1) * slab);
            const f32x4 c = *(const f32x4*)(src + (size_t)(z + 2) * slab);
            const f32x4 d = *(const f32x4*)(src + (size_t)(z + 3) * slab);
            v += (a + b) + (c + d);
        }
        for (; z < p.splitk; ++z) v += *(const f32x4*)(src + (size_t)z * slab);
        const U16x4 o = epilogue_write_bits(e, m, n, epilogue_value_pre(e, m, n, v, b4, r4));
        const int b = m / hw;
        // channels of one lane fall into at most two groups when cg >= 4: merge what shares a group before the atomics
        int gl = (p.gn_coff + n) / cg - g_first, left = cg - (p.gn_coff + n - (g_first + gl) * cg);
        float sm = 0.f, sq = 0.f;
        auto flush = [&]() {
            const long long a = __float2ll_rn(sm * GN_FIX_SUM), q = __float2ll_rn(sq * GN_FIX_SQ);
            long long* dst = use_lds ? acc + (size_t)((b - b_first) * ngl + gl) * 2 : p.gn_stat + ((size_t)b * GN_GROUPS + g_first + gl) * 2;
            if (a) gn_atomic_add(dst, a);
            if (q) gn_atomic_add(dst + 1, q);
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float f = bf16_to_f32(o.v[j]);
            sm += f; sq += f * f;
            if (--left == 0) { flush(); sm = 0.f; sq = 0.f; ++gl; left = cg; }
        }
        if (left != cg) flush();
    }
    if (use_lds) {
        __syncthreads();
        gn_acc_flush(acc, nseg, ngl, b_first, g_first, tid, 256, p.gn_stat);
    }
}

}  // namespace

// tile configurations: index -> (TM, TN, waves m x n, stages)
//   0: 256x128 8 waves (4x2) 3 stages  144 KiB LDS   1 block/CU   85 FLOP per staged byte
//   1: 128x128 4 waves (2x2) 3 stages   96 KiB        1 block/CU   64
//   2: 128x128 4 waves (2x2) 2 stages   64 KiB        2 blocks/CU  64
//   3: 128x64  4 waves (2x2) 3 stages   72 KiB        2 blocks/CU  43
//   4: 64x128  4 waves (2x2) 3 stages   72 KiB        2 blocks/CU  43
//   5: 64x64   4 waves (2x2) 4 stages   64 KiB        2 blocks/CU  32
//   6..11: LDS-staged 3x3 conv tiles (kernels_conv.hip): 256x128, 256x64 (8 waves), 128x128, 128x64, 64x128, 64x64
//   12: 64x64 6 stages (96 KiB), 13: 64x128 5 stages (120 KiB): short-K layers, (nearly) every K-step in flight at once
//   14: 64x160 3 stages (86 KiB), 15: 128x160 3 stages (110 KiB), 16: 64x160 2 stages (57 KiB, 2 blocks/CU): every channel
//       count of the nets is a multiple of 320, so 160-wide column tiles never run a partly empty tile (N = 320 -> 2 x 160
//       instead of 3 x 128 with 17 % of the MFMA work wasted)
//   17: 32x64, 18: 64x32, 19: 32x32 (4 stages): one CU streams at most ~55 GB/s (tools/micro/stream_rate.hip), so a GEMM with
//       fewer blocks than CUs finishes sooner when each block pulls FEWER operand bytes ((TM + TN) * K * 2), not more
//   20..29: in-block K split (KW groups of 4 waves, see gemm_kernel): 32x32 x2 / x4, 64x32 x2 / x4, 64x64 x2 / x4, 32x64 x2 / x4,
//       128x64 x2, 64x128 x2
//   30..37: more waves per CU pulling operands.  A 4-wave workgroup streams ~48 GB/s whatever its ring depth (2, 4 or 8 stages), a CU
//       with 8 waves ~94 GB/s, with 16 waves ~122 GB/s (tools/micro/stream_rate2.hip): the limit is per WAVE.  So: the same tiles with
//       2-stage rings (half the LDS -> twice the resident workgroups): 64x64, 128x64, 64x128, 64x32; and 8-wave workgroups:
//       128x128 (2 stages), 128x64, 64x128 (3), 64x64 (4).  Plain epilogue only (anything else runs on the base configuration).
//   38..40: LDS-staged 3x3 conv tiles with EIGHT waves (kernels_conv.hip): 128x64, 64x128, 128x128; 41: 256x64 with 8 waves (gather /
//       linear, plain epilogue only); 42 / 43: LDS-staged 256x128 / 128x128 with SIXTEEN waves.  Inside the sampling loop the whole-loop tuner (tools/tune_wall.py) moves the heavy shapes onto the
//       eight-wave tiles although they are not faster alone: half the LDS-DMA pieces per wave and K-step.
//   44..49: register-A tiles (gemm_ra_kernel: activations straight into the MFMA's registers, only the weight tile through LDS):
//       256x64 and 256x128 with 8 waves (32 rows each), 128x128 / 128x64 / 128x160 with 4 waves, 256x64 with 4 waves (64 rows each).
//   50: 256x256 with SIXTEEN waves (64x64 each), 2 stages (128 KiB LDS): 128 FLOP per byte pulled out of L2, for the few GEMMs whose M and N
//       both allow it (the K loops of the gather / linear kernel run at three quarters of the L2 -> CU rate: DESIGN.md 4.5); plain epilogue only.
constexpr int N_TILE_CFG = 51;
static const int kTileM[N_TILE_CFG] = {256, 128, 128, 128, 64, 64, 256, 256, 128, 128, 64, 64, 64, 64, 64, 128, 64, 32, 64, 32,
                                       32, 32, 64, 64, 64, 64, 32, 32, 128, 64, 64, 128, 64, 64, 128, 128, 64, 64, 128, 64, 128, 256, 256, 128,
                                       256, 256, 128, 128, 128, 256, 256};
static const int kTileN[N_TILE_CFG] = {128, 128, 128, 64, 128, 64, 128, 64, 128, 64, 128, 64, 64, 128, 160, 160, 160, 64, 32, 32,
                                       32, 32, 32, 32, 64, 64, 64, 64, 64, 128, 64, 64, 128, 32, 128, 64, 128, 64, 64, 128, 128, 64, 128, 128,
                                       64, 128, 128, 64, 160, 64, 256};
static const int kTileKW[N_TILE_CFG] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 4, 2, 4, 2, 4, 2, 4, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1};
static const int kTileLight[N_TILE_CFG] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0,
                                           1, 1, 1, 1, 1, 1, 1};
static const int kTileBase[N_TILE_CFG] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 19, 19, 18, 18, 5, 5, 17, 17, 3, 4,
                                          5, 3, 4, 18, 2, 3, 4, 5, 38, 39, 40, 3, 42, 43,
                                          3, 1, 1, 3, 14, 3, 0};
static const char* const kTileName[N_TILE_CFG] = {"256x128", "128x128_s3", "128x128_s2", "128x64", "64x128", "64x64",
                                                  "patch256x128", "patch256x64", "patch128x128", "patch128x64",
                                                  "patch64x128", "patch64x64", "64x64_s6", "64x128_s5", "64x160", "128x160", "64x160_s2", "32x64", "64x32", "32x32",
                                                  "32x32_k2", "32x32_k4", "64x32_k2", "64x32_k4", "64x64_k2", "64x64_k4", "32x64_k2", "32x64_k4", "128x64_k2", "64x128_k2",
                                                  "64x64_s2", "128x64_s2", "64x128_s2", "64x32_s2", "128x128_w8", "128x64_w8", "64x128_w8", "64x64_w8",
                                                  "patch128x64_w8", "patch64x128_w8", "patch128x128_w8", "256x64_w8", "patch256x128_w16", "patch128x128_w16",
                                                  "ra256x64_w8", "ra256x128_w8", "ra128x128", "ra128x64", "ra128x160", "ra256x64", "256x256_w16"};
static bool is_patch_cfg(int c) { return (c >= 6 && c <= 11) || (c >= 38 && c <= 40) || c == 42 || c == 43; }
bool gemm_cfg_folds_second_input(int c) { return c >= 0 && c < N_TILE_CFG && !is_patch_cfg(c) && !(c >= 44 && c <= 49); }      // gemm_kernel (GemmArgs::A2)
int gemm_num_tile_cfgs() { return N_TILE_CFG; }
const char* gemm_tile_cfg_name(int cfg) { return (cfg >= 0 && cfg < N_TILE_CFG) ? kTileName[cfg] : "?"; }

struct GemmPlan { int cfg, splitk, per; };

struct TunedEntry { int M, N, K, conv, stride, up, cfg, splitk; };
static const TunedEntry kTuned[] = {
#include "gemm_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0}};

static const TunedEntry* tuned_lookup(int M, int N, int K, int conv, int stride, int up) {
    static const bool no_table = getenv("MKD_NO_TABLE") && atoi(getenv("MKD_NO_TABLE")) != 0;      // experiments: heuristic plans only
    if (no_table) return nullptr;
    for (const TunedEntry* e = kTuned; e->M; ++e)
        if (e->M == M && e->N == N && e->K == K && e->conv == conv && e->stride == stride && e->up == up) return e;
    return nullptr;
}

// run-time per-shape overrides (in-eval tuner, tools/tune_ineval.py): consulted before the compiled table
struct ShapeKey { int M, N, K, conv, stride, up; bool operator<(const ShapeKey& o) const {
    return std::tie(M, N, K, conv, stride, up) < std::tie(o.M, o.N, o.K, o.conv, o.stride, o.up); } };
static std::map<ShapeKey, TunedEntry> g_override;
static int g_plan_epoch = 0;
int gemm_plan_epoch() { return g_plan_epoch; }
void gemm_set_override(int M, int N, int K, int conv, int stride, int up, int cfg, int splitk) {
    ++g_plan_epoch;
    if (M <= 0) { g_override.clear(); return; }
    const ShapeKey k{M, N, K, conv, stride, up};
    if (cfg < 0 || cfg >= N_TILE_CFG) { g_override.erase(k); return; }
    TunedEntry e{}; e.M = M; e.N = N; e.K = K; e.conv = conv; e.stride = stride; e.up = up; e.cfg = cfg; e.splitk = splitk < 1 ? 1 : splitk;
    g_override[k] = e;
}

static int g_force_cfg = -1;     // tuner / tests only
static int g_splitk_cap = 0;      // experiment knob (MKD_SPLITK_CAP): 0 = no cap
void gemm_set_splitk_cap(int cap) { if (cap != g_splitk_cap) ++g_plan_epoch; g_splitk_cap = cap; }      // (plans hold split-K decisions: re-build)
// XCD-aware tile order of every launch that does not ask for one itself (MKD_XCD_MODE / mkd_gemm_set_xcd_mode; GemmArgs::xcd_mode).
// Default 0 = launch order: measured over the whole loop both contiguous-run orders lose (batch 8: 27.61 / 27.28 / 27.51 images/s for
// 0 / 1 / 2, batch 1: 6.78 / 6.63 / 6.63, 512x512: 9.22 / 9.07 / 9.05; tools/exp_xcd.sh) - with M-tiles fastest and a multiple of 8 of
// them the launch order already keeps an A tile in one L2, and the weight tiles are small next to 8 x 4 MB of L2.
static int g_xcd_mode = getenv("MKD_XCD_MODE") ? atoi(getenv("MKD_XCD_MODE")) : 0;
void gemm_set_xcd_mode(int mode) { g_xcd_mode = (mode >= 0 && mode <= 2) ? mode : 0; }
void gemm_force_tile_cfg(int cfg) {
    const int c = (cfg >= 0 && cfg < N_TILE_CFG) ? cfg : -1;
    if (c != g_force_cfg) ++g_plan_epoch;       // launch plans hold decisions taken with the previous setting (slab counts of deferred epilogues): re-build
    g_force_cfg = c;
}

// Tile + split-K choice.  Large problems take the big tiles (more FLOP per byte staged through L2 -> LDS,
// which is what bounds these kernels); problems that cannot fill the 256 CUs step down to smaller tiles
// and only then split K (fp32 partial slabs cost 8 B per output element per slice).
// pin_cfg >= 0: that tile configuration, split-K by the heuristic, tuned table / overrides / g_force_cfg ignored
static GemmPlan gemm_plan(int M, int N, int K, int force_splitk, int conv = 0, int stride = 0, int up = 0, int pin_cfg = -1, int K2 = 0) {
    const int nk = (K + BK - 1) / BK;
    const int KL = K - K2;          // the tables know the plain convolution (GemmArgs::K2)
    auto tiles = [&](int c) { return ((M + kTileM[c] - 1) / kTileM[c]) * ((N + kTileN[c] - 1) / kTileN[c]); };
    int cfg;
    const int g_force_cfg = pin_cfg >= 0 ? pin_cfg : ::g_force_cfg;          // (shadows the global on purpose)
    const TunedEntry* te = (g_force_cfg < 0 && force_splitk <= 0) ? tuned_lookup(M, N, KL, conv, stride, up) : nullptr;
    if (g_force_cfg < 0 && force_splitk <= 0 && !g_override.empty()) {
        auto it = g_override.find(ShapeKey{M, N, KL, conv, stride, up});
        if (it != g_override.end()) te = &it->second;
    }
    if (te) {
        GemmPlan g;
        g.cfg = te->cfg;
        const int units = is_patch_cfg(te->cfg) ? (K / 9) / BK : nk;     // patch conv splits over channel chunks
        int s0 = te->splitk < units ? te->splitk : units;
        if (g_splitk_cap > 0 && s0 > g_splitk_cap) s0 = g_splitk_cap;
        g.per = (units + s0 - 1) / s0;
        g.splitk = (units + g.per - 1) / g.per;
        return g;
    }
    if (is_patch_cfg(g_force_cfg)) {
        GemmPlan g;
        g.cfg = g_force_cfg;
        const int units = (K / 9) / BK > 0 ? (K / 9) / BK : 1;
        int s0 = force_splitk > 0 ? force_splitk : 1;
        if (s0 > units) s0 = units;
        g.per = (units + s0 - 1) / s0;
        g.splitk = (units + g.per - 1) / g.per;
        return g;
    }
    if (g_force_cfg >= 0) cfg = g_force_cfg;
    else {
        const bool n128 = (N % 128 == 0);
        const int want = 224;
        if (n128 && tiles(1) >= want) cfg = 1;
        else if (tiles(3) >= want || (!n128 && M > 64)) cfg = (n128 || tiles(3) >= want) ? 3 : 3;
        else cfg = 5;
        if (!n128 && cfg == 1) cfg = 3;
        if (cfg == 3 && tiles(3) < want && n128) cfg = 5;
        if (M <= 64) cfg = 5;
        // shapes the tuned table does not know: the eight-wave siblings of the middle tiles (the whole-loop tuner moved most heavy
        // shapes onto eight-wave tiles: DESIGN.md 4.4); launches that need more than the plain epilogue fall back to kTileBase
        static const bool heur_w8 = getenv("MKD_HEUR_W8") ? atoi(getenv("MKD_HEUR_W8")) != 0 : true;      // (batch 6: +4.7 %, batch 12: +10.7 %, batch 3: +1.4 % images/s)
        if (heur_w8 && M >= 256) cfg = cfg == 1 ? 34 : (cfg == 3 ? 35 : (cfg == 5 ? 37 : cfg));
    }
    int s = 1;
    if (force_splitk > 0) s = force_splitk;
    else {
        const int t = tiles(cfg);
        if (t < 160 && nk >= 8) {
            s = (256 + t - 1) / t;
            if (s > nk / 4) s = nk / 4;      // >= 4 K-steps per slice
            if (s > 16) s = 16;
            if (s < 1) s = 1;
        }
    }
    if (force_splitk <= 0 && g_splitk_cap > 0 && s > g_splitk_cap) s = g_splitk_cap;      // (the experiment cap covers heuristic plans too)
    if (s > nk) s = nk;
    GemmPlan g;
    g.cfg = cfg;
    g.per = (nk + s - 1) / s;
    g.splitk = (nk + g.per - 1) / g.per;
    return g;
}

int gemm_pick_splitk(int M, int N, int K, int conv, int stride, int up) { return gemm_plan(M, N, K, 0, conv, stride, up).splitk; }

// producers of fused-LayerNorm row statistics keep column tiles >= 64 wide (one statistics slot per column tile, <= 20 slots)
static GemmPlan stat_producer_plan(GemmPlan g) {
    if (kTileN[g.cfg] < 64) g.cfg = 5;
    return g;
}

// The ONE place that decides (tile, split-K) for a launch, with the full geometry: the engine sizes its split-K workspace from
// this at plan time and launch_gemm takes the same decision at launch time.  A tuned LDS-patch entry whose spatial tiling does
// not fit THIS geometry (the table is keyed on M, N, K only) falls back to the generic 128x128 tile with heuristic split-K.
static int gemm_resolve_plan(const GemmArgs& a, GemmPlan* out) {
    GemmPlan g = gemm_plan(a.M, a.N, a.K, a.splitk, a.conv, a.conv ? a.stride : 0, a.conv ? a.up : 0, -1, a.A2 ? a.K2 : 0);
    if (a.A2 && !gemm_cfg_folds_second_input(g.cfg))
        return mkd_fail(-4, "gemm: a convolution with a folded second input runs on the gather kernel only (the plan of this shape is LDS-staged or register-A)");
    if (a.ln_s && g.splitk > 1) {          // the LN correction is applied on the full-K accumulator
        g.splitk = 1;
        g.per = (a.K + BK - 1) / BK;
    }
    if (a.stat_out) g = stat_producer_plan(g);
    if (kTileKW[g.cfg] > 1 && ((a.ln_s && a.stat_in) || a.stat_out || (a.gn_stat && g.splitk == 1))) g.cfg = kTileBase[g.cfg];     // plain epilogue (or on-the-fly LayerNorm) only
    if (kTileLight[g.cfg] && (a.ln_s || a.stat_out || (a.gn_stat && g.splitk == 1))) g.cfg = kTileBase[g.cfg];                     // plain epilogue only
    if (is_patch_cfg(g.cfg) && !conv_patch_supported(a, g.cfg)) {
        if (is_patch_cfg(g_force_cfg)) return mkd_fail(-4, "gemm: forced LDS-staged conv tile does not fit this shape");
        g = gemm_plan(a.M, a.N, a.K, 0, 0, 0, 0, /*pin_cfg=*/1);
    }
    *out = g;
    return 0;
}
int gemm_resolve(const GemmArgs& a, int* cfg, int* splitk) {
    GemmPlan g;
    const int rc = gemm_resolve_plan(a, &g);
    if (rc) return rc;
    if (is_patch_cfg(g.cfg)) {             // the patch launcher splits over channel chunks and re-derives the count
        const int nch = a.Cin / 64;
        int s = g.splitk < 1 ? 1 : g.splitk;
        if (s > nch) s = nch;
        const int per = (nch + s - 1) / s;
        g.splitk = (nch + per - 1) / per;
    }
    *cfg = g.cfg; *splitk = g.splitk;
    return 0;
}
int gemm_max_splitk() { return 32; }
int gemm_stat_slots(int M, int N, int K) {
    const GemmPlan g = stat_producer_plan(gemm_plan(M, N, K, 0));
    if (g.splitk > 1) return (N + 255) / 256;
    return (N + kTileN[g.cfg] - 1) / kTileN[g.cfg];            // one slot per column tile
}
int gemm_tile_index(int M, int N, int K, int conv, int stride, int up) { return gemm_plan(M, N, K, 0, conv, stride, up).cfg; }

size_t gemm_ws_bytes(int M, int N, int splitk) {
    return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0;
}

template <int TM, int TN, int WM, int WN, int STAGES, int KW>
static int launch_tile_kw(const GemmArgs& a, int splitk, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, splitk);
    const size_t lds = (size_t)KW * STAGES * (TM * 128 + TN * 128);
    static bool attr_set[2] = {false, false};
    if (lds > 64 * 1024 && !attr_set[a.conv ? 1 : 0]) {
        hipError_t e = a.conv ? hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 1, STAGES, 0, 0, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                              : hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, 0, 0, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
        attr_set[a.conv ? 1 : 0] = true;
    }
    dim3 grid((a.M + TM - 1) / TM, (a.N + TN - 1) / TN, splitk * (second ? 2 : 1));
    dim3 block(64 * WM * WN * KW);
    if (a.ln_s) {          // on-the-fly LayerNorm (resolve keeps in-block K split only for that form)
        static bool lattr = false;
        if (lds > 64 * 1024 && !lattr) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, -1, 0, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
            lattr = true;
        }
        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, -1, 0, KW>), grid, block, lds, stream, ag);
        return 0;
    }
    if (a.conv) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, STAGES, 0, 0, KW>), grid, block, lds, stream, ag);
    else        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, 0, 0, KW>), grid, block, lds, stream, ag);
    return 0;
}

// plain-epilogue-only launcher (tile configurations 30..37): two instantiations per configuration
template <int TM, int TN, int WM, int WN, int STAGES>
static int launch_tile_light(const GemmArgs& a, int splitk, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, splitk);
    const size_t lds = (size_t)STAGES * (TM * 128 + TN * 128) + (size_t)WN * TM * 2 * sizeof(float);
    static bool attr_set[2] = {false, false};
    if (lds > 64 * 1024 && !attr_set[a.conv ? 1 : 0]) {
        hipError_t e = a.conv ? hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 1, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                              : hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
        attr_set[a.conv ? 1 : 0] = true;
    }
    dim3 grid((a.M + TM - 1) / TM, (a.N + TN - 1) / TN, splitk * (second ? 2 : 1));
    dim3 block(64 * WM * WN);
    if (a.conv) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, STAGES>), grid, block, lds, stream, ag);
    else        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES>), grid, block, lds, stream, ag);
    return 0;
}

template <int TM, int TN, int WM, int WN, int STAGES>
static int launch_tile(const GemmArgs& a, int splitk, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, splitk);
    const bool gns = a.gn_stat != nullptr && splitk == 1;
    const size_t lds = (size_t)STAGES * (TM * 128 + TN * 128) + (gns ? (size_t)4096 : (size_t)WN * TM * 2 * sizeof(float));   // ring + tail
    static bool attr_set[2] = {false, false};
    if (lds > 64 * 1024 && !attr_set[a.conv ? 1 : 0]) {
        hipError_t e = a.conv ? hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 1, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                              : hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
        attr_set[a.conv ? 1 : 0] = true;
    }
    dim3 grid((a.M + TM - 1) / TM, (a.N + TN - 1) / TN, splitk * (second ? 2 : 1));
    dim3 block(64 * WM * WN);
    if (gns) {
        static bool gattr[2] = {false, false};
        if (lds > 64 * 1024 && !gattr[a.conv ? 1 : 0]) {
            hipError_t e = a.conv ? hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 1, STAGES, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                  : hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
            gattr[a.conv ? 1 : 0] = true;
        }
        if (a.conv) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, STAGES, 0, 1>), grid, block, lds, stream, ag);
        else        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, 0, 1>), grid, block, lds, stream, ag);
        return 0;
    }
    if (a.ln_s && !a.stat_in) {          // LayerNorm statistics taken by the GEMM itself
        static bool lf_attr = false;
        if (lds > 64 * 1024 && !lf_attr) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
            lf_attr = true;
        }
        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, -1>), grid, block, lds, stream, ag);
        return 0;
    }
    if (a.ln_s) {
        static bool ln_attr = false;
        if (lds > 64 * 1024 && !ln_attr) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_kernel<TM, TN, WM, WN, 0, STAGES, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
            ln_attr = true;
        }
        if (a.stat_in_slots <= 4) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, 1>), grid, block, lds, stream, ag);
        else if (a.stat_in_slots <= 12) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, 3>), grid, block, lds, stream, ag);
        else hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES, 5>), grid, block, lds, stream, ag);
    } else if (a.conv) hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 1, STAGES>), grid, block, lds, stream, ag);
    else        hipLaunchKernelGGL((gemm_kernel<TM, TN, WM, WN, 0, STAGES>), grid, block, lds, stream, ag);
    return 0;
}

bool gemm_same_geometry(const GemmArgs& a, const GemmArgs& b) {
    return a.M == b.M && a.N == b.N && a.K == b.K && a.lda == b.lda && a.ldw == b.ldw && a.conv == b.conv && a.Hin == b.Hin && a.Win == b.Win &&
           a.Cin == b.Cin && a.Hout == b.Hout && a.Wout == b.Wout && a.stride == b.stride && a.up == b.up && a.splitk == b.splitk &&
           a.out_f32 == b.out_f32 && a.act == b.act && a.ldc == b.ldc && a.defer_epilogue == b.defer_epilogue && a.rows_per_batch == b.rows_per_batch &&
           (a.ln_s != nullptr) == (b.ln_s != nullptr) && a.ln_eps == b.ln_eps && !a.stat_in && !b.stat_in && !a.stat_out && !b.stat_out &&
           !a.gn_stat && !b.gn_stat && (a.R != nullptr) == (b.R != nullptr) && a.ldr == b.ldr && (a.rowbias != nullptr) == (b.rowbias != nullptr) &&
           (a.bias != nullptr) == (b.bias != nullptr) && (a.A2 != nullptr) == (b.A2 != nullptr) && a.K2 == b.K2 && a.lda2 == b.lda2;
}

template <int TM, int TN, int NW, int STAGES>
static int launch_tile_ra(const GemmArgs& a, int splitk, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, splitk);
    const size_t lds = (size_t)2 * TN * 128;
    dim3 grid((a.M + TM - 1) / TM, (a.N + TN - 1) / TN, splitk * (second ? 2 : 1));
    dim3 block(64 * NW);
    if (a.conv) hipLaunchKernelGGL((gemm_ra_kernel<TM, TN, NW, 1, STAGES>), grid, block, lds, stream, ag);
    else        hipLaunchKernelGGL((gemm_ra_kernel<TM, TN, NW, 0, STAGES>), grid, block, lds, stream, ag);
    return 0;
}

int launch_gemm(GemmArgs a, hipStream_t stream, const GemmArgs* second) {
    GemmArgs b;
    if (second) {
        if (!gemm_same_geometry(a, *second)) return mkd_fail(-1, "gemm: a grouped launch needs two problems of identical geometry");
        if (!second->A || !second->W || !second->C) return mkd_fail(-1, "gemm: grouped launch with a null operand");
        b = *second; b.zero = a.zero;
    }
    if (!a.xcd_mode) a.xcd_mode = g_xcd_mode;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return mkd_fail(-1, "gemm: empty problem");
    if (a.N % 4) return mkd_fail(-1, "gemm: N must be a multiple of 4");
    if (a.K % 8 || a.ldw % 8 || a.lda % 8) return mkd_fail(-1, "gemm: K, lda, ldw must be multiples of 8");
    if (a.A2 && (!a.conv || a.stride != 1 || a.up != 0 || a.K2 <= 0 || a.K2 % BK || (9 * a.Cin) % BK || a.lda2 % 8 || a.Hin != a.Hout || a.Win != a.Wout))
        return mkd_fail(-1, "gemm: a folded second input needs a stride-1 conv3x3 and K2, 9 * Cin multiples of 64");
    if (!a.A2) a.K2 = 0;
    if (a.conv && (a.Cin % 8 || a.K != 9 * a.Cin + a.K2)) return mkd_fail(-1, "gemm: conv needs Cin % 8 == 0 and K == 9*Cin (+ K2)");
    if (!a.zero) return mkd_fail(-1, "gemm: zero page missing");
    if (!a.out_f32 && (a.ldc % 4)) return mkd_fail(-1, "gemm: ldc must be a multiple of 4");
    if (a.act == 2 && (a.out_f32 || a.R)) return mkd_fail(-1, "gemm: GEGLU epilogue takes no residual and writes bf16");
    if (a.R && (a.ldr % 4)) return mkd_fail(-1, "gemm: ldr must be a multiple of 4");
    if (a.rowbias && a.rows_per_batch <= 0) return mkd_fail(-1, "gemm: rows_per_batch");
    if (a.ln_s && (a.conv || a.K > 8192)) return mkd_fail(-1, "gemm: fused LayerNorm is for linear GEMMs");
    if (a.ln_s && a.stat_in && (a.stat_in_slots <= 0 || a.stat_in_slots > 20))
        return mkd_fail(-1, "gemm: fused LayerNorm with producer statistics needs them in 1..20 column slots");
    if (a.stat_out && (a.conv || a.act == 2 || a.out_f32)) return mkd_fail(-1, "gemm: row statistics are emitted by plain bf16 linear GEMMs only");
    if (a.gn_stat && (a.act == 2 || a.out_f32 || a.stat_out || a.ln_s || a.gn_cg <= 0 || a.gn_hw <= 0 || a.gn_coff < 0 || (a.gn_coff + a.N + a.gn_cg - 1) / a.gn_cg > 32))
        return mkd_fail(-1, "gemm: GroupNorm statistics need a plain bf16 output, rows per sample, channels per group, <= 32 groups");
    GemmPlan g;
    { const int rc = gemm_resolve_plan(a, &g); if (rc) return rc; }
    if (a.defer_epilogue && !is_patch_cfg(g.cfg) && (g.splitk < 2 || (a.expect_splitk > 0 && g.splitk != a.expect_splitk)))
        return mkd_fail(-3, "gemm: deferred split-K epilogue planned for " + std::to_string(a.expect_splitk) + " slabs, the launch resolves to " +
                                std::to_string(g.splitk) + " (tile / split-K settings changed after mkd_prepare: prepare again)");
    if (is_patch_cfg(g.cfg)) return launch_conv_patch(a, g.cfg, g.splitk, stream, second ? &b : nullptr);
    if (g.splitk > 1 && (!a.ws || (second && (!b.ws || b.ws == a.ws)))) return mkd_fail(-1, "gemm: split-K needs a workspace (one per problem)");
    if (second && g.splitk > 1 && gemm_ws_bytes(b.M, b.N, g.splitk) > b.ws_bytes) return mkd_fail(-1, "gemm: split-K workspace of the second problem too small");
    if (g.splitk > 1 && gemm_ws_bytes(a.M, a.N, g.splitk) > a.ws_bytes)
        return mkd_fail(-1, "gemm: split-K workspace too small (" + std::to_string(a.ws_bytes) + " B for " + std::to_string(g.splitk) + " slabs of " +
                                std::to_string(a.M) + " x " + std::to_string(a.N) + ")");
    a.splitk = g.splitk;
    a.ksteps_per_split = g.per;
    const GemmArgs* const sp = second ? &b : nullptr;
    if (second) { b.splitk = a.splitk; b.ksteps_per_split = a.ksteps_per_split; b.xcd_mode = a.xcd_mode; }
    int rc;
    switch (g.cfg) {
        case 0: rc = launch_tile<256, 128, 4, 2, 3>(a, g.splitk, stream, sp); break;
        case 1: rc = launch_tile<128, 128, 2, 2, 3>(a, g.splitk, stream, sp); break;
        case 2: rc = launch_tile<128, 128, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 3: rc = launch_tile<128, 64, 2, 2, 3>(a, g.splitk, stream, sp); break;
        case 4: rc = launch_tile<64, 128, 2, 2, 3>(a, g.splitk, stream, sp); break;
        case 12: rc = launch_tile<64, 64, 2, 2, 6>(a, g.splitk, stream, sp); break;
        case 13: rc = launch_tile<64, 128, 2, 2, 5>(a, g.splitk, stream, sp); break;
        case 14: rc = launch_tile<64, 160, 2, 2, 3>(a, g.splitk, stream, sp); break;
        case 15: rc = launch_tile<128, 160, 2, 2, 3>(a, g.splitk, stream, sp); break;
        case 16: rc = launch_tile<64, 160, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 17: rc = launch_tile<32, 64, 2, 2, 4>(a, g.splitk, stream, sp); break;
        case 18: rc = launch_tile<64, 32, 2, 2, 4>(a, g.splitk, stream, sp); break;
        case 19: rc = launch_tile<32, 32, 2, 2, 4>(a, g.splitk, stream, sp); break;
        case 20: rc = launch_tile_kw<32, 32, 2, 2, 4, 2>(a, g.splitk, stream, sp); break;
        case 21: rc = launch_tile_kw<32, 32, 2, 2, 4, 4>(a, g.splitk, stream, sp); break;
        case 22: rc = launch_tile_kw<64, 32, 2, 2, 4, 2>(a, g.splitk, stream, sp); break;
        case 23: rc = launch_tile_kw<64, 32, 2, 2, 3, 4>(a, g.splitk, stream, sp); break;
        case 24: rc = launch_tile_kw<64, 64, 2, 2, 4, 2>(a, g.splitk, stream, sp); break;
        case 25: rc = launch_tile_kw<64, 64, 2, 2, 2, 4>(a, g.splitk, stream, sp); break;
        case 26: rc = launch_tile_kw<32, 64, 2, 2, 4, 2>(a, g.splitk, stream, sp); break;
        case 27: rc = launch_tile_kw<32, 64, 2, 2, 3, 4>(a, g.splitk, stream, sp); break;
        case 28: rc = launch_tile_kw<128, 64, 2, 2, 3, 2>(a, g.splitk, stream, sp); break;
        case 29: rc = launch_tile_kw<64, 128, 2, 2, 3, 2>(a, g.splitk, stream, sp); break;
        case 30: rc = launch_tile_light<64, 64, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 31: rc = launch_tile_light<128, 64, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 32: rc = launch_tile_light<64, 128, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 33: rc = launch_tile_light<64, 32, 2, 2, 2>(a, g.splitk, stream, sp); break;
        case 34: rc = launch_tile_light<128, 128, 4, 2, 2>(a, g.splitk, stream, sp); break;
        case 35: rc = launch_tile_light<128, 64, 4, 2, 3>(a, g.splitk, stream, sp); break;
        case 36: rc = launch_tile_light<64, 128, 2, 4, 3>(a, g.splitk, stream, sp); break;
        case 37: rc = launch_tile_light<64, 64, 2, 4, 4>(a, g.splitk, stream, sp); break;
        case 41: rc = launch_tile_light<256, 64, 4, 2, 3>(a, g.splitk, stream, sp); break;
        case 50: rc = launch_tile_light<256, 256, 4, 4, 2>(a, g.splitk, stream, sp); break;
        case 44: rc = launch_tile_ra<256, 64, 8, 4>(a, g.splitk, stream, sp); break;
        case 45: rc = launch_tile_ra<256, 128, 8, 3>(a, g.splitk, stream, sp); break;
        case 46: rc = launch_tile_ra<128, 128, 4, 3>(a, g.splitk, stream, sp); break;
        case 47: rc = launch_tile_ra<128, 64, 4, 4>(a, g.splitk, stream, sp); break;
        case 48: rc = launch_tile_ra<128, 160, 4, 3>(a, g.splitk, stream, sp); break;
        case 49: rc = launch_tile_ra<256, 64, 4, 3>(a, g.splitk, stream, sp); break;
        default: rc = launch_tile<64, 64, 2, 2, 4>(a, g.splitk, stream, sp); break;
    }
    if (rc) return rc;
    MKD_LAUNCH_CHECK("gemm_kernel");
    if (g.splitk > 1 && !a.defer_epilogue) return launch_splitk_epilogue(a, stream, sp);
    return 0;
}

int launch_splitk_epilogue(const GemmArgs& a, hipStream_t stream, const GemmArgs* second) {
    const GemmArgs2 ag = gemm_pack2(a, second, 1);
    const int gz = second ? 2 : 1;
#ifdef MKD_EXP_ABLATE
    static const int skip = getenv("MKD_EXP_SKIP") ? atoi(getenv("MKD_EXP_SKIP")) : 0;      // experiment build only: 8 = no split-K reduce launches (WRONG results)
    if (skip & 8) return 0;
#endif
    if (a.gn_stat) {
        dim3 rg((a.N / 4 + 63) / 64, (a.M + 3) / 4, gz);
        hipLaunchKernelGGL(splitk_epilogue_gn_kernel, rg, dim3(256), 0, stream, ag);
        MKD_LAUNCH_CHECK("splitk_epilogue_gn_kernel");
        return 0;
    }
    dim3 rg((a.N / 4 + 63) / 64, (a.M + 3) / 4, gz);
    hipLaunchKernelGGL(splitk_epilogue_kernel, rg, dim3(256), 0, stream, ag);
    MKD_LAUNCH_CHECK("splitk_epilogue_kernel");
    return 0;
}
