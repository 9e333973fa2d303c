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
//   * 2 LDS stages, one barrier per K-step: loads of step k+1 fly under the MFMAs of step k.
//   * the product is computed transposed (D = W_tile . X_tile^T with v_mfma_f32_16x16x32_bf16) so
//     each lane ends with 4 CONSECUTIVE n for one m: 8-byte bf16x4 stores, float4 bias loads.
//   * small-M layers (4x4 / 8x8 latents) are weight-bandwidth bound: split-K over blockIdx.z with
//     fp32 partial slabs + a fused reduce/epilogue kernel fills the 256 CUs.
#include "mkd_common.h"

namespace {

constexpr int TM = 128;
constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int m, int n, f32x4 v) {
    if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
        v += b;
    }
    if (p.rowbias) {
        const f32x4 b = *(const f32x4*)(p.rowbias + (size_t)(m / p.rows_per_batch) * p.ldrb + n);
        v += b;
    }
    v *= p.scale;
    if (p.R) {
        const U16x4 r = *(const U16x4*)(p.R + (size_t)m * p.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bf16_to_f32(r.v[j]);
    }
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    }
    if (p.out_f32) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
    } else {
        U16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = f32_to_bf16(v[j]);
        *(U16x4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
    }
}

template <int TN, int CONV>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XS = TM * 128;          // bytes of one X stage (128 rows x 64 bf16)
    constexpr int WSB = TN * 128;         // bytes of one W stage
    constexpr int STAGE = XS + WSB;
    constexpr int WP = TN / 32;           // W pieces (1 KiB) per wave
    constexpr int NI = TN / 32;           // W fragments (16 rows) per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * p.ksteps_per_split;
    const int kt_end = min(nk_total, kt_begin + p.ksteps_per_split);

    // ---- per-lane staging geometry -----------------------------------------------------------
    const int lrow = lane >> 3;                               // row inside a 1 KiB piece
    const int key = (4 * (w & 1) + (lane >> 4)) & 7;          // == ((tile_row >> 1) & 7) for every piece of this wave
    const int sc = (lane & 7) ^ key;                          // source 16-B chunk inside the 128-B K row

    const bf16_t* xptr[4];
    int xbase[4], uy0[4], ux0[4];
    const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 8 * (w + 4 * i) + lrow;
        if (CONV) {
            xptr[i] = nullptr;
            if (m < p.M) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw;
                const int rem = m - b * hw;
                const int oy = rem / p.Wout;
                const int ox = rem - oy * p.Wout;
                xbase[i] = b * p.Hin * p.Win;
                uy0[i] = oy * p.stride - 1;
                ux0[i] = ox * p.stride - 1;
            } else {
                xbase[i] = 0; uy0[i] = -(1 << 20); ux0[i] = -(1 << 20);
            }
        } else {
            xptr[i] = (m < p.M) ? p.A + (size_t)m * p.lda : nullptr;
            xbase[i] = uy0[i] = ux0[i] = 0;
        }
    }
    const bf16_t* wptr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int n = n0 + 8 * (w + 4 * i) + lrow;
        wptr[i] = (n < p.N) ? p.W + (size_t)n * p.ldw : nullptr;
    }

    auto stage = [&](int buf, int kt) {
        char* xs = smem + buf * STAGE;
        char* wsm = xs + XS;
        const int k = kt * BK + sc * 8;
        const bool kok = k < p.K;
        int ci = 0, ky = 0, kx = 0;
        if (CONV) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            ky = (tap * 11) >> 5;          // tap / 3 for tap in [0, 9)
            kx = tap - 3 * ky;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = p.zero;
            if (CONV) {
                const int uy = uy0[i] + ky, ux = ux0[i] + kx;
                if (kok && (unsigned)uy < (unsigned)Hup && (unsigned)ux < (unsigned)Wup)
                    src = p.A + ((size_t)(xbase[i] + (uy >> p.up) * p.Win + (ux >> p.up)) * p.lda + ci);
            } else {
                if (kok && xptr[i]) src = xptr[i] + k;
            }
            glds16(src, xs + (w + 4 * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const bf16_t* src = (kok && wptr[i]) ? wptr[i] + k : p.zero;
            glds16(src, wsm + (w + 4 * i) * 1024);
        }
    };

    f32x4 acc[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = lane >> 4;

    auto compute = [&](int buf) {
        const char* xs = smem + buf * STAGE;
        const char* wsm = xs + XS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = 4 * kk + fq;
            bf16x8 xf[4], wf[NI];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = wm * 64 + mi * 16 + frow;
                xf[mi] = *(const bf16x8*)(xs + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * (TN / 2) + ni * 16 + frow;
                wf[ni] = *(const bf16x8*)(wsm + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
    };

    if (kt_begin < kt_end) {
        stage(0, kt_begin);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            __syncthreads();               // tile kt landed (vmcnt(0) + barrier); buf^1 is free again
            if (kt + 1 < kt_end) stage(buf ^ 1, kt + 1);
            compute(buf);
        }
    }

    // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of every (ni, mi) fragment ---------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (TN / 2) + ni * 16 + 4 * fq;
            if (n >= p.N) continue;
            if (p.splitk > 1) {
                *(f32x4*)(p.ws + ((size_t)blockIdx.z * p.M + m) * p.N + n) = acc[ni][mi];
            } else {
                epilogue_store(p, m, n, acc[ni][mi]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmArgs p) {
    const int nq = p.N >> 2;
    const int64_t total = (int64_t)p.M * nq;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int m = (int)(idx / nq);
        const int n = (int)(idx - (int64_t)m * nq) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.splitk; ++z) v += *(const f32x4*)(p.ws + ((size_t)z * p.M + m) * p.N + n);
        epilogue_store(p, m, n, v);
    }
}

}  // namespace

int gemm_pick_splitk(int M, int N, int K) {
    const int tn = (N % 128 == 0) ? 128 : 64;
    const int tiles = ((M + TM - 1) / TM) * ((N + tn - 1) / tn);
    const int nk = (K + BK - 1) / BK;
    if (tiles >= 128 || nk < 4) return 1;
    int s = 256 / tiles;
    if (s > nk / 2) s = nk / 2;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

size_t gemm_ws_bytes(int M, int N, int splitk) {
    return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0;
}

int launch_gemm(GemmArgs a, hipStream_t stream) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return mkd_fail(-1, "gemm: empty problem");
    if (a.N % 4) return mkd_fail(-1, "gemm: N must be a multiple of 4");
    if (a.K % 8 || a.ldw % 8 || a.lda % 8) return mkd_fail(-1, "gemm: K, lda, ldw must be multiples of 8");
    if (a.conv && (a.Cin % 8 || a.K != 9 * a.Cin)) return mkd_fail(-1, "gemm: conv needs Cin % 8 == 0 and K == 9*Cin");
    if (!a.zero) return mkd_fail(-1, "gemm: zero page missing");
    if (!a.out_f32 && (a.ldc % 4)) return mkd_fail(-1, "gemm: ldc must be a multiple of 4");
    if (a.R && (a.ldr % 4)) return mkd_fail(-1, "gemm: ldr must be a multiple of 4");
    if (a.rowbias && a.rows_per_batch <= 0) return mkd_fail(-1, "gemm: rows_per_batch");
    const int tn = (a.N % 128 == 0) ? 128 : 64;
    const int nk = (a.K + BK - 1) / BK;
    int s = a.splitk;
    if (s <= 0) s = gemm_pick_splitk(a.M, a.N, a.K);
    if (s > nk) s = nk;
    int per = (nk + s - 1) / s;
    s = (nk + per - 1) / per;
    if (s > 1 && !a.ws) return mkd_fail(-1, "gemm: split-K needs a workspace");
    a.splitk = s;
    a.ksteps_per_split = per;
    dim3 grid((a.M + TM - 1) / TM, (a.N + tn - 1) / tn, s);
    const size_t lds = 2 * (TM * 128 + tn * 128);
    if (tn == 128) {
        if (a.conv) hipLaunchKernelGGL((gemm_kernel<128, 1>), grid, dim3(256), lds, stream, a);
        else        hipLaunchKernelGGL((gemm_kernel<128, 0>), grid, dim3(256), lds, stream, a);
    } else {
        if (a.conv) hipLaunchKernelGGL((gemm_kernel<64, 1>), grid, dim3(256), lds, stream, a);
        else        hipLaunchKernelGGL((gemm_kernel<64, 0>), grid, dim3(256), lds, stream, a);
    }
    MKD_LAUNCH_CHECK("gemm_kernel");
    if (s > 1) {
        const int64_t total = (int64_t)a.M * (a.N / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, a);
        MKD_LAUNCH_CHECK("splitk_epilogue_kernel");
    }
    return 0;
}
