// GroupNorm(32)[+SiLU] and LayerNorm over NHWC bf16 (HBM-bound kernels, fp32 statistics).
// Replaces cldm GroupNorm32 -> SiLU in ResBlock.in_layers/out_layers, SpatialTransformer.norm and
// BasicTransformerBlock.norm1-3, reached from diffmk/makeup_diffuse.py:164-168.
//
// GroupNorm: a sample's statistics span all pixels, so it is two passes over a tensor that is
// L2/Infinity-Cache resident: (1) per-(sample, pixel-chunk) partial sum/sumsq per group, every
// thread owning a FIXED 8-channel vector so sums stay in registers, 16-B coalesced loads;
// (2) fold partials -> mean/rstd, apply gamma/beta (+SiLU), 16-B stores.
// LayerNorm: one 64-lane wavefront per row, row kept in registers, wave-shuffle reductions.
#include "mkd_common.h"
#include <cstdlib>
#include "gemm_device.h"

namespace {
using mkdk::GN_FIX_SUM; using mkdk::GN_FIX_SQ; using mkdk::GN_GROUPS; using mkdk::gn_atomic_add;

constexpr int GN_MAX_GROUPS = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// threads = V * P where V = C/8 vectors per pixel (thread's vector id = tid % V, fixed).
// Deterministic: per-thread register sums -> LDS -> one thread per group adds them in a fixed order.
__global__ void gn_stats_kernel(const bf16_t* __restrict__ x, int ld, int hw, int C, int groups,
                                int rows_per_chunk, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float s_part[];   // [2][T][8]
    const int V = C >> 3;
    const int T = blockDim.x;
    const int P = T / V;
    const int tid = threadIdx.x;
    const int v = tid % V;
    const int pl = tid / V;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(hw, r0 + rows_per_chunk);
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
    const bf16_t* base = x + (size_t)b * hw * ld + v * 8;
    for (int r = r0 + pl; r < r1; r += P) {
        const U16x8 d = *(const U16x8*)(base + (size_t)r * ld);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf16_to_f32(d.v[j]);
            sum[j] += f; sq[j] += f * f;
        }
    }
    float* ssum = s_part;
    float* ssq = s_part + T * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[tid * 8 + j] = sum[j]; ssq[tid * 8 + j] = sq[j]; }
    __syncthreads();
    if (tid < groups) {
        const int cg = C / groups;
        float a = 0.f, q = 0.f;
        for (int p = 0; p < P; ++p)
            for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
                const int idx = (p * V + (c >> 3)) * 8 + (c & 7);
                a += ssum[idx]; q += ssq[idx];
            }
        float* o = partials + ((size_t)(b * gridDim.x + chunk) * groups + tid) * 2;
        o[0] = a; o[1] = q;
    }
}

__global__ void gn_apply_kernel(const bf16_t* __restrict__ x, int ld_in, bf16_t* __restrict__ y, int ld_out,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                int silu, int hw, int C, int groups, int rows_per_chunk, int nchunks,
                                const float* __restrict__ partials) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    const int V = C >> 3;
    const int T = blockDim.x;
    const int P = T / V;
    const int tid = threadIdx.x;
    const int v = tid % V;
    const int pl = tid / V;
    const int b = blockIdx.y, chunk = blockIdx.x;
    if (tid < groups) {
        float s = 0.f, q = 0.f;
        for (int c = 0; c < nchunks; ++c) {
            const float* pp = partials + ((size_t)(b * nchunks + c) * groups + tid) * 2;
            s += pp[0]; q += pp[1];
        }
        const float n = (float)hw * (float)(C / groups);
        const float mean = s / n;
        float var = q / n - mean * mean;
        var = var < 0.f ? 0.f : var;
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    const int cg = C / groups;
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = v * 8 + j;
        const int g = c / cg;
        const float a = gamma[c] * s_rstd[g];
        sa[j] = a;
        sb[j] = beta[c] - s_mean[g] * a;
    }
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(hw, r0 + rows_per_chunk);
    const bf16_t* xin = x + (size_t)b * hw * ld_in + v * 8;
    bf16_t* yout = y + (size_t)b * hw * ld_out + v * 8;
    for (int r = r0 + pl; r < r1; r += P) {
        const U16x8 d = *(const U16x8*)(xin + (size_t)r * ld_in);
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf16_to_f32(d.v[j]) * sa[j] + sb[j];
            if (silu) f = silu_f(f);
            o.v[j] = f32_to_bf16(f);
        }
        *(U16x8*)(yout + (size_t)r * ld_out) = o;
    }
}

// Single-launch GroupNorm: one workgroup per (sample, chunk of `gpb` groups whose channel span is a multiple
// of 8).  NV > 0: the thread's <= NV 16-byte vectors stay in registers between the statistics and the apply, so
// the slab is read from memory exactly once (one latency round trip).  NV == 0: streaming two-pass variant for
// slabs that do not fit the register budget (second pass re-reads, L2 hit).  Statistics: per-thread fp32 sums ->
// LDS -> fixed-shape tree over the pixel lanes (deterministic) -> per-group mean / rstd.
template <int NV>
__global__ __launch_bounds__(NV == 16 ? 512 : 1024) void gn_fused_kernel(const Pair<NormIo> io, int ld_in, int ld_out, float eps, int silu, int hw,
                                                        int C, int groups, int gpb) {
    extern __shared__ __attribute__((aligned(16))) float s_red[];    // [2][NT][8] then per-channel / per-group
    // grouped launch: grid z selects the problem (same geometry, own tensors and affine parameters)
    const bf16_t* __restrict__ const x = io.g[blockIdx.z].x; bf16_t* __restrict__ const y = io.g[blockIdx.z].y;
    const float* __restrict__ const gamma = io.g[blockIdx.z].gamma; const float* __restrict__ const beta = io.g[blockIdx.z].beta;
    const int cg = C / groups;
    const int nch = gpb * cg;                 // channels of this block (multiple of 8)
    const int V = nch >> 3;
    const int NT = blockDim.x;
    const int P = NT / V;
    const int T = V * P;                      // active threads
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * nch;
    const bool active = tid < T;
    const int v = active ? tid % V : 0;
    const int pl = active ? tid / V : 0;
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
    const bf16_t* xin = x + (size_t)b * hw * ld_in + c0 + v * 8;
    constexpr int NR = NV > 0 ? NV : 1;
    U16x8 keep[NR];
    if (active) {
        if (NV > 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                // UNCONDITIONAL loads (rows past the slab re-read its last row and are ignored below): under `if (r < hw)` the
                // compiler waits for every load at the branch join - s_waitcnt vmcnt(0) behind each of the NV loads, NV serial
                // round trips instead of one (round 4: found in the ISA; the "5.1 of 10.9 us waiting for loads" of DESIGN 4.5)
                const int r = pl + i * P;
                keep[i] = *(const U16x8*)(xin + (size_t)(r < hw ? r : hw - 1) * ld_in);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = pl + i * P;
                if (r < hw) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(keep[i].v[j]); sum[j] += f; sq[j] += f * f; }
                }
            }
        } else {
            int r = pl;
            for (; r + 3 * P < hw; r += 4 * P) {          // 4 independent 16-B loads in flight per thread
                U16x8 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = *(const U16x8*)(xin + (size_t)(r + u * P) * ld_in);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(d[u].v[j]); sum[j] += f; sq[j] += f * f; }
            }
            for (; r < hw; r += P) {
                const U16x8 d = *(const U16x8*)(xin + (size_t)r * ld_in);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(d.v[j]); sum[j] += f; sq[j] += f * f; }
            }
        }
    }
    // gamma / beta of this thread's 8 channels: fetched now, next to the slab loads, not after the statistics barriers - except in the
    // 1024-thread instantiation (64 registers per thread: with its 8 slab vectors in flight at once they would spill), which fetches
    // them behind the statistics
    constexpr bool AFFINE_EARLY = NV != 8;
    float pg[8], pb[8];
    auto load_affine = [&]() {
        const f32x4 g0 = *(const f32x4*)(gamma + c0 + v * 8), g1 = *(const f32x4*)(gamma + c0 + v * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c0 + v * 8), b1 = *(const f32x4*)(beta + c0 + v * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[j] = g0[j]; pg[4 + j] = g1[j]; pb[j] = b0[j]; pb[4 + j] = b1[j]; }
    };
    if (AFFINE_EARLY && active) load_affine();
    float* ssum = s_red;                      // [NT][8]
    float* ssq = s_red + NT * 8;              // [NT][8]
    float* csum = s_red + 2 * NT * 8;         // [nch] per-channel totals
    float* csq = csum + nch;
    float* gstat = csq + nch;                 // [gpb][2] mean, rstd
    if (cg >= 8) {
        // Group statistics straight from the registers (block-uniform branch): a thread's 8 channels touch at most two groups;
        // per group a masked butterfly over the wave, then a fixed-order sum over the waves - two barriers, deterministic.
        const int gA = (v * 8) / cg;
        const int split = min(8, (gA + 1) * cg - v * 8);          // channels [0, split) belong to gA, the rest to gA + 1
        float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < split) { a0 += sum[j]; q0 += sq[j]; } else { a1 += sum[j]; q1 += sq[j]; }
            }
        }
        const int nw = NT >> 6, wv = tid >> 6;
        for (int g = 0; g < gpb; ++g) {
            float sg = (gA == g ? a0 : 0.f) + (gA + 1 == g ? a1 : 0.f);
            float qg = (gA == g ? q0 : 0.f) + (gA + 1 == g ? q1 : 0.f);
            sg = wave_sum(sg); qg = wave_sum(qg);
            if ((tid & 63) == 0) { s_red[(wv * gpb + g) * 2] = sg; s_red[(wv * gpb + g) * 2 + 1] = qg; }
        }
        __syncthreads();
        if (tid < gpb) {
            float a = 0.f, q = 0.f;
            for (int k = 0; k < nw; ++k) { a += s_red[(k * gpb + tid) * 2]; q += s_red[(k * gpb + tid) * 2 + 1]; }
            const float n = (float)hw * (float)cg;
            const float mean = a / n;
            float var = q / n - mean * mean;
            var = var < 0.f ? 0.f : var;
            gstat[tid * 2] = mean;
            gstat[tid * 2 + 1] = rsqrtf(var + eps);
        }
        __syncthreads();
    } else {
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { ssum[tid * 8 + j] = sum[j]; ssq[tid * 8 + j] = sq[j]; }
    }
    __syncthreads();
    for (int st = 1; st < P; st <<= 1) {
        if (active && (pl & (2 * st - 1)) == 0 && pl + st < P) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ssum[tid * 8 + j] += ssum[(tid + st * V) * 8 + j];
                ssq[tid * 8 + j] += ssq[(tid + st * V) * 8 + j];
            }
        }
        __syncthreads();
    }
    for (int c = tid; c < nch; c += NT) { csum[c] = ssum[c]; csq[c] = ssq[c]; }
    __syncthreads();
    if (tid < gpb) {
        float a = 0.f, q = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += csum[c]; q += csq[c]; }
        const float n = (float)hw * (float)cg;
        const float mean = a / n;
        float var = q / n - mean * mean;
        var = var < 0.f ? 0.f : var;
        gstat[tid * 2] = mean;
        gstat[tid * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    }
    if (!active) return;
    if (!AFFINE_EARLY) load_affine();
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = v * 8 + j;
        const int g = cl / cg;
        const float a = pg[j] * gstat[g * 2 + 1];
        sa[j] = a;
        sb[j] = pb[j] - gstat[g * 2] * a;
    }
    bf16_t* yout = y + (size_t)b * hw * ld_out + c0 + v * 8;
    auto apply_store = [&](const U16x8& d, int r) {
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf16_to_f32(d.v[j]) * sa[j] + sb[j];
            if (silu) f = silu_f(f);
            o.v[j] = f32_to_bf16(f);
        }
        *(U16x8*)(yout + (size_t)r * ld_out) = o;
    };
    if (NV > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = pl + i * P;
            if (r < hw) apply_store(keep[i], r);
        }
    } else {
        int r = pl;
        for (; r + 3 * P < hw; r += 4 * P) {
            U16x8 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = *(const U16x8*)(xin + (size_t)(r + u * P) * ld_in);
#pragma unroll
            for (int u = 0; u < 4; ++u) apply_store(d[u], r + u * P);
        }
        for (; r < hw; r += P) apply_store(*(const U16x8*)(xin + (size_t)r * ld_in), r);
    }
}

// GroupNorm [+SiLU] fed by the fp32 partial slabs of a split-K GEMM (launch_gemm with defer_epilogue): the workgroup of (sample,
// group chunk) sums the slabs of ITS rows and channels, applies the GEMM epilogue (bias, per-sample row bias, scale, residual) and
// rounds to bf16 exactly as splitk_epilogue_kernel would - the values stay in registers (NV vectors per thread), optionally go to
// memory (raw != null), then statistics and normalisation as in gn_fused_kernel.  Replaces two dependent launches (split-K reduce,
// GroupNorm) and the round trip of the intermediate tensor by one.
struct SlabGn {
    const float* ws; int splitk; int M; int N;
    const float* bias; const float* rowbias; int ldrb; int rpb; float scale; const bf16_t* R; int ldr;
    bf16_t* raw; int ldraw;
};

template <int NV>
__global__ __launch_bounds__(512) void gn_slab_kernel(const Pair<SlabGn> ag, const Pair<NormIo> io, int ld_out,
                                                                       float eps, int silu, int hw, int C, int groups, int gpb) {
    extern __shared__ __attribute__((aligned(16))) float s_red[];
    const SlabGn& a = ag.g[blockIdx.z];          // grouped launch: grid z selects the problem
    bf16_t* __restrict__ const y = io.g[blockIdx.z].y;
    const float* __restrict__ const gamma = io.g[blockIdx.z].gamma; const float* __restrict__ const beta = io.g[blockIdx.z].beta;
    const int cg = C / groups;
    const int nch = gpb * cg;                 // channels of this block (multiple of 8, cg >= 8: launcher)
    const int V = nch >> 3;
    const int NT = blockDim.x;
    const int P = NT / V;
    const int T = V * P;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * nch;
    const bool active = tid < T;
    const int v = active ? tid % V : 0;
    const int pl = active ? tid / V : 0;
    const int n = c0 + v * 8;
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
    U16x8 keep[NV];
    float pg[8], pb[8];
    if (active) {
        // epilogue operands of this thread's 8 channels (applied in splitk_epilogue_kernel's order: results are bit-identical to
        // the two-kernel path)
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f}, rb0 = {0.f, 0.f, 0.f, 0.f}, rb1 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) { e0 = *(const f32x4*)(a.bias + n); e1 = *(const f32x4*)(a.bias + n + 4); }
        if (a.rowbias) {
            const float* rb = a.rowbias + (size_t)(((size_t)b * hw) / a.rpb) * a.ldrb + n;      // one row bias per sample (rpb == hw)
            rb0 = *(const f32x4*)rb; rb1 = *(const f32x4*)(rb + 4);
        }
        const f32x4 g0 = *(const f32x4*)(gamma + n), g1 = *(const f32x4*)(gamma + n + 4);
        const f32x4 b0 = *(const f32x4*)(beta + n), b1 = *(const f32x4*)(beta + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[j] = g0[j]; pg[4 + j] = g1[j]; pb[j] = b0[j]; pb[4 + j] = b1[j]; }
        const size_t slab = (size_t)a.M * a.N;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = pl + i * P;
            if (r < hw) {
                const size_t m = (size_t)b * hw + r;
                const float* src = a.ws + m * a.N + n;
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                int z = 0;
                for (; z + 4 <= a.splitk; z += 4) {           // slabs summed 4 at a time, as splitk_epilogue_kernel does
                    const float* s0 = src + (size_t)z * slab;
                    lo += (*(const f32x4*)s0 + *(const f32x4*)(s0 + slab)) + (*(const f32x4*)(s0 + 2 * slab) + *(const f32x4*)(s0 + 3 * slab));
                    hi += (*(const f32x4*)(s0 + 4) + *(const f32x4*)(s0 + slab + 4)) + (*(const f32x4*)(s0 + 2 * slab + 4) + *(const f32x4*)(s0 + 3 * slab + 4));
                }
                for (; z < a.splitk; ++z) { lo += *(const f32x4*)(src + (size_t)z * slab); hi += *(const f32x4*)(src + (size_t)z * slab + 4); }
                lo += e0; hi += e1;
                if (a.rowbias) { lo += rb0; hi += rb1; }
                lo *= a.scale; hi *= a.scale;
                if (a.R) {
                    const U16x8 rr = *(const U16x8*)(a.R + m * a.ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { lo[j] += bf16_to_f32(rr.v[j]); hi[j] += bf16_to_f32(rr.v[4 + j]); }
                }
                U16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o.v[j] = f32_to_bf16(lo[j]); o.v[4 + j] = f32_to_bf16(hi[j]); }
                keep[i] = o;
                if (a.raw) *(U16x8*)(a.raw + m * a.ldraw + n) = o;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(o.v[j]); sum[j] += f; sq[j] += f * f; }
            }
        }
    }
    float* gstat = s_red + 2 * (NT >> 6) * gpb;          // [gpb][2] mean, rstd (after the per-wave partials)
    {   // group statistics straight from the registers (cg >= 8): as gn_fused_kernel
        const int gA = (v * 8) / cg;
        const int split = min(8, (gA + 1) * cg - v * 8);
        float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < split) { a0 += sum[j]; q0 += sq[j]; } else { a1 += sum[j]; q1 += sq[j]; }
            }
        }
        const int nw = NT >> 6, wv = tid >> 6;
        for (int g = 0; g < gpb; ++g) {
            float sg = (gA == g ? a0 : 0.f) + (gA + 1 == g ? a1 : 0.f);
            float qg = (gA == g ? q0 : 0.f) + (gA + 1 == g ? q1 : 0.f);
            sg = wave_sum(sg); qg = wave_sum(qg);
            if ((tid & 63) == 0) { s_red[(wv * gpb + g) * 2] = sg; s_red[(wv * gpb + g) * 2 + 1] = qg; }
        }
        __syncthreads();
        if (tid < gpb) {
            float s = 0.f, q = 0.f;
            for (int k = 0; k < nw; ++k) { s += s_red[(k * gpb + tid) * 2]; q += s_red[(k * gpb + tid) * 2 + 1]; }
            const float cnt = (float)hw * (float)cg;
            const float mean = s / cnt;
            float var = q / cnt - mean * mean;
            var = var < 0.f ? 0.f : var;
            gstat[tid * 2] = mean;
            gstat[tid * 2 + 1] = rsqrtf(var + eps);
        }
        __syncthreads();
    }
    if (!active) return;
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (v * 8 + j) / cg;
        const float sc = pg[j] * gstat[g * 2 + 1];
        sa[j] = sc;
        sb[j] = pb[j] - gstat[g * 2] * sc;
    }
    bf16_t* yout = y + (size_t)b * hw * ld_out + n;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int r = pl + i * P;
        if (r < hw) {
            U16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = bf16_to_f32(keep[i].v[j]) * sa[j] + sb[j];
                if (silu) f = silu_f(f);
                o.v[j] = f32_to_bf16(f);
            }
            *(U16x8*)(yout + (size_t)r * ld_out) = o;
        }
    }
}

// GroupNorm APPLY with statistics that the producers of x already accumulated (gemm_device.h: gstat[sample][32][2], 64-bit
// fixed point): no reduction, no dependency between workgroups - a plain element-wise kernel over (sample, row chunk) that
// fills the chip.  threads = V * P, V = C / 8 vectors per pixel (a thread keeps its 8 channels: scale / shift in registers).
__global__ __launch_bounds__(320) void gn_apply_stats_kernel(const bf16_t* __restrict__ x, int ld_in, bf16_t* __restrict__ y, int ld_out,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             int silu, int hw, int C, int rows_per_block,
                                                             const long long* __restrict__ gstat) {
    __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
    const int V = C >> 3;
    const int P = blockDim.x / V;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int cg = C / GN_GROUPS;
    if (tid < GN_GROUPS) {
        const long long a = gstat[((size_t)b * GN_GROUPS + tid) * 2], q = gstat[((size_t)b * GN_GROUPS + tid) * 2 + 1];
        const double n = (double)hw * (double)cg;
        const double mean = (double)a / ((double)GN_FIX_SUM * n);
        double var = (double)q / ((double)GN_FIX_SQ * n) - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = q ? rsqrtf((float)var + eps) : 1.0f;      // q == 0: an all-zero group, (0 - 0) * rstd = 0 whatever rstd is
    }
    const int v = tid % V, pl = tid / V;
    float pg[8], pb[8];
    {
        const f32x4 g0 = *(const f32x4*)(gamma + v * 8), g1 = *(const f32x4*)(gamma + v * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(beta + v * 8), b1 = *(const f32x4*)(beta + v * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[j] = g0[j]; pg[4 + j] = g1[j]; pb[j] = b0[j]; pb[4 + j] = b1[j]; }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(hw, r0 + rows_per_block);
    const bf16_t* xin = x + (size_t)b * hw * ld_in + v * 8;
    bf16_t* yout = y + (size_t)b * hw * ld_out + v * 8;
    // the slab loads do not depend on the statistics: issue the first ones before the barrier
    constexpr int U = 4;
    U16x8 d[U];
    int r = r0 + pl;
    if (pl < P) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u * P < r1) d[u] = *(const U16x8*)(xin + (size_t)(r + u * P) * ld_in);
    }
    __syncthreads();
    if (pl >= P) return;
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (v * 8 + j) / cg;
        const float a = pg[j] * s_rstd[g];
        sa[j] = a;
        sb[j] = pb[j] - s_mean[g] * a;
    }
    auto apply_store = [&](const U16x8& in, int row) {
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf16_to_f32(in.v[j]) * sa[j] + sb[j];
            if (silu) f = silu_f(f);
            o.v[j] = f32_to_bf16(f);
        }
        *(U16x8*)(yout + (size_t)row * ld_out) = o;
    };
    for (; r < r1; r += U * P) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u * P < r1) apply_store(d[u], r + u * P);
        const int rn = r + U * P;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (rn + u * P < r1) d[u] = *(const U16x8*)(xin + (size_t)(rn + u * P) * ld_in);
    }
}

// Fallback producer of the same statistics for tensors whose writer cannot emit them (the 4 -> C input convolution, plain
// copies): columns [0, ncols) of x (a slice of a wider consumer tensor starting at consumer column coff).
__global__ __launch_bounds__(320) void gn_colstats_kernel(const bf16_t* __restrict__ x, int ld, int hw, int ncols, int rows_per_block,
                                                          int cg, int coff, long long* __restrict__ gstat) {
    constexpr int CAP = 512;
    __shared__ long long acc[CAP * 2];
    const int V = ncols >> 3;                 // threads = V * P (launcher): thread -> (vector v, row lane pl)
    const int P = blockDim.x / V;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int g_first = coff / cg;
    const int ngl = (coff + ncols - 1) / cg - g_first + 1;
    const bool use_lds = ngl <= CAP;
    if (use_lds)
        for (int i = tid; i < ngl * 2; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const int v = tid % V, pl = tid / V;
    if (pl < P) {
        const bf16_t* base = x + (size_t)b * hw * ld + v * 8;
        float sm[8], sq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sm[j] = 0.f; sq[j] = 0.f; }
        for (int r = r0 + pl; r < r1; r += P) {          // all rows of a block belong to sample b
            const U16x8 d = *(const U16x8*)(base + (size_t)r * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(d.v[j]); sm[j] += f; sq[j] += f * f; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gl = (coff + v * 8 + j) / cg - g_first;
            const long long a = __float2ll_rn(sm[j] * GN_FIX_SUM), q = __float2ll_rn(sq[j] * GN_FIX_SQ);
            long long* dst = use_lds ? acc + (size_t)gl * 2 : gstat + ((size_t)b * GN_GROUPS + g_first + gl) * 2;
            if (a) gn_atomic_add(dst, a);
            if (q) gn_atomic_add(dst + 1, q);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = tid; i < ngl; i += blockDim.x) {
            const long long a = acc[i * 2], q = acc[i * 2 + 1];
            long long* dst = gstat + ((size_t)b * GN_GROUPS + g_first + i) * 2;
            if (a) gn_atomic_add(dst, a);
            if (q) gn_atomic_add(dst + 1, q);
        }
    }
}

// one wave per row; NV = vectors of 8 per lane (d <= 64*8*NV)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const Pair<NormIo> io, float eps, int rows, int d, int ldx) {
    const bf16_t* __restrict__ const x = io.g[blockIdx.y].x; bf16_t* __restrict__ const y = io.g[blockIdx.y].y;      // grouped launch: grid y selects the problem
    const float* __restrict__ const gamma = io.g[blockIdx.y].gamma; const float* __restrict__ const beta = io.g[blockIdx.y].beta;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int V = d >> 3;
    float f[NV][8];
    f32x4 pg[NV][2], pb[NV][2];               // gamma / beta: issued together with the row, consumed after the statistics
    float s = 0.f;
    // every load UNCONDITIONAL (vectors past the row re-read vector 0 and are zeroed / ignored below): a load under a per-lane
    // condition is waited for at the branch join - one serial round trip per vector (round 4)
    U16x8 tv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i, vc = v < V ? v : 0;
        tv[i] = *(const U16x8*)(x + (size_t)row * ldx + vc * 8);
        pg[i][0] = *(const f32x4*)(gamma + vc * 8); pg[i][1] = *(const f32x4*)(gamma + vc * 8 + 4);
        pb[i][0] = *(const f32x4*)(beta + vc * 8); pb[i][1] = *(const f32x4*)(beta + vc * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[i][j] = v < V ? bf16_to_f32(tv[i].v[j]) : 0.f; s += f[i][j]; }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < V) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float c = f[i][j] - mean; q += c * c; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < V) {
            U16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o.v[j] = f32_to_bf16((f[i][j] - mean) * rstd * pg[i][j >> 2][j & 3] + pb[i][j >> 2][j & 3]);
            }
            *(U16x8*)(y + (size_t)row * d + v * 8) = o;
        }
    }
}

// LayerNorm folding (load time): W'[n][k] = bf16(W[n][k] * gamma[k]) written to row dst_row0 + n * dst_row_mul,
// s[n] = sum_k float(W'[n][k]) (of the ROUNDED values the MFMA will see), b'[n] = bias[n] + sum_k W[n][k] * beta[k].
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             int N, int K, bf16_t* __restrict__ w_out, int dst_row0,
                                                             int dst_row_mul, float* __restrict__ s_out, float* __restrict__ b_out) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    const int dr = dst_row0 + n * dst_row_mul;
    float s = 0.f, bb = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wv = w[(size_t)n * K + k];
        const uint16_t r = f32_to_bf16(wv * gamma[k]);
        w_out[(size_t)dr * K + k] = r;
        s += bf16_to_f32(r);
        bb += wv * beta[k];
    }
    s = wave_sum(s); bb = wave_sum(bb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_out[dr] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        b_out[dr] = (bias ? bias[n] : 0.f) + red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

int gn_chunks(int hw) {
    int n = hw / 16;              // >= 16 pixels per chunk
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

}  // namespace

size_t groupnorm_partials_bytes(int batch, int hw, int groups) {
    return (size_t)batch * gn_chunks(hw) * groups * 2 * sizeof(float);
}

// first half of the two-launch GroupNorm on its own: per (sample, row chunk, group) partial (sum, sum of squares) ->
// partials[(b * nchunks + chunk) * groups + g][2]; full-chip grid.  Also the producer of the statistics tfm_head_kernel applies.
int launch_gn_stats(const bf16_t* x, int ld_in, int batch, int hw, int C, int groups, float* partials, hipStream_t stream, int* nchunks_out) {
    if (C % 8 || ld_in % 8 || groups <= 0 || groups > GN_MAX_GROUPS || C % groups) return mkd_fail(-1, "gn_stats: bad geometry");
    if (!partials) return mkd_fail(-1, "gn_stats: partials workspace missing");
    const int V = C / 8;
    if (V > 1024) return mkd_fail(-4, "groupnorm: C > 8192 unsupported");
    int P = 256 / V;
    if (P < 1) P = 1;
    const int threads = V * P;
    const int nchunks = gn_chunks(hw);
    const int rows_per_chunk = (hw + nchunks - 1) / nchunks;
    dim3 grid(nchunks, batch);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(threads), (size_t)threads * 16 * sizeof(float), stream, x, ld_in, hw, C, groups, rows_per_chunk, partials);
    MKD_LAUNCH_CHECK("gn_stats_kernel");
    if (nchunks_out) *nchunks_out = nchunks;
    return 0;
}

int launch_groupnorm(const bf16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu,
                     bf16_t* y, int ld_out, int batch, int hw, int C, int groups, float* partials,
                     hipStream_t stream, const NormIo* second, int two_kernel_min_hw) {
    if (C % 8 || ld_in % 8 || ld_out % 8) return mkd_fail(-1, "groupnorm: C, ld must be multiples of 8");
    Pair<NormIo> io;
    io.g[0] = NormIo{x, y, gamma, beta};
    MKD_PAIR_SET2(io, second ? *second : io.g[0]);
    if (groups > GN_MAX_GROUPS || groups <= 0 || C % groups) return mkd_fail(-1, "groupnorm: bad group count");
    if (!partials) return mkd_fail(-1, "groupnorm: partials workspace missing");
    // single-launch path: smallest group chunk whose channel span is a multiple of 8, slab small enough to
    // be re-read from cache
    {
        const int cg = C / groups;
        int gpb = 1;
        while (gpb <= groups && ((gpb * cg) % 8 || groups % gpb)) ++gpb;
        if (gpb <= groups) {
            const int nch = gpb * cg;
            const size_t slab = (size_t)hw * nch * sizeof(bf16_t);
            if (nch / 8 <= 256 && slab <= (size_t)384 * 1024 && hw < two_kernel_min_hw) {   // (V <= 256 <= blockDim)
                // smallest block (256..1024 threads) whose threads hold their whole share in <= 16 registers-vectors
                const int V = nch / 8;
                // 256 / 512 threads with <= 16 vectors per thread, or 1024 threads with <= 8 (register budget);
                // otherwise the streaming two-pass variant
                int nt = 256, per = 0;
                auto per_of = [&](int t) { const int P = t / V; return (hw + P - 1) / P; };
                if (per_of(256) <= 16) { nt = 256; per = per_of(256); }
                else if (per_of(512) <= 16) { nt = 512; per = per_of(512); }
                else if (per_of(1024) <= 8) { nt = 1024; per = per_of(1024); }
                else { nt = 256; per = 1 << 20; }
                const size_t lds = (size_t)(2 * nt * 8 + 2 * nch + 2 * gpb) * sizeof(float);
                dim3 grid(groups / gpb, batch, second ? 2 : 1);
#define MKD_GN_LAUNCH(NVV)                                                                                              \
    hipLaunchKernelGGL(gn_fused_kernel<NVV>, grid, dim3(nt), lds, stream, io, ld_in, ld_out, eps, silu, hw, C, groups, gpb)
                if (per <= 4) MKD_GN_LAUNCH(4);
                else if (per <= 8) MKD_GN_LAUNCH(8);
                else if (per <= 16) MKD_GN_LAUNCH(16);
                else MKD_GN_LAUNCH(0);
#undef MKD_GN_LAUNCH
                MKD_LAUNCH_CHECK("gn_fused_kernel");
                return 0;
            }
        }
    }
    if (second) {          // the two-kernel path is not grouped: one problem after the other (they share the partials workspace)
        int rc = launch_groupnorm(x, ld_in, gamma, beta, eps, silu, y, ld_out, batch, hw, C, groups, partials, stream, nullptr, two_kernel_min_hw);
        if (rc) return rc;
        return launch_groupnorm(second->x, ld_in, second->gamma, second->beta, eps, silu, second->y, ld_out, batch, hw, C, groups, partials, stream, nullptr, two_kernel_min_hw);
    }
    int nchunks = 0;
    int rc0 = launch_gn_stats(x, ld_in, batch, hw, C, groups, partials, stream, &nchunks);
    if (rc0) return rc0;
    const int V = C / 8;
    int P = 256 / V;
    if (P < 1) P = 1;
    const int threads = V * P;
    const int rows_per_chunk = (hw + nchunks - 1) / nchunks;
    dim3 grid(nchunks, batch);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(threads), 0, stream, x, ld_in, y, ld_out, gamma, beta, eps, silu,
                       hw, C, groups, rows_per_chunk, nchunks, partials);
    MKD_LAUNCH_CHECK("gn_apply_kernel");
    return 0;
}

// geometry of the slab-fed kernel: group chunk with a channel span that is a multiple of 8, at most 16 vectors per thread
static bool gn_slab_shape(int hw, int C, int* gpb_out, int* nt_out, int* per_out) {
    const int groups = 32;
    if (C % groups) return false;
    const int cg = C / groups;
    if (cg < 8) return false;                                   // (register statistics path; smaller groups keep the two kernels)
    int gpb = 1;
    while (gpb <= groups && ((gpb * cg) % 8 || groups % gpb)) ++gpb;
    if (gpb > groups) return false;
    const int V = gpb * cg / 8;
    if (V > 256) return false;
    auto per_of = [&](int t) { const int P = t / V; return (hw + P - 1) / P; };
    int nt, per;
    if (per_of(256) <= 16) { nt = 256; per = per_of(256); }
    else if (per_of(512) <= 16) { nt = 512; per = per_of(512); }
    else return false;                     // (1024 threads would leave 128 registers for 16 slab loads in flight: spills)
    *gpb_out = gpb; *nt_out = nt; *per_out = per;
    return true;
}
bool gn_from_slabs_supported(int batch, int hw, int C) {
    int gpb, nt, per;
    return batch > 0 && C % 8 == 0 && gn_slab_shape(hw, C, &gpb, &nt, &per);
}

static int slab_args(const GemmArgs& a, int batch, int hw, int ld_out, SlabGn* out) {
    if (a.splitk < 2 || !a.ws || a.M != batch * hw || a.N % 8 || ld_out % 8 || a.out_f32 || a.act != 0 || (a.R && a.ldr % 8) ||
        (a.C && a.ldc % 8) || (a.rowbias && a.rows_per_batch != hw))
        return mkd_fail(-1, "gn_from_slabs: needs split-K slabs of a plain bf16 GEMM with one row bias per sample");
    SlabGn sg;
    sg.ws = a.ws; sg.splitk = a.splitk; sg.M = a.M; sg.N = a.N; sg.bias = a.bias; sg.rowbias = a.rowbias; sg.ldrb = a.ldrb;
    sg.rpb = a.rows_per_batch > 0 ? a.rows_per_batch : 1; sg.scale = a.scale; sg.R = a.R; sg.ldr = a.ldr; sg.raw = (bf16_t*)a.C; sg.ldraw = a.ldc;
    *out = sg;
    return 0;
}

int launch_gn_from_slabs(const GemmArgs& a, const float* gamma, const float* beta, float eps, int silu, bf16_t* y, int ld_out,
                         int batch, int hw, hipStream_t stream, const GemmArgs* a2, const NormIo* second) {
    int gpb, nt, per;
    if (!gn_slab_shape(hw, a.N, &gpb, &nt, &per)) return mkd_fail(-4, "gn_from_slabs: geometry does not fit the single-pass kernel");
    if ((a2 != nullptr) != (second != nullptr)) return mkd_fail(-1, "gn_from_slabs: a grouped launch needs both the second GEMM and the second GroupNorm");
    Pair<SlabGn> sg;
    Pair<NormIo> io;
    int rc = slab_args(a, batch, hw, ld_out, &sg.g[0]); if (rc) return rc;
    io.g[0] = NormIo{nullptr, y, gamma, beta};
    MKD_PAIR_SET2(sg, sg.g[0]); MKD_PAIR_SET2(io, io.g[0]);
    if (a2) {
        if (a2->M != a.M || a2->N != a.N || a2->splitk != a.splitk || a2->ws == a.ws) return mkd_fail(-1, "gn_from_slabs: grouped problems must share the geometry and own their slabs");
        rc = slab_args(*a2, batch, hw, ld_out, &sg.g[MKD_PAIR_N - 1]); if (rc) return rc;
        MKD_PAIR_SET2(io, *second);
    }
    const size_t lds = (size_t)(2 * (nt / 64) * gpb + 2 * gpb) * sizeof(float);
    dim3 grid(32 / gpb, batch, a2 ? 2 : 1);
#define MKD_GNS_LAUNCH(NVV) hipLaunchKernelGGL(gn_slab_kernel<NVV>, grid, dim3(nt), lds, stream, sg, io, ld_out, eps, silu, hw, a.N, 32, gpb)
    if (per <= 4) MKD_GNS_LAUNCH(4);
    else if (per <= 8) MKD_GNS_LAUNCH(8);
    else MKD_GNS_LAUNCH(16);
#undef MKD_GNS_LAUNCH
    MKD_LAUNCH_CHECK("gn_slab_kernel");
    return 0;
}

static int gn_thread_shape(int cols, int* threads) {
    const int V = cols / 8;
    if (cols % 8 || V < 1 || V > 320) return mkd_fail(-4, "groupnorm (fused statistics): channel count must be a multiple of 8 and <= 2560");
    const int P = V >= 256 ? 1 : 256 / V;
    *threads = V * P;
    return 0;
}

int launch_gn_apply_stats(const bf16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu, bf16_t* y, int ld_out,
                          int batch, int hw, int C, const long long* gstat, hipStream_t stream) {
    if (C % 32 || ld_in % 8 || ld_out % 8) return mkd_fail(-1, "groupnorm: C must be a multiple of 32 (and of 8), ld of 8");
    if (!gstat) return mkd_fail(-1, "groupnorm: statistics missing");
    int threads;
    int rc = gn_thread_shape(C, &threads); if (rc) return rc;
    const int P = threads / (C / 8);
    int rpb = (int)(((long long)batch * hw + 511) / 512);           // ~512 blocks: two per CU
    rpb = ((rpb + P - 1) / P) * P;
    if (rpb < P) rpb = P;
    dim3 grid((hw + rpb - 1) / rpb, batch);
    hipLaunchKernelGGL(gn_apply_stats_kernel, grid, dim3(threads), 0, stream, x, ld_in, y, ld_out, gamma, beta, eps, silu, hw, C, rpb, gstat);
    MKD_LAUNCH_CHECK("gn_apply_stats_kernel");
    return 0;
}

int launch_gn_colstats(const bf16_t* x, int ld, int batch, int hw, int ncols, int cg, int coff, long long* gstat, hipStream_t stream) {
    if (ld % 8 || cg <= 0 || coff < 0 || (coff + ncols + cg - 1) / cg > 32) return mkd_fail(-1, "gn_colstats: bad geometry");
    int threads;
    int rc = gn_thread_shape(ncols, &threads); if (rc) return rc;
    const int P = threads / (ncols / 8);
    int rpb = (int)(((long long)batch * hw + 255) / 256);
    rpb = ((rpb + P - 1) / P) * P;
    if (rpb < 4 * P) rpb = 4 * P;
    dim3 grid((hw + rpb - 1) / rpb, batch);
    hipLaunchKernelGGL(gn_colstats_kernel, grid, dim3(threads), 0, stream, x, ld, hw, ncols, rpb, cg, coff, gstat);
    MKD_LAUNCH_CHECK("gn_colstats_kernel");
    return 0;
}

int launch_layernorm(const bf16_t* x, const float* gamma, const float* beta, float eps, bf16_t* y,
                     int rows, int d, hipStream_t stream, int ldx, const NormIo* second) {
    if (ldx <= 0) ldx = d;
    if (d % 8 || ldx % 8) return mkd_fail(-1, "layernorm: d and the row stride must be multiples of 8");
    const int V = d / 8;
    Pair<NormIo> io;
    io.g[0] = NormIo{x, y, gamma, beta};
    MKD_PAIR_SET2(io, second ? *second : io.g[0]);
    dim3 grid((rows + 3) / 4, second ? 2 : 1);
    if (V <= 64)       hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, stream, io, eps, rows, d, ldx);
    else if (V <= 128) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, stream, io, eps, rows, d, ldx);
    else if (V <= 192) hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, stream, io, eps, rows, d, ldx);
    else if (V <= 256) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, stream, io, eps, rows, d, ldx);
    else return mkd_fail(-4, "layernorm: d > 2048 unsupported");
    MKD_LAUNCH_CHECK("layernorm_kernel");
    return 0;
}

int launch_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K,
                          bf16_t* w_out, int dst_row0, int dst_row_mul, float* s_out, float* b_out, hipStream_t stream) {
    hipLaunchKernelGGL(fold_layernorm_kernel, dim3(N), dim3(256), 0, stream, w, gamma, beta, bias, N, K, w_out, dst_row0,
                       dst_row_mul, s_out, b_out);
    MKD_LAUNCH_CHECK("fold_layernorm_kernel");
    return 0;
}
