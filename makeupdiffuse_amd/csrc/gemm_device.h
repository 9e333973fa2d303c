// Device-side pieces shared by the implicit-GEMM kernels (kernels_gemm.hip, kernels_conv.hip).
#pragma once
#include "mkd_common.h"

namespace mkdk {

// XCD-aware tile order.  The dispatcher deals workgroups to the 8 XCDs round-robin in launch order (x fastest), so workgroup `lid`
// runs on XCD lid % 8 as the (lid / 8)-th workgroup there.  Give every XCD one CONTIGUOUS run of the tile sequence instead: tiles that
// share an operand tile then sit in the same L2 at the same time.  Uniform (scalar) arithmetic only.
// gz / grp: a grouped launch stacks the second problem's grid above the first one's in z (gz = z extent of one problem); every
// problem is ordered on its own.
__device__ __forceinline__ void xcd_tile_order(int mode, int gz, int grp, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = (int)blockIdx.z - grp * gz;
    if (mode == 0) return;
    const int gx = gridDim.x, gy = gridDim.y, plane = gx * gy, total = plane * gz;
    const int lid = bx + gx * by + plane * bz;
    const int xcd = lid & 7, idx = lid >> 3, q = total >> 3, r = total & 7;
    const int t = xcd * q + (xcd < r ? xcd : r) + idx;       // XCD k owns tiles [k*q + min(k, r), ...): q (+1 for k < r) of them
    bz = t / plane;
    const int rem = t - bz * plane;
    if (mode == 1) { by = rem / gx; bx = rem - by * gx; }    // x fastest: consecutive tiles share the W tile
    else { bx = rem / gy; by = rem - bx * gy; }              // y fastest: consecutive tiles share the A tile
}

// host side: the 2-entry argument table of a (possibly grouped) launch; gz = z extent of one problem's grid
inline GemmArgs2 gemm_pack2(const GemmArgs& a, const GemmArgs* b, int gz) {
    GemmArgs2 r;
    r.g[0] = a; r.g[0].gz = gz;
    if (MKD_PAIR_N > 1) { r.g[MKD_PAIR_N - 1] = b ? *b : a; r.g[MKD_PAIR_N - 1].gz = gz; }
    return r;
}

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

// branch-free select of a load source, made opaque so the compiler keeps ONE global_load_lds per call site
// (a duplicated load would break the exact loads-per-tile count the counted vmcnt waits rely on).
__device__ __forceinline__ const void* select_src(const void* real, const void* zero, bool ok) {
    unsigned long long v = ok ? (unsigned long long)real : (unsigned long long)zero;
    asm volatile("" : "+v"(v));
    return (const void*)v;
}

// (pointers first, then 4-byte fields, no padding holes: keeps the struct in registers after inlining)
struct Epilogue {
    const float* bias; const float* rowbias; const bf16_t* R; void* C; const float* ln_s;
    int ldrb; int rpb; int ldr; int ldc; float scale; int act; int out_f32; int pad_;
};

// ---- GroupNorm statistics emitted by the PRODUCER of a tensor ---------------------------------------------------------
// A GroupNorm needs (sum, sum of squares) over all pixels and the group's channels of one sample: a full pass over a tensor
// that the kernel writing it already holds in registers.  Producers therefore add their share into gstat[sample][group][2],
// 64-bit FIXED-POINT integers (sum * 2^24, sumsq * 2^18): integer addition is associative, so the totals are bit-identical
// whatever order the workgroups (and the device-scope atomics) arrive in - float atomics would make every GroupNorm, and
// with it every evaluation, run-to-run different.  The absolute quantisation error of a partial is 2^-25 (sum) / 2^-19
// (sumsq) per >= 16 values, far below the GroupNorm eps (1e-5 / 1e-6); the range is |x| < ~1e4 for 2^17 values per group.
constexpr float GN_FIX_SUM = 16777216.0f;      // 2^24
constexpr float GN_FIX_SQ = 262144.0f;         // 2^18
constexpr int GN_GROUPS = 32;

__device__ __forceinline__ void gn_atomic_add(long long* p, long long v) {
#ifdef MKD_EXP_NO_ATOMIC
    return;
#endif
    atomicAdd((unsigned long long*)p, (unsigned long long)v);       // two's complement: wraps correctly for negative sums
}

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4), result in every lane of the row; fixed order: deterministic
__device__ __forceinline__ float dpp_row_sum(float v) {
#define MKD_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    MKD_DPP_ADD(0xB1);       // quad_perm [1,0,3,2]
    MKD_DPP_ADD(0x4E);       // quad_perm [2,3,0,1]
    MKD_DPP_ADD(0x141);      // row_half_mirror
    MKD_DPP_ADD(0x140);      // row_mirror
#undef MKD_DPP_ADD
    return v;
}

// FAST path of the tile statistics, one 16-column fragment at a time: every row of the wave belongs to ONE sample (segment
// `seg` of the tile), so the lane's sums over the wave's row fragments (cs / cq: its 4 channels of this column fragment) only
// need a 16-lane row reduction.  All 16 lanes of a row then hold the same 8 totals; lane k < 8 of the row converts total k to
// fixed point and adds it to the tile's LDS accumulator acc[seg][group] (zeroed at kernel start).  No LDS staging, no barrier.
// frag_col0: tile column of the fragment's first channel.
__device__ __forceinline__ void gn_wave_stats(const float (&cs)[4], const float (&cq)[4], int frow, int fq, int frag_col0,
                                              int valid_cols, int cg, int cfirst, int seg, int ngl, long long* acc) {
    float tot[8];
#ifdef MKD_EXP_NO_WAVESTATS
    return;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) { tot[2 * j] = dpp_row_sum(cs[j]); tot[2 * j + 1] = dpp_row_sum(cq[j]); }
    float val = tot[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) val = (frow == i) ? tot[i] : val;
    const int c = frag_col0 + 4 * fq + (frow >> 1);
    if (frow < 8 && c < valid_cols) {
        const int gl = (cfirst + c) / cg - cfirst / cg;
        const long long f = __float2ll_rn(val * ((frow & 1) ? GN_FIX_SQ : GN_FIX_SUM));
        if (f) gn_atomic_add(acc + (size_t)(seg * ngl + gl) * 2 + (frow & 1), f);
    }
}

// ... after a block barrier: one device-scope atomic per (segment, group) entry of the tile's accumulator
__device__ __forceinline__ void gn_acc_flush(const long long* acc, int nseg, int ngl, int b_first, int g_first, int tid, int nthreads,
                                             long long* gstat) {
#ifdef MKD_EXP_NO_FLUSH
    return;
#endif
    for (int i = tid; i < nseg * ngl; i += nthreads) {
        const long long a = acc[i * 2], q = acc[i * 2 + 1];
        long long* dst = gstat + ((size_t)(b_first + i / ngl) * GN_GROUPS + g_first + i % ngl) * 2;
        if (a) gn_atomic_add(dst, a);
        if (q) gn_atomic_add(dst + 1, q);
    }
}

// Column statistics of one output tile that the workgroup has staged in LDS as bf16 [rows][TS] (exactly the values it
// stored to memory).  Thread (chunk, c) walks its rows of column c in order; rows are grouped into SEGMENTS of seg_period
// consecutive rows (segment s of the tile = sample b_first + s; row 0 sits at position phase0 of its segment).  Per
// (segment, group) the per-thread partials are added in LDS (integers: order-free), then ONE global atomic per entry.
// acc: LDS scratch for nseg * ngl pairs (acc_cap pairs available; tiny geometries that exceed it go straight to global).
// Must be called by ALL threads of the block (barriers inside).
__device__ __forceinline__ void gn_tile_stats(const uint16_t* tile, int TS, int TM, int TN, int nthreads, int tid,
                                              int valid_rows, int valid_cols, int seg_period, int phase0, int b_first,
                                              int cg, int cfirst, long long* acc, int acc_cap, long long* gstat) {
    const int nseg = (phase0 + valid_rows - 1) / seg_period + 1;
    const int g_first = cfirst / cg;
    const int ngl = (cfirst + valid_cols - 1) / cg - g_first + 1;
    const bool use_lds = nseg * ngl <= acc_cap;
    if (use_lds)
        for (int i = tid; i < nseg * ngl * 2; i += nthreads) acc[i] = 0;
    __syncthreads();                                   // tile written, acc zeroed
    const int R = nthreads / TN;                       // row chunks (TN <= 160 < 256 threads)
    const int rows_per = (TM + R - 1) / R;
    const int c = tid % TN, chunk = tid / TN;
    if (chunk < R && c < valid_cols) {
        const int lr0 = chunk * rows_per;
        const int lr1 = min(valid_rows, lr0 + rows_per);
        if (lr0 < lr1) {
            const int gl = (cfirst + c) / cg - g_first;
            int seg = (phase0 + lr0) / seg_period;
            int left = seg_period - (phase0 + lr0 - seg * seg_period);
            float sm = 0.f, sq = 0.f;
            auto flush = [&]() {
                const long long a = __float2ll_rn(sm * GN_FIX_SUM), q = __float2ll_rn(sq * GN_FIX_SQ);
                long long* dst = use_lds ? acc + (size_t)(seg * ngl + gl) * 2 : gstat + ((size_t)(b_first + seg) * GN_GROUPS + g_first + gl) * 2;
                if (a) gn_atomic_add(dst, a);
                if (q) gn_atomic_add(dst + 1, q);
            };
            for (int lr = lr0; lr < lr1; ++lr) {
                const float v = bf16_to_f32(tile[lr * TS + c]);
                sm += v; sq += v * v;
                if (--left == 0) { flush(); sm = 0.f; sq = 0.f; ++seg; left = seg_period; }
            }
            if (left != seg_period) flush();
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = tid; i < nseg * ngl; i += nthreads) {
            const long long a = acc[i * 2], q = acc[i * 2 + 1];
            long long* dst = gstat + ((size_t)(b_first + i / ngl) * GN_GROUPS + g_first + i % ngl) * 2;
            if (a) gn_atomic_add(dst, a);
            if (q) gn_atomic_add(dst + 1, q);
        }
    }
}

__device__ __forceinline__ Epilogue make_epilogue(const GemmArgs& p) {
    Epilogue e;
    e.bias = p.bias; e.rowbias = p.rowbias; e.R = p.R; e.C = p.C; e.ln_s = p.ln_s;
    e.ldrb = p.ldrb; e.rpb = p.rows_per_batch; e.ldr = p.ldr; e.ldc = p.ldc; e.scale = p.scale; e.act = p.act;
    e.out_f32 = p.out_f32; e.pad_ = 0;
    return e;
}

// fused LayerNorm: acc = sum_k x_k W'_k on RAW rows; LN(x).W' = rstd * (acc - mu * rowsum(W'))
__device__ __forceinline__ f32x4 ln_correct(const float* ln_s, int n, f32x4 v, float mu, float rstd) {
    const f32x4 s = *(const f32x4*)(ln_s + n);
    return (v - s * mu) * rstd;
}

__device__ __forceinline__ f32x4 epilogue_value(const Epilogue& e, int m, int n, f32x4 v) {
    if (e.bias) v += *(const f32x4*)(e.bias + n);
    if (e.rowbias) v += *(const f32x4*)(e.rowbias + (size_t)(m / e.rpb) * e.ldrb + n);
    v *= e.scale;
    if (e.R) {
        const U16x4 r = *(const U16x4*)(e.R + (size_t)m * e.ldr + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bf16_to_f32(r.v[j]);
    }
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    }
    if (e.act == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = quick_gelu_f(v[j]);
    }
    return v;
}

// same with bias / residual fragments that were loaded before the K loop
__device__ __forceinline__ f32x4 epilogue_value_pre(const Epilogue& e, int m, int n, f32x4 v, f32x4 bias, U16x4 res) {
    v += bias;
    if (e.rowbias) v += *(const f32x4*)(e.rowbias + (size_t)(m / e.rpb) * e.ldrb + n);
    v *= e.scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += bf16_to_f32(res.v[j]);
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
    }
    if (e.act == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = quick_gelu_f(v[j]);
    }
    return v;
}

// stores v (already through epilogue_value); returns the values as the consumer will read them (bf16-rounded)
__device__ __forceinline__ f32x4 epilogue_write(const Epilogue& e, int m, int n, f32x4 v) {
    if (e.act == 2) {
        // GEGLU with interleaved (value, gate) weight rows: columns (n, n+1) and (n+2, n+3) are two
        // (a, g) pairs -> out[m, n/2 .. n/2+1] = a * gelu_erf(g); the output has N/2 columns
        const uint32_t o = (uint32_t)f32_to_bf16(v[0] * gelu_erf_f(v[1])) | ((uint32_t)f32_to_bf16(v[2] * gelu_erf_f(v[3])) << 16);
        *(uint32_t*)((bf16_t*)e.C + (size_t)m * e.ldc + (n >> 1)) = o;
        return v;
    }
    if (e.out_f32) {
        *(f32x4*)((float*)e.C + (size_t)m * e.ldc + n) = v;
        return v;
    }
    U16x4 o;
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o.v[j] = f32_to_bf16(v[j]); r[j] = bf16_to_f32(o.v[j]); }
    *(U16x4*)((bf16_t*)e.C + (size_t)m * e.ldc + n) = o;
    return r;
}

// bf16 store that also hands back the stored bits (GroupNorm statistics are taken of exactly what the consumer will read)
__device__ __forceinline__ U16x4 epilogue_write_bits(const Epilogue& e, int m, int n, f32x4 v) {
    U16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = f32_to_bf16(v[j]);
    *(U16x4*)((bf16_t*)e.C + (size_t)m * e.ldc + n) = o;
    return o;
}

// Straight-line form of epilogue_value_pre + epilogue_write for the common case (bf16 output, no activation; bias and residual
// fragments preloaded, zeros when absent; RB: a per-sample row-bias fragment, valid when all rows of the wave lie in ONE sample): the
// same operations in the same order - bit-identical results - without the five uniform branches, the integer division of the row
// bias and the load-then-wait per fragment that the general form carries.  With one wave per SIMD the epilogue is serial code: the
// general form costs 0.35 us per accumulator fragment of the wave (tools/exp_gemm_fixed.py: 7.1 us for ONE 128x128 tile and one
// K-step against 1.8 us for a 32x32 one), i.e. more than the K loop of most layers.
template <bool RB>
__device__ __forceinline__ void epilogue_fast_store(const Epilogue& e, int m, int n, f32x4 v, f32x4 bias, f32x4 rb, U16x4 res) {
    v += bias;
    if (RB) v += rb;
    v *= e.scale;
    U16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += bf16_to_f32(res.v[j]); o.v[j] = f32_to_bf16(v[j]); }
    *(U16x4*)((bf16_t*)e.C + (size_t)m * e.ldc + n) = o;
}

__device__ __forceinline__ void epilogue_store(const Epilogue e, int m, int n, f32x4 v) {
    epilogue_write(e, m, n, epilogue_value(e, m, n, v));
}

// Row statistics of a fused LayerNorm: the producer GEMM wrote [slots][M][2] partial (sum, sum of squares).
// The 4 lanes that share a row (k-quarters fq = 0..3) split the slots (slot = 4*i + fq); the loads are ISSUED
// early (after the prologue tiles) and only CONSUMED after the K loop, so their latency is hidden.
// loads per lane per row: SL in {1, 3, 5} covers up to 4 / 12 / 20 slots (template parameter of the kernel)

template <int SL>
__device__ __forceinline__ void ln_stats_issue(const float* stat_in, int slots, int M, int m, int fq, float2 (&t)[SL]) {
    // unconditional loads from clamped addresses (a select on the loaded value would force an immediate wait);
    // out-of-range entries are discarded in ln_stats_finish
    const int mc = m < M ? m : M - 1;
    const float* base = stat_in + (size_t)mc * 2;
    const size_t step = (size_t)M * 2;
#pragma unroll
    for (int i = 0; i < SL; ++i) {               // ALWAYS SL loads: the counted vmcnt waits rely on the count
        const int sl = 4 * i + fq;
        t[i] = *(const float2*)(base + (size_t)(sl < slots ? sl : slots - 1) * step);
    }
}

template <int SL>
__device__ __forceinline__ void ln_stats_finish(const float2 (&t)[SL], int slots, int fq, int K, float eps, float& mu,
                                                float& rstd) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < SL; ++i)
        if (4 * i + fq < slots) { a += t[i].x; q += t[i].y; }
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);     // fixed order: deterministic
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float mean = a / (float)K;
    float var = q / (float)K - mean * mean;
    var = var < 0.f ? 0.f : var;
    mu = mean; rstd = rsqrtf(var + eps);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


}  // namespace mkdk
