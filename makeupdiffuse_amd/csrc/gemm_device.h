// Device-side pieces shared by the implicit-GEMM kernels (kernels_gemm.hip, kernels_conv.hip).
#pragma once
#include "mkd_common.h"

namespace mkdk {

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
