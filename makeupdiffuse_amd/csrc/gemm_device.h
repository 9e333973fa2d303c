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

struct Epilogue {
    const float* bias; const float* rowbias; int ldrb; int rpb;
    const bf16_t* R; int ldr; float scale; int act; void* C; int ldc; int out_f32;
};

__device__ __forceinline__ Epilogue make_epilogue(const GemmArgs& p) {
    return Epilogue{p.bias, p.rowbias, p.ldrb, p.rows_per_batch, p.R, p.ldr, p.scale, p.act, p.C, p.ldc, p.out_f32};
}

__device__ __forceinline__ void epilogue_store(const Epilogue e, int m, int n, f32x4 v) {
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
    if (e.act == 2) {
        // GEGLU with interleaved (value, gate) weight rows: columns (n, n+1) and (n+2, n+3) are two
        // (a, g) pairs -> out[m, n/2 .. n/2+1] = a * gelu_erf(g); the output has N/2 columns
        const uint32_t o = (uint32_t)f32_to_bf16(v[0] * gelu_erf_f(v[1])) | ((uint32_t)f32_to_bf16(v[2] * gelu_erf_f(v[3])) << 16);
        *(uint32_t*)((bf16_t*)e.C + (size_t)m * e.ldc + (n >> 1)) = o;
        return;
    }
    if (e.out_f32) {
        *(f32x4*)((float*)e.C + (size_t)m * e.ldc + n) = v;
    } else {
        U16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = f32_to_bf16(v[j]);
        *(U16x4*)((bf16_t*)e.C + (size_t)m * e.ldc + n) = o;
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


}  // namespace mkdk
