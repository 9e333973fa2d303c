// Per-CU streaming rate, fixed launch overhead cancelled: every workgroup (256 threads, one per CU when nwg <= 256) streams S bytes
// with global_load_lds_dwordx4 through a DEPTH-stage LDS ring; S = 1, 2, 4, 8 MB; the INCREMENTAL rate (d bytes / d time) is the
// per-workgroup streaming rate.  (a) private: every workgroup its own region (L2 misses: Infinity Cache / HBM);
// (b) shared: all workgroups read the SAME 1 MB window round and round (L2 hits after the first touch per XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, size_t stride_per_wg, size_t window, int steps, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[DEPTH * 16384];
    const int tid = threadIdx.x, w = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * stride_per_wg;
    auto issue = [&](int s) {
        char* dst = lds + (s % DEPTH) * 16384;
        const size_t off = ((size_t)s * 16384) % window;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + off + (size_t)(w * 4 + i) * 1024 + (tid & 63) * 16), (lds_void*)(dst + (w * 4 + i) * 1024), 16, 0, 0);
    };
    for (int s = 0; s < DEPTH - 1 && s < steps; ++s) issue(s);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (s + DEPTH - 1 < steps) { issue(s + DEPTH - 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += ((const float*)(lds + (s % DEPTH) * 16384))[tid];
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const size_t total = 3ull << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64); hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int depth : {2, 4, 8})
    for (int shared = 0; shared < 2; ++shared)
        for (int nwg : {64, 256, 512, 1024}) {
            double t_prev = 0, b_prev = 0;
            for (int mb : {1, 4}) {
                const size_t S = (size_t)mb << 20;
                if (!shared && (size_t)nwg * S > total) continue;
                const int steps = (int)(S / 16384);
                float best = 1e9;
                for (int r = 0; r < 4; ++r) {
                    hipEventRecord(e0);
                    if (depth == 2) hipLaunchKernelGGL((stream_kernel<2>), dim3(nwg), dim3(256), 0, 0, buf, shared ? (size_t)0 : S, shared ? (size_t)(1 << 20) : S, steps, sink);
                    else if (depth == 4) hipLaunchKernelGGL((stream_kernel<4>), dim3(nwg), dim3(256), 0, 0, buf, shared ? (size_t)0 : S, shared ? (size_t)(1 << 20) : S, steps, sink);
                    else hipLaunchKernelGGL((stream_kernel<8>), dim3(nwg), dim3(256), 0, 0, buf, shared ? (size_t)0 : S, shared ? (size_t)(1 << 20) : S, steps, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                const double t = best * 1e-3, b = (double)S;
                if (t_prev > 0)
                    printf("depth %d %s %4d WGs, %d MB each: %.1f us; incremental %.1f GB/s per WG, %.2f TB/s all\n", depth, shared ? "shared (L2 hits) " : "private (misses)  ", nwg, mb, best * 1e3,
                           (b - b_prev) / (t - t_prev) * 1e-9, (b - b_prev) / (t - t_prev) * 1e-12 * nwg);
                t_prev = t; b_prev = b;
            }
        }
    return 0;
}
