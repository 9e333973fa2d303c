// Per-workgroup streaming rate of the two ways to stage an operand tile in LDS, fixed launch overhead cancelled (incremental rate
// between 1 MB and 4 MB per workgroup, as stream_rate2):
//   dma : global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), DEPTH tiles of 16 KB in flight per workgroup
//   reg : global_load_dwordx4 into VGPRs, DEPTH tiles in flight in registers, ds_write_b128 into a 2-slot LDS buffer
//   regonly : the same loads, consumed in registers (no LDS traffic at all)
// one 256-thread workgroup per CU (256 workgroups), or 2 / 4 per CU; "shared": every workgroup reads the same 1 MB window (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef float __attribute__((ext_vector_type(4))) f4;

template <int DEPTH>
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, size_t stride_per_wg, size_t window, int steps, float* sink) {
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

// MODE 0: registers -> LDS (ds_write_b128) + one read per step; MODE 1: registers only
template <int DEPTH, int MODE>
__global__ __launch_bounds__(256) void reg_kernel(const char* __restrict__ src, size_t stride_per_wg, size_t window, int steps, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 16384];
    const int tid = threadIdx.x;
    const char* base = src + (size_t)blockIdx.x * stride_per_wg + (size_t)tid * 16;
    f4 r[DEPTH][4];
    auto issue = [&](int s, f4 (&dst)[4]) {
        const size_t off = ((size_t)s * 16384) % window;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = __builtin_nontemporal_load((const f4*)(base + off + (size_t)i * 4096));
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, r[d]);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    // steps is a multiple of DEPTH: the ring of register tiles is addressed statically
    for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = s0 + d;
            issue(s + DEPTH - 1 < steps ? s + DEPTH - 1 : s, r[(d + DEPTH - 1) % DEPTH]);
            if (MODE == 0) {
                char* slot = lds + (s & 1) * 16384;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f4*)(slot + i * 4096 + tid * 16) = r[d][i];
                __syncthreads();
                acc += *(const f4*)(slot + ((tid * 16 + 4096) & 16383));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += r[d][i];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <typename F>
static void run(const char* name, F launch, int nwg, bool shared, size_t total) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double t_prev = 0, b_prev = 0;
    for (int mb : {1, 4}) {
        const size_t S = (size_t)mb << 20;
        if (!shared && (size_t)nwg * S > total) continue;
        const int steps = (int)(S / 16384);
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(e0);
            launch(nwg, shared ? (size_t)0 : S, shared ? (size_t)(1 << 20) : S, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double t = best * 1e-3, b = (double)S;
        if (t_prev > 0)
            printf("%-14s %s %4d WGs: incremental %6.1f GB/s per WG, %5.2f TB/s all\n", name, shared ? "shared (L2 hits)" : "private (misses) ", nwg,
                   (b - b_prev) / (t - t_prev) * 1e-9, (b - b_prev) / (t - t_prev) * 1e-12 * nwg);
        t_prev = t; b_prev = b;
    }
}

int main() {
    const size_t total = 3ull << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64); hipMemset(buf, 1, total);
    for (int shared = 1; shared >= 0; --shared)
        for (int nwg : {256, 512, 1024}) {
#define L(K) [&](int n, size_t st, size_t win, int steps) { hipLaunchKernelGGL(K, dim3(n), dim3(256), 0, 0, buf, st, win, steps, sink); }
            run("dma depth 4", L((dma_kernel<4>)), nwg, shared, total);
            run("dma depth 8", L((dma_kernel<8>)), nwg, shared, total);
            run("reg depth 2", L((reg_kernel<2, 0>)), nwg, shared, total);
            run("reg depth 4", L((reg_kernel<4, 0>)), nwg, shared, total);
            run("reg depth 8", L((reg_kernel<8, 0>)), nwg, shared, total);
            run("regonly d4", L((reg_kernel<4, 1>)), nwg, shared, total);
            run("regonly d8", L((reg_kernel<8, 1>)), nwg, shared, total);
        }
    return 0;
}
