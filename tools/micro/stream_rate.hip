// Per-workgroup streaming rate: how fast can ONE workgroup (256 threads) pull a contiguous-row operand tile stream,
// (a) global_load_dwordx4 -> VGPR, (b) global_load_lds_dwordx4 -> LDS, with N workgroups running (80 / 256 / 1024)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, size_t bytes_per_wg, float* sink, int row_stride, int rows) {
    __shared__ __attribute__((aligned(16))) char lds[DEPTH * 16384];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // tile step = 128 rows x 128 B (16 KB): thread -> (row = tid/8 + 32*i, chunk = tid%8), rows row_stride bytes apart
    const char* base = src + (size_t)blockIdx.x * (row_stride == 128 ? bytes_per_wg : (size_t)rows * row_stride);
    const int steps = (int)(bytes_per_wg / 16384);
    const size_t step_bytes = row_stride == 128 ? 16384 : 128;      // pre-tiled: next step = next 16 KB block
    f32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        for (int s = 0; s < steps; ++s) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(base + (size_t)((tid >> 3) + 32 * i) * row_stride + (size_t)s * step_bytes + (tid & 7) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += v[i];
        }
    } else {
        // ring of DEPTH stages, counted waits: 4 loads per wave-lane per stage
        auto issue = [&](int s) {
            char* dst = lds + (s % DEPTH) * 16384;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char* g = base + (size_t)((tid >> 3) + 32 * i) * row_stride + (size_t)s * step_bytes + (tid & 7) * 16;
                __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(dst + (w * 4 + i) * 1024), 16, 0, 0);
            }
        };
        for (int s = 0; s < DEPTH - 1 && s < steps; ++s) issue(s);
        for (int s = 0; s < steps; ++s) {
            if (s + DEPTH - 1 < steps) { issue(s + DEPTH - 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc += *(const f32x4*)(lds + (s % DEPTH) * 16384 + tid * 16);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int NT>
__global__ __launch_bounds__(NT) void stream_vgpr_nt(const char* __restrict__ src, size_t bytes_per_wg, float* sink, int row_stride, int rows) {
    const int tid = threadIdx.x;
    const char* base = src + (size_t)blockIdx.x * (row_stride == 128 ? bytes_per_wg : (size_t)rows * row_stride);
    const int steps = (int)(bytes_per_wg / 16384);
    const size_t step_bytes = row_stride == 128 ? 16384 : 128;
    constexpr int RPT = 128 / (NT / 8);
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        f32x4 v[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) v[i] = *(const f32x4*)(base + (size_t)((tid >> 3) + (NT / 8) * i) * row_stride + (size_t)s * step_bytes + (tid & 7) * 16);
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc += v[i];
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

int main() {
    const size_t total = 1ull << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 64); hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rows = 128;
    for (int row_stride : {2560, 128, 640, 23040}) {      // 128: each 16 KB step fully contiguous (a pre-tiled operand)
        for (int nwg : {80, 256}) {
            // one pass over the WG's 128 rows (2560 B of each row are read; with stride 128 the steps are consecutive 16 KB blocks)
            const size_t bytes_per_wg = (size_t)rows * 2560;
            if ((size_t)nwg * (size_t)rows * (row_stride > 2560 ? row_stride : 2560) > total) continue;
            auto run = [&](int mode, int depth) {
                float best = 1e9;
                for (int r = 0; r < 5; ++r) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL((stream_kernel<0, 1>), dim3(nwg), dim3(256), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    else if (depth == 2) hipLaunchKernelGGL((stream_kernel<1, 2>), dim3(nwg), dim3(256), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    else if (depth == 4) hipLaunchKernelGGL((stream_kernel<1, 4>), dim3(nwg), dim3(256), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    else hipLaunchKernelGGL((stream_kernel<1, 8>), dim3(nwg), dim3(256), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                return best;
            };
            const float t0 = run(0, 1), t2 = run(1, 2), t4 = run(1, 4), t8 = run(1, 8);
            auto run_nt = [&](int nt) {
                float best = 1e9;
                for (int r = 0; r < 5; ++r) {
                    hipEventRecord(e0);
                    if (nt == 128) hipLaunchKernelGGL((stream_vgpr_nt<128>), dim3(nwg), dim3(128), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    else if (nt == 512) hipLaunchKernelGGL((stream_vgpr_nt<512>), dim3(nwg), dim3(512), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    else hipLaunchKernelGGL((stream_vgpr_nt<1024>), dim3(nwg), dim3(1024), 0, 0, buf, bytes_per_wg, sink, row_stride, rows);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                return best;
            };
            const float a = run_nt(128), b = run_nt(512), c = run_nt(1024);
            printf("   threads/WG: 128 -> %.1f us (%.0f GB/s/WG)   512 -> %.1f us (%.0f)   1024 -> %.1f us (%.0f)\n", a * 1e3, bytes_per_wg / 1e9 / (a * 1e-3), b * 1e3, bytes_per_wg / 1e9 / (b * 1e-3), c * 1e3, bytes_per_wg / 1e9 / (c * 1e-3));
            const double gb = bytes_per_wg / 1e9;
            printf("row_stride %d, %4d WGs x %zu KB: vgpr %.1f us (%.0f GB/s/WG)  glds d2 %.1f us (%.0f)  d4 %.1f us (%.0f)  d8 %.1f us (%.0f)\n", row_stride, nwg,
                   bytes_per_wg >> 10, t0 * 1e3, gb / (t0 * 1e-3), t2 * 1e3, gb / (t2 * 1e-3), t4 * 1e3, gb / (t4 * 1e-3), t8 * 1e3, gb / (t8 * 1e-3));
        }
    }
    return 0;
}
