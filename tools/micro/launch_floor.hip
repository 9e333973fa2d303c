// What does one dependent kernel launch cost on this machine?  A captured hipGraph of N kernels in one stream (every node depends on
// the previous one), replayed; time per node for: an empty kernel, an empty full-chip kernel, a kernel where every workgroup loads one
// line written by the previous node and stores one, and an element-wise pass over 5 MB (in -> out, ping-pong).  Also two such chains
// on two streams captured into one graph (fork / join), as the evaluation runs them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_line(const float* __restrict__ in, float* __restrict__ out) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 15);
    if (threadIdx.x < 16) out[i] = in[i] + 1.0f;
}
__global__ __launch_bounds__(256) void k_pass(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { float4 v = in[i]; v.x += 1.f; out[i] = v; }
}

typedef float __attribute__((ext_vector_type(4))) f4;
template <int NT>
__global__ __launch_bounds__(256) void k_pass_nt(const f4* __restrict__ in, f4* __restrict__ out, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        f4 v = (NT & 1) ? __builtin_nontemporal_load(in + i) : in[i];
        v.x += 1.f;
        if (NT & 2) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

int main() {
    const int N = 400;
    float *a, *b, *c, *d;
    const size_t bytes = 5u << 20;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int n4 = (int)(bytes / 16);
    for (int two = 0; two < 2; ++two)
    for (int mode = 0; mode < 8; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        if (two) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
        for (int i = 0; i < N; ++i) {
            for (int st = 0; st <= two; ++st) {
                hipStream_t s = st ? s1 : s0;
                float* in = st ? ((i & 1) ? d : c) : ((i & 1) ? b : a);
                float* out = st ? ((i & 1) ? c : d) : ((i & 1) ? a : b);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); break;
                    case 1: hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); break;
                    case 2: hipLaunchKernelGGL(k_line, dim3(256), dim3(256), 0, s, in, out); break;
                    case 3: hipLaunchKernelGGL(k_pass, dim3(1024), dim3(256), 0, s, (const float4*)in, (float4*)out, n4); break;
                    case 4: hipLaunchKernelGGL(k_pass, dim3(64), dim3(256), 0, s, (const float4*)in, (float4*)out, n4 / 16); break;
                    case 5: hipLaunchKernelGGL(k_pass_nt<1>, dim3(1024), dim3(256), 0, s, (const f4*)in, (f4*)out, n4); break;
                    case 6: hipLaunchKernelGGL(k_pass_nt<2>, dim3(1024), dim3(256), 0, s, (const f4*)in, (f4*)out, n4); break;
                    default: hipLaunchKernelGGL(k_pass_nt<3>, dim3(1024), dim3(256), 0, s, (const f4*)in, (f4*)out, n4); break;
                }
            }
        }
        if (two) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s0)); CK(hipGraphLaunch(ge, s0)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        static const char* names[8] = {"empty <<<1,64>>>", "empty <<<256,256>>>", "one line in, one out per WG (256 WGs)", "5 MB in -> 5 MB out, 1024 WGs", "320 KB in -> out, 64 WGs", "5 MB pass, nontemporal loads", "5 MB pass, nontemporal stores", "5 MB pass, nontemporal loads + stores"};
        printf("%s chain%s of %d nodes: %-42s %6.2f us per node%s\n", two ? "two" : "one", two ? "s" : " ", N, names[mode], best * 1e3 / N, two ? " (pair)" : "");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // two LINEAR graphs, one per stream, launched side by side (instead of one graph with two branches)
    for (int mode = 0; mode < 5; ++mode) {
        hipGraph_t g[2]; hipGraphExec_t ge[2];
        hipStream_t ss[2] = {s0, s1};
        for (int st = 0; st < 2; ++st) {
            CK(hipStreamBeginCapture(ss[st], hipStreamCaptureModeRelaxed));
            for (int i = 0; i < N; ++i) {
                float* in = st ? ((i & 1) ? d : c) : ((i & 1) ? b : a);
                float* out = st ? ((i & 1) ? c : d) : ((i & 1) ? a : b);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, ss[st]); break;
                    case 1: hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, ss[st]); break;
                    case 2: hipLaunchKernelGGL(k_line, dim3(256), dim3(256), 0, ss[st], in, out); break;
                    case 3: hipLaunchKernelGGL(k_pass, dim3(1024), dim3(256), 0, ss[st], (const float4*)in, (float4*)out, n4); break;
                    default: hipLaunchKernelGGL(k_pass, dim3(64), dim3(256), 0, ss[st], (const float4*)in, (float4*)out, n4 / 16); break;
                }
            }
            CK(hipStreamEndCapture(ss[st], &g[st]));
            CK(hipGraphInstantiate(&ge[st], g[st], nullptr, nullptr, 0));
        }
        CK(hipGraphLaunch(ge[0], s0)); CK(hipGraphLaunch(ge[1], s1)); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s0)); CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
            CK(hipGraphLaunch(ge[0], s0)); CK(hipGraphLaunch(ge[1], s1));
            CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        static const char* names[5] = {"empty <<<1,64>>>", "empty <<<256,256>>>", "one line in, one out per WG (256 WGs)", "5 MB in -> 5 MB out, 1024 WGs", "320 KB in -> out, 64 WGs"};
        printf("two linear graphs on two streams, %d nodes each: %-38s %6.2f us per node (pair)\n", N, names[mode], best * 1e3 / N);
    }
    // what a graph BOUNDARY costs: the 400-node chain of 5 MB passes as 1, 4, 16 graphs launched back to back on one stream
    for (int parts : {1, 4, 16}) {
        const int per = N / parts;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < per; ++i) hipLaunchKernelGGL(k_pass, dim3(1024), dim3(256), 0, s0, (const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b), n4);
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int k = 0; k < parts; ++k) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, s0));
            for (int k = 0; k < parts; ++k) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("one chain of %d 5 MB passes as %2d graph launches: %7.1f us total, %5.2f us per node\n", per * parts, parts, best * 1e3, best * 1e3 / (per * parts));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // no graph at all: eager launches on two streams
    for (int mode = 0; mode < 5; mode += 3) {
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0)); CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
            for (int i = 0; i < N; ++i)
                for (int st = 0; st < 2; ++st) {
                    hipStream_t s = st ? s1 : s0;
                    float* in = st ? ((i & 1) ? d : c) : ((i & 1) ? b : a);
                    float* out = st ? ((i & 1) ? c : d) : ((i & 1) ? a : b);
                    if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
                    else hipLaunchKernelGGL(k_pass, dim3(1024), dim3(256), 0, s, (const float4*)in, (float4*)out, n4);
                }
            CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("eager launches on two streams, %d nodes each: %-38s %6.2f us per node (pair, host-bound?)\n", N, mode == 0 ? "empty" : "5 MB pass", best * 1e3 / N);
    }
    return 0;
}
