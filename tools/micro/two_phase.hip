// The shape of one evaluation step, without the library: phase 1 = two independent chains of NA kernels (two nets), join, phase 2 = two
// independent chains of NB kernels (two decoder lanes), join; repeated STEPS times.  (a) ONE captured graph with branches (fork / join
// edges inside), one launch per step; (b) LINEAR graphs per (stream, phase) ordered by events between the launches; (c) eager.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_pass(const float4* __restrict__ in, float4* __restrict__ out, int n, int reps) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 v = in[i];
        for (int r = 0; r < reps; ++r) v.x = v.x * 1.0001f + 0.5f;
        out[i] = v;
    }
}
int main() {
    const int NA = 160, NB = 200, STEPS = 10;
    float* buf[4];
    const size_t bytes = 5u << 20; const int n4 = (int)(bytes / 16);
    for (auto& p : buf) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); }
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ev[4]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    for (int variant = 0; variant < 3; ++variant) {          // kernel size: tiny pass / 5 MB pass / 5 MB pass with 200 FMAs per element (~10 us)
        const int grid = variant == 0 ? 64 : 1024, cnt = variant == 0 ? n4 / 16 : n4, reps = variant == 2 ? 400 : 0;
        auto chain = [&](hipStream_t s, int which, int len) {
            for (int i = 0; i < len; ++i)
                hipLaunchKernelGGL(k_pass, dim3(grid), dim3(256), 0, s, (const float4*)buf[2 * which + (i & 1)], (float4*)buf[2 * which + 1 - (i & 1)], cnt, reps);
        };
        // (a) one graph with branches
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
        CK(hipEventRecord(ev[0], s0)); CK(hipStreamWaitEvent(s1, ev[0], 0));
        chain(s0, 0, NA); chain(s1, 1, NA);
        CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(s0, ev[1], 0));
        CK(hipEventRecord(ev[2], s0)); CK(hipStreamWaitEvent(s1, ev[2], 0));
        chain(s0, 0, NB); chain(s1, 1, NB);
        CK(hipEventRecord(ev[3], s1)); CK(hipStreamWaitEvent(s0, ev[3], 0));
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        // (b) four linear graphs
        hipGraph_t lg[4]; hipGraphExec_t lge[4];
        for (int k = 0; k < 4; ++k) {
            hipStream_t s = (k & 1) ? s1 : s0;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            chain(s, k & 1, k < 2 ? NA : NB);
            CK(hipStreamEndCapture(s, &lg[k]));
            CK(hipGraphInstantiate(&lge[k], lg[k], nullptr, nullptr, 0));
        }
        auto step_linear = [&]() -> int {
            CK(hipEventRecord(ev[0], s0)); CK(hipStreamWaitEvent(s1, ev[0], 0));
            CK(hipGraphLaunch(lge[0], s0)); CK(hipGraphLaunch(lge[1], s1));
            CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(s0, ev[1], 0));
            CK(hipEventRecord(ev[2], s0)); CK(hipStreamWaitEvent(s1, ev[2], 0));
            CK(hipGraphLaunch(lge[2], s0)); CK(hipGraphLaunch(lge[3], s1));
            CK(hipEventRecord(ev[3], s1)); CK(hipStreamWaitEvent(s0, ev[3], 0));
            return 0;
        };
        auto step_eager = [&]() -> int {
            CK(hipEventRecord(ev[0], s0)); CK(hipStreamWaitEvent(s1, ev[0], 0));
            for (int i = 0; i < NA; ++i) { chain(s0, 0, 1); chain(s1, 1, 1); }
            CK(hipEventRecord(ev[1], s1)); CK(hipStreamWaitEvent(s0, ev[1], 0));
            CK(hipEventRecord(ev[2], s0)); CK(hipStreamWaitEvent(s1, ev[2], 0));
            for (int i = 0; i < NB; ++i) { chain(s0, 0, 1); chain(s1, 1, 1); }
            CK(hipEventRecord(ev[3], s1)); CK(hipStreamWaitEvent(s0, ev[3], 0));
            return 0;
        };
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s0));
                for (int st = 0; st < STEPS; ++st) {
                    if (mode == 0) CK(hipGraphLaunch(ge, s0));
                    else if (mode == 1) { if (step_linear()) return 1; }
                    else { if (step_eager()) return 1; }
                }
                CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            static const char* kn[3] = {"320 KB pass, 64 WGs", "5 MB pass", "5 MB pass + 400 FMAs"};
            static const char* mn[3] = {"one graph with branches", "linear graphs + events ", "eager, two streams     "};
            printf("%-22s %s: %8.1f us per step (%d launches), %5.2f us per launch\n", kn[variant], mn[mode], best * 1e3 / STEPS, 2 * (NA + NB), best * 1e3 / STEPS / (2 * (NA + NB)));
        }
    }
    return 0;
}
