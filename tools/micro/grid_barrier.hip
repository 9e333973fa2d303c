// What does a dependency INSIDE a kernel cost on this machine, compared with the 1.6 us of a dependent launch (launch_floor.hip)?
// A persistent kernel of G workgroups runs P phases; in every phase a workgroup writes a tile, all workgroups meet at a grid barrier
// (one device-scope atomic counter, bounded spin), then every workgroup reads and checks the tile ANOTHER workgroup wrote (another
// XCD's, as workgroups are dealt round-robin to the 8 XCDs whose L2s are not coherent with each other).  Three ways to make the tile
// visible across XCDs:
//   mode 0  plain stores / loads + __threadfence() on both sides of the barrier (agent-scope release / acquire = buffer_wbl2 sc1 /
//           buffer_inv sc1: the WHOLE L2 of the XCD is written back / invalidated)
//   mode 1  agent-scope ("sc1") stores and loads - write-through and L2-bypassing for exactly these accesses - no fences
//   mode 2  barrier only, no data (the floor of the barrier itself)
// Every spin is bounded (a stuck barrier sets an error flag and the wave runs to the end), G never exceeds what is co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned __attribute__((ext_vector_type(4))) u4;

// Agent-scope accesses as the compiler emits them for relaxed 64-bit atomics: global_store_dwordx2 / global_load_dwordx2 with sc1, its own
// s_waitcnt placement (hand-written 16-byte asm forms with several outputs were mis-allocated by the register allocator: dropped).
typedef unsigned long long u64;
__device__ __forceinline__ void store_sc1(u4* p, u4 v) {
    __hip_atomic_store((u64*)p, (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((u64*)p + 1, (u64)v.z | ((u64)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u4 load1_sc1(const u4* p) {
    const u64 lo = __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 hi = __hip_atomic_load((const u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}

// MODE: 0 plain + fences, 1 sc1 accesses only, 2 no data, 3 sc1 accesses + buffer_wbl2 sc1 (release WITHOUT the acquire-side invalidate).
// HIER: one counter per group of G / 8 workgroups (blockIdx & 7: the XCD under round-robin dealing), the last arrival of a group adds to
// the global counter everyone polls - 8 + G / 8 serialised atomics per address instead of G.
template <int MODE, int HIER>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned phase, unsigned G, unsigned* err) {
    if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's sc1 stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 0) __threadfence();                                     // release: write the XCD's dirty L2 lines back
        if (MODE == 3) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        unsigned target;
        if (HIER) {
            const unsigned grp = blockIdx.x & 7, per = G >> 3;               // (G is a multiple of 8)
            const unsigned prev = __hip_atomic_fetch_add(counter + 64 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (phase + 1) * per - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            target = (phase + 1) * 8;
        } else {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            target = (phase + 1) * G;
        }
        int spins = 0;
        // bounded, and once ANY barrier got stuck nobody waits again: every wave reaches the end of the kernel within milliseconds
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 && __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (spins > (1 << 18)) { atomicAdd(err + 1, 1u); break; }
        }
        if (MODE == 0) __threadfence();                                     // acquire: invalidate what other XCDs may have changed
    }
    __syncthreads();
}

// V 16-byte vectors per thread and phase (tile = 256 * V * 16 B per workgroup)
template <int MODE, int V, int HIER>
__global__ __launch_bounds__(256) void k_phases(u4* buf, unsigned* counter, unsigned* err, int P) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const int src = (wg + 17) % G;                 // 17 is odd: another XCD (workgroups are dealt to the XCDs round-robin)
    unsigned bad = 0;
    for (int p = 0; p < P; ++p) {
        u4* mine = buf + ((size_t)(p & 1) * G + wg) * 256 * V;
        const u4* theirs = buf + ((size_t)(p & 1) * G + src) * 256 * V;
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const u4 v = {(unsigned)p, (unsigned)wg, (unsigned)(tid + 256 * i), (unsigned)(p * 31 + wg * 7 + tid + i)};
                if (MODE == 1 || MODE == 3) store_sc1(mine + i * 256 + tid, v); else mine[i * 256 + tid] = v;
            }
        }
        grid_barrier<MODE, HIER>(counter, (unsigned)p, (unsigned)G, err);
        if (MODE != 2) {
            u4 r[V];
            if (MODE == 1 || MODE == 3) {
#pragma unroll
                    for (int i = 0; i < V; ++i) r[i] = load1_sc1(theirs + i * 256 + tid);
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) r[i] = theirs[i * 256 + tid];
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const unsigned wrong = (r[i].x != (unsigned)p) | (r[i].y != (unsigned)src) | (r[i].z != (unsigned)(tid + 256 * i)) |
                                       (r[i].w != (unsigned)(p * 31 + src * 7 + tid + i));
                if (wrong && atomicAdd(err + 2, 1u) == 0) {          // the first wrong value of the run, for the log
                    err[3] = p; err[4] = wg; err[5] = tid; err[6] = i; err[7] = r[i].x; err[8] = r[i].y; err[9] = r[i].z; err[10] = r[i].w;
                }
                bad += wrong;
            }
        }
        // (the tile of phase p + 2 overwrites this one only after barrier p + 1, which every reader of phase p has passed)
    }
    if (bad) atomicAdd(err, 1u);          // workgroup-threads that saw at least one wrong value
}

template <int MODE, int V, int HIER>
static int run(const char* what, int G, int P, u4* buf, unsigned* counter, unsigned* err) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; unsigned h_err = 0, h_stuck = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4 * 64 * 9)); CK(hipMemset(err, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_phases<MODE, V, HIER>), dim3(G), dim3(256), 0, 0, buf, counter, err, P);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned ew = 0, cw = 0; CK(hipMemcpy(&ew, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cw, counter, 4, hipMemcpyDeviceToHost)); h_err += ew;      // (CK's own variable is called e)
        CK(hipMemcpy(&ew, err + 1, 4, hipMemcpyDeviceToHost)); h_stuck += ew;
        if (getenv("GB_DEBUG")) {
            unsigned d[11]; CK(hipMemcpy(d, err, 44, hipMemcpyDeviceToHost));
            if (d[2]) printf("   first wrong value: phase %u workgroup %u thread %u vector %u: read (%u, %u, %u, %u)\n", d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]);
        }
        if (getenv("GB_DEBUG")) printf("   rep %d: err word 0x%08x, counter %u (expected %u), %.3f ms\n", rep, ew, cw, (unsigned)P * G, ms);
    }
    printf("%-66s %s G=%4d tile %5d B: %7.3f us per phase   threads with a wrong value %u, stuck barriers %u\n", what, HIER ? "two-level barrier" : "one counter      ",
           G, 256 * V * 16, 1e3f * best / P, h_err, h_stuck);
    fflush(stdout);
    return 0;
}

int main() {
    const int P = 2000, GMAX = 1024;
    u4* buf; unsigned *counter, *err;
    CK(hipMalloc(&buf, (size_t)2 * GMAX * 256 * 4 * 16)); CK(hipMalloc(&counter, 4 * 64 * 9)); CK(hipMalloc(&err, 64));
    CK(hipMemset(buf, 0xff, (size_t)2 * GMAX * 256 * 4 * 16));
    for (int G : {256, 512, 1024}) {
        if (run<2, 1, 0>("barrier only", G, P, buf, counter, err)) return 1;
        if (run<2, 1, 1>("barrier only", G, P, buf, counter, err)) return 1;
        if (run<1, 1, 1>("sc1 stores / loads, no fences", G, P, buf, counter, err)) return 1;
        if (run<1, 4, 1>("sc1 stores / loads, no fences", G, P, buf, counter, err)) return 1;
        if (run<3, 1, 1>("sc1 stores / loads + buffer_wbl2 sc1 before the barrier", G, P, buf, counter, err)) return 1;
        if (run<3, 4, 1>("sc1 stores / loads + buffer_wbl2 sc1 before the barrier", G, P, buf, counter, err)) return 1;
        if (run<0, 1, 1>("plain accesses + agent fences (L2 write-back / invalidate)", G, P, buf, counter, err)) return 1;
        if (run<0, 4, 1>("plain accesses + agent fences (L2 write-back / invalidate)", G, P, buf, counter, err)) return 1;
    }
    return 0;
}
