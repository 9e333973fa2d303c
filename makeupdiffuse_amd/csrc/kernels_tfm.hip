// Row-local tail of a SpatialTransformer block as ONE kernel (gfx950).
//
// After self-attention every op of cldm's BasicTransformerBlock / SpatialTransformer (the two net calls it serves
// are /root/reference/diffmk/makeup_diffuse.py:164-168, sizes from /root/reference/diffmodels/base_diffusion_makeup.yaml:52-84) is
// per token:
//     h1  = attn1.to_out(a1) + h0                       a1 = softmax(q k^T) v of the self-attention (its own kernel), h0 = proj_in(norm(x))
//     q   = attn2.to_q(LN2(h1));  a2 = softmax(q K^T * dh^-0.5) V over the 77 cached context rows;  h2 = attn2.to_out(a2) + h1
//     gg  = GEGLU(ff.net.0.proj(LN3(h2)));  out = proj_out(ff.net.2(gg) + h2) + x_in
// The engine runs this as 7 dependent launches with 6 activation round trips.  Here a workgroup of 8 waves owns
// 64 consecutive tokens of one sample and keeps them in LDS across all five GEMMs:
//   * every product is computed TRANSPOSED, out^T[n][m] = sum_k W[n][k] X[m][k] with v_mfma_f32_16x16x32_bf16: W fragments are the
//     A operand, the 64-token activation tile the B operand.  The waves split the OUTPUT CHANNELS only, so a weight fragment is
//     needed by exactly one wave: it goes from L2 straight into that wave's registers, 1 KiB contiguous per wave-instruction
//     (weights are re-packed once at load time in exactly the order a wave consumes them) by raw buffer loads (descriptor and
//     stream offset in SGPRs: no per-lane 64-bit address arithmetic), TFM_PD k-steps ahead - ACROSS stages: the last k-steps of a
//     stage load the first k-steps of the next one, so the stream stays in flight over the epilogue and the barrier between them;
//     only activations live in LDS;
//   * activations sit in LDS in MFMA-operand order, tile[16-row fragment][32-wide k-step][lane][8]: a B fragment is ONE lane-linear
//     ds_read_b128 (conflict-free), and an epilogue lane (4 consecutive channels of one token) stores ONE 8-byte word; the a1 tile
//     arrives in that order by LDS-DMA (global_load_lds_dwordx4 with a per-lane source address), and so do the block's bias /
//     LayerNorm-correction vectors (27 KB), which every epilogue then reads from LDS;
//   * LayerNorm 2 / 3 are folded into the consumer's weights (W' = W diag(gamma), s = rowsum(W'), b' = b + W beta: the same folded
//     tensors the engine's LayerNorm-on-the-fly GEMM uses) and the epilogue applies rstd (acc - mean s) + b'; the row statistics are
//     taken from the bf16-rounded values the producing epilogue holds (per-wave partials in LDS, summed in fixed order:
//     bit-repeatable);
//   * the 8d-wide GEGLU intermediate is produced in 256-column chunks into a double-buffered LDS tile and consumed chunk by chunk
//     by the [ff.net.2 . proj_out | proj_out] accumulator (the merged K = 5d weight of DESIGN.md §4.2): it never exists in memory;
//   * cross-attention: wave = head.  K and V of the (sample, head) come pre-packed in MFMA-operand order (built once per
//     mkd_prepare beside the K/V cache: the context is constant over the steps), scores are computed transposed so a lane owns
//     one query, P feeds P.V from the accumulator registers (k order permuted identically in the packed V);
//   * the workgroups of an XCD split an L2 warm-up of the weight stream among themselves at kernel start (tfm_prologue).
// Per workgroup: 3.3 MB of weights (d = 320) streamed once from L2, 13.4 K MFMAs; HBM sees the a1 / h0 / x_in tiles in and the
// block output out.  Measured (profiles/exp_r4_tfm_*): 67 us per workgroup generation alone on the chip against
// 94 us for the 7 launches at 8192 rows, 146 against 302 us at 32768; in the loop +4 % (batch 8) to +7 % (interpolation batch).
#include "mkd_common.h"
#include <type_traits>
#include <vector>

namespace {

constexpr int TFM_NW = 8;            // waves per workgroup (= attention heads)
constexpr int TFM_TM = 64;           // tokens per workgroup
constexpr int TFM_MF = TFM_TM / 16;  // 16-token B fragments
constexpr int TFM_CH = 256;          // GEGLU output columns per chunk
constexpr int TFM_GKS = TFM_CH / 32; // k-steps of one chunk in the merged FF GEMM
#ifndef MKD_TFM_PD
#define MKD_TFM_PD 2
#endif
constexpr int TFM_PD = MKD_TFM_PD;   // weight k-steps in flight per wave
constexpr int TFM_KF = 5;            // 16-key fragments of the context (<= 80 keys)
constexpr int TFM_PVS = 3;           // 32-key steps of P.V (96 >= 80)

template <int D>
struct TfmCfg {
    static constexpr int HEADS = TFM_NW, DH = D / HEADS;
    static constexpr int KSD = D / 32;                    // k-steps over D
    static constexpr int NFR = D / 16;                    // W fragments (16 output channels) of an N = D product
    static constexpr int NFB = NFR / TFM_NW;              // fragments per wave: the first NWA waves take NFA = NFB + 1
    static constexpr int NWA = NFR - NFB * TFM_NW;
    static constexpr int NFA = NWA ? NFB + 1 : NFB;
    static constexpr int NCH = 4 * D / TFM_CH;            // GEGLU chunks
    static constexpr int KSM = 5 * D / 32;                // k-steps of the merged FF GEMM
    static constexpr int KSQ = (DH + 31) / 32;            // k-steps of q K^T (head dim zero-padded in the packs)
    static constexpr int MD = (DH + 15) / 16;             // O^T fragments
    static constexpr int KV_UNITS = TFM_KF * KSQ + MD * TFM_PVS;      // 1 KiB units per (sample, head)
    // packed weight offsets, in 1 KiB units (64 lanes x 8 bf16)
    static constexpr int OFF_O1 = 0, OFF_Q = NFR * KSD, OFF_O2 = 2 * NFR * KSD, OFF_G = 3 * NFR * KSD;
    static constexpr int OFF_M = OFF_G + NCH * 4 * TFM_NW * KSD, UNITS = OFF_M + NFR * KSM;
    // packed vectors (floats)
    static constexpr int V_BO1 = 0, V_SQ = D, V_BQ = 2 * D, V_BO2 = 3 * D, V_SV = 4 * D, V_BV = 8 * D, V_SG = 12 * D, V_BG = 16 * D,
                         V_BM = 20 * D, V_TOTAL = 21 * D;
    static constexpr int BUF = TFM_TM * D * 2;            // one activation tile in LDS
    static constexpr int GBUF = TFM_MF * TFM_GKS * 1024;
    static constexpr int PF_OFF = 2 * BUF + GBUF + 2 * TFM_NW * TFM_TM * 8;      // 256 B per wave: landing pad of the L2 warm-up loads
    static constexpr int VEC_OFF = PF_OFF + TFM_NW * 256;                         // the packed vectors, V_TOTAL floats
    static constexpr int LDS = VEC_OFF + V_TOTAL * 4;
    static_assert(D % 64 == 0 && DH % 8 == 0 && (4 * D) % TFM_CH == 0 && GBUF <= BUF, "tile geometry");
};

struct TfmTailArgs {
    const bf16x8* wpk;          // packed weights (TfmCfg::UNITS KiB)
    const float* vec;           // packed bias / LayerNorm-correction vectors
    const bf16_t* a1; int lda;  // self-attention output [M, d]
    const bf16_t* h0; int ldh;  // residual of attn1.to_out
    const bf16_t* xin; int ldx; // the SpatialTransformer's input (residual of proj_out)
    const bf16x8* kvp;          // packed context K / V: [sample][head][KV_UNITS] KiB
    bf16_t* out; int ldo;
    int T;                      // tokens per sample (multiple of TFM_TM)
    int Tk;                     // context keys (<= 16 * TFM_KF)
    float scale_log2e;          // dh^-0.5 * log2(e)
    long long* trace;           // MKD_TFM_TRACE builds only: [workgroup][wave][32] wall_clock64 stamps (tools/exp_r4_tfm_trace.py)
};

#ifdef MKD_TFM_TRACE
#define TFM_STAMP(i) do { if (a.trace && lane == 0) a.trace[((size_t)blockIdx.x * TFM_NW + w) * 32 + (i)] = wall_clock64(); } while (0)
#else
#define TFM_STAMP(i) do { } while (0)
#endif

typedef short s16x8v __attribute__((ext_vector_type(8)));

// byte offset of the 16-byte chunk (token row, 8-channel group c) inside a tile with KS k-steps per row fragment
__device__ __forceinline__ int lds_chunk(int mf, int c, int r, int KS) { return (((mf * KS + (c >> 2)) << 6) + ((c & 3) << 4) + r) << 4; }

__device__ __forceinline__ void warm64(const void* p, char* pad) {      // pull one 64-byte segment per lane towards L2 (lands in a pad nobody reads)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)pad, 4, 0, 0);
}

// The weight stream of a wave: TFM_PD k-steps of NF fragments in registers.  A stage consumes slot ks % TFM_PD at k-step ks and
// refills it with k-step ks + TFM_PD - or, in its last TFM_PD k-steps, with the first k-steps of the NEXT stage's stream, so that
// the loads stay in flight across the epilogue and the workgroup barrier between two stages.
template <int NF> struct Ring { bf16x8 r[TFM_PD][NF]; };
// one 1 KiB unit of a wave's stream by a raw buffer load: resource descriptor and stream offset in SGPRs (scalar adds), the lane's
// 16 bytes as the only VGPR operand - no 64-bit per-lane address arithmetic in the k-loops
typedef int i32x4v __attribute__((ext_vector_type(4)));
struct WStream { __amdgpu_buffer_rsrc_t rs; int off; };          // off: byte offset of the stream inside the buffer (wave-uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 ldw(const WStream& w, unsigned lo, int unit) {
    const i32x4v v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, (int)lo, w.off + unit * 1024, 0);
    return __builtin_bit_cast(bf16x8, v);
}
template <int NF>
__device__ __forceinline__ void ring_fill(Ring<NF>& R, const WStream& wp, unsigned lo) {
#pragma unroll
    for (int p = 0; p < TFM_PD; ++p)
#pragma unroll
        for (int f = 0; f < NF; ++f) R.r[p][f] = ldw(wp, lo, p * NF + f);
}

// acc[f][mf] += W fragment (f, ks) x activation fragment (mf, ks) over KS k-steps.  wp: the wave's packed stream (wave-uniform
// pointer), unit order [ks][f]; lo: the lane's byte offset inside a unit; cur holds its first TFM_PD k-steps; nxt / wpn: ring and stream of the stage that follows
// (NEXT = false: none).  xb: LDS tile + lane * 16 (+ first k-step * 1024); xs: bytes between row fragments.
template <int NF, int KS, bool NEXT, int NFX>
__device__ __forceinline__ void stage_mm(Ring<NF>& cur, const WStream& wp, Ring<NFX>& nxt, const WStream& wpn, unsigned lo,
                                         const char* xb, int xs, f32x4 (&acc)[NF][TFM_MF]) {
    static_assert(KS >= TFM_PD, "a stage is at least TFM_PD k-steps long");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 xf[TFM_MF];
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) xf[mf] = *(const bf16x8*)(xb + mf * xs + ks * 1024);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf) {
#ifdef TFM_EXP_NOMFMA           // (experiment build: operands stay live, no matrix instruction - wrong numbers)
                asm volatile("" :: "v"(cur.r[ks % TFM_PD][f]), "v"(xf[mf]));
#else
                acc[f][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.r[ks % TFM_PD][f], xf[mf], acc[f][mf], 0, 0, 0);
#endif
            }
#ifndef TFM_EXP_NOLOAD          // (experiment build: every k-loop re-uses the first TFM_PD weight steps - wrong numbers, no weight stream)
        if (ks + TFM_PD < KS) {
#pragma unroll
            for (int f = 0; f < NF; ++f) cur.r[ks % TFM_PD][f] = ldw(wp, lo, (ks + TFM_PD) * NF + f);
        } else if (NEXT) {
            const int p = ks + TFM_PD - KS;
#pragma unroll
            for (int f = 0; f < NFX; ++f) nxt.r[p][f] = ldw(wpn, lo, p * NFX + f);
        }
#else
        if (NEXT && ks + TFM_PD >= KS) {
            const int p = ks + TFM_PD - KS;
#pragma unroll
            for (int f = 0; f < NFX; ++f) nxt.r[p][f] = cur.r[p][f < NF ? f : 0];
        }
#endif
    }
}

template <int NF>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NF][TFM_MF]) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) acc[f][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// row statistics of the tile a stage wrote: this wave's partial (sum, sum of squares) per token -> st[wave][token]
__device__ __forceinline__ void put_stats(float (&s)[TFM_MF], float (&q)[TFM_MF], float2* st, int w, int r, int g) {
#pragma unroll
    for (int mf = 0; mf < TFM_MF; ++mf) {
        float a = s[mf], b = q[mf];
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        if (g == 0) st[w * TFM_TM + 16 * mf + r] = make_float2(a, b);
    }
}
template <int D>
__device__ __forceinline__ void get_stats(const float2* st, int r, float (&mean)[TFM_MF], float (&rstd)[TFM_MF]) {
#pragma unroll
    for (int mf = 0; mf < TFM_MF; ++mf) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < TFM_NW; ++w) { const float2 v = st[w * TFM_TM + 16 * mf + r]; a += v.x; b += v.y; }      // fixed order
        const float m = a * (1.0f / D);
        const float var = fmaxf(b * (1.0f / D) - m * m, 0.f);
        mean[mf] = m;
        rstd[mf] = __builtin_amdgcn_rsqf(var + 1e-5f);
    }
}
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }

// Start of a workgroup: the a1 tile and the vectors into LDS by LDS-DMA, the L2 warm-up; completes on the caller's vmcnt(0) + barrier
template <int D>
__device__ __forceinline__ void tfm_prologue(const TfmTailArgs& a, char* smem, const int w, const int lane) {
    using C = TfmCfg<D>;
    const int row0 = blockIdx.x * TFM_TM;
    // a1 tile -> bufA in operand order: 1 KiB block (mf, ks) = 64 lanes x 16 B, lane (g, r) supplies the address of
    // a1[16 mf + r][32 ks + 8 g ..]
    constexpr int PER = TFM_MF * C::KSD / TFM_NW;
    static_assert(TFM_MF * C::KSD % TFM_NW == 0, "a1 tile blocks per wave");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int blk = w + i * TFM_NW, mf = blk / C::KSD, ks = blk - mf * C::KSD;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.a1 + (size_t)(row0 + 16 * mf + (lane & 15)) * a.lda + 32 * ks + 8 * (lane >> 4)),
                                         (__attribute__((address_space(3))) void*)(smem + blk * 1024), 16, 0, 0);
    }
    // bias / LayerNorm-correction vectors -> LDS (every epilogue reads them from there)
    {
        constexpr int NV16 = C::V_TOTAL / 4;                    // 16-byte pieces
        const float4* src = (const float4*)a.vec;
        for (int i = w * 64; i < NV16; i += TFM_NW * 64)
            if (i + lane < NV16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i + lane),
                                                 (__attribute__((address_space(3))) void*)(smem + C::VEC_OFF + i * 16), 16, 0, 0);
    }
    // L2 warm-up.  Every workgroup streams ALL the block's weights (3.3 MB, they fit the XCD's 4 MiB L2), and the workgroups of an
    // XCD run in step: left to demand loads, every line is an HBM miss for all of them at once.  So the workgroups of an XCD split
    // the stream: workgroup i of the XCD (workgroups are dealt round-robin, blockIdx / 8 numbers them - a speed assumption only)
    // touches granule i, i + n, ... once at kernel start (one dword per 64 B, landing in an LDS pad nobody reads).  The same for
    // what this workgroup reads late and only once: its h0 / x_in residual tiles.
    {
        char* pad = smem + C::PF_OFF + w * 256;
        const int nsl = min(16, max(1, (int)gridDim.x >> 3)), sl = ((int)blockIdx.x >> 3) % nsl;
        const char* base = (const char*)a.wpk + lane * 64;
        for (int gi = sl + nsl * w; gi < C::UNITS / 4; gi += nsl * TFM_NW) warm64(base + (size_t)gi * 4096, pad);
        constexpr int SEG = D * 2 / 64;                          // 64-byte segments per tile row
        for (int i = lane; i < (TFM_TM / TFM_NW) * SEG; i += 64) {
            const int rr = (TFM_TM / TFM_NW) * w + i / SEG, sg = i - (i / SEG) * SEG;
            warm64((const char*)(a.xin + (size_t)(row0 + rr) * a.ldx) + sg * 64, pad);
            warm64((const char*)(a.h0 + (size_t)(row0 + rr) * a.ldh) + sg * 64, pad);
        }
    }
}

// the program of one wave; NFN = its share of the 16-channel fragments of an N = D product, fr0 = the first of them.
template <int D, int NFN>
__device__ __forceinline__ void tfm_wave(const TfmTailArgs& a, char* smem, const int w, const int lane, const int fr0) {
    using C = TfmCfg<D>;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * TFM_TM;
    char* const bufA = smem;
    char* const bufB = smem + C::BUF;
    char* const bufG1 = smem + 2 * C::BUF;
    float2* const st1 = (float2*)(smem + 2 * C::BUF + C::GBUF);
    float2* const st2 = st1 + TFM_NW * TFM_TM;
    const float* const vec = (const float*)(smem + C::VEC_OFF);          // the bias / LayerNorm-correction vectors, staged at kernel start
    const int xs = C::KSD * 1024;
    const unsigned lo = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wpk, (unsigned)C::UNITS * 1024u);
    const WStream w_o1{wrs, (C::OFF_O1 + C::KSD * fr0) * 1024}, w_q{wrs, (C::OFF_Q + C::KSD * fr0) * 1024};
    const WStream w_o2{wrs, (C::OFF_O2 + C::KSD * fr0) * 1024}, w_m{wrs, (C::OFF_M + C::KSM * fr0) * 1024};
    const WStream w_g{wrs, (C::OFF_G + w * 4 * C::KSD) * 1024};                   // + chunk * G_CHUNK
    constexpr int G_CHUNK = TFM_NW * 4 * C::KSD * 1024;
    auto at = [](const WStream& b, int bytes) { return WStream{b.rs, b.off + bytes}; };

    // address pieces of this lane's epilogue word: fragment fr -> channels 16 fr + 4 g .. + 3 of token 16 mf + r
    auto out_off = [&](int mf, int fr) { return lds_chunk(mf, 2 * fr + (g >> 1), r, C::KSD) + (g & 1) * 8; };

    TFM_STAMP(0);
    Ring<NFN> ring0, ring1;
    ring_fill<NFN>(ring0, w_o1, lo);                       // the first k-steps of attn1.to_out's weights: in flight under the prologue
    tfm_prologue<D>(a, smem, w, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the a1 tile and the vectors have landed in LDS (LDS-DMA completes on vmcnt)
    __syncthreads();
    TFM_STAMP(1);
    // ---- S1: h1 = attn1.to_out(a1) + h0 -> bufB, row statistics -> st1 -------------------------------------------------
    {
        U16x4 res[NFN][TFM_MF];
#pragma unroll
        for (int f = 0; f < NFN; ++f)
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf)
                res[f][mf] = *(const U16x4*)(a.h0 + (size_t)(row0 + 16 * mf + r) * a.ldh + 16 * (fr0 + f) + 4 * g);
        f32x4 acc[NFN][TFM_MF];
        zero_acc<NFN>(acc);
        stage_mm<NFN, C::KSD, true, NFN>(ring0, w_o1, ring1, w_q, lo, bufA + lane * 16, xs, acc);
        TFM_STAMP(2);
        float s[TFM_MF], q[TFM_MF];
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) { s[mf] = 0.f; q[mf] = 0.f; }
#pragma unroll
        for (int f = 0; f < NFN; ++f) {
            float bv[4]; ld4(vec + C::V_BO1 + 16 * (fr0 + f) + 4 * g, bv);
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf) {
                U16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o.v[i] = f32_to_bf16(acc[f][mf][i] + bv[i] + bf16_to_f32(res[f][mf].v[i]));
                    const float vr = bf16_to_f32(o.v[i]);
                    s[mf] += vr; q[mf] += vr * vr;
                }
                *(U16x4*)(bufB + out_off(mf, fr0 + f)) = o;
            }
        }
        TFM_STAMP(3);
        put_stats(s, q, st1, w, r, g);
    }
    // context K / V of this wave's head, in operand order: in flight under S2
    bf16x8 kf[TFM_KF][C::KSQ];
    const WStream kv{make_rsrc((const char*)a.kvp + (size_t)(row0 / a.T) * C::HEADS * C::KV_UNITS * 1024, C::HEADS * C::KV_UNITS * 1024u), w * C::KV_UNITS * 1024};
#pragma unroll
    for (int i = 0; i < TFM_KF; ++i)
#pragma unroll
        for (int ks = 0; ks < C::KSQ; ++ks) kf[i][ks] = ldw(kv, lo, i * C::KSQ + ks);
    __syncthreads();
    TFM_STAMP(4);

    // ---- S2: q = attn2.to_q(LN2(h1)) -> bufA (a1 is dead) --------------------------------------------------------------
    {
        float mean[TFM_MF], rstd[TFM_MF];
        get_stats<D>(st1, r, mean, rstd);
        f32x4 acc[NFN][TFM_MF];
        zero_acc<NFN>(acc);
        stage_mm<NFN, C::KSD, true, NFN>(ring1, w_q, ring0, w_o2, lo, bufB + lane * 16, xs, acc);
        TFM_STAMP(5);
#pragma unroll
        for (int f = 0; f < NFN; ++f) {
            float sv[4], bv[4];
            ld4(vec + C::V_SQ + 16 * (fr0 + f) + 4 * g, sv); ld4(vec + C::V_BQ + 16 * (fr0 + f) + 4 * g, bv);
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf) {
                U16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o.v[i] = f32_to_bf16(rstd[mf] * (acc[f][mf][i] - mean[mf] * sv[i]) + bv[i]);
                *(U16x4*)(bufA + out_off(mf, fr0 + f)) = o;
            }
        }
    }
    TFM_STAMP(6);
    __syncthreads();
    TFM_STAMP(7);

    // ---- S3: cross-attention over the cached context, wave = head; a2 overwrites q in place --------------------------
    {
        bf16x8 vf[C::MD][TFM_PVS];                      // (V: needed after the first softmax)
#pragma unroll
        for (int md = 0; md < C::MD; ++md)
#pragma unroll
            for (int s = 0; s < TFM_PVS; ++s) vf[md][s] = ldw(kv, lo, TFM_KF * C::KSQ + md * TFM_PVS + s);
        const int c0 = (C::DH / 8) * w;                 // first 8-channel group of this head
#pragma unroll 1
        for (int qf = 0; qf < TFM_MF; ++qf) {
            // Q fragment (B operand): lane holds q[16 qf + r][DH w + 32 ks + 8 g + j]; channels past the head are zero
            bf16x8 qv[C::KSQ];
#pragma unroll
            for (int ks = 0; ks < C::KSQ; ++ks) {
                const bool ok = 32 * ks + 8 * g < C::DH;
                const bf16x8 t = *(const bf16x8*)(bufA + lds_chunk(qf, ok ? c0 + 4 * ks + g : c0, r, C::KSD));
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                qv[ks] = ok ? t : z;
            }
            f32x4 stt[TFM_KF + 1];
#pragma unroll
            for (int i = 0; i < TFM_KF; ++i) {
                stt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < C::KSQ; ++ks) stt[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][ks], qv[ks], stt[i], 0, 0, 0);
            }
            stt[TFM_KF] = f32x4{0.f, 0.f, 0.f, 0.f};
            // lane holds raw scores of query r for keys 16 i + 4 g + j
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < TFM_KF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sc = (16 * i + 4 * g + j < a.Tk) ? stt[i][j] : -INFINITY;
                    stt[i][j] = sc;
                    mx = fmaxf(mx, sc);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m = mx * a.scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < TFM_KF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(stt[i][j], a.scale_log2e, -m));
                    stt[i][j] = p;
                    psum += p;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(psum);
            // O^T[d][q] = V^T[d][key'] P^T[key'][q]; key'(g, j) = 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4): the packed V uses the same order
            f32x4 oacc[C::MD];
#pragma unroll
            for (int md = 0; md < C::MD; ++md) oacc[md] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TFM_PVS; ++s) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[j] = (__bf16)stt[2 * s][j];
                    pf[4 + j] = (__bf16)stt[2 * s + 1][j];
                }
#pragma unroll
                for (int md = 0; md < C::MD; ++md) oacc[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[md][s], pf, oacc[md], 0, 0, 0);
            }
#pragma unroll
            for (int md = 0; md < C::MD; ++md) {
                const int dc = 16 * md + 4 * g;
                if (dc < C::DH) {
                    U16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o.v[i] = f32_to_bf16(oacc[md][i] * inv);
                    const int n = C::DH * w + dc;
                    *(U16x4*)(bufA + lds_chunk(qf, n >> 3, r, C::KSD) + (n & 7) * 2) = o;
                }
            }
        }
    }
    TFM_STAMP(8);
    __syncthreads();
    TFM_STAMP(9);

    // ---- S4: h2 = attn2.to_out(a2) + h1 -> bufB in place, row statistics -> st2 ----------------------------------------
    Ring<4> ringG;
    {
        f32x4 acc[NFN][TFM_MF];
        zero_acc<NFN>(acc);
        stage_mm<NFN, C::KSD, true, 4>(ring0, w_o2, ringG, w_g, lo, bufA + lane * 16, xs, acc);
        TFM_STAMP(10);
        float s[TFM_MF], q[TFM_MF];
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) { s[mf] = 0.f; q[mf] = 0.f; }
#pragma unroll
        for (int f = 0; f < NFN; ++f) {
            float bv[4]; ld4(vec + C::V_BO2 + 16 * (fr0 + f) + 4 * g, bv);
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf) {
                char* const p = bufB + out_off(mf, fr0 + f);
                const U16x4 h1 = *(const U16x4*)p;
                U16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o.v[i] = f32_to_bf16(acc[f][mf][i] + bv[i] + bf16_to_f32(h1.v[i]));
                    const float vr = bf16_to_f32(o.v[i]);
                    s[mf] += vr; q[mf] += vr * vr;
                }
                *(U16x4*)p = o;
            }
        }
        TFM_STAMP(11);
        put_stats(s, q, st2, w, r, g);
    }
    __syncthreads();
    TFM_STAMP(12);

    // ---- S5 / S6: GEGLU chunks -> double-buffered LDS tile -> merged [ff.net.2 . proj_out | proj_out] accumulator -----------
    f32x4 accF[NFN][TFM_MF];
    zero_acc<NFN>(accF);
    {
#pragma unroll 1
        for (int c = 0; c < C::NCH; ++c) {
            char* const G = (c & 1) ? bufG1 : bufA;
            const WStream wmc = at(w_m, c * TFM_GKS * NFN * 1024);
            {
                f32x4 acc[4][TFM_MF];
                zero_acc<4>(acc);
                stage_mm<4, C::KSD, true, NFN>(ringG, at(w_g, c * G_CHUNK), ring1, wmc, lo, bufB + lane * 16, xs, acc);
                if (c == 1) TFM_STAMP(13);
                float mean[TFM_MF], rstd[TFM_MF];               // (re-derived per chunk from the LDS partials: 8 registers fewer across the k-loops)
                get_stats<D>(st2, r, mean, rstd);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int col = 16 * (2 * w + p) + 4 * g;           // column inside the chunk
                    const int j0 = TFM_CH * c + col;                    // GEGLU output column
                    float sv[4], bv[4], sg[4], bg[4];
                    ld4(vec + C::V_SV + j0, sv); ld4(vec + C::V_BV + j0, bv); ld4(vec + C::V_SG + j0, sg); ld4(vec + C::V_BG + j0, bg);
#pragma unroll
                    for (int mf = 0; mf < TFM_MF; ++mf) {
                        U16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float uv = rstd[mf] * (acc[2 * p][mf][i] - mean[mf] * sv[i]) + bv[i];
                            const float ug = rstd[mf] * (acc[2 * p + 1][mf][i] - mean[mf] * sg[i]) + bg[i];
                            o.v[i] = f32_to_bf16(uv * gelu_erf_f(ug));
                        }
                        *(U16x4*)(G + lds_chunk(mf, col >> 3, r, TFM_GKS) + (col & 7) * 2) = o;
                    }
                }
            }
            if (c == 1) TFM_STAMP(14);
            __syncthreads();
            if (c == 1) TFM_STAMP(15);
            // (after the last chunk the ring of the GEGLU stream is refilled with that chunk's first k-steps again: never used)
            const int cn = c + 1 < C::NCH ? c + 1 : c;
            stage_mm<NFN, TFM_GKS, true, 4>(ring1, wmc, ringG, at(w_g, cn * G_CHUNK), lo, G + lane * 16, TFM_GKS * 1024, accF);
            if (c == 1) TFM_STAMP(16);
            if (c == C::NCH - 1) TFM_STAMP(17);
        }
        // the h2 part of the merged GEMM (K columns 4d .. 5d)
        const WStream wt = at(w_m, C::NCH * TFM_GKS * NFN * 1024);
        ring_fill<NFN>(ring1, wt, lo);
        stage_mm<NFN, C::KSD, false, NFN>(ring1, wt, ring1, wt, lo, bufB + lane * 16, xs, accF);
        TFM_STAMP(18);
    }

    // ---- out = accF + (proj_out . b2 + b_proj_out) + x_in ----------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < NFN; ++f) {
        const int n = 16 * (fr0 + f) + 4 * g;
        float bv[4]; ld4(vec + C::V_BM + n, bv);
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) {
            const size_t row = (size_t)(row0 + 16 * mf + r);
            const U16x4 xr = *(const U16x4*)(a.xin + row * a.ldx + n);
            U16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o.v[i] = f32_to_bf16(accF[f][mf][i] + bv[i] + bf16_to_f32(xr.v[i]));
            *(U16x4*)(a.out + row * a.ldo + n) = o;
        }
    }
    TFM_STAMP(19);
}

template <int D>
__global__ __launch_bounds__(64 * TFM_NW) void tfm_tail_kernel(const TfmTailArgs a) {
    using C = TfmCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keeps every per-wave offset in SGPRs
    if (C::NWA == TFM_NW || w < C::NWA) tfm_wave<D, C::NFA>(a, smem, w, lane, w * C::NFA);
    else tfm_wave<D, C::NFB>(a, smem, w, lane, C::NWA * C::NFA + (w - C::NWA) * C::NFB);
}

// ---- the head of the block: GroupNorm apply + proj_in + LayerNorm 1 . (q | k | v) ----------------------------------------------------
// Same machinery as the tail (weights packed in consumption order -> registers of one wave, activations in LDS in operand order).
// The GroupNorm statistics come from a full-chip statistics launch (launch_gn_stats: partial sums per row chunk); the workgroup
// finishes them for its sample, normalises its x tile on the way into LDS, and writes h0 (the tail's residual) and q | k | v.
template <int D>
struct TfmHeadCfg {
    using T = TfmCfg<D>;
    static constexpr int OFF_PI = 0, OFF_QKV = T::NFR * T::KSD, UNITS = 4 * T::NFR * T::KSD;      // proj_in, then the q, k, v thirds
    static constexpr int V_BPI = 0, V_S = D, V_B = 4 * D, V_G = 7 * D, V_BETA = 8 * D, V_TOTAL = 9 * D;
    static constexpr int ST_OFF = 2 * T::BUF;                      // LayerNorm 1 partials [waves][tokens] float2
    static constexpr int VEC_OFF = ST_OFF + TFM_NW * TFM_TM * 8;
    static constexpr int AB_OFF = VEC_OFF + V_TOTAL * 4;           // GroupNorm scale / shift per channel [D] float2
    static constexpr int GS_OFF = AB_OFF + D * 8;                  // GroupNorm (mean, rstd) per group [32] float2
    static constexpr int PF_OFF = GS_OFF + 32 * 8;                 // 256 B per wave: landing pad of the L2 warm-up loads
    static constexpr int LDS = PF_OFF + TFM_NW * 256;
};

struct TfmHeadArgs {
    const bf16x8* wpk; const float* vec;
    const bf16_t* x; int ldx;
    const float* part; int nchunks; float gn_eps;       // GroupNorm partials [(b * nchunks + chunk) * 32 + g][2]
    bf16_t* h0; bf16_t* qkv;
    int T;
};

// one third (q, k or v: N = D output channels) of the LayerNorm-folded projection, written straight to global
template <int D, int NFN, bool NEXT>
__device__ __forceinline__ void head_third(const int t, Ring<NFN>& cur, Ring<NFN>& nxt, const WStream& ws, const WStream& wn, unsigned lo,
                                           const char* xb, int xs, const float* vec, const float (&mean)[TFM_MF], const float (&rstd)[TFM_MF],
                                           bf16_t* qkv, int row0, int fr0, int r, int g) {
    using C = TfmCfg<D>;
    using H = TfmHeadCfg<D>;
    f32x4 acc[NFN][TFM_MF];
    zero_acc<NFN>(acc);
    stage_mm<NFN, C::KSD, NEXT, NFN>(cur, ws, nxt, wn, lo, xb, xs, acc);
#pragma unroll
    for (int f = 0; f < NFN; ++f) {
        const int n = 16 * (fr0 + f) + 4 * g;
        float sv[4], bv[4];
        ld4(vec + H::V_S + t * D + n, sv); ld4(vec + H::V_B + t * D + n, bv);
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) {
            U16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o.v[i] = f32_to_bf16(rstd[mf] * (acc[f][mf][i] - mean[mf] * sv[i]) + bv[i]);
            *(U16x4*)(qkv + (size_t)(row0 + 16 * mf + r) * (3 * D) + t * D + n) = o;
        }
    }
}

template <int D, int NFN>
__device__ __forceinline__ void tfm_head_wave(const TfmHeadArgs& a, char* smem, const int w, const int lane, const int fr0) {
    using C = TfmCfg<D>;
    using H = TfmHeadCfg<D>;
    const int tid = w * 64 + lane;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * TFM_TM;
    const int b = row0 / a.T;
    char* const bufA = smem;
    char* const bufB = smem + C::BUF;
    float2* const st1 = (float2*)(smem + H::ST_OFF);
    const float* const vec = (const float*)(smem + H::VEC_OFF);
    float2* const ab = (float2*)(smem + H::AB_OFF);
    float2* const gs = (float2*)(smem + H::GS_OFF);
    const int xs = C::KSD * 1024;
    const unsigned lo = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.wpk, (unsigned)H::UNITS * 1024u);
    const WStream w_pi{wrs, (H::OFF_PI + C::KSD * fr0) * 1024};
    auto w_third = [&](int t) { return WStream{wrs, (H::OFF_QKV + t * C::NFR * C::KSD + C::KSD * fr0) * 1024}; };
    auto out_off = [&](int mf, int fr) { return lds_chunk(mf, 2 * fr + (g >> 1), r, C::KSD) + (g & 1) * 8; };

    Ring<NFN> ring0, ring1;
    ring_fill<NFN>(ring0, w_pi, lo);
    // x tile -> registers in operand order (chunk idx = block (mf, ks) * 64 + (g, r)); vectors -> LDS by LDS-DMA
    constexpr int PER = TFM_MF * C::KSD / TFM_NW;
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    u32x4v xr[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int blk = w + i * TFM_NW, mf = blk / C::KSD, ks = blk - mf * C::KSD;
        xr[i] = *(const u32x4v*)(a.x + (size_t)(row0 + 16 * mf + r) * a.ldx + 32 * ks + 8 * g);
    }
    {
        constexpr int NV16 = H::V_TOTAL / 4;
        const float4* src = (const float4*)a.vec;
        for (int i = w * 64; i < NV16; i += TFM_NW * 64)
            if (i + lane < NV16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i + lane),
                                                 (__attribute__((address_space(3))) void*)(smem + H::VEC_OFF + i * 16), 16, 0, 0);
    }
    // GroupNorm statistics of this sample: thread (slice = tid / 32, group = tid % 32) sums the partials of row chunks slice,
    // slice + 16, ... (at most 4: launch_gn_stats cuts a sample into <= 64 chunks; all loads in flight at once, fixed order), then
    // 32 threads add the 16 slices in order
    float2* const red = (float2*)bufB;                      // [16][32], free until S1's epilogue
    {
        const int sl = tid >> 5, gi = tid & 31;
        float2 pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = sl + 16 * i;
            pp[i] = *(const float2*)(a.part + ((size_t)(b * a.nchunks + (c < a.nchunks ? c : 0)) * 32 + gi) * 2);
        }
        // L2 warm-up of the weight stream, split among the workgroups of an XCD as in tfm_prologue
        {
            char* pad = smem + H::PF_OFF + w * 256;
            const int nsl = min(16, max(1, (int)gridDim.x >> 3)), sx = ((int)blockIdx.x >> 3) % nsl;
            const char* base = (const char*)a.wpk + lane * 64;
            for (int q = sx + nsl * w; q < H::UNITS / 4; q += nsl * TFM_NW) warm64(base + (size_t)q * 4096, pad);
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (sl + 16 * i < a.nchunks) { s += pp[i].x; q += pp[i].y; }
        red[sl * 32 + gi] = make_float2(s, q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // vectors landed in LDS (LDS-DMA completes on vmcnt)
    __syncthreads();
    if (tid < 32) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float2 v = red[i * 32 + tid]; s += v.x; q += v.y; }
        const float n = (float)a.T * (float)(D / 32);
        const float mean = s / n;
        const float var = fmaxf(q / n - mean * mean, 0.f);
        gs[tid] = make_float2(mean, rsqrtf(var + a.gn_eps));
    }
    __syncthreads();
    if (tid < D) {                                          // per channel: y = x * a + b
        const float2 ms = gs[tid / (D / 32)];
        const float sc = vec[H::V_G + tid] * ms.y;
        ab[tid] = make_float2(sc, vec[H::V_BETA + tid] - ms.x * sc);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int blk = w + i * TFM_NW, ks = blk % C::KSD;
        const int c0 = 32 * ks + 8 * g;
        U16x8 t = __builtin_bit_cast(U16x8, xr[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float2 p = ab[c0 + j]; t.v[j] = f32_to_bf16(bf16_to_f32(t.v[j]) * p.x + p.y); }
        *(U16x8*)(bufA + (blk * 64 + lane) * 16) = t;
    }
    __syncthreads();

    // ---- S1: h0 = proj_in(g) -> bufB and global, LayerNorm 1 partials -> st1 -------------------------------------------------
    {
        f32x4 acc[NFN][TFM_MF];
        zero_acc<NFN>(acc);
        stage_mm<NFN, C::KSD, true, NFN>(ring0, w_pi, ring1, w_third(0), lo, bufA + lane * 16, xs, acc);
        float s[TFM_MF], q[TFM_MF];
#pragma unroll
        for (int mf = 0; mf < TFM_MF; ++mf) { s[mf] = 0.f; q[mf] = 0.f; }
#pragma unroll
        for (int f = 0; f < NFN; ++f) {
            const int n = 16 * (fr0 + f) + 4 * g;
            float bv[4]; ld4(vec + H::V_BPI + n, bv);
#pragma unroll
            for (int mf = 0; mf < TFM_MF; ++mf) {
                U16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o.v[i] = f32_to_bf16(acc[f][mf][i] + bv[i]);
                    const float vr = bf16_to_f32(o.v[i]);
                    s[mf] += vr; q[mf] += vr * vr;
                }
                *(U16x4*)(bufB + out_off(mf, fr0 + f)) = o;
                *(U16x4*)(a.h0 + (size_t)(row0 + 16 * mf + r) * D + n) = o;
            }
        }
        put_stats(s, q, st1, w, r, g);
    }
    __syncthreads();
    // ---- S2: q | k | v = LN1(h0) . W'^T, one third (N = D) at a time; straight to global ------------------------------------
    {
        float mean[TFM_MF], rstd[TFM_MF];
        get_stats<D>(st1, r, mean, rstd);
        // (no barrier separates the thirds: without the scheduling fences the compiler hoists the later thirds' weight loads to the
        // front and spills 60 registers)
        head_third<D, NFN, true>(0, ring1, ring0, w_third(0), w_third(1), lo, bufB + lane * 16, xs, vec, mean, rstd, a.qkv, row0, fr0, r, g);
        __builtin_amdgcn_sched_barrier(0);
        head_third<D, NFN, true>(1, ring0, ring1, w_third(1), w_third(2), lo, bufB + lane * 16, xs, vec, mean, rstd, a.qkv, row0, fr0, r, g);
        __builtin_amdgcn_sched_barrier(0);
        head_third<D, NFN, false>(2, ring1, ring0, w_third(2), w_third(2), lo, bufB + lane * 16, xs, vec, mean, rstd, a.qkv, row0, fr0, r, g);
    }
}

template <int D>
__global__ __launch_bounds__(64 * TFM_NW) void tfm_head_kernel(const TfmHeadArgs a) {
    using C = TfmCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (C::NWA == TFM_NW || w < C::NWA) tfm_head_wave<D, C::NFA>(a, smem, w, lane, w * C::NFA);
    else tfm_head_wave<D, C::NFB>(a, smem, w, lane, C::NWA * C::NFA + (w - C::NWA) * C::NFB);
}

// ---- packing ---------------------------------------------------------------------------------------------------------------
// one 1 KiB unit per block: out[u][lane][j] = W[rows[u][lane & 15]][32 ks[u] + 8 (lane >> 4) + j]
__global__ void tfm_pack_units_kernel(const bf16_t* __restrict__ W, int ldw, const int* __restrict__ tab, bf16_t* __restrict__ out) {
    const int u = blockIdx.x, lane = threadIdx.x;
    const int* t = tab + (size_t)u * 17;
    const int row = t[lane & 15], ks = t[16];
    *(U16x8*)(out + ((size_t)u * 64 + lane) * 8) = *(const U16x8*)(W + (size_t)row * ldw + 32 * ks + 8 * (lane >> 4));
}

// context K / V of one (sample, head) in operand order, zero padded: kv [B * Tk, 2 d] (K | V), heads of dh channels
template <int D>
__global__ void tfm_pack_kv_kernel(const bf16_t* __restrict__ kv, int ldkv, int Tk, bf16_t* __restrict__ out) {
    using C = TfmCfg<D>;
    const int bh = blockIdx.x, b = bh / C::HEADS, h = bh - b * C::HEADS;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const bf16_t* base = kv + (size_t)b * Tk * ldkv + h * C::DH;
    bf16_t* o = out + (size_t)bh * C::KV_UNITS * 512;
    for (int u = threadIdx.x >> 6; u < C::KV_UNITS; u += blockDim.x >> 6) {
        U16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = 0;
        if (u < TFM_KF * C::KSQ) {                       // K fragment (key block i, k-step ks): K[16 i + r][32 ks + 8 g + j]
            const int i = u / C::KSQ, ks = u - i * C::KSQ;
            const int key = 16 * i + r, d0 = 32 * ks + 8 * g;
            if (key < Tk && d0 < C::DH) v = *(const U16x8*)(base + (size_t)key * ldkv + d0);
        } else {                                         // V^T fragment (channel block md, key step s): V[key'(g, j)][16 md + r]
            const int uu = u - TFM_KF * C::KSQ, md = uu / TFM_PVS, s = uu - md * TFM_PVS;
            const int dch = 16 * md + r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
                if (key < Tk && dch < C::DH) v.v[j] = base[(size_t)key * ldkv + D + dch];
            }
        }
        *(U16x8*)(o + ((size_t)u * 64 + lane) * 8) = v;
    }
}

template <int D>
int pack_stage(const bf16_t* W, int ldw, const std::vector<int>& tab, bf16_t* out_units, hipStream_t stream, std::vector<void*>& tmp) {
    const int n = (int)(tab.size() / 17);
    int* d = nullptr;
    MKD_HIP_CHECK(hipMalloc((void**)&d, tab.size() * sizeof(int)));
    tmp.push_back(d);
    MKD_HIP_CHECK(hipMemcpyAsync(d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(tfm_pack_units_kernel, dim3(n), dim3(64), 0, stream, W, ldw, d, out_units);
    MKD_LAUNCH_CHECK("tfm_pack_units_kernel");
    return 0;
}

template <int D>
int pack_weights_t(const TfmTailWeights& s, bf16_t* wpk, float* vec, hipStream_t stream) {
    using C = TfmCfg<D>;
    std::vector<void*> tmp;
    auto first_frag = [](int w) { return w < C::NWA ? w * C::NFA : C::NWA * C::NFA + (w - C::NWA) * C::NFB; };
    auto nfrag = [](int w) { return w < C::NWA ? C::NFA : C::NFB; };
    // an N = D product with KS k-steps: wave streams one after the other, unit order [ks][f]; fragment fr = rows 16 fr .. + 15
    auto plain = [&](int KS) {
        std::vector<int> tab;
        for (int w = 0; w < TFM_NW; ++w)
            for (int ks = 0; ks < KS; ++ks)
                for (int f = 0; f < nfrag(w); ++f) {
                    for (int i = 0; i < 16; ++i) tab.push_back(16 * (first_frag(w) + f) + i);
                    tab.push_back(ks);
                }
        return tab;
    };
    int rc = pack_stage<D>(s.w_o1, D, plain(C::KSD), wpk + (size_t)C::OFF_O1 * 512, stream, tmp);
    if (!rc) rc = pack_stage<D>(s.w_q, D, plain(C::KSD), wpk + (size_t)C::OFF_Q * 512, stream, tmp);
    if (!rc) rc = pack_stage<D>(s.w_o2, D, plain(C::KSD), wpk + (size_t)C::OFF_O2 * 512, stream, tmp);
    if (!rc) rc = pack_stage<D>(s.w_m, 5 * D, plain(C::KSM), wpk + (size_t)C::OFF_M * 512, stream, tmp);
    if (!rc) {
        // GEGLU projection, source rows interleaved (2 j = value_j, 2 j + 1 = gate_j): chunk c, wave w, fragments
        // (value, gate) of output block 2 w, then of 2 w + 1
        std::vector<int> tab;
        for (int c = 0; c < C::NCH; ++c)
            for (int w = 0; w < TFM_NW; ++w)
                for (int ks = 0; ks < C::KSD; ++ks)
                    for (int f = 0; f < 4; ++f) {
                        const int j0 = TFM_CH * c + 16 * (2 * w + (f >> 1));
                        for (int i = 0; i < 16; ++i) tab.push_back(2 * (j0 + i) + (f & 1));
                        tab.push_back(ks);
                    }
        rc = pack_stage<D>(s.w_g, D, tab, wpk + (size_t)C::OFF_G * 512, stream, tmp);
    }
    auto cp = [&](int off, const float* src, int n) { return hipMemcpyAsync(vec + off, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, stream); };
    auto cp2 = [&](int off, const float* src, int n) { return hipMemcpy2DAsync(vec + off, 4, src, 8, 4, n, hipMemcpyDeviceToDevice, stream); };     // every second float
    if (!rc) {
        MKD_HIP_CHECK(cp(C::V_BO1, s.b_o1, D));
        MKD_HIP_CHECK(cp(C::V_SQ, s.s_q, D));
        MKD_HIP_CHECK(cp(C::V_BQ, s.b_q, D));
        MKD_HIP_CHECK(cp(C::V_BO2, s.b_o2, D));
        MKD_HIP_CHECK(cp2(C::V_SV, s.s_g, 4 * D));
        MKD_HIP_CHECK(cp2(C::V_BV, s.b_g, 4 * D));
        MKD_HIP_CHECK(cp2(C::V_SG, s.s_g + 1, 4 * D));
        MKD_HIP_CHECK(cp2(C::V_BG, s.b_g + 1, 4 * D));
        MKD_HIP_CHECK(cp(C::V_BM, s.b_m, D));
    }
    hipError_t e = hipStreamSynchronize(stream);
    for (void* p : tmp) hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) return mkd_fail(-2, std::string("tfm_tail pack: ") + hipGetErrorString(e));
    return 0;
}

}  // namespace

static long long* g_tfm_trace = nullptr;
void tfm_tail_set_trace(long long* buf) { g_tfm_trace = buf; }       // MKD_TFM_TRACE builds: device buffer [workgroups][8][32]

bool tfm_tail_supported(int d, int heads, int T, int Tk) {
    return d == 320 && heads == TFM_NW && T > 0 && T % TFM_TM == 0 && Tk > 0 && Tk <= 16 * TFM_KF;
}
size_t tfm_tail_weight_bytes(int d) { return d == 320 ? (size_t)TfmCfg<320>::UNITS * 1024 : 0; }
size_t tfm_tail_vec_bytes(int d) { return d == 320 ? (size_t)TfmCfg<320>::V_TOTAL * sizeof(float) : 0; }
size_t tfm_tail_kv_bytes(int d, int batch) { return d == 320 ? (size_t)batch * TFM_NW * TfmCfg<320>::KV_UNITS * 1024 : 0; }
double tfm_tail_flops(int d, int M, int Tk) { return 2.0 * M * (16.0 * d * d + 2.0 * Tk * d); }

int tfm_tail_pack_weights(int d, const TfmTailWeights& src, bf16_t* wpk, float* vec, hipStream_t stream) {
    if (d != 320) return mkd_fail(-4, "tfm_tail: only d = 320 is built");
    return pack_weights_t<320>(src, wpk, vec, stream);
}

int launch_tfm_tail_pack_kv(int d, const bf16_t* kv, int ldkv, int batch, int Tk, bf16_t* out, hipStream_t stream) {
    if (d != 320) return mkd_fail(-4, "tfm_tail: only d = 320 is built");
    if (Tk <= 0 || Tk > 16 * TFM_KF || ldkv % 8) return mkd_fail(-1, "tfm_tail kv pack: 1..80 keys, ld multiple of 8");
    hipLaunchKernelGGL(tfm_pack_kv_kernel<320>, dim3(batch * TFM_NW), dim3(256), 0, stream, kv, ldkv, Tk, out);
    MKD_LAUNCH_CHECK("tfm_pack_kv_kernel");
    return 0;
}

int launch_tfm_tail(int d, const bf16_t* wpk, const float* vec, const bf16_t* a1, int lda, const bf16_t* h0, int ldh, const bf16_t* xin, int ldx,
                    const bf16_t* kvp, bf16_t* out, int ldo, int M, int T, int Tk, hipStream_t stream) {
    if (!tfm_tail_supported(d, TFM_NW, T, Tk) || M <= 0 || M % TFM_TM) return mkd_fail(-4, "tfm_tail: unsupported shape");
    if (lda % 8 || ldh % 4 || ldx % 4 || ldo % 4) return mkd_fail(-1, "tfm_tail: strides must be multiples of 8 (a1) / 4");
    using C = TfmCfg<320>;
    TfmTailArgs a;
    a.wpk = (const bf16x8*)wpk; a.vec = vec; a.a1 = a1; a.lda = lda; a.h0 = h0; a.ldh = ldh; a.xin = xin; a.ldx = ldx;
    a.kvp = (const bf16x8*)kvp; a.out = out; a.ldo = ldo; a.T = T; a.Tk = Tk;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)C::DH);
    a.trace = g_tfm_trace;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)tfm_tail_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(tfm_tail LDS): ") + hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL(tfm_tail_kernel<320>, dim3(M / TFM_TM), dim3(64 * TFM_NW), C::LDS, stream, a);
    MKD_LAUNCH_CHECK("tfm_tail_kernel");
    return 0;
}

size_t tfm_head_weight_bytes(int d) { return d == 320 ? (size_t)TfmHeadCfg<320>::UNITS * 1024 : 0; }
size_t tfm_head_vec_bytes(int d) { return d == 320 ? (size_t)TfmHeadCfg<320>::V_TOTAL * sizeof(float) : 0; }

int tfm_head_pack_weights(int d, const TfmHeadWeights& s, bf16_t* wpk, float* vec, hipStream_t stream) {
    if (d != 320) return mkd_fail(-4, "tfm_head: only d = 320 is built");
    using C = TfmCfg<320>; using H = TfmHeadCfg<320>;
    constexpr int D = 320;
    std::vector<void*> tmp;
    auto first_frag = [](int w) { return w < C::NWA ? w * C::NFA : C::NWA * C::NFA + (w - C::NWA) * C::NFB; };
    auto nfrag = [](int w) { return w < C::NWA ? C::NFA : C::NFB; };
    auto plain = [&](int row0) {
        std::vector<int> tab;
        for (int w = 0; w < TFM_NW; ++w)
            for (int ks = 0; ks < C::KSD; ++ks)
                for (int f = 0; f < nfrag(w); ++f) {
                    for (int i = 0; i < 16; ++i) tab.push_back(row0 + 16 * (first_frag(w) + f) + i);
                    tab.push_back(ks);
                }
        return tab;
    };
    int rc = pack_stage<D>(s.w_pi, D, plain(0), wpk + (size_t)H::OFF_PI * 512, stream, tmp);
    for (int t = 0; t < 3 && !rc; ++t)
        rc = pack_stage<D>(s.w_qkv, D, plain(t * D), wpk + (size_t)(H::OFF_QKV + t * C::NFR * C::KSD) * 512, stream, tmp);
    auto cp = [&](int off, const float* src, int n) { return hipMemcpyAsync(vec + off, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, stream); };
    if (!rc) {
        MKD_HIP_CHECK(cp(H::V_BPI, s.b_pi, D));
        MKD_HIP_CHECK(cp(H::V_S, s.s_qkv, 3 * D));
        MKD_HIP_CHECK(cp(H::V_B, s.b_qkv, 3 * D));
        MKD_HIP_CHECK(cp(H::V_G, s.gn_gamma, D));
        MKD_HIP_CHECK(cp(H::V_BETA, s.gn_beta, D));
    }
    hipError_t e = hipStreamSynchronize(stream);
    for (void* p : tmp) (void)hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) return mkd_fail(-2, std::string("tfm_head pack: ") + hipGetErrorString(e));
    return 0;
}

int launch_tfm_head(int d, const bf16_t* wpk, const float* vec, const bf16_t* x, int ldx, const float* gn_partials, int gn_chunks, float gn_eps,
                    bf16_t* h0, bf16_t* qkv, int M, int T, hipStream_t stream) {
    if (d != 320 || T <= 0 || T % TFM_TM || M <= 0 || M % T) return mkd_fail(-4, "tfm_head: unsupported shape");
    if (ldx % 8 || gn_chunks <= 0 || gn_chunks > 64 || !gn_partials) return mkd_fail(-1, "tfm_head: x stride must be a multiple of 8; GroupNorm partials of <= 64 row chunks required");
    using H = TfmHeadCfg<320>;
    TfmHeadArgs a;
    a.wpk = (const bf16x8*)wpk; a.vec = vec; a.x = x; a.ldx = ldx; a.part = gn_partials; a.nchunks = gn_chunks; a.gn_eps = gn_eps;
    a.h0 = h0; a.qkv = qkv; a.T = T;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)tfm_head_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, H::LDS);
        if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(tfm_head LDS): ") + hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL(tfm_head_kernel<320>, dim3(M / TFM_TM), dim3(64 * TFM_NW), H::LDS, stream, a);
    MKD_LAUNCH_CHECK("tfm_head_kernel");
    return 0;
}
