// softmax(Q K^T * scale) V per (sample, head) on the matrix cores, flash-style (no T x T matrix in HBM).
// Replaces CrossAttention.forward's einsum / softmax / einsum for attn1 (self,
// 16..4096 tokens) and attn2 (cross, 77 context tokens), reached from diffmk/makeup_diffuse.py:164-168.
//
// One workgroup = NW (4 or 8) wavefronts = 16*NW queries of one (sample, head); each wave owns 16 queries.  Long
// sequences use NW = 8: every K/V tile fetched from L2 then serves 128 queries (K/V re-reads, not MFMA, bound T >= 1024).
// The next K/V tile is prefetched into registers while the current one is being consumed.
// K/V tiles of 64 keys are staged through LDS, both row-major; PV's V^T operand comes from ds_read_b64_tr_b16, the
// gfx950 transposing LDS read (no software transpose on the store side).  The score product is computed TRANSPOSED (S^T = K Q^T with v_mfma_f32_16x16x32_bf16)
// so a lane owns ONE query column: the softmax row-reduction is 15 in-register max/sum ops + two
// cross-lane shuffles, and S^T's accumulator registers are, after a bf16 pack, directly the B
// operand of O^T += V^T P^T (k-order permuted identically in both operands) - P never touches LDS.
// Head dims 40 / 80 / 160 are zero-padded to the 32-deep MFMA K step inside LDS only.
#include "mkd_common.h"
#include <cstdlib>

namespace {

// keys per tile: 64 (self-attention streams tiles with an online softmax) or 96 (KT template parameter): the 77 context keys of
// the cross-attention then fit ONE tile - no second, 80 %-masked tile, no rescale of O, one barrier pair instead of two.
// MKD_ATTN_TAIL (compile time): the last 8 / 16 channels of dh = 40 / 80 go through one 16-deep MFMA (40 padded to 48, not 64), with
// an accumulator of its own (see the hazard note at the MFMA).  1 (default): in the one-tile cross-attention kernel (KT = 96) only,
// 2: everywhere, 0: nowhere.  Measured at batch 8 (tools/exp_r3_attn_tail.sh, us per launch without / with): cross-attention
// 4096 x 77 dh 40 24.6 / 22.1, 1024 x 77 8.2 / 7.3, 256 x 77 dh 80 5.7 / 5.5; self-attention 4096^2 dh 40 448 / 462, 1024^2 32.0 /
// 33.2 (the streaming kernel is VALU-bound and the join costs 16 adds per tile), dh 80 45.2 / 45.0.
#ifndef MKD_ATTN_TAIL
#define MKD_ATTN_TAIL 1
#endif

template <int DH, int KT, int MSUM = 0, int NWV = 4>
struct AttnCfg {
    static constexpr int TAIL = ((MKD_ATTN_TAIL == 2 || (MKD_ATTN_TAIL == 1 && KT == 96)) && (DH % 32)) ? 1 : 0;   // one 16-deep step (v_mfma_f32_16x16x16_bf16) for dh = 8, 16, 40, 80
    static constexpr int KS = TAIL ? DH / 32 : (DH + 31) / 32;          // 32-deep k-steps of QK^T
    static constexpr int DHP = 32 * KS + 16 * TAIL;      // QK^T contraction depth, zero padded (40 -> 48, not 64)
    static constexpr int DVP = (DH + 15) / 16 * 16;      // output rows of O^T (padded)
    static constexpr int MD = DVP / 16;                  // O^T fragments
    // Row sums of P on the matrix cores (MSUM): when the padded V tile has a spare column (dh = 40 -> 48), column DH holds 1.0 for
    // every valid key, so row DH of O^T = V^T P^T accumulates sum_key bf16(p) - the softmax denominator - inside the P.V MFMAs that
    // run anyway: 16 v_add_f32 per tile leave a VALU-bound loop (rocprofv3, 4096 keys: the SIMDs' vector ALUs are busy 65-70 % of the
    // kernel, the matrix cores 26 %: profiles/exp_r4_attn_pmc_*.csv), and the denominator is the sum of exactly the rounded weights
    // the numerator uses.  The running rescale of O rescales it with the rest.  Chosen by the launcher for >= 2048 keys (dh 40,
    // 4096 keys: 445 -> 427 us).  Also tried in round 4 and dropped: two K/V tile buffers with ONE barrier per tile (+7 % at dh 40,
    // equal at dh 80).
    static constexpr int ONES = (MSUM && DVP > DH) ? 1 : 0;
    // MSUM == 2, additionally (needs a spare column in the padded K tile too, dh 40 -> 64): the softmax's  s c - m  on the matrix cores.
    // Q is pre-scaled by c = scale log2(e), column DH of the K tile holds 1.0 and element DH of the query's Q fragment holds -m', the
    // row's REFERENCE (a bf16 value, so that it can live in the fragment): the MFMAs deliver s c - m' and the vector ALUs only take
    // exp2 and a max per score.  The reference is updated lazily (cdna_hip_programming.md T13): only when some score of the tile
    // exceeds it by more than OFFS_THRESH (p <= 2^OFFS_THRESH is harmless in bf16 / fp32), or on the first tile; then the tile's scores
    // are corrected, O (with its denominator row) rescaled and the Q element rewritten.  Any reference gives the same softmax as long as
    // numerator, denominator and rescale use the same one.
    static constexpr int OFFS = (MSUM == 2 && ONES && KS * 32 > DH && !TAIL) ? 1 : 0;
    static constexpr float OFFS_THRESH = 6.0f;
    // ASYNC: unconditional tile loads, padding applied at the LDS store (see the kernel).  Measured per shape on one box (round 4, us,
    // conditional / unconditional): 4096^2 dh 40 419 / 387, 1024^2 dh 40 32.8 / 29.4, 256^2 dh 80 11.1 / 9.8 - but 1024^2 dh 80 46.0 / 47.7
    // (one more live 128-bit register per thread: a wave of occupancy), 256^2 dh 160 14.0 / 14.7, one-tile cross-attention +1-2 %.
    static constexpr int ASYNC = ((KT == 64 || KT == 128) && (DH < 80 || (DH == 80 && NWV == 4))) ? 1 : 0;
    static constexpr int KROW = DHP * 2 + 16;            // K tile row stride in bytes (pad: bank spread)
    // V tile stays ROW-MAJOR [key][d] and is read transposed by ds_read_b64_tr_b16 (gfx950).  A 32-lane half reads 8
    // consecutive key rows x 4 column quads: conflict-free when the row stride in dwords is 8 * odd.
    static constexpr int VR0 = DVP / 2;
    static constexpr int VROW = 4 * (VR0 + ((8 - VR0 % 16) + 16) % 16);
    static constexpr int KBYTES = KT * KROW;
    static constexpr int VBYTES = KT * VROW;
    static_assert(DH % 8 == 0 && (!TAIL || DH % 32 <= 16), "head dim: multiple of 8 (tail of at most 16 past the 32-deep steps)");
};

#ifdef MKD_ATTN_TRACE
__device__ long long* g_attn_trace_dev = nullptr;      // experiment builds: [workgroup][wave][8] cycle sums per phase (tools/exp_r4_attn_trace.py)
#define ATT_T(i) do { const long long now_ = clock64(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define ATT_T(i) do { } while (0)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

template <int DH, int NW, int KT, int MSUM = 0>
__global__ __launch_bounds__(64 * NW) void attention_kernel(const Pair<AttnIo> io, int ldq, int ldk, int ldv, int ldo,
                                                        int Tq, int Tk, int heads, float scale_log2e, int causal) {
    // grouped launch: grid z selects the problem (same geometry, own tensors)
    const bf16_t* __restrict__ const Q = io.g[blockIdx.z].q; const bf16_t* __restrict__ const K = io.g[blockIdx.z].k;
    const bf16_t* __restrict__ const V = io.g[blockIdx.z].v; bf16_t* __restrict__ const O = io.g[blockIdx.z].o;
    using C = AttnCfg<DH, KT, MSUM, NW>;
    static_assert(KT % 32 == 0, "key tile: whole 32-key PV steps");
    constexpr int KB = KT / 16;          // 16-key blocks of S^T
    extern __shared__ __attribute__((aligned(16))) char smem[];        // C::KBYTES + C::VBYTES (66 KB for dh 160 with 96-key tiles)
    char* ks = smem;
    char* vs = smem + C::KBYTES;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int qc = lane & 15;          // query column owned by this lane
    const int g = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = blockIdx.x * (16 * NW) + w * 16;
    const int qi = q0 + qc;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[qi][32*s + 8*g + j]; tail step: Q[qi][32*KS + 4*g + j]
    bf16x8 qf[C::KS ? C::KS : 1];
    bf16x4v qt = {0, 0, 0, 0};
    {
        const bf16_t* qrow = Q + ((size_t)b * Tq + (qi < Tq ? qi : 0)) * ldq + h * DH;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            U16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t.v[j] = 0;
            if (qi < Tq && 32 * s + 8 * g < DH) t = *(const U16x8*)(qrow + 32 * s + 8 * g);
            qf[s] = __builtin_bit_cast(bf16x8, t);
        }
        if (C::TAIL) {
            const int d = 32 * C::KS + 4 * g;
            U16x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t.v[j] = 0;
            if (qi < Tq && d < DH) t = *(const U16x4*)(qrow + d);
            qt = __builtin_bit_cast(bf16x4v, t);
        }
    }

    if (C::OFFS) {          // pre-scaled queries: the MFMA delivers scores in the log2 domain
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (__bf16)((float)qf[s][j] * scale_log2e);
    }
    f32x4 oacc[C::MD];
#pragma unroll
    for (int i = 0; i < C::MD; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    float m_ref = 0.f;        // OFFS: the row's reference, held as -m_ref in element DH of the Q fragment (0 until the first tile sets it)

    const bf16_t* kbase = K + (size_t)b * Tk * ldk + h * DH;
    const bf16_t* vbase = V + (size_t)b * Tk * ldv + h * DH;
    const int ntiles = (Tk + KT - 1) / KT;
    const int key_lim = causal ? (qi < Tk ? qi + 1 : Tk) : Tk;      // causal (CLIP text): query i sees keys 0..i; key 0 is always visible

    // staging geometry: chunk = 8 channels (16 B) of one key; thread tid owns chunks tid, tid + 64*NW, ...
    constexpr int NT = 64 * NW;
    constexpr int KCHUNKS = KT * (C::DHP / 8), VCHUNKS = KT * (C::DVP / 8);
    constexpr int KPT = (KCHUNKS + NT - 1) / NT, VPT = (VCHUNKS + NT - 1) / NT;
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    u32x4v kreg[KPT], vreg[VPT];          // (opaque 128-bit values until they are stored: nothing to unpack, nothing for the compiler to hoist)
    // The tile's global loads are UNCONDITIONAL (out-of-range chunks read the head's first row instead): a load under a divergent
    // branch makes the compiler wait for it at the join (s_waitcnt vmcnt(0) right behind the "prefetch": a full L2 / HBM round trip
    // exposed per key tile - 1425 of 4150 cycles per wave and tile in the phase trace, profiles/exp_r4_attn_trace.txt).  Zero padding,
    // the ones column (ONES) and the reference column (OFFS) are applied when the registers are stored to LDS.
    auto prefetch = [&](int key0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (C::DHP / 8), c = idx - r * (C::DHP / 8);
            const bool ok = idx < KCHUNKS && key0 + r < Tk && c * 8 < DH;
            const size_t off = ok ? (size_t)(key0 + r) * ldk + c * 8 : 0;
            if (C::ASYNC) kreg[i] = *(const u32x4v*)(kbase + off);
            else { const u32x4v z = {0u, 0u, 0u, 0u}; kreg[i] = z; if (ok) kreg[i] = *(const u32x4v*)(kbase + off); }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (C::DVP / 8), c = idx - r * (C::DVP / 8);
            const bool ok = idx < VCHUNKS && key0 + r < Tk && c * 8 < DH;
            const size_t off = ok ? (size_t)(key0 + r) * ldv + c * 8 : 0;
            if (C::ASYNC) vreg[i] = *(const u32x4v*)(vbase + off);
            else { const u32x4v z = {0u, 0u, 0u, 0u}; vreg[i] = z; if (ok) vreg[i] = *(const u32x4v*)(vbase + off); }
        }
    };
    // the chunk as it goes to LDS: zero outside the tensor; 1.0 in column DH of valid keys where the kernel uses that column
    auto kfix = [&](int i, int key0) {
        const int idx = tid + i * NT;
        const int r = idx / (C::DHP / 8), c = idx - r * (C::DHP / 8);
        u32x4v d = kreg[i];
        if (C::ASYNC) {
            asm volatile("" : "+v"(d));           // the value is looked at HERE, one tile after its load was issued
            const bool ok = key0 + r < Tk && c * 8 < DH;
            const u32x4v z = {0u, 0u, 0u, 0u};
            d = ok ? d : z;
        }
        if (C::OFFS && key0 + r < Tk && c * 8 == DH) d.x = 0x3F80u;       // bf16 1.0 in column DH (the rest of the chunk is padding): adds the query's -m_ref
        return d;
    };
    auto vfix = [&](int i, int key0) {
        const int idx = tid + i * NT;
        const int r = idx / (C::DVP / 8), c = idx - r * (C::DVP / 8);
        u32x4v d = vreg[i];
        if (C::ASYNC) {
            asm volatile("" : "+v"(d));
            const bool ok = key0 + r < Tk && c * 8 < DH;
            const u32x4v z = {0u, 0u, 0u, 0u};
            d = ok ? d : z;
        }
        if (C::ONES && key0 + r < Tk && c * 8 == DH) d.x = 0x3F80u;       // bf16 1.0 in column DH
        return d;
    };
    // registers -> LDS: K tile [KT][DHP] and V tile [KT][DVP], both row-major and zero padded
    auto stage = [&](char* kd, char* vd, int key0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (C::DHP / 8), c = idx - r * (C::DHP / 8);
            if (idx < KCHUNKS) *(u32x4v*)(kd + r * C::KROW + c * 16) = kfix(i, key0);
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (C::DVP / 8), c = idx - r * (C::DVP / 8);
            if (idx < VCHUNKS) *(u32x4v*)(vd + r * C::VROW + c * 16) = vfix(i, key0);
        }
    };
    // DB: two tile buffers - tile t + 1 is stored while tile t is consumed, ONE barrier per tile
#ifdef MKD_ATTN_DB
    constexpr bool DB = C::OFFS != 0;
#else
    constexpr bool DB = false;
#endif
    constexpr int TILE_BYTES = C::KBYTES + C::VBYTES;
    prefetch(0);
    if (DB) {
        stage(smem, smem + C::KBYTES, 0);
        if (ntiles > 1) prefetch(KT);
    }
#ifdef MKD_ATTN_TRACE
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
#endif

    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        if (DB) { ks = smem + (t & 1) * TILE_BYTES; vs = ks + C::KBYTES; }
        ATT_T(6);
        __syncthreads();                                   // previous tile fully consumed (DB: and this tile stored by everyone)
        ATT_T(0);
        if (DB) {
            if (t + 1 < ntiles) {
                char* const kn = smem + ((t + 1) & 1) * TILE_BYTES;
                stage(kn, kn + C::KBYTES, key0 + KT);
                ATT_T(1);
                if (t + 2 < ntiles) prefetch(key0 + 2 * KT);
            }
            ATT_T(3);
        } else {
            stage(ks, vs, key0);
            ATT_T(1);
            __syncthreads();
            ATT_T(2);
            if (t + 1 < ntiles) prefetch(key0 + KT);       // in flight while this tile is consumed
            ATT_T(3);
        }

        // ---- S^T[key][q] for KB key blocks of 16 ------------------------------------------------
        f32x4 st[KB];
#pragma unroll
        for (int mf = 0; mf < KB; ++mf) {
            st[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
            const char* krow = ks + (16 * mf + qc) * C::KROW;
            // The 16-deep tail step keeps an accumulator of its OWN and joins the 32-deep chain with a VALU add.  Chaining it
            // through SrcC is the hazard that made this variant sporadically wrong in the 8-wave build: an XDL result read as
            // SrcC of a DIFFERENT opcode with a different vDst gets no forwarding and needs the producer's full pass count in
            // wait states (CDNA3 ISA 4.5, table "XDL write VGPR -> XDL read SrcC, overlapped, different vDst": passes + 1); the
            // compiler emitted none between v_mfma_f32_16x16x16_bf16 and v_mfma_f32_16x16x32_bf16.  XDL write -> VALU read is
            // the dependency every GEMM epilogue has, and its wait states are inserted correctly.
            f32x4 tl = f32x4{0.f, 0.f, 0.f, 0.f};
            if (C::TAIL) {
                const bf16x4v kt = *(const bf16x4v*)(krow + (32 * C::KS + 4 * g) * 2);
                tl = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, kt), __builtin_bit_cast(s16x4, qt), tl, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x8 kf = *(const bf16x8*)(krow + (32 * s + 8 * g) * 2);
                st[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], st[mf], 0, 0, 0);
            }
            if (C::TAIL) st[mf] += tl;
        }
        // lane holds RAW scores of query qc for keys key0 + 16*mf + 4*g + r.  The softmax runs in the log2 domain with the
        // scale folded into one FMA per score: p = exp2(s*c - m), m = running max of s*c (c = scale*log2(e) > 0).
        // VALU, not MFMA, bounds this kernel (16 scores per lane per tile), so every per-score op counts: masking only
        // on a partial / causal tile, raw v_exp_f32, O rescaled only when some row's max moved.
#ifdef MKD_ATTN_TRACE
        asm volatile("" :: "v"(st[0]), "v"(st[KB - 1]));
#endif
        ATT_T(4);
        if (C::OFFS) {
            // st = s c - m_ref already.  Mask a partial last tile, take the tile's maximum (only to detect a reference that has
            // fallen too far behind), exponentiate.
            float mx = -INFINITY;
            if (key0 + KT > Tk) {
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = (key0 + 16 * mf + 4 * g + r < Tk) ? st[mf][r] : -INFINITY;
                        st[mf][r] = sc;
                        mx = fmaxf(mx, sc);
                    }
            } else {
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[mf][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const bool need = t == 0 || mx > C::OFFS_THRESH;
            if (__any(need)) {                                     // wave-uniform; rare after the first tiles
                const float m_new = need ? (float)(__bf16)(m_ref + mx) : m_ref;          // the new reference must be a bf16 value
                const float delta = m_new - m_ref;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref = m_new;
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[mf][r] -= delta;
#pragma unroll
                for (int i = 0; i < C::MD; ++i) oacc[i] *= alpha;
                if (g == (DH % 32) / 8) qf[DH / 32][DH % 8] = (__bf16)(-m_ref);
            }
#pragma unroll
            for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[mf][r] = __builtin_amdgcn_exp2f(st[mf][r]);
        } else {
            float mx = -INFINITY;
            if (causal || key0 + KT > Tk) {
    #pragma unroll
                for (int mf = 0; mf < KB; ++mf)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + 16 * mf + 4 * g + r;
                        const float s = key < key_lim ? st[mf][r] : -INFINITY;
                        st[mf][r] = s;
                        mx = fmaxf(mx, s);
                    }
            } else {
    #pragma unroll
                for (int mf = 0; mf < KB; ++mf)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[mf][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx * scale_log2e);   // finite: the first tile always holds a visible key
            float psum = 0.f;
    #pragma unroll
            for (int mf = 0; mf < KB; ++mf)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[mf][r], scale_log2e, -m_new));
                    st[mf][r] = pv;
                    if (!C::ONES) psum += pv;
                }
            if (__any(m_new != m_run)) {                           // wave-uniform: after the first tiles the max rarely moves
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
    #pragma unroll
                for (int i = 0; i < C::MD; ++i) oacc[i] *= alpha;
            }
            l_run += psum;

        }

        ATT_T(5);
        // ---- O^T[d][q] += V^T[d][key'] P^T[key'][q]; key'(g, j) = 32*s + (j<4 ? 4g+j : 16+4g+j-4) -----
#pragma unroll
        for (int s = 0; s < KT / 32; ++s) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (__bf16)st[2 * s][j];
                pf[4 + j] = (__bf16)st[2 * s + 1][j];
            }
            // V^T fragment through the hardware transpose read: a 16-lane group fetches a 4-key x 16-channel block and
            // lane i receives channel 16*md + i of those 4 keys (lane 4q + p supplies the address of key row q, quad p)
            const char* vblk = vs + (32 * s + 4 * g + (qc >> 2)) * C::VROW + (qc & 3) * 8;
#pragma unroll
            for (int md = 0; md < C::MD; ++md) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + md * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 16 * C::VROW + md * 32));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[md], 0, 0, 0);
            }
        }
    }

#ifdef MKD_ATTN_TRACE
    ATT_T(6);
    if (g_attn_trace_dev && lane == 0) {
        long long* o_ = g_attn_trace_dev + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + w) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = tacc[i];
    }
#endif
    if (C::ONES) {
        l_run = __shfl(oacc[DH / 16][DH % 4], 16 * ((DH % 16) / 4) + qc, 64);      // row DH of O^T: lane group (DH % 16) / 4, register DH % 4
    } else {
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_run;
    if (qi < Tq) {
        bf16_t* orow = O + ((size_t)b * Tq + qi) * ldo + h * DH;
#pragma unroll
        for (int md = 0; md < C::MD; ++md) {
            const int d = 16 * md + 4 * g;
            if (d < DH) {
                U16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.v[r] = f32_to_bf16(oacc[md][r] * inv);
                *(U16x4*)(orow + d) = o;
            }
        }
    }
}


// ---- long self-attention, d_head 40: K / V tiles by LDS-DMA ---------------------------------------------------------------------
// The streaming kernel above spends 40 % of a wave's tile time between its MFMA / softmax phases: the tile's global loads land in
// registers, are patched (zero padding, the ones columns) and stored to LDS between two barriers (profiles/exp_r4_attn_trace_after.txt:
// barrier 419 | LDS stores 421 | barrier 214 | prefetch issue 415 of 3655 cycles per wave and 128-key tile).  Here the tile goes
// global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write), into one of TWO buffers, one barrier per tile:
//   K tile  [KT][40] bf16, rows of 80 B (5 chunks of 16 B; slot p of the tile = row p / 5, chunk p % 5)
//   V tile  [KT][48] bf16, rows of 96 B (8 * odd dwords: conflict-free ds_read_b64_tr_b16); chunk 5 of every row comes from a constant
//           16-byte block {1.0, 0 ...}: column 40 = the ones column that makes row 40 of O^T the softmax denominator (AttnCfg::ONES)
// The padded K columns of the second k-step never exist in LDS: the Q fragment is zero beyond column 40 (so K may hold anything
// finite there) except element 40 = -m_ref (AttnCfg::OFFS), which meets K column 40 = 1.0 - lanes of group g >= 1 read their second
// K fragment from one constant 16-byte block in LDS instead of the tile.  Keys past the end of a partial last tile are CLAMPED to
// the last valid row on the load side (finite values) and masked to -inf / p = 0 by the softmax, as above.
// Numerics identical to attention_kernel<40, 8, KT, 2> (same MFMA order, same softmax): test_attention* compare both with torch.
__device__ __attribute__((aligned(16))) const uint16_t g_attn_ones_chunk[8] = {0x3F80u, 0, 0, 0, 0, 0, 0, 0};

template <int KT>
struct AttnDmaCfg {
    static constexpr int DH = 40, NW = 8, KROW = 80, VROW = 96;
    static constexpr int KSLOTS = KT * 5, VSLOTS = KT * 6, SLOTS = KSLOTS + VSLOTS;      // 16-byte slots of one tile buffer
    static constexpr int TILE_BYTES = SLOTS * 16;
    static constexpr int ROUNDS = (SLOTS + 64 * NW - 1) / (64 * NW);                      // LDS-DMA instructions per thread and tile
    static constexpr int ONES_OFF = 2 * TILE_BYTES;
    static constexpr int LDS = ONES_OFF + 16;
    static_assert(SLOTS % 64 == 0, "whole wave-instructions");
};

template <int KT>
__global__ __launch_bounds__(512) void attention_dma40_kernel(const Pair<AttnIo> io, int ldq, int ldk, int ldv, int ldo,
                                                              int Tq, int Tk, int heads, float scale_log2e) {
    using D = AttnDmaCfg<KT>;
    using C = AttnCfg<40, KT, 2, 8>;
    static_assert(C::OFFS && C::ONES && C::KS == 2 && C::MD == 3 && !C::TAIL, "the dh-40 reference-on-the-matrix-cores configuration");
    constexpr int DH = 40, NW = 8, KB = KT / 16, NT = 64 * NW;
    const bf16_t* __restrict__ const Q = io.g[blockIdx.z].q; const bf16_t* __restrict__ const K = io.g[blockIdx.z].k;
    const bf16_t* __restrict__ const V = io.g[blockIdx.z].v; bf16_t* __restrict__ const O = io.g[blockIdx.z].o;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / heads, h = bh - b * heads;
    const int q0 = blockIdx.x * (16 * NW) + w * 16;
    const int qi = q0 + qc;

    // staging geometry of this thread's slots (fixed per kernel): source row inside the tile, element offset, K or V, ones chunk
    const bf16_t* kbase = K + (size_t)b * Tk * ldk + h * DH;
    const bf16_t* vbase = V + (size_t)b * Tk * ldv + h * DH;
    int srow[D::ROUNDS], scol[D::ROUNDS];
    bool sv[D::ROUNDS], sone[D::ROUNDS];
#pragma unroll
    for (int i = 0; i < D::ROUNDS; ++i) {
        const int p = i * NT + tid;
        const bool isv = p >= D::KSLOTS;
        const int pp = isv ? p - D::KSLOTS : p;
        const int per = isv ? 6 : 5;
        srow[i] = pp / per; scol[i] = (pp - srow[i] * per) * 8; sv[i] = isv; sone[i] = isv && scol[i] == 40;
    }
    // (all of a tile's wave-instructions are issued together right behind the barrier: 516 cycles per wave and 128-key tile in the phase
    // trace, profiles/exp_r4_attn_trace_dma.txt - an LDS-DMA instruction costs its wave ~170 cycles of issue wherever it stands: spread
    // one by one between the QK^T / softmax / P.V phases the same cycles reappear inside those phases, 330 -> 341 us at 4096 keys)
    auto issue = [&](int buf, int key0) {
        char* dst = smem + buf * D::TILE_BYTES;
#pragma unroll
        for (int i = 0; i < D::ROUNDS; ++i) {
            if ((i * NT + w * 64) < D::SLOTS) {                      // wave-uniform: the last round is partial
                int row = key0 + srow[i];
                row = row < Tk ? row : Tk - 1;
                const bf16_t* src = sv[i] ? vbase + (size_t)row * ldv + scol[i] : kbase + (size_t)row * ldk + scol[i];
                unsigned long long sa = sone[i] ? (unsigned long long)g_attn_ones_chunk : (unsigned long long)src;
                asm volatile("" : "+v"(sa));                      // (one load per call site)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                                 (__attribute__((address_space(3))) void*)(dst + (i * NT + w * 64) * 16), 16, 0, 0);
            }
        }
    };
    const int ntiles = (Tk + KT - 1) / KT;
    issue(0, 0);
    if (tid == 0) *(uint4*)(smem + D::ONES_OFF) = make_uint4(0x3F80u, 0u, 0u, 0u);

    // Q fragments (B operand of S^T = K Q^T), pre-scaled by scale * log2(e): the MFMAs deliver scores in the log2 domain
    bf16x8 qf[2];
    {
        const bf16_t* qrow = Q + ((size_t)b * Tq + (qi < Tq ? qi : 0)) * ldq + h * DH;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            U16x8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t.v[j] = 0;
            if (qi < Tq && 32 * s + 8 * g < DH) t = *(const U16x8*)(qrow + 32 * s + 8 * g);
            qf[s] = __builtin_bit_cast(bf16x8, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (__bf16)((float)qf[s][j] * scale_log2e);
        }
    }
    f32x4 oacc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_ref = 0.f;
    const int k1s = g == 0 ? 16 * D::KROW : 0;          // second K fragment: the tile's chunk 4 (g = 0) or the constant ones block

#ifdef MKD_ATTN_TRACE
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
#endif
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * KT;
        ATT_T(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's part of tile t has landed
        ATT_T(0);
        __syncthreads();                                       // ... and everyone's; tile t - 1 is consumed: its buffer is free
        ATT_T(1);
        if (t + 1 < ntiles) issue((t + 1) & 1, key0 + KT);
        ATT_T(3);
        const char* ks = smem + (t & 1) * D::TILE_BYTES;
        const char* vs = ks + D::KSLOTS * 16;

        f32x4 st[KB];
        {
            const char* k0p = ks + qc * D::KROW + g * 16;
            const char* k1p = g == 0 ? ks + qc * D::KROW + 64 : smem + D::ONES_OFF;
#pragma unroll
            for (int mf = 0; mf < KB; ++mf) {
                st[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16x8 kf0 = *(const bf16x8*)(k0p + mf * 16 * D::KROW);
                const bf16x8 kf1 = *(const bf16x8*)(k1p + mf * k1s);
                st[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0], st[mf], 0, 0, 0);
                st[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1], st[mf], 0, 0, 0);
            }
        }
#ifdef MKD_ATTN_TRACE
        asm volatile("" :: "v"(st[0]), "v"(st[KB - 1]));
#endif
        ATT_T(4);
        // st = s c - m_ref (see attention_kernel, AttnCfg::OFFS)
        {
            float mx = -INFINITY;
            if (key0 + KT > Tk) {
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = (key0 + 16 * mf + 4 * g + r < Tk) ? st[mf][r] : -INFINITY;
                        st[mf][r] = sc;
                        mx = fmaxf(mx, sc);
                    }
            } else {
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[mf][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const bool need = t == 0 || mx > C::OFFS_THRESH;
            if (__any(need)) {
                const float m_new = need ? (float)(__bf16)(m_ref + mx) : m_ref;
                const float delta = m_new - m_ref;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref = m_new;
#pragma unroll
                for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[mf][r] -= delta;
#pragma unroll
                for (int i = 0; i < 3; ++i) oacc[i] *= alpha;
                if (g == 1) qf[1][0] = (__bf16)(-m_ref);
            }
#pragma unroll
            for (int mf = 0; mf < KB; ++mf)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[mf][r] = __builtin_amdgcn_exp2f(st[mf][r]);
        }
        ATT_T(5);
        // O^T[d][q] += V^T[d][key'] P^T[key'][q] (attention_kernel's operand order)
#pragma unroll
        for (int s = 0; s < KT / 32; ++s) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (__bf16)st[2 * s][j];
                pf[4 + j] = (__bf16)st[2 * s + 1][j];
            }
            const char* vblk = vs + (32 * s + 4 * g + (qc >> 2)) * D::VROW + (qc & 3) * 8;
#pragma unroll
            for (int md = 0; md < 3; ++md) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + md * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vblk + 16 * D::VROW + md * 32));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                oacc[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[md], 0, 0, 0);
            }
        }
    }
#ifdef MKD_ATTN_TRACE
    ATT_T(6);
    if (g_attn_trace_dev && lane == 0) {
        long long* o_ = g_attn_trace_dev + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + w) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = tacc[i];
    }
#endif
    const float l_run = __shfl(oacc[2][0], 32 + qc, 64);        // row 40 of O^T: fragment 2, lane group 2, register 0
    const float inv = 1.0f / l_run;
    if (qi < Tq) {
        bf16_t* orow = O + ((size_t)b * Tq + qi) * ldo + h * DH;
#pragma unroll
        for (int md = 0; md < 3; ++md) {
            const int d = 16 * md + 4 * g;
            if (d < DH) {
                U16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.v[r] = f32_to_bf16(oacc[md][r] * inv);
                *(U16x4*)(orow + d) = o;
            }
        }
    }
}

}  // namespace

int attn_set_trace(long long* buf) {
#ifdef MKD_ATTN_TRACE
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace_dev), &buf, sizeof(buf));
    return e == hipSuccess ? 0 : -2;
#else
    (void)buf; return 0;
#endif
}

int launch_attention(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv,
                     bf16_t* o, int ldo, int batch, int Tq, int Tk, int heads, int dh, float scale,
                     hipStream_t stream, int causal, const AttnIo* second) {
    if (Tq <= 0 || Tk <= 0 || batch <= 0 || heads <= 0) return mkd_fail(-1, "attention: empty problem");
    Pair<AttnIo> io;
    io.g[0] = AttnIo{q, k, v, o};
    MKD_PAIR_SET2(io, second ? *second : io.g[0]);
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || dh % 8) return mkd_fail(-1, "attention: strides/dh must be multiples of 8");
    const float sl = scale * 1.4426950408889634f;
    const bool wide = Tq >= 1024 && Tk >= 1024;             // long self-attention: 128 queries share each K/V tile
    static const bool kt96 = !(getenv("MKD_ATTN_KT96") && atoi(getenv("MKD_ATTN_KT96")) == 0);      // (A/B knob)
    const bool one96 = kt96 && !wide && !causal && Tk > 64 && Tk <= 96;      // cross-attention (77 context keys): one 96-key tile
#ifndef MKD_ATTN_KT_WIDE
#define MKD_ATTN_KT_WIDE 128     // keys per tile of the dh-40 long self-attention (round 4: 1024 keys 30.2 -> 28.6 us, 4096 keys equal; half the barriers per key)
#endif
#ifndef MKD_ATTN_OFFS_MIN
#define MKD_ATTN_OFFS_MIN 1024
#endif
    // softmax work moved onto the matrix cores in the long self-attention (AttnCfg::ONES / OFFS): 2 = reference subtraction and
    // denominators (head dims with spare K and V columns in their padded tiles: 40), 1 = denominators only, from 2048 keys
    int msum = (wide && !causal) ? ((Tk >= MKD_ATTN_OFFS_MIN && dh == 40) ? 2 : (Tk >= 2048 ? 1 : 0)) : 0;
#ifdef MKD_ATTN_TRACE
    if (const char* fm = getenv("MKD_ATTN_MODE")) msum = (wide && !causal) ? atoi(fm) : 0;      // (experiment builds only)
#endif
#ifdef MKD_ATTN_FORCE_MODE
    msum = (wide && !causal) ? MKD_ATTN_FORCE_MODE : 0;
#endif
    const int qb = wide ? 128 : 64;
    dim3 grid((Tq + qb - 1) / qb, batch * heads, second ? 2 : 1);
#define MKD_ATTN_LAUNCH(D, NWV, KTV, MS)                                                                      \
    do {                                                                                                      \
        constexpr int lds = AttnCfg<D, KTV>::KBYTES + AttnCfg<D, KTV>::VBYTES;                                  \
        static bool attr = false;                                                                             \
        if (lds > 64 * 1024 && !attr) {                                                                       \
            hipError_t e = hipFuncSetAttribute((const void*)attention_kernel<D, NWV, KTV, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return mkd_fail(-2, std::string("hipFuncSetAttribute(attention LDS): ") + hipGetErrorString(e));            \
            attr = true;                                                                                      \
        }                                                                                                     \
        hipLaunchKernelGGL((attention_kernel<D, NWV, KTV, MS>), grid, dim3(64 * NWV), lds, stream, io, ldq, ldk, ldv, ldo, Tq, Tk,     \
                           heads, sl, causal);                                                                \
    } while (0)
#define MKD_ATTN_CASE(D)                                                                                      \
    case D:                                                                                                   \
        if (wide && msum == 2) MKD_ATTN_LAUNCH(D, 8, MKD_ATTN_KT_WIDE, 2);                                    \
        else if (wide && msum) MKD_ATTN_LAUNCH(D, 8, 64, 1);                                                  \
        else if (wide) MKD_ATTN_LAUNCH(D, 8, 64, 0);                                                          \
        else if (one96) MKD_ATTN_LAUNCH(D, 4, 96, 0);                                                         \
        else MKD_ATTN_LAUNCH(D, 4, 64, 0);                                                                    \
        break;
    static const bool dma40 = !(getenv("MKD_ATTN_DMA") && atoi(getenv("MKD_ATTN_DMA")) == 0);      // (A/B knob)
    if (wide && msum == 2 && dh == 40 && dma40) {          // K / V tiles by LDS-DMA, two buffers, one barrier per tile
        using D = AttnDmaCfg<MKD_ATTN_KT_WIDE>;
        hipLaunchKernelGGL((attention_dma40_kernel<MKD_ATTN_KT_WIDE>), grid, dim3(512), D::LDS, stream, io, ldq, ldk, ldv, ldo, Tq, Tk, heads, sl);
        MKD_LAUNCH_CHECK("attention_dma40_kernel");
        return 0;
    }
    switch (dh) {
        MKD_ATTN_CASE(8)
        MKD_ATTN_CASE(16)
        MKD_ATTN_CASE(32)
        MKD_ATTN_CASE(40)
        MKD_ATTN_CASE(64)
        MKD_ATTN_CASE(80)
        MKD_ATTN_CASE(160)
        default:
            return mkd_fail(-4, "attention: unsupported head dim " + std::to_string(dh));
    }
#undef MKD_ATTN_CASE
#undef MKD_ATTN_LAUNCH
    MKD_LAUNCH_CHECK("attention_kernel");
    return 0;
}
