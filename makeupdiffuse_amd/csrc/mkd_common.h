// Shared declarations for libmkd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef uint16_t bf16_t;   // raw bf16 bits on the host side of the launchers

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void mkd_set_error(const std::string& msg);
int mkd_fail(int code, const std::string& msg);

#define MKD_HIP_CHECK(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return mkd_fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

#define MKD_LAUNCH_CHECK(what)                                                         \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess)                                                          \
            return mkd_fail(-2, std::string(what) + ": " + hipGetErrorString(_e));     \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even via the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// Activation math on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): an IEEE division costs ~10 VALU
// ops and libm erff ~40, and the element-wise kernels here are VALU-bound on the few CUs a GroupNorm can occupy.
__device__ __forceinline__ float sigmoid_scaled_f(float x, float k) {            // 1 / (1 + exp(-k x))
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.4426950408889634f * k)));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_scaled_f(x, 1.0f); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_scaled_f(x, 1.702f); }      // CLIP: x * sigmoid(1.702 x)
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 output step): GELU(x) = x/2 * (1 + erf(x / sqrt 2))
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float e = 1.0f - p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);      // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

struct __attribute__((aligned(16))) U16x8 { uint16_t v[8]; };
struct __attribute__((aligned(8)))  U16x4 { uint16_t v[4]; };
#endif

// ---------------------------------------------------------------------------------------------
// kernel launch parameter blocks (host + device)
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const float* bias;
    const float* rowbias; int ldrb; int rows_per_batch;
    const bf16_t* R; int ldr;
    float scale; int act;
    void* C; int ldc; int out_f32;
    int M, N, K;
    int conv; int Hin, Win, Cin, Hout, Wout, stride, up;
    float* ws; size_t ws_bytes; int splitk; int ksteps_per_split;     // split-K fp32 partial slabs: pointer and capacity
    int tile_h, tile_w, tile_imgs;          // spatial tile of the LDS-staged conv kernel (set by its launcher)
    const float* ln_s; float ln_eps;        // fused LayerNorm on the A rows: ln_s[n] = sum_k W'[n][k] (W' = W*gamma), else null
    const float* stat_in; int stat_in_slots; // ... whose row sums were emitted by the producer: [slots][M][2] (sum, sumsq)
    float* stat_out;                        // producer side: emit per-column-slot partial row sums of the (rounded) output
    // GroupNorm statistics of the tensor this GEMM writes (part of): gn_stat[sample][32][2] int64 fixed point (gemm_device.h);
    // gn_cg channels per group OF THE CONSUMER's tensor, gn_coff = column of this output inside it (concat halves), gn_hw rows per sample
    long long* gn_stat; int gn_cg; int gn_coff; int gn_hw;
    int defer_epilogue;                     // split-K launches: leave the fp32 partial slabs in ws; the caller's next kernel reduces them (launch_gn_from_slabs)
    int expect_splitk;                      // ... which was planned for exactly this many slabs: the launch fails if it resolves to another count (0: unchecked)
    // workgroup -> tile order for the 8 XCDs (workgroups are dealt round-robin to them in launch order, each XCD has its own L2):
    // 0 launch order; 1 an XCD owns runs of M-tiles of one N-tile (the weight tile lives in ONE L2); 2 runs of N-tiles of one M-tile
    int xcd_mode;
    int gz;               // z extent of ONE problem's grid (set by the launchers); a grouped launch stacks the second problem above it
    const bf16_t* zero;   // >= 16 bytes of zeros
    // conv3x3 with a folded 1x1 convolution of a SECOND input (ResBlock: out = conv3x3(h) + skip_connection(x), UPSTREAM ResBlock._forward):
    // K = 9 * Cin + K2, W = [W_conv | W_skip] (row-major over that K), the last K2 columns contract with row m of A2 ([M, K2] at lda2).
    // Gather / linear kernel only (stride 1, no upsampling); the tile plan is the one of the plain convolution (K - K2 in the tables)
    const bf16_t* A2; int lda2; int K2;
};

// ---------------------------------------------------------------------------------------------
// Grouped launches.  The ControlNet and the UNet encoder + middle block (reference diffmk/makeup_diffuse.py:164-168: the two
// calls of apply_model) run the SAME op sequence on identical shapes with different weights / inputs.  One launch can carry both
// problems: the kernel's argument block is a 2-entry table and a grid coordinate (z, or y for 1-D grids) selects the entry, so
// the pair costs ONE dependent dispatch instead of two on two contending queues.  Same kernels, same tiles, same order of
// operations per output: results are bit-identical to two separate launches.
// ---------------------------------------------------------------------------------------------
// MKD_PAIR_N = 1 (default build since round 4): single-entry argument tables.  The 2-entry form on every launch cost the DEFAULT
// two-chain plan 1.9 % (doubled kernarg blocks, a blockIdx.z-indexed indirection in every kernel: profiles/exp_r4_pair_form.txt,
// 5.39 vs 5.29 ms per evaluation, three alternating rounds), and its only user, the grouped encoder chain
// (MKD_ENC_GROUP, 10 % slower than two chains), is an experiment: build it with tools/build_variant.sh group -DMKD_PAIR_N=2.
#ifndef MKD_PAIR_N
#define MKD_PAIR_N 1
#endif
template <typename T> struct Pair { T g[MKD_PAIR_N]; };
#define MKD_PAIR_SET2(tab, val) do { if (MKD_PAIR_N > 1) (tab).g[MKD_PAIR_N - 1] = (val); } while (0)
typedef Pair<GemmArgs> GemmArgs2;
struct NormIo { const bf16_t* x; bf16_t* y; const float* gamma; const float* beta; };     // GroupNorm / LayerNorm operands of one problem
struct AttnIo { const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; };
struct ConvInIo { const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* add; };     // 4 -> C input convolution (same x for both nets)

// device-resident DDIM step state (hipGraph replay): tables of n_steps entries, counter runs n_steps-1 .. 0
constexpr int MKD_MAX_STEPS = 1024;
struct StepState {
    int counter;
    float cur[4];                          // sqrt(1/a_t), sqrt(a_prev), sqrt(1-a_prev), sqrt(1-a_t) of the current step
    int64_t timesteps[MKD_MAX_STEPS];
    float coef[4 * MKD_MAX_STEPS];
    // eta > 0 (cddim.py:74-78): x_prev += sigma_t * noise * temperature.  noise: [n_steps][n] fp32 on the device, row k = the draw of the
    // k-th EXECUTED step (index n_steps - 1 - k); null: deterministic loop.  cur_sigma / cur_row: this step's, set by step_setup_kernel
    float sigma[MKD_MAX_STEPS];
    const float* noise; float temperature; int n_steps;
    float cur_sigma; int cur_row;
};

// Time embedding of a sampling call: row `step` of tab[k] ([steps, n[k]] fp32, one table per net) is copied into every one of
// the `batch` rows of proj[k] ([batch, n[k]]: what the ResBlock epilogues read as their per-sample row bias).  n[k] % 4 == 0; n[k] = 0: absent.
struct TembSel { const float* tab[2]; float* proj[2]; int n[2]; int batch; };

// launchers (each only enqueues on `stream`)
// picks tile + split-K (a.splitk==0: auto).  second != null: grouped launch of two problems of identical geometry (see Pair)
int  launch_gemm(GemmArgs a, hipStream_t stream, const GemmArgs* second = nullptr);
bool gemm_same_geometry(const GemmArgs& a, const GemmArgs& b);     // may the two run as one grouped launch?
int  gemm_pick_splitk(int M, int N, int K, int conv, int stride, int up);
bool gemm_cfg_folds_second_input(int cfg);     // may this tile configuration take GemmArgs::A2 (gather / linear kernel)?
int  gemm_resolve(const GemmArgs& a, int* cfg, int* splitk);       // the (tile, split-K) launch_gemm will use for exactly these arguments
int  gemm_stat_slots(int M, int N, int K);   // column slots a linear GEMM of this shape writes row statistics in
int  gemm_tile_index(int M, int N, int K, int conv, int stride, int up);   // index into the tile-config table of kernels_gemm.hip
void gemm_set_xcd_mode(int mode);         // tests / experiments: 0 launch order, 1 / 2 contiguous runs per XCD (GemmArgs::xcd_mode)
void gemm_force_tile_cfg(int cfg);           // tuner/tests: force a tile config (-1 = heuristic)
void gemm_set_splitk_cap(int cap);
void gemm_set_override(int M, int N, int K, int conv, int stride, int up, int cfg, int splitk);   // in-eval tuner; M <= 0 clears all
int  gemm_plan_epoch();                       // bumped by every override change: launch plans re-build on the next mkd_prepare
int  gemm_num_tile_cfgs();
const char* gemm_tile_cfg_name(int cfg);
int  launch_splitk_epilogue(const GemmArgs& a, hipStream_t stream, const GemmArgs* second = nullptr);
bool conv_patch_supported(const GemmArgs& a, int cfg);
int  launch_conv_patch(GemmArgs a, int cfg, int splitk, hipStream_t stream, const GemmArgs* second = nullptr);
size_t gemm_ws_bytes(int M, int N, int splitk);

int launch_groupnorm(const bf16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu,
                     bf16_t* y, int ld_out, int batch, int hw, int C, int groups, float* partials,
                     hipStream_t stream, const NormIo* second = nullptr,       // second: same geometry, grouped launch
                     int two_kernel_min_hw = 1 << 30);      // >= this many pixels per sample: the two full-chip launches (mkd_ctx::gn_2k_min_hw)
size_t groupnorm_partials_bytes(int batch, int hw, int groups);
int launch_gn_stats(const bf16_t* x, int ld_in, int batch, int hw, int C, int groups, float* partials, hipStream_t stream, int* nchunks_out);
// GroupNorm with producer-emitted statistics (gstat[batch][32][2] int64 fixed point, see gemm_device.h): element-wise apply,
// and the stand-alone producer of the same statistics for tensors whose writer cannot emit them
int launch_gn_apply_stats(const bf16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu, bf16_t* y, int ld_out,
                          int batch, int hw, int C, const long long* gstat, hipStream_t stream);
// GroupNorm [+SiLU] of a split-K GEMM's output straight from its fp32 partial slabs (launched with defer_epilogue): one kernel
// does the slab reduction, the GEMM epilogue (bias, row bias, scale, residual, bf16 rounding) and the normalisation; the raw
// output is also stored when a.C is non-null.  a = the GEMM's arguments with ws / splitk as launch_gemm resolved them.
// Returns -4 when the geometry does not fit the single-pass kernel (the caller then runs the two kernels separately).
int launch_gn_from_slabs(const GemmArgs& a, const float* gamma, const float* beta, float eps, int silu, bf16_t* y, int ld_out,
                         int batch, int hw, hipStream_t stream, const GemmArgs* a2 = nullptr, const NormIo* second = nullptr);
bool gn_from_slabs_supported(int batch, int hw, int C);
int launch_gn_colstats(const bf16_t* x, int ld, int batch, int hw, int ncols, int cg, int coff, long long* gstat, hipStream_t stream);
int launch_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K,
                          bf16_t* w_out, int dst_row0, int dst_row_mul, float* s_out, float* b_out, hipStream_t stream);
int launch_layernorm(const bf16_t* x, const float* gamma, const float* beta, float eps, bf16_t* y,
                     int rows, int d, hipStream_t stream, int ldx = 0, const NormIo* second = nullptr);      // ldx: input row stride (0 = d); y is dense
int launch_merge_ff_out(const float* P, const float* W2, const float* b2, const float* bp, bf16_t* Wm, float* bias_m, int d,
                        hipStream_t stream);
// ---- fused row-local transformer tail (kernels_tfm.hip) ----------------------------------------------------------------------
// The tensors the engine's unfused path uses, all device pointers: attn1.to_out [d][d] + bias; attn2.to_q folded with LayerNorm 2
// (W' = W diag(gamma), s = rowsum(W'), b' = W beta); attn2.to_out [d][d] + bias; the GEGLU projection folded with LayerNorm 3 with
// (value, gate) rows interleaved [8d][d] (+ s, b' in the same order); the merged [ff.net.2 . proj_out | proj_out] weight [d][5d] + bias.
struct TfmTailWeights {
    const bf16_t* w_o1; const float* b_o1;
    const bf16_t* w_q; const float* s_q; const float* b_q;
    const bf16_t* w_o2; const float* b_o2;
    const bf16_t* w_g; const float* s_g; const float* b_g;
    const bf16_t* w_m; const float* b_m;
};
// ---- fused head of a d = 320 SpatialTransformer (kernels_tfm.hip): GroupNorm apply (statistics from launch_gn_stats) + proj_in +
// LayerNorm 1 folded into the q | k | v projection, one launch per 64-token tile: h0 [M, d] and qkv [M, 3d] out
struct TfmHeadWeights {
    const float* gn_gamma; const float* gn_beta;
    const bf16_t* w_pi; const float* b_pi;                       // proj_in [d][d]
    const bf16_t* w_qkv; const float* s_qkv; const float* b_qkv; // [3d][d] folded with LayerNorm 1 (W' = W diag(gamma), s, b')
};
size_t tfm_head_weight_bytes(int d);
size_t tfm_head_vec_bytes(int d);
int tfm_head_pack_weights(int d, const TfmHeadWeights& src, bf16_t* wpk, float* vec, hipStream_t stream);       // synchronous
int launch_tfm_head(int d, const bf16_t* wpk, const float* vec, const bf16_t* x, int ldx, const float* gn_partials, int gn_chunks, float gn_eps,
                    bf16_t* h0, bf16_t* qkv, int M, int T, hipStream_t stream);
void   tfm_tail_set_trace(long long* buf);
bool   tfm_tail_supported(int d, int heads, int T, int Tk);      // T tokens per sample, Tk context keys
size_t tfm_tail_weight_bytes(int d);
size_t tfm_tail_vec_bytes(int d);
size_t tfm_tail_kv_bytes(int d, int batch);
double tfm_tail_flops(int d, int M, int Tk);
int tfm_tail_pack_weights(int d, const TfmTailWeights& src, bf16_t* wpk, float* vec, hipStream_t stream);       // synchronous
int launch_tfm_tail_pack_kv(int d, const bf16_t* kv, int ldkv, int batch, int Tk, bf16_t* out, hipStream_t stream);
int launch_tfm_tail(int d, const bf16_t* wpk, const float* vec, const bf16_t* a1, int lda, const bf16_t* h0, int ldh, const bf16_t* xin, int ldx,
                    const bf16_t* kvp, bf16_t* out, int ldo, int M, int T, int Tk, hipStream_t stream);
int launch_attention(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv,
                     bf16_t* o, int ldo, int batch, int Tq, int Tk, int heads, int dh, float scale,
                     hipStream_t stream, int causal = 0, const AttnIo* second = nullptr);
int attn_set_trace(long long* buf);      // -DMKD_ATTN_TRACE builds: per-phase cycle sums of the attention kernel
int launch_geglu(const bf16_t* x, bf16_t* y, int rows, int inner, hipStream_t stream);
int launch_conv3x3_direct(const void* x, int in_nchw_f32, const bf16_t* w, const float* bias, void* y,
                          int out_nchw_f32, int act, const bf16_t* add, int batch, int Hin, int Win,
                          int Cin, int Cout, int stride, hipStream_t stream, const ConvInIo* second = nullptr);   // second: 4 -> C form only
int launch_pack_conv_weight(const float* w, bf16_t* out, int Cout, int Cin, int kh, int kw, hipStream_t stream);
int launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t stream);
int launch_timestep_embedding(const int64_t* t, bf16_t* out, int batch, int dim, hipStream_t stream);
int launch_copy_strided(const bf16_t* src, int ld_src, bf16_t* dst, int ld_dst, int rows, int cols,
                        hipStream_t stream);
int launch_ddim_step(const float* x, const float* eps_c, const float* eps_u, float cfg_scale, float a_t,
                     float a_prev, float sigma_t, float s1m, const float* noise, float temperature,
                     float* x_prev, float* pred_x0, int64_t n, hipStream_t stream);
int launch_repeat_batch(const float* x, float* y, int64_t n_per, int reps, hipStream_t stream);
int launch_fill_i64(int64_t* p, int64_t v, int n, hipStream_t stream);
// first kernel of a replayed step: timestep / coefficients of step st->counter (+ the step's time-embedding rows when ts != null).
// The counter itself is advanced by the step's LAST kernel (launch_ddim_step_state), so every workgroup here reads the same value.
int launch_step_setup(StepState* st, int64_t* t_out, int batch, hipStream_t stream, const TembSel* ts = nullptr);
int launch_temb_select(const TembSel& ts, int step, hipStream_t stream);          // the same copy with a host-side step index (eager loop)
int launch_ddim_step_state(float* x, const float* eps_c, const float* eps_u, float cfg_scale, StepState* st, int64_t n,
                           hipStream_t stream);
int launch_softmax_rows(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t stream);
int launch_post_quant(const float* z, const bf16_t* w, const float* bias, float inv_scale, float* out, int batch, int C, int hw,
                      hipStream_t stream);
int launch_clip_embed(const int32_t* tokens, const bf16_t* tok_emb, const bf16_t* pos_emb, bf16_t* out, int batch, int T, int width,
                      int vocab, hipStream_t stream);
int launch_bf16_to_f32(const bf16_t* x, float* y, int64_t n, hipStream_t stream);
int launch_blend(const bf16_t* a, const bf16_t* b, const float* alpha, bf16_t* y, int64_t per_sample, int batch, hipStream_t stream);
