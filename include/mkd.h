/*
 * mkd.h — C ABI of libmkd.so: the MI355X (gfx950) DDIM-sampling hot path of MakeupDiffuse.
 *
 * The reference (jiean001/MakeupDiffuse) has NO C/FFI boundary: its hot path is Python
 * duck-typing over torch tensors.  Each entry point below names the
 * reference interface (file:line under /root/reference) whose arithmetic it replaces; the
 * Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - functions only ENQUEUE work on `stream` (no host sync) unless stated otherwise;
 *   - return 0 on success, <0 on error; mkd_last_error() gives the message (thread-local);
 *   - external layout is the reference's: NCHW fp32 latents/images, [B,77,C] fp32 context,
 *     int64 timesteps.  Internally activations are NHWC bf16, accumulation fp32.
 *   - no internal host threads; a context is not re-entrant (one stream at a time).
 *   - host synchronisation points (everything else only enqueues): mkd_ctx_create / mkd_weights_finalize / mkd_vae_finalize / mkd_clip_finalize
 *     and the first mkd_prepare of a new shape (plan building, hipDeviceSynchronize); mkd_sample(use_graph != 0), which waits on the host
 *     for the context's PREVIOUS graph-replayed loop before it rewrites the pinned step table (hipStreamSynchronize of the
 *     private loop stream) and then returns with the new loop enqueued; mkd_eps_profile (measures, so it waits); mkd_ctx_destroy.
 */
#ifndef MKD_H
#define MKD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mkd_ctx mkd_ctx;

/* yaml control_stage_config / unet_config (diffmodels/base_diffusion_makeup.yaml:52-84). */
typedef struct mkd_net_config {
    int32_t in_channels;            /* 4   */
    int32_t out_channels;           /* 4   */
    int32_t hint_channels;          /* 6   (src_img ‖ ref_img, makeup_diffuse.py:56) */
    int32_t model_channels;         /* 320 */
    int32_t num_res_blocks;         /* 2   */
    int32_t n_levels;               /* len(channel_mult) = 4 */
    int32_t channel_mult[8];        /* 1,2,4,4 */
    int32_t n_attention_resolutions;/* 3 */
    int32_t attention_resolutions[8];/* 4,2,1 */
    int32_t num_heads;              /* 8   */
    int32_t transformer_depth;      /* 1 (only 1 is supported) */
    int32_t context_dim;            /* 768 */
    int32_t hint_widths[7];         /* 16,16,32,32,96,96,256 (cldm input_hint_block) */
} mkd_net_config;

#define MKD_OK              0
#define MKD_ERR_ARG        -1
#define MKD_ERR_HIP        -2
#define MKD_ERR_STATE      -3
#define MKD_ERR_UNSUPPORTED -4
#define MKD_ERR_MISSING    -5

const char* mkd_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int mkd_abi_version(void);
/* 1 when the library was built with 2-entry kernel argument tables (-DMKD_PAIR_N=2): the grouped encoder chain experiment
 * (MKD_ENC_GROUP=1) is available; the default build has single-entry tables (1.9 % faster on the default plan) and refuses it. */
int mkd_grouped_launches_available(void);

/* ---- context ---------------------------------------------------------------------------- */
/* Replaces cldm.model.create_model(yaml) for the two nets (runs/test.py:27). */
int  mkd_ctx_create(const mkd_net_config* cfg, mkd_ctx** out);
void mkd_ctx_destroy(mkd_ctx* ctx);

/* Replaces model.load_state_dict (runs/test.py:59-60) for keys under
 * "model.diffusion_model." and "control_model." (upstream names).
 * `data` is fp32, host OR device, `shape` is host. Synchronous. Unknown names -> MKD_ERR_ARG.
 * Loading a net weight invalidates the prepared conditioning: mkd_eps / mkd_sample fail (MKD_ERR_STATE) until mkd_weights_finalize +
 * mkd_prepare ran again; the weight forms mkd_weights_finalize derives are rebuilt and their previous generation is freed. */
int mkd_load_weight(mkd_ctx* ctx, const char* name, const float* data, int ndim, const int64_t* shape);
/* Checks every expected tensor was loaded, builds fused/packed weights. Synchronous. */
int mkd_weights_finalize(mkd_ctx* ctx);
/* Number of parameters expected (for the 859.5 M / 361.3 M check); which: 0 unet, 1 control. */
int64_t mkd_param_count(const mkd_ctx* ctx, int which);
/* Enumerate the expected state_dict entries (sorted by name): count, name, shape (returns ndim <= 4). */
int mkd_param_total(const mkd_ctx* ctx);
const char* mkd_param_name(const mkd_ctx* ctx, int index);
int mkd_param_shape(const mkd_ctx* ctx, int index, int64_t* shape4);

/* Per-context plan switches (round 4: what used to be read from MKD_* environment variables once per process; the variables still
 * give the initial values).  Takes effect at the next mkd_prepare (the launch plan is re-built).  Names:
 *   "tfm_tail"             fused row-local transformer tail: 0 off, 1 wherever the kernel covers the shape, -1 shape policy (default)
 *   "tfm_tail_min_rows"    ... the policy's threshold on the rows (samples x tokens) of a block (default 4096)
 *   "skip_fold"            ... ResBlocks with a 1x1 skip_connection: conv2 and the skip as one implicit GEMM (K = 9 Cout + Cin) where
 *                          conv2's plan is the gather kernel: 0 off, 1 on (default)
 *   "tfm_head"             ... and the block's head (GroupNorm apply + proj_in + LayerNorm 1 . q|k|v) as one launch behind a GroupNorm
 *                          statistics launch wherever the tail is fused: 0 off, 1 on (default)
 *   "gn_2k_min_hw"         GroupNorm over >= this many pixels per sample: two full-chip launches (default 4096)
 *   "xcd_auto_ratio"       XCD-aware tile order where M <= ratio x N (default 1; 0 = launch order everywhere)
 *   "dec_lanes"            decoder batch lanes 0 / 2 / 4 (default 2)
 *   "ln_fly"               bit mask: LayerNorm taken on the fly by 1 = q|k|v, 2 = attn2.to_q, 4 = GEGLU projection (default 2)
 *   "gn_slab_min_channels" slab-fed GroupNorm from this many channels (default 1280)
 *   "graph_steps"          DDIM steps per captured graph (default 5)
 * Unknown names return MKD_ERR_ARG.  The tile tuner's state (mkd_gemm_force_tile / _set_xcd_mode / _set_override) stays process-global
 * by design (single-kernel entries, tuners): a change makes EVERY live context re-plan at its next mkd_prepare; mkd_live_contexts()
 * says how many there are. */
int mkd_ctx_set_option(mkd_ctx* ctx, const char* name, double value);
int mkd_ctx_get_option(const mkd_ctx* ctx, const char* name, double* value);
int mkd_live_contexts(void);

/* ---- conditioning ----------------------------------------------------------------------- */
/* Binds the step-invariant conditioning for a batch (cond dict of makeup_diffuse.py:42-57,
 * 152-166): hint = cat(c_concat,1) [B,hint_channels,8h,8w] in [0,1]; context = cat(c_crossattn,1)
 * [B,77,context_dim]; control_scales host [13] (makeup_diffuse.py:166) or NULL for all-ones;
 * only_mid_control (makeup_diffuse.py:162,168).  hint == NULL selects the `c_concat is None`
 * branch (makeup_diffuse.py:160-162).  Computes and caches the ControlNet hint embedding and
 * every cross-attention K/V projection (both independent of x and t).  (Re)allocates the
 * workspace when batch/h/w change (the only place that allocates). */
int mkd_prepare(mkd_ctx* ctx, int batch, int h, int w, const float* hint, const float* context,
                const float* control_scales, int only_mid_control, void* stream);

/* Makeup INTERPOLATION between two references (BUILD-DEFINED: the reference only shows a figure, README.md:23-25; SURVEY.md
 * §8f rank 2): like mkd_prepare, but the cached ControlNet hint embedding is the per-sample blend
 * (1 - alpha[b]) * E(hint_a[b]) + alpha[b] * E(hint_b[b]); hint_a = src||ref1, hint_b = src||ref2, alpha device fp32 [B]. */
int mkd_prepare_interp(mkd_ctx* ctx, int batch, int h, int w, const float* hint_a, const float* hint_b, const float* alpha,
                       const float* context, const float* control_scales, int only_mid_control, void* stream);

/* ---- one eps evaluation ----------------------------------------------------------------- */
/* Replaces apply_model (makeup_diffuse.py:152-170): ControlNet -> 13 residuals x scale ->
 * ControlledUnet.  x [B,4,h,w] fp32 NCHW, t [B] int64 (device), eps_out [B,4,h,w] fp32. */
int mkd_eps(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream);

/* ---- DDIM update ------------------------------------------------------------------------ */
/* Replaces cddim.py:39-40 (CFG combine, eps_u may be NULL) and :56-78 (x0 / x_{t-1}).
 * All tensors have n elements; noise may be NULL (sigma_t == 0); pred_x0 may be NULL. */
int mkd_ddim_step(const float* x, const float* eps_c, const float* eps_u, float cfg_scale,
                  float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at,
                  const float* noise, float temperature,
                  float* x_prev, float* pred_x0, int64_t n, void* stream);

/* ---- whole reverse loop ------------------------------------------------------------------ */
/* Replaces MKDDIMSampler.reconstruct (cddim.py:81-100) / DDIMSampler.ddim_sampling reached from
 * sample_log (diffusion_makeup.py:393-408) for eta == 0.  The context must have been prepared with
 * batch B (cfg_scale == 1) or 2B with the UNCONDITIONAL conditioning first (cddim.py:25-31).
 * Tables are host arrays of length n_steps indexed like ddim_alphas[index]; the loop runs
 * index = n_steps-1 .. 0 with timestep = timesteps[index].  x_T, x_out: [B,4,h,w] fp32.
 * use_graph != 0 captures one step into a hipGraph and replays it; such a call first waits (host) until the previous
 * graph-replayed loop of this context has finished, see "host synchronisation points" above. */
int mkd_sample(mkd_ctx* ctx, const float* x_T, int batch, int n_steps, const int64_t* timesteps,
               const float* alphas, const float* alphas_prev, const float* sqrt_one_minus_alphas,
               float cfg_scale, float* x_out, int use_graph, void* stream);

/* The same loop with eta > 0 (cddim.py:56-78 / UPSTREAM DDIMSampler.p_sample_ddim: dir_xt = sqrt(1 - a_prev - sigma_t^2) e_t,
 * x_prev += sigma_t * noise * temperature).  sigmas: host array like the other tables (ddim_sigmas[index]); noise: DEVICE array
 * [n_steps][B*4*h*w] fp32, row k = the draw of the k-th executed step (the caller draws them in loop order, as the reference's
 * noise_like does step by step); both NULL, or every sigma 0: mkd_sample.  The graph replays unchanged (the step reads its
 * sigma / noise row from the device-resident step state).  `noise` is read by the enqueued loop: it must stay valid until the work
 * on `stream` has completed. */
int mkd_sample_eta(mkd_ctx* ctx, const float* x_T, int batch, int n_steps, const int64_t* timesteps,
                   const float* alphas, const float* alphas_prev, const float* sqrt_one_minus_alphas,
                   const float* sigmas, const float* noise, float temperature,
                   float cfg_scale, float* x_out, int use_graph, void* stream);

/* ---- first-stage decoder ------------------------------------------------ */
/* yaml first_stage_config.params.ddconfig (diffmodels/base_diffusion_makeup.yaml:86-107), decoder half only. */
typedef struct mkd_vae_config {
    int32_t z_channels;      /* 4 */
    int32_t embed_dim;       /* 4 */
    int32_t ch;              /* 128 */
    int32_t n_levels;        /* 4 */
    int32_t ch_mult[8];      /* 1,2,4,4 */
    int32_t num_res_blocks;  /* 2 */
    int32_t out_ch;          /* 3 */
} mkd_vae_config;
/* Adds the "first_stage_model.post_quant_conv.*" / "first_stage_model.decoder.*" entries to the expected state_dict
 * (load them with mkd_load_weight).  Optional: the sampler works without it. */
int mkd_vae_configure(mkd_ctx* ctx, const mkd_vae_config* cfg);
int mkd_vae_finalize(mkd_ctx* ctx);
/* Replaces decode_first_stage (diffmk/diffusion_makeup.py:396,409; diffmk/makeups.py:260-262): z / scale_factor ->
 * post_quant_conv -> Decoder.  z [B,4,h,w] fp32 NCHW -> images [B,3,8h,8w] fp32 NCHW (unclamped, nominally [-1,1]). */
int mkd_decode(mkd_ctx* ctx, const float* z, int batch, int h, int w, float scale_factor, float* images, void* stream);
double mkd_decode_flops(const mkd_ctx* ctx);

/* ---- CLIP text encoder -------------------------------------------------- */
/* yaml cond_stage_config FrozenCLIPEmbedder (diffmodels/base_diffusion_makeup.yaml:109-110), i.e. UPSTREAM transformers
 * CLIPTextModel: token + position embeddings, `layers` pre-LN blocks (causal self-attention, quick-GELU MLP), final LN. */
typedef struct mkd_clip_config {
    int32_t vocab_size;      /* 49408 */
    int32_t max_positions;   /* 77 */
    int32_t width;           /* 768 */
    int32_t layers;          /* 12 */
    int32_t heads;           /* 12 */
    int32_t intermediate;    /* 3072 */
    float   ln_eps;          /* 1e-5 */
} mkd_clip_config;
/* Adds the "cond_stage_model.transformer.text_model.*" entries to the expected state_dict.  Optional. */
int mkd_clip_configure(mkd_ctx* ctx, const mkd_clip_config* cfg);
int mkd_clip_finalize(mkd_ctx* ctx);
/* Replaces get_learned_conditioning / FrozenCLIPEmbedder.encode after tokenisation (diffmk/makeup_teacher.py:33-42,
 * diffmk/diffusion_makeup.py:400): tokens [B, n_tokens] int32 (device; padded ids included, CLIP applies only the causal
 * mask) -> last_hidden_state [B, n_tokens, width] fp32 (device). */
int mkd_clip_encode(mkd_ctx* ctx, const int32_t* tokens, int batch, int n_tokens, float* out, void* stream);

/* ---- introspection for bench.py ----------------------------------------------------------- */
/* Executed matmul/conv FLOPs (2 per MAC) of one mkd_eps at the prepared shape. */
double  mkd_eps_flops(const mkd_ctx* ctx);
/* Number of kernel launches of one mkd_eps at the prepared shape. */
int     mkd_eps_launches(const mkd_ctx* ctx);
/* Number of kernel launches of one DDIM step inside mkd_sample at the prepared shape (the time-embedding chain of mkd_eps is
 * computed once per call there, see mkd_sample; + the step setup and the x_{t-1} update). */
int     mkd_step_launches(const mkd_ctx* ctx);                           /* graph replay, no guidance */
int     mkd_step_launches_ex(const mkd_ctx* ctx, int use_graph, int cfg_on); /* as the loop is run: eager adds the timestep fill / table-row select, guidance the batch doubling */
/* Kernel classes of the launch plan, and one mkd_eps with a hipEvent pair around every launch group:
 * per-class device milliseconds, executed FLOPs and launch counts (arrays of mkd_kind_count()). Synchronous.
 * csv_path (host string, may be NULL): also write one line per launch group (op,kind,label,ms,gflop). */
int mkd_kind_count(void);
const char* mkd_kind_name(int kind);
int mkd_eps_profile(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream,
                    double* ms_per_kind, double* flops_per_kind, int* launches_per_kind, const char* csv_path);
/* The same, plus per class: the ALGORITHMIC HBM bytes of the memory-bound launches (GroupNorm / LayerNorm: one read + one write of
 * the tensor; slab-fed GroupNorm: its fp32 slabs in, bf16 out) and the class's launches replayed back to back between ONE event pair
 * (per-launch time without the event overhead of the per-launch pass).  Either array may be NULL. */
int mkd_eps_profile2(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream,
                     double* ms_per_kind, double* flops_per_kind, int* launches_per_kind, double* bytes_per_kind,
                     double* ms_back_to_back_per_kind, const char* csv_path);
/* Bytes of device memory held by the context (weights + workspace). */
int64_t mkd_device_bytes(const mkd_ctx* ctx);

/* ResBlock tail as ONE implicit GEMM (UPSTREAM ResBlock._forward: out = conv3x3(h) + skip_connection(x), the 1x1 skip of the blocks whose
 * channel count changes; reached from diffmk/makeup_diffuse.py:164-168): y[m, :] = conv3x3(x)[m, :] + x2[m, :] . W_skip^T + bias.
 * w_fold: [N][9 * Cin + K2] bf16, row n = (the packed 3x3 weight row of mkd_pack_conv_weight, tap-major) | W_skip[n, :]; x [batch*H*W, Cin]
 * and x2 [batch*H*W, K2] NHWC at ldx / ldx2; stride 1, pad 1.  Runs on the gather kernel (tile plan of the plain convolution);
 * MKD_ERR_UNSUPPORTED when that shape's plan is an LDS-staged tile. */
int mkd_conv3x3_fold_bf16(const uint16_t* x, int ldx, const uint16_t* w_fold, const float* bias, const uint16_t* x2, int ldx2, int K2,
                          uint16_t* y, int ldy, int batch, int H, int W, int Cin, int N, int splitk, void* stream);

/* ---- single kernels (unit parity tests; bf16 = uint16_t device buffers) -------------------- */
/* C[M,N] = act((A[M,K] . W[N,K]^T + bias[N] + rowbias[m / rows_per_batch][n]) * scale + R[M,N]).
 * conv3x3 != 0: A is NHWC [B,Hin,Win,Cin] (pixel stride lda), K = 9*Cin ordered (ky,kx,ci),
 * output pixel grid Hout x Wout with `stride`, input nearest-upsampled by 2^up first; pad 1.
 * out_f32 selects an fp32 C.  splitk 0 = auto. */
int mkd_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                  const float* rowbias, int ldrb, int rows_per_batch,
                  const uint16_t* R, int ldr, float scale, int act,
                  void* C, int ldc, int out_f32, int M, int N, int K,
                  int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout,
                  int stride, int up, int splitk, void* stream);
/* LayerNorm fused into a linear GEMM: C = act(LN(A) . W^T + b) computed as rstd*(A.W'^T - mu*s) + b' on the RAW rows of A.
 * mkd_fold_layernorm builds W' = bf16(W*gamma) (written to rows dst_row0 + n*dst_row_mul of w_out), s = rowsum(W'),
 * b' = bias + W.beta from fp32 W [N,K] (device).  mkd_gemm_ln_bf16 runs the fused GEMM (no split-K).  row_stats == NULL (the
 * form the engine uses): the GEMM takes the row statistics itself from the A fragments it holds (ones . A^T and the diagonal of
 * A . A^T on the matrix cores) - any A, no producer involved.  Otherwise the row sums come from the GEMM that produced A:
 * row_stats [stat_slots][M][2] = partial (sum, sum of squares) per column slot.
 * mkd_gemm_rowstats_bf16 is that producer: C = A.W^T + bias + R (bf16) and the partial row sums of the rounded C. */
int mkd_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K,
                       uint16_t* w_out, int dst_row0, int dst_row_mul, float* s_out, float* b_out, void* stream);
int mkd_gemm_ln_bf16(const uint16_t* A, int lda, const uint16_t* Wfold, int ldw, const float* bias_fold, const float* ln_s,
                     const float* row_stats, int stat_slots, float eps, int act, void* C, int ldc, int M, int N, int K,
                     void* stream);
int mkd_gemm_rowstats_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* R, int ldr,
                           uint16_t* C, int ldc, int M, int N, int K, float* stat_out, int stat_capacity_slots, int* slots_out,
                           void* stream);
/* Tuner / tests only: force the GEMM tile configuration (index into the table in kernels_gemm.hip; -1 = heuristic). */
int mkd_gemm_force_tile(int cfg);
/* Tests / experiments: workgroup -> tile order of the GEMM and LDS-staged conv kernels with respect to the 8 XCDs (each has its own
 * L2; workgroups are dealt to them round-robin in launch order).  0 (default): launch order; 1: every XCD gets one contiguous run
 * of the tile sequence with M-tiles fastest (a weight tile lives in one L2); 2: the same with N-tiles fastest.  Results do not
 * depend on it (bit-identical).  Applies to launches issued afterwards. */
int mkd_gemm_set_xcd_mode(int mode);
/* Tests only (race detector): overwrite every buffer one mkd_eps produces (activations, temporaries, workspaces) with NaN
 * patterns, so that a kernel running ahead of its producer cannot see the previous call's values.  Synchronous. */
int mkd_debug_poison(mkd_ctx* ctx);
/* In-eval tuner (tools/tune_ineval.py): per-shape (tile config, split-K) override consulted before the compiled table;
 * cfg < 0 removes one entry, M <= 0 clears all.  Takes effect at the next mkd_prepare (plans re-build). */
int mkd_gemm_set_override(int M, int N, int K, int conv3x3, int stride, int up, int cfg, int splitk);
/* 1 when tile configuration `cfg` can run this shape (the LDS-staged conv tiles have geometry limits). */
int mkd_gemm_cfg_supported(int cfg, int M, int N, int K, int conv3x3, int Hin, int Win, int Cin, int Hout, int Wout,
                           int stride, int up);
/* GroupNorm(32 groups, fp32 statistics) [+SiLU] over NHWC bf16 (pixel stride ld_in). */
int mkd_groupnorm(const uint16_t* x, int ld_in, const float* gamma, const float* beta, float eps,
                  int silu, uint16_t* y, int ld_out, int batch, int hw, int C, int groups, void* stream);
/* GroupNorm split in two so that the statistics pass disappears into the kernel that WRITES the tensor (UPSTREAM GroupNorm32 of
 * ResBlock.in_layers/out_layers and SpatialTransformer.norm, reached from diffmk/makeup_diffuse.py:164-168):
 *   gstat[batch][32][2] = (sum * 2^24, sum of squares * 2^18) as 64-bit integers, zeroed by the caller, accumulated with integer
 *   atomics (order-independent, so results are bit-repeatable) by every producer of the tensor;
 *   mkd_gemm_gnstat_bf16  = mkd_gemm_bf16 whose epilogue also adds the statistics of its bf16 output (columns gn_coff.. of a
 *                           consumer tensor with gn_cg channels per group, gn_hw rows per sample);
 *   mkd_gn_colstats       = stand-alone producer of the same statistics for columns [0, ncols) of x;
 *   mkd_gn_apply_stats    = y = (x - mean) * rstd * gamma + beta [-> SiLU], element-wise, statistics read from gstat. */
int mkd_gemm_gnstat_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                         const float* rowbias, int ldrb, int rows_per_batch,
                         const uint16_t* R, int ldr, float scale, int act, void* C, int ldc, int out_f32, int M, int N, int K,
                         int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout,
                         int stride, int upsample, int splitk, int64_t* gn_stat, int gn_cg, int gn_coff, int gn_hw, void* stream);
int mkd_gn_colstats(const uint16_t* x, int ld, int batch, int hw, int ncols, int cg, int coff, int64_t* gstat, void* stream);
int mkd_gn_apply_stats(const uint16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu, uint16_t* y,
                       int ld_out, int batch, int hw, int C, const int64_t* gstat, void* stream);
/* A split-K GEMM / conv3x3 whose output feeds a GroupNorm(32) [+SiLU]: the GEMM leaves its fp32 partial slabs and ONE kernel does
 * slab reduction + GEMM epilogue (bias, per-sample row bias, scale, residual, bf16 rounding) + GroupNorm -> y; the raw GEMM output
 * is also written to C when write_raw != 0 (C must then be valid).  Same arguments as mkd_gemm_bf16 (act must be 0, bf16 output,
 * rows_per_batch = rows per sample when rowbias is given).  Returns -4 when this shape would not be split over K (splitk = 0 picks
 * the tuned value) or the GroupNorm geometry does not fit the single-pass kernel: call mkd_gemm_bf16 + mkd_groupnorm then. */
int mkd_gemm_groupnorm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                            const float* rowbias, int ldrb, int rows_per_batch,
                            const uint16_t* R, int ldr, float scale, void* C, int ldc, int write_raw, int M, int N, int K,
                            int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout,
                            int stride, int upsample, int splitk, int rows_per_sample,
                            const float* gamma, const float* beta, float eps, int silu, uint16_t* y, int ld_y, void* stream);
/* ---- fused row-local tail of a SpatialTransformer block (kernels_tfm.hip), stand-alone form ---------------------------------
 * UPSTREAM cldm BasicTransformerBlock after the self-attention product, reached from diffmk/makeup_diffuse.py:164-168:
 *   h1 = attn1.to_out(a1) + h0;  h2 = attn2.to_out(softmax(attn2.to_q(LN2 h1) K^T dh^-0.5) V) + h1;
 *   out = proj_out(ff.net.2(GEGLU(ff.net.0.proj(LN3 h2))) + h2) + x_in
 * as ONE kernel per 64-token row tile (d = 320, 8 heads).  mkd_tfm_tail_create takes the block's fp32 DEVICE weights under their
 * upstream shapes ([d,d], [8d,d], [d,4d], vectors) and builds the packed operand-order copies; mkd_tfm_tail_set_context packs the
 * cross-attention K | V projections of the context (kv: [batch * Tk, 2d] bf16, K in columns [0,d), V in [d,2d); Tk <= 80);
 * mkd_tfm_tail_run: a1, h0, x_in, out are [M, d] bf16 with row strides, M = samples * T, T a multiple of 64.  The engine uses the
 * same kernel inside mkd_eps for its d = 320 blocks (mkd_set_option "tfm_tail"). */
typedef struct mkd_tfm_tail mkd_tfm_tail;
int  mkd_tfm_tail_create(int d, const float* to_out1_w, const float* to_out1_b, const float* norm2_g, const float* norm2_b,
                         const float* to_q2_w, const float* to_out2_w, const float* to_out2_b, const float* norm3_g, const float* norm3_b,
                         const float* ff0_w, const float* ff0_b, const float* ff2_w, const float* ff2_b, const float* proj_out_w,
                         const float* proj_out_b, mkd_tfm_tail** out);
void mkd_tfm_tail_destroy(mkd_tfm_tail* h);
/* Experiment builds (-DMKD_TFM_TRACE) only: device buffer [workgroups][8][32] of int64 time stamps; a no-op in the product build. */
int  mkd_debug_tfm_trace(long long* buf);
int  mkd_debug_attn_trace(long long* buf);      /* -DMKD_ATTN_TRACE builds: [workgroup][wave][8] cycle sums per phase of attention_kernel */
int  mkd_tfm_tail_set_context(mkd_tfm_tail* h, const uint16_t* kv, int ldkv, int batch, int Tk, void* stream);
int  mkd_tfm_tail_run(mkd_tfm_tail* h, const uint16_t* a1, int lda, const uint16_t* h0, int ldh, const uint16_t* xin, int ldx,
                      uint16_t* out, int ldo, int M, int T, void* stream);
/* The head of the same block as ONE kernel behind a GroupNorm statistics launch: GroupNorm(32, eps) apply -> proj_in -> LayerNorm 1
 * folded into to_q | to_k | to_v: x [B * T, d] (row stride ldx, T a multiple of 64) -> h0 [B * T, d] and qkv [B * T, 3d] (dense).
 * fp32 DEVICE weights under their upstream shapes (SpatialTransformer.norm, .proj_in; transformer_blocks.0.norm1, attn1.to_q/k/v). */
typedef struct mkd_tfm_head mkd_tfm_head;
int  mkd_tfm_head_create(int d, const float* gn_g, const float* gn_b, const float* proj_in_w, const float* proj_in_b, const float* norm1_g,
                         const float* norm1_b, const float* to_q_w, const float* to_k_w, const float* to_v_w, mkd_tfm_head** out);
void mkd_tfm_head_destroy(mkd_tfm_head* h);
int  mkd_tfm_head_run(mkd_tfm_head* h, const uint16_t* x, int ldx, float gn_eps, uint16_t* h0, uint16_t* qkv, int batch, int T, void* stream);
/* LayerNorm over the last dim of [rows, d] bf16. */
int mkd_layernorm(const uint16_t* x, const float* gamma, const float* beta, float eps,
                  uint16_t* y, int rows, int d, void* stream);
/* ... with a row stride on x (the engine normalises the h2 columns of its [gg | h2] buffer); y is dense. */
int mkd_layernorm_ld(const uint16_t* x, int ldx, const float* gamma, const float* beta, float eps,
                     uint16_t* y, int rows, int d, void* stream);
/* softmax(q k^T * scale) v per (batch, head); q rows b*Tq+i, k/v rows b*Tk+j, head h at column h*dh. */
int mkd_attention(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv,
                  uint16_t* o, int ldo, int batch, int Tq, int Tk, int heads, int dh, float scale,
                  void* stream);
/* same with the causal mask of the CLIP text encoder: query i attends keys 0..i (Tq == Tk). */
int mkd_attention_causal(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv,
                         uint16_t* o, int ldo, int batch, int Tq, int Tk, int heads, int dh, float scale,
                         void* stream);
/* y[m, j] = x[m, j] * gelu_erf(x[m, inner + j]). */
int mkd_geglu(const uint16_t* x, uint16_t* y, int rows, int inner, void* stream);
/* direct 3x3 conv, pad 1, fp32 accumulate.  in_nchw_f32: x is fp32 NCHW else bf16 NHWC;
 * out_nchw_f32 likewise; act 1 = SiLU; add (bf16 NHWC, may be NULL) is added after act. */
int mkd_conv3x3_direct(const void* x, int in_nchw_f32, const uint16_t* w, const float* bias,
                       void* y, int out_nchw_f32, int act, const uint16_t* add,
                       int batch, int Hin, int Win, int Cin, int Cout, int stride, void* stream);
/* fp32 [Cout,Cin,kh,kw] -> bf16 [Cout][kh][kw][Cin]. */
int mkd_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int kh, int kw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MKD_H */
