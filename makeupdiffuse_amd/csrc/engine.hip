// libmkd engine: owns the weights, the workspace and the static launch plan of one eps evaluation
// (ControlNet -> 13 scaled residuals -> ControlledUnet; reference diffmk/makeup_diffuse.py:152-170),
// plus the DDIM reverse loop (reference diffmk/cddim.py:81-100).  One process per GPU, one stream.
//
// Data layout in HBM
//   activations : NHWC bf16, pixel stride `ld` (>= C) so channel-concats are just strided writes:
//                 every decoder block reads one [B,H,W,C_h+C_skip] buffer whose two halves were
//                 written in place by the previous block and by the zero-conv "combine" GEMM.
//   weights     : bf16, conv [Cout][ky][kx][Cin], linear [N][K]; vectors (bias, norm) fp32.
//   step-invariant caches (built in mkd_prepare): hint embedding, cross-attention K/V of every
//                 transformer (context is constant over steps, SURVEY.md finding 5).
#include "mkd_common.h"
#include "../../include/mkd.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace {

struct Tensor {
    bf16_t* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0, ld = 0;
    // GroupNorm statistics of this tensor, accumulated by the kernels that WRITE it (gemm_device.h): [B][32][2] int64, or null
    // when the GroupNorm that reads it computes them itself
    long long* gst = nullptr;
    int rows() const { return B * H * W; }
};

// where a producer adds the GroupNorm statistics of what it writes: buffer of the consumer tensor, its channels per group, the
// column this output starts at inside it (concat halves), rows per sample
struct GnOut {
    long long* gst = nullptr; int cg = 0, coff = 0, hw = 0;
};
static GnOut gn_of(const Tensor& t, int coff = 0) {
    GnOut g;
    if (t.gst) { g.gst = t.gst; g.cg = t.C / 32; g.coff = coff; g.hw = t.H * t.W; }
    return g;
}

struct BlockSpec {
    int kind;  // 0 conv_in, 1 res, 2 down
    int cin, cout;
    bool attn, up;
    int ds;
};

struct Param {
    std::vector<int64_t> shape;
    int which = 0;          // 0 unet, 1 control
    bool loaded = false;
    void* dev = nullptr;    // bf16 packed (ndim >= 2) or f32 (ndim == 1)
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct Arena {
    char* base = nullptr;
    size_t off = 0, high = 0, cap = 0;
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base + off;
        off += bytes;
        if (off > high) high = off;
        return p;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    void reset() { off = 0; high = 0; }
};

struct Epi {
    const float* bias = nullptr;
    const float* rowbias = nullptr; int ldrb = 0; int rpb = 1;
    const bf16_t* R = nullptr; int ldr = 0;
    float scale = 1.f; int act = 0;
    const float* ln_s = nullptr;          // fused LayerNorm of the A rows (weights pre-folded with gamma/beta)
    const float* stat_in = nullptr; int stat_slots = 0;   // ... with the row sums the producer GEMM emitted
    float* stat_out = nullptr;            // emit row sums of this GEMM's output for a downstream fused LayerNorm
    GnOut gn;                             // emit GroupNorm statistics of this GEMM's output
};

typedef std::function<int(hipStream_t)> OpFn;

// What a plan op launches, kept beside its closure for the ops that have a GROUPED form (mkd_common.h: Pair): when the ControlNet
// and the UNet encoder emit the same op on the same geometry, the pair becomes one launch (mkd_ctx::group_ops).
enum DescType { D_NONE = 0, D_GEMM, D_GN, D_GN_SLAB, D_LN, D_ATTN, D_CONV_IN };
struct OpDesc {
    int type = D_NONE;
    int sid = 0;                 // workspace set of the op (split-K slabs, GroupNorm partials)
    GemmArgs gemm;               // D_GEMM; D_GN_SLAB: the split GEMM whose slabs are reduced
    NormIo nio{};                // D_GN, D_GN_SLAB (x unused), D_LN
    float eps = 0.f; int silu = 0, ld_in = 0, ld_out = 0, nb = 0, hw = 0, C = 0;      // D_LN: nb = rows, C = d, ld_in = row stride
    int raw = 0, sk = 0;         // D_GN_SLAB: also store the raw GEMM output; slab count
    AttnIo aio{}; int ldq = 0, ldk = 0, ldv = 0, ldo = 0, Tq = 0, Tk = 0, heads = 0, dh = 0;      // D_ATTN (nb = samples)
    ConvInIo cio{}; size_t xoff = 0; int cin = 0, cout = 0, hh = 0, ww = 0;                      // D_CONV_IN (nb = samples)
    OpDesc() { memset(&gemm, 0, sizeof(gemm)); }
};

struct Op {
    OpFn fn;
    OpDesc d;
    bool temb = false;      // part of the time-embedding chain (left out of the step when mkd_sample runs from its table)
    int kind = 0;
    double flops = 0;
    double bytes = 0;       // algorithmic HBM bytes of the memory-bound ops (norms, split-K reduce inside slab-fed GroupNorm): what MUST move
    int launches = 0;
    std::string label;
    int sid = 0;            // 0: caller's stream, 1..3: side streams, HELPER_BASE + k: helper work on side stream k
    int cap_sid = -1;       // stream while a hipGraph is being captured (-1: same)
    int edge_from = -1, edge_to = -1; bool edge_in_graph = false;       // cross-stream edge ops (op_edge)
};

// kinds: [K_GEMM_CONV, K_GEMM_LIN) conv GEMM by tile config, [K_GEMM_LIN, K_GROUPNORM) linear GEMM by tile config (both ranges
// must hold gemm_num_tile_cfgs() entries: checked in mkd_ctx_create), then the rest
enum OpKind { K_GEMM_CONV = 0, K_GEMM_LIN = 64, K_GROUPNORM = 128, K_LAYERNORM, K_ATTENTION, K_GEGLU, K_CONV_DIRECT, K_TFM_TAIL, K_MISC, K_COUNT };

std::string kind_name(int k) {
    static const char* const rest[] = {"groupnorm", "layernorm", "attention", "geglu", "conv3x3_direct", "tfm_tail", "misc"};
    if (k < K_GEMM_LIN) return std::string("gemm_conv3x3_") + gemm_tile_cfg_name(k);
    if (k < K_GROUPNORM) return std::string("gemm_linear_") + gemm_tile_cfg_name(k - K_GEMM_LIN);
    return rest[k - K_GROUPNORM];
}

}  // namespace

struct mkd_ctx {
    mkd_net_config cfg;
    std::map<std::string, Param> params;
    std::vector<std::string> res_prefixes[2];     // per net, in execution order
    std::vector<std::string> st_prefixes[2];
    bool finalized = false;
    // LayerNorm folded into its consumer GEMMs (row sums emitted by the producer GEMM's epilogue).  Correct and tested,
    // but measured neutral-to-slower in the pipeline (the consumer inherits the producer's write-back wait that the
    // LayerNorm kernel used to absorb), so it is opt-in: MKD_FUSE_LN=1.
    bool fuse_ln = false;
    // LayerNorm "on the fly" (MKD_LN_FLY): the consumer GEMM (attn1 q|k|v, attn2 to_q, the GEGLU projection) runs on the raw rows with
    // the gamma-folded weights and takes the row statistics itself from its A fragments (two extra MFMAs per row fragment and
    // k-step); no LayerNorm kernel, nothing asked of the producer.  Removes 96 launches per evaluation.
    // Bit mask: 1 = norm1 (-> q|k|v), 2 = norm2 (-> attn2.to_q), 4 = norm3 (-> GEGLU projection).  Measured at batch 8, 256x256
    // (images/s): 0: 26.13, 2: 26.53, 3: 26.41, 7: 25.78 - every column tile of a wide consumer (N = 3d, 8d) repeats the statistics
    // MFMAs (+26 % on the M = 8192, N = 2560 GEGLU projection), the d x d to_q GEMM pays 2-7 % for a 5 us LayerNorm launch -> 2.
    int ln_fly = getenv("MKD_LN_FLY") ? atoi(getenv("MKD_LN_FLY")) : 2;
    // Transformers of at most ln_fly_rows rows (lane batch x tokens) take the mask ln_fly_small instead: with few rows the repeated
    // statistics cost nothing and only the removed launch counts.  The default threshold follows the evaluated batch (MKD_LN_FLY_ROWS
    // overrides; tools/exp_lnrows.sh, images/s without / with): batch 1 -> every transformer (6.51 / 6.70), batch 2 -> 256 rows
    // (11.75 / 11.90), otherwise 128 rows (batch 4: 18.75 / 18.84; batch 8 and 16 unchanged, 256 rows would cost them 1.2 %: those
    // are the split-K weight-streaming layers of the half-batch lanes, and statistics over the whole row forbid splitting K).
    int ln_fly_small = getenv("MKD_LN_FLY_SMALL") ? atoi(getenv("MKD_LN_FLY_SMALL")) : 7;
    int ln_fly_rows = getenv("MKD_LN_FLY_ROWS") ? atoi(getenv("MKD_LN_FLY_ROWS")) : -1;
    // MKD_GN_FUSED=1: GroupNorm statistics emitted by the producing kernel's epilogue (deterministic fixed-point atomics) + an
    // element-wise apply kernel, instead of the two-phase GroupNorm kernel.  Built, parity-tested - and OFF by default: measured at
    // batch 8, 256x256 the producers' device-scope atomics cost 0.36 ms per evaluation (6.43 vs 6.18 ms; with the atomics stubbed
    // out the same plan takes 6.06 ms: DESIGN.md §4.3).
    bool gn_fused = getenv("MKD_GN_FUSED") ? atoi(getenv("MKD_GN_FUSED")) != 0 : false;
    bool gn_colstats_only = getenv("MKD_GN_FUSED") && atoi(getenv("MKD_GN_FUSED")) == 2;      // experiment: statistics by a stand-alone kernel after every producer
    Arena gstat; char* gstat_base = nullptr; size_t gstat_cap = 0;

    // fused weights (built in finalize)
    std::map<std::string, bf16_t*> qkv_w, kv_w;   // by transformer prefix
    std::map<std::string, bf16_t*> ffg_w; std::map<std::string, float*> ffg_b;   // GEGLU proj with (value, gate) rows interleaved
    // LayerNorm folded into its consumer GEMMs: W' = W*gamma (bf16), s = rowsum(W'), b' = b + W.beta
    std::map<std::string, bf16_t*> q2_w; std::map<std::string, float*> qkv_s, qkv_b, q2_s, q2_b, ffg_s;
    std::map<std::string, bf16_t*> qkv_plain, ffp_w; std::map<std::string, float*> ffp_b;     // the unfolded counterparts
    std::map<std::string, bf16_t*> ffm_w; std::map<std::string, float*> ffm_b;                // [P.W2 | P], P.b2 + bp: FF2 + proj_out as one GEMM
    bool merge_ffout = getenv("MKD_MERGE_FFOUT") ? atoi(getenv("MKD_MERGE_FFOUT")) != 0 : true;
    // Fused row-local tail of the d = 320 transformer blocks (kernels_tfm.hip): everything after the self-attention product as ONE
    // launch per block instead of 7.  tfm_w / tfm_v: the block's weights re-packed in the order a wave consumes them (built in
    // finalize from the SAME folded tensors the unfused plan uses), tfm_kv: the cached context K / V in MFMA-operand order (built in
    // the prepare plan).  tfm_tail: 0 off, 1 on where the kernel covers the shape, -1 (default) the shape policy of use_tfm_tail().
    int tfm_tail = getenv("MKD_TFM_TAIL") ? atoi(getenv("MKD_TFM_TAIL")) : -1;
    std::map<std::string, bf16_t*> tfm_w; std::map<std::string, float*> tfm_v; std::map<std::string, bf16_t*> tfm_kv;
    // ... and the head of the same blocks (GroupNorm apply + proj_in + LayerNorm 1 . q|k|v as one launch behind a GroupNorm statistics
    // launch: 4 launches -> 2): tfm_head 0 off, 1 wherever the tail is fused (default)
    int tfm_head = getenv("MKD_TFM_HEAD") ? atoi(getenv("MKD_TFM_HEAD")) : 1;
    std::map<std::string, bf16_t*> tfm_hw; std::map<std::string, float*> tfm_hv;
    // ResBlocks with a 1x1 skip_connection: conv2 and the skip as ONE implicit GEMM over K = 9 Cout + Cin_block (GemmArgs::A2) where
    // conv2's plan is the gather kernel - [W_conv2 | W_skip] and b_conv2 + b_skip built at load time.  skip_fold: 0 off, 1 on (default)
    int skip_fold = getenv("MKD_SKIP_FOLD") ? atoi(getenv("MKD_SKIP_FOLD")) : 1;
    std::map<std::string, bf16_t*> fold_w; std::map<std::string, float*> fold_b;
    std::map<std::string, float*> f32_keep;      // fp32 copies of the weights that get folded (kept for re-finalize)
    bf16_t* emb_w[2] = {nullptr, nullptr};
    float* emb_b[2] = {nullptr, nullptr};
    int emb_total[2] = {0, 0};
    std::map<std::string, int> emb_off;           // resblock prefix -> column offset in emb projection
    std::vector<void*> owned;                     // every hipMalloc'd weight block
    int64_t weight_bytes = 0;
    bf16_t* zero_page = nullptr;

    // prepared state
    bool prepared = false;
    int B = 0, h = 0, w = 0;
    bool has_control = false, only_mid = false;
    float scales[64];
    int n_ctrl() const { return (int)encoder_spec().size() + 1; }
    static constexpr int NS = 4;              // streams / temp arenas: 0 = caller's stream, 1..3 = side streams
    static constexpr int HELPER_BASE = 16;    // sid HELPER_BASE + k: decoder helper GEMMs on side stream k (arena k); while capturing a
                                              // graph they run on their lane's own stream (Op::cap_sid)
    int lane_main = 0, helper_stream = 1;     // the lane being emitted and the stream its helpers use
    int cur_cap_sid = -1;
    static constexpr int SID_AUX = 4;         // first-stage decoder / text encoder plans: private workspaces, so that they may run
                                              // on another stream concurrently with an evaluation (pipelined decode)
    static constexpr int NA = NS + 1;         // workspace sets: one per stream + SID_AUX
    Arena persist, temp_arena[NA];
    int cur_sid = 0;
    static int arena_of(int sid) { return sid >= HELPER_BASE ? sid - HELPER_BASE : sid; }
    Arena& TA() { return temp_arena[arena_of(cur_sid)]; }
    char* persist_base = nullptr; char* temp_base[NA] = {};
    size_t persist_cap = 0, temp_cap[NA] = {};
    float* splitk_ws[NA] = {}; size_t splitk_ws_bytes[NA] = {}, splitk_need = 0;
    float* gn_ws[NA] = {}; size_t gn_ws_bytes[NA] = {}, gn_need = 0;
    hipStream_t side_streams[NS] = {};        // [0] unused (the caller's stream)
    hipStream_t stream_of(int sid) const { return (arena_of(sid) == 0 || run_serial) ? run_main : side_streams[arena_of(sid)]; }
    hipStream_t run_main = nullptr; bool run_serial = false;
    bool dual_stream = getenv("MKD_DUAL_STREAM") ? atoi(getenv("MKD_DUAL_STREAM")) != 0 : true;      // 0: the whole plan on the caller's stream (experiments)
    // measured at batch 8, 256x256 (ms per evaluation): no lanes 6.78, 2 decoder lanes 6.58, 4 decoder lanes 6.89, encoder lanes
    // on top +0.25: the encoder phase already runs two nets side by side, a third and fourth stream only add contention
    int dec_lanes = getenv("MKD_DEC_LANES") ? atoi(getenv("MKD_DEC_LANES")) : 2;          // 0 / 2 / 4 half- or quarter-batch decoder lanes
    // 2 lanes + a helper stream each (zero-conv combine / 1x1 skip GEMMs off the lanes' chains): 6.13 vs 6.20 ms per evaluation with the
    // round-2 tile table (it lost, 6.30 vs 6.01, with round 1's)
    bool lane_helpers = getenv("MKD_LANE_HELPERS") ? atoi(getenv("MKD_LANE_HELPERS")) != 0 : true;
    int dec_lanes_from = getenv("MKD_DEC_LANES_FROM") ? atoi(getenv("MKD_DEC_LANES_FROM")) : 0;   // deepest blocks as one full-batch chain first
    bool enc_lanes = getenv("MKD_ENC_LANES") ? atoi(getenv("MKD_ENC_LANES")) != 0 : false;
    bool dec_overlap = getenv("MKD_DEC_OVERLAP") ? atoi(getenv("MKD_DEC_OVERLAP")) != 0 : true;
    bool capturing = false;
    int plan_epoch = 0;
    int opt_epoch = 0, opt_epoch_planned = 0;      // bumped by mkd_ctx_set_option: the next mkd_prepare re-plans
    int graph_steps = getenv("MKD_GRAPH_STEPS") ? atoi(getenv("MKD_GRAPH_STEPS")) : 5;      // DDIM steps per captured graph
    // Shape policy (round 4, profiles/exp_r4_bigcfg_switches.txt): a GroupNorm over >= gn_2k_min_hw pixels per sample runs as the two
    // full-chip launches (statistics, apply) instead of the single launch that keeps a (sample, group chunk) slab in registers on
    // 32-128 workgroups: at 64x64 latents (512x512 images) the single launch holds too few CUs for too long (16.51 -> 16.26 ms per
    // evaluation); at 32x32 it is the other way round (8.36 -> 8.63 ms with guidance), so the rule is by tensor size.
    int gn_2k_min_hw = getenv("MKD_GN_2K_MINHW") ? atoi(getenv("MKD_GN_2K_MINHW")) : 4096;
    size_t persist_eps_begin = 0;
    std::vector<hipEvent_t> aux_ev; int aux_used = 0;        // cross-stream edges inside the decoder (re-used across plan rebuilds)
    std::vector<Op> plan_prepare, plan_eps;
    double flops_eps = 0; int launches_eps = 0;
    bool dry = false; bool counting_eps = false;
    std::map<std::string, Tensor> kv_cache;       // transformer prefix -> [B*77, 2d]
    Tensor hint_emb;
    bf16_t* ctx_bf16 = nullptr;
    const float* in_hint = nullptr; const float* in_context = nullptr;
    const float* in_hint2 = nullptr; const float* in_alpha = nullptr; bool has_interp = false;   // makeup interpolation (build-defined)
    // ---- first-stage decoder (AutoencoderKL.decode) ----
    bool vae_configured = false, vae_finalized = false;
    bool clip_configured = false, clip_finalized = false; mkd_clip_config ccfg{};
    Arena carena; char* carena_base = nullptr; size_t carena_cap = 0;
    std::vector<Op> plan_clip; int clip_B = 0, clip_T = 0;
    const int32_t* io_tokens = nullptr; float* io_ctx_out = nullptr;
    std::map<std::string, bf16_t*> clip_qkv_w; std::map<std::string, float*> clip_qkv_b;
    mkd_vae_config vcfg;
    std::map<std::string, bf16_t*> vae_fused;           // mid attention [Wq;Wk]
    std::map<std::string, float*> vae_fused_b;
    Arena varena; char* varena_base = nullptr; size_t varena_cap = 0;
    std::vector<Op> plan_vae; int vae_B = 0, vae_h = 0, vae_w = 0;
    const float* io_z = nullptr; float* io_img = nullptr; float io_inv_scale = 1.f;
    double flops_vae = 0;
    // per-call io
    const float* io_x = nullptr; const int64_t* io_t = nullptr; float* io_out = nullptr;
    // sampler buffers
    float* s_xa = nullptr; float* s_xb = nullptr; float* s_xin = nullptr; float* s_eps = nullptr;
    int64_t* s_t = nullptr;
    StepState* s_state = nullptr; StepState* h_state = nullptr;
    hipStream_t loop_stream = nullptr; hipEvent_t ev_loop_in = nullptr, ev_loop_out = nullptr;
    hipGraphExec_t multi_graph = nullptr; int multi_graph_steps = 0;      // MKD_GRAPH_STEPS consecutive steps as one graph
    hipGraphExec_t step_graph = nullptr; int step_graph_cfg = -1; float step_graph_scale = 0.f; int plan_generation = 0, step_graph_gen = -1;
    int step_graph_temb = -1, step_graph_batch = -1;
    // Graph mode 2 (MKD_GRAPH_MODE=2; default 1 = one captured graph per step): one step = LINEAR graphs, one per (stream, stretch
    // between two cross-stream edges), launched on their own streams and ordered by events.  A captured graph with two BRANCHES is
    // replayed with its branches serialised node by node (tools/micro/launch_floor.hip: 3.2 us per pair of empty nodes, 5.3-5.9 us
    // per pair of small kernels; tools/micro/two_phase.hip: 1836 us for a 720-launch step of small kernels), two linear graphs on two
    // streams run side by side (1.7 / 1.9-3.4 us per pair; 725 us per step).  In the real evaluation the launch floor does drop
    // (every kernel empty: 1.65 -> 0.93 ms) but the GroupNorm / LayerNorm / attention kernels no longer hide in the dispatcher's
    // gaps (0.09 -> 1.0 ms) and the GEMM work is 4.0 ms either way: 5.96 vs 5.75 ms per evaluation at batch 8, 3.12 vs 2.96 at
    // batch 1 - measured, kept as a switch, off.
    struct SegAction { int type; int sid; int idx; };      // type 0: launch seg_graphs[idx] / seg_eager[idx] on stream sid; 1: record event idx on sid; 2: sid waits for event idx
    struct Segment { hipGraphExec_t graph = nullptr; std::vector<OpFn> eager; };
    std::vector<Segment> segs; std::vector<SegAction> seg_actions; std::vector<hipEvent_t> seg_events;
    int seg_gen = -1, seg_cfg = -1, seg_temb = -1, seg_batch = -1; float seg_scale = 0.f;
    int graph_mode = getenv("MKD_GRAPH_MODE") ? atoi(getenv("MKD_GRAPH_MODE")) : 1;

    // ---------------------------------------------------------------------------------------------
    int ctx_len() const { return 77; }
    int temb_dim() const { return 4 * cfg.model_channels; }

    std::vector<BlockSpec> encoder_spec() const {
        std::vector<BlockSpec> v;
        const int mc = cfg.model_channels;
        v.push_back({0, cfg.in_channels, mc, false, false, 1});
        int ch = mc, ds = 1;
        for (int level = 0; level < cfg.n_levels; ++level) {
            const int mult = cfg.channel_mult[level];
            for (int i = 0; i < cfg.num_res_blocks; ++i) {
                v.push_back({1, ch, mult * mc, attn_at(ds), false, ds});
                ch = mult * mc;
            }
            if (level != cfg.n_levels - 1) {
                ds *= 2;
                v.push_back({2, ch, ch, false, false, ds});
            }
        }
        return v;
    }
    std::vector<BlockSpec> decoder_spec() const {
        std::vector<BlockSpec> enc = encoder_spec(), out;
        std::vector<int> chans;
        for (auto& b : enc) chans.push_back(b.cout);
        int ch = enc.back().cout, ds = enc.back().ds;
        const int mc = cfg.model_channels;
        for (int level = cfg.n_levels - 1; level >= 0; --level) {
            const int mult = cfg.channel_mult[level];
            for (int i = 0; i <= cfg.num_res_blocks; ++i) {
                const int ich = chans.back(); chans.pop_back();
                const bool up = level > 0 && i == cfg.num_res_blocks;
                out.push_back({1, ch + ich, mc * mult, attn_at(ds), up, ds});
                ch = mc * mult;
                if (up) ds /= 2;
            }
        }
        return out;
    }
    bool attn_at(int ds) const {
        for (int i = 0; i < cfg.n_attention_resolutions; ++i)
            if (cfg.attention_resolutions[i] == ds) return true;
        return false;
    }

    void add_param(const std::string& name, std::vector<int64_t> shape, int which) {
        Param p; p.shape = std::move(shape); p.which = which;
        params[name] = p;
    }
    void add_res(const std::string& p, int cin, int cout, int which) {
        const int te = temb_dim();
        add_param(p + ".in_layers.0.weight", {cin}, which); add_param(p + ".in_layers.0.bias", {cin}, which);
        add_param(p + ".in_layers.2.weight", {cout, cin, 3, 3}, which); add_param(p + ".in_layers.2.bias", {cout}, which);
        add_param(p + ".emb_layers.1.weight", {cout, te}, which); add_param(p + ".emb_layers.1.bias", {cout}, which);
        add_param(p + ".out_layers.0.weight", {cout}, which); add_param(p + ".out_layers.0.bias", {cout}, which);
        add_param(p + ".out_layers.3.weight", {cout, cout, 3, 3}, which); add_param(p + ".out_layers.3.bias", {cout}, which);
        if (cin != cout) {
            add_param(p + ".skip_connection.weight", {cout, cin, 1, 1}, which);
            add_param(p + ".skip_connection.bias", {cout}, which);
        }
        res_prefixes[which].push_back(p);
    }
    void add_st(const std::string& p, int ch, int which) {
        const int cd = cfg.context_dim;
        add_param(p + ".norm.weight", {ch}, which); add_param(p + ".norm.bias", {ch}, which);
        add_param(p + ".proj_in.weight", {ch, ch, 1, 1}, which); add_param(p + ".proj_in.bias", {ch}, which);
        add_param(p + ".proj_out.weight", {ch, ch, 1, 1}, which); add_param(p + ".proj_out.bias", {ch}, which);
        const std::string t = p + ".transformer_blocks.0";
        for (int a = 1; a <= 2; ++a) {
            const std::string ap = t + ".attn" + std::to_string(a);
            const int kd = a == 1 ? ch : cd;
            add_param(ap + ".to_q.weight", {ch, ch}, which);
            add_param(ap + ".to_k.weight", {ch, kd}, which);
            add_param(ap + ".to_v.weight", {ch, kd}, which);
            add_param(ap + ".to_out.0.weight", {ch, ch}, which);
            add_param(ap + ".to_out.0.bias", {ch}, which);
        }
        for (int n = 1; n <= 3; ++n) {
            add_param(t + ".norm" + std::to_string(n) + ".weight", {ch}, which);
            add_param(t + ".norm" + std::to_string(n) + ".bias", {ch}, which);
        }
        add_param(t + ".ff.net.0.proj.weight", {8 * ch, ch}, which); add_param(t + ".ff.net.0.proj.bias", {8 * ch}, which);
        add_param(t + ".ff.net.2.weight", {ch, 4 * ch}, which); add_param(t + ".ff.net.2.bias", {ch}, which);
        st_prefixes[which].push_back(p);
    }
    static std::string net_prefix(int which) { return which == 0 ? "model.diffusion_model." : "control_model."; }

    void build_param_spec() {
        const int mc = cfg.model_channels, te = temb_dim();
        for (int which = 0; which < 2; ++which) {
            const std::string P = net_prefix(which);
            add_param(P + "time_embed.0.weight", {te, mc}, which); add_param(P + "time_embed.0.bias", {te}, which);
            add_param(P + "time_embed.2.weight", {te, te}, which); add_param(P + "time_embed.2.bias", {te}, which);
            auto enc = encoder_spec();
            for (size_t i = 0; i < enc.size(); ++i) {
                const std::string p = P + "input_blocks." + std::to_string(i);
                const BlockSpec& b = enc[i];
                if (b.kind == 0) {
                    add_param(p + ".0.weight", {b.cout, b.cin, 3, 3}, which); add_param(p + ".0.bias", {b.cout}, which);
                } else if (b.kind == 1) {
                    add_res(p + ".0", b.cin, b.cout, which);
                    if (b.attn) add_st(p + ".1", b.cout, which);
                } else {
                    add_param(p + ".0.op.weight", {b.cout, b.cin, 3, 3}, which); add_param(p + ".0.op.bias", {b.cout}, which);
                }
            }
            const int ch = enc.back().cout;
            add_res(P + "middle_block.0", ch, ch, which);
            add_st(P + "middle_block.1", ch, which);
            add_res(P + "middle_block.2", ch, ch, which);
            if (which == 0) {
                auto dec = decoder_spec();
                for (size_t i = 0; i < dec.size(); ++i) {
                    const std::string p = P + "output_blocks." + std::to_string(i);
                    const BlockSpec& b = dec[i];
                    add_res(p + ".0", b.cin, b.cout, which);
                    int k = 1;
                    if (b.attn) { add_st(p + ".1", b.cout, which); k = 2; }
                    if (b.up) {
                        add_param(p + "." + std::to_string(k) + ".conv.weight", {b.cout, b.cout, 3, 3}, which);
                        add_param(p + "." + std::to_string(k) + ".conv.bias", {b.cout}, which);
                    }
                }
                add_param(P + "out.0.weight", {mc}, which); add_param(P + "out.0.bias", {mc}, which);
                add_param(P + "out.2.weight", {cfg.out_channels, mc, 3, 3}, which); add_param(P + "out.2.bias", {cfg.out_channels}, which);
            } else {
                int widths[9];
                widths[0] = cfg.hint_channels;
                for (int j = 0; j < 7; ++j) widths[j + 1] = cfg.hint_widths[j];
                widths[8] = mc;
                for (int j = 0; j < 8; ++j) {
                    add_param(P + "input_hint_block." + std::to_string(2 * j) + ".weight", {widths[j + 1], widths[j], 3, 3}, which);
                    add_param(P + "input_hint_block." + std::to_string(2 * j) + ".bias", {widths[j + 1]}, which);
                }
                for (size_t i = 0; i < enc.size(); ++i) {
                    add_param(P + "zero_convs." + std::to_string(i) + ".0.weight", {enc[i].cout, enc[i].cout, 1, 1}, which);
                    add_param(P + "zero_convs." + std::to_string(i) + ".0.bias", {enc[i].cout}, which);
                }
                add_param(P + "middle_block_out.0.weight", {ch, ch, 1, 1}, which);
                add_param(P + "middle_block_out.0.bias", {ch}, which);
            }
        }
    }

    // ---- weights ----------------------------------------------------------------------------------
    // Weight forms DERIVED at finalize() (concatenated / folded / packed copies) are rebuilt by every finalize after a weight was
    // loaded: they live in their own list and the previous generation is freed first (a reloaded checkpoint must not leave 1-2 GB of
    // stale copies behind; no plan can still use them: mkd_load_weight invalidates the prepared plan)
    std::vector<void*> derived; int64_t derived_bytes = 0; bool deriving = false;
    int dev_alloc(void** out, size_t bytes) {
        MKD_HIP_CHECK(hipMalloc(out, bytes ? bytes : 16));
        if (deriving) { derived.push_back(*out); derived_bytes += (int64_t)bytes; } else owned.push_back(*out);
        weight_bytes += (int64_t)bytes;
        return 0;
    }
    const bf16_t* wb(const std::string& name) const {
        auto it = params.find(name);
        return it == params.end() ? nullptr : (const bf16_t*)it->second.dev;
    }
    const float* wf(const std::string& name) const {
        auto it = params.find(name);
        return it == params.end() ? nullptr : (const float*)it->second.dev;
    }

    int load_weight(const char* name, const float* data, int ndim, const int64_t* shape) {
        auto it = params.find(name);
        if (it == params.end()) return mkd_fail(MKD_ERR_ARG, std::string("unknown weight name: ") + name);
        Param& p = it->second;
        if ((int)p.shape.size() != ndim) return mkd_fail(MKD_ERR_ARG, std::string("rank mismatch for ") + name);
        for (int i = 0; i < ndim; ++i)
            if (p.shape[i] != shape[i]) return mkd_fail(MKD_ERR_ARG, std::string("shape mismatch for ") + name);
        const int64_t n = p.numel();
        float* stage = nullptr;
        MKD_HIP_CHECK(hipMalloc((void**)&stage, n * sizeof(float)));
        hipError_t e = hipMemcpy(stage, data, n * sizeof(float), hipMemcpyDefault);
        if (e != hipSuccess) { hipFree(stage); return mkd_fail(MKD_ERR_HIP, std::string("hipMemcpy weight: ") + hipGetErrorString(e)); }
        int rc = 0;
        if (ndim == 1) {
            if (!p.dev) rc = dev_alloc(&p.dev, n * sizeof(float));
            if (!rc) {
                e = hipMemcpy(p.dev, stage, n * sizeof(float), hipMemcpyDeviceToDevice);
                if (e != hipSuccess) rc = mkd_fail(MKD_ERR_HIP, hipGetErrorString(e));
            }
        } else {
            if (!p.dev) rc = dev_alloc(&p.dev, n * sizeof(bf16_t));
            if (!rc) {
                if (ndim == 4) rc = launch_pack_conv_weight(stage, (bf16_t*)p.dev, (int)shape[0], (int)shape[1], (int)shape[2], (int)shape[3], 0);
                else rc = launch_f32_to_bf16(stage, (bf16_t*)p.dev, n, 0);
            }
        }
        e = hipDeviceSynchronize();
        const std::string nm(name);
        auto ends = [&](const char* suf) { const size_t L = strlen(suf); return nm.size() >= L && nm.compare(nm.size() - L, L, suf) == 0; };
        if (!rc && e == hipSuccess && (ends(".attn1.to_q.weight") || ends(".attn1.to_k.weight") || ends(".attn1.to_v.weight") ||
                                      ends(".attn2.to_q.weight") || ends(".ff.net.0.proj.weight") || ends(".ff.net.2.weight") ||
                                      ends(".proj_out.weight"))) {
            auto it2 = f32_keep.find(nm);
            if (it2 != f32_keep.end()) hipFree(it2->second); else weight_bytes += n * (int64_t)sizeof(float);
            f32_keep[nm] = stage;
        } else {
            hipFree(stage);
        }
        if (rc) return rc;
        if (e != hipSuccess) return mkd_fail(MKD_ERR_HIP, std::string("load_weight sync: ") + hipGetErrorString(e));
        p.loaded = true;
        if (p.which == 2) vae_finalized = false; else if (p.which == 3) clip_finalized = false; else { finalized = false; prepared = false; }
        return 0;
    }

    int concat_rows(bf16_t** out, const std::vector<std::string>& names) {
        size_t total = 0;
        for (auto& n : names) total += (size_t)params.at(n).numel();
        void* d = nullptr;
        int rc = dev_alloc(&d, total * sizeof(bf16_t));
        if (rc) return rc;
        size_t off = 0;
        for (auto& n : names) {
            const Param& p = params.at(n);
            MKD_HIP_CHECK(hipMemcpy((bf16_t*)d + off, p.dev, p.numel() * sizeof(bf16_t), hipMemcpyDeviceToDevice));
            off += p.numel();
        }
        *out = (bf16_t*)d;
        return 0;
    }

    int finalize() {
        for (auto& kv : params)
            if (kv.second.which < 2 && !kv.second.loaded) return mkd_fail(MKD_ERR_MISSING, "weight not loaded: " + kv.first);
        if (finalized) return 0;
        if (!zero_page) {
            void* z = nullptr;
            int rc = dev_alloc(&z, 4096);
            if (rc) return rc;
            MKD_HIP_CHECK(hipMemset(z, 0, 4096));
            zero_page = (bf16_t*)z;
        }
        if (!derived.empty()) {
            MKD_HIP_CHECK(hipDeviceSynchronize());
            for (void* q : derived) hipFree(q);
            derived.clear(); weight_bytes -= derived_bytes; derived_bytes = 0;
        }
        struct Guard { bool& f; Guard(bool& b) : f(b) { f = true; } ~Guard() { f = false; } } guard(deriving);
        qkv_w.clear(); kv_w.clear(); emb_off.clear(); ffg_w.clear(); ffg_b.clear();
        q2_w.clear(); qkv_s.clear(); qkv_b.clear(); q2_s.clear(); q2_b.clear(); ffg_s.clear(); qkv_plain.clear(); ffp_w.clear(); ffp_b.clear(); ffm_w.clear(); ffm_b.clear();
        tfm_w.clear(); tfm_v.clear(); tfm_hw.clear(); tfm_hv.clear(); fold_w.clear(); fold_b.clear();
        for (int which = 0; which < 2; ++which) {
            for (auto& p : st_prefixes[which]) {
                const std::string t = p + ".transformer_blocks.0";
                bf16_t* k = nullptr;
                int rc = concat_rows(&k, {t + ".attn2.to_k.weight", t + ".attn2.to_v.weight"});
                if (rc) return rc;
                kv_w[p] = k;
                {   // unfolded: [to_q; to_k; to_v] and the (value, gate)-interleaved GEGLU projection
                    bf16_t* q = nullptr;
                    rc = concat_rows(&q, {t + ".attn1.to_q.weight", t + ".attn1.to_k.weight", t + ".attn1.to_v.weight"});
                    if (rc) return rc;
                    qkv_plain[p] = q;
                    const Param& pw = params.at(t + ".ff.net.0.proj.weight");
                    const Param& pb = params.at(t + ".ff.net.0.proj.bias");
                    const int64_t inner = pw.shape[0] / 2, kd = pw.shape[1];
                    void* w2 = nullptr; void* b2 = nullptr;
                    rc = dev_alloc(&w2, pw.numel() * sizeof(bf16_t)); if (rc) return rc;
                    rc = dev_alloc(&b2, pb.numel() * sizeof(float)); if (rc) return rc;
                    const size_t rowb = kd * sizeof(bf16_t);
                    MKD_HIP_CHECK(hipMemcpy2D(w2, 2 * rowb, pw.dev, rowb, rowb, inner, hipMemcpyDeviceToDevice));
                    MKD_HIP_CHECK(hipMemcpy2D((char*)w2 + rowb, 2 * rowb, (const char*)pw.dev + inner * rowb, rowb, rowb, inner, hipMemcpyDeviceToDevice));
                    MKD_HIP_CHECK(hipMemcpy2D(b2, 8, pb.dev, 4, 4, inner, hipMemcpyDeviceToDevice));
                    MKD_HIP_CHECK(hipMemcpy2D((char*)b2 + 4, 8, (const char*)pb.dev + inner * 4, 4, 4, inner, hipMemcpyDeviceToDevice));
                    ffp_w[p] = (bf16_t*)w2; ffp_b[p] = (float*)b2;
                    // FF2 and proj_out are both linear with nothing in between: one GEMM over [gg | h2] (K = 5d)
                    auto kp = f32_keep.find(p + ".proj_out.weight"); auto kw = f32_keep.find(t + ".ff.net.2.weight");
                    if (merge_ffout && kp != f32_keep.end() && kw != f32_keep.end()) {
                        const int dd = (int)params.at(p + ".norm.weight").numel();
                        void* wm = nullptr; void* bm = nullptr;
                        rc = dev_alloc(&wm, (size_t)dd * 5 * dd * sizeof(bf16_t)); if (rc) return rc;
                        rc = dev_alloc(&bm, (size_t)dd * sizeof(float)); if (rc) return rc;
                        rc = launch_merge_ff_out(kp->second, kw->second, wf(t + ".ff.net.2.bias"), wf(p + ".proj_out.bias"), (bf16_t*)wm, (float*)bm, dd, 0);
                        if (rc) return rc;
                        ffm_w[p] = (bf16_t*)wm; ffm_b[p] = (float*)bm;
                    }
                }
                const int d = (int)params.at(p + ".norm.weight").numel();
                auto keep = [&](const std::string& n) -> const float* {
                    auto it = f32_keep.find(n);
                    return it == f32_keep.end() ? nullptr : it->second;
                };
                auto fvec = [&](int n, float** out) { void* v = nullptr; int r = dev_alloc(&v, (size_t)n * sizeof(float)); *out = (float*)v; return r; };
                // attn1: [to_q; to_k; to_v] . LN1
                {
                    void* wq = nullptr; float* sv = nullptr; float* bv = nullptr;
                    rc = dev_alloc(&wq, (size_t)3 * d * d * sizeof(bf16_t)); if (rc) return rc;
                    rc = fvec(3 * d, &sv); if (rc) return rc;
                    rc = fvec(3 * d, &bv); if (rc) return rc;
                    const char* parts[3] = {".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_v.weight"};
                    for (int j = 0; j < 3; ++j) {
                        const float* wsrc = keep(t + parts[j]);
                        if (!wsrc) return mkd_fail(MKD_ERR_MISSING, "fp32 copy missing for " + t + parts[j]);
                        rc = launch_fold_layernorm(wsrc, wf(t + ".norm1.weight"), wf(t + ".norm1.bias"), nullptr, d, d, (bf16_t*)wq,
                                                   j * d, 1, sv, bv, 0);
                        if (rc) return rc;
                    }
                    qkv_w[p] = (bf16_t*)wq; qkv_s[p] = sv; qkv_b[p] = bv;
                }
                // attn2.to_q . LN2
                {
                    void* wq = nullptr; float* sv = nullptr; float* bv = nullptr;
                    rc = dev_alloc(&wq, (size_t)d * d * sizeof(bf16_t)); if (rc) return rc;
                    rc = fvec(d, &sv); if (rc) return rc;
                    rc = fvec(d, &bv); if (rc) return rc;
                    const float* wsrc = keep(t + ".attn2.to_q.weight");
                    if (!wsrc) return mkd_fail(MKD_ERR_MISSING, "fp32 copy missing for " + t + ".attn2.to_q.weight");
                    rc = launch_fold_layernorm(wsrc, wf(t + ".norm2.weight"), wf(t + ".norm2.bias"), nullptr, d, d, (bf16_t*)wq, 0, 1, sv, bv, 0);
                    if (rc) return rc;
                    q2_w[p] = (bf16_t*)wq; q2_s[p] = sv; q2_b[p] = bv;
                }
                // ff.net.0.proj [8d][d] . LN3: rows [0,4d) = value, [4d,8d) = gate  ->  row 2j = value_j, row 2j+1 = gate_j
                {
                    const int inner = 4 * d;
                    void* w2 = nullptr; float* sv = nullptr; float* bv = nullptr;
                    rc = dev_alloc(&w2, (size_t)2 * inner * d * sizeof(bf16_t)); if (rc) return rc;
                    rc = fvec(2 * inner, &sv); if (rc) return rc;
                    rc = fvec(2 * inner, &bv); if (rc) return rc;
                    const float* wsrc = keep(t + ".ff.net.0.proj.weight");
                    if (!wsrc) return mkd_fail(MKD_ERR_MISSING, "fp32 copy missing for " + t + ".ff.net.0.proj.weight");
                    const float* bsrc = wf(t + ".ff.net.0.proj.bias");
                    for (int half = 0; half < 2; ++half) {
                        rc = launch_fold_layernorm(wsrc + (size_t)half * inner * d, wf(t + ".norm3.weight"), wf(t + ".norm3.bias"),
                                                   bsrc + half * inner, inner, d, (bf16_t*)w2, half, 2, sv, bv, 0);
                        if (rc) return rc;
                    }
                    ffg_w[p] = (bf16_t*)w2; ffg_b[p] = bv; ffg_s[p] = sv;
                }
                if (tfm_tail_weight_bytes(d) && cfg.num_heads == 8 && ffm_w.count(p)) {
                    void* wp = nullptr; void* vp = nullptr;
                    rc = dev_alloc(&wp, tfm_tail_weight_bytes(d)); if (rc) return rc;
                    rc = dev_alloc(&vp, tfm_tail_vec_bytes(d)); if (rc) return rc;
                    TfmTailWeights tw{wb(t + ".attn1.to_out.0.weight"), wf(t + ".attn1.to_out.0.bias"), q2_w.at(p), q2_s.at(p), q2_b.at(p),
                                      wb(t + ".attn2.to_out.0.weight"), wf(t + ".attn2.to_out.0.bias"), ffg_w.at(p), ffg_s.at(p), ffg_b.at(p),
                                      ffm_w.at(p), ffm_b.at(p)};
                    rc = tfm_tail_pack_weights(d, tw, (bf16_t*)wp, (float*)vp, 0);
                    if (rc) return rc;
                    tfm_w[p] = (bf16_t*)wp; tfm_v[p] = (float*)vp;
                    void* hwp = nullptr; void* hvp = nullptr;
                    rc = dev_alloc(&hwp, tfm_head_weight_bytes(d)); if (rc) return rc;
                    rc = dev_alloc(&hvp, tfm_head_vec_bytes(d)); if (rc) return rc;
                    TfmHeadWeights hw{wf(p + ".norm.weight"), wf(p + ".norm.bias"), wb(p + ".proj_in.weight"), wf(p + ".proj_in.bias"),
                                      qkv_w.at(p), qkv_s.at(p), qkv_b.at(p)};
                    rc = tfm_head_pack_weights(d, hw, (bf16_t*)hwp, (float*)hvp, 0);
                    if (rc) return rc;
                    tfm_hw[p] = (bf16_t*)hwp; tfm_hv[p] = (float*)hvp;
                }
            }
            for (auto& p : res_prefixes[which]) {
                auto sk = params.find(p + ".skip_connection.weight");
                if (sk == params.end()) continue;
                const Param& cw = params.at(p + ".out_layers.3.weight");
                const int cout = (int)cw.shape[0], cs = (int)sk->second.shape[1];
                if (cw.shape[1] != cout || (9 * cout) % 64 || cs % 64) continue;
                const size_t kc = (size_t)9 * cout, kt = kc + cs;
                void* wm = nullptr; void* bm = nullptr;
                int rc = dev_alloc(&wm, (size_t)cout * kt * sizeof(bf16_t)); if (rc) return rc;
                rc = dev_alloc(&bm, (size_t)cout * sizeof(float)); if (rc) return rc;
                MKD_HIP_CHECK(hipMemcpy2D(wm, kt * sizeof(bf16_t), cw.dev, kc * sizeof(bf16_t), kc * sizeof(bf16_t), cout, hipMemcpyDeviceToDevice));
                MKD_HIP_CHECK(hipMemcpy2D((char*)wm + kc * sizeof(bf16_t), kt * sizeof(bf16_t), sk->second.dev, (size_t)cs * sizeof(bf16_t), (size_t)cs * sizeof(bf16_t), cout,
                                          hipMemcpyDeviceToDevice));
                std::vector<float> b1(cout), b2(cout);
                MKD_HIP_CHECK(hipMemcpy(b1.data(), wf(p + ".out_layers.3.bias"), cout * sizeof(float), hipMemcpyDeviceToHost));
                MKD_HIP_CHECK(hipMemcpy(b2.data(), wf(p + ".skip_connection.bias"), cout * sizeof(float), hipMemcpyDeviceToHost));
                for (int i = 0; i < cout; ++i) b1[i] += b2[i];
                MKD_HIP_CHECK(hipMemcpy(bm, b1.data(), cout * sizeof(float), hipMemcpyHostToDevice));
                fold_w[p] = (bf16_t*)wm; fold_b[p] = (float*)bm;
            }
            std::vector<std::string> wn;
            int off = 0;
            for (auto& p : res_prefixes[which]) {
                wn.push_back(p + ".emb_layers.1.weight");
                emb_off[p] = off;
                off += (int)params.at(p + ".emb_layers.1.bias").numel();
            }
            emb_total[which] = off;
            int rc = concat_rows(&emb_w[which], wn);
            if (rc) return rc;
            void* b = nullptr;
            rc = dev_alloc(&b, off * sizeof(float));
            if (rc) return rc;
            emb_b[which] = (float*)b;
            for (auto& p : res_prefixes[which]) {
                const Param& bp = params.at(p + ".emb_layers.1.bias");
                MKD_HIP_CHECK(hipMemcpy(emb_b[which] + emb_off[p], bp.dev, bp.numel() * sizeof(float), hipMemcpyDeviceToDevice));
            }
        }
        MKD_HIP_CHECK(hipDeviceSynchronize());
        finalized = true;
        prepared = false;
        return 0;
    }

    // ---- plan building ------------------------------------------------------------------------------
    void push(std::vector<Op>& plan, OpFn f, int launches, double flops, int kind = K_MISC, const std::string& label = "") {
        if (counting_eps) { launches_eps += launches; flops_eps += flops; }
        if (!dry) {
            Op o; o.fn = std::move(f); o.kind = kind; o.flops = flops; o.launches = launches; o.label = label; o.sid = cur_sid; o.cap_sid = cur_cap_sid;
            o.temb = cur_temb;
            plan.push_back(std::move(o));
        }
    }
    std::vector<Op>* cur_plan = nullptr;
    bool cur_temb = false;
    OpDesc& last_desc() { static OpDesc dummy; return dry ? dummy : cur_plan->back().d; }      // descriptor of the op just pushed

    Tensor talloc(Arena& a, int B_, int H_, int W_, int C_) {
        Tensor t; t.B = B_; t.H = H_; t.W = W_; t.C = C_; t.ld = C_;
        t.p = (bf16_t*)a.alloc((size_t)B_ * H_ * W_ * C_ * sizeof(bf16_t));
        return t;
    }

    // Per-launch XCD tile order (GemmArgs::xcd_mode).  In launch order workgroup (m-tile x, n-tile y) runs on XCD (x + gx y) mod 8: an A
    // tile is pulled through ONE of the 8 non-coherent L2s (all its n-tiles run there), but every XCD walks ALL n-tiles, i.e. each L2
    // streams the whole weight matrix from the Infinity Cache - and L2 misses are what this loop waits for (19 GB per evaluation,
    // DESIGN.md 4.5).  Mode 1 gives every XCD one contiguous run of the tile sequence with m-tiles fastest: a weight tile lives in ONE L2,
    // an A tile in up to gy of them.  So: mode 1 where the weights outweigh the activations, M <= xcd_auto_ratio * N (both operands have
    // K columns).  MKD_XCD_AUTO_RATIO (0 = launch order everywhere); measured at batch 8, 256x256, ms per evaluation, two rounds
    // (tools/exp_r3_xcd_auto.sh): off 5.604 / 5.610, ratio 1 5.557 / 5.549, 2 5.547 / 5.548, 4 5.557 / 5.579, 8 5.570 / 5.564, 16 5.637 /
    // 5.648 (= mode 1 everywhere, MKD_XCD_MODE=1: 5.645); by M alone (N >= 640): M <= 512 5.576 / 5.564, <= 2048 5.565 / 5.544.
    // Results do not depend on the order (bit-identical: test_xcd_auto_order_changes_no_bit).
    // round 4: default 1 (was 2): equal at batch 8 / guidance / interpolation, +0.8 % at 512x512 (profiles/exp_r4_bigcfg_switches.txt, exp_r3_xcd_configs.txt)
    float xcd_auto_ratio = getenv("MKD_XCD_AUTO_RATIO") ? (float)atof(getenv("MKD_XCD_AUTO_RATIO")) : 1.0f;
    void op_gemm(GemmArgs a, int force_splitk = 0) {
        a.zero = zero_page;
        a.splitk = force_splitk;
        if (xcd_auto_ratio > 0.f && !a.xcd_mode && (float)a.M <= xcd_auto_ratio * (float)a.N) a.xcd_mode = 1;
        if (gn_colstats_only && a.gn_stat) {
            GnOut g; g.gst = a.gn_stat; g.cg = a.gn_cg; g.coff = a.gn_coff; g.hw = a.gn_hw;
            a.gn_stat = nullptr;
            op_gemm(a, force_splitk);
            op_colstats((const bf16_t*)a.C, a.ldc, a.M / g.hw, g.hw, a.N, g);
            return;
        }
        int cfg_i = 0, s = 1;
        if (gemm_resolve(a, &cfg_i, &s)) { cfg_i = 1; s = 1; }       // (an unsupported forced tile fails again, loudly, at launch)
        const size_t need = gemm_ws_bytes(a.M, a.N, s);
        if (need > splitk_need) splitk_need = need;
        mkd_ctx* self = this;
        const int sid = cur_sid;
        push(*cur_plan, [self, a, sid](hipStream_t st) {
                 GemmArgs b = a; b.ws = self->splitk_ws[arena_of(sid)]; b.ws_bytes = self->splitk_ws_bytes[arena_of(sid)];
                 return launch_gemm(b, st); },
             s > 1 ? 2 : 1, 2.0 * a.M * a.N * a.K, (a.conv ? K_GEMM_CONV : K_GEMM_LIN) + cfg_i,
             "M=" + std::to_string(a.M) + " N=" + std::to_string(a.N) + " K=" + std::to_string(a.K) + " conv=" + std::to_string(a.conv) +
             " stride=" + std::to_string(a.stride) + " up=" + std::to_string(a.up) + " splitk=" + std::to_string(s) +
             " res=" + std::to_string(a.R != nullptr) + " f32=" + std::to_string(a.out_f32) + " Hin=" + std::to_string(a.Hin) +
             " Win=" + std::to_string(a.Win) + " Cin=" + std::to_string(a.Cin) + " Hout=" + std::to_string(a.Hout) + " Wout=" + std::to_string(a.Wout) +
             " gn=" + std::to_string(a.gn_stat != nullptr));
        OpDesc& d = last_desc(); d.type = D_GEMM; d.sid = sid; d.gemm = a;
    }
    void op_linear(const bf16_t* A, int lda, int M, int K, const bf16_t* W, int N, const Epi& e, void* C, int ldc, bool f32out = false, int force_splitk = 0) {
        GemmArgs a; memset(&a, 0, sizeof(a));
        a.A = A; a.lda = lda; a.W = W; a.ldw = K; a.bias = e.bias; a.rowbias = e.rowbias; a.ldrb = e.ldrb;
        a.rows_per_batch = e.rpb; a.R = e.R; a.ldr = e.ldr; a.scale = e.scale; a.act = e.act;
        a.C = C; a.ldc = ldc; a.out_f32 = f32out ? 1 : 0; a.M = M; a.N = N; a.K = K; a.conv = 0;
        a.ln_s = e.ln_s; a.ln_eps = 1e-5f; a.stat_in = e.stat_in; a.stat_in_slots = e.stat_slots; a.stat_out = e.stat_out;
        a.gn_stat = e.gn.gst; a.gn_cg = e.gn.cg; a.gn_coff = e.gn.coff; a.gn_hw = e.gn.hw;
        op_gemm(a, force_splitk);
    }
    // 3x3 conv, pad 1
    void op_conv(const Tensor& in, const bf16_t* W, int N, int stride, int up, const Epi& e, bf16_t* C, int ldc) {
        op_gemm(conv_args(in, W, N, stride, up, e, C, ldc));
    }
    GemmArgs conv_args(const Tensor& in, const bf16_t* W, int N, int stride, int up, const Epi& e, bf16_t* C, int ldc) {
        GemmArgs a; memset(&a, 0, sizeof(a));
        const int Hs = in.H << up, Ws = in.W << up;
        const int Ho = (Hs - 1) / stride + 1, Wo = (Ws - 1) / stride + 1;
        a.A = in.p; a.lda = in.ld; a.W = W; a.ldw = 9 * in.C; a.bias = e.bias; a.rowbias = e.rowbias; a.ldrb = e.ldrb;
        a.rows_per_batch = e.rpb; a.R = e.R; a.ldr = e.ldr; a.scale = e.scale; a.act = e.act;
        a.C = C; a.ldc = ldc; a.out_f32 = 0; a.M = in.B * Ho * Wo; a.N = N; a.K = 9 * in.C; a.conv = 1;
        a.Hin = in.H; a.Win = in.W; a.Cin = in.C; a.Hout = Ho; a.Wout = Wo; a.stride = stride; a.up = up;
        a.gn_stat = e.gn.gst; a.gn_cg = e.gn.cg; a.gn_coff = e.gn.coff; a.gn_hw = e.gn.hw;
        return a;
    }
    // A split-K GEMM whose output is read by a GroupNorm: instead of [GEMM -> partial slabs][reduce + epilogue -> tensor][GroupNorm]
    // emit [GEMM -> partial slabs][one kernel: reduce + epilogue + GroupNorm (+ the raw tensor when `raw` says it is needed)].
    // Returns false (nothing emitted) when the GEMM of this shape is not split or the geometry does not fit: the caller then
    // emits the separate ops.  MKD_GN_SLAB=0 turns it off; MKD_GN_SLAB_MINC: only for at least this many channels.  Measured at batch
    // 8, 256x256 (ms per evaluation): off 6.12, all ResBlocks 6.14 (40 launches fewer, but with C = 320 / 640 the GroupNorm grid of
    // 64 / 128 workgroups reads 31 MB of slabs slower than the full-chip reduce kernel), C >= 1280 only 6.10 -> default 1280.
    bool gn_slab = getenv("MKD_GN_SLAB") ? atoi(getenv("MKD_GN_SLAB")) != 0 : true;
    int gn_slab_minc = getenv("MKD_GN_SLAB_MINC") ? atoi(getenv("MKD_GN_SLAB_MINC")) : 1280;
    bool op_gemm_then_gn(GemmArgs a, bool raw, int nb, int hw, const float* gamma, const float* beta, float eps, int silu, bf16_t* y, int ld_y) {
        if (!gn_slab || a.gn_stat || a.N < gn_slab_minc || !gn_from_slabs_supported(nb, hw, a.N) || a.M != nb * hw) return false;
        a.zero = zero_page; a.splitk = 0;
        int cfg_i = 0, sk = 1;
        if (gemm_resolve(a, &cfg_i, &sk) || sk < 2) return false;
        a.defer_epilogue = 1; a.expect_splitk = sk;   // (launch_gemm fails if it would resolve to another slab count than the one planned here)
        op_gemm(a);                                   // (counts 2 launches for a split GEMM: corrected below)
        if (counting_eps) --launches_eps;
        if (!dry) cur_plan->back().launches = 1;
        mkd_ctx* self = this;
        const int sid = cur_sid;
        push(*cur_plan, [self, a, sid, sk, raw, nb, hw, gamma, beta, eps, silu, y, ld_y](hipStream_t st) {
            GemmArgs b = a; b.ws = self->splitk_ws[arena_of(sid)]; b.splitk = sk;
            if (!raw) b.C = nullptr;
            return launch_gn_from_slabs(b, gamma, beta, eps, silu, y, ld_y, nb, hw, st);
        }, 1, 0.0, K_GROUPNORM, "slab B=" + std::to_string(nb) + " HW=" + std::to_string(hw) + " C=" + std::to_string(a.N) + " splitk=" + std::to_string(sk));
        OpDesc& d = last_desc(); d.type = D_GN_SLAB; d.sid = sid; d.gemm = a; d.sk = sk; d.raw = raw ? 1 : 0; d.nio = NormIo{nullptr, y, gamma, beta};
        d.eps = eps; d.silu = silu; d.ld_out = ld_y; d.nb = nb; d.hw = hw;
        if (!dry) cur_plan->back().bytes = (double)a.M * a.N * (sizeof(float) * sk + sizeof(bf16_t) * (raw ? 2 : 1));      // sk fp32 slabs in, bf16 out
        return true;
    }
    void op_gn(const Tensor& in, const float* gamma, const float* beta, float eps, int silu, bf16_t* out, int ld_out) {
        if (in.gst) {          // statistics were accumulated by the producers of `in`: element-wise apply
            Tensor t = in;
            push(*cur_plan, [t, gamma, beta, eps, silu, out, ld_out](hipStream_t st) {
                return launch_gn_apply_stats(t.p, t.ld, gamma, beta, eps, silu, out, ld_out, t.B, t.H * t.W, t.C, t.gst, st);
            }, 1, 0.0, K_GROUPNORM, "apply B=" + std::to_string(in.B) + " HW=" + std::to_string(in.H * in.W) + " C=" + std::to_string(in.C));
            return;
        }
        const size_t need = groupnorm_partials_bytes(in.B, in.H * in.W, 32);
        if (need > gn_need) gn_need = need;
        mkd_ctx* self = this;
        Tensor t = in;
        const int sid = cur_sid;
        push(*cur_plan, [self, t, gamma, beta, eps, silu, out, ld_out, sid](hipStream_t st) {
            return launch_groupnorm(t.p, t.ld, gamma, beta, eps, silu, out, ld_out, t.B, t.H * t.W, t.C, 32, self->gn_ws[arena_of(sid)], st, nullptr, self->gn_2k_min_hw);
        }, 1, 0.0, K_GROUPNORM, "B=" + std::to_string(in.B) + " HW=" + std::to_string(in.H * in.W) + " C=" + std::to_string(in.C));
        OpDesc& d = last_desc(); d.type = D_GN; d.sid = sid; d.nio = NormIo{in.p, out, gamma, beta}; d.eps = eps; d.silu = silu;
        d.ld_in = in.ld; d.ld_out = ld_out; d.nb = in.B; d.hw = in.H * in.W; d.C = in.C;
        if (!dry) cur_plan->back().bytes = 2.0 * sizeof(bf16_t) * in.B * in.H * in.W * (double)in.C;          // one read + one write of the tensor
    }
    void op_ln(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int rows, int d, int ldx = 0) {
        push(*cur_plan, [=](hipStream_t st) { return launch_layernorm(x, gamma, beta, 1e-5f, y, rows, d, st, ldx); }, 1, 0.0, K_LAYERNORM,
             "rows=" + std::to_string(rows) + " d=" + std::to_string(d));
        OpDesc& ds = last_desc(); ds.type = D_LN; ds.nio = NormIo{x, y, gamma, beta}; ds.eps = 1e-5f; ds.nb = rows; ds.C = d; ds.ld_in = ldx;
        if (!dry) cur_plan->back().bytes = 2.0 * sizeof(bf16_t) * rows * (double)d;
    }
    void op_attn(const bf16_t* q, int ldq, const bf16_t* k, int ldk, const bf16_t* v, int ldv, bf16_t* o, int ldo,
                 int B_, int Tq, int Tk, int heads, int dh) {
        const float scale = 1.0f / sqrtf((float)dh);
        push(*cur_plan, [=](hipStream_t st) { return launch_attention(q, ldq, k, ldk, v, ldv, o, ldo, B_, Tq, Tk, heads, dh, scale, st); },
             1, 4.0 * B_ * heads * (double)Tq * Tk * dh, K_ATTENTION,
             "B=" + std::to_string(B_) + " Tq=" + std::to_string(Tq) + " Tk=" + std::to_string(Tk) + " heads=" + std::to_string(heads) + " dh=" + std::to_string(dh));
        OpDesc& d = last_desc(); d.type = D_ATTN; d.aio = AttnIo{q, k, v, o}; d.ldq = ldq; d.ldk = ldk; d.ldv = ldv; d.ldo = ldo; d.nb = B_; d.Tq = Tq; d.Tk = Tk;
        d.heads = heads; d.dh = dh;
    }
    void op_geglu(const bf16_t* x, bf16_t* y, int rows, int inner) {
        push(*cur_plan, [=](hipStream_t st) { return launch_geglu(x, y, rows, inner, st); }, 1, 0.0, K_GEGLU, "rows=" + std::to_string(rows) + " inner=" + std::to_string(inner));
    }
    void op_copy(const bf16_t* src, int ld_src, bf16_t* dst, int ld_dst, int rows, int cols) {
        push(*cur_plan, [=](hipStream_t st) { return launch_copy_strided(src, ld_src, dst, ld_dst, rows, cols, st); }, 1, 0.0);
    }

    // ResBlock (App. A.2).  x may be a concat buffer (ld == C).  Writes [rows, cout] at (out, ldo).
    // side_skip: run the 1x1 skip_connection GEMM on the side stream, concurrently with GN -> conv -> GN (decoder only:
    // the side stream is idle there).
    void resblock(const std::string& p, const Tensor& x, int cout, const float* embproj, int ld_emb, bf16_t* out, int ldo,
                  bool side_skip = false, const GnOut& go = GnOut()) {
        const size_t mk = TA().mark();
        const int rows = x.rows(), hw = x.H * x.W;
        Tensor t4;
#ifdef MKD_EXP_ABLATE
        // experiment build only (MKD_EXP_SKIP & 32): no skip_connection GEMMs - conv2 takes its own input as residual (WRONG results):
        // the bound for folding the 1x1 skip into conv2's K loop
        static const bool no_skip = getenv("MKD_EXP_SKIP") && (atoi(getenv("MKD_EXP_SKIP")) & 32);
#else
        constexpr bool no_skip = false;
#endif
        if (no_skip && x.C != cout) side_skip = false;
        // conv2 + skip as one implicit GEMM when conv2's plan for this shape is the gather kernel (the decoder lanes' shapes)
        GemmArgs fa; memset(&fa, 0, sizeof(fa));
        bool fold = false;
        if (x.C != cout && skip_fold && !no_skip && fold_w.count(p) && x.ld % 8 == 0) {
            Tensor t3s; t3s.p = nullptr; t3s.B = x.B; t3s.H = x.H; t3s.W = x.W; t3s.C = cout; t3s.ld = cout;
            Epi ef; ef.bias = fold_b.at(p); ef.gn = go;
            fa = conv_args(t3s, fold_w.at(p), cout, 1, 0, ef, out, ldo);
            fa.A2 = x.p; fa.lda2 = x.ld; fa.K2 = x.C; fa.K += x.C; fa.ldw = fa.K; fa.zero = zero_page;
            int cfg_i = 0, sk_i = 1;
            fold = gemm_resolve(fa, &cfg_i, &sk_i) == 0;
        }
        if (fold) side_skip = false;
        if (x.C != cout && side_skip) {
            t4 = talloc(TA(), x.B, x.H, x.W, cout);
            op_edge(lane_main, helper_stream);           // helper stream waits for the block input (written on the lane's stream)
            cur_sid = HELPER_BASE + helper_stream; cur_cap_sid = lane_main;
            Epi es; es.bias = wf(p + ".skip_connection.bias");
            op_linear(x.p, x.ld, rows, x.C, wb(p + ".skip_connection.weight"), cout, es, t4.p, t4.ld);
            cur_sid = lane_main; cur_cap_sid = -1;
        }
        Tensor t1 = talloc(TA(), x.B, x.H, x.W, x.C);
        op_gn(x, wf(p + ".in_layers.0.weight"), wf(p + ".in_layers.0.bias"), 1e-5f, 1, t1.p, t1.ld);
        Tensor t2 = talloc(TA(), x.B, x.H, x.W, cout);
        want_stats(t2);
        Epi e1; e1.bias = wf(p + ".in_layers.2.bias"); e1.rowbias = embproj + emb_off.at(p); e1.ldrb = ld_emb; e1.rpb = hw; e1.gn = gn_of(t2);
        Tensor t3 = talloc(TA(), x.B, x.H, x.W, cout);
        // conv1 -> GroupNorm -> conv2: nothing else reads conv1's output, so a split conv1 feeds the GroupNorm straight from its slabs
        if (!op_gemm_then_gn(conv_args(t1, wb(p + ".in_layers.2.weight"), cout, 1, 0, e1, t2.p, t2.ld), /*raw=*/false, x.B, hw,
                             wf(p + ".out_layers.0.weight"), wf(p + ".out_layers.0.bias"), 1e-5f, 1, t3.p, t3.ld)) {
            op_conv(t1, wb(p + ".in_layers.2.weight"), cout, 1, 0, e1, t2.p, t2.ld);
            op_gn(t2, wf(p + ".out_layers.0.weight"), wf(p + ".out_layers.0.bias"), 1e-5f, 1, t3.p, t3.ld);
        }
        if (fold) {
            fa.A = t3.p; fa.lda = t3.ld;
            op_gemm(fa);
            TA().release(mk);
            return;
        }
        Epi e2; e2.bias = wf(p + ".out_layers.3.bias"); e2.gn = go;
        if (x.C != cout && side_skip) {
            op_edge(helper_stream, lane_main);           // the lane waits for the skip GEMM
            e2.R = t4.p; e2.ldr = t4.ld;
        } else if (x.C != cout && no_skip) {
            e2.R = t3.p; e2.ldr = t3.ld;
        } else if (x.C != cout) {
            t4 = talloc(TA(), x.B, x.H, x.W, cout);
            Epi es; es.bias = wf(p + ".skip_connection.bias");
            op_linear(x.p, x.ld, rows, x.C, wb(p + ".skip_connection.weight"), cout, es, t4.p, t4.ld);
            e2.R = t4.p; e2.ldr = t4.ld;
        } else {
            e2.R = x.p; e2.ldr = x.ld;
        }
        op_conv(t3, wb(p + ".out_layers.3.weight"), cout, 1, 0, e2, out, ldo);
        TA().release(mk);
    }

    // Where the fused tail replaces the 7 launches (tools/bench_tfm_tail.py, profiles/exp_r4_tfm_tail_standalone.txt: 0.49 of the
    // chain's time at M >= 16384 rows, 0.73 at 8192, 1.08 at 4096 - a workgroup streams the block's 3.3 MB of weights whatever M is).
    bool use_tfm_tail(const std::string& p, int d, int M, int T, bool producer_ln, const GnOut& go) const {
        if (tfm_tail == 0 || producer_ln || go.gst || !tfm_w.count(p) || !tfm_kv.count(p)) return false;
        if (!tfm_tail_supported(d, cfg.num_heads, T, ctx_len()) || M % 64) return false;
        return tfm_tail > 0 || M >= tfm_tail_min_rows;
    }
    int tfm_tail_min_rows = getenv("MKD_TFM_TAIL_MINROWS") ? atoi(getenv("MKD_TFM_TAIL_MINROWS")) : 4096;
    void emit_tfm_tail(const std::string& p, int d, int M, int T, const bf16_t* a1, const bf16_t* h0, const Tensor& x, bf16_t* out, int ldo, int b0) {
        const bf16_t* wpk = tfm_w.at(p); const float* vec = tfm_v.at(p);
        const bf16_t* kvp = tfm_kv.at(p) + (size_t)b0 * (tfm_tail_kv_bytes(d, 1) / sizeof(bf16_t));
        const bf16_t* xin = x.p; const int ldx = x.ld, Tk = ctx_len();
        push(*cur_plan, [=](hipStream_t st) { return launch_tfm_tail(d, wpk, vec, a1, d, h0, d, xin, ldx, kvp, out, ldo, M, T, Tk, st); },
             1, tfm_tail_flops(d, M, Tk), K_TFM_TAIL, "M=" + std::to_string(M) + " d=" + std::to_string(d) + " T=" + std::to_string(T));
    }
    // SpatialTransformer, depth 1 (App. A.2). x: [B,H,W,d] contiguous or strided. Writes at (out, ldo).
    // b0: first sample of x inside the prepared batch (decoder lanes run on a batch slice; the cross-attention K/V cache is
    // indexed by absolute sample)
    void spatial_transformer(const std::string& p, const Tensor& x, bf16_t* out, int ldo, int b0 = 0, const GnOut& go = GnOut()) {
        const size_t mk = TA().mark();
        const int d = x.C, M = x.rows(), T = x.H * x.W, heads = cfg.num_heads, dh = d / heads;
        const std::string t = p + ".transformer_blocks.0";
        auto buf = [&](int cols) { return (bf16_t*)TA().alloc((size_t)M * cols * sizeof(bf16_t)); };
        const bool fused_tail = use_tfm_tail(p, d, M, T, fuse_ln && gemm_stat_slots(M, d, d) <= 20, go);
        if (fused_tail && tfm_head != 0 && tfm_hw.count(p) && !x.gst) {
            // head as two launches: GroupNorm statistics (full chip), then GroupNorm apply + proj_in + LayerNorm 1 . q|k|v per 64-token tile
            bf16_t* h0f = buf(d); bf16_t* qkvf = buf(3 * d);
            const size_t need = groupnorm_partials_bytes(x.B, T, 32);
            if (need > gn_need) gn_need = need;
            mkd_ctx* self = this;
            const int sid = cur_sid;
            const Tensor xt = x;
            const bf16_t* hwp = tfm_hw.at(p); const float* hvp = tfm_hv.at(p);
            push(*cur_plan, [self, xt, T, d, sid, hwp, hvp, h0f, qkvf, M](hipStream_t st) {
                int nch = 0;
                int rc = launch_gn_stats(xt.p, xt.ld, xt.B, T, d, 32, self->gn_ws[arena_of(sid)], st, &nch);
                if (rc) return rc;
                return launch_tfm_head(d, hwp, hvp, xt.p, xt.ld, self->gn_ws[arena_of(sid)], nch, 1e-6f, h0f, qkvf, M, T, st);
            }, 2, 2.0 * M * 4.0 * d * d, K_TFM_TAIL, "head M=" + std::to_string(M) + " d=" + std::to_string(d) + " T=" + std::to_string(T));
            bf16_t* a1f = buf(d);
            op_attn(qkvf, 3 * d, qkvf + d, 3 * d, qkvf + 2 * d, 3 * d, a1f, d, x.B, T, T, heads, dh);
            emit_tfm_tail(p, d, M, T, a1f, h0f, x, out, ldo, b0);
            TA().release(mk);
            return;
        }
        bf16_t* g = buf(d);
        op_gn(x, wf(p + ".norm.weight"), wf(p + ".norm.bias"), 1e-6f, 0, g, d);
        // Two variants.  fuse_ln: LayerNorm1/2/3 never run as kernels - the GEMM that PRODUCES h0/h1/h2 emits per-column-
        // tile partial row sums of its rounded output and the GEMM that CONSUMES LN(h) applies rstd*(acc - mu*rowsum(W'))
        // in its epilogue.  Default: LayerNorm kernels feeding the same folded... no: the plain weights.
        // Third variant (ln_fly, the default): the consumer takes the row statistics itself (gemm_kernel LN < 0); producers untouched.
        const int slots = gemm_stat_slots(M, d, d);
        const bool fo = fuse_ln && slots <= 20;                  // producer-statistics form, all three norms
        const int fly_rows = ln_fly_rows >= 0 ? ln_fly_rows : (B <= 1 ? 1 << 30 : (B == 2 ? 256 : 128));
        const int fly = getenv("MKD_LN_FLY") ? ln_fly : (M <= fly_rows ? ln_fly_small : ln_fly);      // an explicit MKD_LN_FLY applies to every size
        const bool fy1 = !fuse_ln && (fly & 1), fy2 = !fuse_ln && (fly & 2), fy3 = !fuse_ln && (fly & 4);
        const bool fl = fo;
        auto sbuf = [&]() { return fl ? (float*)TA().alloc((size_t)slots * M * 2 * sizeof(float)) : nullptr; };
        float* st0 = sbuf(); float* st1 = sbuf(); float* st2 = sbuf();
        auto ln_input = [&](const bf16_t* hsrc, const std::string& norm, int ldh = 0, bool fused = false) -> const bf16_t* {      // plain path: LN kernel
            if (fused) return hsrc;
            bf16_t* y = buf(d);
            op_ln(hsrc, wf(t + norm + ".weight"), wf(t + norm + ".bias"), y, M, d, ldh);
            return y;
        };
        const bool mf = !fl && ffm_w.count(p);              // FF2 + proj_out as one GEMM over [gg | h2]
        bf16_t* cat5 = mf ? buf(5 * d) : nullptr;
        bf16_t* h0 = buf(d);
        { Epi e; e.bias = wf(p + ".proj_in.bias"); e.stat_out = st0; op_linear(g, d, M, d, wb(p + ".proj_in.weight"), d, e, h0, d); }
        // self attention
        bf16_t* qkv = buf(3 * d);
        { Epi e; const bool f = fl || fy1; const bf16_t* a_in = ln_input(h0, ".norm1", 0, f);
          if (f) { e.bias = qkv_b.at(p); e.ln_s = qkv_s.at(p); e.stat_in = fl ? st0 : nullptr; e.stat_slots = fl ? slots : 0; }
          op_linear(a_in, d, M, d, f ? qkv_w.at(p) : qkv_plain.at(p), 3 * d, e, qkv, 3 * d); }
        bf16_t* a1 = buf(d);
        op_attn(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, a1, d, x.B, T, T, heads, dh);
        if (fused_tail) {          // everything below as one launch per 64-token tile
            emit_tfm_tail(p, d, M, T, a1, h0, x, out, ldo, b0);
            TA().release(mk);
            return;
        }
        bf16_t* h1 = buf(d);
        { Epi e; e.bias = wf(t + ".attn1.to_out.0.bias"); e.R = h0; e.ldr = d; e.stat_out = st1;
          op_linear(a1, d, M, d, wb(t + ".attn1.to_out.0.weight"), d, e, h1, d); }
        // cross attention (K/V cached at prepare)
        bf16_t* q2 = buf(d);
        { Epi e; const bool f = fl || fy2; const bf16_t* a_in = ln_input(h1, ".norm2", 0, f);
          if (f) { e.bias = q2_b.at(p); e.ln_s = q2_s.at(p); e.stat_in = fl ? st1 : nullptr; e.stat_slots = fl ? slots : 0; }
          op_linear(a_in, d, M, d, f ? q2_w.at(p) : wb(t + ".attn2.to_q.weight"), d, e, q2, d); }
        Tensor kv = kv_cache.at(p);
        kv.p += (size_t)b0 * ctx_len() * kv.ld;
        bf16_t* a2 = buf(d);
        op_attn(q2, d, kv.p, 2 * d, kv.p + d, 2 * d, a2, d, x.B, T, ctx_len(), heads, dh);
        bf16_t* h2 = mf ? cat5 + 4 * d : buf(d);
        const int ldh2 = mf ? 5 * d : d;
        { Epi e; e.bias = wf(t + ".attn2.to_out.0.bias"); e.R = h1; e.ldr = d; e.stat_out = st2;
          op_linear(a2, d, M, d, wb(t + ".attn2.to_out.0.weight"), d, e, h2, ldh2); }
        // GEGLU feed-forward: Linear(d, 8d) + GEGLU in one GEMM (epilogue writes a * gelu(gate), 4d columns)
        bf16_t* gg = mf ? cat5 : buf(4 * d);
        const int ldg = mf ? 5 * d : 4 * d;
        { Epi e; const bool f = fl || fy3; const bf16_t* a_in = ln_input(h2, ".norm3", ldh2, f); e.act = 2;
          if (f) { e.bias = ffg_b.at(p); e.ln_s = ffg_s.at(p); e.stat_in = fl ? st2 : nullptr; e.stat_slots = fl ? slots : 0; }
          else e.bias = ffp_b.at(p);
          op_linear(a_in, f ? ldh2 : d, M, d, f ? ffg_w.at(p) : ffp_w.at(p), 8 * d, e, gg, ldg); }
        if (mf) {
            Epi e; e.bias = ffm_b.at(p); e.R = x.p; e.ldr = x.ld; e.gn = go;
            op_linear(cat5, 5 * d, M, 5 * d, ffm_w.at(p), d, e, out, ldo);
        } else {
            bf16_t* h3 = buf(d);
            { Epi e; e.bias = wf(t + ".ff.net.2.bias"); e.R = h2; e.ldr = d;
              op_linear(gg, 4 * d, M, 4 * d, wb(t + ".ff.net.2.weight"), d, e, h3, d); }
            { Epi e; e.bias = wf(p + ".proj_out.bias"); e.R = x.p; e.ldr = x.ld; e.gn = go;
              op_linear(h3, d, M, d, wb(p + ".proj_out.weight"), d, e, out, ldo); }
        }
        TA().release(mk);
    }

    // time embedding MLP + every ResBlock's emb projection in one GEMM -> fp32 [rows, emb_total] at `proj`.  The three GEMMs never
    // split K (force_splitk = 1): an output element then sees the same sequence of K-steps whatever the tile, so a row of the
    // per-call table (rows = the sampling call's steps, build_temb_table) is bit-identical to the same timestep evaluated per step.
    void emit_time_embedding(int which, int rows, const int64_t* const* tsrc, bf16_t* s0, bf16_t* s1, bf16_t* s2, float* proj) {
        const std::string P = net_prefix(which);
        const int mc = cfg.model_channels, te = temb_dim();
        const bool keep = cur_temb;
        cur_temb = true;
        push(*cur_plan, [tsrc, s0, rows, mc](hipStream_t st) { return launch_timestep_embedding(*tsrc, s0, rows, mc, st); }, 1, 0.0);
        { Epi e; e.bias = wf(P + "time_embed.0.bias"); e.act = 1; op_linear(s0, mc, rows, mc, wb(P + "time_embed.0.weight"), te, e, s1, te, false, 1); }
        // every consumer applies SiLU to emb first (emb_layers = [SiLU, Linear]) -> fold it here
        { Epi e; e.bias = wf(P + "time_embed.2.bias"); e.act = 1; op_linear(s1, te, rows, te, wb(P + "time_embed.2.weight"), te, e, s2, te, false, 1); }
        { Epi e; e.bias = emb_b[which]; op_linear(s2, te, rows, te, emb_w[which], emb_total[which], e, proj, emb_total[which], true, 1); }
        cur_temb = keep;
    }
    // ... of the evaluated batch, from the caller's t (one mkd_eps): buffers in the persistent arena
    float* time_embedding(int which) {
        const int mc = cfg.model_channels, te = temb_dim();
        bf16_t* s0 = (bf16_t*)persist.alloc((size_t)B * mc * sizeof(bf16_t));
        bf16_t* s1 = (bf16_t*)persist.alloc((size_t)B * te * sizeof(bf16_t));
        bf16_t* s2 = (bf16_t*)persist.alloc((size_t)B * te * sizeof(bf16_t));
        float* proj = (float*)persist.alloc((size_t)B * emb_total[which] * sizeof(float));
        emit_time_embedding(which, B, &io_t, s0, s1, s2, proj);
        temb_proj[which] = proj;
        return proj;
    }

    // ---- time embedding of a whole sampling call ---------------------------------------------------------------------------
    // The timesteps of mkd_sample are a host table (reference diffmk/cddim.py:83-95: ts = full((b,), step)), every sample of a step
    // shares one, and the embedding chain (sinusoid -> 2 linear layers -> every ResBlock's emb_layers projection: 75 MB of weights
    // for 8 rows) depends on nothing else.  So one pass computes [n_steps, emb_total] per net before the loop, and a step only copies
    // its row into the [B, emb_total] buffers the ResBlock epilogues read (step_setup_kernel / temb_select_kernel): 8-10 dependent
    // launches leave every step.  MKD_TEMB_TABLE=0: per-step chain as in mkd_eps.
    bool temb_table = getenv("MKD_TEMB_TABLE") ? atoi(getenv("MKD_TEMB_TABLE")) != 0 : true;
    bool temb_skip = false;                       // set while a sampling loop runs from the table: the plan's own chain is left out
    float* temb_proj[2] = {nullptr, nullptr};     // [B, emb_total] per net (persistent arena)
    float* temb_tab[2] = {nullptr, nullptr};      // [steps, emb_total] per net
    int64_t* temb_t = nullptr;                    // device copy of the call's timesteps
    int temb_steps = 0, temb_gen = -1, temb_cap = 0;
    bf16_t* temb_s[2][3] = {};                    // sinusoid and the two hidden layers of the table pass, [temb_cap, .]
    std::vector<void*> temb_owned;
    std::vector<Op> plan_temb_tab;
    void drop_temb_table() {
        for (void* q : temb_owned) hipFree(q);
        temb_owned.clear(); plan_temb_tab.clear();
        temb_tab[0] = temb_tab[1] = nullptr; temb_t = nullptr; temb_steps = 0; temb_gen = -1; temb_cap = 0;
    }
    // buffers grow only (captured step graphs hold the table pointers: they are dropped when the table moves); the launch list is
    // re-emitted whenever the step count or the plan changes
    int build_temb_table(int n_steps) {
        if (temb_gen == plan_generation && temb_steps == n_steps) return 0;
        const int nets = has_control ? 2 : 1;
        const bool realloc = n_steps > temb_cap || !temb_t || !temb_tab[0] || (nets == 2 && !temb_tab[1]);
        const int mc = cfg.model_channels, te = temb_dim();
        if (realloc) {
            MKD_HIP_CHECK(hipDeviceSynchronize());      // (a previous loop may still read the old table)
            drop_temb_table();
            drop_graph();
            auto dalloc = [&](size_t bytes, void** out) -> int {
                MKD_HIP_CHECK(hipMalloc(out, bytes));
                temb_owned.push_back(*out);
                return 0;
            };
            const int cap = std::max(64, n_steps);
            int rc = dalloc((size_t)cap * sizeof(int64_t), (void**)&temb_t);
            for (int which = 0; which < nets && !rc; ++which) {
                rc = dalloc((size_t)cap * mc * sizeof(bf16_t), (void**)&temb_s[which][0]);
                if (!rc) rc = dalloc((size_t)cap * te * sizeof(bf16_t), (void**)&temb_s[which][1]);
                if (!rc) rc = dalloc((size_t)cap * te * sizeof(bf16_t), (void**)&temb_s[which][2]);
                if (!rc) rc = dalloc((size_t)cap * emb_total[which] * sizeof(float), (void**)&temb_tab[which]);
            }
            if (rc) { drop_temb_table(); return rc; }
            temb_cap = cap;
        }
        plan_temb_tab.clear();
        std::vector<Op>* keep_plan = cur_plan; const int keep_sid = cur_sid; const bool keep_count = counting_eps;
        cur_plan = &plan_temb_tab; cur_sid = 0; counting_eps = false;
        for (int which = 0; which < nets; ++which)
            emit_time_embedding(which, n_steps, &temb_t, temb_s[which][0], temb_s[which][1], temb_s[which][2], temb_tab[which]);
        cur_plan = keep_plan; cur_sid = keep_sid; counting_eps = keep_count;
        temb_steps = n_steps; temb_gen = plan_generation;
        return 0;
    }
    TembSel temb_sel() const {
        TembSel ts;
        for (int k = 0; k < 2; ++k) { ts.tab[k] = temb_tab[k]; ts.proj[k] = temb_proj[k]; ts.n[k] = (temb_tab[k] && temb_proj[k]) ? emb_total[k] : 0; }
        ts.batch = B;
        return ts;
    }
    // fills the table for this call's timesteps (host array) on `stream`
    int run_temb_table(int n_steps, const int64_t* timesteps, hipStream_t stream) {
        int rc = build_temb_table(n_steps); if (rc) return rc;
        MKD_HIP_CHECK(hipMemcpyAsync(temb_t, timesteps, (size_t)n_steps * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        for (auto& op : plan_temb_tab) { rc = op.fn(stream); if (rc) return rc; }
        return 0;
    }

    static Tensor slice(Tensor t, int b0, int nb) {
        t.p += (size_t)b0 * t.H * t.W * t.ld;
        if (t.gst) t.gst += (size_t)b0 * 64;
        t.B = nb;
        return t;
    }
    // this tensor will be read by a GroupNorm: give it a statistics buffer that its producers fill
    void want_stats(Tensor& t) {
        t.gst = nullptr;
        if (!gn_fused || t.C % 32 || t.C % 8 || t.C / 8 > 320) return;
        t.gst = (long long*)gstat.alloc((size_t)t.B * 64 * sizeof(long long));
    }
    // producers that cannot emit statistics themselves (direct convolution, copies): a stand-alone pass over what they wrote
    void op_colstats(const bf16_t* x, int ld, int nb, int hw, int ncols, const GnOut& g) {
        if (!g.gst) return;
        push(*cur_plan, [=](hipStream_t st) { return launch_gn_colstats(x, ld, nb, hw, ncols, g.cg, g.coff, g.gst, st); }, 1, 0.0, K_GROUPNORM,
             "colstats B=" + std::to_string(nb) + " HW=" + std::to_string(hw) + " C=" + std::to_string(ncols));
    }

    // whole-batch outputs of one net's encoder: feats[i] = output of input_blocks[i]; mid = middle block output
    void alloc_encoder(std::vector<Tensor>& feats, Tensor& mid) {
        auto enc = encoder_spec();
        int H = h, W = w;
        for (size_t i = 0; i < enc.size(); ++i) {
            if (enc[i].kind == 2) { H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1; }
            feats.push_back(talloc(persist, B, H, W, enc[i].cout));
            // read by a GroupNorm when the next block is a ResBlock (the last one feeds middle_block.0); a Downsample conv reads it raw
            if (i + 1 == enc.size() || enc[i + 1].kind == 1) want_stats(feats.back());
        }
        mid = talloc(persist, B, H, W, enc.back().cout);
    }

    // encoder + middle block of one net over samples [b0, b0 + nb) (every op is per-sample, so a batch slice is a valid lane)
    void encoder_lane(int which, const float* embproj_all, const std::vector<Tensor>& feats, const Tensor& mid, int b0, int nb) {
        const std::string P = net_prefix(which);
        auto enc = encoder_spec();
        const int ld_emb = emb_total[which];
        const float* embproj = embproj_all + (size_t)b0 * ld_emb;
        mkd_ctx* self = this;
        Tensor hcur;
        for (size_t i = 0; i < enc.size(); ++i) {
            const BlockSpec& b = enc[i];
            const std::string p = P + "input_blocks." + std::to_string(i);
            const Tensor o = slice(feats[i], b0, nb);
            if (b.kind == 0) {
                const bf16_t* wgt = wb(p + ".0.weight"); const float* bias = wf(p + ".0.bias");
                const bf16_t* add = which == 1 ? slice(hint_emb, b0, nb).p : nullptr;
                const int hh = h, ww = w, cin = b.cin, cout = b.cout;
                const size_t xoff = (size_t)b0 * cin * hh * ww;
                push(*cur_plan, [self, wgt, bias, o, add, nb, hh, ww, cin, cout, xoff](hipStream_t st) {
                    return launch_conv3x3_direct(self->io_x + xoff, 1, wgt, bias, o.p, 0, 0, add, nb, hh, ww, cin, cout, 1, st);
                }, 1, 2.0 * nb * h * w * b.cout * 9 * b.cin, K_CONV_DIRECT);
                { OpDesc& d = last_desc(); d.type = D_CONV_IN; d.cio = ConvInIo{wgt, bias, o.p, add}; d.xoff = xoff; d.nb = nb; d.hh = hh; d.ww = ww; d.cin = cin; d.cout = cout; }
                op_colstats(o.p, o.ld, o.B, o.H * o.W, o.C, gn_of(o));
            } else if (b.kind == 1) {
                if (b.attn) {
                    const size_t mk = TA().mark();
                    Tensor r = talloc(TA(), hcur.B, hcur.H, hcur.W, b.cout);
                    want_stats(r);
                    resblock(p + ".0", hcur, b.cout, embproj, ld_emb, r.p, r.ld, false, gn_of(r));
                    spatial_transformer(p + ".1", r, o.p, o.ld, b0, gn_of(o));
                    TA().release(mk);
                } else {
                    resblock(p + ".0", hcur, b.cout, embproj, ld_emb, o.p, o.ld, false, gn_of(o));
                }
            } else {
                Epi e; e.bias = wf(p + ".0.op.bias"); e.gn = gn_of(o);
                op_conv(hcur, wb(p + ".0.op.weight"), b.cout, 2, 0, e, o.p, o.ld);
            }
            hcur = o;
        }
        const int ch = hcur.C;
        const size_t mk = TA().mark();
        Tensor m1 = talloc(TA(), hcur.B, hcur.H, hcur.W, ch);
        want_stats(m1);
        resblock(P + "middle_block.0", hcur, ch, embproj, ld_emb, m1.p, m1.ld, false, gn_of(m1));
        Tensor m2 = talloc(TA(), hcur.B, hcur.H, hcur.W, ch);
        want_stats(m2);
        spatial_transformer(P + "middle_block.1", m1, m2.p, m2.ld, b0, gn_of(m2));
        const Tensor m3 = slice(mid, b0, nb);
        resblock(P + "middle_block.2", m2, ch, embproj, ld_emb, m3.p, m3.ld);
        TA().release(mk);
    }

    void build_prepare_plan() {
        cur_plan = &plan_prepare;
        counting_eps = false;
        mkd_ctx* self = this;
        const int L = ctx_len(), cd = cfg.context_dim;
        ctx_bf16 = (bf16_t*)persist.alloc((size_t)B * L * cd * sizeof(bf16_t));
        {
            bf16_t* dst = ctx_bf16; const int64_t n = (int64_t)B * L * cd;
            push(*cur_plan, [self, dst, n](hipStream_t st) { return launch_f32_to_bf16(self->in_context, dst, n, st); }, 1, 0.0);
        }
        kv_cache.clear(); tfm_kv.clear();
        for (int which = 0; which < 2; ++which) {
            if (which == 1 && !has_control) continue;
            for (auto& p : st_prefixes[which]) {
                const int d = (int)params.at(p + ".norm.weight").numel();
                Tensor kv; kv.B = B; kv.H = 1; kv.W = L; kv.C = 2 * d; kv.ld = 2 * d;
                kv.p = (bf16_t*)persist.alloc((size_t)B * L * 2 * d * sizeof(bf16_t));
                Epi e;
                op_linear(ctx_bf16, cd, B * L, cd, kv_w.at(p), 2 * d, e, kv.p, 2 * d);
                kv_cache[p] = kv;
                if (tfm_w.count(p) && tfm_tail != 0 && L <= 80) {
                    bf16_t* kp = (bf16_t*)persist.alloc(tfm_tail_kv_bytes(d, B));
                    const bf16_t* src = kv.p; const int Bn = B;
                    push(*cur_plan, [src, d, Bn, L, kp](hipStream_t st) { return launch_tfm_tail_pack_kv(d, src, 2 * d, Bn, L, kp, st); }, 1, 0.0, K_MISC, "tfm_kv_pack");
                    tfm_kv[p] = kp;
                }
            }
        }
        if (has_control) {
            const std::string P = net_prefix(1);
            const int strides[8] = {1, 1, 2, 1, 2, 1, 2, 1};
            int widths[9];
            widths[0] = cfg.hint_channels;
            for (int j = 0; j < 7; ++j) widths[j + 1] = cfg.hint_widths[j];
            widths[8] = cfg.model_channels;
            const int H0 = 8 * h, W0 = 8 * w;
            // input_hint_block on hint (which == 0) or on the second reference's hint (which == 1, interpolation)
            auto hint_chain = [&](int which2) -> Tensor {
                const size_t mk = TA().mark();
                Tensor cur = talloc(TA(), B, H0, W0, widths[1]);
                {
                    const bf16_t* wgt = wb(P + "input_hint_block.0.weight"); const float* bias = wf(P + "input_hint_block.0.bias");
                    const int Bn = B, cin = widths[0], cout = widths[1];
                    push(*cur_plan, [self, wgt, bias, cur, Bn, H0, W0, cin, cout, which2](hipStream_t st) {
                        return launch_conv3x3_direct(which2 ? self->in_hint2 : self->in_hint, 1, wgt, bias, cur.p, 0, 1, nullptr, Bn, H0,
                                                     W0, cin, cout, 1, st);
                    }, 1, 0.0);
                }
                for (int j = 1; j < 8; ++j) {
                    const int s = strides[j];
                    const int Ho = (cur.H - 1) / s + 1, Wo = (cur.W - 1) / s + 1;
                    Tensor nxt = (j == 7) ? talloc(persist, B, Ho, Wo, widths[j + 1]) : talloc(TA(), B, Ho, Wo, widths[j + 1]);
                    Epi e; e.bias = wf(P + "input_hint_block." + std::to_string(2 * j) + ".bias"); e.act = (j == 7) ? 0 : 1;
                    op_conv(cur, wb(P + "input_hint_block." + std::to_string(2 * j) + ".weight"), widths[j + 1], s, 0, e, nxt.p, nxt.ld);
                    cur = nxt;
                }
                TA().release(mk);
                return cur;
            };
            hint_emb = hint_chain(0);
            if (has_interp) {
                Tensor e2 = hint_chain(1);
                Tensor he = hint_emb; const int Bn = B;
                const int64_t per = (int64_t)he.H * he.W * he.C;
                push(*cur_plan, [self, he, e2, per, Bn](hipStream_t st) {
                    return launch_blend(he.p, e2.p, self->in_alpha, he.p, per, Bn, st);
                }, 1, 0.0);
            }
        }
    }

    void build_eps_plan() {
        cur_plan = &plan_eps;
        counting_eps = true;
        flops_eps = 0; launches_eps = 0;
        mkd_ctx* self = this;
        std::vector<Tensor> cn_feats, hs;
        Tensor cn_mid, u_mid;
        // The ControlNet and the UNet encoder+middle only meet at the zero-conv "combine", and every op is per-sample: the two
        // nets run on different streams, each optionally as two half-batch lanes (UNet lanes on streams 0 / 2, ControlNet lanes
        // on 1 / 3), with the host enqueue order interleaved so all streams are fed from the first launch on.  These kernels are
        // latency-bound (a batch-2 evaluation takes 60 % of a batch-8 one), so concurrency is what pays.
        // enc_group (MKD_ENC_GROUP=1): the two nets become ONE chain of grouped launches instead (group_ops): built to test whether the
        // dispatcher cost of two dependent chains (tools/micro/launch_floor.hip) outweighs what their overlap hides - it does not, see
        // the measurement at enc_group.
        const bool grouped = enc_group && has_control;
        const int EL = (!grouped && enc_lanes && B >= 2) ? 2 : 1;
        float* ep0 = nullptr; float* ep1 = nullptr;
        temb_proj[0] = temb_proj[1] = nullptr;
        cur_plan = &plan_eps;
        cur_sid = 0;
        if (gn_fused)           // the GroupNorm statistics of this evaluation start from zero (one memset node for all of them)
            push(*cur_plan, [self](hipStream_t st) {
                // a kernel, not hipMemsetAsync: as a memset NODE of the captured step graph it costs 1.6 ms per replay (measured)
                return self->gstat.high ? launch_fill_i64((int64_t*)self->gstat_base, 0, (int)(self->gstat.high / 8), st) : 0;
            }, 1, 0.0, K_MISC, "gn_stat_zero");
        if (has_control && !grouped) op_edge(0, 1, true, true);      // side stream starts after everything already enqueued by the caller
        std::vector<Op> lists[NS];
        cur_plan = &lists[0]; cur_sid = 0; ep0 = time_embedding(0);
        if (has_control) { cur_plan = &lists[1]; cur_sid = 1; ep1 = time_embedding(1); }
        cur_plan = &plan_eps; cur_sid = 0;
        if (EL == 2) {          // (the lanes fork after the time embedding of their net)
            for (int k = 0; k < 2; ++k) { for (auto& o : lists[k]) plan_eps.push_back(std::move(o)); lists[k].clear(); }
            op_edge(0, 2, true, true); if (has_control) op_edge(1, 3, true, true);
        }
        alloc_encoder(hs, u_mid);
        if (has_control) alloc_encoder(cn_feats, cn_mid);
        for (int l = 0; l < EL; ++l) {
            const int nb0 = EL == 2 ? B / 2 : B;
            const int b0 = l ? nb0 : 0, nb = l ? B - nb0 : nb0;
            if (has_control) { cur_plan = &lists[2 * l + 1]; cur_sid = 2 * l + 1; encoder_lane(1, ep1, cn_feats, cn_mid, b0, nb); }
            cur_plan = &lists[2 * l]; cur_sid = 2 * l; encoder_lane(0, ep0, hs, u_mid, b0, nb);
        }
        cur_plan = &plan_eps; cur_sid = 0;
        if (grouped) group_ops(lists[0], lists[1], plan_eps);
        else {
            size_t idx[NS] = {};
            for (bool more = true; more;) {
                more = false;
                for (int k : {1, 0, 3, 2})
                    if (idx[k] < lists[k].size()) { plan_eps.push_back(std::move(lists[k][idx[k]++])); more = true; }
            }
            for (int k = 1; k < NS; ++k)
                if (!lists[k].empty()) op_edge(k, 0, true, true);      // join: the decoder needs every encoder lane
        }

        auto dec = decoder_spec();
        const std::string P = net_prefix(0), PC = net_prefix(1);
        // Concat buffers of all decoder blocks: [h (from the previous block / mid) | skip (+ control residual)].
        int n_skip = (int)hs.size();
        std::vector<Tensor> cats(dec.size());
        Tensor final_t;
        {
            int Hc = hs[n_skip - 1].H, Wc = hs[n_skip - 1].W;
            for (size_t i = 0; i < dec.size(); ++i) {
                cats[i] = talloc(persist, B, Hc, Wc, dec[i].cin);
                want_stats(cats[i]);
                if (dec[i].up) { Hc *= 2; Wc *= 2; }
            }
            final_t = talloc(persist, B, Hc, Wc, dec.back().cout);
            want_stats(final_t);
        }
        // One decoder pass over samples [b0, b0 + nb).  helpers_on_side: the zero-conv "combine" GEMMs (skip + scale *
        // zero_conv(cn_feat), written straight into the concat buffer's skip half) and the ResBlocks' 1x1 skip GEMMs only depend
        // on the encoders / the block input, so they run on the side stream under the main GN -> conv -> GN chain.
        // blocks [i0, i1); first_combined: block i0's concat input was already completed by the previous phase
        auto emit_decoder = [&](int b0, int nb, bool helpers_on_side, size_t i0, size_t i1, bool first_combined) {
            const int main_sid = cur_sid;
            const int help_sid = helpers_on_side ? HELPER_BASE + helper_stream : main_sid;
            const int help_cap = helpers_on_side ? main_sid : -1;
            auto combine = [&](size_t i) {          // emits on the CURRENT sid
                const BlockSpec& bs = dec[i];
                const int si = n_skip - 1 - (int)i;
                const Tensor skip = slice(hs[si], b0, nb);
                const int ch_h = bs.cin - skip.C;
                const Tensor cat = slice(cats[i], b0, nb);
                if (i == 0) {
                    const Tensor um = slice(u_mid, b0, nb);
                    if (has_control) {
                        const Tensor cm = slice(cn_mid, b0, nb);
                        Epi e; e.bias = wf(PC + "middle_block_out.0.bias"); e.scale = scales[n_ctrl() - 1]; e.R = um.p; e.ldr = um.ld;
                        e.gn = gn_of(cat, 0);
                        op_linear(cm.p, cm.ld, cm.rows(), cm.C, wb(PC + "middle_block_out.0.weight"), cm.C, e, cat.p, cat.ld);
                    } else {
                        op_copy(um.p, um.ld, cat.p, cat.ld, um.rows(), um.C);
                        op_colstats(cat.p, cat.ld, cat.B, cat.H * cat.W, um.C, gn_of(cat, 0));
                    }
                }
                if (has_control && !only_mid) {
                    const Tensor cf = slice(cn_feats[si], b0, nb);
                    Epi e; e.bias = wf(PC + "zero_convs." + std::to_string(si) + ".0.bias"); e.scale = scales[si]; e.R = skip.p; e.ldr = skip.ld;
                    e.gn = gn_of(cat, ch_h);
                    op_linear(cf.p, cf.ld, cf.rows(), cf.C, wb(PC + "zero_convs." + std::to_string(si) + ".0.weight"), cf.C, e, cat.p + ch_h, cat.ld);
                } else {
                    op_copy(skip.p, skip.ld, cat.p + ch_h, cat.ld, skip.rows(), skip.C);
                    op_colstats(cat.p + ch_h, cat.ld, cat.B, cat.H * cat.W, skip.C, gn_of(cat, ch_h));
                }
            };
            if (helpers_on_side) op_edge(main_sid, helper_stream);     // helper: both encoders are complete (joined on the lane's stream)
            if (!first_combined) { cur_sid = help_sid; cur_cap_sid = help_cap; combine(i0); cur_sid = main_sid; cur_cap_sid = -1; }
            Tensor cat;
            for (size_t i = i0; i < i1; ++i) {
                const BlockSpec& bs = dec[i];
                cat = slice(cats[i], b0, nb);
                if (helpers_on_side) op_edge(helper_stream, main_sid);  // lane: this block's concat input is complete
                if (i + 1 < dec.size()) { cur_sid = help_sid; cur_cap_sid = help_cap; combine(i + 1); cur_sid = main_sid; cur_cap_sid = -1; }   // next block's combine runs under this block
                // where does this block's output go?  next concat buffer's h half (or the final tensor)
                const std::string p = P + "output_blocks." + std::to_string(i);
                const Tensor nxt = slice(i + 1 < dec.size() ? cats[i + 1] : final_t, b0, nb);
                bf16_t* dst = nxt.p; const int dst_ld = nxt.ld;
                const size_t mk = TA().mark();
                const int stages = 1 + (bs.attn ? 1 : 0) + (bs.up ? 1 : 0);
                Tensor cur_in = cat;
                int stage = 0;
                const GnOut go_dst = gn_of(nxt, 0);          // the block's last kernel also accumulates the next GroupNorm's statistics
                {   // ResBlock
                    ++stage;
                    Tensor o;
                    GnOut go;
                    if (stage == stages) { o = nxt; o.C = bs.cout; o.gst = nullptr; go = go_dst; }
                    else {
                        o = talloc(TA(), cat.B, cat.H, cat.W, bs.cout);
                        if (bs.attn) { want_stats(o); go = gn_of(o); }      // read by the transformer's GroupNorm (an Upsample conv reads it raw)
                    }
                    resblock(p + ".0", cur_in, bs.cout, ep0 + (size_t)b0 * emb_total[0], emb_total[0], o.p, o.ld, /*side_skip=*/helpers_on_side, go);
                    cur_in = o;
                }
                if (bs.attn) {
                    ++stage;
                    Tensor o;
                    GnOut go;
                    if (stage == stages) { o = nxt; o.C = bs.cout; o.gst = nullptr; go = go_dst; }
                    else o = talloc(TA(), cat.B, cat.H, cat.W, bs.cout);
                    spatial_transformer(p + ".1", cur_in, o.p, o.ld, b0, go);
                    cur_in = o;
                }
                if (bs.up) {
                    const int k = bs.attn ? 2 : 1;
                    Epi e; e.bias = wf(p + "." + std::to_string(k) + ".conv.bias"); e.gn = go_dst;
                    op_conv(cur_in, wb(p + "." + std::to_string(k) + ".conv.weight"), bs.cout, 1, 1, e, dst, dst_ld);
                }
                TA().release(mk);
            }
            // out: GN32 + SiLU + conv3x3 C -> out_channels (fp32 NCHW)
            if (i1 == dec.size()) {
                const Tensor fin = slice(final_t, b0, nb);
                const size_t mk = TA().mark();
                Tensor g = talloc(TA(), fin.B, fin.H, fin.W, fin.C);
                op_gn(fin, wf(P + "out.0.weight"), wf(P + "out.0.bias"), 1e-5f, 1, g.p, g.ld);
                const bf16_t* wgt = wb(P + "out.2.weight"); const float* bias = wf(P + "out.2.bias");
                const int hh = h, ww = w, cin = fin.C, cout = cfg.out_channels;
                const size_t out_off = (size_t)b0 * cout * hh * ww;
                push(*cur_plan, [self, g, wgt, bias, nb, hh, ww, cin, cout, out_off](hipStream_t st) {
                    return launch_conv3x3_direct(g.p, 0, wgt, bias, self->io_out + out_off, 1, 0, nullptr, nb, hh, ww, cin, cout, 1, st);
                }, 1, 2.0 * nb * h * w * cfg.out_channels * 9 * fin.C, K_CONV_DIRECT);
                TA().release(mk);
            }
        };
        const int DL = (dec_lanes >= 4 && B >= 4) ? 4 : ((dec_lanes >= 2 && B >= 2) ? 2 : 1);
        if (DL > 1) {
            // Optional first phase: the deepest blocks [0, lanes_from) as ONE full-batch chain (they stream weights; a second
            // lane would stream them again), helpers on the side stream; then batch-slice lanes.
            const size_t from = (size_t)std::min<int>(std::max(dec_lanes_from, 0), (int)dec.size() - 1);
            if (from > 0) {
                emit_decoder(0, B, dec_overlap, 0, from, false);
                if (dec_overlap) op_edge(1, 0);     // the helper's combine of block `from` (lane 1 shares the helper's stream)
            }
            // Batch-slice lanes: every op is per-sample, so lane l decodes its share of the samples on stream l, host enqueue
            // interleaved.  Also inside a hipGraph capture (one fork / join edge per extra lane).
            std::vector<Op> lane_ops[NS];
            for (int l = 0; l < DL; ++l) {
                const int b0 = (int)((int64_t)B * l / DL), b1 = (int)((int64_t)B * (l + 1) / DL);
                cur_plan = &lane_ops[l]; cur_sid = l;
                const bool lh = lane_helpers && DL == 2;          // lane l's helper GEMMs on stream 2 + l
                lane_main = l; helper_stream = lh ? 2 + l : 1;
                emit_decoder(b0, b1 - b0, lh, from, dec.size(), from > 0);
                lane_main = 0; helper_stream = 1;
            }
            cur_plan = &plan_eps; cur_sid = 0;
            for (int l = 1; l < DL; ++l) op_edge(0, l, true, true);     // lanes start after the encoders (all joined on main)
            size_t idx[NS] = {};
            for (bool more = true; more;) {
                more = false;
                for (int l = 0; l < DL; ++l)
                    if (idx[l] < lane_ops[l].size()) { plan_eps.push_back(std::move(lane_ops[l][idx[l]++])); more = true; }
            }
            for (int l = 1; l < DL; ++l) op_edge(l, 0, true, true);     // the evaluation ends when every lane has
        } else {
            emit_decoder(0, B, dec_overlap, 0, dec.size(), false);
        }
        counting_eps = false;
        if (!dry) {          // (grouping changed the launch count: take it from the plan itself)
            launches_eps = 0; launches_temb = 0; flops_eps = 0;
            for (auto& op : plan_eps) { launches_eps += op.launches; flops_eps += op.flops; if (op.temb) launches_temb += op.launches; }
        }
    }

    // ---- grouped launches of the encoder phase ----------------------------------------------------------------------------------
    // MKD_ENC_GROUP=1: ControlNet and UNet encoder + middle block as one chain of 2-problem launches on the caller's stream; 0 (the
    // default): two chains on two streams.  Measured at batch 8, 256x256, graph replay, latents only (tools/exp_r3_group.sh, two
    // alternating rounds on one box): two chains 5.64 ms per evaluation (728 launches), grouped 6.19-6.25 ms (576-582 launches) -
    // 146 launches fewer and 10 % SLOWER.  A grouped kernel takes twice the time of one of its halves (the serial sum of kernel
    // times barely moves: 10.35 -> 9.4 ms with ~4 us of event overhead per launch in both), so what grouping removes is one
    // fixed cost per pair - and that is exactly what the second chain already hides: while one chain's kernel drains, dispatches and
    // fills its first tiles, the other chain's kernel has the CUs (pair = max(L + W, 2 W) on two chains, L + 2 W grouped).
    bool enc_group = getenv("MKD_ENC_GROUP") ? atoi(getenv("MKD_ENC_GROUP")) != 0 : false;
    int launches_temb = 0;
    // u / c: the op lists of the UNet and the ControlNet encoder (same architecture, emitted by the same code).  Ops whose
    // descriptors agree in kind and geometry become ONE launch over both problems; anything else runs one after the other.  Either
    // way everything lands on stream 0: the ControlNet keeps its own temporaries and workspaces (arena 1), only its stream goes.
    void group_ops(std::vector<Op>& u, std::vector<Op>& c, std::vector<Op>& out) {
        auto single = [&](Op& o) { o.sid = 0; o.cap_sid = -1; out.push_back(std::move(o)); };
        if (u.size() != c.size()) {          // (not the same op sequence after all: no pairing)
            for (auto& o : u) single(o);
            for (auto& o : c) single(o);
            return;
        }
        for (size_t i = 0; i < u.size(); ++i) {
            Op g;
            if (make_group(u[i], c[i], &g)) out.push_back(std::move(g));
            else { single(u[i]); single(c[i]); }
        }
    }
    bool make_group(const Op& a, const Op& b, Op* out) {
        const OpDesc& x = a.d; const OpDesc& y = b.d;
        if (x.type == D_NONE || x.type != y.type || a.kind != b.kind || a.launches != b.launches || a.temb != b.temb) return false;
        if (arena_of(x.sid) == arena_of(y.sid) && (x.type == D_GEMM || x.type == D_GN_SLAB)) return false;      // (split-K slabs per problem)
        mkd_ctx* self = this;
        OpFn fn;
        switch (x.type) {
            case D_GEMM: {
                if (!gemm_same_geometry(x.gemm, y.gemm)) return false;
                const GemmArgs ga = x.gemm, gb = y.gemm; const int sa = x.sid, sb = y.sid;
                fn = [self, ga, gb, sa, sb](hipStream_t st) {
                    GemmArgs p = ga, q = gb;
                    p.ws = self->splitk_ws[arena_of(sa)]; p.ws_bytes = self->splitk_ws_bytes[arena_of(sa)];
                    q.ws = self->splitk_ws[arena_of(sb)]; q.ws_bytes = self->splitk_ws_bytes[arena_of(sb)];
                    return launch_gemm(p, st, &q);
                };
                break;
            }
            case D_GN_SLAB: {
                if (!gemm_same_geometry(x.gemm, y.gemm) || x.sk != y.sk || x.raw != y.raw || x.eps != y.eps || x.silu != y.silu || x.ld_out != y.ld_out ||
                    x.nb != y.nb || x.hw != y.hw) return false;
                const OpDesc dx = x, dy = y;
                fn = [self, dx, dy](hipStream_t st) {
                    GemmArgs p = dx.gemm, q = dy.gemm;
                    p.ws = self->splitk_ws[arena_of(dx.sid)]; p.splitk = dx.sk; q.ws = self->splitk_ws[arena_of(dy.sid)]; q.splitk = dy.sk;
                    if (!dx.raw) { p.C = nullptr; q.C = nullptr; }
                    return launch_gn_from_slabs(p, dx.nio.gamma, dx.nio.beta, dx.eps, dx.silu, dx.nio.y, dx.ld_out, dx.nb, dx.hw, st, &q, &dy.nio);
                };
                break;
            }
            case D_GN: {
                if (x.eps != y.eps || x.silu != y.silu || x.ld_in != y.ld_in || x.ld_out != y.ld_out || x.nb != y.nb || x.hw != y.hw || x.C != y.C) return false;
                const OpDesc dx = x, dy = y;
                fn = [self, dx, dy](hipStream_t st) {
                    return launch_groupnorm(dx.nio.x, dx.ld_in, dx.nio.gamma, dx.nio.beta, dx.eps, dx.silu, dx.nio.y, dx.ld_out, dx.nb, dx.hw, dx.C, 32,
                                            self->gn_ws[arena_of(dx.sid)], st, &dy.nio, self->gn_2k_min_hw);
                };
                break;
            }
            case D_LN: {
                if (x.eps != y.eps || x.nb != y.nb || x.C != y.C || x.ld_in != y.ld_in) return false;
                const OpDesc dx = x, dy = y;
                fn = [dx, dy](hipStream_t st) { return launch_layernorm(dx.nio.x, dx.nio.gamma, dx.nio.beta, dx.eps, dx.nio.y, dx.nb, dx.C, st, dx.ld_in, &dy.nio); };
                break;
            }
            case D_ATTN: {
                if (x.ldq != y.ldq || x.ldk != y.ldk || x.ldv != y.ldv || x.ldo != y.ldo || x.nb != y.nb || x.Tq != y.Tq || x.Tk != y.Tk || x.heads != y.heads ||
                    x.dh != y.dh) return false;
                const OpDesc dx = x, dy = y;
                fn = [dx, dy](hipStream_t st) {
                    return launch_attention(dx.aio.q, dx.ldq, dx.aio.k, dx.ldk, dx.aio.v, dx.ldv, dx.aio.o, dx.ldo, dx.nb, dx.Tq, dx.Tk, dx.heads, dx.dh,
                                            1.0f / sqrtf((float)dx.dh), st, 0, &dy.aio);
                };
                break;
            }
            case D_CONV_IN: {
                if (x.xoff != y.xoff || x.nb != y.nb || x.hh != y.hh || x.ww != y.ww || x.cin != y.cin || x.cout != y.cout || x.cin != 4 || x.cout < 64 || x.cout > 512)
                    return false;
                const OpDesc dx = x, dy = y;
                fn = [self, dx, dy](hipStream_t st) {
                    return launch_conv3x3_direct(self->io_x + dx.xoff, 1, dx.cio.w, dx.cio.bias, dx.cio.y, 0, 0, dx.cio.add, dx.nb, dx.hh, dx.ww, dx.cin, dx.cout, 1, st,
                                                 &dy.cio);
                };
                break;
            }
            default: return false;
        }
        Op g;
        g.fn = std::move(fn); g.kind = a.kind; g.flops = a.flops + b.flops; g.bytes = a.bytes + b.bytes; g.launches = a.launches; g.label = a.label + " x2"; g.sid = 0; g.cap_sid = -1;
        g.temb = a.temb;
        *out = std::move(g);
        return true;
    }

    int ensure(void** p, size_t* have, size_t need) {
        if (*have >= need && *p) return 0;
        if (*p) { hipFree(*p); *p = nullptr; *have = 0; }
        MKD_HIP_CHECK(hipMalloc(p, need ? need : 256));
        *have = need;
        return 0;
    }

    int prepare(int batch, int hh, int ww, const float* hint, const float* context, const float* control_scales,
                int only_mid_control, hipStream_t stream, const float* hint2 = nullptr, const float* alpha = nullptr) {
        if (!finalized) return mkd_fail(MKD_ERR_STATE, "mkd_prepare before mkd_weights_finalize");
        if (batch <= 0 || hh <= 0 || ww <= 0 || !context) return mkd_fail(MKD_ERR_ARG, "mkd_prepare: bad arguments");
        const int down = 1 << (cfg.n_levels - 1);
        if (hh % down || ww % down) return mkd_fail(MKD_ERR_ARG, "mkd_prepare: latent h, w must be multiples of " + std::to_string(down));
        const bool ctrl = hint != nullptr;
        const bool interp = hint2 != nullptr;
        if (interp && (!ctrl || !alpha)) return mkd_fail(MKD_ERR_ARG, "mkd_prepare_interp: needs hint, hint2 and alpha");
        const bool same = prepared && batch == B && hh == h && ww == w && ctrl == has_control && (only_mid_control != 0) == only_mid &&
                          interp == has_interp && plan_epoch == gemm_plan_epoch() && opt_epoch_planned == opt_epoch;
        bool same_scales = same;
        for (int i = 0; i < n_ctrl() && same_scales; ++i) same_scales = scales[i] == (control_scales ? control_scales[i] : 1.f);
        in_hint = hint; in_context = context; in_hint2 = hint2; in_alpha = alpha;
        if (!same_scales) {
            plan_epoch = gemm_plan_epoch(); opt_epoch_planned = opt_epoch;
            B = batch; h = hh; w = ww; has_control = ctrl; only_mid = only_mid_control != 0; has_interp = interp;
            for (int i = 0; i < n_ctrl(); ++i) scales[i] = control_scales ? control_scales[i] : 1.f;
            prepared = false;
            // pass 1: dry run to size the arenas (pointers are offsets from null and never dereferenced)
            dry = true;
            persist.base = nullptr; persist.reset();
            gstat.base = nullptr; gstat.reset();
            for (int i = 0; i < NS; ++i) { temp_arena[i].base = nullptr; temp_arena[i].reset(); }
            splitk_need = 0; gn_need = 0;
            plan_prepare.clear(); plan_eps.clear();
            cur_sid = 0;
            build_prepare_plan(); build_eps_plan();
            int rc = ensure((void**)&persist_base, &persist_cap, persist.high + 256); if (rc) return rc;
            rc = ensure((void**)&gstat_base, &gstat_cap, gstat.high + 256); if (rc) return rc;
            for (int i = 0; i < NS; ++i) {
                rc = ensure((void**)&temp_base[i], &temp_cap[i], temp_arena[i].high + 256); if (rc) return rc;
                rc = ensure((void**)&splitk_ws[i], &splitk_ws_bytes[i], splitk_need); if (rc) return rc;
                rc = ensure((void**)&gn_ws[i], &gn_ws_bytes[i], gn_need); if (rc) return rc;
            }
            for (int i = 1; i < NS; ++i)
                if (!side_streams[i]) MKD_HIP_CHECK(hipStreamCreateWithFlags(&side_streams[i], hipStreamNonBlocking));
            // pass 2: real plan
            dry = false;
            persist.base = persist_base; persist.reset();
            gstat.base = gstat_base; gstat.reset();
            for (int i = 0; i < NS; ++i) { temp_arena[i].base = temp_base[i]; temp_arena[i].reset(); }
            cur_sid = 0; aux_used = 0;
            build_prepare_plan();
            persist_eps_begin = persist.off;           // everything above this is written by mkd_eps itself
            build_eps_plan();
            ++plan_generation; drop_graph();
            // sampler buffers
            const size_t lat = (size_t)B * cfg.in_channels * h * w * sizeof(float);
            for (float** q : {&s_xa, &s_xb, &s_xin, &s_eps}) {
                if (*q) { hipFree(*q); *q = nullptr; }
                MKD_HIP_CHECK(hipMalloc((void**)q, lat));
            }
            if (s_t) { hipFree(s_t); s_t = nullptr; }
            MKD_HIP_CHECK(hipMalloc((void**)&s_t, (size_t)B * sizeof(int64_t)));
        }
        for (auto& op : plan_prepare) { int rc = op.fn(stream); if (rc) return rc; }
        prepared = true;
        return 0;
    }

    int eps(const float* x, const int64_t* t, float* out, hipStream_t stream) {
        if (!prepared) return mkd_fail(MKD_ERR_STATE, "mkd_eps before mkd_prepare");
        if (!x || !t || !out) return mkd_fail(MKD_ERR_ARG, "mkd_eps: null pointer");
        io_x = x; io_t = t; io_out = out;
        run_main = stream; run_serial = !dual_stream;
        for (auto& op : plan_eps) {
            if (op.temb && temb_skip) continue;          // (mkd_sample filled the rows from its per-call table)
            const int sid = (capturing && op.cap_sid >= 0) ? op.cap_sid : op.sid;
#ifdef MKD_EXP_ABLATE
            // experiment build only (tools/exp_ablate.sh): leave a whole kernel class out of the evaluation (WRONG results) to bound what
            // any optimisation of that class could gain inside the concurrent loop.  1 GroupNorm, 2 LayerNorm, 4 attention
            static const int skip = getenv("MKD_EXP_SKIP") ? atoi(getenv("MKD_EXP_SKIP")) : 0;
            if (op.launches > 0 && (((skip & 1) && op.kind == K_GROUPNORM) || ((skip & 2) && op.kind == K_LAYERNORM) ||
                                    ((skip & 4) && op.kind == K_ATTENTION))) continue;
            // MKD_EXP_EMPTY: same classes, but the launch stays and only its work goes (a one-element fill kernel in its place):
            // separates what a class costs as launches from what it costs as work
            static const int empty = getenv("MKD_EXP_EMPTY") ? atoi(getenv("MKD_EXP_EMPTY")) : 0;
            if (op.launches > 0 && (((empty & 1) && op.kind == K_GROUPNORM) || ((empty & 2) && op.kind == K_LAYERNORM) ||
                                    ((empty & 4) && op.kind == K_ATTENTION) || ((empty & 16) && op.kind < K_GROUPNORM))) {
                static int64_t* dummy = nullptr;
                if (!dummy) MKD_HIP_CHECK(hipMalloc((void**)&dummy, 256));
                int rc = launch_fill_i64(dummy, 0, 1, stream_of(sid));
                if (rc) return rc;
                continue;
            }
#endif
            int rc = op.fn(stream_of(sid));
            if (rc) return rc;
        }
        return 0;
    }

    // ---- fine-grained cross-stream edges (decoder): event e recorded on stream `from`, awaited by stream `to` ----
    int ev_new() {
        if (dry) return -1;
        if (aux_used == (int)aux_ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
            aux_ev.push_back(e);
        }
        return aux_used++;
    }
    // in_graph: also taken while capturing a hipGraph (the two decoder lanes: 2 edges per evaluation); the per-block helper
    // edges are not (graph-side cross-stream edges cost more than they hide)
    void op_edge(int from_sid, int to_sid, bool in_graph = false, bool always = false) {   // `to` waits for everything enqueued so far on `from`
        if (!always && !dec_overlap && !dec_lanes) return;
        const int e = ev_new();
        mkd_ctx* self = this;
        push(*cur_plan, [self, e, from_sid, to_sid, in_graph](hipStream_t) {
            if (self->run_serial || (self->capturing && !in_graph) || e < 0) return 0;      // graph nodes pay for every cross-stream edge: keep the decoder linear there
            MKD_HIP_CHECK(hipEventRecord(self->aux_ev[e], self->stream_of(from_sid)));
            MKD_HIP_CHECK(hipStreamWaitEvent(self->stream_of(to_sid), self->aux_ev[e], 0));
            return 0;
        }, 0, 0.0, K_MISC, "edge " + std::to_string(from_sid) + "->" + std::to_string(to_sid));
        if (!dry) { Op& o = cur_plan->back(); o.edge_from = from_sid; o.edge_to = to_sid; o.edge_in_graph = in_graph; }
    }

    // one eps with a hipEvent pair around every plan op: per-kernel-class device time (bench roofline)
    // bytes (may be null): algorithmic HBM bytes per class.  ms_b2b (may be null): the launches of each class replayed BACK TO BACK
    // between one event pair (3 rounds, averaged) - per-launch durations without the 3-5 us that an event pair adds to every launch
    // it brackets in the per-op pass; inputs are whatever the full pass left in the buffers.
    int eps_profile(const float* x, const int64_t* t, float* out, hipStream_t stream, double* ms, double* flops, int* launches,
                    const char* csv_path = nullptr, double* bytes = nullptr, double* ms_b2b = nullptr) {
        if (!prepared) return mkd_fail(MKD_ERR_STATE, "mkd_eps_profile before mkd_prepare");
        io_x = x; io_t = t; io_out = out;
        run_main = stream; run_serial = true;          // profile on ONE stream: per-launch times are not overlapped
        const size_t n = plan_eps.size();
        std::vector<hipEvent_t> ev(n + 1);
        for (auto& e : ev) MKD_HIP_CHECK(hipEventCreate(&e));
        int rc = 0;
        MKD_HIP_CHECK(hipEventRecord(ev[0], stream));
        for (size_t i = 0; i < n && !rc; ++i) {
            rc = plan_eps[i].fn(stream);
            if (!rc && hipEventRecord(ev[i + 1], stream) != hipSuccess) rc = mkd_fail(MKD_ERR_HIP, "hipEventRecord");
        }
        if (!rc && hipStreamSynchronize(stream) != hipSuccess) rc = mkd_fail(MKD_ERR_HIP, "hipStreamSynchronize");
        for (int k = 0; k < K_COUNT; ++k) { ms[k] = 0; flops[k] = 0; launches[k] = 0; }
        FILE* csv = (csv_path && !rc) ? fopen(csv_path, "w") : nullptr;
        if (csv) fprintf(csv, "op,kind,label,ms,gflop,stream\n");
        for (size_t i = 0; i < n && !rc; ++i) {
            float dt = 0.f;
            if (hipEventElapsedTime(&dt, ev[i], ev[i + 1]) != hipSuccess) { rc = mkd_fail(MKD_ERR_HIP, "hipEventElapsedTime"); break; }
            const Op& op = plan_eps[i];
            ms[op.kind] += dt; flops[op.kind] += op.flops; launches[op.kind] += op.launches;
            if (csv) fprintf(csv, "%zu,%s,%s,%.5f,%.4f,%d\n", i, kind_name(op.kind).c_str(), op.label.c_str(), dt, op.flops / 1e9, op.sid);
        }
        if (csv) fclose(csv);
        if (bytes) {
            for (int k = 0; k < K_COUNT; ++k) bytes[k] = 0;
            for (auto& op : plan_eps) bytes[op.kind] += op.bytes;
        }
        if (ms_b2b && !rc) {
            constexpr int ROUNDS = 3;
            for (int k = 0; k < K_COUNT && !rc; ++k) {
                ms_b2b[k] = 0;
                if (!launches[k]) continue;
                for (int r = 0; r <= ROUNDS && !rc; ++r) {          // round 0 warms up
                    if (hipEventRecord(ev[0], stream) != hipSuccess) rc = mkd_fail(MKD_ERR_HIP, "hipEventRecord");
                    for (size_t i = 0; i < n && !rc; ++i)
                        if (plan_eps[i].kind == k && plan_eps[i].launches > 0) rc = plan_eps[i].fn(stream);
                    if (!rc && (hipEventRecord(ev[1], stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) rc = mkd_fail(MKD_ERR_HIP, "class replay");
                    float dt = 0.f;
                    if (!rc && hipEventElapsedTime(&dt, ev[0], ev[1]) != hipSuccess) rc = mkd_fail(MKD_ERR_HIP, "hipEventElapsedTime");
                    if (r > 0) ms_b2b[k] += dt / ROUNDS;
                }
            }
        }
        for (auto& e : ev) hipEventDestroy(e);
        return rc;
    }

    void drop_segments() {
        for (auto& sg : segs) if (sg.graph) hipGraphExecDestroy(sg.graph);
        segs.clear(); seg_actions.clear(); seg_gen = -1;
    }
    // One reverse step as per-stream linear graphs (graph mode 2).  Walks the step's launches in plan order; every cross-stream edge
    // that the single-graph capture keeps (in_graph) closes the pending stretch of both its streams.
    int build_segments(int batch, bool cfg_on, float cfg_scale) {
        drop_segments();
        const int64_t n = (int64_t)batch * cfg.in_channels * h * w;
        mkd_ctx* self = this;
        struct Item { OpFn fn; int sid; int from, to; };
        std::vector<Item> items;
        const int Bn = B;
        items.push_back({[self, Bn](hipStream_t st) {
            const TembSel ts = self->temb_sel();
            return launch_step_setup(self->s_state, self->s_t, Bn, st, self->temb_skip ? &ts : nullptr); }, 0, -1, -1});
        const float* ec; const float* eu = nullptr;
        if (cfg_on) {
            items.push_back({[self, n](hipStream_t st) { return launch_repeat_batch(self->s_xa, self->s_xin, n, 2, st); }, 0, -1, -1});
            io_x = s_xin; eu = s_eps; ec = s_eps + n;
        } else { io_x = s_xa; ec = s_eps; }
        io_t = s_t; io_out = s_eps;
        for (auto& op : plan_eps) {
            if (op.edge_from >= 0) { if (op.edge_in_graph) items.push_back({nullptr, 0, arena_of(op.edge_from), arena_of(op.edge_to)}); continue; }
            if (op.launches == 0 && op.kind == K_MISC && op.label.rfind("edge", 0) == 0) continue;
            if (op.temb && temb_skip) continue;
#ifdef MKD_EXP_ABLATE
            {   // experiment build only: the same class switches as in eps()
                static const int skip = getenv("MKD_EXP_SKIP") ? atoi(getenv("MKD_EXP_SKIP")) : 0;
                static const int empty = getenv("MKD_EXP_EMPTY") ? atoi(getenv("MKD_EXP_EMPTY")) : 0;
                auto hit = [&](int m) { return op.launches > 0 && (((m & 1) && op.kind == K_GROUPNORM) || ((m & 2) && op.kind == K_LAYERNORM) ||
                                                                    ((m & 4) && op.kind == K_ATTENTION) || ((m & 16) && op.kind < K_GROUPNORM)); };
                if (hit(skip & 7)) continue;
                if (hit(empty)) {
                    static int64_t* dummy = nullptr;
                    if (!dummy) MKD_HIP_CHECK(hipMalloc((void**)&dummy, 256));
                    int64_t* dd = dummy;
                    items.push_back({[dd](hipStream_t st) { return launch_fill_i64(dd, 0, 1, st); }, arena_of(op.cap_sid >= 0 ? op.cap_sid : op.sid), -1, -1});
                    continue;
                }
            }
#endif
            items.push_back({op.fn, arena_of(op.cap_sid >= 0 ? op.cap_sid : op.sid), -1, -1});
        }
        items.push_back({[self, ec, eu, cfg_scale, n](hipStream_t st) { return launch_ddim_step_state(self->s_xa, ec, eu, cfg_scale, self->s_state, n, st); }, 0, -1, -1});
        std::vector<OpFn> pend[NS];
        int rc = 0;
        auto flush = [&](int sid) -> int {
            if (pend[sid].empty()) return 0;
            Segment sg;
            if (pend[sid].size() <= 2) sg.eager = pend[sid];
            else {
                hipStream_t st = stream_of(sid);
                hipGraph_t g = nullptr;
                MKD_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                int r = 0;
                for (auto& f : pend[sid]) { r = f(st); if (r) break; }
                hipError_t e = hipStreamEndCapture(st, &g);
                if (r) { if (g) hipGraphDestroy(g); return r; }
                if (e != hipSuccess) return mkd_fail(MKD_ERR_HIP, std::string("hipStreamEndCapture (segment): ") + hipGetErrorString(e));
                e = hipGraphInstantiate(&sg.graph, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (e != hipSuccess) return mkd_fail(MKD_ERR_HIP, std::string("hipGraphInstantiate (segment): ") + hipGetErrorString(e));
            }
            segs.push_back(std::move(sg));
            seg_actions.push_back({0, sid, (int)segs.size() - 1});
            pend[sid].clear();
            return 0;
        };
        int n_ev = 0;
        capturing = true;          // (ops that look at the flag behave as in the single-graph capture)
        for (auto& it : items) {
            if (it.from >= 0) {
                if (it.from == it.to) continue;
                if ((rc = flush(it.from))) break;
                if (n_ev == (int)seg_events.size()) {
                    hipEvent_t e = nullptr;
                    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = mkd_fail(MKD_ERR_HIP, "hipEventCreate (segment)"); break; }
                    seg_events.push_back(e);
                }
                seg_actions.push_back({1, it.from, n_ev});
                if ((rc = flush(it.to))) break;
                seg_actions.push_back({2, it.to, n_ev});
                ++n_ev;
            } else pend[it.sid].push_back(it.fn);
        }
        for (int sid = NS - 1; sid >= 0 && !rc; --sid) rc = flush(sid);      // (every side stream was joined by an edge: only stream 0 has work left)
        capturing = false;
        if (rc) { drop_segments(); return rc; }
        seg_gen = plan_generation; seg_cfg = (int)cfg_on; seg_scale = cfg_scale; seg_temb = (int)temb_skip; seg_batch = batch;
        return 0;
    }
    int run_segments() {
        for (auto& a : seg_actions) {
            hipStream_t st = stream_of(a.sid);
            if (a.type == 0) {
                Segment& sg = segs[a.idx];
                if (sg.graph) MKD_HIP_CHECK(hipGraphLaunch(sg.graph, st));
                else for (auto& f : sg.eager) { int rc = f(st); if (rc) return rc; }
            } else if (a.type == 1) MKD_HIP_CHECK(hipEventRecord(seg_events[a.idx], st));
            else MKD_HIP_CHECK(hipStreamWaitEvent(st, seg_events[a.idx], 0));
        }
        return 0;
    }

    void drop_graph() {
        drop_segments();
        if (step_graph) { hipGraphExecDestroy(step_graph); step_graph = nullptr; }
        if (multi_graph) { hipGraphExecDestroy(multi_graph); multi_graph = nullptr; multi_graph_steps = 0; }
        step_graph_gen = -1;
    }

    // enqueue ONE reverse step that reads its timestep / coefficients from the device step state (graph body)
    int enqueue_state_step(int batch, bool cfg_on, float cfg_scale, hipStream_t stream) {
        const int64_t n = (int64_t)batch * cfg.in_channels * h * w;
        const TembSel ts = temb_sel();
        int rc = launch_step_setup(s_state, s_t, B, stream, temb_skip ? &ts : nullptr); if (rc) return rc;
        const float* ec; const float* eu = nullptr;
        if (cfg_on) {
            rc = launch_repeat_batch(s_xa, s_xin, n, 2, stream); if (rc) return rc;
            rc = eps(s_xin, s_t, s_eps, stream); if (rc) return rc;
            eu = s_eps; ec = s_eps + n;
        } else {
            rc = eps(s_xa, s_t, s_eps, stream); if (rc) return rc;
            ec = s_eps;
        }
        return launch_ddim_step_state(s_xa, ec, eu, cfg_scale, s_state, n, stream);
    }

    int sample(const float* x_T, int batch, int n_steps, const int64_t* timesteps, const float* alphas,
               const float* alphas_prev, const float* s1m, float cfg_scale, float* x_out, int use_graph, hipStream_t stream,
               const float* sigmas = nullptr, const float* noise = nullptr, float temperature = 1.0f) {
        const int rc = sample_impl(x_T, batch, n_steps, timesteps, alphas, alphas_prev, s1m, cfg_scale, x_out, use_graph, stream, sigmas, noise, temperature);
        temb_skip = false;          // (a later mkd_eps runs its own time-embedding chain)
        return rc;
    }
    int sample_impl(const float* x_T, int batch, int n_steps, const int64_t* timesteps, const float* alphas,
               const float* alphas_prev, const float* s1m, float cfg_scale, float* x_out, int use_graph, hipStream_t stream,
               const float* sigmas, const float* noise, float temperature) {
        if (!prepared) return mkd_fail(MKD_ERR_STATE, "mkd_sample before mkd_prepare");
        bool stochastic = false;
        if (sigmas) for (int i = 0; i < n_steps; ++i) {
            if (!(sigmas[i] >= 0.f) || 1.0f - alphas_prev[i] - sigmas[i] * sigmas[i] < 0.f) return mkd_fail(MKD_ERR_ARG, "mkd_sample_eta: sigma out of range");
            stochastic = stochastic || sigmas[i] != 0.f;
        }
        if (stochastic && !noise) return mkd_fail(MKD_ERR_ARG, "mkd_sample_eta: sigma > 0 needs the noise draws");
        const bool cfg_on = cfg_scale != 1.0f;
        if (cfg_on ? (B != 2 * batch) : (B != batch))
            return mkd_fail(MKD_ERR_ARG, "mkd_sample: prepared batch must be B (cfg_scale == 1) or 2B (uncond first)");
        if (n_steps <= 0 || !timesteps || !alphas || !alphas_prev || !s1m || !x_T || !x_out)
            return mkd_fail(MKD_ERR_ARG, "mkd_sample: bad arguments");
        const int64_t n = (int64_t)batch * cfg.in_channels * h * w;
        MKD_HIP_CHECK(hipMemcpyAsync(s_xa, x_T, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        if (use_graph) {
            // --- hipGraph path: one captured step (both streams, fork/join included), replayed n_steps times ---
            if (n_steps > MKD_MAX_STEPS) return mkd_fail(MKD_ERR_ARG, "mkd_sample: too many steps for the graph path");
            if (!h_state) MKD_HIP_CHECK(hipHostMalloc((void**)&h_state, sizeof(StepState)));
            if (!s_state) MKD_HIP_CHECK(hipMalloc((void**)&s_state, sizeof(StepState)));
            if (!loop_stream) {
                // the caller's stream may be the legacy null stream, which cannot be captured: run the loop on a
                // private stream ordered against the caller's with events
                MKD_HIP_CHECK(hipStreamCreateWithFlags(&loop_stream, hipStreamNonBlocking));
                MKD_HIP_CHECK(hipEventCreateWithFlags(&ev_loop_in, hipEventDisableTiming));
                MKD_HIP_CHECK(hipEventCreateWithFlags(&ev_loop_out, hipEventDisableTiming));
            }
            MKD_HIP_CHECK(hipStreamSynchronize(loop_stream));     // h_state may still feed a previous call's copy
            h_state->counter = n_steps - 1;
            for (int i = 0; i < n_steps; ++i) {
                h_state->timesteps[i] = timesteps[i];
                h_state->coef[4 * i + 0] = 1.0f / sqrtf(alphas[i]);
                h_state->coef[4 * i + 1] = sqrtf(alphas_prev[i]);
                const float sg = stochastic ? sigmas[i] : 0.f;
                h_state->coef[4 * i + 2] = sqrtf(1.0f - alphas_prev[i] - sg * sg);
                h_state->coef[4 * i + 3] = s1m[i];
                h_state->sigma[i] = sg;
            }
            h_state->noise = stochastic ? noise : nullptr; h_state->temperature = temperature; h_state->n_steps = n_steps;
            h_state->cur_sigma = 0.f; h_state->cur_row = 0;
            MKD_HIP_CHECK(hipEventRecord(ev_loop_in, stream));
            MKD_HIP_CHECK(hipStreamWaitEvent(loop_stream, ev_loop_in, 0));
            MKD_HIP_CHECK(hipMemcpyAsync(s_state, h_state, sizeof(StepState), hipMemcpyHostToDevice, loop_stream));
            if (temb_table) {
                int rc = run_temb_table(n_steps, timesteps, loop_stream); if (rc) return rc;
                temb_skip = true;
            }
            if (graph_mode == 2 && dual_stream) {
                run_main = loop_stream; run_serial = false;
                if (segs.empty() || seg_gen != plan_generation || seg_cfg != (int)cfg_on || seg_scale != cfg_scale || seg_temb != (int)temb_skip || seg_batch != batch) {
                    int rc = build_segments(batch, cfg_on, cfg_scale); if (rc) return rc;
                }
                for (int i = 0; i < n_steps; ++i) { int rc = run_segments(); if (rc) return rc; }
                MKD_HIP_CHECK(hipMemcpyAsync(x_out, s_xa, n * sizeof(float), hipMemcpyDeviceToDevice, loop_stream));
                MKD_HIP_CHECK(hipEventRecord(ev_loop_out, loop_stream));
                MKD_HIP_CHECK(hipStreamWaitEvent(stream, ev_loop_out, 0));
                return 0;
            }
            // both graphs are keyed on everything their nodes depend on (plan, guidance, batch, where the time embedding comes from)
            if (!step_graph || step_graph_gen != plan_generation || step_graph_cfg != (int)cfg_on || step_graph_scale != cfg_scale ||
                step_graph_temb != (int)temb_skip || step_graph_batch != batch) {
                drop_graph();
                hipGraph_t g = nullptr;
                MKD_HIP_CHECK(hipStreamBeginCapture(loop_stream, hipStreamCaptureModeRelaxed));
                capturing = true;
                int rc = enqueue_state_step(batch, cfg_on, cfg_scale, loop_stream);
                capturing = false;
                hipError_t e = hipStreamEndCapture(loop_stream, &g);
                if (rc) { if (g) hipGraphDestroy(g); return rc; }
                if (e != hipSuccess) return mkd_fail(MKD_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                e = hipGraphInstantiate(&step_graph, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (e != hipSuccess) { step_graph = nullptr; return mkd_fail(MKD_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
                step_graph_gen = plan_generation; step_graph_cfg = (int)cfg_on; step_graph_scale = cfg_scale;
                step_graph_temb = (int)temb_skip; step_graph_batch = batch;
            }
            // (multi_graph needs no key of its own: whenever step_graph's key above changes, drop_graph() destroys both)
            // MKD_GRAPH_STEPS = k > 1: k consecutive steps captured as ONE graph (the step reads its index from the device-resident
            // counter, so the same capture repeated k times is k different steps); the remainder runs on the single-step graph
            // (default 5: a graph boundary costs ~30 us; batch 8: 5.85 -> 5.82 ms per evaluation, batch 1: 2.99 -> 2.97)
            const int gsteps = graph_steps;
            int done = 0;
            if (gsteps > 1 && n_steps >= gsteps) {
                if (!multi_graph || multi_graph_steps != gsteps) {
                    if (multi_graph) { hipGraphExecDestroy(multi_graph); multi_graph = nullptr; }
                    hipGraph_t g = nullptr;
                    MKD_HIP_CHECK(hipStreamBeginCapture(loop_stream, hipStreamCaptureModeRelaxed));
                    capturing = true;
                    int rc = 0;
                    for (int k = 0; k < gsteps && !rc; ++k) rc = enqueue_state_step(batch, cfg_on, cfg_scale, loop_stream);
                    capturing = false;
                    hipError_t e = hipStreamEndCapture(loop_stream, &g);
                    if (rc) { if (g) hipGraphDestroy(g); return rc; }
                    if (e != hipSuccess) return mkd_fail(MKD_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                    e = hipGraphInstantiate(&multi_graph, g, nullptr, nullptr, 0);
                    hipGraphDestroy(g);
                    if (e != hipSuccess) { multi_graph = nullptr; return mkd_fail(MKD_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
                    multi_graph_steps = gsteps;
                }
                for (; done + gsteps <= n_steps; done += gsteps) MKD_HIP_CHECK(hipGraphLaunch(multi_graph, loop_stream));
            }
            for (int i = done; i < n_steps; ++i) MKD_HIP_CHECK(hipGraphLaunch(step_graph, loop_stream));
            MKD_HIP_CHECK(hipMemcpyAsync(x_out, s_xa, n * sizeof(float), hipMemcpyDeviceToDevice, loop_stream));
            MKD_HIP_CHECK(hipEventRecord(ev_loop_out, loop_stream));
            MKD_HIP_CHECK(hipStreamWaitEvent(stream, ev_loop_out, 0));
            return 0;
        }
        if (temb_table) {
            int rc = run_temb_table(n_steps, timesteps, stream); if (rc) return rc;
            temb_skip = true;
        }
        float* xa = s_xa; float* xb = s_xb;
        for (int i = 0; i < n_steps; ++i) {
            const int index = n_steps - 1 - i;
            int rc = launch_fill_i64(s_t, timesteps[index], B, stream); if (rc) return rc;
            if (temb_skip) { rc = launch_temb_select(temb_sel(), index, stream); if (rc) return rc; }
            const float* ec; const float* eu = nullptr;
            if (cfg_on) {
                rc = launch_repeat_batch(xa, s_xin, n, 2, stream); if (rc) return rc;
                rc = eps(s_xin, s_t, s_eps, stream); if (rc) return rc;
                eu = s_eps; ec = s_eps + n;
            } else {
                rc = eps(xa, s_t, s_eps, stream); if (rc) return rc;
                ec = s_eps;
            }
            const float sg = stochastic ? sigmas[index] : 0.f;
            rc = launch_ddim_step(xa, ec, eu, cfg_scale, alphas[index], alphas_prev[index], sg, s1m[index], sg != 0.f ? noise + (int64_t)i * n : nullptr,
                                  temperature, xb, nullptr, n, stream);
            if (rc) return rc;
            float* tmp = xa; xa = xb; xb = tmp;
        }
        MKD_HIP_CHECK(hipMemcpyAsync(x_out, xa, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return 0;
    }

    // ---------------------------------------------------------------------------------------------------------------
    // first-stage decoder
    // ---------------------------------------------------------------------------------------------------------------
    static std::string vae_prefix() { return "first_stage_model."; }
    void vae_add_res(const std::string& p, int cin, int cout) {
        add_param(p + ".norm1.weight", {cin}, 2); add_param(p + ".norm1.bias", {cin}, 2);
        add_param(p + ".conv1.weight", {cout, cin, 3, 3}, 2); add_param(p + ".conv1.bias", {cout}, 2);
        add_param(p + ".norm2.weight", {cout}, 2); add_param(p + ".norm2.bias", {cout}, 2);
        add_param(p + ".conv2.weight", {cout, cout, 3, 3}, 2); add_param(p + ".conv2.bias", {cout}, 2);
        if (cin != cout) { add_param(p + ".nin_shortcut.weight", {cout, cin, 1, 1}, 2); add_param(p + ".nin_shortcut.bias", {cout}, 2); }
    }
    int vae_configure(const mkd_vae_config* c) {
        if (vae_configured) return mkd_fail(MKD_ERR_STATE, "mkd_vae_configure: already configured");
        if (c->n_levels < 1 || c->n_levels > 8 || c->ch % 32 || c->z_channels != c->embed_dim || c->z_channels > 8 ||
            (c->out_ch != 3 && c->out_ch != 4))
            return mkd_fail(MKD_ERR_UNSUPPORTED, "mkd_vae_configure: unsupported decoder configuration");
        vcfg = *c;
        const std::string P = vae_prefix(), D = P + "decoder.";
        add_param(P + "post_quant_conv.weight", {c->z_channels, c->embed_dim, 1, 1}, 2);
        add_param(P + "post_quant_conv.bias", {c->z_channels}, 2);
        int bi = c->ch * c->ch_mult[c->n_levels - 1];
        add_param(D + "conv_in.weight", {bi, c->z_channels, 3, 3}, 2); add_param(D + "conv_in.bias", {bi}, 2);
        vae_add_res(D + "mid.block_1", bi, bi);
        for (const char* n : {"q", "k", "v", "proj_out"}) {
            add_param(D + "mid.attn_1." + n + ".weight", {bi, bi, 1, 1}, 2); add_param(D + "mid.attn_1." + n + ".bias", {bi}, 2);
        }
        add_param(D + "mid.attn_1.norm.weight", {bi}, 2); add_param(D + "mid.attn_1.norm.bias", {bi}, 2);
        vae_add_res(D + "mid.block_2", bi, bi);
        for (int lvl = c->n_levels - 1; lvl >= 0; --lvl) {
            const int bo = c->ch * c->ch_mult[lvl];
            for (int j = 0; j <= c->num_res_blocks; ++j) { vae_add_res(D + "up." + std::to_string(lvl) + ".block." + std::to_string(j), bi, bo); bi = bo; }
            if (lvl != 0) {
                add_param(D + "up." + std::to_string(lvl) + ".upsample.conv.weight", {bi, bi, 3, 3}, 2);
                add_param(D + "up." + std::to_string(lvl) + ".upsample.conv.bias", {bi}, 2);
            }
        }
        add_param(D + "norm_out.weight", {bi}, 2); add_param(D + "norm_out.bias", {bi}, 2);
        add_param(D + "conv_out.weight", {c->out_ch, bi, 3, 3}, 2); add_param(D + "conv_out.bias", {c->out_ch}, 2);
        vae_configured = true;
        return 0;
    }
    int vae_finalize() {
        if (!vae_configured) return mkd_fail(MKD_ERR_STATE, "decoder not configured (mkd_vae_configure)");
        for (auto& kv : params)
            if (kv.second.which == 2 && !kv.second.loaded) return mkd_fail(MKD_ERR_MISSING, "weight not loaded: " + kv.first);
        if (vae_finalized) return 0;
        if (!zero_page) {
            void* z = nullptr;
            int rc = dev_alloc(&z, 4096); if (rc) return rc;
            MKD_HIP_CHECK(hipMemset(z, 0, 4096));
            zero_page = (bf16_t*)z;
        }
        const std::string A = vae_prefix() + "decoder.mid.attn_1";
        bf16_t* qk = nullptr;
        int rc = concat_rows(&qk, {A + ".q.weight", A + ".k.weight"}); if (rc) return rc;
        vae_fused[A] = qk;
        const int c = (int)params.at(A + ".q.bias").numel();
        void* b = nullptr;
        rc = dev_alloc(&b, 2 * c * sizeof(float)); if (rc) return rc;
        MKD_HIP_CHECK(hipMemcpy(b, params.at(A + ".q.bias").dev, c * sizeof(float), hipMemcpyDeviceToDevice));
        MKD_HIP_CHECK(hipMemcpy((float*)b + c, params.at(A + ".k.bias").dev, c * sizeof(float), hipMemcpyDeviceToDevice));
        vae_fused_b[A] = (float*)b;
        MKD_HIP_CHECK(hipDeviceSynchronize());
        vae_finalized = true;
        return 0;
    }

    // VAE ResnetBlock: GN(eps 1e-6)+SiLU -> conv3x3 -> GN+SiLU -> conv3x3 (+ x or nin_shortcut(x))
    void vae_resblock(const std::string& p, const Tensor& x, int cout, bf16_t* out) {
        const size_t mk = varena.mark();
        auto tal = [&](int C_) { Tensor t = x; t.C = C_; t.ld = C_; t.p = (bf16_t*)varena.alloc((size_t)x.rows() * C_ * sizeof(bf16_t)); return t; };
        Tensor t1 = tal(x.C);
        op_gn(x, wf(p + ".norm1.weight"), wf(p + ".norm1.bias"), 1e-6f, 1, t1.p, t1.ld);
        Tensor t2 = tal(cout);
        { Epi e; e.bias = wf(p + ".conv1.bias"); op_conv(t1, wb(p + ".conv1.weight"), cout, 1, 0, e, t2.p, t2.ld); }
        Tensor t3 = tal(cout);
        op_gn(t2, wf(p + ".norm2.weight"), wf(p + ".norm2.bias"), 1e-6f, 1, t3.p, t3.ld);
        Epi e2; e2.bias = wf(p + ".conv2.bias");
        if (x.C != cout) {
            Tensor t4 = tal(cout);
            Epi es; es.bias = wf(p + ".nin_shortcut.bias");
            op_linear(x.p, x.ld, x.rows(), x.C, wb(p + ".nin_shortcut.weight"), cout, es, t4.p, t4.ld);
            e2.R = t4.p; e2.ldr = t4.ld;
        } else { e2.R = x.p; e2.ldr = x.ld; }
        op_conv(t3, wb(p + ".conv2.weight"), cout, 1, 0, e2, out, cout);
        varena.release(mk);
    }

    // single-head attention over the hw tokens of each sample, built from GEMMs (c = 512 does not fit the flash kernel)
    void vae_attn(const std::string& p, const Tensor& x, bf16_t* out) {
        const size_t mk = varena.mark();
        const int c = x.C, T = x.H * x.W, M = x.rows();
        auto buf = [&](size_t n) { return (bf16_t*)varena.alloc(n * sizeof(bf16_t)); };
        bf16_t* g = buf((size_t)M * c);
        op_gn(x, wf(p + ".norm.weight"), wf(p + ".norm.bias"), 1e-6f, 0, g, c);
        bf16_t* qk = buf((size_t)M * 2 * c);
        { Epi e; e.bias = vae_fused_b.at(p); op_linear(g, c, M, c, vae_fused.at(p), 2 * c, e, qk, 2 * c); }
        bf16_t* vt = buf((size_t)x.B * c * T);          // V^T per sample: [c][T] = Wv . g_b^T  (v bias added after PV: softmax rows sum to 1)
        bf16_t* sc = buf((size_t)x.B * T * T);
        bf16_t* pr = buf((size_t)x.B * T * T);
        bf16_t* o = buf((size_t)M * c);
        const float scale = 1.0f / sqrtf((float)c);
        for (int b = 0; b < x.B; ++b) {
            { Epi e; op_linear(wb(p + ".v.weight"), c, c, c, g + (size_t)b * T * c, T, e, vt + (size_t)b * c * T, T); }
            { Epi e; e.scale = scale;
              GemmArgs a; memset(&a, 0, sizeof(a));
              a.A = qk + (size_t)b * T * 2 * c; a.lda = 2 * c; a.W = qk + (size_t)b * T * 2 * c + c; a.ldw = 2 * c;
              a.scale = scale; a.C = sc + (size_t)b * T * T; a.ldc = T; a.M = T; a.N = T; a.K = c; a.rows_per_batch = 1;
              op_gemm(a); }
        }
        {
            bf16_t* s_ = sc; bf16_t* p_ = pr; const int rows = x.B * T, cols = T;
            push(*cur_plan, [=](hipStream_t st) { return launch_softmax_rows(s_, p_, rows, cols, st); }, 1, 0.0, K_MISC, "softmax_rows");
        }
        for (int b = 0; b < x.B; ++b) {
            Epi e; e.bias = wf(p + ".v.bias");
            op_linear(pr + (size_t)b * T * T, T, T, T, vt + (size_t)b * c * T, c, e, o + (size_t)b * T * c, c);
        }
        { Epi e; e.bias = wf(p + ".proj_out.bias"); e.R = x.p; e.ldr = x.ld; op_linear(o, c, M, c, wb(p + ".proj_out.weight"), c, e, out, c); }
        varena.release(mk);
    }

    void build_vae_plan(int Bn, int hh, int ww) {
        cur_plan = &plan_vae; cur_sid = SID_AUX; counting_eps = false;
        mkd_ctx* self = this;
        const std::string P = vae_prefix(), D = P + "decoder.";
        const int zc = vcfg.z_channels;
        // largest activation of the decoder: (2^(L-1) h)^2 pixels x ch*ch_mult[1 or 0] channels
        size_t max_act = 0;
        {
            int bi = vcfg.ch * vcfg.ch_mult[vcfg.n_levels - 1], H = hh, W = ww;
            max_act = (size_t)Bn * H * W * bi;
            for (int lvl = vcfg.n_levels - 1; lvl >= 0; --lvl) {
                const int bo = vcfg.ch * vcfg.ch_mult[lvl];
                max_act = std::max(max_act, (size_t)Bn * H * W * std::max(bi, bo));
                bi = bo;
                if (lvl != 0) { H *= 2; W *= 2; max_act = std::max(max_act, (size_t)Bn * H * W * bi); }
            }
        }
        bf16_t* X[2] = {(bf16_t*)varena.alloc(max_act * sizeof(bf16_t)), (bf16_t*)varena.alloc(max_act * sizeof(bf16_t))};
        float* zq = (float*)varena.alloc((size_t)Bn * zc * hh * ww * sizeof(float));
        {
            const bf16_t* w_ = wb(P + "post_quant_conv.weight"); const float* b_ = wf(P + "post_quant_conv.bias");
            const int hw = hh * ww;
            push(*cur_plan, [self, w_, b_, zq, Bn, zc, hw](hipStream_t st) {
                return launch_post_quant(self->io_z, w_, b_, self->io_inv_scale, zq, Bn, zc, hw, st); }, 1, 0.0, K_MISC, "post_quant");
        }
        int bi = vcfg.ch * vcfg.ch_mult[vcfg.n_levels - 1];
        int cur = 0;
        Tensor h; h.B = Bn; h.H = hh; h.W = ww; h.C = bi; h.ld = bi; h.p = X[cur];
        {
            const bf16_t* w_ = wb(D + "conv_in.weight"); const float* b_ = wf(D + "conv_in.bias"); bf16_t* dst = h.p;
            push(*cur_plan, [w_, b_, zq, dst, Bn, hh, ww, zc, bi](hipStream_t st) {
                return launch_conv3x3_direct(zq, 1, w_, b_, dst, 0, 0, nullptr, Bn, hh, ww, zc, bi, 1, st); }, 1,
                2.0 * Bn * hh * ww * bi * 9 * zc, K_CONV_DIRECT);
        }
        auto step = [&](int cout) { cur ^= 1; Tensor o = h; o.C = cout; o.ld = cout; o.p = X[cur]; return o; };
        { Tensor o = step(bi); vae_resblock(D + "mid.block_1", h, bi, o.p); h = o; }
        { Tensor o = step(bi); vae_attn(D + "mid.attn_1", h, o.p); h = o; }
        { Tensor o = step(bi); vae_resblock(D + "mid.block_2", h, bi, o.p); h = o; }
        for (int lvl = vcfg.n_levels - 1; lvl >= 0; --lvl) {
            const int bo = vcfg.ch * vcfg.ch_mult[lvl];
            for (int j = 0; j <= vcfg.num_res_blocks; ++j) {
                Tensor o = step(bo);
                vae_resblock(D + "up." + std::to_string(lvl) + ".block." + std::to_string(j), h, bo, o.p);
                h = o;
            }
            if (lvl != 0) {
                Tensor o = step(h.C); o.H = h.H * 2; o.W = h.W * 2;
                Epi e; e.bias = wf(D + "up." + std::to_string(lvl) + ".upsample.conv.bias");
                op_conv(h, wb(D + "up." + std::to_string(lvl) + ".upsample.conv.weight"), h.C, 1, 1, e, o.p, o.ld);
                h = o;
            }
        }
        {
            Tensor o = step(h.C);
            op_gn(h, wf(D + "norm_out.weight"), wf(D + "norm_out.bias"), 1e-6f, 1, o.p, o.ld);
            const bf16_t* w_ = wb(D + "conv_out.weight"); const float* b_ = wf(D + "conv_out.bias");
            const int H2 = h.H, W2 = h.W, cin = h.C, cout = vcfg.out_ch;
            push(*cur_plan, [self, o, w_, b_, Bn, H2, W2, cin, cout](hipStream_t st) {
                return launch_conv3x3_direct(o.p, 0, w_, b_, self->io_img, 1, 0, nullptr, Bn, H2, W2, cin, cout, 1, st); }, 1,
                2.0 * Bn * H2 * W2 * cout * 9 * cin, K_CONV_DIRECT);
        }
    }

    int decode(const float* z, int Bn, int hh, int ww, float scale_factor, float* images, hipStream_t stream) {
        if (!vae_finalized) { int rc = vae_finalize(); if (rc) return rc; }
        if (!z || !images || Bn <= 0 || hh <= 0 || ww <= 0 || scale_factor == 0.f) return mkd_fail(MKD_ERR_ARG, "mkd_decode: bad arguments");
        if (Bn != vae_B || hh != vae_h || ww != vae_w) {
            // same two-pass scheme as mkd_prepare: dry run sizes the arena, second pass binds pointers
            const size_t keep_sk = splitk_need, keep_gn = gn_need;
            splitk_need = 0; gn_need = 0;
            dry = true; varena.base = nullptr; varena.reset(); plan_vae.clear(); flops_vae = 0;
            build_vae_plan(Bn, hh, ww);
            int rc = ensure((void**)&varena_base, &varena_cap, varena.high + 256); if (rc) return rc;
            rc = ensure((void**)&splitk_ws[SID_AUX], &splitk_ws_bytes[SID_AUX], std::max(splitk_need, splitk_ws_bytes[SID_AUX])); if (rc) return rc;
            rc = ensure((void**)&gn_ws[SID_AUX], &gn_ws_bytes[SID_AUX], std::max(gn_need, gn_ws_bytes[SID_AUX])); if (rc) return rc;
            splitk_need = keep_sk; gn_need = keep_gn;          // (the evaluation plans keep their own maxima)
            dry = false; varena.base = varena_base; varena.reset(); plan_vae.clear();
            build_vae_plan(Bn, hh, ww);
            splitk_need = keep_sk; gn_need = keep_gn;
            for (auto& op : plan_vae) flops_vae += op.flops;
            vae_B = Bn; vae_h = hh; vae_w = ww;
        }
        io_z = z; io_img = images; io_inv_scale = 1.0f / scale_factor;
        for (auto& op : plan_vae) { int rc = op.fn(stream); if (rc) return rc; }
        return 0;
    }


    // ---- CLIP text encoder ---------------------------------------------------------------
    // cond_stage_config FrozenCLIPEmbedder (reference diffmodels/base_diffusion_makeup.yaml:109-110; called through
    // get_learned_conditioning at diffmk/makeup_teacher.py:33-42 and get_unconditional_conditioning at
    // diffmk/diffusion_makeup.py:400): UPSTREAM = transformers CLIPTextModel(...).last_hidden_state over 77 padded tokens.
    static std::string clip_prefix() { return "cond_stage_model.transformer.text_model."; }

    int clip_configure(const mkd_clip_config* c) {
        if (clip_configured) return mkd_fail(MKD_ERR_STATE, "mkd_clip_configure: already configured");
        if (c->vocab_size <= 0 || c->max_positions <= 0 || c->width <= 0 || c->layers <= 0 || c->heads <= 0 || c->intermediate <= 0 ||
            c->width % c->heads || c->width % 8 || c->intermediate % 8)
            return mkd_fail(MKD_ERR_UNSUPPORTED, "mkd_clip_configure: unsupported text-encoder configuration");
        const int dh = c->width / c->heads;
        if (!(dh == 8 || dh == 16 || dh == 32 || dh == 40 || dh == 64 || dh == 80 || dh == 160))
            return mkd_fail(MKD_ERR_UNSUPPORTED, "mkd_clip_configure: head dim " + std::to_string(dh) + " has no attention kernel");
        ccfg = *c;
        if (ccfg.ln_eps <= 0.f) ccfg.ln_eps = 1e-5f;
        const std::string P = clip_prefix();
        const int64_t W = c->width, I = c->intermediate;
        add_param(P + "embeddings.token_embedding.weight", {c->vocab_size, W}, 3);
        add_param(P + "embeddings.position_embedding.weight", {c->max_positions, W}, 3);
        for (int l = 0; l < c->layers; ++l) {
            const std::string L = P + "encoder.layers." + std::to_string(l);
            for (const char* n : {"q_proj", "k_proj", "v_proj", "out_proj"}) {
                add_param(L + ".self_attn." + n + ".weight", {W, W}, 3); add_param(L + ".self_attn." + n + ".bias", {W}, 3);
            }
            add_param(L + ".layer_norm1.weight", {W}, 3); add_param(L + ".layer_norm1.bias", {W}, 3);
            add_param(L + ".mlp.fc1.weight", {I, W}, 3); add_param(L + ".mlp.fc1.bias", {I}, 3);
            add_param(L + ".mlp.fc2.weight", {W, I}, 3); add_param(L + ".mlp.fc2.bias", {W}, 3);
            add_param(L + ".layer_norm2.weight", {W}, 3); add_param(L + ".layer_norm2.bias", {W}, 3);
        }
        add_param(P + "final_layer_norm.weight", {W}, 3); add_param(P + "final_layer_norm.bias", {W}, 3);
        clip_configured = true;
        return 0;
    }

    int clip_finalize() {
        if (!clip_configured) return mkd_fail(MKD_ERR_STATE, "text encoder not configured (mkd_clip_configure)");
        for (auto& kv : params)
            if (kv.second.which == 3 && !kv.second.loaded) return mkd_fail(MKD_ERR_MISSING, "weight not loaded: " + kv.first);
        if (clip_finalized) return 0;
        if (!zero_page) {
            void* z = nullptr;
            int rc = dev_alloc(&z, 4096); if (rc) return rc;
            MKD_HIP_CHECK(hipMemset(z, 0, 4096));
            zero_page = (bf16_t*)z;
        }
        const int W = ccfg.width;
        for (int l = 0; l < ccfg.layers; ++l) {          // one [3W, W] projection per layer: rows q | k | v
            const std::string A = clip_prefix() + "encoder.layers." + std::to_string(l) + ".self_attn";
            if (clip_qkv_w.count(A)) {                   // re-finalize after a weight reload: refresh in place
                bf16_t* d = clip_qkv_w[A]; float* b = clip_qkv_b[A]; int j = 0;
                for (const char* n : {".q_proj", ".k_proj", ".v_proj"}) {
                    MKD_HIP_CHECK(hipMemcpy(d + (size_t)j * W * W, params.at(A + n + ".weight").dev, (size_t)W * W * sizeof(bf16_t), hipMemcpyDeviceToDevice));
                    MKD_HIP_CHECK(hipMemcpy(b + (size_t)j * W, params.at(A + n + ".bias").dev, W * sizeof(float), hipMemcpyDeviceToDevice));
                    ++j;
                }
                continue;
            }
            bf16_t* w = nullptr;
            int rc = concat_rows(&w, {A + ".q_proj.weight", A + ".k_proj.weight", A + ".v_proj.weight"}); if (rc) return rc;
            void* b = nullptr;
            rc = dev_alloc(&b, 3 * W * sizeof(float)); if (rc) return rc;
            int j = 0;
            for (const char* n : {".q_proj.bias", ".k_proj.bias", ".v_proj.bias"}) {
                MKD_HIP_CHECK(hipMemcpy((float*)b + (size_t)j * W, params.at(A + n).dev, W * sizeof(float), hipMemcpyDeviceToDevice));
                ++j;
            }
            clip_qkv_w[A] = w; clip_qkv_b[A] = (float*)b;
        }
        MKD_HIP_CHECK(hipDeviceSynchronize());
        clip_finalized = true;
        clip_B = 0;
        return 0;
    }

    void build_clip_plan(int Bn, int T) {
        cur_plan = &plan_clip; cur_sid = SID_AUX; counting_eps = false;
        mkd_ctx* self = this;
        const std::string P = clip_prefix();
        const int W = ccfg.width, I = ccfg.intermediate, heads = ccfg.heads, dh = W / heads, rows = Bn * T;
        auto buf = [&](int cols) { return (bf16_t*)carena.alloc((size_t)rows * cols * sizeof(bf16_t)); };
        bf16_t* X[2] = {buf(W), buf(W)};
        bf16_t* ln = buf(W); bf16_t* qkv = buf(3 * W); bf16_t* att = buf(W); bf16_t* hid = buf(I);
        {
            const bf16_t* te = wb(P + "embeddings.token_embedding.weight"); const bf16_t* pe = wb(P + "embeddings.position_embedding.weight");
            bf16_t* dst = X[0]; const int V = ccfg.vocab_size;
            push(*cur_plan, [self, te, pe, dst, Bn, T, W, V](hipStream_t st) {
                return launch_clip_embed(self->io_tokens, te, pe, dst, Bn, T, W, V, st); }, 1, 0.0, K_MISC, "clip_embed");
        }
        int cur = 0;
        const float eps = ccfg.ln_eps;
        auto op_ln_eps = [&](const bf16_t* x, const std::string& n, bf16_t* y) {
            const float* g = wf(n + ".weight"); const float* b = wf(n + ".bias");
            push(*cur_plan, [=](hipStream_t st) { return launch_layernorm(x, g, b, eps, y, rows, W, st); }, 1, 0.0, K_LAYERNORM,
                 "rows=" + std::to_string(rows) + " d=" + std::to_string(W));
        };
        for (int l = 0; l < ccfg.layers; ++l) {
            const std::string L = P + "encoder.layers." + std::to_string(l), A = L + ".self_attn";
            op_ln_eps(X[cur], L + ".layer_norm1", ln);
            { Epi e; e.bias = dry ? nullptr : clip_qkv_b.at(A); op_linear(ln, W, rows, W, dry ? nullptr : clip_qkv_w.at(A), 3 * W, e, qkv, 3 * W); }
            {   // causal self-attention over the T padded tokens (CLIP applies no padding mask, only the causal one)
                const float scale = 1.0f / sqrtf((float)dh);
                const bf16_t* q = qkv; bf16_t* o = att;
                push(*cur_plan, [=](hipStream_t st) { return launch_attention(q, 3 * W, q + W, 3 * W, q + 2 * W, 3 * W, o, W, Bn, T, T, heads, dh, scale, st, 1); },
                     1, 4.0 * Bn * heads * (double)T * T * dh, K_ATTENTION, "clip causal B=" + std::to_string(Bn) + " T=" + std::to_string(T));
            }
            { Epi e; e.bias = wf(A + ".out_proj.bias"); e.R = X[cur]; e.ldr = W; op_linear(att, W, rows, W, wb(A + ".out_proj.weight"), W, e, X[cur ^ 1], W); }
            cur ^= 1;
            op_ln_eps(X[cur], L + ".layer_norm2", ln);
            { Epi e; e.bias = wf(L + ".mlp.fc1.bias"); e.act = 3; op_linear(ln, W, rows, W, wb(L + ".mlp.fc1.weight"), I, e, hid, I); }
            { Epi e; e.bias = wf(L + ".mlp.fc2.bias"); e.R = X[cur]; e.ldr = W; op_linear(hid, I, rows, I, wb(L + ".mlp.fc2.weight"), W, e, X[cur ^ 1], W); }
            cur ^= 1;
        }
        op_ln_eps(X[cur], P + "final_layer_norm", ln);
        {
            const int64_t n = (int64_t)rows * W;
            push(*cur_plan, [self, ln, n](hipStream_t st) { return launch_bf16_to_f32(ln, self->io_ctx_out, n, st); }, 1, 0.0, K_MISC, "clip_out");
        }
    }

    int clip_encode(const int32_t* tokens, int Bn, int T, float* out, hipStream_t stream) {
        if (!clip_finalized) { int rc = clip_finalize(); if (rc) return rc; }
        if (!tokens || !out || Bn <= 0 || T <= 0) return mkd_fail(MKD_ERR_ARG, "mkd_clip_encode: bad arguments");
        if (T > ccfg.max_positions) return mkd_fail(MKD_ERR_ARG, "mkd_clip_encode: more tokens than position embeddings");
        if (Bn != clip_B || T != clip_T) {
            const size_t keep_sk = splitk_need;
            splitk_need = 0;
            dry = true; carena.base = nullptr; carena.reset(); plan_clip.clear();
            build_clip_plan(Bn, T);
            int rc = ensure((void**)&carena_base, &carena_cap, carena.high + 256); if (rc) return rc;
            rc = ensure((void**)&splitk_ws[SID_AUX], &splitk_ws_bytes[SID_AUX], std::max(splitk_need, splitk_ws_bytes[SID_AUX])); if (rc) return rc;
            dry = false; carena.base = carena_base; carena.reset(); plan_clip.clear();
            build_clip_plan(Bn, T);
            splitk_need = keep_sk;
            clip_B = Bn; clip_T = T;
        }
        io_tokens = tokens; io_ctx_out = out;
        for (auto& op : plan_clip) { int rc = op.fn(stream); if (rc) return rc; }
        return 0;
    }

    // race detector for the tests: fill every buffer that one mkd_eps produces (activations, temporaries, workspaces) with
    // NaN patterns, so a kernel that runs ahead of its producer reads garbage instead of the previous call's (equal) values
    int debug_poison() {
        if (!prepared) return mkd_fail(MKD_ERR_STATE, "mkd_debug_poison before mkd_prepare");
        MKD_HIP_CHECK(hipDeviceSynchronize());
        if (persist_cap > persist_eps_begin) MKD_HIP_CHECK(hipMemset(persist_base + persist_eps_begin, 0xFF, persist_cap - persist_eps_begin));
        if (gstat_base) MKD_HIP_CHECK(hipMemset(gstat_base, 0xFF, gstat_cap));
        for (int i = 0; i < NS; ++i) {
            if (temp_base[i]) MKD_HIP_CHECK(hipMemset(temp_base[i], 0xFF, temp_cap[i]));
            if (splitk_ws[i]) MKD_HIP_CHECK(hipMemset(splitk_ws[i], 0xFF, splitk_ws_bytes[i]));
            if (gn_ws[i]) MKD_HIP_CHECK(hipMemset(gn_ws[i], 0xFF, gn_ws_bytes[i]));
        }
        MKD_HIP_CHECK(hipDeviceSynchronize());
        return 0;
    }

    int64_t device_bytes() const {
        return weight_bytes + (int64_t)varena_cap + (int64_t)carena_cap + (int64_t)persist_cap + (int64_t)gstat_cap + [&] { int64_t t = 0; for (int i = 0; i < NA; ++i) t += (int64_t)(temp_cap[i] + splitk_ws_bytes[i] + gn_ws_bytes[i]); return t; }();
    }

    ~mkd_ctx() {
        for (void* p : owned) hipFree(p);
        for (void* p : derived) hipFree(p);
        if (gstat_base) hipFree(gstat_base);
        if (varena_base) hipFree(varena_base);
        if (carena_base) hipFree(carena_base);
        for (auto& kv : f32_keep) hipFree(kv.second);
        drop_graph();
        drop_temb_table();
        for (hipEvent_t e : aux_ev) hipEventDestroy(e);
        for (hipEvent_t e : seg_events) hipEventDestroy(e);
        if (loop_stream) { hipStreamSynchronize(loop_stream); hipStreamDestroy(loop_stream); hipEventDestroy(ev_loop_in); hipEventDestroy(ev_loop_out); }
        if (h_state) hipHostFree(h_state);
        if (s_state) hipFree(s_state);
        for (int i = 1; i < NS; ++i)
            if (side_streams[i]) { hipStreamSynchronize(side_streams[i]); hipStreamDestroy(side_streams[i]); }
        for (int i = 0; i < NA; ++i)
            for (void* p : {(void*)temp_base[i], (void*)splitk_ws[i], (void*)gn_ws[i]})
                if (p) hipFree(p);
        for (void* p : {(void*)persist_base, (void*)s_xa, (void*)s_xb,
                        (void*)s_xin, (void*)s_eps, (void*)s_t})
            if (p) hipFree(p);
    }
};

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void mkd_set_error(const std::string& msg) { g_last_error = msg; }
int mkd_fail(int code, const std::string& msg) { g_last_error = msg; return code; }

static int g_live_contexts = 0;

extern "C" {

const char* mkd_last_error(void) { return g_last_error.c_str(); }
int mkd_abi_version(void) { return 1; }
int mkd_grouped_launches_available(void) { return MKD_PAIR_N > 1 ? 1 : 0; }

int mkd_ctx_create(const mkd_net_config* cfg, mkd_ctx** out) {
    if (!cfg || !out) return mkd_fail(MKD_ERR_ARG, "mkd_ctx_create: null argument");
    if (cfg->transformer_depth != 1) return mkd_fail(MKD_ERR_UNSUPPORTED, "only transformer_depth == 1 is supported");
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->n_attention_resolutions < 0 || cfg->n_attention_resolutions > 8)
        return mkd_fail(MKD_ERR_ARG, "mkd_ctx_create: bad level / attention_resolutions count");
    if (cfg->model_channels % 32) return mkd_fail(MKD_ERR_ARG, "model_channels must be a multiple of 32 (GroupNorm32)");
    if ((cfg->model_channels / cfg->num_heads) % 8) return mkd_fail(MKD_ERR_UNSUPPORTED, "head dim must be a multiple of 8");
    if (cfg->context_dim % 8) return mkd_fail(MKD_ERR_UNSUPPORTED, "context_dim must be a multiple of 8");
    for (int j = 0; j < 7; ++j)
        if (cfg->hint_widths[j] % 8) return mkd_fail(MKD_ERR_UNSUPPORTED, "hint widths must be multiples of 8");
    // A sampling step is replayed as ONE captured graph whose two branches (ControlNet / UNet encoder, then the decoder lanes) run on
    // different hardware queues and meet at join nodes: barrier packets that wait on the completion signals of kernels of ANOTHER
    // queue.  ROC_SYSTEM_SCOPE_SIGNAL=0 makes the runtime create those signals without system scope; a cross-queue barrier-AND on
    // such a signal was observed never to be satisfied (round 3, profiles/exp_r3_rt_env2.txt: the first graph replay did not
    // complete in 420 s).  Refuse the setting instead of hanging the device.
    if (const char* ss = getenv("ROC_SYSTEM_SCOPE_SIGNAL"))
        if (ss[0] && atoi(ss) == 0)
            return mkd_fail(MKD_ERR_UNSUPPORTED, "ROC_SYSTEM_SCOPE_SIGNAL=0 is not supported: the step graph's cross-queue join barriers wait on "
                                                 "completion signals that need system scope (a graph replay never completes); unset it");
    if (MKD_PAIR_N < 2 && getenv("MKD_ENC_GROUP") && atoi(getenv("MKD_ENC_GROUP")) != 0)
        return mkd_fail(MKD_ERR_UNSUPPORTED, "MKD_ENC_GROUP=1 needs a build with 2-entry argument tables (tools/build_variant.sh group -DMKD_PAIR_N=2)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mkd_fail(MKD_ERR_HIP, "no HIP device visible: libmkd has no CPU path");
    if (gemm_num_tile_cfgs() > K_GEMM_LIN - K_GEMM_CONV || gemm_num_tile_cfgs() > K_GROUPNORM - K_GEMM_LIN)
        return mkd_fail(MKD_ERR_STATE, "kernel-class table too small for the tile configurations of kernels_gemm.hip (OpKind)");
    mkd_ctx* c = new mkd_ctx();
    c->cfg = *cfg;
    if (const char* fl = getenv("MKD_FUSE_LN")) c->fuse_ln = fl[0] == '1';
    if (const char* sc = getenv("MKD_SPLITK_CAP")) gemm_set_splitk_cap(atoi(sc));
    c->build_param_spec();
    *out = c;
    ++g_live_contexts;
    return 0;
}

void mkd_ctx_destroy(mkd_ctx* ctx) { if (ctx) { --g_live_contexts; delete ctx; } }

int mkd_load_weight(mkd_ctx* ctx, const char* name, const float* data, int ndim, const int64_t* shape) {
    if (!ctx || !name || !data || !shape) return mkd_fail(MKD_ERR_ARG, "mkd_load_weight: null argument");
    return ctx->load_weight(name, data, ndim, shape);
}
int mkd_weights_finalize(mkd_ctx* ctx) { return ctx ? ctx->finalize() : mkd_fail(MKD_ERR_ARG, "null ctx"); }

int64_t mkd_param_count(const mkd_ctx* ctx, int which) {
    if (!ctx) return -1;
    int64_t n = 0;
    for (auto& kv : ctx->params) if (kv.second.which == which) n += kv.second.numel();
    return n;
}

int mkd_param_total(const mkd_ctx* ctx) { return ctx ? (int)ctx->params.size() : -1; }
const char* mkd_param_name(const mkd_ctx* ctx, int index) {
    if (!ctx || index < 0 || index >= (int)ctx->params.size()) return nullptr;
    auto it = ctx->params.begin();
    std::advance(it, index);
    return it->first.c_str();
}
int mkd_param_shape(const mkd_ctx* ctx, int index, int64_t* shape4) {
    if (!ctx || !shape4 || index < 0 || index >= (int)ctx->params.size()) return -1;
    auto it = ctx->params.begin();
    std::advance(it, index);
    for (size_t i = 0; i < it->second.shape.size(); ++i) shape4[i] = it->second.shape[i];
    return (int)it->second.shape.size();
}

int mkd_prepare(mkd_ctx* ctx, int batch, int h, int w, const float* hint, const float* context,
                const float* control_scales, int only_mid_control, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->prepare(batch, h, w, hint, context, control_scales, only_mid_control, (hipStream_t)stream);
}
int mkd_prepare_interp(mkd_ctx* ctx, int batch, int h, int w, const float* hint_a, const float* hint_b, const float* alpha,
                       const float* context, const float* control_scales, int only_mid_control, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    if (!hint_a || !hint_b || !alpha) return mkd_fail(MKD_ERR_ARG, "mkd_prepare_interp: null pointer");
    return ctx->prepare(batch, h, w, hint_a, context, control_scales, only_mid_control, (hipStream_t)stream, hint_b, alpha);
}
int mkd_eps(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->eps(x, t, eps_out, (hipStream_t)stream);
}
int mkd_ddim_step(const float* x, const float* eps_c, const float* eps_u, float cfg_scale, float a_t, float a_prev,
                  float sigma_t, float sqrt_one_minus_at, const float* noise, float temperature, float* x_prev,
                  float* pred_x0, int64_t n, void* stream) {
    if (!x || !eps_c || !x_prev) return mkd_fail(MKD_ERR_ARG, "mkd_ddim_step: null pointer");
    return launch_ddim_step(x, eps_c, eps_u, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise, temperature, x_prev,
                            pred_x0, n, (hipStream_t)stream);
}
int mkd_sample(mkd_ctx* ctx, const float* x_T, int batch, int n_steps, const int64_t* timesteps, const float* alphas,
               const float* alphas_prev, const float* sqrt_one_minus_alphas, float cfg_scale, float* x_out, int use_graph,
               void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->sample(x_T, batch, n_steps, timesteps, alphas, alphas_prev, sqrt_one_minus_alphas, cfg_scale, x_out, use_graph,
                       (hipStream_t)stream);
}
int mkd_sample_eta(mkd_ctx* ctx, const float* x_T, int batch, int n_steps, const int64_t* timesteps, const float* alphas,
                   const float* alphas_prev, const float* sqrt_one_minus_alphas, const float* sigmas, const float* noise, float temperature,
                   float cfg_scale, float* x_out, int use_graph, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->sample(x_T, batch, n_steps, timesteps, alphas, alphas_prev, sqrt_one_minus_alphas, cfg_scale, x_out, use_graph,
                       (hipStream_t)stream, sigmas, noise, temperature);
}
// The tile tuner's state (forced tile, XCD mode, per-shape overrides) is PROCESS-global by design: it belongs to the single-kernel
// entries and to the tuners.  A change bumps the global plan epoch, so EVERY live context re-plans at its next mkd_prepare and a plan
// never runs with decisions of another setting (launch_gemm also checks planned slab counts).  Per-context plan switches:
// mkd_ctx_set_option.  mkd_live_contexts() tells a tuner whether it is alone.
int mkd_live_contexts(void) { return g_live_contexts; }
int mkd_gemm_force_tile(int cfg) { gemm_force_tile_cfg(cfg); return 0; }
int mkd_gemm_set_xcd_mode(int mode) { gemm_set_xcd_mode(mode); return 0; }
int mkd_debug_poison(mkd_ctx* ctx) { return ctx ? ctx->debug_poison() : mkd_fail(MKD_ERR_ARG, "null ctx"); }
int mkd_ctx_set_option(mkd_ctx* ctx, const char* name, double value) {
    if (!ctx || !name) return mkd_fail(MKD_ERR_ARG, "mkd_ctx_set_option: null argument");
    const std::string n(name);
    const int iv = (int)value;
    if (n == "tfm_tail") ctx->tfm_tail = iv;
    else if (n == "tfm_tail_min_rows") ctx->tfm_tail_min_rows = iv;
    else if (n == "tfm_head") ctx->tfm_head = iv;
    else if (n == "skip_fold") ctx->skip_fold = iv;
    else if (n == "gn_2k_min_hw") ctx->gn_2k_min_hw = iv;
    else if (n == "xcd_auto_ratio") ctx->xcd_auto_ratio = (float)value;
    else if (n == "dec_lanes") { if (iv != 0 && iv != 2 && iv != 4) return mkd_fail(MKD_ERR_ARG, "dec_lanes: 0, 2 or 4"); ctx->dec_lanes = iv; }
    else if (n == "ln_fly") ctx->ln_fly = iv & 7;
    else if (n == "gn_slab_min_channels") ctx->gn_slab_minc = iv;
    else if (n == "graph_steps") { if (iv < 1) return mkd_fail(MKD_ERR_ARG, "graph_steps >= 1"); ctx->graph_steps = iv; ctx->drop_graph(); return 0; }
    else return mkd_fail(MKD_ERR_ARG, "mkd_ctx_set_option: unknown option '" + n + "'");
    ++ctx->opt_epoch;
    return 0;
}
int mkd_ctx_get_option(const mkd_ctx* ctx, const char* name, double* value) {
    if (!ctx || !name || !value) return mkd_fail(MKD_ERR_ARG, "mkd_ctx_get_option: null argument");
    const std::string n(name);
    if (n == "tfm_tail") *value = ctx->tfm_tail;
    else if (n == "tfm_tail_min_rows") *value = ctx->tfm_tail_min_rows;
    else if (n == "tfm_head") *value = ctx->tfm_head;
    else if (n == "skip_fold") *value = ctx->skip_fold;
    else if (n == "gn_2k_min_hw") *value = ctx->gn_2k_min_hw;
    else if (n == "xcd_auto_ratio") *value = ctx->xcd_auto_ratio;
    else if (n == "dec_lanes") *value = ctx->dec_lanes;
    else if (n == "ln_fly") *value = ctx->ln_fly;
    else if (n == "gn_slab_min_channels") *value = ctx->gn_slab_minc;
    else if (n == "graph_steps") *value = ctx->graph_steps;
    else return mkd_fail(MKD_ERR_ARG, "mkd_ctx_get_option: unknown option '" + n + "'");
    return 0;
}
int mkd_gemm_set_override(int M, int N, int K, int conv3x3, int stride, int up, int cfg, int splitk) {
    gemm_set_override(M, N, K, conv3x3, stride, up, cfg, splitk);
    return 0;
}
int mkd_gemm_cfg_supported(int cfg, int M, int N, int K, int conv3x3, int Hin, int Win, int Cin, int Hout, int Wout, int stride, int up) {
    if (cfg < 0 || cfg >= gemm_num_tile_cfgs()) return 0;
    if (!((cfg >= 6 && cfg <= 11) || (cfg >= 38 && cfg <= 40) || cfg == 42 || cfg == 43)) return 1;          // (only the LDS-staged conv tiles depend on the geometry)
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.conv = conv3x3; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.stride = stride; a.up = up;
    return conv_patch_supported(a, cfg) ? 1 : 0;
}
int mkd_kind_count(void) { return K_COUNT; }
const char* mkd_kind_name(int kind) {
    static thread_local std::string nm;
    if (kind < 0 || kind >= K_COUNT) return nullptr;
    nm = kind_name(kind);
    return nm.c_str();
}
int mkd_eps_profile(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream, double* ms_per_kind,
                    double* flops_per_kind, int* launches_per_kind, const char* csv_path) {
    if (!ctx || !ms_per_kind || !flops_per_kind || !launches_per_kind) return mkd_fail(MKD_ERR_ARG, "mkd_eps_profile: null argument");
    return ctx->eps_profile(x, t, eps_out, (hipStream_t)stream, ms_per_kind, flops_per_kind, launches_per_kind, csv_path);
}
int mkd_eps_profile2(mkd_ctx* ctx, const float* x, const int64_t* t, float* eps_out, void* stream, double* ms_per_kind,
                     double* flops_per_kind, int* launches_per_kind, double* bytes_per_kind, double* ms_back_to_back_per_kind, const char* csv_path) {
    if (!ctx || !ms_per_kind || !flops_per_kind || !launches_per_kind) return mkd_fail(MKD_ERR_ARG, "mkd_eps_profile2: null argument");
    return ctx->eps_profile(x, t, eps_out, (hipStream_t)stream, ms_per_kind, flops_per_kind, launches_per_kind, csv_path, bytes_per_kind,
                            ms_back_to_back_per_kind);
}
int mkd_vae_configure(mkd_ctx* ctx, const mkd_vae_config* cfg) {
    if (!ctx || !cfg) return mkd_fail(MKD_ERR_ARG, "mkd_vae_configure: null argument");
    return ctx->vae_configure(cfg);
}
int mkd_vae_finalize(mkd_ctx* ctx) { return ctx ? ctx->vae_finalize() : mkd_fail(MKD_ERR_ARG, "null ctx"); }
int mkd_decode(mkd_ctx* ctx, const float* z, int batch, int h, int w, float scale_factor, float* images, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->decode(z, batch, h, w, scale_factor, images, (hipStream_t)stream);
}
double mkd_decode_flops(const mkd_ctx* ctx) { return ctx ? ctx->flops_vae : 0.0; }
int mkd_clip_configure(mkd_ctx* ctx, const mkd_clip_config* cfg) {
    if (!ctx || !cfg) return mkd_fail(MKD_ERR_ARG, "mkd_clip_configure: null argument");
    return ctx->clip_configure(cfg);
}
int mkd_clip_finalize(mkd_ctx* ctx) { return ctx ? ctx->clip_finalize() : mkd_fail(MKD_ERR_ARG, "null ctx"); }
int mkd_clip_encode(mkd_ctx* ctx, const int32_t* tokens, int batch, int n_tokens, float* out, void* stream) {
    if (!ctx) return mkd_fail(MKD_ERR_ARG, "null ctx");
    return ctx->clip_encode(tokens, batch, n_tokens, out, (hipStream_t)stream);
}
double mkd_eps_flops(const mkd_ctx* ctx) { return ctx ? ctx->flops_eps : 0.0; }
int mkd_eps_launches(const mkd_ctx* ctx) { return ctx ? ctx->launches_eps : 0; }
// what sample_impl / build_segments enqueue per step: the evaluation (without its own time-embedding chain when the per-call table
// is on) + graph replay: step setup (timestep, coefficients, table rows) and the state update; eager: timestep fill, table-row
// select (table on) and the update; + the batch doubling of x with guidance
int mkd_step_launches_ex(const mkd_ctx* ctx, int use_graph, int cfg_on) {
    if (!ctx) return 0;
    const int eval = ctx->launches_eps - (ctx->temb_table ? ctx->launches_temb : 0);
    return eval + (use_graph ? 2 : (ctx->temb_table ? 3 : 2)) + (cfg_on ? 1 : 0);
}
int mkd_step_launches(const mkd_ctx* ctx) { return mkd_step_launches_ex(ctx, 1, 0); }
int64_t mkd_device_bytes(const mkd_ctx* ctx) { return ctx ? ctx->device_bytes() : 0; }

// ---- single-kernel entry points ----------------------------------------------------------------------
static bf16_t* g_zero = nullptr;
static float* g_ws = nullptr; static size_t g_ws_bytes = 0;
static float* g_gn = nullptr; static size_t g_gn_bytes = 0;

static int scratch(float** p, size_t* have, size_t need) {
    if (*have >= need && *p) return 0;
    if (*p) { hipDeviceSynchronize(); hipFree(*p); *p = nullptr; }
    MKD_HIP_CHECK(hipMalloc((void**)p, need ? need : 256));
    *have = need;
    return 0;
}

static int gemm_entry(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* rowbias, int ldrb,
                      int rows_per_batch, const uint16_t* R, int ldr, float scale, int act, void* C, int ldc, int out_f32, int M,
                      int N, int K, int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int stride, int up,
                      int splitk, long long* gn_stat, int gn_cg, int gn_coff, int gn_hw, void* stream) {
    if (!A || !W || !C) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_bf16: null pointer");
    if (!g_zero) {
        MKD_HIP_CHECK(hipMalloc((void**)&g_zero, 4096));
        MKD_HIP_CHECK(hipMemset(g_zero, 0, 4096));
    }
    if (conv3x3 && M != batch * Hout * Wout) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_bf16: M != batch*Hout*Wout");
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.rowbias = rowbias; a.ldrb = ldrb; a.rows_per_batch = rows_per_batch;
    a.R = R; a.ldr = ldr; a.scale = scale; a.act = act; a.C = C; a.ldc = ldc; a.out_f32 = out_f32; a.M = M; a.N = N; a.K = K;
    a.conv = conv3x3 ? 1 : 0; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.stride = stride; a.up = up;
    a.zero = g_zero;
    a.gn_stat = gn_stat; a.gn_cg = gn_cg; a.gn_coff = gn_coff; a.gn_hw = gn_hw;
    a.splitk = splitk > 0 ? splitk : 0;
    int cfg_i = 0, s = 1;
    int rc = gemm_resolve(a, &cfg_i, &s);          // the decision launch_gemm will take, fallbacks included
    if (rc) return rc;
    rc = scratch(&g_ws, &g_ws_bytes, gemm_ws_bytes(M, N, s > 1 ? s : 2));
    if (rc) return rc;
    a.ws = g_ws; a.ws_bytes = g_ws_bytes;
    return launch_gemm(a, (hipStream_t)stream);
}
int mkd_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* rowbias, int ldrb,
                  int rows_per_batch, const uint16_t* R, int ldr, float scale, int act, void* C, int ldc, int out_f32, int M,
                  int N, int K, int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int stride, int up,
                  int splitk, void* stream) {
    return gemm_entry(A, lda, W, ldw, bias, rowbias, ldrb, rows_per_batch, R, ldr, scale, act, C, ldc, out_f32, M, N, K, conv3x3, batch,
                      Hin, Win, Cin, Hout, Wout, stride, up, splitk, nullptr, 0, 0, 0, stream);
}
int mkd_conv3x3_fold_bf16(const uint16_t* x, int ldx, const uint16_t* w_fold, const float* bias, const uint16_t* x2, int ldx2, int K2, uint16_t* y,
                          int ldy, int batch, int H, int W, int Cin, int N, int splitk, void* stream) {
    if (!x || !w_fold || !x2 || !y) return mkd_fail(MKD_ERR_ARG, "mkd_conv3x3_fold_bf16: null pointer");
    if (!g_zero) {
        MKD_HIP_CHECK(hipMalloc((void**)&g_zero, 4096));
        MKD_HIP_CHECK(hipMemset(g_zero, 0, 4096));
    }
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = x; a.lda = ldx; a.W = w_fold; a.ldw = 9 * Cin + K2; a.bias = bias; a.scale = 1.0f; a.C = y; a.ldc = ldy; a.M = batch * H * W; a.N = N;
    a.K = 9 * Cin + K2; a.conv = 1; a.Hin = H; a.Win = W; a.Cin = Cin; a.Hout = H; a.Wout = W; a.stride = 1; a.up = 0;
    a.A2 = x2; a.lda2 = ldx2; a.K2 = K2; a.zero = g_zero; a.splitk = splitk > 0 ? splitk : 0;
    int cfg_i = 0, s = 1;
    int rc = gemm_resolve(a, &cfg_i, &s);
    if (rc) return rc;
    rc = scratch(&g_ws, &g_ws_bytes, gemm_ws_bytes(a.M, N, s > 1 ? s : 2));
    if (rc) return rc;
    a.ws = g_ws; a.ws_bytes = g_ws_bytes;
    return launch_gemm(a, (hipStream_t)stream);
}
int mkd_gemm_gnstat_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* rowbias, int ldrb,
                         int rows_per_batch, const uint16_t* R, int ldr, float scale, int act, void* C, int ldc, int out_f32, int M,
                         int N, int K, int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int stride, int up,
                         int splitk, int64_t* gn_stat, int gn_cg, int gn_coff, int gn_hw, void* stream) {
    if (!gn_stat) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_gnstat_bf16: null statistics buffer");
    return gemm_entry(A, lda, W, ldw, bias, rowbias, ldrb, rows_per_batch, R, ldr, scale, act, C, ldc, out_f32, M, N, K, conv3x3, batch,
                      Hin, Win, Cin, Hout, Wout, stride, up, splitk, (long long*)gn_stat, gn_cg, gn_coff, gn_hw, stream);
}
int mkd_gemm_groupnorm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* rowbias, int ldrb,
                            int rows_per_batch, const uint16_t* R, int ldr, float scale, void* C, int ldc, int write_raw, int M, int N,
                            int K, int conv3x3, int batch, int Hin, int Win, int Cin, int Hout, int Wout, int stride, int up, int splitk,
                            int rows_per_sample, const float* gamma, const float* beta, float eps, int silu, uint16_t* y, int ld_y,
                            void* stream) {
    if (!A || !W || (!C && write_raw) || !gamma || !beta || !y) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_groupnorm_bf16: null pointer");
    if (rows_per_sample <= 0 || M % rows_per_sample) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_groupnorm_bf16: M must be a multiple of rows_per_sample");
    if (!g_zero) {
        MKD_HIP_CHECK(hipMalloc((void**)&g_zero, 4096));
        MKD_HIP_CHECK(hipMemset(g_zero, 0, 4096));
    }
    if (conv3x3 && M != batch * Hout * Wout) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_groupnorm_bf16: M != batch*Hout*Wout");
    const int nb = M / rows_per_sample;
    if (!gn_from_slabs_supported(nb, rows_per_sample, N)) return mkd_fail(MKD_ERR_UNSUPPORTED, "mkd_gemm_groupnorm_bf16: GroupNorm geometry does not fit the single-pass kernel");
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.rowbias = rowbias; a.ldrb = ldrb; a.rows_per_batch = rows_per_batch;
    a.R = R; a.ldr = ldr; a.scale = scale; a.act = 0; a.C = C; a.ldc = ldc; a.out_f32 = 0; a.M = M; a.N = N; a.K = K;
    a.conv = conv3x3 ? 1 : 0; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.stride = stride; a.up = up;
    a.zero = g_zero; a.splitk = splitk > 0 ? splitk : 0;
    int cfg_i = 0, s = 1;
    int rc = gemm_resolve(a, &cfg_i, &s);
    if (rc) return rc;
    if (s < 2) return mkd_fail(MKD_ERR_UNSUPPORTED, "mkd_gemm_groupnorm_bf16: this shape is not split over K");
    rc = scratch(&g_ws, &g_ws_bytes, gemm_ws_bytes(M, N, s));
    if (rc) return rc;
    a.ws = g_ws; a.ws_bytes = g_ws_bytes; a.defer_epilogue = 1;
    rc = launch_gemm(a, (hipStream_t)stream);
    if (rc) return rc;
    a.splitk = s;
    if (!write_raw) a.C = nullptr;
    return launch_gn_from_slabs(a, gamma, beta, eps, silu, y, ld_y, nb, rows_per_sample, (hipStream_t)stream);
}
int mkd_gn_colstats(const uint16_t* x, int ld, int batch, int hw, int ncols, int cg, int coff, int64_t* gstat, void* stream) {
    if (!x || !gstat) return mkd_fail(MKD_ERR_ARG, "mkd_gn_colstats: null pointer");
    return launch_gn_colstats(x, ld, batch, hw, ncols, cg, coff, (long long*)gstat, (hipStream_t)stream);
}
int mkd_gn_apply_stats(const uint16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu, uint16_t* y,
                       int ld_out, int batch, int hw, int C, const int64_t* gstat, void* stream) {
    if (!x || !y || !gamma || !beta || !gstat) return mkd_fail(MKD_ERR_ARG, "mkd_gn_apply_stats: null pointer");
    return launch_gn_apply_stats(x, ld_in, gamma, beta, eps, silu, y, ld_out, batch, hw, C, (const long long*)gstat, (hipStream_t)stream);
}
int mkd_fold_layernorm(const float* w, const float* gamma, const float* beta, const float* bias, int N, int K, uint16_t* w_out,
                       int dst_row0, int dst_row_mul, float* s_out, float* b_out, void* stream) {
    if (!w || !gamma || !beta || !w_out || !s_out || !b_out) return mkd_fail(MKD_ERR_ARG, "mkd_fold_layernorm: null pointer");
    return launch_fold_layernorm(w, gamma, beta, bias, N, K, w_out, dst_row0, dst_row_mul, s_out, b_out, (hipStream_t)stream);
}
int mkd_gemm_ln_bf16(const uint16_t* A, int lda, const uint16_t* Wfold, int ldw, const float* bias_fold, const float* ln_s,
                     const float* row_stats, int stat_slots, float eps, int act, void* C, int ldc, int M, int N, int K, void* stream) {
    if (!A || !Wfold || !C || !ln_s) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_ln_bf16: null pointer");          // row_stats null: on the fly
    if (!g_zero) {
        MKD_HIP_CHECK(hipMalloc((void**)&g_zero, 4096));
        MKD_HIP_CHECK(hipMemset(g_zero, 0, 4096));
    }
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = Wfold; a.ldw = ldw; a.bias = bias_fold; a.scale = 1.f; a.act = act; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.rows_per_batch = 1; a.zero = g_zero; a.ln_s = ln_s; a.ln_eps = eps;
    a.stat_in = row_stats; a.stat_in_slots = stat_slots;
    return launch_gemm(a, (hipStream_t)stream);
}
int mkd_gemm_rowstats_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* R, int ldr,
                           uint16_t* C, int ldc, int M, int N, int K, float* stat_out, int stat_capacity_slots, int* slots_out,
                           void* stream) {
    if (!A || !W || !C || !stat_out || !slots_out) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_rowstats_bf16: null pointer");
    if (!g_zero) {
        MKD_HIP_CHECK(hipMalloc((void**)&g_zero, 4096));
        MKD_HIP_CHECK(hipMemset(g_zero, 0, 4096));
    }
    const int slots = gemm_stat_slots(M, N, K);
    if (slots > stat_capacity_slots) return mkd_fail(MKD_ERR_ARG, "mkd_gemm_rowstats_bf16: statistics buffer too small");
    *slots_out = slots;
    GemmArgs a; memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.R = R; a.ldr = ldr; a.scale = 1.f; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.rows_per_batch = 1; a.zero = g_zero; a.stat_out = stat_out;
    int cfg_i = 0, s = 1;
    int rc = gemm_resolve(a, &cfg_i, &s);
    if (rc) return rc;
    rc = scratch(&g_ws, &g_ws_bytes, gemm_ws_bytes(M, N, s > 1 ? s : 2));
    if (rc) return rc;
    a.ws = g_ws; a.ws_bytes = g_ws_bytes;
    return launch_gemm(a, (hipStream_t)stream);
}
int mkd_groupnorm(const uint16_t* x, int ld_in, const float* gamma, const float* beta, float eps, int silu, uint16_t* y,
                  int ld_out, int batch, int hw, int C, int groups, void* stream) {
    int rc = scratch(&g_gn, &g_gn_bytes, groupnorm_partials_bytes(batch, hw, groups));
    if (rc) return rc;
    return launch_groupnorm(x, ld_in, gamma, beta, eps, silu, y, ld_out, batch, hw, C, groups, g_gn, (hipStream_t)stream);
}
int mkd_layernorm(const uint16_t* x, const float* gamma, const float* beta, float eps, uint16_t* y, int rows, int d, void* stream) {
    return launch_layernorm(x, gamma, beta, eps, y, rows, d, (hipStream_t)stream);
}
int mkd_layernorm_ld(const uint16_t* x, int ldx, const float* gamma, const float* beta, float eps, uint16_t* y, int rows, int d, void* stream) {
    return launch_layernorm(x, gamma, beta, eps, y, rows, d, (hipStream_t)stream, ldx);
}
int mkd_attention(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, uint16_t* o, int ldo,
                  int batch, int Tq, int Tk, int heads, int dh, float scale, void* stream) {
    return launch_attention(q, ldq, k, ldk, v, ldv, o, ldo, batch, Tq, Tk, heads, dh, scale, (hipStream_t)stream);
}
int mkd_attention_causal(const uint16_t* q, int ldq, const uint16_t* k, int ldk, const uint16_t* v, int ldv, uint16_t* o, int ldo,
                         int batch, int Tq, int Tk, int heads, int dh, float scale, void* stream) {
    return launch_attention(q, ldq, k, ldk, v, ldv, o, ldo, batch, Tq, Tk, heads, dh, scale, (hipStream_t)stream, 1);
}
int mkd_geglu(const uint16_t* x, uint16_t* y, int rows, int inner, void* stream) {
    return launch_geglu(x, y, rows, inner, (hipStream_t)stream);
}
int mkd_conv3x3_direct(const void* x, int in_nchw_f32, const uint16_t* w, const float* bias, void* y, int out_nchw_f32, int act,
                       const uint16_t* add, int batch, int Hin, int Win, int Cin, int Cout, int stride, void* stream) {
    return launch_conv3x3_direct(x, in_nchw_f32, w, bias, y, out_nchw_f32, act, add, batch, Hin, Win, Cin, Cout, stride,
                                 (hipStream_t)stream);
}
int mkd_pack_conv_weight(const float* w, uint16_t* out, int Cout, int Cin, int kh, int kw, void* stream) {
    return launch_pack_conv_weight(w, out, Cout, Cin, kh, kw, (hipStream_t)stream);
}

// ---- fused transformer tail, stand-alone (unit parity test, tools/bench_tfm_tail.py) ------------------------------------------
struct mkd_tfm_tail { int d = 0; bf16_t* wpk = nullptr; float* vec = nullptr; bf16_t* kvp = nullptr; int kv_batch = 0, Tk = 0; std::vector<void*> owned; };
void mkd_tfm_tail_destroy(mkd_tfm_tail* h) {
    if (!h) return;
    for (void* p : h->owned) hipFree(p);
    delete h;
}
int mkd_tfm_tail_create(int d, const float* to_out1_w, const float* to_out1_b, const float* norm2_g, const float* norm2_b,
                        const float* to_q2_w, const float* to_out2_w, const float* to_out2_b, const float* norm3_g, const float* norm3_b,
                        const float* ff0_w, const float* ff0_b, const float* ff2_w, const float* ff2_b, const float* proj_out_w,
                        const float* proj_out_b, mkd_tfm_tail** out) {
    if (!out) return mkd_fail(MKD_ERR_ARG, "null out");
    if (!tfm_tail_weight_bytes(d)) return mkd_fail(MKD_ERR_UNSUPPORTED, "tfm_tail: only d = 320 is built");
    mkd_tfm_tail* h = new mkd_tfm_tail; h->d = d;
    auto dal = [&](size_t bytes, void** o) -> int { MKD_HIP_CHECK(hipMalloc(o, bytes)); h->owned.push_back(*o); return 0; };
    void *wo1 = nullptr, *wq = nullptr, *sq = nullptr, *bq = nullptr, *wo2 = nullptr, *wg = nullptr, *sg = nullptr, *bg = nullptr, *wm = nullptr, *bm = nullptr;
    const size_t dd = (size_t)d * d;
    int rc = dal(dd * 2, &wo1);
    if (!rc) rc = dal(dd * 2, &wq); if (!rc) rc = dal(d * 4, &sq); if (!rc) rc = dal(d * 4, &bq);
    if (!rc) rc = dal(dd * 2, &wo2);
    if (!rc) rc = dal(8 * dd * 2, &wg); if (!rc) rc = dal(8 * d * 4, &sg); if (!rc) rc = dal(8 * d * 4, &bg);
    if (!rc) rc = dal(5 * dd * 2, &wm); if (!rc) rc = dal(d * 4, &bm);
    if (!rc) rc = dal(tfm_tail_weight_bytes(d), (void**)&h->wpk);
    if (!rc) rc = dal(tfm_tail_vec_bytes(d), (void**)&h->vec);
    if (!rc) rc = launch_f32_to_bf16(to_out1_w, (bf16_t*)wo1, (int64_t)dd, 0);
    if (!rc) rc = launch_f32_to_bf16(to_out2_w, (bf16_t*)wo2, (int64_t)dd, 0);
    if (!rc) rc = launch_fold_layernorm(to_q2_w, norm2_g, norm2_b, nullptr, d, d, (bf16_t*)wq, 0, 1, (float*)sq, (float*)bq, 0);
    for (int half = 0; half < 2 && !rc; ++half)      // rows [0, 4d) value, [4d, 8d) gate -> row 2 j = value_j, 2 j + 1 = gate_j (as mkd_ctx::finalize)
        rc = launch_fold_layernorm(ff0_w + (size_t)half * 4 * dd, norm3_g, norm3_b, ff0_b + half * 4 * d, 4 * d, d, (bf16_t*)wg, half, 2, (float*)sg, (float*)bg, 0);
    if (!rc) rc = launch_merge_ff_out(proj_out_w, ff2_w, ff2_b, proj_out_b, (bf16_t*)wm, (float*)bm, d, 0);
    if (!rc) {
        TfmTailWeights s{(const bf16_t*)wo1, to_out1_b, (const bf16_t*)wq, (const float*)sq, (const float*)bq, (const bf16_t*)wo2, to_out2_b,
                         (const bf16_t*)wg, (const float*)sg, (const float*)bg, (const bf16_t*)wm, (const float*)bm};
        rc = tfm_tail_pack_weights(d, s, h->wpk, h->vec, 0);
    }
    if (rc) { mkd_tfm_tail_destroy(h); return rc; }
    *out = h;
    return 0;
}
/* experiment builds (-DMKD_TFM_TRACE) only: device buffer [workgroups][8][32] of int64 time stamps; a no-op in the product build */
int mkd_debug_tfm_trace(long long* buf) { tfm_tail_set_trace(buf); return 0; }
int mkd_debug_attn_trace(long long* buf) { return attn_set_trace(buf); }
int mkd_tfm_tail_set_context(mkd_tfm_tail* h, const uint16_t* kv, int ldkv, int batch, int Tk, void* stream) {
    if (!h) return mkd_fail(MKD_ERR_ARG, "null handle");
    if (batch > h->kv_batch) {
        void* p = nullptr;
        MKD_HIP_CHECK(hipMalloc(&p, tfm_tail_kv_bytes(h->d, batch)));
        h->owned.push_back(p); h->kvp = (bf16_t*)p; h->kv_batch = batch;
    }
    h->Tk = Tk;
    return launch_tfm_tail_pack_kv(h->d, kv, ldkv, batch, Tk, h->kvp, (hipStream_t)stream);
}
int mkd_tfm_tail_run(mkd_tfm_tail* h, const uint16_t* a1, int lda, const uint16_t* h0, int ldh, const uint16_t* xin, int ldx, uint16_t* out,
                     int ldo, int M, int T, void* stream) {
    if (!h || !h->kvp) return mkd_fail(MKD_ERR_STATE, "tfm_tail: set the context first");
    if (T <= 0 || M % T || M / T > h->kv_batch) return mkd_fail(MKD_ERR_ARG, "tfm_tail: M must be samples x T within the packed context");
    return launch_tfm_tail(h->d, h->wpk, h->vec, a1, lda, h0, ldh, xin, ldx, h->kvp, out, ldo, M, T, h->Tk, (hipStream_t)stream);
}

// ---- fused transformer head, stand-alone (unit parity test) ---------------------------------------------------------------------
struct mkd_tfm_head { int d = 0; bf16_t* wpk = nullptr; float* vec = nullptr; float* part = nullptr; size_t part_bytes = 0; std::vector<void*> owned; };
void mkd_tfm_head_destroy(mkd_tfm_head* h) {
    if (!h) return;
    for (void* p : h->owned) hipFree(p);
    delete h;
}
int mkd_tfm_head_create(int d, const float* gn_g, const float* gn_b, const float* proj_in_w, const float* proj_in_b, const float* norm1_g,
                        const float* norm1_b, const float* to_q_w, const float* to_k_w, const float* to_v_w, mkd_tfm_head** out) {
    if (!out) return mkd_fail(MKD_ERR_ARG, "null out");
    if (!tfm_head_weight_bytes(d)) return mkd_fail(MKD_ERR_UNSUPPORTED, "tfm_head: only d = 320 is built");
    mkd_tfm_head* h = new mkd_tfm_head; h->d = d;
    auto dal = [&](size_t bytes, void** o) -> int { MKD_HIP_CHECK(hipMalloc(o, bytes)); h->owned.push_back(*o); return 0; };
    void *wpi = nullptr, *wq = nullptr, *sq = nullptr, *bq = nullptr;
    const size_t dd = (size_t)d * d;
    int rc = dal(dd * 2, &wpi);
    if (!rc) rc = dal(3 * dd * 2, &wq); if (!rc) rc = dal(3 * d * 4, &sq); if (!rc) rc = dal(3 * d * 4, &bq);
    if (!rc) rc = dal(tfm_head_weight_bytes(d), (void**)&h->wpk);
    if (!rc) rc = dal(tfm_head_vec_bytes(d), (void**)&h->vec);
    if (!rc) rc = launch_f32_to_bf16(proj_in_w, (bf16_t*)wpi, (int64_t)dd, 0);
    const float* parts[3] = {to_q_w, to_k_w, to_v_w};
    for (int j = 0; j < 3 && !rc; ++j)
        rc = launch_fold_layernorm(parts[j], norm1_g, norm1_b, nullptr, d, d, (bf16_t*)wq, j * d, 1, (float*)sq, (float*)bq, 0);
    if (!rc) {
        TfmHeadWeights s{gn_g, gn_b, (const bf16_t*)wpi, proj_in_b, (const bf16_t*)wq, (const float*)sq, (const float*)bq};
        rc = tfm_head_pack_weights(d, s, h->wpk, h->vec, 0);
    }
    if (rc) { mkd_tfm_head_destroy(h); return rc; }
    *out = h;
    return 0;
}
int mkd_tfm_head_run(mkd_tfm_head* h, const uint16_t* x, int ldx, float gn_eps, uint16_t* h0, uint16_t* qkv, int batch, int T, void* stream) {
    if (!h) return mkd_fail(MKD_ERR_ARG, "null handle");
    const size_t need = groupnorm_partials_bytes(batch, T, 32);
    if (need > h->part_bytes) {
        void* p = nullptr;
        MKD_HIP_CHECK(hipMalloc(&p, need));
        h->owned.push_back(p); h->part = (float*)p; h->part_bytes = need;
    }
    int nch = 0;
    int rc = launch_gn_stats(x, ldx, batch, T, h->d, 32, h->part, (hipStream_t)stream, &nch);
    if (rc) return rc;
    return launch_tfm_head(h->d, h->wpk, h->vec, x, ldx, h->part, nch, gn_eps, h0, qkv, batch * T, T, (hipStream_t)stream);
}

}  // extern "C"
