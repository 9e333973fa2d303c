// Small HBM-bound kernels of the path: DDIM update (+CFG combine), GEGLU, sinusoidal timestep
// embedding, the three tiny-channel 3x3 convs (4->C, 6->16, C->4), weight packing, layout glue.
#include "mkd_common.h"
#include <cstring>

namespace {

// ---- DDIM x0 / x_{t-1} update, reference diffmk/cddim.py:39-40 (CFG) and :63,74-78 ----------------
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                 const float* __restrict__ eps_u, float cfg_scale, float sqrt_at_inv,
                                 float sqrt_aprev, float dir_coef, float sigma_t, float s1m,
                                 const float* __restrict__ noise, float temperature,
                                 float* __restrict__ x_prev, float* __restrict__ pred_x0, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = eps_c[i];
        if (eps_u) {
            const float u = eps_u[i];
            e = u + cfg_scale * (e - u);            // model_uncond + s * (model_t - model_uncond)
        }
        const float xv = x[i];
        const float p0 = (xv - s1m * e) * sqrt_at_inv;   // (x - sqrt(1-a_t) e) / sqrt(a_t)
        float xp = sqrt_aprev * p0 + dir_coef * e;       // sqrt(a_prev) x0 + sqrt(1-a_prev-sigma^2) e
        if (noise) xp += sigma_t * noise[i] * temperature;
        x_prev[i] = xp;
        if (pred_x0) pred_x0[i] = p0;
    }
}

// ---- device-resident step state for hipGraph replay of the DDIM loop -------------------------------------
// One graph = one reverse step; what changes between steps (timestep, schedule coefficients) is read from
// device tables through a counter that the first kernel of the graph advances, so the SAME graph replays.
__device__ __forceinline__ void temb_copy_rows(const TembSel& ts, int step, int gtid, int gthreads) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int nv = ts.n[k] >> 2;
        if (!nv) continue;
        const f32x4* src = (const f32x4*)(ts.tab[k] + (size_t)step * ts.n[k]);
        f32x4* dst = (f32x4*)ts.proj[k];
        for (int i = gtid; i < nv * ts.batch; i += gthreads) dst[i] = src[i % nv];
    }
}

__global__ void step_setup_kernel(StepState* st, int64_t* t_out, int batch, const TembSel ts) {
    const int i = st->counter;          // read-only in this kernel: ddim_step_state_kernel, the step's last, advances it
    if (blockIdx.x == 0) {
        for (int b = threadIdx.x; b < batch; b += blockDim.x) t_out[b] = st->timesteps[i];
        if (threadIdx.x == 0) {
            st->cur[0] = st->coef[4 * i + 0]; st->cur[1] = st->coef[4 * i + 1];
            st->cur[2] = st->coef[4 * i + 2]; st->cur[3] = st->coef[4 * i + 3];
            st->cur_sigma = st->sigma[i]; st->cur_row = st->n_steps - 1 - i;
        }
    }
    temb_copy_rows(ts, i, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

__global__ void temb_select_kernel(const TembSel ts, int step) {
    temb_copy_rows(ts, step, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// in-place x <- x_{t-1} (eta == 0), coefficients from the step state; ends the step: the counter moves to the next one
__global__ void ddim_step_state_kernel(float* __restrict__ x, const float* __restrict__ eps_c, const float* __restrict__ eps_u,
                                       float cfg_scale, StepState* __restrict__ st, int64_t n) {
    const float sqrt_at_inv = st->cur[0], sqrt_aprev = st->cur[1], dir_coef = st->cur[2], s1m = st->cur[3];
    const float* __restrict__ const nz = (st->noise && st->cur_sigma != 0.f) ? st->noise + (int64_t)st->cur_row * n : nullptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->counter = st->counter - 1;      // (nothing else in this kernel reads it)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = eps_c[i];
        if (eps_u) { const float u = eps_u[i]; e = u + cfg_scale * (e - u); }
        const float p0 = (x[i] - s1m * e) * sqrt_at_inv;
        float xp = sqrt_aprev * p0 + dir_coef * e;
        if (nz) xp += st->cur_sigma * nz[i] * st->temperature;          // (ddim_step_kernel's order of operations: same bits as the eager loop)
        x[i] = xp;
    }

}

// ---- GEGLU: y = a * gelu_erf(gate) ----------------------------------------------------------------
__global__ void geglu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int rows, int inner) {
    const int vper = inner >> 3;
    const int64_t total = (int64_t)rows * vper;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / vper);
        const int c = (int)(idx - (int64_t)r * vper) << 3;
        const U16x8 a = *(const U16x8*)(x + (size_t)r * 2 * inner + c);
        const U16x8 gt = *(const U16x8*)(x + (size_t)r * 2 * inner + inner + c);
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = f32_to_bf16(bf16_to_f32(a.v[j]) * gelu_erf_f(bf16_to_f32(gt.v[j])));
        *(U16x8*)(y + (size_t)r * inner + c) = o;
    }
}

// ---- sinusoidal timestep embedding: out[b] = [cos(t f_k), sin(t f_k)], f_k = exp(-ln(1e4) k / half) --
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, bf16_t* __restrict__ out, int batch, int dim) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * half) return;
    const int b = idx / half, k = idx - b * half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = (float)t[b] * freq;
    out[(size_t)b * dim + k] = f32_to_bf16(cosf(a));
    out[(size_t)b * dim + half + k] = f32_to_bf16(sinf(a));
}

// ---- direct 3x3 conv (pad 1), one thread per output element, fp32 accumulate ----------------------
// Only used where the channel count is too small for the MFMA path (Cin 4/6, or Cout 4).
__global__ void conv3x3_direct_kernel(const void* __restrict__ xin, int in_nchw_f32, const bf16_t* __restrict__ w,
                                      const float* __restrict__ bias, void* __restrict__ yout, int out_nchw_f32,
                                      int act, const bf16_t* __restrict__ add, int batch, int Hin, int Win,
                                      int Cin, int Cout, int Hout, int Wout, int stride) {
    const int64_t total = (int64_t)batch * Hout * Wout * Cout;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout);
        int64_t pix = idx / Cout;
        const int ox = (int)(pix % Wout); pix /= Wout;
        const int oy = (int)(pix % Hout);
        const int b = (int)(pix / Hout);
        float acc = bias ? bias[co] : 0.f;
        const bf16_t* wr = w + (size_t)co * 9 * Cin;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + ky - 1;
            if ((unsigned)iy >= (unsigned)Hin) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + kx - 1;
                if ((unsigned)ix >= (unsigned)Win) continue;
                const bf16_t* wt = wr + (ky * 3 + kx) * Cin;
                if (in_nchw_f32) {
                    const float* xp = (const float*)xin + ((size_t)b * Cin * Hin + iy) * Win + ix;
                    for (int ci = 0; ci < Cin; ++ci)
                        acc += xp[(size_t)ci * Hin * Win] * bf16_to_f32(wt[ci]);
                } else {
                    const bf16_t* xp = (const bf16_t*)xin + ((size_t)(b * Hin + iy) * Win + ix) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) acc += bf16_to_f32(xp[ci]) * bf16_to_f32(wt[ci]);
                }
            }
        }
        if (act == 1) acc = silu_f(acc);
        const size_t opix = ((size_t)b * Hout + oy) * Wout + ox;
        if (add) acc += bf16_to_f32(add[opix * Cout + co]);
        if (out_nchw_f32) ((float*)yout)[(((size_t)b * Cout + co) * Hout + oy) * Wout + ox] = acc;
        else ((bf16_t*)yout)[opix * Cout + co] = f32_to_bf16(acc);
    }
}

// ---- 3x3 conv with a tiny Cin read from fp32 NCHW (the 4 -> 320 input conv): thread = output channel, the
// thread's 9*CIN weights live in registers, a block walks pixels with the 9*CIN input patch shared through LDS.
template <int CIN>
__global__ __launch_bounds__(512) void conv3x3_fewin_kernel(const float* __restrict__ x, const Pair<ConvInIo> io, int act, int batch, int H, int W,
                                                            int Cout, int pix_per_block) {
    // grouped launch: grid y selects the problem (both nets convolve the SAME x with their own weights)
    const bf16_t* __restrict__ const w = io.g[blockIdx.y].w; const float* __restrict__ const bias = io.g[blockIdx.y].bias;
    bf16_t* __restrict__ const y = io.g[blockIdx.y].y; const bf16_t* __restrict__ const add = io.g[blockIdx.y].add;
    constexpr int KK = 9 * CIN;
    constexpr int PPB_MAX = 16;
    __shared__ float patch[PPB_MAX][KK];
    const int co = threadIdx.x;
    const bool live = co < Cout;
    float wr[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) wr[k] = live ? bf16_to_f32(w[(size_t)co * KK + k]) : 0.f;
    const float bz = (live && bias) ? bias[co] : 0.f;
    const int npix = batch * H * W;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(npix, p0 + pix_per_block);
    // every input patch of the block's pixels in ONE round of loads (a per-pixel prefetch only covers one iteration, ~0.1 us,
    // of a ~1 us load latency: the old loop paid that latency per pixel)
    for (int idx = threadIdx.x; idx < (p1 - p0) * KK; idx += blockDim.x) {
        const int pp = idx / KK, k = idx - pp * KK;
        const int tap = k / CIN, ci = k - tap * CIN;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int pix = p0 + pp;
        const int b = pix / (H * W);
        const int rem = pix - b * H * W;
        const int oy = rem / W, ox = rem - oy * W;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((size_t)b * CIN + ci) * H + iy) * W + ix];
        patch[pp][k] = v;
    }
    __syncthreads();
    if (!live) return;
    for (int pix = p0; pix < p1; ++pix) {
        float acc = bz;
#pragma unroll
        for (int k = 0; k < KK; ++k) acc += patch[pix - p0][k] * wr[k];
        if (act == 1) acc = silu_f(acc);
        if (add) acc += bf16_to_f32(add[(size_t)pix * Cout + co]);
        y[(size_t)pix * Cout + co] = f32_to_bf16(acc);
    }
}

// ---- 3x3 conv with a tiny Cout (the UNet's final C -> 4): one wavefront per output pixel ---------------
// lanes split the (tap, ci) reduction in 16-B vectors, COUT accumulators per lane, wave-shuffle tree at the end.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y_nchw,
                                                             int batch, int H, int W, int Cin) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int npix = batch * H * W;
    if (pix >= npix) return;
    const int b = pix / (H * W);
    const int rem = pix - b * H * W;
    const int oy = rem / W, ox = rem - oy * W;
    const int V = Cin >> 3;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
    for (int idx = lane; idx < 9 * V; idx += 64) {
        const int tap = idx / V, v = idx - tap * V;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        const U16x8 xv = *(const U16x8*)(x + ((size_t)(b * H + iy) * W + ix) * Cin + v * 8);
        float xf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = bf16_to_f32(xv.v[j]);
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            const U16x8 wv = *(const U16x8*)(w + ((size_t)c * 9 + tap) * Cin + v * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[c] += xf[j] * bf16_to_f32(wv.v[j]);
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        float a = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) y_nchw[(((size_t)b * COUT + c) * H + oy) * W + ox] = a + (bias ? bias[c] : 0.f);
    }
}

// ---- row softmax over bf16 [rows, cols] (VAE mid-block attention scores), one 256-thread block per row ----------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int cols) {
    __shared__ float red[4];
    const bf16_t* xr = x + (size_t)blockIdx.x * cols;
    bf16_t* yr = y + (size_t)blockIdx.x * cols;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, bf16_to_f32(xr[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) sum += __expf(bf16_to_f32(xr[c]) - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = f32_to_bf16(__expf(bf16_to_f32(xr[c]) - mx) * inv);
}

// ---- post_quant_conv: 1x1 conv over fp32 NCHW latents with the 1/scale_factor of decode_first_stage folded in -------
__global__ void post_quant_kernel(const float* __restrict__ z, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                  float inv_scale, float* __restrict__ out, int batch, int C, int hw) {
    const int64_t total = (int64_t)batch * C * hw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx % hw);
        const int co = (int)((idx / hw) % C);
        const int b = (int)(idx / ((int64_t)hw * C));
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < C; ++ci) acc += bf16_to_f32(w[co * C + ci]) * (z[((size_t)b * C + ci) * hw + p] * inv_scale);
        out[idx] = acc;
    }
}

// ---- per-sample blend of two bf16 tensors: y = (1 - alpha[b]) * a + alpha[b] * b (makeup interpolation) ---------------
__global__ void blend_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ alpha,
                             bf16_t* __restrict__ y, int64_t per_sample, int batch) {
    const int64_t total = per_sample * batch / 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const float al = alpha[(idx * 8) / per_sample];
        const U16x8 va = *(const U16x8*)(a + idx * 8);
        const U16x8 vb = *(const U16x8*)(b + idx * 8);
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = f32_to_bf16((1.0f - al) * bf16_to_f32(va.v[j]) + al * bf16_to_f32(vb.v[j]));
        *(U16x8*)(y + idx * 8) = o;
    }
}

// ---- fp32 [Cout,Cin,kh,kw] -> bf16 [Cout][kh][kw][Cin] ----------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int kh, int kw) {
    const int64_t total = (int64_t)Cout * Cin * kh * kw;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(idx % Cin);
        int64_t r = idx / Cin;
        const int x = (int)(r % kw); r /= kw;
        const int y = (int)(r % kh);
        const int co = (int)(r / kh);
        out[idx] = f32_to_bf16(w[(((size_t)co * Cin + ci) * kh + y) * kw + x]);
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = f32_to_bf16(x[i]);
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = bf16_to_f32(x[i]);
}

// CLIP text embeddings: out[b, t, :] = token_embedding[tokens[b, t]] + position_embedding[t]   (8 channels per thread)
__global__ void clip_embed_kernel(const int32_t* __restrict__ tokens, const bf16_t* __restrict__ tok_emb,
                                  const bf16_t* __restrict__ pos_emb, bf16_t* __restrict__ out, int rows, int T, int width, int vocab) {
    const int vper = width >> 3;
    const int64_t total = (int64_t)rows * vper;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / vper);
        const int c = (int)(idx - (int64_t)r * vper) << 3;
        int id = tokens[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);        // the host wrapper rejects out-of-range ids; never read OOB
        const U16x8 a = *(const U16x8*)(tok_emb + (size_t)id * width + c);
        const U16x8 b = *(const U16x8*)(pos_emb + (size_t)(r % T) * width + c);
        U16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = f32_to_bf16(bf16_to_f32(a.v[j]) + bf16_to_f32(b.v[j]));
        *(U16x8*)(out + (size_t)r * width + c) = o;
    }
}

// Load-time fold of the transformer's last two linear maps: out = proj_out(h2 + FF2(gg)) + x_in
//   = [gg | h2] . [P.W2 | P]^T + (P.b2 + bp) + x_in      (P = proj_out [d,d], W2 = ff.net.2 [d,4d]; fp32 product, bf16 result)
__global__ void merge_ff_out_kernel(const float* __restrict__ P, const float* __restrict__ W2, const float* __restrict__ b2,
                                    const float* __restrict__ bp, bf16_t* __restrict__ Wm, float* __restrict__ bias_m, int d) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int K4 = 4 * d;
    if (k < K4) {
        float a = 0.f;
        for (int j = 0; j < d; ++j) a += P[(size_t)n * d + j] * W2[(size_t)j * K4 + k];
        Wm[(size_t)n * 5 * d + k] = f32_to_bf16(a);
    } else if (k < 5 * d) {
        Wm[(size_t)n * 5 * d + k] = f32_to_bf16(P[(size_t)n * d + (k - K4)]);
    }
    if (k == 0) {
        float a = bp[n];
        for (int j = 0; j < d; ++j) a += P[(size_t)n * d + j] * b2[j];
        bias_m[n] = a;
    }
}

__global__ void copy_strided_kernel(const bf16_t* __restrict__ src, int ld_src, bf16_t* __restrict__ dst, int ld_dst,
                                    int rows, int cols) {
    const int vper = cols >> 3;
    const int64_t total = (int64_t)rows * vper;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / vper);
        const int c = (int)(idx - (int64_t)r * vper) << 3;
        *(U16x8*)(dst + (size_t)r * ld_dst + c) = *(const U16x8*)(src + (size_t)r * ld_src + c);
    }
}

__global__ void repeat_batch_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_per, int reps) {
    const int64_t total = n_per * reps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i % n_per];
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline int grid_for(int64_t total, int block = 256, int cap = 2048) {
    int64_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

int launch_ddim_step(const float* x, const float* eps_c, const float* eps_u, float cfg_scale, float a_t,
                     float a_prev, float sigma_t, float s1m, const float* noise, float temperature,
                     float* x_prev, float* pred_x0, int64_t n, hipStream_t stream) {
    if (n <= 0) return mkd_fail(-1, "ddim_step: empty");
    const float sqrt_at_inv = 1.0f / sqrtf(a_t);
    const float sqrt_aprev = sqrtf(a_prev);
    const float dir_coef = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_c, eps_u, cfg_scale,
                       sqrt_at_inv, sqrt_aprev, dir_coef, sigma_t, s1m, noise, temperature, x_prev, pred_x0, n);
    MKD_LAUNCH_CHECK("ddim_step_kernel");
    return 0;
}

int launch_geglu(const bf16_t* x, bf16_t* y, int rows, int inner, hipStream_t stream) {
    if (inner % 8) return mkd_fail(-1, "geglu: inner must be a multiple of 8");
    hipLaunchKernelGGL(geglu_kernel, dim3(grid_for((int64_t)rows * (inner / 8), 256, 4096)), dim3(256), 0, stream, x, y, rows, inner);
    MKD_LAUNCH_CHECK("geglu_kernel");
    return 0;
}

int launch_timestep_embedding(const int64_t* t, bf16_t* out, int batch, int dim, hipStream_t stream) {
    if (dim % 2) return mkd_fail(-1, "timestep_embedding: odd dim");
    const int total = batch * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t, out, batch, dim);
    MKD_LAUNCH_CHECK("timestep_embedding_kernel");
    return 0;
}

int launch_conv3x3_direct(const void* x, int in_nchw_f32, const bf16_t* w, const float* bias, void* y,
                          int out_nchw_f32, int act, const bf16_t* add, int batch, int Hin, int Win,
                          int Cin, int Cout, int stride, hipStream_t stream, const ConvInIo* second) {
    if (stride != 1 && stride != 2) return mkd_fail(-1, "conv3x3_direct: stride must be 1 or 2");
    if (in_nchw_f32 && !out_nchw_f32 && stride == 1 && Cin == 4 && Cout >= 64 && Cout <= 512) {
        const int npix = batch * Hin * Win;
        const int ppb = 16;            // <= PPB_MAX of the kernel
        const int threads = (Cout + 63) / 64 * 64;
        Pair<ConvInIo> io;
        io.g[0] = ConvInIo{w, bias, (bf16_t*)y, add};
        MKD_PAIR_SET2(io, second ? *second : io.g[0]);
        hipLaunchKernelGGL(conv3x3_fewin_kernel<4>, dim3((npix + ppb - 1) / ppb, second ? 2 : 1), dim3(threads), 0, stream, (const float*)x, io,
                           act, batch, Hin, Win, Cout, ppb);
        MKD_LAUNCH_CHECK("conv3x3_fewin_kernel");
        return 0;
    }
    if (second) return mkd_fail(-1, "conv3x3_direct: only the 4 -> C input convolution has a grouped form");
    if (!in_nchw_f32 && out_nchw_f32 && (Cout == 4 || Cout == 3) && stride == 1 && act == 0 && !add && Cin % 8 == 0) {
        const int npix = batch * Hin * Win;
        if (Cout == 4)
            hipLaunchKernelGGL(conv3x3_fewout_kernel<4>, dim3((npix + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, w, bias,
                               (float*)y, batch, Hin, Win, Cin);
        else
            hipLaunchKernelGGL(conv3x3_fewout_kernel<3>, dim3((npix + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, w, bias,
                               (float*)y, batch, Hin, Win, Cin);
        MKD_LAUNCH_CHECK("conv3x3_fewout_kernel");
        return 0;
    }
    const int Hout = (Hin + 2 - 3) / stride + 1, Wout = (Win + 2 - 3) / stride + 1;
    const int64_t total = (int64_t)batch * Hout * Wout * Cout;
    hipLaunchKernelGGL(conv3x3_direct_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, stream, x, in_nchw_f32, w, bias,
                       y, out_nchw_f32, act, add, batch, Hin, Win, Cin, Cout, Hout, Wout, stride);
    MKD_LAUNCH_CHECK("conv3x3_direct_kernel");
    return 0;
}

int launch_pack_conv_weight(const float* w, bf16_t* out, int Cout, int Cin, int kh, int kw, hipStream_t stream) {
    const int64_t total = (int64_t)Cout * Cin * kh * kw;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, stream, w, out, Cout, Cin, kh, kw);
    MKD_LAUNCH_CHECK("pack_conv_weight_kernel");
    return 0;
}

int launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, x, y, n);
    MKD_LAUNCH_CHECK("f32_to_bf16_kernel");
    return 0;
}

int launch_bf16_to_f32(const bf16_t* x, float* y, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, x, y, n);
    MKD_LAUNCH_CHECK("bf16_to_f32_kernel");
    return 0;
}

int launch_clip_embed(const int32_t* tokens, const bf16_t* tok_emb, const bf16_t* pos_emb, bf16_t* out, int batch, int T, int width,
                      int vocab, hipStream_t stream) {
    if (width % 8) return mkd_fail(-1, "clip_embed: width must be a multiple of 8");
    hipLaunchKernelGGL(clip_embed_kernel, dim3(grid_for((int64_t)batch * T * (width / 8))), dim3(256), 0, stream, tokens, tok_emb, pos_emb,
                       out, batch * T, T, width, vocab);
    MKD_LAUNCH_CHECK("clip_embed_kernel");
    return 0;
}

int launch_merge_ff_out(const float* P, const float* W2, const float* b2, const float* bp, bf16_t* Wm, float* bias_m, int d,
                        hipStream_t stream) {
    hipLaunchKernelGGL(merge_ff_out_kernel, dim3((5 * d + 255) / 256, d), dim3(256), 0, stream, P, W2, b2, bp, Wm, bias_m, d);
    MKD_LAUNCH_CHECK("merge_ff_out_kernel");
    return 0;
}

int launch_copy_strided(const bf16_t* src, int ld_src, bf16_t* dst, int ld_dst, int rows, int cols, hipStream_t stream) {
    if (cols % 8 || ld_src % 8 || ld_dst % 8) return mkd_fail(-1, "copy_strided: multiples of 8 required");
    hipLaunchKernelGGL(copy_strided_kernel, dim3(grid_for((int64_t)rows * (cols / 8))), dim3(256), 0, stream, src, ld_src, dst,
                       ld_dst, rows, cols);
    MKD_LAUNCH_CHECK("copy_strided_kernel");
    return 0;
}

int launch_repeat_batch(const float* x, float* y, int64_t n_per, int reps, hipStream_t stream) {
    hipLaunchKernelGGL(repeat_batch_kernel, dim3(grid_for(n_per * reps)), dim3(256), 0, stream, x, y, n_per, reps);
    MKD_LAUNCH_CHECK("repeat_batch_kernel");
    return 0;
}

int launch_fill_i64(int64_t* p, int64_t v, int n, hipStream_t stream) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, v, n);
    MKD_LAUNCH_CHECK("fill_i64_kernel");
    return 0;
}

static int temb_blocks(const TembSel& ts) {
    const long long v = ((long long)(ts.n[0] >> 2) + (ts.n[1] >> 2)) * ts.batch;
    long long b = (v + 1023) / 1024;          // ~4 float4 per thread
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
int launch_temb_select(const TembSel& ts, int step, hipStream_t stream) {
    if ((ts.n[0] | ts.n[1]) & 3) return mkd_fail(-1, "temb_select: row lengths must be multiples of 4");
    if (!ts.n[0] && !ts.n[1]) return 0;
    hipLaunchKernelGGL(temb_select_kernel, dim3(temb_blocks(ts)), dim3(256), 0, stream, ts, step);
    MKD_LAUNCH_CHECK("temb_select_kernel");
    return 0;
}
int launch_step_setup(StepState* st, int64_t* t_out, int batch, hipStream_t stream, const TembSel* tsp) {
    TembSel ts; memset(&ts, 0, sizeof(ts));
    if (tsp) ts = *tsp;
    if ((ts.n[0] | ts.n[1]) & 3) return mkd_fail(-1, "step_setup: row lengths must be multiples of 4");
    const int blocks = (ts.n[0] || ts.n[1]) ? temb_blocks(ts) : 1;
    hipLaunchKernelGGL(step_setup_kernel, dim3(blocks), dim3(256), 0, stream, st, t_out, batch, ts);
    MKD_LAUNCH_CHECK("step_setup_kernel");
    return 0;
}

int launch_ddim_step_state(float* x, const float* eps_c, const float* eps_u, float cfg_scale, StepState* st, int64_t n,
                           hipStream_t stream) {
    hipLaunchKernelGGL(ddim_step_state_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, eps_c, eps_u, cfg_scale, st, n);
    MKD_LAUNCH_CHECK("ddim_step_state_kernel");
    return 0;
}

int launch_softmax_rows(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t stream) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, y, cols);
    MKD_LAUNCH_CHECK("softmax_rows_kernel");
    return 0;
}

int launch_post_quant(const float* z, const bf16_t* w, const float* bias, float inv_scale, float* out, int batch, int C, int hw,
                      hipStream_t stream) {
    hipLaunchKernelGGL(post_quant_kernel, dim3(grid_for((int64_t)batch * C * hw)), dim3(256), 0, stream, z, w, bias, inv_scale, out,
                       batch, C, hw);
    MKD_LAUNCH_CHECK("post_quant_kernel");
    return 0;
}

int launch_blend(const bf16_t* a, const bf16_t* b, const float* alpha, bf16_t* y, int64_t per_sample, int batch, hipStream_t stream) {
    if (per_sample % 8) return mkd_fail(-1, "blend: per-sample size must be a multiple of 8");
    hipLaunchKernelGGL(blend_kernel, dim3(grid_for(per_sample * batch / 8)), dim3(256), 0, stream, a, b, alpha, y, per_sample, batch);
    MKD_LAUNCH_CHECK("blend_kernel");
    return 0;
}
