// Pieces of the Transformer 0D encoder (src/models/transformer.py, nn.TransformerEncoderLayer post-norm form; SURVEY 8a row
// a12): fused residual-add + LayerNorm, the attention core of nn.MultiheadAttention with an additive mask, and the two GELU
// forms the reference uses.  Sequences are tiny (S = 21 tokens, d_model 128-256, head dim 16-32): one workgroup per row /
// per (batch, head); everything fixed-order fp32.
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads, red[4]
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// out[row] = LayerNorm(a[row] + b[row]) * gamma + beta  (b may be NULL).  Saves xhat [rows][D] and rstd [rows].
__global__ __launch_bounds__(256) void k_add_ln_fwd(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int D, float eps,
                                                   float* __restrict__ out, float* __restrict__ xhat, float* __restrict__ rstd,
                                                   float* __restrict__ sum_out) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float* ar = a + row * D; const float* br = b ? b + row * D : nullptr;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += ar[i] + (br ? br[i] : 0.f);
  const float mean = block_sum(s, red) / (float)D;
  float v = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { const float d = ar[i] + (br ? br[i] : 0.f) - mean; v += d * d; }
  const float var = block_sum(v, red) / (float)D;
  const float rs = 1.f / sqrtf(var + eps);
  for (int i = threadIdx.x; i < D; i += 256) {
    const float xh = (ar[i] + (br ? br[i] : 0.f) - mean) * rs;
    xhat[row * D + i] = xh;
    out[row * D + i] = xh * gamma[i] + beta[i];
    if (sum_out) sum_out[row * D + i] = ar[i] + (br ? br[i] : 0.f);      // the pre-norm residual stream (ViViT.py:108-111)
  }
  if (threadIdx.x == 0) rstd[row] = rs;
}
// dx[row] = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dout * gamma   (the same dx goes to both summands;
// dres = gradient that reached the sum through the residual stream)
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ dout, const float* __restrict__ gamma,
                                                   const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                   const float* __restrict__ dres, int D, float* __restrict__ dx) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { const float g = dout[row * D + i] * gamma[i]; s1 += g; s2 += g * xhat[row * D + i]; }
  const float m1 = block_sum(s1, red) / (float)D;
  const float m2 = block_sum(s2, red) / (float)D;
  const float rs = rstd[row];
  for (int i = threadIdx.x; i < D; i += 256) {
    const float g = dout[row * D + i] * gamma[i];
    dx[row * D + i] = rs * (g - m1 - xhat[row * D + i] * m2) + (dres ? dres[row * D + i] : 0.f);
  }
}
// dgamma[i] = sum_rows dout * xhat, dbeta[i] = sum_rows dout   (one thread per feature; blockIdx.y = chunk of rows, in order;
// with more than one chunk the outputs are per-chunk partials [chunks][D] that k_ln_param_sum adds in chunk order)
#define LN_CHUNKS 128
__global__ __launch_bounds__(256) void k_ln_param_grad(const float* __restrict__ dout, const float* __restrict__ xhat, int rows, int D,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  const int per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float g = 0.f, bsum = 0.f;
  for (int r = r0; r < r1; ++r) { const float d = dout[(size_t)r * D + i]; g = fmaf(d, xhat[(size_t)r * D + i], g); bsum += d; }
  dgamma[(size_t)blockIdx.y * D + i] = g; dbeta[(size_t)blockIdx.y * D + i] = bsum;
}
__global__ __launch_bounds__(256) void k_ln_param_sum(const float* __restrict__ pg, const float* __restrict__ pb, int chunks, int D,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  float g = 0.f, b = 0.f;
  for (int c = 0; c < chunks; ++c) { g += pg[(size_t)c * D + i]; b += pb[(size_t)c * D + i]; }
  dgamma[i] = g; dbeta[i] = b;
}

// Attention core.  qkv [S][B][3*D] (q | k | v as nn.MultiheadAttention's in_proj lays them out), head h uses features
// h*dh .. (h+1)*dh of each part.  probs [B*H][S][S] (after softmax, before dropout) kept for the backward; drop (may be NULL)
// [B*H][S][S] holds mask/keep factors of the attention dropout.  out [S][B][D].  mask [S][S] additive (may be NULL).
// One workgroup per (batch*head, block of ATT_RB query rows): the score rows of the block live in LDS, K and V are read from
// L2 (S = 21 for the 0D encoder, 197 for ViViT's spatial transformer).
#define ATT_RB 16
// row of token i of sequence b: [S][B][.] (nn.MultiheadAttention) or, batch-first, [B][S][.] (ViViT.py:69-73)
#define ROW(i) (bf ? (size_t)b * S + (i) : (size_t)(i) * B + b)
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                 const float* __restrict__ drop, int S, int B, int D, int H, int bf, float scale,
                                                 float* __restrict__ probs, float* __restrict__ out) {
  extern __shared__ float sm[];        // p [rb][S]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, dh = D / H, t = threadIdx.x;
  const int i0 = blockIdx.y * ATT_RB, rb = min(ATT_RB, S - i0);
  float* p = sm;
  for (int e = t; e < rb * S; e += 256) {
    const int r = e / S, j = e - r * S, i = i0 + r;
    const float* q = qkv + ROW(i) * 3 * D + h * dh;
    const float* k = qkv + ROW(j) * 3 * D + D + h * dh;
    float a = 0.f;
    for (int c = 0; c < dh; ++c) a = fmaf(q[c], k[c], a);
    p[e] = a * scale + (mask ? mask[(size_t)i * S + j] : 0.f);
  }
  __syncthreads();
  {   // softmax of each row: 16 lanes per row (fixed-order tree inside the 16 lanes)
    const int r = t >> 4, l = t & 15;
    if (r < rb) {
      float mx = -INFINITY;
      for (int j = l; j < S; j += 16) mx = fmaxf(mx, p[r * S + j]);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
      for (int j = l; j < S; j += 16) { const float e = expf(p[r * S + j] - mx); p[r * S + j] = e; sum += e; }
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.f / sum;
      for (int j = l; j < S; j += 16) p[r * S + j] *= inv;
    }
  }
  __syncthreads();
  for (int e = t; e < rb * S; e += 256) {
    const int r = e / S, j = e - r * S;
    const size_t g = ((size_t)bh * S + i0 + r) * S + j;
    probs[g] = p[e];
    if (drop) p[e] *= drop[g];
  }
  __syncthreads();
  for (int e = t; e < rb * dh; e += 256) {
    const int r = e / dh, c = e - r * dh;
    float a = 0.f;
    for (int j = 0; j < S; ++j) a = fmaf(p[r * S + j], qkv[ROW(j) * 3 * D + 2 * D + h * dh + c], a);
    out[ROW(i0 + r) * D + h * dh + c] = a;
  }
}
// Backward, pass A (per block of query rows): dS = P * (dP - sum_j dP*P) * scale with dP = (dout v^T) * drop, written to
// ds [B*H][S][S]; dq = dS k.
__global__ __launch_bounds__(256) void k_attn_bwd_q(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                   const float* __restrict__ drop, const float* __restrict__ dout, int S, int B,
                                                   int D, int H, int bf, float scale, float* __restrict__ ds_out, float* __restrict__ dqkv) {
  extern __shared__ float sm[];        // ds [rb][S]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, dh = D / H, t = threadIdx.x;
  const int i0 = blockIdx.y * ATT_RB, rb = min(ATT_RB, S - i0);
  float* ds = sm;
  for (int e = t; e < rb * S; e += 256) {
    const int r = e / S, j = e - r * S, i = i0 + r;
    float a = 0.f;
    for (int c = 0; c < dh; ++c)
      a = fmaf(dout[ROW(i) * D + h * dh + c], qkv[ROW(j) * 3 * D + 2 * D + h * dh + c], a);
    const size_t g = ((size_t)bh * S + i) * S + j;
    ds[e] = a * (drop ? drop[g] : 1.f);
  }
  __syncthreads();
  {
    const int r = t >> 4, l = t & 15;
    if (r < rb) {
      const float* pr = probs + ((size_t)bh * S + i0 + r) * S;
      float dot = 0.f;
      for (int j = l; j < S; j += 16) dot = fmaf(ds[r * S + j], pr[j], dot);
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
      for (int j = l; j < S; j += 16) ds[r * S + j] = pr[j] * (ds[r * S + j] - dot) * scale;
    }
  }
  __syncthreads();
  for (int e = t; e < rb * S; e += 256) { const int r = e / S, j = e - r * S; ds_out[((size_t)bh * S + i0 + r) * S + j] = ds[e]; }
  for (int e = t; e < rb * dh; e += 256) {
    const int r = e / dh, c = e - r * dh;
    float a = 0.f;
    for (int j = 0; j < S; ++j) a = fmaf(ds[r * S + j], qkv[ROW(j) * 3 * D + D + h * dh + c], a);
    dqkv[ROW(i0 + r) * 3 * D + h * dh + c] = a;
  }
}
// Backward, pass B (per block of key rows j): dk[j] = sum_i dS[i][j] q[i];  dv[j] = sum_i (P*drop)[i][j] dout[i].
__global__ __launch_bounds__(256) void k_attn_bwd_kv(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                    const float* __restrict__ drop, const float* __restrict__ dout,
                                                    const float* __restrict__ ds, int S, int B, int D, int H, int bf,
                                                    float* __restrict__ dqkv) {
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, dh = D / H, t = threadIdx.x;
  const int j0 = blockIdx.y * ATT_RB, rb = min(ATT_RB, S - j0);
  for (int e = t; e < rb * dh; e += 256) {
    const int r = e / dh, c = e - r * dh, j = j0 + r;
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < S; ++i) {
      const size_t g = ((size_t)bh * S + i) * S + j;
      ak = fmaf(ds[g], qkv[ROW(i) * 3 * D + h * dh + c], ak);
      av = fmaf(probs[g] * (drop ? drop[g] : 1.f), dout[ROW(i) * D + h * dh + c], av);
    }
    dqkv[ROW(j) * 3 * D + D + h * dh + c] = ak;
    dqkv[ROW(j) * 3 * D + 2 * D + h * dh + c] = av;
  }
}

#undef ROW

// GELU.  kind 0: exact, 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU, transformer.py:85); kind 1: the reference's own tanh form
// 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) (transformer.py:36-38).  bwd: dx = dy * gelu'(x).
__global__ __launch_bounds__(256) void k_gelu(const float* __restrict__ x, const float* __restrict__ dy, int kind, int64_t n,
                                             float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    float y, d;
    if (kind == 0) {
      const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
      y = v * cdf; d = cdf + v * 0.3989422804014327f * expf(-0.5f * v * v);
    } else {
      const float c = 0.7978845608028654f, u = c * (v + 0.044715f * v * v * v), th = tanhf(u);
      y = 0.5f * v * (1.f + th);
      d = 0.5f * (1.f + th) + 0.5f * v * (1.f - th * th) * c * (1.f + 3.f * 0.044715f * v * v);
    }
    out[i] = dy ? dy[i] * d : y;
  }
}

// ELU (ViViT.py:166, MultiModal heads): y = x > 0 ? x : alpha (e^x - 1);  bwd: dx = dy * (x > 0 ? 1 : alpha e^x)
__global__ __launch_bounds__(256) void k_elu(const float* __restrict__ x, const float* __restrict__ dy, float alpha, int64_t n,
                                            float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i], e = alpha * expf(fminf(v, 0.f));
    const bool pos = !(v <= 0.f);                       // NaN stays NaN, as in ATen
    out[i] = dy ? dy[i] * (pos ? 1.f : e) : (pos ? v : e - alpha);
  }
}
// Tensor-fusion outer product (MultiModal.py:216-220): with A = [1 | a[b]] (Da+1) and C = [1 | c[b]] (Dc+1),
// out[b][i][j] = A_i * C_j.  bwd: da[b][i-1] = sum_j dout[b][i][j] C_j;  dc[b][j-1] = sum_i dout[b][i][j] A_i  (fixed order).
__global__ __launch_bounds__(256) void k_outer_fwd(const float* __restrict__ a, const float* __restrict__ c, int B, int Da, int Dc,
                                                  float* __restrict__ out) {
  const int64_t n = (int64_t)B * (Da + 1) * (Dc + 1);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % (Dc + 1)); const int64_t r = e / (Dc + 1);
    const int i = (int)(r % (Da + 1)), b = (int)(r / (Da + 1));
    out[e] = (i ? a[(size_t)b * Da + i - 1] : 1.f) * (j ? c[(size_t)b * Dc + j - 1] : 1.f);
  }
}
__global__ __launch_bounds__(256) void k_outer_bwd(const float* __restrict__ a, const float* __restrict__ c, const float* __restrict__ dout,
                                                  int Da, int Dc, float* __restrict__ da, float* __restrict__ dc) {
  const int b = blockIdx.x;
  const float* g = dout + (size_t)b * (Da + 1) * (Dc + 1);
  for (int i = 1 + threadIdx.x; i <= Da; i += 256) {
    float s = g[(size_t)i * (Dc + 1)];
    for (int j = 1; j <= Dc; ++j) s = fmaf(g[(size_t)i * (Dc + 1) + j], c[(size_t)b * Dc + j - 1], s);
    da[(size_t)b * Da + i - 1] = s;
  }
  for (int j = 1 + threadIdx.x; j <= Dc; j += 256) {
    float s = g[j];
    for (int i = 1; i <= Da; ++i) s = fmaf(g[(size_t)i * (Dc + 1) + j], a[(size_t)b * Da + i - 1], s);
    dc[(size_t)b * Dc + j - 1] = s;
  }
}
extern "C" int md_outer_fwd(const float* a, const float* c, int32_t B, int32_t Da, int32_t Dc, float* out, void* stream) {
  if (!a || !c || !out) return MD_ERR_NULL;
  if (B <= 0 || Da <= 0 || Dc <= 0) return MD_ERR_BAD_SHAPE;
  int64_t blocks = ((int64_t)B * (Da + 1) * (Dc + 1) + 255) / 256; if (blocks > 8192) blocks = 8192;
  MD_KLAUNCH(k_outer_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, c, B, Da, Dc, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_outer_bwd(const float* a, const float* c, const float* dout, int32_t B, int32_t Da, int32_t Dc, float* da, float* dc,
                            void* stream) {
  if (!a || !c || !dout || !da || !dc) return MD_ERR_NULL;
  if (B <= 0 || Da <= 0 || Dc <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_outer_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, a, c, dout, Da, Dc, da, dc);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_elu(const float* x, const float* dy, float alpha, int64_t n, float* out, void* stream) {
  if (!x || !out) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  int64_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
  MD_KLAUNCH(k_elu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, alpha, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int64_t rows, int32_t D,
                                    float eps, float* out, float* xhat, float* rstd, float* sum_out, void* stream) {
  if (!a || !gamma || !beta || !out || !xhat || !rstd) return MD_ERR_NULL;
  if (rows <= 0 || D <= 0 || rows > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_add_ln_fwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta, D, eps, out, xhat, rstd, sum_out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" size_t md_add_layernorm_bwd_scratch_floats(int64_t rows, int32_t D) {
  return rows > 1024 ? (size_t)2 * LN_CHUNKS * (size_t)D : 0;
}
extern "C" int md_add_layernorm_bwd(const float* dout, const float* gamma, const float* xhat, const float* rstd, const float* dres,
                                    int64_t rows, int32_t D, float* dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
  if (!dout || !gamma || !xhat || !rstd || !dx || !dgamma || !dbeta) return MD_ERR_NULL;
  if (rows <= 0 || D <= 0 || rows > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  const bool chunked = md_add_layernorm_bwd_scratch_floats(rows, D) != 0;
  if (chunked && !scratch) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  MD_KLAUNCH(k_add_ln_bwd, dim3((unsigned)rows), dim3(256), 0, s, dout, gamma, xhat, rstd, dres, D, dx);
  MD_CHECK_LAUNCH();
  if (!chunked) {
    MD_KLAUNCH(k_ln_param_grad, dim3(md_cdiv(D, 256), 1), dim3(256), 0, s, dout, xhat, (int)rows, D, dgamma, dbeta);
    MD_CHECK_LAUNCH();
  } else {
    float* pg = scratch; float* pb = scratch + (size_t)LN_CHUNKS * D;
    MD_KLAUNCH(k_ln_param_grad, dim3(md_cdiv(D, 256), LN_CHUNKS), dim3(256), 0, s, dout, xhat, (int)rows, D, pg, pb);
    MD_CHECK_LAUNCH();
    MD_KLAUNCH(k_ln_param_sum, dim3(md_cdiv(D, 256)), dim3(256), 0, s, (const float*)pg, (const float*)pb, LN_CHUNKS, D, dgamma, dbeta);
    MD_CHECK_LAUNCH();
  }
  return MD_OK;
}
extern "C" int md_attention_fwd(const float* qkv, const float* mask, const float* drop, int32_t S, int32_t B, int32_t D, int32_t H,
                                int32_t batch_first, float* probs, float* out, void* stream) {
  if (!qkv || !probs || !out) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)ATT_RB * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_attn_fwd, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), lds, (hipStream_t)stream, qkv, mask, drop, S, B, D, H,
             batch_first ? 1 : 0, 1.f / sqrtf((float)(D / H)), probs, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_attention_bwd(const float* qkv, const float* probs, const float* drop, const float* dout, int32_t S, int32_t B,
                                int32_t D, int32_t H, int32_t batch_first, float* dqkv, float* ds_scratch, void* stream) {
  if (!qkv || !probs || !dout || !dqkv || !ds_scratch) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)ATT_RB * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)(D / H));
  const int bf = batch_first ? 1 : 0;
  MD_KLAUNCH(k_attn_bwd_q, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), lds, s, qkv, probs, drop, dout, S, B, D, H, bf, scale, ds_scratch, dqkv);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_attn_bwd_kv, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), 0, s, qkv, probs, drop, dout, (const float*)ds_scratch, S, B, D, H, bf, dqkv);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_gelu(const float* x, const float* dy, int32_t kind, int64_t n, float* out, void* stream) {
  if (!x || !out) return MD_ERR_NULL;
  if (n <= 0 || (kind != 0 && kind != 1)) return MD_ERR_BAD_SHAPE;
  int64_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
  MD_KLAUNCH(k_gelu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, kind, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
