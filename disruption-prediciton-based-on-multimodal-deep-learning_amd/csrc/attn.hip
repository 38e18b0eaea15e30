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
                                                   float* __restrict__ out, float* __restrict__ xhat, float* __restrict__ rstd) {
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
  }
  if (threadIdx.x == 0) rstd[row] = rs;
}
// dx[row] = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dout * gamma   (the same dx goes to both summands)
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ dout, const float* __restrict__ gamma,
                                                   const float* __restrict__ xhat, const float* __restrict__ rstd, int D,
                                                   float* __restrict__ dx) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { const float g = dout[row * D + i] * gamma[i]; s1 += g; s2 += g * xhat[row * D + i]; }
  const float m1 = block_sum(s1, red) / (float)D;
  const float m2 = block_sum(s2, red) / (float)D;
  const float rs = rstd[row];
  for (int i = threadIdx.x; i < D; i += 256) {
    const float g = dout[row * D + i] * gamma[i];
    dx[row * D + i] = rs * (g - m1 - xhat[row * D + i] * m2);
  }
}
// dgamma[i] = sum_rows dout * xhat, dbeta[i] = sum_rows dout   (one thread per feature, rows in order)
__global__ __launch_bounds__(256) void k_ln_param_grad(const float* __restrict__ dout, const float* __restrict__ xhat, int rows, int D,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  float g = 0.f, bsum = 0.f;
  for (int r = 0; r < rows; ++r) { const float d = dout[(size_t)r * D + i]; g = fmaf(d, xhat[(size_t)r * D + i], g); bsum += d; }
  dgamma[i] = g; dbeta[i] = bsum;
}

// Attention core.  qkv [S][B][3*D] (q | k | v as nn.MultiheadAttention's in_proj lays them out), head h uses features
// h*dh .. (h+1)*dh of each part.  probs [B*H][S][S] (after softmax, before dropout) kept for the backward; drop (may be NULL)
// [B*H][S][S] holds mask/keep factors of the attention dropout.  out [S][B][D].  mask [S][S] additive (may be NULL).
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ qkv, const float* __restrict__ mask,
                                                 const float* __restrict__ drop, int S, int B, int D, int H, float scale,
                                                 float* __restrict__ probs, float* __restrict__ out) {
  extern __shared__ float sm[];        // p [S][S]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, dh = D / H, t = threadIdx.x;
  float* p = sm;
  for (int e = t; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    const float* q = qkv + ((size_t)i * B + b) * 3 * D + h * dh;
    const float* k = qkv + ((size_t)j * B + b) * 3 * D + D + h * dh;
    float a = 0.f;
    for (int c = 0; c < dh; ++c) a = fmaf(q[c], k[c], a);
    p[e] = a * scale + (mask ? mask[e] : 0.f);
  }
  __syncthreads();
  for (int i = t; i < S; i += 256) {          // softmax of row i
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) mx = fmaxf(mx, p[i * S + j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) { const float e = expf(p[i * S + j] - mx); p[i * S + j] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int j = 0; j < S; ++j) p[i * S + j] *= inv;
  }
  __syncthreads();
  for (int e = t; e < S * S; e += 256) {
    probs[(size_t)bh * S * S + e] = p[e];
    if (drop) p[e] *= drop[(size_t)bh * S * S + e];
  }
  __syncthreads();
  for (int e = t; e < S * dh; e += 256) {
    const int i = e / dh, c = e - i * dh;
    float a = 0.f;
    for (int j = 0; j < S; ++j) a = fmaf(p[i * S + j], qkv[((size_t)j * B + b) * 3 * D + 2 * D + h * dh + c], a);
    out[((size_t)i * B + b) * D + h * dh + c] = a;
  }
}
// dqkv [S][B][3*D] from dout [S][B][D]
__global__ __launch_bounds__(256) void k_attn_bwd(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                 const float* __restrict__ drop, const float* __restrict__ dout, int S, int B,
                                                 int D, int H, float scale, float* __restrict__ dqkv) {
  extern __shared__ float sm[];        // pd [S][S] (dropped probs), ds [S][S]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, dh = D / H, t = threadIdx.x;
  float* pd = sm; float* ds = sm + S * S;
  const float* pr = probs + (size_t)bh * S * S;
  const float* dr = drop ? drop + (size_t)bh * S * S : nullptr;
  for (int e = t; e < S * S; e += 256) pd[e] = pr[e] * (dr ? dr[e] : 1.f);
  __syncthreads();
  // dv[j][c] = sum_i pd[i][j] * dout[i][c]
  for (int e = t; e < S * dh; e += 256) {
    const int j = e / dh, c = e - j * dh;
    float a = 0.f;
    for (int i = 0; i < S; ++i) a = fmaf(pd[i * S + j], dout[((size_t)i * B + b) * D + h * dh + c], a);
    dqkv[((size_t)j * B + b) * 3 * D + 2 * D + h * dh + c] = a;
  }
  // dP[i][j] = (sum_c dout[i][c] v[j][c]) * drop;  dS = P * (dP - sum_j dP*P)
  for (int e = t; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    float a = 0.f;
    for (int c = 0; c < dh; ++c) a = fmaf(dout[((size_t)i * B + b) * D + h * dh + c], qkv[((size_t)j * B + b) * 3 * D + 2 * D + h * dh + c], a);
    ds[e] = a * (dr ? dr[e] : 1.f);
  }
  __syncthreads();
  for (int i = t; i < S; i += 256) {
    float dot = 0.f;
    for (int j = 0; j < S; ++j) dot = fmaf(ds[i * S + j], pr[i * S + j], dot);
    for (int j = 0; j < S; ++j) ds[i * S + j] = pr[i * S + j] * (ds[i * S + j] - dot) * scale;
  }
  __syncthreads();
  for (int e = t; e < S * dh; e += 256) {       // dq[i][c] = sum_j ds[i][j] k[j][c];  dk[j][c] = sum_i ds[i][j] q[i][c]
    const int i = e / dh, c = e - i * dh;
    float aq = 0.f, ak = 0.f;
    for (int j = 0; j < S; ++j) {
      aq = fmaf(ds[i * S + j], qkv[((size_t)j * B + b) * 3 * D + D + h * dh + c], aq);
      ak = fmaf(ds[j * S + i], qkv[((size_t)j * B + b) * 3 * D + h * dh + c], ak);
    }
    dqkv[((size_t)i * B + b) * 3 * D + h * dh + c] = aq;
    dqkv[((size_t)i * B + b) * 3 * D + D + h * dh + c] = ak;
  }
}

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

extern "C" int md_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int64_t rows, int32_t D,
                                    float eps, float* out, float* xhat, float* rstd, void* stream) {
  if (!a || !gamma || !beta || !out || !xhat || !rstd) return MD_ERR_NULL;
  if (rows <= 0 || D <= 0 || rows > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_add_ln_fwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta, D, eps, out, xhat, rstd);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_add_layernorm_bwd(const float* dout, const float* gamma, const float* xhat, const float* rstd, int64_t rows,
                                    int32_t D, float* dx, float* dgamma, float* dbeta, void* stream) {
  if (!dout || !gamma || !xhat || !rstd || !dx || !dgamma || !dbeta) return MD_ERR_NULL;
  if (rows <= 0 || D <= 0 || rows > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  MD_KLAUNCH(k_add_ln_bwd, dim3((unsigned)rows), dim3(256), 0, s, dout, gamma, xhat, rstd, D, dx);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_ln_param_grad, dim3(md_cdiv(D, 256)), dim3(256), 0, s, dout, xhat, (int)rows, D, dgamma, dbeta);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_attention_fwd(const float* qkv, const float* mask, const float* drop, int32_t S, int32_t B, int32_t D, int32_t H,
                                float* probs, float* out, void* stream) {
  if (!qkv || !probs || !out) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)S * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_attn_fwd, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, mask, drop, S, B, D, H,
             1.f / sqrtf((float)(D / H)), probs, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_attention_bwd(const float* qkv, const float* probs, const float* drop, const float* dout, int32_t S, int32_t B,
                                int32_t D, int32_t H, float* dqkv, void* stream) {
  if (!qkv || !probs || !dout || !dqkv) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)2 * S * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_attn_bwd, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, probs, drop, dout, S, B, D, H,
             1.f / sqrtf((float)(D / H)), dqkv);
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
