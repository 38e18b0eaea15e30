// Pieces of the Transformer 0D encoder (src/models/transformer.py, nn.TransformerEncoderLayer post-norm form; SURVEY 8a row
// a12): fused residual-add + LayerNorm, the attention core of nn.MultiheadAttention with an additive mask, and the two GELU
// forms the reference uses.  Sequences are tiny (S = 21 tokens, d_model 128-256, head dim 16-32): one workgroup per row /
// per (batch, head); everything fixed-order fp32.
#include "common.h"
#include "patch_common.h"
#include "philox.h"
#include <map>
#include <mutex>

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
// dgamma[i] = sum_rows dout * xhat, dbeta[i] = sum_rows dout.  Workgroup = 64 features x 4 row lanes over one chunk of rows
// (blockIdx.y; every lane walks its rows in order, the four lanes are added in a fixed order); with more than one chunk the
// outputs are per-chunk partials [chunks][D] that k_ln_param_sum adds in a fixed tree.
#define LN_CHUNKS 256
__global__ __launch_bounds__(256) void k_ln_param_grad(const float* __restrict__ dout, const float* __restrict__ xhat, int rows, int D,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, i = blockIdx.x * 64 + c;
  const int per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float g = 0.f, bsum = 0.f;
  if (i < D)
    for (int r = r0 + rl; r < r1; r += 4) { const float d = dout[(size_t)r * D + i]; g = fmaf(d, xhat[(size_t)r * D + i], g); bsum += d; }
  red[0][rl][c] = g; red[1][rl][c] = bsum;
  __syncthreads();
  if (rl == 0 && i < D) {
    dgamma[(size_t)blockIdx.y * D + i] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    dbeta[(size_t)blockIdx.y * D + i] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}
// 16 features x 16 chunk lanes per workgroup: lane l adds chunks l, l + 16, ... in order, then the 16 lanes in order
__global__ __launch_bounds__(256) void k_ln_param_sum(const float* __restrict__ pg, const float* __restrict__ pb, int chunks, int D,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][16][16];
  const int c = threadIdx.x & 15, l = threadIdx.x >> 4, i = blockIdx.x * 16 + c;
  float g = 0.f, b = 0.f;
  if (i < D)
    for (int k = l; k < chunks; k += 16) { g += pg[(size_t)k * D + i]; b += pb[(size_t)k * D + i]; }
  red[0][l][c] = g; red[1][l][c] = b;
  __syncthreads();
  if (l == 0 && i < D) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sg += red[0][k][c]; sb += red[1][k][c]; }
    dgamma[i] = sg; dbeta[i] = sb;
  }
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

// ------------------------------------------------------------------------------------------------------------------------
// Matrix-core attention (exact fp32, v_mfma_f32_16x16x4_f32) for the unmasked, dropout-free case of ViViT's Attention
// (ViViT.py:69-88; BASELINE cfg3: 84 sequences x 4 heads x 197 tokens x d_head 64).  S <= 256, d_head in {16, 32, 64}.
// A wave owns 16 query (or key) rows; a workgroup (4 waves) 64.  Operand mapping as in conv_gemm.hip: lane (li = lane&15,
// lg = lane>>4) supplies A[row li][k] and B[k][col li] with k = 4*lg + e for MFMA e of a group of 4 (a permutation of the K axis,
// the same on both sides); the result tile holds D[row 4*lg + r][col li] in register r.
//   scores (16 x S) = Qfrag . K^T : K rows staged 64 at a time in LDS (pitch 72: conflict-free ds_read_b128), all 16 x S
//     scores live in <= 16 accumulator tiles per wave, so the softmax is done in registers (row = (lg, r): reduce over tiles and
//     over the 16 lanes of the group);
//   out (16 x dh) = P . V : P goes through a per-wave LDS buffer to turn the result layout into the A layout; V rows staged 64
//     at a time (pitch 68: conflict-free ds_read_b32 for the four rows 4*lg + e).
// Backward pass A has the same shape with (dO, V) in the first product and (dS, K) in the second; pass B (per 16 keys) reads
// dS^T / P^T straight from global as A operands and stages Q / dO.
#define AT_KP 72
#define AT_VP 68
#define AT_MAXT 16
typedef const float* cfp;

template <int DH, int P>
__device__ __forceinline__ void at_stage(float* tile, cfp base, int width, int r0, int S, int B, int b, int bf) {
  constexpr int CH = DH / 4;
#pragma unroll
  for (int k = 0; k < 64 * CH / 256; ++k) {
    const int e = threadIdx.x + 256 * k, r = e / CH, c = e - r * CH, i = r0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < S) v = *(const float4*)(base + (bf ? (size_t)b * S + i : (size_t)i * B + b) * width + c * 4);
    *(float4*)(tile + r * P + c * 4) = v;
  }
}
// the same staging in two halves, so that the global loads of the next 64 rows are in flight while the current tile is used
template <int DH> struct AtRegs { float4 v[64 * (DH / 4) / 256]; };
template <int DH>
__device__ __forceinline__ void at_fetch(AtRegs<DH>& rg, cfp base, int width, int r0, int S, int B, int b, int bf) {
  constexpr int CH = DH / 4;
#pragma unroll
  for (int k = 0; k < 64 * CH / 256; ++k) {
    const int e = threadIdx.x + 256 * k, r = e / CH, c = e - r * CH, i = r0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < S) v = *(const float4*)(base + (bf ? (size_t)b * S + i : (size_t)i * B + b) * width + c * 4);
    rg.v[k] = v;
  }
}
// acc[t] (t < nt) = afrag (16 x DH) . rows^T for the rows of `base` (staged 64 at a time)
template <int DH>
__device__ __forceinline__ void at_scores(f32x4 (&acc)[AT_MAXT], const f32x4 (&a)[DH / 16], float* tile, cfp base, int width, int S,
                                          int B, int b, int bf, int nt, int li, int lg) {
#pragma unroll
  for (int kt = 0; kt < AT_MAXT / 4; ++kt) {
    if (kt * 4 < nt) {
      __syncthreads();
      at_stage<DH, AT_KP>(tile, base, width, kt * 64, S, B, b, bf);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kt * 4 + j < nt) {
#pragma unroll
          for (int s = 0; s < DH / 16; ++s) {
            const f32x4 bv = *(const f32x4*)(tile + (16 * j + li) * AT_KP + (s * 4 + lg) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[kt * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][e], bv[e], acc[kt * 4 + j], 0, 0, 0);
          }
        }
      }
    }
  }
}
// o (16 x DH) = Pw (16 x nt*16, per-wave LDS buffer, pitch pp) . rows of `base` (staged 64 at a time)
template <int DH>
__device__ __forceinline__ void at_apply(f32x4 (&o)[DH / 16], const float* pw, int pp, float* tile, cfp base, int width, int S, int B,
                                         int b, int bf, int nt, int li, int lg) {
#pragma unroll
  for (int jn = 0; jn < DH / 16; ++jn) o[jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt * 4 < nt; ++kt) {
    __syncthreads();
    at_stage<DH, AT_VP>(tile, base, width, kt * 64, S, B, b, bf);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (kt * 4 + ks < nt) {
        const f32x4 av = *(const f32x4*)(pw + li * pp + kt * 64 + ks * 16 + lg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* row = tile + (ks * 16 + lg * 4 + e) * AT_VP + li;
#pragma unroll
          for (int jn = 0; jn < DH / 16; ++jn) o[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], row[jn * 16], o[jn], 0, 0, 0);
        }
      }
    }
  }
}
// Staged rows as two 16-bit images (hi, lo: [64][DH] halves, row pitch AT_SPB bytes, the lo image AT_SLO bytes after the hi one):
// split ONCE per workgroup while staging, so the four waves read ready B operands (16 bytes per lane, k = 8 lg .. 8 lg + 7).
// Same LDS footprint as the fp32 tile (64 x 72 floats).
#define AT_SPB(DH_) ((DH_) * 2 + 16)
#define AT_SLO(DH_) (64 * AT_SPB(DH_))
template <bool F16>
__device__ __forceinline__ void at_split4(float4 v, uint2& h, uint2& l) {
  if (F16) {
    h.x = pk_f16(v.x, v.y); h.y = pk_f16(v.z, v.w);
    const f16x2 a = __builtin_bit_cast(f16x2, h.x), b = __builtin_bit_cast(f16x2, h.y);
    l.x = pk_f16(v.x - (float)a[0], v.y - (float)a[1]); l.y = pk_f16(v.z - (float)b[0], v.w - (float)b[1]);
  } else {
    h.x = pk_bf16(v.x, v.y); h.y = pk_bf16(v.z, v.w);
    l.x = pk_bf16(v.x - __builtin_bit_cast(float, h.x << 16), v.y - __builtin_bit_cast(float, h.x & 0xffff0000u));
    l.y = pk_bf16(v.z - __builtin_bit_cast(float, h.y << 16), v.w - __builtin_bit_cast(float, h.y & 0xffff0000u));
  }
}
template <int DH, bool F16>
__device__ __forceinline__ void at_commit_split(char* img, const AtRegs<DH>& rg) {
  constexpr int CH = DH / 4;
#pragma unroll
  for (int k = 0; k < 64 * CH / 256; ++k) {
    const int e = threadIdx.x + 256 * k, r = e / CH, c = e - r * CH;
    uint2 h, l;
    at_split4<F16>(rg.v[k], h, l);
    *(uint2*)(img + r * AT_SPB(DH) + c * 8) = h;
    *(uint2*)(img + AT_SLO(DH) + r * AT_SPB(DH) + c * 8) = l;
  }
}
// Transposing LDS read (ds_read_b64_tr_b16) of a [row][column] 16-bit image: within a 16-lane group, lane 4 lq + lp addresses the
// 8-byte chunk (columns 4 lp .. 4 lp + 3) of row lq, and receives column (its own index in the group) of rows 0..3.  Two reads give
// the 8 reduction slots of one v_mfma_f32_16x16x32 operand: slots 0-3 = rows r .. r + 3, slots 4-7 = rows r + 16 .. r + 19.
typedef short at_s16x4 __attribute__((ext_vector_type(4)));
typedef short at_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint4 at_tr2(const char* p0, const char* p1) {
  const at_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s16x4 __attribute__((address_space(3)))*)p0);
  const at_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_s16x4 __attribute__((address_space(3)))*)p1);
  at_s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(uint4, r);
}
// ---- split-precision form of the two products (every fp32 product as three 16-bit MFMAs on hi + lo halves, exactly as in the
// convolution kernels: fp16 halves in the forward pass, bf16 in the backward pass; v_mfma_f32_16x16x32: 5.3x the rate of the fp32
// instruction for the same contraction).  The LDS tiles stay fp32; a lane reads the 8 values it feeds to one instruction (k = 8 lg
// .. 8 lg + 7 of a 32-wide k block, the same permutation on both sides) and splits them in registers.
template <bool F16>
__device__ __forceinline__ void at_split8(const float* v, uint4& h, uint4& l) {
  if (F16) split8_f16(v, h, l); else split8(v, h, l);
}
template <bool F16>
__device__ __forceinline__ f32x4 at_mma3(uint4 ah, uint4 al, uint4 bh, uint4 bl, f32x4 c) {
  c = mma<F16>(ah, bh, c);
  c = mma<F16>(ah, bl, c);
  return mma<F16>(al, bh, c);
}
// acc[t] (t < nt) = afrag (16 x DH, pre-split per 32-wide k block) . rows^T
template <int DH, bool F16>
__device__ __forceinline__ void at_scores_sp(f32x4 (&acc)[AT_MAXT], const uint4 (&ah)[DH / 32], const uint4 (&al)[DH / 32], float* tile,
                                             cfp base, int width, int S, int B, int b, int bf, int nt, int li, int lg) {
  AtRegs<DH> rg;
  at_fetch<DH>(rg, base, width, 0, S, B, b, bf);
#pragma unroll
  for (int kt = 0; kt < AT_MAXT / 4; ++kt) {
    if (kt * 4 < nt) {
      __syncthreads();
      at_commit_split<DH, F16>((char*)tile, rg);
      __syncthreads();
      if ((kt + 1) * 4 < nt) at_fetch<DH>(rg, base, width, (kt + 1) * 64, S, B, b, bf);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kt * 4 + j < nt) {
#pragma unroll
          for (int s = 0; s < DH / 32; ++s) {
            const char* p = (const char*)tile + (16 * j + li) * AT_SPB(DH) + s * 64 + lg * 16;
            const uint4 bh = *(const uint4*)p, bl = *(const uint4*)(p + AT_SLO(DH));
            acc[kt * 4 + j] = at_mma3<F16>(ah[s], al[s], bh, bl, acc[kt * 4 + j]);
          }
        }
      }
    }
  }
}
// o (16 x DH) = Pw (16 x 32*ceil(nt/2), per-wave LDS buffer, pitch pp, columns past the last tile zero) . rows of `base`
// (`rg`: rows 0..63 of `base`, fetched by the caller -- before its softmax / dS arithmetic, which hides that latency)
template <int DH, bool F16>
__device__ __forceinline__ void at_apply_sp(f32x4 (&o)[DH / 16], const float* pw, int pp, float* tile, AtRegs<DH>& rg, cfp base, int width,
                                            int S, int B, int b, int bf, int nt, int li, int lg) {
  const int lq = li >> 2, lp = li & 3;
#pragma unroll
  for (int jn = 0; jn < DH / 16; ++jn) o[jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt * 4 < nt; ++kt) {
    __syncthreads();
    at_commit_split<DH, F16>((char*)tile, rg);
    __syncthreads();
    if ((kt + 1) * 4 < nt) at_fetch<DH>(rg, base, width, (kt + 1) * 64, S, B, b, bf);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kt * 4 + kb * 2 < nt) {
        // reduction slots of this 32-key block: lane group lg takes keys 4 lg .. 4 lg + 3 and 16 + 4 lg .. 16 + 4 lg + 3
        const float* pa = pw + li * pp + kt * 64 + kb * 32 + lg * 4;
        float v[8];
        *(float4*)v = *(const float4*)pa; *(float4*)(v + 4) = *(const float4*)(pa + 16);
        uint4 ph, pl;
        at_split8<F16>(v, ph, pl);
        const char* r0 = (const char*)tile + (kb * 32 + lg * 4 + lq) * AT_SPB(DH) + lp * 8;
        const char* r1 = r0 + 16 * AT_SPB(DH);
#pragma unroll
        for (int jn = 0; jn < DH / 16; ++jn) {
          const uint4 vh = at_tr2(r0 + jn * 32, r1 + jn * 32);
          const uint4 vl = at_tr2(r0 + AT_SLO(DH) + jn * 32, r1 + AT_SLO(DH) + jn * 32);
          o[jn] = at_mma3<F16>(ph, pl, vh, vl, o[jn]);
        }
      }
    }
  }
}
__device__ __forceinline__ float at_group_max(float v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float at_group_sum(float v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// LSE (split form only): `probs` receives one value per query row, max + log(sum) of its scores, instead of the S x S matrix --
// the backward kernels k_attn_lse_bwd_* recompute P from q, k and that value.
template <int DH, bool SP = false, bool LSE = false>
__global__ __launch_bounds__(256) void k_attn_mfma_fwd(cfp qkv, int S, int B, int D, int H, int bf, float scale, int pp,
                                                      float* __restrict__ probs, float* __restrict__ out) {
  extern __shared__ float sm[];
  float* tile = sm;                                  // 64 x 72
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  float* pw = sm + 64 * AT_KP + wave * 16 * pp;      // per-wave 16 x pp
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, nt = (S + 15) >> 4;
  const int q0 = blockIdx.y * 64 + wave * 16;
  f32x4 acc[AT_MAXT];
#pragma unroll
  for (int t = 0; t < AT_MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  AtRegs<DH> vpre;                                   // (split form) first 64 rows of the second product's staged operand
  if constexpr (SP) {
    uint4 ah[DH / 32], al[DH / 32];
    const int i = q0 + li;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (i < S) {
        const float* p = qkv + (bf ? (size_t)b * S + i : (size_t)i * B + b) * 3 * D + h * DH + s * 32 + lg * 8;
        *(float4*)v = *(const float4*)p; *(float4*)(v + 4) = *(const float4*)(p + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= scale;
      at_split8<true>(v, ah[s], al[s]);
    }
    at_scores_sp<DH, true>(acc, ah, al, tile, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
    at_fetch<DH>(vpre, qkv + 2 * D + h * DH, 3 * D, 0, S, B, b, bf);
  } else {
    f32x4 a[DH / 16];
    const int i = q0 + li;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < S) v = *(const f32x4*)(qkv + (bf ? (size_t)b * S + i : (size_t)i * B + b) * 3 * D + h * DH + (s * 4 + lg) * 4);
      a[s] = v * scale;
    }
    at_scores<DH>(acc, a, tile, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  }
  // softmax over the key axis (columns 16 t + li), one row per (lg, r)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) if (t < nt && 16 * t + li < S) mx = fmaxf(mx, acc[t][r]);
    mx = at_group_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      if (t < nt) { const float e = (16 * t + li < S) ? expf(acc[t][r] - mx) : 0.f; acc[t][r] = e; sum += e; }
    }
    const float tot = at_group_sum(sum);
    const float inv = 1.f / tot;
    const int i = q0 + 4 * lg + r;
    if (LSE && li == 0 && i < S) probs[(size_t)bh * S + i] = mx + logf(tot);
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      if (t < nt) {
        const float pv = acc[t][r] * inv;
        pw[(4 * lg + r) * pp + 16 * t + li] = pv;
        if (!LSE && i < S && 16 * t + li < S) probs[((size_t)bh * S + i) * S + 16 * t + li] = pv;
      }
    }
  }
  f32x4 o[DH / 16];
  if constexpr (SP) {
    if (nt & 1) {                                     // the second half of the last 32-key block
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(4 * lg + r) * pp + 16 * nt + li] = 0.f;
    }
    at_apply_sp<DH, true>(o, pw, pp, tile, vpre, qkv + 2 * D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  } else {
    at_apply<DH>(o, pw, pp, tile, qkv + 2 * D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * lg + r;
    if (i < S) {
      float* dst = out + (bf ? (size_t)b * S + i : (size_t)i * B + b) * D + h * DH + li;
#pragma unroll
      for (int jn = 0; jn < DH / 16; ++jn) dst[jn * 16] = o[jn][r];
    }
  }
}

// pass A: dP = dO V^T; dS = P (dP - rowsum(dP P)) scale -> ds_out; dq = dS K
template <int DH, bool SP = false>
__global__ __launch_bounds__(256) void k_attn_mfma_bwd_q(cfp qkv, cfp probs, cfp dout, int S, int B, int D, int H, int bf, float scale,
                                                        int pp, float* __restrict__ ds_out, float* __restrict__ dqkv) {
  extern __shared__ float sm[];
  float* tile = sm;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  float* pw = sm + 64 * AT_KP + wave * 16 * pp;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, nt = (S + 15) >> 4;
  const int q0 = blockIdx.y * 64 + wave * 16;
  f32x4 acc[AT_MAXT];
#pragma unroll
  for (int t = 0; t < AT_MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  AtRegs<DH> vpre;
  if constexpr (SP) {
    uint4 ah[DH / 32], al[DH / 32];
    const int i = q0 + li;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (i < S) {
        const float* p = dout + (bf ? (size_t)b * S + i : (size_t)i * B + b) * D + h * DH + s * 32 + lg * 8;
        *(float4*)v = *(const float4*)p; *(float4*)(v + 4) = *(const float4*)(p + 4);
      }
      at_split8<false>(v, ah[s], al[s]);
    }
    at_scores_sp<DH, false>(acc, ah, al, tile, qkv + 2 * D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
    at_fetch<DH>(vpre, qkv + D + h * DH, 3 * D, 0, S, B, b, bf);
  } else {
    f32x4 a[DH / 16];
    const int i = q0 + li;
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < S) v = *(const f32x4*)(dout + (bf ? (size_t)b * S + i : (size_t)i * B + b) * D + h * DH + (s * 4 + lg) * 4);
      a[s] = v;
    }
    at_scores<DH>(acc, a, tile, qkv + 2 * D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * lg + r;
    float pr[AT_MAXT];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      pr[t] = 0.f;
      if (t < nt) {
        if (i < S && 16 * t + li < S) pr[t] = probs[((size_t)bh * S + i) * S + 16 * t + li];
        dot = fmaf(acc[t][r], pr[t], dot);
      }
    }
    dot = at_group_sum(dot);
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      if (t < nt) {
        const float dsv = pr[t] * (acc[t][r] - dot) * scale;
        pw[(4 * lg + r) * pp + 16 * t + li] = dsv;
        if (i < S && 16 * t + li < S) ds_out[((size_t)bh * S + i) * S + 16 * t + li] = dsv;
      }
    }
  }
  f32x4 o[DH / 16];
  if constexpr (SP) {
    if (nt & 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(4 * lg + r) * pp + 16 * nt + li] = 0.f;
    }
    at_apply_sp<DH, false>(o, pw, pp, tile, vpre, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  } else {
    at_apply<DH>(o, pw, pp, tile, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * lg + r;
    if (i < S) {
      float* dst = dqkv + (bf ? (size_t)b * S + i : (size_t)i * B + b) * 3 * D + h * DH + li;
#pragma unroll
      for (int jn = 0; jn < DH / 16; ++jn) dst[jn * 16] = o[jn][r];
    }
  }
}
// pass B (per 16 keys j): dk[j] = sum_i dS[i][j] q[i];  dv[j] = sum_i P[i][j] dO[i]
template <int DH, bool SP = false>
__global__ __launch_bounds__(256) void k_attn_mfma_bwd_kv(cfp qkv, cfp probs, cfp dout, cfp ds, int S, int B, int D, int H, int bf,
                                                         float* __restrict__ dqkv) {
  extern __shared__ float sm[];
  float* tq = sm;                      // 64 x 68
  float* td = sm + 64 * AT_VP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, nt = (S + 15) >> 4;
  const int j0 = blockIdx.y * 64 + wave * 16, j = j0 + li;
  f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
  for (int jn = 0; jn < DH / 16; ++jn) { dk[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float* dsb = ds + (size_t)bh * S * S;
  const float* pb = probs + (size_t)bh * S * S;
  for (int it = 0; it * 4 < nt; ++it) {
    __syncthreads();
    at_stage<DH, AT_VP>(tq, qkv + h * DH, 3 * D, it * 64, S, B, b, bf);
    at_stage<DH, AT_VP>(td, dout + h * DH, D, it * 64, S, B, b, bf);
    __syncthreads();
    if constexpr (SP) {
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        if (it * 4 + ib * 2 < nt) {
          float ads[8], ap[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int i = it * 64 + ib * 32 + lg * 8 + e;
            const bool ok = i < S && j < S;
            ads[e] = ok ? dsb[(size_t)i * S + j] : 0.f;
            ap[e] = ok ? pb[(size_t)i * S + j] : 0.f;
          }
          uint4 dsh, dsl, ph, pl;
          at_split8<false>(ads, dsh, dsl);
          at_split8<false>(ap, ph, pl);
#pragma unroll
          for (int jn = 0; jn < DH / 16; ++jn) {
            float wq[8], wd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              wq[e] = tq[(ib * 32 + lg * 8 + e) * AT_VP + jn * 16 + li];
              wd[e] = td[(ib * 32 + lg * 8 + e) * AT_VP + jn * 16 + li];
            }
            uint4 qh, ql, oh, ol;
            at_split8<false>(wq, qh, ql);
            at_split8<false>(wd, oh, ol);
            dk[jn] = at_mma3<false>(dsh, dsl, qh, ql, dk[jn]);
            dv[jn] = at_mma3<false>(ph, pl, oh, ol, dv[jn]);
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int is = 0; is < 4; ++is) {
      if (it * 4 + is < nt) {
        float ads[4], ap[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = it * 64 + is * 16 + lg * 4 + e;
          const bool ok = i < S && j < S;
          ads[e] = ok ? dsb[(size_t)i * S + j] : 0.f;
          ap[e] = ok ? pb[(size_t)i * S + j] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* rq = tq + (is * 16 + lg * 4 + e) * AT_VP + li;
          const float* rd = td + (is * 16 + lg * 4 + e) * AT_VP + li;
#pragma unroll
          for (int jn = 0; jn < DH / 16; ++jn) {
            dk[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(ads[e], rq[jn * 16], dk[jn], 0, 0, 0);
            dv[jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[e], rd[jn * 16], dv[jn], 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jj = j0 + 4 * lg + r;
    if (jj < S) {
      float* dst = dqkv + (bf ? (size_t)b * S + jj : (size_t)jj * B + b) * 3 * D + h * DH + li;
#pragma unroll
      for (int jn = 0; jn < DH / 16; ++jn) { dst[D + jn * 16] = dk[jn][r]; dst[2 * D + jn * 16] = dv[jn][r]; }
    }
  }
}
// ---- backward without the S x S matrices: P is recomputed from q, k and the forward pass's per-row log-sum-exp (the same fp16
// split products as the forward pass), so neither P nor dS travels through memory (cfg3: 52 MB each, per attention call).
// pass A per 16 queries: S = (q scale) K^T, dP = dO V^T, P = exp(S - lse), delta = rowsum(P dP) -> delta_out, dS = P (dP - delta) scale,
// dq = dS K.
#define AT_TP 72          // pitch of the per-wave 16 x 64 transposed blocks of pass B
template <int DH>
__global__ __launch_bounds__(256) void k_attn_lse_bwd_q(cfp qkv, cfp lse, cfp dout, int S, int B, int D, int H, int bf, float scale, int pp,
                                                       float* __restrict__ delta_out, float* __restrict__ dqkv) {
  extern __shared__ float sm[];
  float* tile = sm;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  float* pw = sm + 64 * AT_KP + wave * 16 * pp;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, nt = (S + 15) >> 4;
  const int q0 = blockIdx.y * 64 + wave * 16;
  f32x4 acs[AT_MAXT], acd[AT_MAXT];
#pragma unroll
  for (int t = 0; t < AT_MAXT; ++t) { acs[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acd[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  {
    uint4 ah[DH / 32], al[DH / 32];
    const int i = q0 + li;
    const size_t row = bf ? (size_t)b * S + i : (size_t)i * B + b;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (i < S) {
        const float* p = qkv + row * 3 * D + h * DH + s * 32 + lg * 8;
        *(float4*)v = *(const float4*)p; *(float4*)(v + 4) = *(const float4*)(p + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= scale;
      at_split8<true>(v, ah[s], al[s]);
    }
    at_scores_sp<DH, true>(acs, ah, al, tile, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (i < S) {
        const float* p = dout + row * D + h * DH + s * 32 + lg * 8;
        *(float4*)v = *(const float4*)p; *(float4*)(v + 4) = *(const float4*)(p + 4);
      }
      at_split8<false>(v, ah[s], al[s]);
    }
    at_scores_sp<DH, false>(acd, ah, al, tile, qkv + 2 * D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
  }
  AtRegs<DH> vpre;
  at_fetch<DH>(vpre, qkv + D + h * DH, 3 * D, 0, S, B, b, bf);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * lg + r;
    const float l = i < S ? lse[(size_t)bh * S + i] : 0.f;
    float pr[AT_MAXT];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      pr[t] = 0.f;
      if (t < nt) {
        if (i < S && 16 * t + li < S) pr[t] = expf(acs[t][r] - l);
        dot = fmaf(acd[t][r], pr[t], dot);
      }
    }
    dot = at_group_sum(dot);
    if (li == 0 && i < S) delta_out[(size_t)bh * S + i] = dot;
#pragma unroll
    for (int t = 0; t < AT_MAXT; ++t) {
      if (t < nt) pw[(4 * lg + r) * pp + 16 * t + li] = pr[t] * (acd[t][r] - dot) * scale;
    }
  }
  if (nt & 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[(4 * lg + r) * pp + 16 * nt + li] = 0.f;
  }
  f32x4 o[DH / 16];
  at_apply_sp<DH, false>(o, pw, pp, tile, vpre, qkv + D + h * DH, 3 * D, S, B, b, bf, nt, li, lg);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q0 + 4 * lg + r;
    if (i < S) {
      float* dst = dqkv + (bf ? (size_t)b * S + i : (size_t)i * B + b) * 3 * D + h * DH + li;
#pragma unroll
      for (int jn = 0; jn < DH / 16; ++jn) dst[jn * 16] = o[jn][r];
    }
  }
}
// pass B per 16 keys j: for every block of 64 queries, S^T = K (q scale)^T and dP^T = V dO^T for the wave's keys (the staged
// query / dO rows are the B operands), P^T = exp(S^T - lse_i), dS^T = P^T (dP^T - delta_i) scale; both go through the wave's LDS
// blocks to become A operands of dv += P^T dO and dk += dS^T q.
template <int DH>
__global__ __launch_bounds__(256) void k_attn_lse_bwd_kv(cfp qkv, cfp lse, cfp delta, cfp dout, int S, int B, int D, int H, int bf, float scale,
                                                        float* __restrict__ dqkv) {
  extern __shared__ float sm[];
  float* tq = sm;                      // 64 x AT_VP fp32 (fp16 halves for S^T, bf16 halves for dk: split per use)
  char* td = (char*)(sm + 64 * AT_VP);  // dO rows as bf16 hi | lo images: natural reads for dP^T, transposing reads for dv
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  float* pT = sm + 64 * AT_VP + 2 * AT_SLO(64) / 4 + wave * 2 * 16 * AT_TP;
  float* dsT = pT + 16 * AT_TP;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H, nt = (S + 15) >> 4;
  const int j0 = blockIdx.y * 64 + wave * 16;
  uint4 kh[DH / 32], kl[DH / 32], vh[DH / 32], vl[DH / 32];
  {
    const int j = j0 + li;
    const size_t row = bf ? (size_t)b * S + j : (size_t)j * B + b;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (j < S) {
        const float* p = qkv + row * 3 * D + D + h * DH + s * 32 + lg * 8;
        *(float4*)v = *(const float4*)p; *(float4*)(v + 4) = *(const float4*)(p + 4);
        *(float4*)w = *(const float4*)(p + D); *(float4*)(w + 4) = *(const float4*)(p + D + 4);
      }
      at_split8<true>(v, kh[s], kl[s]);
      at_split8<false>(w, vh[s], vl[s]);
    }
  }
  f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
  for (int jn = 0; jn < DH / 16; ++jn) { dk[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[jn] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int it = 0; it * 4 < nt; ++it) {
    __syncthreads();
    at_stage<DH, AT_VP>(tq, qkv + h * DH, 3 * D, it * 64, S, B, b, bf);
    {
      AtRegs<DH> rd;
      at_fetch<DH>(rd, dout + h * DH, D, it * 64, S, B, b, bf);
      at_commit_split<DH, false>(td, rd);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int i = it * 64 + 16 * t + li;          // this lane's query column
      if (it * 4 + t < nt) {
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          const float* pq = tq + (16 * t + li) * AT_VP + s * 32 + lg * 8;
          const char* pd = td + (16 * t + li) * AT_SPB(DH) + s * 64 + lg * 16;
          float v[8];
          *(float4*)v = *(const float4*)pq; *(float4*)(v + 4) = *(const float4*)(pq + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= scale;
          uint4 qh, ql;
          at_split8<true>(v, qh, ql);
          const uint4 oh = *(const uint4*)pd, ol = *(const uint4*)(pd + AT_SLO(DH));
          st = at_mma3<true>(kh[s], kl[s], qh, ql, st);
          dp = at_mma3<false>(vh[s], vl[s], oh, ol, dp);
        }
      }
      const bool iv = i < S;
      const float l = iv ? lse[(size_t)bh * S + i] : 0.f, de = iv ? delta[(size_t)bh * S + i] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = iv && j0 + 4 * lg + r < S;
        const float pv = ok ? expf(st[r] - l) : 0.f;
        pT[(4 * lg + r) * AT_TP + 16 * t + li] = pv;
        dsT[(4 * lg + r) * AT_TP + 16 * t + li] = pv * (dp[r] - de) * scale;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (it * 4 + kb * 2 < nt) {
        // dv: reduction slots in the transposing reads' order (queries 4 lg .. + 3 and 16 + 4 lg .. + 3 of the block); dk: 8 lg .. + 7
        float ap[8], ads[8];
        const float* pp_ = pT + li * AT_TP + kb * 32 + lg * 4;
        const float* pd_ = dsT + li * AT_TP + kb * 32 + lg * 8;
        *(float4*)ap = *(const float4*)pp_; *(float4*)(ap + 4) = *(const float4*)(pp_ + 16);
        *(float4*)ads = *(const float4*)pd_; *(float4*)(ads + 4) = *(const float4*)(pd_ + 4);
        uint4 ph, pl, dsh, dsl;
        at_split8<false>(ap, ph, pl);
        at_split8<false>(ads, dsh, dsl);
        const char* r0 = td + (kb * 32 + lg * 4 + lq) * AT_SPB(DH) + lp * 8;
        const char* r1 = r0 + 16 * AT_SPB(DH);
#pragma unroll
        for (int jn = 0; jn < DH / 16; ++jn) {
          float wq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wq[e] = tq[(kb * 32 + lg * 8 + e) * AT_VP + jn * 16 + li];
          uint4 qh, ql;
          at_split8<false>(wq, qh, ql);
          const uint4 oh = at_tr2(r0 + jn * 32, r1 + jn * 32);
          const uint4 ol = at_tr2(r0 + AT_SLO(DH) + jn * 32, r1 + AT_SLO(DH) + jn * 32);
          dk[jn] = at_mma3<false>(dsh, dsl, qh, ql, dk[jn]);
          dv[jn] = at_mma3<false>(ph, pl, oh, ol, dv[jn]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jj = j0 + 4 * lg + r;
    if (jj < S) {
      float* dst = dqkv + (bf ? (size_t)b * S + jj : (size_t)jj * B + b) * 3 * D + h * DH + li;
#pragma unroll
      for (int jn = 0; jn < DH / 16; ++jn) { dst[D + jn * 16] = dk[jn][r]; dst[2 * D + jn * 16] = dv[jn][r]; }
    }
  }
}

static bool attn_use_mfma(const float* mask, const float* drop, int S, int D, int H) {
  static const bool off = [] { const char* e = getenv("MD_ATTN_SCALAR"); return e && atoi(e) != 0; }();
  const int dh = D / H;
  return !off && !mask && !drop && S <= 16 * AT_MAXT && (dh == 16 || dh == 32 || dh == 64) && D % 4 == 0;
}
static bool attn_mfma_prepare() {        // raise the dynamic LDS limit of the attention kernels, once per device
  static std::mutex mu;
  static std::map<int, bool> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(dev);
  if (it != done.end()) return it->second;
  const bool ok = [] {
    bool good = true;
    auto set = [&](const void* f) { good = good && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    set((const void*)k_attn_mfma_fwd<16>); set((const void*)k_attn_mfma_fwd<32>); set((const void*)k_attn_mfma_fwd<64>);
    set((const void*)k_attn_mfma_bwd_q<16>); set((const void*)k_attn_mfma_bwd_q<32>); set((const void*)k_attn_mfma_bwd_q<64>);
    set((const void*)k_attn_mfma_fwd<32, true>); set((const void*)k_attn_mfma_fwd<64, true>);
    set((const void*)k_attn_mfma_fwd<32, true, true>); set((const void*)k_attn_mfma_fwd<64, true, true>);
    set((const void*)k_attn_lse_bwd_q<32>); set((const void*)k_attn_lse_bwd_q<64>);
    set((const void*)k_attn_lse_bwd_kv<32>); set((const void*)k_attn_lse_bwd_kv<64>);
    set((const void*)k_attn_mfma_bwd_q<32, true>); set((const void*)k_attn_mfma_bwd_q<64, true>);
    return good;
  }();
  done[dev] = ok;
  return ok;
}
// pitch of the per-wave P buffer: = 8 (mod 16) floats; the split-precision kernels contract over 32 keys per instruction
static int attn_pp(int S, bool sp = false) { return sp ? ((S + 31) & ~31) + 8 : ((S + 15) & ~15) + 8; }
static size_t attn_mfma_lds(int S, bool sp = false) { return (size_t)(64 * AT_KP + 4 * 16 * attn_pp(S, sp)) * 4; }
extern "C" int md_get_exact_fp32(void);
// split-precision products (three 16-bit MFMAs per fp32 product) unless the library is in exact-fp32 mode; d_head >= 32
static bool attn_split(int dh) {
  static const bool off = [] { const char* e = getenv("MD_ATTN_SPLIT"); return e && atoi(e) == 0; }();
  return !off && !md_get_exact_fp32() && dh >= 32;
}

// GELU.  kind 0: exact, 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU, transformer.py:85); kind 1: the reference's own tanh form
// 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) (transformer.py:36-38).  bwd: dx = dy * gelu'(x).
__device__ __forceinline__ void gelu_val(float v, int kind, float& y, float& d) {
#pragma clang fp contract(off)      // every fused multiply-add is written out: k_gelu and k_bias_gelu_drop then round identically
  if (kind == 0) {
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    y = v * cdf; d = fmaf(v * 0.3989422804014327f, expf(-0.5f * v * v), cdf);
  } else {
    const float c = 0.7978845608028654f, u = c * fmaf(0.044715f * v * v, v, v), th = tanhf(u);
    y = 0.5f * v * (1.f + th);
    d = fmaf(0.5f * v * (1.f - th * th) * c, fmaf(3.f * 0.044715f * v, v, 1.f), 0.5f * (1.f + th));
  }
}
__global__ __launch_bounds__(256) void k_gelu(const float* __restrict__ x, const float* __restrict__ dy, int kind, int64_t n,
                                             float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float y, d;
    gelu_val(x[i], kind, y, d);
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

extern "C" size_t md_add_layernorm_bwd_scratch_floats(int64_t rows, int32_t D);
// One step of a pre-norm residual stream with the branch's tail folded in (ViViT.py:31-46,85-91,108-111): the branch ends in
// Linear (+ bias) -> nn.Dropout, then  s = branch + stream,  h = LayerNorm(s).  y is the Linear's raw product; this pass applies bias and
// the (regenerated, philox.h) dropout decisions on the way in instead of two elementwise passes over y before it:
//   s = (y + bias) * keep-decision * scale + stream;  h = LayerNorm(s) * gamma + beta.
// Four consecutive features per lane (one Philox call), L = 16 / 32 / 64 lanes per row, row sums by shuffles inside the L lanes.
template <int L>
__device__ __forceinline__ float lanes_sum(float v) {
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int L>
__global__ __launch_bounds__(256) void k_branch_ln_fwd(const float* __restrict__ y, const float* __restrict__ bias, const int64_t* __restrict__ key,
                                                      int tag, float keep, const float* __restrict__ stream_in, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int64_t rows, int D4, float eps, float* __restrict__ out,
                                                      float* __restrict__ xhat, float* __restrict__ rstd, float* __restrict__ sum_out) {
  constexpr int RPB = 256 / L;                           // rows per workgroup
  const int c4 = threadIdx.x % L;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / L;
  const bool act = row < rows && c4 < D4;
  const int64_t i4 = act ? row * D4 + c4 : 0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    const float4 a = ((const float4*)y)[i4], r = ((const float4*)stream_in)[i4];
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ((const float4*)bias)[c4];
    float h[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    if (key) {
      float m[4];
      md_drop_keep4(md_drop_key(key), tag, i4, keep, m);
      const float sc = 1.f / keep;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = h[e] * m[e] * sc;
    }
    v[0] = h[0] + r.x; v[1] = h[1] + r.y; v[2] = h[2] + r.z; v[3] = h[3] + r.w;
  }
  const float D = (float)(D4 * 4);
  const float mean = lanes_sum<L>((v[0] + v[1]) + (v[2] + v[3])) / D;
  float d[4], q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { d[e] = act ? v[e] - mean : 0.f; q += d[e] * d[e]; }
  const float var = lanes_sum<L>(q) / D;
  const float rs = 1.f / sqrtf(var + eps);
  if (act) {
    const float4 g = ((const float4*)gamma)[c4], bt = ((const float4*)beta)[c4];
    const float4 xh = make_float4(d[0] * rs, d[1] * rs, d[2] * rs, d[3] * rs);
    ((float4*)xhat)[i4] = xh;
    ((float4*)out)[i4] = make_float4(xh.x * g.x + bt.x, xh.y * g.y + bt.y, xh.z * g.z + bt.z, xh.w * g.w + bt.w);
    ((float4*)sum_out)[i4] = make_float4(v[0], v[1], v[2], v[3]);
    if (c4 == 0) rstd[row] = rs;
  }
}
// dstream = rstd * (g - mean(g) - xhat * mean(g * xhat)) + dres,  g = dout * gamma   (gradient of the sum: what the residual stream
// carries on); dbranch = dstream * keep-decision * scale (gradient of the Linear's raw product).  The three column sums the step needs
// ride along: dgamma = sum_rows dout * xhat, dbeta = sum_rows dout, dbias = sum_rows dbranch -- every workgroup walks its row groups
// (blockIdx.x, + gridDim.x, ...) in order, adds the 256 / L row lanes in a fixed order at the end and writes one partial row
// [3][D]; k_colsum3_final adds the partial rows in a fixed tree.  (Separate passes before: k_ln_param_grad + k_ln_param_sum +
// k_colsum_partial + k_colsum_final, each re-reading an 8.5 MB tensor.)
template <int L>
__global__ __launch_bounds__(256) void k_branch_ln_bwd(const float* __restrict__ dout, const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                      const float* __restrict__ rstd, const float* __restrict__ dres, const int64_t* __restrict__ key,
                                                      int tag, float keep, int64_t rows, int D4, float* __restrict__ dstream,
                                                      float* __restrict__ dbranch, float* __restrict__ partial) {
  constexpr int RPB = 256 / L;
  extern __shared__ float sred[];                        // [RPB][3][D4 * 4]
  const int c4 = threadIdx.x % L, rl = threadIdx.x / L;
  const float D = (float)(D4 * 4);
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < D4) gm = ((const float4*)gamma)[c4];
  float acc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  MdDropKey dk; dk.k0 = dk.k1 = dk.step = 0;
  if (key) dk = md_drop_key(key);
  const float sc = key ? 1.f / keep : 1.f;
  const int64_t ngroups = (rows + RPB - 1) / RPB;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t row = grp * RPB + rl;
    const bool act = row < rows && c4 < D4;
    const int64_t i4 = act ? row * D4 + c4 : 0;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, xh[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
      const float4 t = ((const float4*)dout)[i4], x = ((const float4*)xhat)[i4];
      a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
      g[0] = t.x * gm.x; g[1] = t.y * gm.y; g[2] = t.z * gm.z; g[3] = t.w * gm.w;
      xh[0] = x.x; xh[1] = x.y; xh[2] = x.z; xh[3] = x.w;
    }
    const float m1 = lanes_sum<L>((g[0] + g[1]) + (g[2] + g[3])) / D;
    const float m2 = lanes_sum<L>((g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3])) / D;
    if (!act) continue;
    const float rs = rstd[row];
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (dres) { const float4 t = ((const float4*)dres)[i4]; r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
    float dx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[e] = rs * (g[e] - m1 - xh[e] * m2) + r[e];
    ((float4*)dstream)[i4] = make_float4(dx[0], dx[1], dx[2], dx[3]);
    if (key) {
      float m[4];
      md_drop_keep4(dk, tag, i4, keep, m);
#pragma unroll
      for (int e = 0; e < 4; ++e) dx[e] = dx[e] * m[e] * sc;
    }
    ((float4*)dbranch)[i4] = make_float4(dx[0], dx[1], dx[2], dx[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[0][e] = fmaf(a[e], xh[e], acc[0][e]); acc[1][e] += a[e]; acc[2][e] += dx[e]; }
  }
  // the row lanes of this workgroup, in order
  const int Dw = D4 * 4;
  if (c4 < D4) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) sred[(rl * 3 + q) * Dw + c4 * 4 + e] = acc[q][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * Dw; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RPB; ++k) t += sred[k * 3 * Dw + i];
    partial[(size_t)blockIdx.x * 3 * Dw + i] = t;
  }
}
// out[q][i] = sum over partial rows k of partial[k][q][i], q = 0..2 (dgamma, dbeta, dbias): 16 features x 16 row lanes per workgroup,
// lane l adds rows l, l + 16, ... in order, then the 16 lanes in order
__global__ __launch_bounds__(256) void k_colsum3_final(const float* __restrict__ partial, int nrows, int D, float* __restrict__ o0,
                                                      float* __restrict__ o1, float* __restrict__ o2) {
  __shared__ float red[16][16];
  const int c = threadIdx.x & 15, l = threadIdx.x >> 4, i = blockIdx.x * 16 + c, q = blockIdx.y;
  float* out = q == 0 ? o0 : q == 1 ? o1 : o2;
  if (!out) return;
  float t = 0.f;
  if (i < D)
    for (int k = l; k < nrows; k += 16) t += partial[((size_t)k * 3 + q) * D + i];
  red[l][c] = t;
  __syncthreads();
  if (l == 0 && i < D) {
    float sg = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sg += red[k][c];
    out[i] = sg;
  }
}
#define BLN_PARTIAL_ROWS 256
extern "C" int md_branch_layernorm_supported(int64_t rows, int32_t D) { return rows > 0 && D > 0 && (D & 3) == 0 && D <= 256 ? 1 : 0; }
extern "C" int md_branch_layernorm_fwd(const float* y, const float* bias, const int64_t* key, int32_t tag, float keep, const float* stream_in,
                                       const float* gamma, const float* beta, int64_t rows, int32_t D, float eps, float* out, float* xhat,
                                       float* rstd, float* sum_out, void* stream) {
  if (!y || !stream_in || !gamma || !beta || !out || !xhat || !rstd || !sum_out) return MD_ERR_NULL;
  if (!md_branch_layernorm_supported(rows, D) || (key && !(keep > 0.f && keep <= 1.f))) return MD_ERR_BAD_SHAPE;
  if ((((uintptr_t)y | (uintptr_t)bias | (uintptr_t)stream_in | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out | (uintptr_t)xhat |
        (uintptr_t)sum_out) & 15) != 0) return MD_ERR_BAD_SHAPE;
  const int D4 = D / 4;
  hipStream_t s = (hipStream_t)stream;
#define BLN_FWD(L_) MD_KLAUNCH(k_branch_ln_fwd<L_>, dim3((unsigned)((rows + 256 / L_ - 1) / (256 / L_))), dim3(256), 0, s, y, bias, key, tag, keep, \
                               stream_in, gamma, beta, rows, D4, eps, out, xhat, rstd, sum_out)
  if (D4 <= 16) BLN_FWD(16); else if (D4 <= 32) BLN_FWD(32); else BLN_FWD(64);
#undef BLN_FWD
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" size_t md_branch_layernorm_bwd_scratch_floats(int64_t rows, int32_t D) {
  return md_branch_layernorm_supported(rows, D) ? (size_t)BLN_PARTIAL_ROWS * 3 * (size_t)D : 0;
}
extern "C" int md_branch_layernorm_bwd(const float* dout, const float* gamma, const float* xhat, const float* rstd, const float* dres,
                                       const int64_t* key, int32_t tag, float keep, int64_t rows, int32_t D, float* dstream, float* dbranch,
                                       float* dgamma, float* dbeta, float* dbias, float* scratch, void* stream) {
  if (!dout || !gamma || !xhat || !rstd || !dstream || !dbranch || !dgamma || !dbeta || !scratch) return MD_ERR_NULL;
  if (!md_branch_layernorm_supported(rows, D) || (key && !(keep > 0.f && keep <= 1.f))) return MD_ERR_BAD_SHAPE;
  if ((((uintptr_t)dout | (uintptr_t)gamma | (uintptr_t)xhat | (uintptr_t)dres | (uintptr_t)dstream | (uintptr_t)dbranch) & 15) != 0) return MD_ERR_BAD_SHAPE;
  const int D4 = D / 4;
  hipStream_t s = (hipStream_t)stream;
  int nwg = 0;
#define BLN_BWD(L_)                                                                                                                   \
  do {                                                                                                                                \
    const int64_t groups = (rows + 256 / L_ - 1) / (256 / L_);                                                                        \
    nwg = (int)(groups < BLN_PARTIAL_ROWS ? groups : BLN_PARTIAL_ROWS);                                                               \
    MD_KLAUNCH(k_branch_ln_bwd<L_>, dim3((unsigned)nwg), dim3(256), (size_t)(256 / L_) * 3 * D * 4, s, dout, gamma, xhat, rstd, dres, key, tag, keep, \
               rows, D4, dstream, dbranch, scratch);                                                                                  \
  } while (0)
  if (D4 <= 16) BLN_BWD(16); else if (D4 <= 32) BLN_BWD(32); else BLN_BWD(64);
#undef BLN_BWD
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_colsum3_final, dim3(md_cdiv(D, 16), 3), dim3(256), 0, s, (const float*)scratch, nwg, D, dgamma, dbeta, dbias);
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
    MD_KLAUNCH(k_ln_param_grad, dim3(md_cdiv(D, 64), 1), dim3(256), 0, s, dout, xhat, (int)rows, D, dgamma, dbeta);
    MD_CHECK_LAUNCH();
  } else {
    float* pg = scratch; float* pb = scratch + (size_t)LN_CHUNKS * D;
    MD_KLAUNCH(k_ln_param_grad, dim3(md_cdiv(D, 64), LN_CHUNKS), dim3(256), 0, s, dout, xhat, (int)rows, D, pg, pb);
    MD_CHECK_LAUNCH();
    MD_KLAUNCH(k_ln_param_sum, dim3(md_cdiv(D, 16)), dim3(256), 0, s, (const float*)pg, (const float*)pb, LN_CHUNKS, D, dgamma, dbeta);
    MD_CHECK_LAUNCH();
  }
  return MD_OK;
}
extern "C" int md_attention_fwd(const float* qkv, const float* mask, const float* drop, int32_t S, int32_t B, int32_t D, int32_t H,
                                int32_t batch_first, float* probs, float* out, void* stream) {
  if (!qkv || !probs || !out) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  if (attn_use_mfma(mask, drop, S, D, H)) {
    if (!attn_mfma_prepare()) return MD_ERR_LAUNCH;
    const dim3 grid(B * H, md_cdiv(S, 64));
    const bool sp = attn_split(D / H);
    const size_t l = attn_mfma_lds(S, sp);
    const float sc = 1.f / sqrtf((float)(D / H));
    const int bf = batch_first ? 1 : 0, pp = attn_pp(S, sp);
    hipStream_t st = (hipStream_t)stream;
    if (sp) {
      if (D / H == 32) MD_KLAUNCH((k_attn_mfma_fwd<32, true>), grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, probs, out);
      else MD_KLAUNCH((k_attn_mfma_fwd<64, true>), grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, probs, out);
    } else switch (D / H) {
      case 16: MD_KLAUNCH(k_attn_mfma_fwd<16>, grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, probs, out); break;
      case 32: MD_KLAUNCH(k_attn_mfma_fwd<32>, grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, probs, out); break;
      default: MD_KLAUNCH(k_attn_mfma_fwd<64>, grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, probs, out); break;
    }
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  const size_t lds = (size_t)ATT_RB * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_attn_fwd, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), lds, (hipStream_t)stream, qkv, mask, drop, S, B, D, H,
             batch_first ? 1 : 0, 1.f / sqrtf((float)(D / H)), probs, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// ---- recomputing form (no S x S matrices): see k_attn_lse_bwd_*.  Unmasked, dropout-free attention in split precision only.
extern "C" int32_t md_attention_lse_supported(int32_t S, int32_t D, int32_t H) {
  static const bool off = [] { const char* e = getenv("MD_ATTN_LSE"); return e && atoi(e) == 0; }();
  if (off || S <= 0 || D <= 0 || H <= 0 || D % H) return 0;
  return attn_use_mfma(nullptr, nullptr, S, D, H) && attn_split(D / H) ? 1 : 0;
}
extern "C" int md_attention_lse_fwd(const float* qkv, int32_t S, int32_t B, int32_t D, int32_t H, int32_t batch_first, float* lse,
                                    float* out, void* stream) {
  if (!qkv || !lse || !out) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  if (!md_attention_lse_supported(S, D, H)) return MD_ERR_UNSUPPORTED;
  if (!attn_mfma_prepare()) return MD_ERR_LAUNCH;
  const dim3 grid(B * H, md_cdiv(S, 64));
  const size_t l = attn_mfma_lds(S, true);
  const float sc = 1.f / sqrtf((float)(D / H));
  const int bf = batch_first ? 1 : 0, pp = attn_pp(S, true);
  hipStream_t st = (hipStream_t)stream;
  if (D / H == 32) MD_KLAUNCH((k_attn_mfma_fwd<32, true, true>), grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, lse, out);
  else MD_KLAUNCH((k_attn_mfma_fwd<64, true, true>), grid, dim3(256), l, st, qkv, S, B, D, H, bf, sc, pp, lse, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_attention_lse_bwd(const float* qkv, const float* lse, const float* dout, int32_t S, int32_t B, int32_t D, int32_t H,
                                    int32_t batch_first, float* dqkv, float* delta, void* stream) {
  if (!qkv || !lse || !dout || !dqkv || !delta) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  if (!md_attention_lse_supported(S, D, H)) return MD_ERR_UNSUPPORTED;
  if (!attn_mfma_prepare()) return MD_ERR_LAUNCH;
  const dim3 grid(B * H, md_cdiv(S, 64));
  const size_t la = attn_mfma_lds(S, true), lb = (size_t)(64 * AT_VP + 4 * 2 * 16 * AT_TP) * 4 + 2 * AT_SLO(64);
  const float sc = 1.f / sqrtf((float)(D / H));
  const int bf = batch_first ? 1 : 0, pp = attn_pp(S, true);
  hipStream_t st = (hipStream_t)stream;
  const float* de = delta;
  if (D / H == 32) {
    MD_KLAUNCH(k_attn_lse_bwd_q<32>, grid, dim3(256), la, st, qkv, lse, dout, S, B, D, H, bf, sc, pp, delta, dqkv);
    MD_CHECK_LAUNCH();
    MD_KLAUNCH(k_attn_lse_bwd_kv<32>, grid, dim3(256), lb, st, qkv, lse, de, dout, S, B, D, H, bf, sc, dqkv);
  } else {
    MD_KLAUNCH(k_attn_lse_bwd_q<64>, grid, dim3(256), la, st, qkv, lse, dout, S, B, D, H, bf, sc, pp, delta, dqkv);
    MD_CHECK_LAUNCH();
    MD_KLAUNCH(k_attn_lse_bwd_kv<64>, grid, dim3(256), lb, st, qkv, lse, de, dout, S, B, D, H, bf, sc, dqkv);
  }
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_attention_bwd(const float* qkv, const float* probs, const float* drop, const float* dout, int32_t S, int32_t B,
                                int32_t D, int32_t H, int32_t batch_first, float* dqkv, float* ds_scratch, void* stream) {
  if (!qkv || !probs || !dout || !dqkv || !ds_scratch) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || D <= 0 || H <= 0 || D % H) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)(D / H));
  const int bf = batch_first ? 1 : 0;
  if (attn_use_mfma(nullptr, drop, S, D, H)) {
    if (!attn_mfma_prepare()) return MD_ERR_LAUNCH;
    const dim3 grid(B * H, md_cdiv(S, 64));
    const bool sp = attn_split(D / H);
    const size_t la = attn_mfma_lds(S, sp), lb = (size_t)2 * 64 * AT_VP * 4;
    const int pp = attn_pp(S, sp);
    const float* dsc = ds_scratch;
    if (sp) {
      if (D / H == 32) {
        MD_KLAUNCH((k_attn_mfma_bwd_q<32, true>), grid, dim3(256), la, s, qkv, probs, dout, S, B, D, H, bf, scale, pp, ds_scratch, dqkv);
        MD_CHECK_LAUNCH();
        MD_KLAUNCH((k_attn_mfma_bwd_kv<32, true>), grid, dim3(256), lb, s, qkv, probs, dout, dsc, S, B, D, H, bf, dqkv);
      } else {
        MD_KLAUNCH((k_attn_mfma_bwd_q<64, true>), grid, dim3(256), la, s, qkv, probs, dout, S, B, D, H, bf, scale, pp, ds_scratch, dqkv);
        MD_CHECK_LAUNCH();
        MD_KLAUNCH((k_attn_mfma_bwd_kv<64, true>), grid, dim3(256), lb, s, qkv, probs, dout, dsc, S, B, D, H, bf, dqkv);
      }
    } else switch (D / H) {
      case 16:
        MD_KLAUNCH(k_attn_mfma_bwd_q<16>, grid, dim3(256), la, s, qkv, probs, dout, S, B, D, H, bf, scale, pp, ds_scratch, dqkv);
        MD_CHECK_LAUNCH();
        MD_KLAUNCH(k_attn_mfma_bwd_kv<16>, grid, dim3(256), lb, s, qkv, probs, dout, dsc, S, B, D, H, bf, dqkv); break;
      case 32:
        MD_KLAUNCH(k_attn_mfma_bwd_q<32>, grid, dim3(256), la, s, qkv, probs, dout, S, B, D, H, bf, scale, pp, ds_scratch, dqkv);
        MD_CHECK_LAUNCH();
        MD_KLAUNCH(k_attn_mfma_bwd_kv<32>, grid, dim3(256), lb, s, qkv, probs, dout, dsc, S, B, D, H, bf, dqkv); break;
      default:
        MD_KLAUNCH(k_attn_mfma_bwd_q<64>, grid, dim3(256), la, s, qkv, probs, dout, S, B, D, H, bf, scale, pp, ds_scratch, dqkv);
        MD_CHECK_LAUNCH();
        MD_KLAUNCH(k_attn_mfma_bwd_kv<64>, grid, dim3(256), lb, s, qkv, probs, dout, dsc, S, B, D, H, bf, dqkv); break;
    }
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  const size_t lds = (size_t)ATT_RB * S * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_attn_bwd_q, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), lds, s, qkv, probs, drop, dout, S, B, D, H, bf, scale, ds_scratch, dqkv);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_attn_bwd_kv, dim3(B * H, md_cdiv(S, ATT_RB)), dim3(256), 0, s, qkv, probs, drop, dout, (const float*)ds_scratch, S, B, D, H, bf, dqkv);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// Linear bias + GELU + inverted dropout of a FeedForward's hidden activation in ONE pass each way (ViViT.py:31-46, transformer.py:85):
// h = x + bias[c], y = gelu(h), out = y * mask * scale (mask == nullptr: no dropout) -- the same arithmetic, in the same order, as
// md_channel_bias_fwd -> md_gelu -> md_mask_scale, so results are bit-identical to the three passes (68 MB each way per pass at cfg3).
// Backward: dx = dout * mask * scale * gelu'(x + bias); the bias gradient is the column sum of dx (md_channel_bias_bwd).
template <bool BWD>
__global__ __launch_bounds__(256) void k_bias_gelu_drop(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ mask,
                                                       const float* __restrict__ dout, float scale, int kind, int64_t n4, int c4,
                                                       float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = ((const float4*)x)[i], b = ((const float4*)bias)[i % c4];
    const float h[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    float m[4] = {1.f, 1.f, 1.f, 1.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, r[4];
    if (mask) { const float4 mm = ((const float4*)mask)[i]; m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w; }
    if (BWD) { const float4 gg = ((const float4*)dout)[i]; g[0] = gg.x; g[1] = gg.y; g[2] = gg.z; g[3] = gg.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y, d;
      gelu_val(h[e], kind, y, d);
      if (BWD) r[e] = (mask ? g[e] * m[e] * scale : g[e]) * d;
      else r[e] = mask ? y * m[e] * scale : y;
    }
    ((float4*)out)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}
extern "C" int md_bias_gelu_drop(const float* x, const float* bias, const float* mask, const float* dout, float scale, int32_t kind,
                                 int64_t rows, int32_t C, float* out, void* stream) {
  if (!x || !bias || !out) return MD_ERR_NULL;
  if (rows <= 0 || C <= 0 || (C & 3) || (kind != 0 && kind != 1)) return MD_ERR_BAD_SHAPE;
  if ((((uintptr_t)x | (uintptr_t)bias | (uintptr_t)mask | (uintptr_t)dout | (uintptr_t)out) & 15) != 0) return MD_ERR_BAD_SHAPE;
  const int64_t n4 = rows * (C / 4);
  int64_t blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  if (dout) MD_KLAUNCH(k_bias_gelu_drop<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, mask, dout, scale, kind, n4, C / 4, out);
  else MD_KLAUNCH(k_bias_gelu_drop<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, mask, dout, scale, kind, n4, C / 4, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// The same pass with the dropout decisions regenerated from a counter-based generator instead of read from a mask tensor (philox.h): at
// cfg3 the FeedForward hidden activation is 68 MB, and its mask cost one 68 MB write (torch's bernoulli_) and two 68 MB reads per layer.
template <bool BWD>
__global__ __launch_bounds__(256) void k_bias_gelu_drop_ctr(const float* __restrict__ x, const float* __restrict__ bias, const int64_t* __restrict__ state,
                                                           int tag, float keep, const float* __restrict__ dout, float scale, int kind, int64_t n4,
                                                           int c4, float* __restrict__ out) {
  const MdDropKey key = md_drop_key(state);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = ((const float4*)x)[i], b = ((const float4*)bias)[i % c4];
    const float h[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
    float m[4], g[4] = {0.f, 0.f, 0.f, 0.f}, r[4];
    md_drop_keep4(key, tag, i, keep, m);
    if (BWD) { const float4 gg = ((const float4*)dout)[i]; g[0] = gg.x; g[1] = gg.y; g[2] = gg.z; g[3] = gg.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y, d;
      gelu_val(h[e], kind, y, d);
      if (BWD) r[e] = (g[e] * m[e] * scale) * d;
      else r[e] = y * m[e] * scale;
    }
    ((float4*)out)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}
extern "C" int md_bias_gelu_drop_ctr(const float* x, const float* bias, const int64_t* state, int32_t tag, float keep, const float* dout,
                                     float scale, int32_t kind, int64_t rows, int32_t C, float* out, void* stream) {
  if (!x || !bias || !state || !out) return MD_ERR_NULL;
  if (rows <= 0 || C <= 0 || (C & 3) || (kind != 0 && kind != 1) || !(keep > 0.f && keep <= 1.f)) return MD_ERR_BAD_SHAPE;
  if ((((uintptr_t)x | (uintptr_t)bias | (uintptr_t)dout | (uintptr_t)out) & 15) != 0) return MD_ERR_BAD_SHAPE;
  const int64_t n4 = rows * (C / 4);
  int64_t blocks = (n4 + 255) / 256; if (blocks > 8192) blocks = 8192;
  if (dout) MD_KLAUNCH(k_bias_gelu_drop_ctr<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, state, tag, keep, dout, scale, kind, n4, C / 4, out);
  else MD_KLAUNCH(k_bias_gelu_drop_ctr<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, state, tag, keep, dout, scale, kind, n4, C / 4, out);
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
