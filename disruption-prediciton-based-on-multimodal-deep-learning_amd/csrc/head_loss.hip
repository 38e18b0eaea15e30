// Classifier head (Linear -> BatchNorm1d -> ELU -> Linear, src/models/R2Plus1D.py:243-248) and the fused
// softmax + Focal / LDAM / CE loss (src/loss.py:14-81) with argmax bookkeeping (src/train.py:70).
// These operate on (B, <=512) tensors: a single workgroup each, latency bound, written for exactness.
#include "common.h"

// save layout: xhat[B*Hd] | hn[B*Hd] | he[B*Hd] | invstd[Hd]
extern "C" size_t md_head_save_floats(int32_t B, int32_t D, int32_t Hd) { (void)D; return (size_t)3 * B * Hd + Hd; }

__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ feat, int B, int D, int Hd, int K,
                                                  const float* __restrict__ w0, const float* __restrict__ b0,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ w1, const float* __restrict__ b1, float alpha,
                                                  float eps, float momentum, int training, float* __restrict__ rmean,
                                                  float* __restrict__ rvar, float* __restrict__ logits,
                                                  float* __restrict__ save, int stage_w) {
  extern __shared__ float sm[];
  float* h0 = sm;            // [B][Hd]
  float* he = h0 + B * Hd;   // [B][Hd]
  const int t = threadIdx.x, nt = blockDim.x;
  if (stage_w) {
    // w0 and feat staged with coalesced loads (row pitch D+1: the per-thread rows below then fall on different banks);
    // the dot products keep their serial order, so the result is bit-identical to reading global memory directly
    float* w0s = he + B * Hd;          // [Hd][D+1]
    float* fs = w0s + Hd * (D + 1);    // [B][D]
    for (int e = t; e < Hd * D; e += nt) { const int j = e / D, d = e - j * D; w0s[j * (D + 1) + d] = w0[e]; }
    for (int e = t; e < B * D; e += nt) fs[e] = feat[e];
    __syncthreads();
    for (int e = t; e < B * Hd; e += nt) {
      const int b = e / Hd, j = e - b * Hd;
      float a = b0[j];
      for (int d = 0; d < D; ++d) a = fmaf(fs[b * D + d], w0s[j * (D + 1) + d], a);
      h0[e] = a;
    }
  } else {
    for (int e = t; e < B * Hd; e += nt) {
      const int b = e / Hd, j = e - b * Hd;
      float a = b0[j];
      for (int d = 0; d < D; ++d) a = fmaf(feat[b * D + d], w0[j * D + d], a);
      h0[e] = a;
    }
  }
  __syncthreads();
  float* xhat = save; float* hn = save + B * Hd; float* hes = save + 2 * B * Hd; float* istd = save + 3 * B * Hd;
  for (int j = t; j < Hd; j += nt) {
    float mean, var;
    if (training) {
      double s = 0.0; for (int b = 0; b < B; ++b) s += h0[b * Hd + j];
      const double m = s / B;
      double v = 0.0; for (int b = 0; b < B; ++b) { const double d = h0[b * Hd + j] - m; v += d * d; }
      v /= B;
      mean = (float)m; var = (float)v;
      if (rmean) rmean[j] = (1.f - momentum) * rmean[j] + momentum * mean;
      if (rvar) rvar[j] = (1.f - momentum) * rvar[j] + momentum * (float)(B > 1 ? v * B / (B - 1) : v);
    } else { mean = rmean[j]; var = rvar[j]; }
    const float is = 1.f / sqrtf(var + eps);
    istd[j] = is;
    for (int b = 0; b < B; ++b) {
      const float xh = (h0[b * Hd + j] - mean) * is;
      const float y = xh * gamma[j] + beta[j];
      const float z = y > 0.f ? y : (alpha >= 0.f ? alpha * expm1f(y) : -alpha * y);       // ELU(alpha) | LeakyReLU(-alpha)
      xhat[b * Hd + j] = xh; hn[b * Hd + j] = y; hes[b * Hd + j] = z; he[b * Hd + j] = z;
    }
  }
  __syncthreads();
  for (int e = t; e < B * K; e += nt) {
    const int b = e / K, k = e - b * K;
    float a = b1[k];
    for (int j = 0; j < Hd; ++j) a = fmaf(he[b * Hd + j], w1[k * Hd + j], a);
    logits[e] = a;
  }
}

__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dlogits, const float* __restrict__ feat, int B,
                                                  int D, int Hd, int K, const float* __restrict__ w0,
                                                  const float* __restrict__ gamma, const float* __restrict__ w1,
                                                  float alpha, const float* __restrict__ save, float* __restrict__ dfeat,
                                                  float* __restrict__ dw0, float* __restrict__ db0,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  float* __restrict__ dw1, float* __restrict__ db1) {
  extern __shared__ float sm[];
  float* dh = sm;            // [B][Hd]  dhn then dh0
  const int t = threadIdx.x, nt = blockDim.x;
  const float* xhat = save; const float* hn = save + B * Hd; const float* hes = save + 2 * B * Hd;
  const float* istd = save + 3 * B * Hd;
  for (int k = t; k < K; k += nt) { float a = 0.f; for (int b = 0; b < B; ++b) a += dlogits[b * K + k]; db1[k] = a; }
  for (int e = t; e < K * Hd; e += nt) {
    const int k = e / Hd, j = e - k * Hd;
    float a = 0.f; for (int b = 0; b < B; ++b) a = fmaf(dlogits[b * K + k], hes[b * Hd + j], a);
    dw1[e] = a;
  }
  for (int e = t; e < B * Hd; e += nt) {
    const int b = e / Hd, j = e - b * Hd;
    float a = 0.f; for (int k = 0; k < K; ++k) a = fmaf(dlogits[b * K + k], w1[k * Hd + j], a);
    const float y = hn[e];
    dh[e] = a * (y > 0.f ? 1.f : (alpha >= 0.f ? alpha * expf(y) : -alpha));     // ELU' | LeakyReLU'
  }
  __syncthreads();
  for (int j = t; j < Hd; j += nt) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < B; ++b) { s1 += dh[b * Hd + j]; s2 += (double)dh[b * Hd + j] * xhat[b * Hd + j]; }
    dbeta[j] = (float)s1; dgamma[j] = (float)s2;
    const float c1 = (float)(s1 / B), c2 = (float)(s2 / B), k = gamma[j] * istd[j];
    float sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float v = k * (dh[b * Hd + j] - c1 - xhat[b * Hd + j] * c2);
      dh[b * Hd + j] = v; sb += v;
    }
    db0[j] = sb;
  }
  __syncthreads();
  for (int e = t; e < Hd * D; e += nt) {
    const int j = e / D, d = e - j * D;
    float a = 0.f; for (int b = 0; b < B; ++b) a = fmaf(dh[b * Hd + j], feat[b * D + d], a);
    dw0[e] = a;
  }
  for (int e = t; e < B * D; e += nt) {
    const int b = e / D, d = e - b * D;
    float a = 0.f; for (int j = 0; j < Hd; ++j) a = fmaf(dh[b * Hd + j], w0[j * D + d], a);
    dfeat[e] = a;
  }
}

extern "C" int md_head_fwd(const float* feat, int32_t B, int32_t D, int32_t Hd, int32_t K, const float* w0,
                           const float* b0, const float* gamma, const float* beta, const float* w1, const float* b1,
                           float alpha, float eps, float momentum, int training, float* running_mean,
                           float* running_var, float* logits, float* save, void* stream) {
  if (!feat || !w0 || !b0 || !gamma || !beta || !w1 || !b1 || !logits || !save) return MD_ERR_NULL;
  if (!training && (!running_mean || !running_var)) return MD_ERR_NULL;
  if (B <= 0 || D <= 0 || Hd <= 0 || K <= 0) return MD_ERR_BAD_SHAPE;
  size_t lds = (size_t)2 * B * Hd * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  const size_t lds_w = lds + ((size_t)Hd * (D + 1) + (size_t)B * D) * 4;
  const int stage_w = lds_w <= 60000;
  if (stage_w) lds = lds_w;
  MD_KLAUNCH(k_head_fwd, dim3(1), dim3(256), lds, (hipStream_t)stream, feat, B, D, Hd, K, w0, b0, gamma, beta, w1,
                     b1, alpha, eps, momentum, training, running_mean, running_var, logits, save, stage_w);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_head_bwd(const float* dlogits, const float* feat, int32_t B, int32_t D, int32_t Hd, int32_t K,
                           const float* w0, const float* gamma, const float* w1, float alpha, const float* save,
                           float* dfeat, float* dw0, float* db0, float* dgamma, float* dbeta, float* dw1, float* db1,
                           void* stream) {
  if (!dlogits || !feat || !w0 || !gamma || !w1 || !save || !dfeat || !dw0 || !db0 || !dgamma || !dbeta || !dw1 || !db1)
    return MD_ERR_NULL;
  if (B <= 0 || D <= 0 || Hd <= 0 || K <= 0) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)B * Hd * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_head_bwd, dim3(1), dim3(256), lds, (hipStream_t)stream, dlogits, feat, B, D, Hd, K, w0, gamma, w1,
                     alpha, save, dfeat, dw0, db0, dgamma, dbeta, dw1, db1);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// ---------------------------------------------------------------- fused softmax + loss
// kind 0 focal : L = sum_i w[y_i] (1-p_i)^gamma ce_i                      (src/loss.py:25-34)
// kind 1 LDAM  : z = s*(x - m[y] onehot); L = sum_i w[y_i] nll_i / sum_i w[y_i]   (src/loss.py:58-69)
// kind 2 CE    : L = sum_i w[y_i] nll_i                                   (src/loss.py:80-81)
// pred = argmax_k softmax(x) on the UNMODIFIED logits, first maximal index (src/train.py:70).
#define MAXK 16
__global__ __launch_bounds__(256) void k_softmax_loss(int kind, const float* __restrict__ x, const int64_t* __restrict__ y,
                                                      int B, int K, const float* __restrict__ cw,
                                                      const float* __restrict__ margins, float gs,
                                                      float* __restrict__ loss, float* __restrict__ dx,
                                                      int64_t* __restrict__ pred) {
  __shared__ double red[256], redw[256];
  const int t = threadIdx.x;
  double lsum = 0.0, wsum = 0.0;
  for (int b = t; b < B; b += blockDim.x) {
    const int yy = (int)y[b];
    float z[MAXK];
    float mx = -INFINITY; int arg = 0; float rawmx = -INFINITY;
    for (int k = 0; k < K; ++k) {
      float v = x[b * K + k];
      if (v > rawmx) { rawmx = v; arg = k; }
      if (kind == 1) { if (k == yy && margins) v -= margins[k]; v *= gs; }
      z[k] = v; mx = fmaxf(mx, v);
    }
    if (pred) pred[b] = arg;
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(z[k] - mx);
    const float lse = mx + logf(se);
    const float ce = lse - z[yy];
    const float w = cw ? cw[yy] : 1.f;
    float coef;   // d L_i / d ce_i
    if (kind == 0) {
      const float p = expf(-ce);
      const float q = 1.f - p;
      const float qg = powf(q, gs);
      lsum += (double)(w * qg * ce);
      float dq = 0.f;
      if (gs != 0.f && q > 0.f) dq = gs * powf(q, gs - 1.f) * p * ce;
      coef = w * (qg + dq);
    } else {
      lsum += (double)(w * ce);
      coef = w;
    }
    wsum += (double)w;
    if (dx) {
      for (int k = 0; k < K; ++k) {
        float d = expf(z[k] - lse) - (k == yy ? 1.f : 0.f);
        d *= coef;
        if (kind == 1) d *= gs;
        dx[b * K + k] = d;    // LDAM: divided by sum of weights below
      }
    }
  }
  red[t] = lsum; redw[t] = wsum;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) { if (t < s) { red[t] += red[t + s]; redw[t] += redw[t + s]; } __syncthreads(); }
  const double W = redw[0];
  if (t == 0) loss[0] = (float)(kind == 1 ? red[0] / W : red[0]);
  if (kind == 1 && dx) {
    const float inv = (float)(1.0 / W);
    for (int e = t; e < B * K; e += blockDim.x) dx[e] *= inv;
  }
}

extern "C" int md_softmax_loss(int32_t kind, const float* logits, const int64_t* target, int32_t B, int32_t K,
                               const float* class_weight, const float* margins, float gamma_or_s, float* loss,
                               float* dlogits, int64_t* pred, void* stream) {
  if (!logits || !target || !loss) return MD_ERR_NULL;
  if (kind < 0 || kind > 2) return MD_ERR_UNSUPPORTED;
  if (B <= 0 || K <= 0) return MD_ERR_BAD_SHAPE;
  if (K > MAXK) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_softmax_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, kind, logits, target, B, K, class_weight,
                     margins, gamma_or_s, loss, dlogits, pred);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
