// Gradient-norm clipping + AdamW over every parameter tensor of the model in two launches
// (reference: torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step, src/train.py:64-66).
//
// The caller describes the tensors once in a device table (MdOptTensor per tensor, MdOptChunk per 4096-element
// chunk); both kernels walk the chunks, so one launch covers ~100 tensors of very different sizes with every
// workgroup busy.  HBM-bound: 4 B/param read for the norm; 16 B read + 16 B written per parameter for the update.
#include "common.h"

#define OPT_CHUNK 4096

struct OptTensor { float* p; float* g; float* m; float* v; long long n; };
struct OptChunk { int tensor; int offset; };          // offset in units of OPT_CHUNK elements

// partial[chunk] = sum of squares of the chunk's gradient elements (fixed order inside the chunk)
__global__ __launch_bounds__(256) void k_opt_sumsq(const OptTensor* __restrict__ tens, const OptChunk* __restrict__ chunks,
                                                   float* __restrict__ partial) {
  __shared__ float red[4];
  const OptChunk c = chunks[blockIdx.x];
  const OptTensor tt = tens[c.tensor];
  const long long base = (long long)c.offset * OPT_CHUNK;
  const int cnt = (int)min((long long)OPT_CHUNK, tt.n - base);
  const float* g = tt.g + base;
  float s = 0.f;
  const bool vec = ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (vec) {
    for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
      if (i + 3 < cnt) { const float4 x = *(const float4*)(g + i); s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
      else for (int e = i; e < cnt; ++e) s += g[e] * g[e];
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// total norm = sqrt(sum partial) in fp64, fixed order; clip coefficient as clip_grad_norm_ computes it:
// min(1, max_norm / (norm + 1e-6)).  out[0] = norm, out[1] = coefficient (1 when max_norm <= 0: no clipping).
__global__ __launch_bounds__(256) void k_opt_norm(const float* __restrict__ partial, int nchunks, float max_norm,
                                                  float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    float coef = 1.f;
    if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); coef = coef > 1.f ? 1.f : coef; }   // NaN norm -> NaN coef, as torch
    out[0] = norm; out[1] = coef;
  }
}

// One AdamW step (decoupled weight decay, no amsgrad) with the clip coefficient applied to the gradient, which is also
// written back scaled (clip_grad_norm_ clips in place).
__global__ __launch_bounds__(256) void k_opt_adamw(const OptTensor* __restrict__ tens, const OptChunk* __restrict__ chunks,
                                                   const float* __restrict__ norm_coef, float lr, float beta1, float beta2,
                                                   float eps, float weight_decay, float step_size, float inv_bc2_sqrt,
                                                   const float* __restrict__ ok_flag) {
  // ok_flag (optional, device scalar): anything but exactly 1 skips the whole update -- parameters, moments and gradients stay
  // as they are.  The data-parallel loop passes the rank-averaged "loss is finite" flag here, so the reference's
  // `if not torch.isfinite(loss): continue` (src/train.py:56-58) is decided on the device, collectively, without a host sync.
  if (ok_flag && *ok_flag != 1.f) return;
  const OptChunk c = chunks[blockIdx.x];
  const OptTensor tt = tens[c.tensor];
  const long long base = (long long)c.offset * OPT_CHUNK;
  const int cnt = (int)min((long long)OPT_CHUNK, tt.n - base);
  float* p = tt.p + base; float* g = tt.g + base; float* m = tt.m + base; float* v = tt.v + base;
  const float coef = norm_coef ? norm_coef[1] : 1.f;
  const float decay = 1.f - lr * weight_decay;
  auto upd = [&](float& pp, float& gg, float& mm, float& vv) {
    gg *= coef;
    pp *= decay;
    mm = mm + (gg - mm) * (1.f - beta1);
    vv = beta2 * vv + (1.f - beta2) * gg * gg;
    const float denom = sqrtf(vv) * inv_bc2_sqrt + eps;
    pp -= step_size * (mm / denom);
  };
  const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                      reinterpret_cast<uintptr_t>(v)) & 15) == 0);
  if (vec) {
    for (int i = threadIdx.x * 4; i < cnt; i += 1024) {
      if (i + 3 < cnt) {
        float4 P = *(float4*)(p + i), G = *(float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        *(float4*)(p + i) = P; *(float4*)(g + i) = G; *(float4*)(m + i) = M; *(float4*)(v + i) = V;
      } else {
        for (int e = i; e < cnt; ++e) upd(p[e], g[e], m[e], v[e]);
      }
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) upd(p[i], g[i], m[i], v[i]);
  }
}

extern "C" int md_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int md_opt_grad_norm(const void* tensors, const void* chunks, int32_t nchunks, float max_norm, float* partial,
                                float* norm_coef, void* stream) {
  if (!tensors || !chunks || !partial || !norm_coef) return MD_ERR_NULL;
  if (nchunks <= 0) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  MD_KLAUNCH(k_opt_sumsq, dim3(nchunks), dim3(256), 0, s, (const OptTensor*)tensors, (const OptChunk*)chunks, partial);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_opt_norm, dim3(1), dim3(256), 0, s, partial, nchunks, max_norm, norm_coef);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_opt_adamw_step_if(const void* tensors, const void* chunks, int32_t nchunks, const float* norm_coef, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* ok_flag,
                                    void* stream) {
  if (!tensors || !chunks) return MD_ERR_NULL;
  if (nchunks <= 0 || step < 1) return MD_ERR_BAD_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  MD_KLAUNCH(k_opt_adamw, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)tensors,
             (const OptChunk*)chunks, norm_coef, lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt, ok_flag);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_opt_adamw_step(const void* tensors, const void* chunks, int32_t nchunks, const float* norm_coef, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  if (!tensors || !chunks) return MD_ERR_NULL;
  if (nchunks <= 0 || step < 1) return MD_ERR_BAD_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  MD_KLAUNCH(k_opt_adamw, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)tensors,
             (const OptChunk*)chunks, norm_coef, lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
             (const float*)nullptr);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
