// Elementwise pieces of the other encoders, streaming kernels, HBM-bound:
//   SwishEfficient (src/models/resnet.py:70-81): y = x * sigmoid(x); dx = dy * s * (1 + x * (1 - s)), s = sigmoid(x)
//   NoiseLayer (src/models/NoiseLayer.py:5-16, training branch): out = x + (mean + noise * std)
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int MODE>     // 0 swish forward, 1 swish backward, 2 add noise
__global__ __launch_bounds__(256) void k_elem(const float* __restrict__ a, const float* __restrict__ b, float p0, float p1,
                                              int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    float x[4], y[4] = {0.f, 0.f, 0.f, 0.f}, r[4];
    const bool full = i + 3 < n && ((reinterpret_cast<uintptr_t>(a + i) | reinterpret_cast<uintptr_t>(out + i) |
                                     (b ? reinterpret_cast<uintptr_t>(b + i) : 0)) & 15) == 0;
    if (full) {
      const float4 v = *(const float4*)(a + i); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      if (MODE != 0) { const float4 w = *(const float4*)(b + i); y[0] = w.x; y[1] = w.y; y[2] = w.z; y[3] = w.w; }
    } else {
      for (int e = 0; e < 4; ++e) { x[e] = i + e < n ? a[i + e] : 0.f; if (MODE != 0) y[e] = i + e < n ? b[i + e] : 0.f; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (MODE == 0) r[e] = x[e] * sigmoidf_(x[e]);
      else if (MODE == 1) { const float s = sigmoidf_(x[e]); r[e] = y[e] * (s * (1.f + x[e] * (1.f - s))); }
      else r[e] = x[e] + (p0 + y[e] * p1);
    }
    if (full) *(float4*)(out + i) = make_float4(r[0], r[1], r[2], r[3]);
    else for (int e = 0; e < 4; ++e) if (i + e < n) out[i + e] = r[e];
  }
}

static int elem_blocks(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (int)b;
}

extern "C" int md_swish_fwd(const float* x, int64_t n, float* y, void* stream) {
  if (!x || !y) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_elem<0>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, 0.f, 0.f, n, y);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_swish_bwd(const float* x, const float* dy, int64_t n, float* dx, void* stream) {
  if (!x || !dy || !dx) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_elem<1>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, dy, 0.f, 0.f, n, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_add_noise(const float* x, const float* noise, float mean, float std, int64_t n, float* out, void* stream) {
  if (!x || !noise || !out) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_elem<2>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, noise, mean, std, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
