// Elementwise pieces of the other encoders, streaming kernels, HBM-bound:
//   SwishEfficient (src/models/resnet.py:70-81): y = x * sigmoid(x); dx = dy * s * (1 + x * (1 - s)), s = sigmoid(x)
//   NoiseLayer (src/models/NoiseLayer.py:5-16, training branch): out = x + (mean + noise * std)
#include "common.h"
#include "philox.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <int MODE>     // 0 swish forward, 1 swish backward, 2 add noise, 3 leaky(a + b, p0), 4 its backward (a = output, b = dout)
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
      else if (MODE == 2) r[e] = x[e] + (p0 + y[e] * p1);
      else if (MODE == 3) r[e] = md_leaky(x[e] + y[e], p0);
      else r[e] = y[e] * (x[e] > 0.f ? 1.f : p0);
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
// leaky(a + b, alpha): the residual close of a stand-alone SpatioTemporalResBlock (reference R2Plus1D.py:183-187); the backward takes
// the OUTPUT (its sign is the sign of a + b for alpha >= 0) and gives the one gradient both summands share.
extern "C" int md_add_leaky_fwd(const float* a, const float* b, float alpha, int64_t n, float* out, void* stream) {
  if (!a || !b || !out) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_elem<3>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, 0.f, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_add_leaky_bwd(const float* out, const float* dout, float alpha, int64_t n, float* dx, void* stream) {
  if (!out || !dout || !dx) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_elem<4>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, dout, alpha, 0.f, n, dx);
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

// ------------------------------------------------------------------------------------------------
// Squeeze-and-excitation gate + Swish of Bottleneck3D (src/models/resnet.py:182-190) and its residual close (:196-198),
// on the (N,C,T,H,W) tensors that cross the module boundary (one contiguous row of thw values per (n,c)).
//   pool[n,c] = mean_thw a;  h = relu(W1 pool + b1);  gate = sigmoid(W2 h + b2);  out = swish(a * gate)
// Backward: q = a*gate, dq = dout * swish'(q);  da = dq*gate + dpool/thw;  dgate[n,c] = sum_thw dq*a, then through the two
// tiny fully connected layers.  All reductions are fixed-order trees (bitwise reproducible).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {      // red: 4 floats of shared memory
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// MODE 0: out[row] = mean(a[row][:]);  MODE 1: out[row] = sum_thw dout * swish'(a*g) * a   (g = gate[row])
template <int MODE>
__device__ __forceinline__ float row_term(float x, float d, float g) {
  if (MODE == 0) return x;
  if (MODE == 2) return d * x;                                      // plain gate (no Swish): dgate = sum dout * a
  const float q = x * g, sg = sigmoidf_(q);
  return d * (sg * (1.f + q * (1.f - sg))) * x;
}
template <int MODE>
__device__ __forceinline__ float row_term4(float4 x, float4 d, float g) {
  return (row_term<MODE>(x.x, d.x, g) + row_term<MODE>(x.y, d.y, g)) + (row_term<MODE>(x.z, d.z, g) + row_term<MODE>(x.w, d.w, g));
}
template <int MODE>
__global__ __launch_bounds__(256) void k_row_reduce(const float* __restrict__ a, const float* __restrict__ gate,
                                                    const float* __restrict__ dout, int64_t thw, float* __restrict__ out) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* ar = a + row * thw;
  const float* dr = MODE == 0 ? ar : dout + row * thw;
  const float g = MODE == 1 ? gate[row] : 0.f;
  float s = 0.f;
  if ((thw & 3) == 0 && (((uintptr_t)a | (uintptr_t)dr) & 15) == 0) {
    // 16-byte loads, four per tensor in flight per lane (one workgroup per row: few waves per CU, so depth per lane matters);
    // four running sums per lane, combined in a fixed order
    const float4* a4 = (const float4*)ar; const float4* d4 = (const float4*)dr;
    const int64_t n4 = thw >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t i = threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i + 3 * 256 < n4; i += 4 * 256) {
      const float4 x0 = a4[i], x1 = a4[i + 256], x2 = a4[i + 512], x3 = a4[i + 768];
      float4 e0 = z, e1 = z, e2 = z, e3 = z;
      if (MODE != 0) { e0 = d4[i]; e1 = d4[i + 256]; e2 = d4[i + 512]; e3 = d4[i + 768]; }
      s0 += row_term4<MODE>(x0, e0, g); s1 += row_term4<MODE>(x1, e1, g);
      s2 += row_term4<MODE>(x2, e2, g); s3 += row_term4<MODE>(x3, e3, g);
    }
    for (; i < n4; i += 256) s0 += row_term4<MODE>(a4[i], MODE != 0 ? d4[i] : z, g);
    s = (s0 + s1) + (s2 + s3);
  } else {
    for (int64_t i = threadIdx.x; i < thw; i += 256) s += row_term<MODE>(ar[i], MODE != 0 ? dr[i] : 0.f, g);
  }
  const float t = block_sum256(s, red);
  if (threadIdx.x == 0) out[row] = MODE == 0 ? t / (float)thw : t;
}

// gate forward: one workgroup per sample.  hidden[N][Wd] (post-ReLU) is kept for the backward.
__global__ __launch_bounds__(256) void k_se_gate_fwd(const float* __restrict__ pool, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, int N, int Cc, int Wd,
                                                     float* __restrict__ hidden, float* __restrict__ gate) {
  const int n = blockIdx.x;
  for (int j = threadIdx.x; j < Wd; j += 256) {
    float acc = b1[j];
    for (int c = 0; c < Cc; ++c) acc = fmaf(w1[j * Cc + c], pool[n * Cc + c], acc);
    hidden[n * Wd + j] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();      // (the hidden values of this sample were written by this workgroup's threads)
  __threadfence_block();
  for (int c = threadIdx.x; c < Cc; c += 256) {
    float acc = b2[c];
    for (int j = 0; j < Wd; ++j) acc = fmaf(w2[c * Wd + j], hidden[n * Wd + j], acc);
    gate[n * Cc + c] = sigmoidf_(acc);
  }
}

// gate backward, per sample (one workgroup each): dgate[n][C] -> dz2[n][C], dh[n][Wd], dpool[n][C]
__global__ __launch_bounds__(256) void k_se_gate_bwd_n(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                       const float* __restrict__ hidden, const float* __restrict__ w1,
                                                       const float* __restrict__ w2, int Cc, int Wd, float* __restrict__ dz2,
                                                       float* __restrict__ dh, float* __restrict__ dpool) {
  const int n = blockIdx.x, t = threadIdx.x;
  for (int c = t; c < Cc; c += 256) { const float s = gate[n * Cc + c]; dz2[n * Cc + c] = dgate[n * Cc + c] * s * (1.f - s); }
  __syncthreads(); __threadfence_block();
  for (int j = t; j < Wd; j += 256) {
    float acc = 0.f; for (int c = 0; c < Cc; ++c) acc = fmaf(dz2[n * Cc + c], w2[c * Wd + j], acc);
    dh[n * Wd + j] = hidden[n * Wd + j] > 0.f ? acc : 0.f;
  }
  __syncthreads(); __threadfence_block();
  for (int c = t; c < Cc; c += 256) {
    float acc = 0.f; for (int j = 0; j < Wd; ++j) acc = fmaf(dh[n * Wd + j], w1[j * Cc + c], acc);
    dpool[n * Cc + c] = acc;
  }
}
// gate backward, parameters: one thread per element of dw2[C][Wd] | db2[C] | dw1[Wd][C] | db1[Wd], samples in order
__global__ __launch_bounds__(256) void k_se_gate_bwd_w(const float* __restrict__ dz2, const float* __restrict__ dh,
                                                       const float* __restrict__ hidden, const float* __restrict__ pool, int N, int Cc,
                                                       int Wd, float* __restrict__ dw1, float* __restrict__ db1,
                                                       float* __restrict__ dw2, float* __restrict__ db2) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e < Cc * Wd) {
    const int c = e / Wd, j = e - c * Wd;
    float acc = 0.f; for (int n = 0; n < N; ++n) acc = fmaf(dz2[n * Cc + c], hidden[n * Wd + j], acc);
    dw2[e] = acc; return;
  }
  e -= Cc * Wd;
  if (e < Cc) { float acc = 0.f; for (int n = 0; n < N; ++n) acc += dz2[n * Cc + e]; db2[e] = acc; return; }
  e -= Cc;
  if (e < Wd * Cc) {
    const int j = e / Cc, c = e - j * Cc;
    float acc = 0.f; for (int n = 0; n < N; ++n) acc = fmaf(dh[n * Wd + j], pool[n * Cc + c], acc);
    dw1[e] = acc; return;
  }
  e -= Wd * Cc;
  if (e < Wd) { float acc = 0.f; for (int n = 0; n < N; ++n) acc += dh[n * Wd + e]; db1[e] = acc; }
}

// MODE 0: out = swish(a*g);  MODE 1: da = dout*swish'(a*g)*g + dpool/thw;  MODE 2: out = relu(a+b);  MODE 3: dx = dout*(out>0)
// MODE 4: out = a*g;  MODE 5: da = dout*g + dpool/thw   (squeeze-excitation without Swish, MLSTM_FCN.py:17-33)
template <int MODE>
__device__ __forceinline__ float row_elem1(float a, float b, float g, float extra) {
  if (MODE == 0) { const float q = a * g; return q * sigmoidf_(q); }
  if (MODE == 1) { const float q = a * g, sg = sigmoidf_(q); return b * (sg * (1.f + q * (1.f - sg))) * g + extra; }
  if (MODE == 2) { const float s = a + b; return s > 0.f ? s : 0.f; }
  if (MODE == 3) return a > 0.f ? b : 0.f;
  if (MODE == 4) return a * g;
  return b * g + extra;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_row_elem(const float* __restrict__ a, const float* __restrict__ b,
                                                  const float* __restrict__ rowv, const float* __restrict__ rowv2,
                                                  int64_t thw, int64_t n, float* __restrict__ out) {
  constexpr bool ROWS = MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5;       // per-row constants
  constexpr bool HAS_B = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5;
  constexpr bool EXTRA = MODE == 1 || MODE == 5;
  const bool al16 = (((uintptr_t)a | (uintptr_t)out | (HAS_B ? (uintptr_t)b : (uintptr_t)0)) & 15) == 0;
  if ((thw & 3) == 0 && al16 && n < ((int64_t)1 << 31)) {
    // 16 bytes per lane (the four elements share a row: thw % 4 == 0), row index by a 32-bit division
    const uint32_t n4 = (uint32_t)(n >> 2), t4 = (uint32_t)(thw >> 2), stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n4; i += stride) {
      const float4 x = ((const float4*)a)[i];
      float4 y = x;
      if (HAS_B) y = ((const float4*)b)[i];
      float g = 0.f, e = 0.f;
      if (ROWS) { const uint32_t row = i / t4; g = rowv[row]; if (EXTRA) e = rowv2[row] / (float)thw; }
      float4 r;
      r.x = row_elem1<MODE>(x.x, y.x, g, e); r.y = row_elem1<MODE>(x.y, y.y, g, e);
      r.z = row_elem1<MODE>(x.z, y.z, g, e); r.w = row_elem1<MODE>(x.w, y.w, g, e);
      ((float4*)out)[i] = r;
    }
    return;
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float g = 0.f, e = 0.f;
    if (ROWS) { const int64_t row = i / thw; g = rowv[row]; if (EXTRA) e = rowv2[row] / (float)thw; }
    out[i] = row_elem1<MODE>(a[i], HAS_B ? b[i] : 0.f, g, e);
  }
}

extern "C" int md_se_swish_fwd(const float* a, int32_t N, int32_t Cc, int64_t thw, int32_t Wd, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* pool, float* hidden, float* gate, float* out,
                               void* stream) {
  if (!a || !w1 || !b1 || !w2 || !b2 || !pool || !hidden || !gate || !out) return MD_ERR_NULL;
  if (N <= 0 || Cc <= 0 || thw <= 0 || Wd <= 0) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)N * Cc * thw;
  MD_KLAUNCH(k_row_reduce<0>, dim3(N * Cc), dim3(256), 0, s, a, (const float*)nullptr, (const float*)nullptr, thw, pool);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_fwd, dim3(N), dim3(256), 0, s, (const float*)pool, w1, b1, w2, b2, N, Cc, Wd, hidden, gate);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_row_elem<0>, dim3(elem_blocks(n)), dim3(256), 0, s, a, (const float*)nullptr, (const float*)gate,
             (const float*)nullptr, thw, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// scratch: 3*N*C + N*Wd floats
extern "C" int md_se_swish_bwd(const float* a, const float* dout, int32_t N, int32_t Cc, int64_t thw, int32_t Wd,
                               const float* w1, const float* w2, const float* pool, const float* hidden, const float* gate,
                               float* da, float* dw1, float* db1, float* dw2, float* db2, float* scratch, void* stream) {
  if (!a || !dout || !w1 || !w2 || !pool || !hidden || !gate || !da || !dw1 || !db1 || !dw2 || !db2 || !scratch)
    return MD_ERR_NULL;
  if (N <= 0 || Cc <= 0 || thw <= 0 || Wd <= 0) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)N * Cc * thw;
  float* dgate = scratch; float* dz2 = scratch + (size_t)N * Cc; float* dpool = dz2 + (size_t)N * Cc; float* dh = dpool + (size_t)N * Cc;
  MD_KLAUNCH(k_row_reduce<1>, dim3(N * Cc), dim3(256), 0, s, a, gate, dout, thw, dgate);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_bwd_n, dim3(N), dim3(256), 0, s, (const float*)dgate, gate, hidden, w1, w2, Cc, Wd, dz2, dh, dpool);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_bwd_w, dim3(md_cdiv(2 * Cc * Wd + Cc + Wd, 256)), dim3(256), 0, s, (const float*)dz2, (const float*)dh, hidden, pool,
             N, Cc, Wd, dw1, db1, dw2, db2);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_row_elem<1>, dim3(elem_blocks(n)), dim3(256), 0, s, a, dout, gate, (const float*)dpool, thw, n, da);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// Squeeze-excitation WITHOUT the Swish (MLSTM_FCN's SqueezeExciteBlock, MLSTM_FCN.py:17-33): out = a * gate; same gate
// network (pass zero biases for its bias-free Linears).
extern "C" int md_se_scale_fwd(const float* a, int32_t N, int32_t Cc, int64_t thw, int32_t Wd, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* pool, float* hidden, float* gate, float* out,
                               void* stream) {
  if (!a || !w1 || !b1 || !w2 || !b2 || !pool || !hidden || !gate || !out) return MD_ERR_NULL;
  if (N <= 0 || Cc <= 0 || thw <= 0 || Wd <= 0) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)N * Cc * thw;
  MD_KLAUNCH(k_row_reduce<0>, dim3(N * Cc), dim3(256), 0, s, a, (const float*)nullptr, (const float*)nullptr, thw, pool);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_fwd, dim3(N), dim3(256), 0, s, (const float*)pool, w1, b1, w2, b2, N, Cc, Wd, hidden, gate);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_row_elem<4>, dim3(elem_blocks(n)), dim3(256), 0, s, a, (const float*)nullptr, (const float*)gate,
             (const float*)nullptr, thw, n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_se_scale_bwd(const float* a, const float* dout, int32_t N, int32_t Cc, int64_t thw, int32_t Wd,
                               const float* w1, const float* w2, const float* pool, const float* hidden, const float* gate,
                               float* da, float* dw1, float* db1, float* dw2, float* db2, float* scratch, void* stream) {
  if (!a || !dout || !w1 || !w2 || !pool || !hidden || !gate || !da || !dw1 || !db1 || !dw2 || !db2 || !scratch)
    return MD_ERR_NULL;
  if (N <= 0 || Cc <= 0 || thw <= 0 || Wd <= 0) return MD_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)N * Cc * thw;
  float* dgate = scratch; float* dz2 = scratch + (size_t)N * Cc; float* dpool = dz2 + (size_t)N * Cc; float* dh = dpool + (size_t)N * Cc;
  MD_KLAUNCH(k_row_reduce<2>, dim3(N * Cc), dim3(256), 0, s, a, gate, dout, thw, dgate);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_bwd_n, dim3(N), dim3(256), 0, s, (const float*)dgate, gate, hidden, w1, w2, Cc, Wd, dz2, dh, dpool);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_se_gate_bwd_w, dim3(md_cdiv(2 * Cc * Wd + Cc + Wd, 256)), dim3(256), 0, s, (const float*)dz2, (const float*)dh, hidden, pool,
             N, Cc, Wd, dw1, db1, dw2, db2);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_row_elem<5>, dim3(elem_blocks(n)), dim3(256), 0, s, a, dout, gate, (const float*)dpool, thw, n, da);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_add_relu_fwd(const float* a, const float* b, int64_t n, float* out, void* stream) {
  if (!a || !b || !out) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_row_elem<2>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, (const float*)nullptr,
             (const float*)nullptr, (int64_t)((n & 3) == 0 ? 4 : 1), n, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_add_relu_bwd(const float* out, const float* dout, int64_t n, float* dx, void* stream) {
  if (!out || !dout || !dx) return MD_ERR_NULL;
  if (n <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_row_elem<3>, dim3(elem_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, dout, (const float*)nullptr,
             (const float*)nullptr, (int64_t)((n & 3) == 0 ? 4 : 1), n, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// Pieces of ResNet3D / SlowFast outside the bottleneck blocks (src/models/resnet.py:221-226, slowfast.py:33-34,86-87), on
// (N,C,T,H,W) tensors: MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1)) and AdaptiveAvgPool3d(1).
// ------------------------------------------------------------------------------------------------
// forward: out[nct][ho][wo] = max over the 3x3 window (padding never wins: -inf); idx = flat h*W+w of the first maximum in
// row-major window order (PyTorch's tie rule).  backward (gather form, deterministic): dx[h][w] = sum of dout over the
// windows whose idx is (h,w).
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, int64_t planes, int H, int W, int Ho, int Wo,
                                                    float* __restrict__ out, int* __restrict__ idx) {
  const int64_t n = planes * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int wo = (int)(i % Wo); const int64_t r = i / Wo; const int ho = (int)(r % Ho); const int64_t p = r / Ho;
    const float* xp = x + p * H * W;
    float best = -INFINITY; int bi = -1;
    for (int dh = 0; dh < 3; ++dh) {
      const int h = ho * 2 - 1 + dh; if (h < 0 || h >= H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = wo * 2 - 1 + dw; if (w < 0 || w >= W) continue;
        const float v = xp[h * W + w];
        if (bi < 0 || v > best || v != v) { best = v; bi = h * W + w; }      // ATen's rule: (val > max) || isnan(val); first maximum wins
      }
    }
    out[i] = best; idx[i] = bi;
  }
}
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ dout, const int* __restrict__ idx, int64_t planes,
                                                    int H, int W, int Ho, int Wo, float* __restrict__ dx) {
  const int64_t n = planes * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W); const int64_t r = i / W; const int h = (int)(r % H); const int64_t p = r / H;
    const int me = h * W + w;
    float s = 0.f;
    // windows ho with 2*ho-1 <= h <= 2*ho+1  <=>  ho in [ceil((h-1)/2), floor((h+1)/2)]
    for (int ho = (h >= 1 ? (h - 1 + 1) / 2 : 0); ho <= (h + 1) / 2 && ho < Ho; ++ho)
      for (int wo = (w >= 1 ? (w - 1 + 1) / 2 : 0); wo <= (w + 1) / 2 && wo < Wo; ++wo) {
        const int64_t o = (p * Ho + ho) * Wo + wo;
        if (idx[o] == me) s += dout[o];
      }
    dx[i] = s;
  }
}
// dx[row][i] = dmean[row] / thw
__global__ __launch_bounds__(256) void k_rowmean_bwd(const float* __restrict__ dmean, int64_t thw, int64_t n, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = dmean[i / thw] / (float)thw;
}

extern "C" int md_maxpool_1x3x3_fwd(const float* x, int64_t planes, int32_t H, int32_t W, float* out, int32_t* idx, void* stream) {
  if (!x || !out || !idx) return MD_ERR_NULL;
  if (planes <= 0 || H <= 0 || W <= 0) return MD_ERR_BAD_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  MD_KLAUNCH(k_maxpool_fwd, dim3(elem_blocks(planes * Ho * Wo * 4)), dim3(256), 0, (hipStream_t)stream, x, planes, H, W, Ho, Wo, out, idx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_maxpool_1x3x3_bwd(const float* dout, const int32_t* idx, int64_t planes, int32_t H, int32_t W, float* dx, void* stream) {
  if (!dout || !idx || !dx) return MD_ERR_NULL;
  if (planes <= 0 || H <= 0 || W <= 0) return MD_ERR_BAD_SHAPE;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  MD_KLAUNCH(k_maxpool_bwd, dim3(elem_blocks(planes * H * W * 4)), dim3(256), 0, (hipStream_t)stream, dout, idx, planes, H, W, Ho, Wo, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_rowmean_fwd(const float* x, int64_t rows, int64_t thw, float* mean, void* stream) {
  if (!x || !mean) return MD_ERR_NULL;
  if (rows <= 0 || thw <= 0 || rows > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_row_reduce<0>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
             (const float*)nullptr, thw, mean);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_rowmean_bwd(const float* dmean, int64_t rows, int64_t thw, float* dx, void* stream) {
  if (!dmean || !dx) return MD_ERR_NULL;
  if (rows <= 0 || thw <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_rowmean_bwd, dim3(elem_blocks(rows * thw * 4)), dim3(256), 0, (hipStream_t)stream, dmean, thw, rows * thw, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------
// Small pieces of the 0D encoders (src/models/CnnLSTM.py): a per-channel bias on an (N,C,L) tensor (Conv1d bias that is NOT
// followed directly by a normalisation, :42) and the sequence reduction that the attention pooling reduces to (:76-97).
// ------------------------------------------------------------------------------------------------
// MODE 0: out[n][c][l] = x[n][c][l] + bias[c];  MODE 1: out[b][d] = scale * sum_s x[b][s][d];  MODE 2: out[b][s][d] = scale * g[b][d]
// MODE 3: out[i] = x[i] * v[i] * scale   (inverted-dropout mask between LSTM layers; A*Bn*Cn elements)
template <int MODE>
__global__ __launch_bounds__(256) void k_small(const float* __restrict__ x, const float* __restrict__ v, float scale, int A,
                                               int Bn, int Cn, float* __restrict__ out) {
  const int64_t n = MODE == 1 ? (int64_t)A * Cn : (int64_t)A * Bn * Cn;
  // 16-byte form of the two streaming modes (Linear bias over rows: Cn == 1, Bn % 4 == 0; dropout mask): a float per thread and a
  // 64-bit division per element made these launches 2-3x slower than the bytes they move
  if ((MODE == 3 || (MODE == 0 && Cn == 1 && (Bn & 3) == 0)) && (n & 3) == 0 &&
      ((((uintptr_t)x | (uintptr_t)v | (uintptr_t)out) & 15) == 0)) {
    const int64_t n4 = n >> 2;
    const int b4 = Bn >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 a = ((const float4*)x)[i];
      const float4 b = MODE == 3 ? ((const float4*)v)[i] : ((const float4*)v)[i % b4];
      float4 r;
      if (MODE == 3) { r.x = a.x * b.x * scale; r.y = a.y * b.y * scale; r.z = a.z * b.z * scale; r.w = a.w * b.w * scale; }
      else { r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; r.w = a.w + b.w; }
      ((float4*)out)[i] = r;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (MODE == 3) out[i] = x[i] * v[i] * scale;
    else if (MODE == 0) out[i] = x[i] + v[(i / Cn) % Bn];                  // A = N, Bn = C, Cn = L
    else if (MODE == 1) {                                                  // A = B, Bn = S, Cn = D
      const int64_t b = i / Cn, d = i - b * Cn;
      float s = 0.f;
      for (int t = 0; t < Bn; ++t) s += x[(b * Bn + t) * Cn + d];
      out[i] = s * scale;
    } else {                                                               // A = B, Bn = S, Cn = D: broadcast back
      const int64_t b = i / ((int64_t)Bn * Cn), d = i % Cn;
      out[i] = v[b * Cn + d] * scale;
    }
  }
}
// db[c] = sum over n, l of dout[n][c][l] (fixed order)
__global__ __launch_bounds__(256) void k_channel_bias_bwd(const float* __restrict__ dout, int Nn, int Cc, int L, float* __restrict__ db) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < Nn * L; i += 256) { const int n = i / L, l = i - n * L; s += dout[((size_t)n * Cc + c) * L + l]; }
  const float t = block_sum256(s, red);
  if (threadIdx.x == 0) db[c] = t;
}

extern "C" int md_channel_bias_fwd(const float* x, const float* bias, int32_t N, int32_t C, int32_t L, float* out, void* stream) {
  if (!x || !bias || !out) return MD_ERR_NULL;
  if (N <= 0 || C <= 0 || L <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_small<0>, dim3(elem_blocks((int64_t)N * C * L * 4)), dim3(256), 0, (hipStream_t)stream, x, bias, 1.f, N, C, L, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// L == 1 and few rows (a table broadcast over the batch: ViViT's positional embedding, 794 k "channels" x 4 clips): one thread
// per channel, rows in order, coalesced across channels.
__global__ __launch_bounds__(256) void k_channel_bias_bwd_rows(const float* __restrict__ dout, int Nn, int Cc, float* __restrict__ db) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Cc) return;
  float s = 0.f;
  for (int n = 0; n < Nn; ++n) s += dout[(size_t)n * Cc + c];
  db[c] = s;
}
// L == 1 and many rows (a Linear bias over 16 548 tokens): column sums in two fixed-order levels.  Level 1: workgroup = 64
// columns x 4 row lanes over one of CB_CHUNKS row chunks -> partial [chunk][C]; level 2: 16 columns x 16 chunk lanes.
#define CB_CHUNKS 128
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ dout, int rows, int Cc, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, i = blockIdx.x * 64 + c;
  const int per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  if (i < Cc)
    for (int r = r0 + rl; r < r1; r += 4) s += dout[(size_t)r * Cc + i];
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && i < Cc) part[(size_t)blockIdx.y * Cc + i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int chunks, int Cc, float* __restrict__ db) {
  __shared__ float red[16][16];
  const int c = threadIdx.x & 15, l = threadIdx.x >> 4, i = blockIdx.x * 16 + c;
  float s = 0.f;
  if (i < Cc)
    for (int k = l; k < chunks; k += 16) s += part[(size_t)k * Cc + i];
  red[l][c] = s;
  __syncthreads();
  if (l == 0 && i < Cc) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    db[i] = t;
  }
}
extern "C" size_t md_channel_bias_bwd_scratch_floats(int32_t N, int32_t C, int32_t L) {
  return (L == 1 && N >= 2048) ? (size_t)CB_CHUNKS * (size_t)C : 0;
}
extern "C" int md_channel_bias_bwd(const float* dout, int32_t N, int32_t C, int32_t L, float* dbias, float* scratch, void* stream) {
  if (!dout || !dbias) return MD_ERR_NULL;
  if (N <= 0 || C <= 0 || L <= 0) return MD_ERR_BAD_SHAPE;
  if (scratch && md_channel_bias_bwd_scratch_floats(N, C, L)) {
    MD_KLAUNCH(k_colsum_partial, dim3(md_cdiv(C, 64), CB_CHUNKS), dim3(256), 0, (hipStream_t)stream, dout, N, C, scratch);
    MD_CHECK_LAUNCH();
    MD_KLAUNCH(k_colsum_final, dim3(md_cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, CB_CHUNKS, C, dbias);
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  if (L == 1 && N <= 64 && C >= 4096) {
    MD_KLAUNCH(k_channel_bias_bwd_rows, dim3(md_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dout, N, C, dbias);
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  MD_KLAUNCH(k_channel_bias_bwd, dim3(C), dim3(256), 0, (hipStream_t)stream, dout, N, C, L, dbias);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_seq_sum_fwd(const float* x, int32_t B, int32_t S, int32_t D, float scale, float* out, void* stream) {
  if (!x || !out) return MD_ERR_NULL;
  if (B <= 0 || S <= 0 || D <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_small<1>, dim3(elem_blocks((int64_t)B * D * 4)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, scale, B, S, D, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_seq_sum_bwd(const float* dout, int32_t B, int32_t S, int32_t D, float scale, float* dx, void* stream) {
  if (!dout || !dx) return MD_ERR_NULL;
  if (B <= 0 || S <= 0 || D <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_small<2>, dim3(elem_blocks((int64_t)B * S * D * 4)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, dout, scale, B, S, D, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// out[i] = x[i] * scale where element i is kept, 0 otherwise; kept <=> Philox word (state, tag, i) < keep (philox.h).  Forward and
// backward of a mask-free nn.Dropout are this one kernel (on the activation, then on the gradient).
__global__ __launch_bounds__(256) void k_dropout_ctr(const float* __restrict__ x, const int64_t* __restrict__ state, int tag, float keep,
                                                    float scale, int64_t n, int vec, float* __restrict__ out) {
  const MdDropKey key = md_drop_key(state);
  const int64_t n4 = (n + 3) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float m[4];
    md_drop_keep4(key, tag, i, keep, m);
    if (vec) {
      const float4 a = ((const float4*)x)[i];
      ((float4*)out)[i] = make_float4(a.x * m[0] * scale, a.y * m[1] * scale, a.z * m[2] * scale, a.w * m[3] * scale);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i * 4 + e < n) out[i * 4 + e] = x[i * 4 + e] * m[e] * scale;
    }
  }
}
extern "C" int md_dropout_ctr(const float* x, const int64_t* state, int32_t tag, float keep, float scale, int64_t n, float* out, void* stream) {
  if (!x || !state || !out) return MD_ERR_NULL;
  if (n <= 0 || !(keep > 0.f && keep <= 1.f)) return MD_ERR_BAD_SHAPE;
  const int vec = (n & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
  MD_KLAUNCH(k_dropout_ctr, dim3(elem_blocks((n + 3) >> 2)), dim3(256), 0, (hipStream_t)stream, x, state, tag, keep, scale, n, vec, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_mask_scale(const float* x, const float* mask, float scale, int64_t n, float* out, void* stream) {
  if (!x || !mask || !out) return MD_ERR_NULL;
  if (n <= 0 || n > 0x7fffffff) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_small<3>, dim3(elem_blocks(n * 4)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, (int)n, 1, 1, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}


// Device-side tail of DatasetForVideo.get_video_data (reference src/dataset.py:124-144 without the cv2 augmentations): centre
// crop (:241-246), BGR mean subtraction (:203-207) and (T,H,W,C) -> (C,T,H,W) (:229-230), from the uint8 frames cv2.imread
// returns.  frames [B][T][Hr][Wr][3] uint8; layout 0: out [B][3][T][S][S] fp32 (the module-boundary layout); layout 1:
// out [B][T][S][S][4] fp32, channel 3 = 0 (the kernels' channels-last layout: skips md_nchw_to_cl).  One thread per output
// pixel: 3 bytes in, 12-16 bytes out (HBM-bound byte kernel, no reuse).
__global__ __launch_bounds__(256) void k_clip_preprocess(const unsigned char* __restrict__ frames, int B, int T, int Hr, int Wr, int S,
                                                        int y0, int x0, float m0, float m1, float m2, int layout,
                                                        float* __restrict__ out) {
  const int64_t n = (int64_t)B * T * S * S;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % S); int64_t r = e / S;
    const int y = (int)(r % S); r /= S;
    const int t = (int)(r % T), b = (int)(r / T);
    const unsigned char* px = frames + ((((int64_t)b * T + t) * Hr + (y0 + y)) * Wr + (x0 + x)) * 3;
    const float v0 = (float)px[0] - m0, v1 = (float)px[1] - m1, v2 = (float)px[2] - m2;
    if (layout == 0) {
      const int64_t plane = (int64_t)T * S * S, o = (int64_t)b * 3 * plane + ((int64_t)t * S + y) * S + x;
      out[o] = v0; out[o + plane] = v1; out[o + 2 * plane] = v2;
    } else {
      *(float4*)(out + e * 4) = make_float4(v0, v1, v2, 0.f);
    }
  }
}
extern "C" int md_clip_preprocess(const uint8_t* frames, int32_t B, int32_t T, int32_t Hr, int32_t Wr, int32_t S, const float* mean_bgr,
                                  int32_t layout, float* out, void* stream) {
  if (!frames || !mean_bgr || !out) return MD_ERR_NULL;
  if (B <= 0 || T <= 0 || Hr <= 0 || Wr <= 0 || S <= 0 || (S & 1) || S > Hr || S > Wr || (layout != 0 && layout != 1)) return MD_ERR_BAD_SHAPE;
  const int64_t n = (int64_t)B * T * S * S;
  int64_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
  MD_KLAUNCH(k_clip_preprocess, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames, B, T, Hr, Wr, S, Hr / 2 - S / 2,
             Wr / 2 - S / 2, mean_bgr[0], mean_bgr[1], mean_bgr[2], layout, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}


// The six augmentations of DatasetForVideo.get_video_data (src/dataset.py:129-135, 152-227) between the crop and the mean
// subtraction, decided on the host (src/utils/clip_preprocess.py::draw_augmentation, same random-number calls as the
// reference) and applied here per output pixel: brightness (add + clip to [10,255], or add + horizontal flip), contrast
// (cv2.convertScaleAbs: |alpha x| rounded half to even, saturated to 0..255), Gaussian blur (separable, row pass then column
// pass in fp32, BORDER_REFLECT_101), the two edge masks the reference calls shifts.  par [B][10] int32: mode_b (0 none, 1 add+clip,
// 2 add+flip), bright, contrast on, alpha, blur on, ksize, row_lo, row_hi, col_lo, col_hi.  gk: ksize Gaussian weights.
__device__ __forceinline__ float aug_point(const unsigned char* fr, int Wr, int y0, int x0, int S, int yy, int xx, int c, int mode_b,
                                           float bright, int contrast, float alpha) {
  const int xs = mode_b == 2 ? S - 1 - xx : xx;
  float v = (float)fr[((int64_t)(y0 + yy) * Wr + (x0 + xs)) * 3 + c];
  if (mode_b == 1) v = fminf(fmaxf(__fadd_rn(v, bright), 10.f), 255.f);
  else if (mode_b == 2) v = __fadd_rn(v, bright);
  if (contrast) v = fminf(fmaxf(rintf(fabsf(__fmul_rn(v, alpha))), 0.f), 255.f);
  return v;
}
__global__ __launch_bounds__(256) void k_clip_augment(const unsigned char* __restrict__ frames, int B, int T, int Hr, int Wr, int S,
                                                     int y0, int x0, float m0, float m1, float m2, int layout,
                                                     const int* __restrict__ par, const float* __restrict__ gk,
                                                     float* __restrict__ out) {
  const int64_t n = (int64_t)B * T * S * S;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % S); int64_t r = e / S;
    const int y = (int)(r % S); r /= S;
    const int t = (int)(r % T), b = (int)(r / T);
    const int* q = par + b * 10;
    const int mode_b = q[0], contrast = q[2], blur = q[4], ks = q[5];
    const float bright = (float)q[1], alpha = (float)q[3];
    const unsigned char* fr = frames + ((int64_t)b * T + t) * Hr * Wr * 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (y >= q[6] && y < q[7] && x >= q[8] && x < q[9]) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (!blur) { v[c] = aug_point(fr, Wr, y0, x0, S, y, x, c, mode_b, bright, contrast, alpha); continue; }
        const int h = ks >> 1;
        float acc = 0.f;
        for (int i = 0; i < ks; ++i) {
          int yy = y + i - h; yy = yy < 0 ? -yy : yy; yy = yy >= S ? 2 * S - 2 - yy : yy;
          float row = 0.f;
          for (int j = 0; j < ks; ++j) {
            int xx = x + j - h; xx = xx < 0 ? -xx : xx; xx = xx >= S ? 2 * S - 2 - xx : xx;
            row = __fadd_rn(row, __fmul_rn(gk[j], aug_point(fr, Wr, y0, x0, S, yy, xx, c, mode_b, bright, contrast, alpha)));
          }
          acc = __fadd_rn(acc, __fmul_rn(gk[i], row));
        }
        v[c] = acc;
      }
    }
    const float v0 = v[0] - m0, v1 = v[1] - m1, v2 = v[2] - m2;
    if (layout == 0) {
      const int64_t plane = (int64_t)T * S * S, o = (int64_t)b * 3 * plane + ((int64_t)t * S + y) * S + x;
      out[o] = v0; out[o + plane] = v1; out[o + 2 * plane] = v2;
    } else {
      *(float4*)(out + e * 4) = make_float4(v0, v1, v2, 0.f);
    }
  }
}
extern "C" int md_clip_augment_preprocess(const uint8_t* frames, int32_t B, int32_t T, int32_t Hr, int32_t Wr, int32_t S,
                                          const float* mean_bgr, int32_t layout, const int32_t* params, const float* gauss, float* out,
                                          void* stream) {
  if (!frames || !mean_bgr || !out || !params || !gauss) return MD_ERR_NULL;
  if (B <= 0 || T <= 0 || Hr <= 0 || Wr <= 0 || S <= 0 || (S & 1) || S > Hr || S > Wr || (layout != 0 && layout != 1)) return MD_ERR_BAD_SHAPE;
  const int64_t n = (int64_t)B * T * S * S;
  int64_t blocks = (n + 255) / 256; if (blocks > 65536) blocks = 65536;
  MD_KLAUNCH(k_clip_augment, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames, B, T, Hr, Wr, S, Hr / 2 - S / 2,
             Wr / 2 - S / 2, mean_bgr[0], mean_bgr[1], mean_bgr[2], layout, params, gauss, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}


// ---------------------------------------------------------------- class-balanced re-sampling (ImbalancedDatasetSampler)
// torch.multinomial(weights, n, replacement=True) on the CPU (the reference's sampler, src/utils/sampler.py:29-32) is: cumulative
// distribution of the normalised weights, then for each draw a uniform double u and the LEFTMOST category whose cumulative
// probability is >= u (aten/src/ATen/native/cpu/MultinomialKernel.cpp).  The uniforms are a serial Mersenne-twister stream and
// stay on the host generator; the search -- the O(n log C) part -- runs here, and only over this rank's share of the stream:
// rank r of W takes draws r, r + W, ... (DistributedSampler's partition applied to the resampled index stream).
// out[k] = index_map[search(u[rank + k * world])]   (index_map optional: the sampler's `indices` list)
__global__ __launch_bounds__(256) void k_multinomial_shard(const double* __restrict__ cum, int64_t ncat, const double* __restrict__ u,
                                                           int64_t nsamples, int rank, int world, const int64_t* __restrict__ index_map,
                                                           int64_t* __restrict__ out, int64_t nout) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nout) return;
  const double x = u[rank + k * (int64_t)world];
  int64_t left = 0, right = ncat;
  while (right - left > 0) {
    const int64_t mid = left + (right - left) / 2;
    if (cum[mid] < x) left = mid + 1; else right = mid;
  }
  if (left >= ncat) left = ncat - 1;               // (cum[ncat-1] == 1 >= u always; guard against a malformed table)
  out[k] = index_map ? index_map[left] : left;
}
extern "C" int64_t md_multinomial_shard_count(int64_t nsamples, int32_t rank, int32_t world) {
  if (nsamples <= 0 || world <= 0 || rank < 0 || rank >= world || rank >= nsamples) return 0;
  return (nsamples - rank + world - 1) / world;
}
extern "C" int md_multinomial_shard(const double* cum_dist, int64_t ncat, const double* uniforms, int64_t nsamples, int32_t rank,
                                    int32_t world, const int64_t* index_map, int64_t* out, void* stream) {
  if (!cum_dist || !uniforms || !out) return MD_ERR_NULL;
  if (ncat <= 0 || nsamples <= 0 || world <= 0 || rank < 0 || rank >= world) return MD_ERR_BAD_SHAPE;
  const int64_t nout = md_multinomial_shard_count(nsamples, rank, world);
  if (nout == 0) return MD_OK;
  MD_KLAUNCH(k_multinomial_shard, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cum_dist, ncat, uniforms,
             nsamples, rank, world, index_map, out, nout);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
