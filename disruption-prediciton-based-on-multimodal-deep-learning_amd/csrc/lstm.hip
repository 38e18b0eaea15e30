// One direction of one nn.LSTM layer (batch_first=False, zero initial state), forward and backward through time
// (reference users: src/models/CnnLSTM.py:51,93-96 and src/models/MLSTM_FCN.py).
// PyTorch conventions: gates in the order i, f, g, o; pre-activation = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh;
// c_t = f*c_{t-1} + i*g;  h_t = o * tanh(c_t).
// The sequences are tiny (S <= a few dozen steps, H <= 128) and strictly sequential in time: one workgroup per batch element
// walks the steps with h_{t-1} in LDS (latency-bound by construction); the weight gradients are a separate reduction over
// all (t, b) pairs.  Everything is fixed-order fp32 (bitwise reproducible).
#include "common.h"

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// x [S][B][I], h_all / c_all [S][B][H], gates [S][B][4H] (activated i, f, g, o); reverse != 0 walks t = S-1 .. 0
__global__ __launch_bounds__(256) void k_lstm_fwd(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                 const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                 const float* __restrict__ b_hh, int S, int B, int I, int H, int reverse,
                                                 float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ gates) {
  extern __shared__ float sm[];
  float* hprev = sm;            // [H]
  float* cprev = sm + H;        // [H]
  float* xs = sm + 2 * H;       // [I]
  float* pre = xs + I;          // [4H]
  const int b = blockIdx.x, t = threadIdx.x;
  for (int k = t; k < H; k += 256) { hprev[k] = 0.f; cprev[k] = 0.f; }
  for (int step = 0; step < S; ++step) {
    const int ts = reverse ? S - 1 - step : step;
    const float* xt = x + ((size_t)ts * B + b) * I;
    for (int i = t; i < I; i += 256) xs[i] = xt[i];
    __syncthreads();
    for (int j = t; j < 4 * H; j += 256) {
      float a = b_ih[j] + b_hh[j];
      const float* wi = w_ih + (size_t)j * I; const float* wh = w_hh + (size_t)j * H;
      for (int i = 0; i < I; ++i) a = fmaf(wi[i], xs[i], a);
      for (int k = 0; k < H; ++k) a = fmaf(wh[k], hprev[k], a);
      pre[j] = a;
    }
    __syncthreads();
    float* gt = gates + ((size_t)ts * B + b) * 4 * H;
    for (int k = t; k < H; k += 256) {
      const float ig = sigm(pre[k]), fg = sigm(pre[H + k]), gg = tanhf(pre[2 * H + k]), og = sigm(pre[3 * H + k]);
      const float c = fg * cprev[k] + ig * gg;
      const float h = og * tanhf(c);
      gt[k] = ig; gt[H + k] = fg; gt[2 * H + k] = gg; gt[3 * H + k] = og;
      c_all[((size_t)ts * B + b) * H + k] = c;
      h_all[((size_t)ts * B + b) * H + k] = h;
      cprev[k] = c; hprev[k] = h;
    }
    __syncthreads();
  }
}

// dh_all [S][B][H] = gradient w.r.t. every output h_t.  Writes dpre [S][B][4H] (gradient w.r.t. the gate pre-activations)
// and dx [S][B][I].
__global__ __launch_bounds__(256) void k_lstm_bwd(const float* __restrict__ dh_all, const float* __restrict__ w_ih,
                                                 const float* __restrict__ w_hh, const float* __restrict__ c_all,
                                                 const float* __restrict__ gates, int S, int B, int I, int H, int reverse,
                                                 float* __restrict__ dpre, float* __restrict__ dx) {
  extern __shared__ float sm[];
  float* dh = sm;               // [H] recurrent part of dL/dh_t
  float* dc = sm + H;           // [H] dL/dc_t carried backwards
  float* dg = sm + 2 * H;       // [4H]
  const int b = blockIdx.x, t = threadIdx.x;
  for (int k = t; k < H; k += 256) { dh[k] = 0.f; dc[k] = 0.f; }
  __syncthreads();
  for (int step = S - 1; step >= 0; --step) {
    const int ts = reverse ? S - 1 - step : step;                  // the step-th processed time index
    const int tprev = reverse ? ts + 1 : ts - 1;                   // time index processed just before it (none at step 0)
    const size_t row = (size_t)ts * B + b;
    const float* gt = gates + row * 4 * H;
    for (int k = t; k < H; k += 256) {
      const float ig = gt[k], fg = gt[H + k], gg = gt[2 * H + k], og = gt[3 * H + k];
      const float c = c_all[row * H + k];
      const float cp = step > 0 ? c_all[((size_t)tprev * B + b) * H + k] : 0.f;
      const float tc = tanhf(c);
      const float dht = dh_all[row * H + k] + dh[k];
      const float dct = dc[k] + dht * og * (1.f - tc * tc);
      dg[k] = dct * gg * ig * (1.f - ig);                          // d pre_i
      dg[H + k] = dct * cp * fg * (1.f - fg);                      // d pre_f
      dg[2 * H + k] = dct * ig * (1.f - gg * gg);                  // d pre_g
      dg[3 * H + k] = dht * tc * og * (1.f - og);                  // d pre_o
      dc[k] = dct * fg;
    }
    __syncthreads();
    float* dp = dpre + row * 4 * H;
    for (int j = t; j < 4 * H; j += 256) dp[j] = dg[j];
    for (int k = t; k < H; k += 256) {                             // dh_{prev} = W_hh^T dpre
      float a = 0.f;
      for (int j = 0; j < 4 * H; ++j) a = fmaf(w_hh[(size_t)j * H + k], dg[j], a);
      dh[k] = a;
    }
    for (int i = t; i < I; i += 256) {                             // dx_t = W_ih^T dpre
      float a = 0.f;
      for (int j = 0; j < 4 * H; ++j) a = fmaf(w_ih[(size_t)j * I + i], dg[j], a);
      dx[row * I + i] = a;
    }
    __syncthreads();
  }
}

// dW_ih [4H][I], dW_hh [4H][H], db [4H] (the same for b_ih and b_hh): sums over all (t, b), rows in index order.
__global__ __launch_bounds__(256) void k_lstm_wgrad(const float* __restrict__ dpre, const float* __restrict__ x,
                                                   const float* __restrict__ h_all, int S, int B, int I, int H, int reverse,
                                                   float* __restrict__ dw_ih, float* __restrict__ dw_hh, float* __restrict__ db) {
  const int total = 4 * H * (I + H + 1);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int j = e / (I + H + 1), c = e - j * (I + H + 1);
    float a = 0.f;
    for (int ts = 0; ts < S; ++ts) {
      const int tprev = reverse ? ts + 1 : ts - 1;
      const bool has_prev = reverse ? ts + 1 < S : ts > 0;
      for (int b = 0; b < B; ++b) {
        const float d = dpre[((size_t)ts * B + b) * 4 * H + j];
        float v;
        if (c < I) v = x[((size_t)ts * B + b) * I + c];
        else if (c < I + H) v = has_prev ? h_all[((size_t)tprev * B + b) * H + (c - I)] : 0.f;
        else v = 1.f;
        a = fmaf(d, v, a);
      }
    }
    if (c < I) dw_ih[(size_t)j * I + c] = a;
    else if (c < I + H) dw_hh[(size_t)j * H + (c - I)] = a;
    else db[j] = a;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Register-resident recurrence (H = 64 or 128; MLSTM_FCN's and CnnLSTM's sizes).  The kernels above stream W_ih and W_hh from
// memory at every time step, one uncoalesced row per thread: 73 us per step at H = 128, I = 256, all of it on the sequential
// path (cfg5: 30 of 41 ms per training step).  Here the input projection x W_ih^T for ALL steps is one MFMA GEMM outside
// (LinearRowsFunction, which also owns dx and dW_ih through its own backward), and the workgroup of 4H threads keeps W_hh in
// registers for the whole sequence: thread j holds row j (forward: pre_j = xproj_j + b_j + <W_hh[j], h>), or, backward, column k of
// one quarter of the rows (dh_k = sum_j W_hh[j][k] dpre_j as four partial sums added in a fixed order).  Per step only h / dpre
// travel through LDS as broadcast ds_read_b128.
template <int H>
__device__ __forceinline__ void lstm_rec_fwd_body(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                  const float* __restrict__ b_ih, const float* __restrict__ b_hh, int S, int B,
                                                  int reverse, float* __restrict__ h_all, float* __restrict__ c_all,
                                                  float* __restrict__ gates) {
  __shared__ __attribute__((aligned(16))) float hprev[H];
  __shared__ float cprev[H];
  __shared__ float pre[4 * H];
  const int b = blockIdx.x, j = threadIdx.x;
  float w[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 v = *(const float4*)(w_hh + (size_t)j * H + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  const float bias = b_ih[j] + b_hh[j];
  if (j < H) { hprev[j] = 0.f; cprev[j] = 0.f; }
  __syncthreads();
  for (int step = 0; step < S; ++step) {
    const int ts = reverse ? S - 1 - step : step;
    const size_t row = (size_t)ts * B + b;
    float a = xproj[row * 4 * H + j] + bias;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      const float4 hv = *(const float4*)(hprev + k);
      a = fmaf(w[k], hv.x, a); a = fmaf(w[k + 1], hv.y, a); a = fmaf(w[k + 2], hv.z, a); a = fmaf(w[k + 3], hv.w, a);
    }
    pre[j] = a;
    __syncthreads();
    if (j < H) {
      const float ig = sigm(pre[j]), fg = sigm(pre[H + j]), gg = tanhf(pre[2 * H + j]), og = sigm(pre[3 * H + j]);
      const float c = fg * cprev[j] + ig * gg;
      const float h = og * tanhf(c);
      float* gt = gates + row * 4 * H;
      gt[j] = ig; gt[H + j] = fg; gt[2 * H + j] = gg; gt[3 * H + j] = og;
      c_all[row * H + j] = c; h_all[row * H + j] = h;
      cprev[j] = c; hprev[j] = h;
    }
    __syncthreads();
  }
}
template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_rec_fwd(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh, int S, int B,
                                                       int reverse, float* __restrict__ h_all, float* __restrict__ c_all,
                                                       float* __restrict__ gates) {
  lstm_rec_fwd_body<H>(xproj, w_hh, b_ih, b_hh, S, B, reverse, h_all, c_all, gates);
}
// Both directions of a bidirectional layer in ONE launch (blockIdx.y = direction): the two recurrences are independent chains of S
// dependent steps (2 us each), and one after the other they were 8 x 45 us on cfg5's critical path for a model that occupies 8 CUs.
struct LstmRec2 { const float* xproj[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2]; float* h_all[2]; float* c_all[2];
                  float* gates[2]; };
template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_rec_fwd2(LstmRec2 a, int S, int B) {
  const int d = blockIdx.y;
  lstm_rec_fwd_body<H>(a.xproj[d], a.w_hh[d], a.b_ih[d], a.b_hh[d], S, B, d, a.h_all[d], a.c_all[d], a.gates[d]);
}
template <int H>
__device__ __forceinline__ void lstm_rec_bwd_body(const float* __restrict__ dh_all, const float* __restrict__ w_hh,
                                                  const float* __restrict__ c_all, const float* __restrict__ gates, int S, int B,
                                                  int reverse, float* __restrict__ dpre) {
  __shared__ float dh[H];
  __shared__ float dc[H];
  __shared__ __attribute__((aligned(16))) float dg[4 * H];
  __shared__ float part[4][H];
  const int b = blockIdx.x, t = threadIdx.x, p = t / H, k = t - p * H;
  float w[H];                                            // w[jj] = W_hh[p H + jj][k]
#pragma unroll
  for (int jj = 0; jj < H; ++jj) w[jj] = w_hh[(size_t)(p * H + jj) * H + k];
  if (t < H) { dh[t] = 0.f; dc[t] = 0.f; }
  __syncthreads();
  for (int step = S - 1; step >= 0; --step) {
    const int ts = reverse ? S - 1 - step : step;
    const int tprev = reverse ? ts + 1 : ts - 1;
    const size_t row = (size_t)ts * B + b;
    if (t < H) {
      const float* gt = gates + row * 4 * H;
      const float ig = gt[t], fg = gt[H + t], gg = gt[2 * H + t], og = gt[3 * H + t];
      const float c = c_all[row * H + t];
      const float cp = step > 0 ? c_all[((size_t)tprev * B + b) * H + t] : 0.f;
      const float tc = tanhf(c);
      const float dht = dh_all[row * H + t] + dh[t];
      const float dct = dc[t] + dht * og * (1.f - tc * tc);
      dg[t] = dct * gg * ig * (1.f - ig);
      dg[H + t] = dct * cp * fg * (1.f - fg);
      dg[2 * H + t] = dct * ig * (1.f - gg * gg);
      dg[3 * H + t] = dht * tc * og * (1.f - og);
      dc[t] = dct * fg;
    }
    __syncthreads();
    dpre[row * 4 * H + t] = dg[t];
    float a = 0.f;
#pragma unroll
    for (int jj = 0; jj < H; jj += 4) {
      const float4 d = *(const float4*)(dg + p * H + jj);
      a = fmaf(w[jj], d.x, a); a = fmaf(w[jj + 1], d.y, a); a = fmaf(w[jj + 2], d.z, a); a = fmaf(w[jj + 3], d.w, a);
    }
    part[p][k] = a;
    __syncthreads();
    if (t < H) dh[t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    __syncthreads();
  }
}
template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_rec_bwd(const float* __restrict__ dh_all, const float* __restrict__ w_hh,
                                                       const float* __restrict__ c_all, const float* __restrict__ gates, int S, int B,
                                                       int reverse, float* __restrict__ dpre) {
  lstm_rec_bwd_body<H>(dh_all, w_hh, c_all, gates, S, B, reverse, dpre);
}
struct LstmRecBwd2 { const float* dh_all[2]; const float* w_hh[2]; const float* h_all[2]; const float* c_all[2]; const float* gates[2];
                     float* dpre[2]; float* dw_hh[2]; float* db[2]; };
template <int H>
__global__ __launch_bounds__(4 * H) void k_lstm_rec_bwd2(LstmRecBwd2 a, int S, int B) {
  const int d = blockIdx.y;
  lstm_rec_bwd_body<H>(a.dh_all[d], a.w_hh[d], a.c_all[d], a.gates[d], S, B, d, a.dpre[d]);
}
// dW_hh / db of both directions (k_lstm_wgrad with I = 0), blockIdx.y = direction
__global__ __launch_bounds__(256) void k_lstm_wgrad2(LstmRecBwd2 a, int S, int B, int H) {
  const int d = blockIdx.y;
  const float* __restrict__ dpre = a.dpre[d]; const float* __restrict__ h_all = a.h_all[d];
  const int total = 4 * H * (H + 1);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int j = e / (H + 1), c = e - j * (H + 1);
    float acc = 0.f;
    for (int ts = 0; ts < S; ++ts) {
      const int tprev = d ? ts + 1 : ts - 1;
      const bool has_prev = d ? ts + 1 < S : ts > 0;
      for (int b = 0; b < B; ++b) {
        const float g = dpre[((size_t)ts * B + b) * 4 * H + j];
        const float v = c < H ? (has_prev ? h_all[((size_t)tprev * B + b) * H + c] : 0.f) : 1.f;
        acc = fmaf(g, v, acc);
      }
    }
    if (c < H) a.dw_hh[d][(size_t)j * H + c] = acc; else a.db[d][j] = acc;
  }
}
extern "C" int md_lstm_rec_supported(int32_t H) { return (H == 64 || H == 128) ? 1 : 0; }
// Both directions of one bidirectional layer: index 0 = forward in time, 1 = reverse; arrays of two device pointers each.
extern "C" int md_lstm_rec_fwd2(const float* const* xproj, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                                int32_t S, int32_t B, int32_t H, float* const* h_all, float* const* c_all, float* const* gates, void* stream) {
  if (!xproj || !w_hh || !b_ih || !b_hh || !h_all || !c_all || !gates) return MD_ERR_NULL;
  LstmRec2 a;
  for (int d = 0; d < 2; ++d) {
    if (!xproj[d] || !w_hh[d] || !b_ih[d] || !b_hh[d] || !h_all[d] || !c_all[d] || !gates[d]) return MD_ERR_NULL;
    a.xproj[d] = xproj[d]; a.w_hh[d] = w_hh[d]; a.b_ih[d] = b_ih[d]; a.b_hh[d] = b_hh[d]; a.h_all[d] = h_all[d]; a.c_all[d] = c_all[d];
    a.gates[d] = gates[d];
  }
  if (S <= 0 || B <= 0) return MD_ERR_BAD_SHAPE;
  if (!md_lstm_rec_supported(H)) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (H == 64) MD_KLAUNCH(k_lstm_rec_fwd2<64>, dim3(B, 2), dim3(256), 0, s, a, S, B);
  else MD_KLAUNCH(k_lstm_rec_fwd2<128>, dim3(B, 2), dim3(512), 0, s, a, S, B);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// dw_hh / db: both NULL arrays = the caller reduces dpre itself (as md_lstm_rec_bwd)
extern "C" int md_lstm_rec_bwd2(const float* const* dh_all, const float* const* w_hh, const float* const* h_all, const float* const* c_all,
                                const float* const* gates, int32_t S, int32_t B, int32_t H, float* const* dpre, float* const* dw_hh,
                                float* const* db, void* stream) {
  if (!dh_all || !w_hh || !h_all || !c_all || !gates || !dpre || (!dw_hh != !db)) return MD_ERR_NULL;
  LstmRecBwd2 a;
  for (int d = 0; d < 2; ++d) {
    if (!dh_all[d] || !w_hh[d] || !h_all[d] || !c_all[d] || !gates[d] || !dpre[d]) return MD_ERR_NULL;
    if (dw_hh && (!dw_hh[d] || !db[d])) return MD_ERR_NULL;
    a.dh_all[d] = dh_all[d]; a.w_hh[d] = w_hh[d]; a.h_all[d] = h_all[d]; a.c_all[d] = c_all[d]; a.gates[d] = gates[d]; a.dpre[d] = dpre[d];
    a.dw_hh[d] = dw_hh ? dw_hh[d] : nullptr; a.db[d] = db ? db[d] : nullptr;
  }
  if (S <= 0 || B <= 0) return MD_ERR_BAD_SHAPE;
  if (!md_lstm_rec_supported(H)) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (H == 64) MD_KLAUNCH(k_lstm_rec_bwd2<64>, dim3(B, 2), dim3(256), 0, s, a, S, B);
  else MD_KLAUNCH(k_lstm_rec_bwd2<128>, dim3(B, 2), dim3(512), 0, s, a, S, B);
  MD_CHECK_LAUNCH();
  if (!dw_hh) return MD_OK;
  MD_KLAUNCH(k_lstm_wgrad2, dim3(md_cdiv(4 * H * (H + 1), 256), 2), dim3(256), 0, s, a, S, B, H);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// xproj [S][B][4H] = x W_ih^T (no bias); the biases are added here.
extern "C" int md_lstm_rec_fwd(const float* xproj, const float* w_hh, const float* b_ih, const float* b_hh, int32_t S, int32_t B, int32_t H,
                               int32_t reverse, float* h_all, float* c_all, float* gates, void* stream) {
  if (!xproj || !w_hh || !b_ih || !b_hh || !h_all || !c_all || !gates) return MD_ERR_NULL;
  if (S <= 0 || B <= 0) return MD_ERR_BAD_SHAPE;
  if (!md_lstm_rec_supported(H)) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (H == 64) MD_KLAUNCH(k_lstm_rec_fwd<64>, dim3(B), dim3(256), 0, s, xproj, w_hh, b_ih, b_hh, S, B, reverse, h_all, c_all, gates);
  else MD_KLAUNCH(k_lstm_rec_fwd<128>, dim3(B), dim3(512), 0, s, xproj, w_hh, b_ih, b_hh, S, B, reverse, h_all, c_all, gates);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
// dpre [S][B][4H] (= gradient of xproj), dw_hh [4H][H], db [4H]
extern "C" int md_lstm_rec_bwd(const float* dh_all, const float* w_hh, const float* h_all, const float* c_all, const float* gates,
                               int32_t S, int32_t B, int32_t H, int32_t reverse, float* dpre, float* dw_hh, float* db, void* stream) {
  if (!dh_all || !w_hh || !h_all || !c_all || !gates || !dpre || (!dw_hh != !db)) return MD_ERR_NULL;
  if (S <= 0 || B <= 0) return MD_ERR_BAD_SHAPE;
  if (!md_lstm_rec_supported(H)) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (H == 64) MD_KLAUNCH(k_lstm_rec_bwd<64>, dim3(B), dim3(256), 0, s, dh_all, w_hh, c_all, gates, S, B, reverse, dpre);
  else MD_KLAUNCH(k_lstm_rec_bwd<128>, dim3(B), dim3(512), 0, s, dh_all, w_hh, c_all, gates, S, B, reverse, dpre);
  MD_CHECK_LAUNCH();
  if (!dw_hh) return MD_OK;          // the caller reduces dpre itself (many rows: MFMA weight gradient + column sums)
  const int total = 4 * H * (H + 1);
  MD_KLAUNCH(k_lstm_wgrad, dim3(md_cdiv(total, 256)), dim3(256), 0, s, (const float*)dpre, (const float*)nullptr, h_all, S, B, 0, H, reverse,
             (float*)nullptr, dw_hh, db);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                           int32_t S, int32_t B, int32_t I, int32_t H, int32_t reverse, float* h_all, float* c_all,
                           float* gates, void* stream) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !h_all || !c_all || !gates) return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || I <= 0 || H <= 0) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)(6 * H + I) * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  MD_KLAUNCH(k_lstm_fwd, dim3(B), dim3(256), lds, (hipStream_t)stream, x, w_ih, w_hh, b_ih, b_hh, S, B, I, H, reverse, h_all,
             c_all, gates);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_lstm_bwd(const float* dh_all, const float* x, const float* w_ih, const float* w_hh, const float* h_all,
                           const float* c_all, const float* gates, int32_t S, int32_t B, int32_t I, int32_t H, int32_t reverse,
                           float* dx, float* dw_ih, float* dw_hh, float* db, float* dpre_scratch, void* stream) {
  if (!dh_all || !x || !w_ih || !w_hh || !h_all || !c_all || !gates || !dx || !dw_ih || !dw_hh || !db || !dpre_scratch)
    return MD_ERR_NULL;
  if (S <= 0 || B <= 0 || I <= 0 || H <= 0) return MD_ERR_BAD_SHAPE;
  const size_t lds = (size_t)(6 * H) * 4;
  if (lds > 60000) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  MD_KLAUNCH(k_lstm_bwd, dim3(B), dim3(256), lds, s, dh_all, w_ih, w_hh, c_all, gates, S, B, I, H, reverse, dpre_scratch, dx);
  MD_CHECK_LAUNCH();
  const int total = 4 * H * (I + H + 1);
  MD_KLAUNCH(k_lstm_wgrad, dim3(md_cdiv(total, 256)), dim3(256), 0, s, (const float*)dpre_scratch, x, h_all, S, B, I, H, reverse,
             dw_ih, dw_hh, db);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
