// BatchNorm3d (train) + LeakyReLU + residual pieces that sit between two convolutions of the reference's
// Conv3dBlock / SpatioTemporalResBlock (src/models/R2Plus1D.py:53-58, :179-187), for channels-last
// fp32 tensors [rows][Cp].  All kernels are HBM-bound streaming passes: every thread owns one fixed
// 16-byte channel chunk (so per-channel constants are loaded once) and walks rows; consecutive threads
// touch consecutive 16-byte chunks, i.e. fully coalesced 1 KiB wave accesses.
#include "common.h"
#include "patch_common.h"

struct View {
  const float* p; const float* scale; const float* shift; float slope;
};
static inline View to_view(const MdActView* v) {
  View o; o.p = v ? v->data : nullptr; o.scale = v ? v->scale : nullptr; o.shift = v ? v->shift : nullptr;
  o.slope = v ? v->slope : 1.f; return o;
}

struct ChanConst { float4 sc, sh; };
__device__ __forceinline__ ChanConst load_cc(const View& v, int c4) {
  ChanConst k;
  if (v.scale) { k.sc = *(const float4*)(v.scale + c4 * 4); k.sh = *(const float4*)(v.shift + c4 * 4); }
  else { k.sc = make_float4(1.f, 1.f, 1.f, 1.f); k.sh = make_float4(0.f, 0.f, 0.f, 0.f); }
  return k;
}
__device__ __forceinline__ float4 pre_of(const View& v, const ChanConst& k, float4 x) {
  if (!v.scale) return x;
  return make_float4(fmaf(x.x, k.sc.x, k.sh.x), fmaf(x.y, k.sc.y, k.sh.y), fmaf(x.z, k.sc.z, k.sh.z), fmaf(x.w, k.sc.w, k.sh.w));
}
__device__ __forceinline__ float4 act_of(const View& v, float4 pre) {
  if (!v.scale) return pre;
  return make_float4(md_leaky(pre.x, v.slope), md_leaky(pre.y, v.slope), md_leaky(pre.z, v.slope), md_leaky(pre.w, v.slope));
}
__device__ __forceinline__ float4 dact_of(const View& v, float4 pre) {
  if (!v.scale) return make_float4(1.f, 1.f, 1.f, 1.f);
  return make_float4(md_dleaky(pre.x, v.slope), md_dleaky(pre.y, v.slope), md_dleaky(pre.z, v.slope), md_dleaky(pre.w, v.slope));
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// d_raw of one element from g, the raw conv output and the per-channel constants; ONE definition for the fp32 and the
// pre-split apply kernels so that both evaluate it with the same operations (their outputs must agree bit for bit).
__device__ __forceinline__ float md_bn_apply1(float gq, float raw, float mu, float is, float sc, float c1, float c2) {
  const float xh = __fmul_rn(__fsub_rn(raw, mu), is);
  return __fmul_rn(sc, __fsub_rn(__fsub_rn(gq, c1), __fmul_rn(xh, c2)));
}

// Row walk shared by the streaming kernels: thread -> (chunk c4, row lane r); block -> row range.
struct RowWalk { int c4, r, nr; int64_t beg, end; bool active; };
// `bid`: which contiguous slice of rows this block walks (blockIdx.x, or the mirrored index for a pass that should
// start where the previous kernel finished writing, while those lines are still in L2 / Infinity Cache).
__device__ __forceinline__ RowWalk row_walk(int64_t rows, int C4, int bid = -1) {
  RowWalk w;
  if (bid < 0) bid = blockIdx.x;
  w.nr = blockDim.x / C4;
  w.c4 = threadIdx.x % C4;
  w.r = threadIdx.x / C4;
  w.active = w.r < w.nr;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  w.beg = (int64_t)bid * per;
  w.end = w.beg + per < rows ? w.beg + per : rows;
  return w;
}

static int stream_blocks(int64_t rows, int C4) {
  const int nr = 256 / C4;
  int64_t b = md_cdiv64(rows, (int64_t)nr * 8);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

// ---------------------------------------------------------------- forward elementwise
__global__ __launch_bounds__(256) void k_bn_act(View v, int64_t rows, int C4, float* __restrict__ out) {
  RowWalk w = row_walk(rows, C4);
  if (!w.active) return;
  const ChanConst k = load_cc(v, w.c4);
  for (int64_t row = w.beg + w.r; row < w.end; row += w.nr) {
    const size_t o = ((size_t)row * C4 + w.c4) * 4;
    *(float4*)(out + o) = act_of(v, pre_of(v, k, *(const float4*)(v.p + o)));
  }
}

__global__ __launch_bounds__(256) void k_residual_fwd(View skip, View main, float alpha, int64_t rows, int C4,
                                                      float* __restrict__ z) {
  RowWalk w = row_walk(rows, C4);
  if (!w.active) return;
  const ChanConst ks = load_cc(skip, w.c4), km = load_cc(main, w.c4);
  for (int64_t row = w.beg + w.r; row < w.end; row += w.nr) {
    const size_t o = ((size_t)row * C4 + w.c4) * 4;
    const float4 a = act_of(skip, pre_of(skip, ks, *(const float4*)(skip.p + o)));
    const float4 b = act_of(main, pre_of(main, km, *(const float4*)(main.p + o)));
    const float4 s = add4(a, b);
    *(float4*)(z + o) = make_float4(md_leaky(s.x, alpha), md_leaky(s.y, alpha), md_leaky(s.z, alpha), md_leaky(s.w, alpha));
  }
}

// Sum the per-workgroup partial rows [blocks][2][Cp] for channels 4*blockIdx.x .. +3 in fp64: 256 row lanes (row r,
// r+256, ...), then a fixed shuffle / LDS tree -- deterministic for a given partial buffer.  Result (8 doubles) in out[] of every thread.
__device__ __forceinline__ void reduce_partials4(const float* __restrict__ partial, int blocks, int Cp, double out[8]) {
  __shared__ double red[4][8];
  const int r = threadIdx.x;
  const int c0 = blockIdx.x * 4;
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = r; i < blocks; i += 256) {
    const float4 s1 = *(const float4*)(partial + (size_t)i * 2 * Cp + c0);
    const float4 s2 = *(const float4*)(partial + (size_t)i * 2 * Cp + Cp + c0);
    a[0] += s1.x; a[1] += s1.y; a[2] += s1.z; a[3] += s1.w;
    a[4] += s2.x; a[5] += s2.y; a[6] += s2.z; a[7] += s2.w;
  }
  // inside a wave by shuffles (no barriers: the eight-barrier LDS tree of round 2 was most of this 5 us kernel, which sits 64 times
  // per step on the dependent chain), then the four waves through LDS -- a fixed order either way
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += __shfl_xor(a[k], m);
  }
  if ((r & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[r >> 6][k] = a[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

// ---------------------------------------------------------------- BatchNorm statistics finalize
// grid = Cp/4 blocks of 256 row lanes (reduce_partials4); threads 0..3 then finish one channel each.
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ partial, int blocks, int C, int Cp,
                                                     double inv_count, double unbias, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float momentum,
                                                     float* __restrict__ rmean, float* __restrict__ rvar,
                                                     float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                     float* __restrict__ scale_o, float* __restrict__ shift_o) {
  double tot[8];
  reduce_partials4(partial, blocks, Cp, tot);
  const int cl = threadIdx.x, r = threadIdx.x >= 4 ? 1 : 0;      // threads 0..3 finish one channel each
  const int c = blockIdx.x * 4 + (cl & 3);
  double s1[1][4] = {{tot[0], tot[1], tot[2], tot[3]}}, s2[1][4] = {{tot[4], tot[5], tot[6], tot[7]}};
  if (r == 0 && c < Cp) {
    float mean = 0.f, invstd = 0.f, sc = 0.f, sh = 0.f;
    if (c < C) {
      const double m = s1[0][cl & 3] * inv_count;
      double var = s2[0][cl & 3] * inv_count - m * m;
      if (var < 0.0) var = 0.0;
      const double is = 1.0 / sqrt(var + (double)eps);
      mean = (float)m; invstd = (float)is;
      sc = gamma[c] * invstd;
      sh = beta[c] - mean * sc;
      if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);
    }
    mean_o[c] = mean; invstd_o[c] = invstd; scale_o[c] = sc; shift_o[c] = sh;
  }
}

// eval mode: scale/shift from the running statistics (nn.BatchNorm3d.eval()).
__global__ void k_bn_eval_params(int C, int Cp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                 float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                 float* __restrict__ shift_o) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float mean = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
  if (c < C) { mean = rmean[c]; is = 1.f / sqrtf(rvar[c] + eps); sc = gamma[c] * is; sh = beta[c] - mean * sc; }
  mean_o[c] = mean; invstd_o[c] = is; scale_o[c] = sc; shift_o[c] = sh;
}

// ---------------------------------------------------------------- BatchNorm backward
// g = dA * leaky'_main(pre_main); closing form: dS = dZ * leaky'_alpha(act(skip) + act(main)), g = dS * leaky'(pre).
// GIN: dA already holds g (the fused reduction of the consumer's data gradient multiplied by leaky' on the way out).
// Fused finalize (APPLY only, fin.partial != nullptr): instead of reading the coefficients that md_bn_bwd_finalize would have
// written, every workgroup sums the partial rows itself (<= 256 rows of the 1024-thread reduction pass or of a persistent data
// gradient; fp64, fixed order: row lanes r, r + nr, ..., then lane by lane) -- one dependent launch less per unit on the backward
// chain; workgroup 0 also writes dgamma / dbeta.
struct BnFin { const float* partial; int blocks; double inv_count; float* dgamma; float* dbeta; int C; };
template <bool APPLY, bool GIN = false, int NTH = 256>
__global__ __launch_bounds__(NTH) void k_bn_bwd(const float* __restrict__ dA, View main, View skip, int flags,
                                                float alpha, const float* __restrict__ mean,
                                                const float* __restrict__ invstd, const float* __restrict__ coef,
                                                int64_t rows, int C4, float* __restrict__ partial,
                                                float* __restrict__ d_raw, float* __restrict__ dS, BnFin fin) {
  __shared__ float4 red[NTH];
  // The reduction pass walks the tensor back to front: dA was just written front to back by the data-gradient kernel,
  // and the apply pass that follows (front to back) then starts on what this pass touched last.
  const int has_skip = flags & 1;
  const int bid = (APPLY || (flags & 2)) ? (int)blockIdx.x : (int)(gridDim.x - 1 - blockIdx.x);
  RowWalk w = row_walk(rows, C4, bid);
  const int Cp = C4 * 4;
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  float4 fc1 = make_float4(0.f, 0.f, 0.f, 0.f), fc2 = fc1;
  if (APPLY && fin.partial) {
    __shared__ double fred[8][257];
    const int nrp = 256 / C4;                      // row lanes of the prologue: the first nrp * C4 <= 256 threads
    const bool pl = w.r < nrp;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pl) {
      for (int i = w.r; i < fin.blocks; i += nrp) {
        const float4 s1 = *(const float4*)(fin.partial + (size_t)i * 2 * Cp + w.c4 * 4);
        const float4 s2 = *(const float4*)(fin.partial + (size_t)i * 2 * Cp + Cp + w.c4 * 4);
        a[0] += s1.x; a[1] += s1.y; a[2] += s1.z; a[3] += s1.w;
        a[4] += s2.x; a[5] += s2.y; a[6] += s2.z; a[7] += s2.w;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) fred[k][threadIdx.x] = a[k];
    }
    __syncthreads();
    if (pl && w.r == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { double t = fred[k][w.c4]; for (int r = 1; r < nrp; ++r) t += fred[k][r * C4 + w.c4]; fred[k][w.c4] = t; }
    }
    __syncthreads();
    if (w.active) {
      double g[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = fred[k][w.c4];
      fc1 = make_float4((float)(g[0] * fin.inv_count), (float)(g[1] * fin.inv_count), (float)(g[2] * fin.inv_count), (float)(g[3] * fin.inv_count));
      fc2 = make_float4((float)(g[4] * fin.inv_count), (float)(g[5] * fin.inv_count), (float)(g[6] * fin.inv_count), (float)(g[7] * fin.inv_count));
      if (blockIdx.x == 0 && w.r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = w.c4 * 4 + e;
          if (c < fin.C) { if (fin.dbeta) fin.dbeta[c] = (float)g[e]; if (fin.dgamma) fin.dgamma[c] = (float)g[4 + e]; }
        }
      }
    }
  }
  if (w.active) {
    const ChanConst km = load_cc(main, w.c4);
    ChanConst ks; if (has_skip) ks = load_cc(skip, w.c4);
    const float4 mu = *(const float4*)(mean + w.c4 * 4), is = *(const float4*)(invstd + w.c4 * 4);
    float4 c1, c2;
    if (APPLY && !fin.partial) { c1 = *(const float4*)(coef + w.c4 * 4); c2 = *(const float4*)(coef + Cp + w.c4 * 4); }
    if (APPLY && fin.partial) { c1 = fc1; c2 = fc2; }
    // one row: same arithmetic and the same accumulation order whatever the unrolling below
    auto one = [&](size_t o, float4 raw, float4 d, float4 sraw) {
      const float4 pre = pre_of(main, km, raw);
      if (has_skip) {
        const float4 sv = act_of(skip, pre_of(skip, ks, sraw));
        const float4 sum = add4(sv, act_of(main, pre));
        d = mul4(d, make_float4(md_dleaky(sum.x, alpha), md_dleaky(sum.y, alpha), md_dleaky(sum.z, alpha), md_dleaky(sum.w, alpha)));
        if (APPLY) *(float4*)(dS + o) = d;
      }
      const float4 gq = GIN ? d : mul4(d, dact_of(main, pre));
      const float4 xh = make_float4((raw.x - mu.x) * is.x, (raw.y - mu.y) * is.y, (raw.z - mu.z) * is.z, (raw.w - mu.w) * is.w);
      if (APPLY) {
        float4 r;
        r.x = md_bn_apply1(gq.x, raw.x, mu.x, is.x, km.sc.x, c1.x, c2.x);
        r.y = md_bn_apply1(gq.y, raw.y, mu.y, is.y, km.sc.y, c1.y, c2.y);
        r.z = md_bn_apply1(gq.z, raw.z, mu.z, is.z, km.sc.z, c1.z, c2.z);
        r.w = md_bn_apply1(gq.w, raw.w, mu.w, is.w, km.sc.w, c1.w, c2.w);
        *(float4*)(d_raw + o) = r;
      } else {
        a1 = add4(a1, gq);
        a2 = add4(a2, mul4(gq, xh));
      }
    };
    // four rows per trip: 8-12 independent 16-byte loads in flight per lane (the pass is HBM-bound)
    int64_t row = w.beg + w.r;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; row + 3 * (int64_t)w.nr < w.end; row += 4 * (int64_t)w.nr) {
      size_t o[4]; float4 raw[4], d[4], sk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o[u] = ((size_t)(row + u * (int64_t)w.nr) * C4 + w.c4) * 4;
        raw[u] = *(const float4*)(main.p + o[u]);
        d[u] = *(const float4*)(dA + o[u]);
        sk[u] = has_skip ? *(const float4*)(skip.p + o[u]) : z4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(o[u], raw[u], d[u], sk[u]);
    }
    for (; row < w.end; row += w.nr) {
      const size_t o = ((size_t)row * C4 + w.c4) * 4;
      one(o, *(const float4*)(main.p + o), *(const float4*)(dA + o), has_skip ? *(const float4*)(skip.p + o) : z4);
    }
  }
  if (!APPLY) {
    // reduce over the row lanes of this block (fixed order), one partial row per block
    float* sp = partial + (size_t)bid * 2 * Cp;
    red[threadIdx.x] = a1;
    __syncthreads();
    if (w.r == 0 && w.active) {
      float4 s = red[w.c4];
      for (int r = 1; r < w.nr; ++r) s = add4(s, red[r * C4 + w.c4]);
      *(float4*)(sp + w.c4 * 4) = s;
    }
    __syncthreads();
    red[threadIdx.x] = a2;
    __syncthreads();
    if (w.r == 0 && w.active) {
      float4 s = red[w.c4];
      for (int r = 1; r < w.nr; ++r) s = add4(s, red[r * C4 + w.c4]);
      *(float4*)(sp + Cp + w.c4 * 4) = s;
    }
  }
}

// Apply pass writing d_raw in the pre-split bf16 format the data-gradient / weight-gradient kernels stage without any
// arithmetic: [row][C8 chunks]{hi 8 x bf16 | lo 8 x bf16}, 32 bytes per 8 channels (same bytes as fp32, Cp % 8 == 0, so
// the pass may run in place).  A thread owns one 8-channel chunk (per-channel constants loaded once) and walks rows.
// The hi/lo split is exactly the kernels' own split8(), so results are bit-identical to staging the fp32 tensor.
template <bool GIN>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_split(const float* __restrict__ dA, View main, View skip, int has_skip,
                                                            float alpha, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ coef,
                                                            int64_t rows, int C8, uint4* __restrict__ d_split,
                                                            float* __restrict__ dS) {
  const int nr = blockDim.x / C8, c8 = threadIdx.x % C8, r0 = threadIdx.x / C8;
  if (r0 >= nr) return;
  const int Cp = C8 * 8;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < rows ? beg + per : rows;
  float sc[8], sh[8], mu[8], is[8], c1[8], c2[8], ssc[8], ssh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    sc[e] = main.scale[c]; sh[e] = main.shift[c]; mu[e] = mean[c]; is[e] = invstd[c]; c1[e] = coef[c]; c2[e] = coef[Cp + c];
    ssc[e] = (has_skip && skip.scale) ? skip.scale[c] : 1.f; ssh[e] = (has_skip && skip.scale) ? skip.shift[c] : 0.f;
  }
  auto one = [&](int64_t row) {
    const size_t o = ((size_t)row * C8 + c8) * 8;
    const float4 ra = *(const float4*)(main.p + o), rb = *(const float4*)(main.p + o + 4);
    const float4 da = *(const float4*)(dA + o), db = *(const float4*)(dA + o + 4);
    float raw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
    float d[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
    if (has_skip) {
      const float4 sa = *(const float4*)(skip.p + o), sb = *(const float4*)(skip.p + o + 4);
      const float sk[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sv = skip.scale ? md_leaky(fmaf(sk[e], ssc[e], ssh[e]), skip.slope) : sk[e];
        const float sum = sv + md_leaky(fmaf(raw[e], sc[e], sh[e]), main.slope);
        d[e] *= md_dleaky(sum, alpha);
      }
      *(float4*)(dS + o) = make_float4(d[0], d[1], d[2], d[3]);
      *(float4*)(dS + o + 4) = make_float4(d[4], d[5], d[6], d[7]);
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pre = fmaf(raw[e], sc[e], sh[e]);
      const float gq = GIN ? d[e] : __fmul_rn(d[e], md_dleaky(pre, main.slope));
      r[e] = md_bn_apply1(gq, raw[e], mu[e], is[e], sc[e], c1[e], c2[e]);
      // the split below must see the ROUNDED fp32 value, as a consumer staging the fp32 tensor would: without this the
      // compiler contracts the final multiply into the (x - hi) subtraction of split8 and lo comes out different
      asm volatile("" : "+v"(r[e]));
    }
    uint4 hi, lo;
    split8(r, hi, lo);
    d_split[((size_t)row * C8 + c8) * 2] = hi;
    d_split[((size_t)row * C8 + c8) * 2 + 1] = lo;
  };
  int64_t row = beg + r0;
  for (; row + (int64_t)nr < end; row += 2 * (int64_t)nr) { one(row); one(row + nr); }
  for (; row < end; row += nr) one(row);
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partial, int blocks, int C, int Cp,
                                                         double inv_count, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ coef) {
  double tot[8];
  reduce_partials4(partial, blocks, Cp, tot);
  const int cl = threadIdx.x, r = threadIdx.x >= 4 ? 1 : 0;
  const int c = blockIdx.x * 4 + (cl & 3);
  double s1[1][4] = {{tot[0], tot[1], tot[2], tot[3]}}, s2[1][4] = {{tot[4], tot[5], tot[6], tot[7]}};
  if (r == 0 && c < Cp) {
    const double g1 = s1[0][cl & 3], g2 = s2[0][cl & 3];
    if (c < C) { if (dbeta) dbeta[c] = (float)g1; if (dgamma) dgamma[c] = (float)g2; }
    coef[c] = (float)(g1 * inv_count);
    coef[Cp + c] = (float)(g2 * inv_count);
  }
}

// ---------------------------------------------------------------- layout conversion, pooling
__global__ __launch_bounds__(256) void k_nchw_to_cl(const float* __restrict__ x, int C, int C4, int64_t thw, int64_t total,
                                                    float* __restrict__ out) {
  // one thread per (pixel, 16-byte chunk); reads are coalesced along the pixel axis of each plane
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t pix = idx / C4; const int c4 = (int)(idx - pix * C4);
  const int64_t b = pix / thw, p = pix - b * thw;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int c = c4 * 4 + e; v[e] = c < C ? x[((size_t)b * C + c) * thw + p] : 0.f; }
  *(float4*)(out + (size_t)idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void k_cl_to_nchw(const float* __restrict__ x, int C, int Cp, int64_t thw, int64_t total,
                                                    float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [B][C][thw]
  if (idx >= total) return;
  const int64_t p = idx % thw; const int64_t bc = idx / thw; const int c = (int)(bc % C); const int64_t b = bc / C;
  out[idx] = x[((size_t)b * thw + p) * Cp + c];
}

__global__ __launch_bounds__(256) void k_avgpool_fwd(const float* __restrict__ x, int C, int C4, int64_t thw, float* __restrict__ feat) {
  __shared__ float4 red[256];
  const int nr = blockDim.x / C4, c4 = threadIdx.x % C4, r = threadIdx.x / C4;
  const float* xb = x + (size_t)blockIdx.x * thw * C4 * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < nr) for (int64_t row = r; row < thw; row += nr) a = add4(a, *(const float4*)(xb + ((size_t)row * C4 + c4) * 4));
  red[threadIdx.x] = a;
  __syncthreads();
  if (r == 0) {
    float4 s = red[c4];
    for (int q = 1; q < nr; ++q) s = add4(s, red[q * C4 + c4]);
    const float inv = 1.f / (float)thw;
    const float v[4] = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
    for (int e = 0; e < 4; ++e) if (c4 * 4 + e < C) feat[(size_t)blockIdx.x * C + c4 * 4 + e] = v[e];
  }
}
__global__ __launch_bounds__(256) void k_avgpool_bwd(const float* __restrict__ dfeat, int C, int C4, int64_t thw, int64_t total,
                                                     float* __restrict__ dx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [B][thw][C4]
  if (idx >= total) return;
  const int c4 = (int)(idx % C4); const int64_t b = idx / C4 / thw;
  const float inv = 1.f / (float)thw;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int c = c4 * 4 + e; v[e] = c < C ? dfeat[(size_t)b * C + c] * inv : 0.f; }
  *(float4*)(dx + (size_t)idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------- host side
extern "C" int md_bn_finalize(const float* stat_partial, int32_t blocks, int32_t C, int64_t count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* mean, float* invstd, float* scale, float* shift, void* stream) {
  if (!stat_partial || !gamma || !beta || !mean || !invstd || !scale || !shift) return MD_ERR_NULL;
  if (C <= 0 || blocks <= 0 || count <= 0) return MD_ERR_BAD_SHAPE;
  const int Cp = md_cpad(C);
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  MD_KLAUNCH(k_bn_finalize, dim3(md_cdiv(Cp, 4)), dim3(256), 0, (hipStream_t)stream, stat_partial, blocks, C, Cp,
                     1.0 / (double)count, unbias, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                     scale, shift);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_bn_eval_params(int32_t C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                 float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !rmean || !rvar || !mean || !invstd || !scale || !shift) return MD_ERR_NULL;
  if (C <= 0) return MD_ERR_BAD_SHAPE;
  const int Cp = md_cpad(C);
  MD_KLAUNCH(k_bn_eval_params, dim3(md_cdiv(Cp, 256)), dim3(256), 0, (hipStream_t)stream, C, Cp, gamma, beta, rmean,
                     rvar, eps, mean, invstd, scale, shift);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

static int check_rows(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return MD_ERR_BAD_SHAPE;
  if (md_cpad(C) / 4 > 256) return MD_ERR_UNSUPPORTED;
  return MD_OK;
}

extern "C" int md_bn_act(const MdActView* x, int64_t rows, int32_t C, float* out, void* stream) {
  if (!x || !x->data || !out) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  MD_KLAUNCH(k_bn_act, dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, to_view(x), rows, C4, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_residual_fwd(const MdActView* skip, const MdActView* main, float alpha, int64_t rows, int32_t C,
                               float* z, void* stream) {
  if (!skip || !main || !skip->data || !main->data || !z) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  MD_KLAUNCH(k_residual_fwd, dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, to_view(skip),
                     to_view(main), alpha, rows, C4, z);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// Geometry of the BatchNorm-backward reduction: up to MD_BN_RED_CAP (2048) workgroups of 256 threads, 16 rows per row lane; past
// the cap, 1024-thread workgroups.  (A cap of 256 -- few partial rows, for md_bn_bwd_apply_fused -- measured 0.06 ms per step
// slower on the R(2+1)D bench, profiles/r03_bn_fused_finalize.txt.)
static int bn_pass_geom(int64_t rows, int C4, int per, int* nth) {
  static const int cap = getenv("MD_BN_RED_CAP") ? atoi(getenv("MD_BN_RED_CAP")) : 2048;
  int64_t b = md_cdiv64(rows, (int64_t)(256 / C4) * per);
  *nth = 256;
  if (b > cap) { *nth = 1024; b = md_cdiv64(rows, (int64_t)(1024 / C4) * per); }
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}
extern "C" int32_t md_bn_bwd_blocks(int64_t rows, int32_t C) {
  if (check_rows(rows, C) != MD_OK) return 0;
  static const int per = getenv("MD_BN_RED_ROWS") ? atoi(getenv("MD_BN_RED_ROWS")) : 16;
  int nth;
  return bn_pass_geom(rows, md_cpad(C) / 4, per, &nth);
}
static const BnFin kNoFin = {nullptr, 0, 0.0, nullptr, nullptr, 0};

extern "C" int md_bn_bwd_reduce(const float* dA, const MdActView* main, const MdActView* skip, float alpha,
                                const float* mean, const float* invstd, int64_t rows, int32_t C, float* partial,
                                void* stream) {
  if (!dA || !main || !main->data || !mean || !invstd || !partial) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  static const int fwd_order = getenv("MD_BN_RED_FWD") ? 2 * (atoi(getenv("MD_BN_RED_FWD")) != 0) : 0;
  static const int per = getenv("MD_BN_RED_ROWS") ? atoi(getenv("MD_BN_RED_ROWS")) : 16;
  int nth; const int nb = bn_pass_geom(rows, C4, per, &nth);
  const int flags = (skip != nullptr ? 1 : 0) | fwd_order;
  if (nth == 1024)
    MD_KLAUNCH((k_bn_bwd<false, false, 1024>), dim3(nb), dim3(1024), 0, (hipStream_t)stream, dA, to_view(main), to_view(skip), flags, alpha,
               mean, invstd, (const float*)nullptr, rows, C4, partial, (float*)nullptr, (float*)nullptr, kNoFin);
  else
    MD_KLAUNCH((k_bn_bwd<false, false, 256>), dim3(nb), dim3(256), 0, (hipStream_t)stream, dA, to_view(main), to_view(skip), flags, alpha,
               mean, invstd, (const float*)nullptr, rows, C4, partial, (float*)nullptr, (float*)nullptr, kNoFin);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_bn_bwd_finalize(const float* partial, int32_t blocks, int32_t C, int64_t count, float* dgamma,
                                  float* dbeta, float* coef, void* stream) {
  if (!partial || !coef) return MD_ERR_NULL;
  if (C <= 0 || blocks <= 0 || count <= 0) return MD_ERR_BAD_SHAPE;
  const int Cp = md_cpad(C);
  MD_KLAUNCH(k_bn_bwd_finalize, dim3(md_cdiv(Cp, 4)), dim3(256), 0, (hipStream_t)stream, partial, blocks, C, Cp,
                     1.0 / (double)count, dgamma, dbeta, coef);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_bn_bwd_apply(const float* dA, const MdActView* main, const MdActView* skip, float alpha,
                               const float* mean, const float* invstd, const float* coef, int64_t rows, int32_t C,
                               float* d_raw, float* dS, void* stream) {
  if (!dA || !main || !main->data || !mean || !invstd || !coef || !d_raw) return MD_ERR_NULL;
  if (skip != nullptr && !dS) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  MD_KLAUNCH(k_bn_bwd<true>, dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, dA, to_view(main),
                     to_view(skip), skip != nullptr ? 1 : 0, alpha, mean, invstd, coef, rows, C4, (float*)nullptr,
                     d_raw, dS, kNoFin);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// Finalize + apply in one launch (fp32 d_raw): the apply pass sums the `blocks` partial rows of the reduction pass -- or of
// a data gradient's fused reduction, g_in != 0 -- itself and writes dgamma / dbeta; no coefficient buffer, no finalize launch.
extern "C" int md_bn_bwd_apply_fused(const float* dA, int g_in, const MdActView* main, const MdActView* skip, float alpha,
                                     const float* mean, const float* invstd, const float* partial, int32_t blocks, int64_t count,
                                     float* dgamma, float* dbeta, int64_t rows, int32_t C, float* d_raw, float* dS, void* stream) {
  if (!dA || !main || !main->data || !mean || !invstd || !partial || !d_raw) return MD_ERR_NULL;
  if (skip != nullptr && !dS) return MD_ERR_NULL;
  if (g_in && (skip != nullptr || !main->scale)) return MD_ERR_UNSUPPORTED;
  if (blocks <= 0 || count <= 0) return MD_ERR_BAD_SHAPE;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  // geometry of the plain apply pass (many 256-thread workgroups: the streaming part measured 14 % slower as 256 workgroups of
  // 1024 threads, profiles/r03k); MD_BN_APPLY_1024=1 keeps that form for experiments
  static const int big = getenv("MD_BN_APPLY_1024") && atoi(getenv("MD_BN_APPLY_1024")) == 1;
  int nth = 256; int nb = stream_blocks(rows, C4);
  if (big) nb = bn_pass_geom(rows, C4, 8, &nth);
  const BnFin fin = {partial, blocks, 1.0 / (double)count, dgamma, dbeta, C};
  const int flags = skip != nullptr ? 1 : 0;
#define LAUNCH_APPLY_FUSED(GIN_, NTH_)                                                                                            \
  MD_KLAUNCH((k_bn_bwd<true, GIN_, NTH_>), dim3(nb), dim3(NTH_), 0, (hipStream_t)stream, dA, to_view(main), to_view(skip), flags, alpha, \
             mean, invstd, (const float*)nullptr, rows, C4, (float*)nullptr, d_raw, dS, fin)
  if (g_in) { if (nth == 1024) LAUNCH_APPLY_FUSED(true, 1024); else LAUNCH_APPLY_FUSED(true, 256); }
  else { if (nth == 1024) LAUNCH_APPLY_FUSED(false, 1024); else LAUNCH_APPLY_FUSED(false, 256); }
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// General form of the apply pass (md_bn_bwd_apply / md_bn_bwd_apply_g are the fp32-output special cases): g_in != 0 means
// dA already holds g; split_out != 0 writes d_raw in the pre-split bf16 format (needs md_cpad(C) % 8 == 0).
extern "C" int md_bn_bwd_apply_fmt(const float* dA, int g_in, const MdActView* main, const MdActView* skip, float alpha,
                                   const float* mean, const float* invstd, const float* coef, int64_t rows, int32_t C,
                                   void* d_raw, int split_out, float* dS, void* stream) {
  if (!dA || !main || !main->data || !mean || !invstd || !coef || !d_raw) return MD_ERR_NULL;
  if (skip != nullptr && !dS) return MD_ERR_NULL;
  if (g_in && skip != nullptr) return MD_ERR_UNSUPPORTED;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int Cp = md_cpad(C), C4 = Cp / 4;
  if (!split_out) {
    if (g_in) {
      if (!main->scale) return MD_ERR_NULL;
      MD_KLAUNCH((k_bn_bwd<true, true>), dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, dA, to_view(main),
                 to_view(nullptr), 0, 1.f, mean, invstd, coef, rows, C4, (float*)nullptr, (float*)d_raw, (float*)nullptr, kNoFin);
    } else {
      MD_KLAUNCH(k_bn_bwd<true>, dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, dA, to_view(main),
                 to_view(skip), skip != nullptr ? 1 : 0, alpha, mean, invstd, coef, rows, C4, (float*)nullptr, (float*)d_raw, dS, kNoFin);
    }
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  if ((Cp & 7) || !main->scale || !main->shift) return MD_ERR_UNSUPPORTED;
  const int C8 = Cp / 8;
  if (C8 > 256) return MD_ERR_UNSUPPORTED;
  const int nr = 256 / C8;
  int64_t blocks = md_cdiv64(rows, (int64_t)nr * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  if (g_in) MD_KLAUNCH((k_bn_bwd_apply_split<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dA, to_view(main),
                       to_view(nullptr), 0, 1.f, mean, invstd, coef, rows, C8, (uint4*)d_raw, (float*)nullptr);
  else MD_KLAUNCH((k_bn_bwd_apply_split<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dA, to_view(main),
                  to_view(skip), skip != nullptr ? 1 : 0, alpha, mean, invstd, coef, rows, C8, (uint4*)d_raw, dS);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// The activation of a unit (leaky(scale * y + shift)), or a materialised tensor as it is (v.scale == nullptr), written in the
// pre-split bf16 format the weight-gradient kernels stage by plain copy: [row][C8 = ceil(Cp / 8) chunks]{hi 8 x bf16 | lo 8 x bf16}.
// Same arithmetic as the in-kernel staging (bn_leaky8: packed fma, max(x, slope x); split8), so a weight gradient reading this
// copy is bit-identical to one that applies BatchNorm-on-read itself.  Cp % 8 == 4: the upper half of the last chunk is zero.
// Runs on the executor's side stream during the forward pass (which leaves that stream idle): the consumer, the unit's weight
// gradient, is on that stream as well, a whole backward pass later.
__global__ __launch_bounds__(256) void k_bn_act_split(View v, int64_t rows, int Cp, int C8, uint4* __restrict__ out) {
  const int nr = blockDim.x / C8, c8 = threadIdx.x % C8, r0 = threadIdx.x / C8;
  if (r0 >= nr) return;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < rows ? beg + per : rows;
  const bool upper = c8 * 8 + 4 < Cp;               // the chunk's channels 4..7 exist
  f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
  if (v.scale) {
    sc0 = *(const f32x4*)(v.scale + c8 * 8); sh0 = *(const f32x4*)(v.shift + c8 * 8);
    if (upper) { sc1 = *(const f32x4*)(v.scale + c8 * 8 + 4); sh1 = *(const f32x4*)(v.shift + c8 * 8 + 4); }
  }
  auto one = [&](int64_t row) {
    const float* s = v.p + (size_t)row * Cp + c8 * 8;
    const float4 a = *(const float4*)s;
    const float4 b = upper ? *(const float4*)(s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float r[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (v.scale) {
      bn_leaky8(r, sc0, sc1, sh0, sh1, v.slope);
      if (!upper) { r[4] = r[5] = r[6] = r[7] = 0.f; }
    }
    uint4 hi, lo;
    split8(r, hi, lo);
    out[((size_t)row * C8 + c8) * 2] = hi;
    out[((size_t)row * C8 + c8) * 2 + 1] = lo;
  };
  int64_t row = beg + r0;
  for (; row + (int64_t)nr < end; row += 2 * (int64_t)nr) { one(row); one(row + nr); }
  for (; row < end; row += nr) one(row);
}
extern "C" size_t md_bn_act_split_floats(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)rows * ((md_cpad(C) + 7) / 8) * 8;
}
extern "C" int md_bn_act_split(const MdActView* x, int64_t rows, int32_t C, void* out, void* stream) {
  if (!x || !x->data || !out) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int Cp = md_cpad(C), C8 = (Cp + 7) / 8;
  if (C8 > 256) return MD_ERR_UNSUPPORTED;
  const int nr = 256 / C8;
  int64_t blocks = md_cdiv64(rows, (int64_t)nr * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  MD_KLAUNCH(k_bn_act_split, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, to_view(x), rows, Cp, C8, (uint4*)out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_bn_bwd_apply_g(const float* g, const MdActView* main, const float* mean, const float* invstd,
                                 const float* coef, int64_t rows, int32_t C, float* d_raw, void* stream) {
  if (!g || !main || !main->data || !main->scale || !mean || !invstd || !coef || !d_raw) return MD_ERR_NULL;
  int rc = check_rows(rows, C); if (rc) return rc;
  const int C4 = md_cpad(C) / 4;
  MD_KLAUNCH((k_bn_bwd<true, true>), dim3(stream_blocks(rows, C4)), dim3(256), 0, (hipStream_t)stream, g, to_view(main),
             to_view(nullptr), 0, 1.f, mean, invstd, coef, rows, C4, (float*)nullptr, d_raw, (float*)nullptr, kNoFin);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_nchw_to_cl(const float* x, int32_t B, int32_t C, int64_t thw, float* out, void* stream) {
  if (!x || !out) return MD_ERR_NULL;
  if (B <= 0 || C <= 0 || thw <= 0) return MD_ERR_BAD_SHAPE;
  const int C4 = md_cpad(C) / 4;
  const int64_t total = (int64_t)B * thw * C4;
  MD_KLAUNCH(k_nchw_to_cl, dim3((unsigned)md_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, C, C4, thw,
                     total, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_cl_to_nchw(const float* x, int32_t B, int32_t C, int64_t thw, float* out, void* stream) {
  if (!x || !out) return MD_ERR_NULL;
  if (B <= 0 || C <= 0 || thw <= 0) return MD_ERR_BAD_SHAPE;
  const int64_t total = (int64_t)B * C * thw;
  MD_KLAUNCH(k_cl_to_nchw, dim3((unsigned)md_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, C, md_cpad(C),
                     thw, total, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// Channel concatenation of two channels-last tensors (SlowFast's lateral connection, slowfast.py:26-40: torch.cat([slow, lateral],
// dim=1)): out[row][0..Ca) = a[row][0..Ca), out[row][Ca..Ca+Cb) = b[row][0..Cb), padding channels zero.  One thread per output
// channel quad of a row; the backward splits the gradient the same way (padding channels of da / db zero).
__global__ __launch_bounds__(256) void k_cat_cl(const float* __restrict__ a, int Ca, int Cpa, const float* __restrict__ b, int Cb,
                                                int Cpb, int Cpo, int64_t total, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int q = Cpo >> 2;
  const int64_t row = i / q; const int c0 = (int)(i - row * q) * 4;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    v[e] = c < Ca ? a[row * Cpa + c] : (c < Ca + Cb ? b[row * Cpb + (c - Ca)] : 0.f);
  }
  *(float4*)(out + row * Cpo + c0) = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void k_split_cl(const float* __restrict__ g, int Cpo, int Ca, int Cpa, int Cb, int Cpb, int64_t total,
                                                  float* __restrict__ da, float* __restrict__ db) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int qa = Cpa >> 2, q = qa + (Cpb >> 2);
  const int64_t row = i / q; const int k = (int)(i - row * q);
  const bool isa = k < qa;
  const int c0 = (isa ? k : k - qa) * 4, n = isa ? Ca : Cb, off = isa ? 0 : Ca;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = c0 + e < n ? g[row * Cpo + off + c0 + e] : 0.f;
  *(float4*)((isa ? da + row * Cpa : db + row * Cpb) + c0) = make_float4(v[0], v[1], v[2], v[3]);
}
extern "C" int md_cat_cl(const float* a, int32_t Ca, const float* b, int32_t Cb, int64_t rows, float* out, void* stream) {
  if (!a || !b || !out) return MD_ERR_NULL;
  if (Ca <= 0 || Cb <= 0 || rows <= 0) return MD_ERR_BAD_SHAPE;
  const int Cpo = md_cpad(Ca + Cb);
  const int64_t total = rows * (Cpo / 4);
  MD_KLAUNCH(k_cat_cl, dim3((unsigned)md_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, a, Ca, md_cpad(Ca), b, Cb, md_cpad(Cb),
             Cpo, total, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_split_cl(const float* g, int32_t Ca, int32_t Cb, int64_t rows, float* da, float* db, void* stream) {
  if (!g || !da || !db) return MD_ERR_NULL;
  if (Ca <= 0 || Cb <= 0 || rows <= 0) return MD_ERR_BAD_SHAPE;
  const int Cpa = md_cpad(Ca), Cpb = md_cpad(Cb);
  const int64_t total = rows * ((Cpa + Cpb) / 4);
  MD_KLAUNCH(k_split_cl, dim3((unsigned)md_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, g, md_cpad(Ca + Cb), Ca, Cpa, Cb, Cpb,
             total, da, db);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_avgpool_fwd(const float* x, int32_t B, int32_t C, int64_t thw, float* feat, void* stream) {
  if (!x || !feat) return MD_ERR_NULL;
  int rc = check_rows(thw, C); if (rc) return rc;
  if (B <= 0) return MD_ERR_BAD_SHAPE;
  MD_KLAUNCH(k_avgpool_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, x, C, md_cpad(C) / 4, thw, feat);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_avgpool_bwd(const float* dfeat, int32_t B, int32_t C, int64_t thw, float* dx, void* stream) {
  if (!dfeat || !dx) return MD_ERR_NULL;
  if (B <= 0 || C <= 0 || thw <= 0) return MD_ERR_BAD_SHAPE;
  const int C4 = md_cpad(C) / 4;
  const int64_t total = (int64_t)B * thw * C4;
  MD_KLAUNCH(k_avgpool_bwd, dim3((unsigned)md_cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, dfeat, C, C4, thw,
                     total, dx);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
