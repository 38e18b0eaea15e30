// Implicit-GEMM convolution for gfx950 (MI355X): forward, data gradient and weight gradient of the
// factored (1,k,k) / (k,1,1) Conv3d of the reference's Conv3dBlock (src/models/R2Plus1D.py:25-58).
//
// Data layout: channels-last activations [N][T][H][W][Cp] (Cp multiple of 4 -> every pixel is a run of
// 16-byte chunks).  The GEMM's K axis is the flat (tap, channel) axis cut into 16-byte chunks, so a
// 32-deep K stage may straddle taps and no MFMA work is wasted on per-tap channel padding.
//
// Matrix core mapping (exact fp32, v_mfma_f32_16x16x4_f32): a wave owns a 32 x (16*nrep) output tile.
// Lane (i = lane&15, g = lane>>4) reads ONE 16-byte chunk per operand row per 16-deep K step
// (ds_read_b128) and feeds its 4 floats to 4 consecutive MFMAs; lane group g therefore supplies
// k = 16*s + 4*g + e to MFMA e, identically for A and B, which is a permutation of the K axis and
// leaves the product unchanged.  LDS row pitch 40 floats makes those b128 reads bank-conflict free.
#include <cstdlib>
#include "common.h"
#include <vector>
#include "patch_common.h"

#define BM 128          // destination rows per workgroup
#define KB 32           // K depth per LDS stage (floats)
#define PA 40           // LDS row pitch (floats): (PA/4) % 4 == 2 -> conflict-free ds_read_b128
#define NREP_MAX 9      // up to 144 destination channels per workgroup
#define MAXC_PROLOGUE 320

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_gemm(
    Geom g, const float* __restrict__ src, const float* __restrict__ pscale,
    const float* __restrict__ pshift, float pslope, const float* __restrict__ wp,
    float* __restrict__ dst, float* __restrict__ stat_partial, int accumulate, int n_per_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                                  // [BM][PA]
  float* sB = sA + BM * PA;                          // [NREP_MAX*16][PA]
  int4* sRow = (int4*)(sB + NREP_MAX * 16 * PA);     // [BM]
  float* sScale = (float*)(sRow + BM);               // [MAXC_PROLOGUE] scale, then shift
  float* sShift = sScale + MAXC_PROLOGUE;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * n_per_blk;
  const int ncols = min(n_per_blk, g.N16 - n0);
  const int nrep = ncols >> 4;
  const bool prologue = pscale != nullptr;

  if (t < BM) {
    int m = m0 + t;
    int4 ri;
    if (m < g.M) {
      int ox = m % g.Wo; int r = m / g.Wo;
      int oy = r % g.Ho; r /= g.Ho;
      int ot = r % g.To; int n = r / g.To;
      ri.x = n * g.Ti * g.Hi * g.Wi;
      ri.y = ot * g.sn_t + g.off_t;
      ri.z = oy * g.sn_h + g.off_h;
      ri.w = ox * g.sn_w + g.off_w;
    } else {
      ri.x = 0; ri.y = -(1 << 28); ri.z = 0; ri.w = 0;
    }
    sRow[t] = ri;
  }
  if (prologue) {
    for (int c = t; c < g.Cpi; c += 256) { sScale[c] = pscale[c]; sShift[c] = pshift[c]; }
  }
  __syncthreads();

  const int cj = t & 7, rg = t >> 3;
  const int C4 = g.Cpi >> 2;
  float4 ra[4], rb[5];

  auto load_stage = [&](int kb) {
    const int q = kb * 8 + cj;
    const bool qv = q < g.Kc;
    const int tap = q / C4;
    const int c4 = q - tap * C4;
    int dt = tap / g.khw;
    const int r2 = tap - dt * g.khw;
    int dy = r2 / g.kw;
    int dx = r2 - dy * g.kw;
    dt *= g.sign; dy *= g.sign; dx *= g.sign;
    float4 sc, sh;
    if (prologue && qv) { sc = *(const float4*)(sScale + c4 * 4); sh = *(const float4*)(sShift + c4 * 4); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int4 ri = sRow[rg + 32 * i];
      const int nt = ri.y + dt, nh = ri.z + dy, nw = ri.w + dx;
      int it = nt, ih = nh, iw = nw;
      bool v = qv;
      if (g.sd_t != 1) { it = nt / g.sd_t; v = v && (it * g.sd_t == nt); }
      if (g.sd_h != 1) { ih = nh / g.sd_h; v = v && (ih * g.sd_h == nh); }
      if (g.sd_w != 1) { iw = nw / g.sd_w; v = v && (iw * g.sd_w == nw); }
      v = v && ((unsigned)it < (unsigned)g.Ti) && ((unsigned)ih < (unsigned)g.Hi) && ((unsigned)iw < (unsigned)g.Wi);
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v) {
        const size_t pix = (size_t)(ri.x + (it * g.Hi + ih) * g.Wi + iw);
        val = *(const float4*)(src + pix * g.Cpi + c4 * 4);
        if (prologue) {
          val.x = md_leaky(fmaf(val.x, sc.x, sh.x), pslope);
          val.y = md_leaky(fmaf(val.y, sc.y, sh.y), pslope);
          val.z = md_leaky(fmaf(val.z, sc.z, sh.z), pslope);
          val.w = md_leaky(fmaf(val.w, sc.w, sh.w), pslope);
        }
      }
      ra[i] = val;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int n = rg + 32 * i;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < ncols) val = *(const float4*)(wp + (size_t)(n0 + n) * g.Kp + kb * KB + cj * 4);
      rb[i] = val;
    }
  };

  f32x4 acc[2][NREP_MAX];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < NREP_MAX; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_stage(0);
  for (int kb = 0; kb < g.nstages; ++kb) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float4*)(sA + (rg + 32 * i) * PA + cj * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int n = rg + 32 * i;
      if (n < NREP_MAX * 16) *(float4*)(sB + n * PA + cj * 4) = rb[i];
    }
    __syncthreads();
    if (kb + 1 < g.nstages) load_stage(kb + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 a0 = *(const f32x4*)(sA + (wave * 32 + li) * PA + (s * 4 + lg) * 4);
      const f32x4 a1 = *(const f32x4*)(sA + (wave * 32 + 16 + li) * PA + (s * 4 + lg) * 4);
#pragma unroll
      for (int j = 0; j < NREP_MAX; ++j) {
        if (j < nrep) {
          const f32x4 b = *(const f32x4*)(sB + (j * 16 + li) * PA + (s * 4 + lg) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[e], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[e], acc[1][j], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: store (C/D map: col = lane&15, row = 4*(lane>>4) + reg) ----
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int j = 0; j < NREP_MAX; ++j) {
      if (j < nrep) {
        const int col = n0 + j * 16 + li;
        if (col < g.Cpo) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 32 + a * 16 + lg * 4 + r;
            if (m < g.M) {
              float* p = dst + (size_t)m * g.Cpo + col;
              float v = acc[a][j][r];
              if (accumulate) v += *p;
              *p = v;
            }
          }
        }
      }
    }
  }

  // ---- epilogue: BatchNorm partial statistics of the raw output (rows >= M contribute exact zeros) ----
  if (stat_partial != nullptr) {
    __syncthreads();
    float* red = sA;  // [4 waves][2][NREP_MAX*16]
#pragma unroll
    for (int j = 0; j < NREP_MAX; ++j) {
      if (j < nrep) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[a][j][r]; s1 += v; s2 = fmaf(v, v, s2); }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lg == 0) {
          red[(wave * 2 + 0) * (NREP_MAX * 16) + j * 16 + li] = s1;
          red[(wave * 2 + 1) * (NREP_MAX * 16) + j * 16 + li] = s2;
        }
      }
    }
    __syncthreads();
    if (t < ncols && n0 + t < g.Cpo) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s1 += red[(w * 2 + 0) * (NREP_MAX * 16) + t];
        s2 += red[(w * 2 + 1) * (NREP_MAX * 16) + t];
      }
      float* sp = stat_partial + (size_t)blockIdx.x * 2 * g.Cpo;
      sp[n0 + t] = s1;
      sp[g.Cpo + n0 + t] = s2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel: dw[cout][cin][tap] += sum_m Xg[m][(tap,cin)] * dY[m][cout]
// Workgroup = 64 flat-k values x up to 144 output channels x a slice of the pixels; wave w owns k rows
// 16w..16w+15.  Operands are read from row-major [pixel][.] LDS tiles with ds_read_b32 (lanes of a
// 16-lane group on consecutive floats, the two groups of a half-wave 16 banks apart: conflict free).
// ------------------------------------------------------------------------------------------------
#define WK 64
#define WR 32
#define PX 80    // 64 + 16
#define PY 176   // 144 + 32, == 16 (mod 32)

__global__ __launch_bounds__(256) void k_conv_wgrad(
    Geom g, const float* __restrict__ src, const float* __restrict__ pscale,
    const float* __restrict__ pshift, float pslope, const float* __restrict__ dy,
    float* __restrict__ dw, int Cin, int Cout, int taps, int rows_per_blk, int n_per_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                 // [WR][PX]
  float* sY = sX + WR * PX;         // [WR][PY]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int ktile = blockIdx.x;
  const int mbeg = blockIdx.y * rows_per_blk;
  const int mend = min(g.M, mbeg + rows_per_blk);
  const int n0 = blockIdx.z * n_per_blk;
  const int N16 = (Cout + 15) & ~15;
  const int ncols = min(n_per_blk, N16 - n0);
  const int nrep = ncols >> 4;
  const int nch = min(ncols, g.Cpo - n0) >> 2;   // valid 16-byte chunks of a dY row in this column block
  const bool prologue = pscale != nullptr;

  // this thread's fixed flat-k chunk
  const int cj = t & 15, rg = t >> 4;   // rows rg, rg+16
  const int q = ktile * 16 + cj;
  const bool qv = q < g.Kc;
  const int C4 = g.Cpi >> 2;
  const int tap = q / C4;
  const int c4 = q - tap * C4;
  const int dt = tap / g.khw;
  const int r2 = tap - dt * g.khw;
  const int dyy = r2 / g.kw;
  const int dxx = r2 - dyy * g.kw;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (prologue && qv) { sc = *(const float4*)(pscale + c4 * 4); sh = *(const float4*)(pshift + c4 * 4); }

  float4 rx[2], ry[5];
  auto load_stage = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + rg + 16 * i;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qv && m < mend) {
        int ox = m % g.Wo; int r = m / g.Wo;
        int oy = r % g.Ho; r /= g.Ho;
        int ot = r % g.To; int n = r / g.To;
        const int it = ot * g.sn_t + g.off_t + dt;
        const int ih = oy * g.sn_h + g.off_h + dyy;
        const int iw = ox * g.sn_w + g.off_w + dxx;
        if (((unsigned)it < (unsigned)g.Ti) && ((unsigned)ih < (unsigned)g.Hi) && ((unsigned)iw < (unsigned)g.Wi)) {
          const size_t pix = (size_t)((n * g.Ti + it) * g.Hi + ih) * g.Wi + iw;
          val = *(const float4*)(src + pix * g.Cpi + c4 * 4);
          if (prologue) {
            val.x = md_leaky(fmaf(val.x, sc.x, sh.x), pslope);
            val.y = md_leaky(fmaf(val.y, sc.y, sh.y), pslope);
            val.z = md_leaky(fmaf(val.z, sc.z, sh.z), pslope);
            val.w = md_leaky(fmaf(val.w, sc.w, sh.w), pslope);
          }
        }
      }
      rx[i] = val;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = t + 256 * i;            // element over [WR][36 chunks]
      const int row = e / 36, ch = e - row * 36;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      const int m = mb + row;
      if (row < WR && ch < nch && m < mend) val = *(const float4*)(dy + (size_t)m * g.Cpo + n0 + ch * 4);
      ry[i] = val;
    }
  };

  f32x4 acc[NREP_MAX];
#pragma unroll
  for (int j = 0; j < NREP_MAX; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (mbeg < mend) load_stage(mbeg);
  for (int mb = mbeg; mb < mend; mb += WR) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) *(float4*)(sX + (rg + 16 * i) * PX + cj * 4) = rx[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = t + 256 * i;
      const int row = e / 36, ch = e - row * 36;
      if (row < WR) *(float4*)(sY + row * PY + ch * 4) = ry[i];
    }
    __syncthreads();
    if (mb + WR < mend) load_stage(mb + WR);
#pragma unroll
    for (int s = 0; s < WR / 4; ++s) {
      const float a = sX[(s * 4 + lg) * PX + wave * 16 + li];
#pragma unroll
      for (int j = 0; j < NREP_MAX; ++j) {
        if (j < nrep) {
          const float b = sY[(s * 4 + lg) * PY + j * 16 + li];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: D[row = k index][col = cout] -> reference layout (Cout, Cin, taps), float atomics
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = ktile * WK + wave * 16 + lg * 4 + r;
    const int ktap = k / g.Cpi;
    const int c = k - ktap * g.Cpi;
    if (k < g.Kc * 4 && c < Cin) {
#pragma unroll
      for (int j = 0; j < NREP_MAX; ++j) {
        if (j < nrep) {
          const int co = n0 + j * 16 + li;
          if (co < Cout) atomicAdd(dw + ((size_t)co * Cin + c) * taps + ktap, acc[j][r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: reference layout (Cout,Cin,taps) -> GEMM operands [n][flat k], zero padded.
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_weights(const float* __restrict__ w, int Cout, int Cin, int taps,
                               float* __restrict__ wf, int Cpi, int KpF, int N16F,
                               float* __restrict__ wd, int Cpo, int KpD, int N16D) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nF = N16F * KpF;
  if (wf != nullptr && idx < nF) {
    const int n = idx / KpF, k = idx - n * KpF;
    const int tap = k / Cpi, c = k - tap * Cpi;
    float v = 0.f;
    if (n < Cout && tap < taps && c < Cin) v = w[((size_t)n * Cin + c) * taps + tap];
    wf[idx] = v;
  }
  const int nD = N16D * KpD;
  if (wd != nullptr && idx < nD) {
    const int n = idx / KpD, k = idx - n * KpD;   // n = cin, k = tap*Cpo + cout
    const int tap = k / Cpo, co = k - tap * Cpo;
    float v = 0.f;
    if (n < Cin && tap < taps && co < Cout) v = w[((size_t)co * Cin + n) * taps + tap];
    wd[idx] = v;
  }
}

// the same for up to GPACK_MAX weights in one launch (blockIdx.y = item): the Linears of the transformer models whose operands are
// not in the patch format
#define GPACK_MAX 32
struct GPackItem { const float* w; float* wf; float* wd; int Cout, Cin, taps, CpiF, KpF, N16F, CpoD, KpD, N16D; };
struct GPackBatch { GPackItem it[GPACK_MAX]; };
__global__ void k_pack_weights_many(GPackBatch pb) {
  const GPackItem& q = pb.it[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nF = q.N16F * q.KpF;
  if (q.wf != nullptr && idx < nF) {
    const int n = idx / q.KpF, k = idx - n * q.KpF;
    const int tap = k / q.CpiF, c = k - tap * q.CpiF;
    float v = 0.f;
    if (n < q.Cout && tap < q.taps && c < q.Cin) v = q.w[((size_t)n * q.Cin + c) * q.taps + tap];
    q.wf[idx] = v;
  }
  const int nD = q.N16D * q.KpD;
  if (q.wd != nullptr && idx < nD) {
    const int n = idx / q.KpD, k = idx - n * q.KpD;
    const int tap = k / q.CpoD, co = k - tap * q.CpoD;
    float v = 0.f;
    if (n < q.Cin && tap < q.taps && co < q.Cout) v = q.w[((size_t)co * q.Cin + n) * q.taps + tap];
    q.wd[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_desc(const MdConvDesc* d) {
  if (!d) return MD_ERR_NULL;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->kt <= 0 || d->kh <= 0 || d->kw <= 0) return MD_ERR_BAD_SHAPE;
  if (d->st <= 0 || d->sh <= 0 || d->sw <= 0) return MD_ERR_BAD_SHAPE;
  if (d->To != (d->Ti + 2 * d->pt - d->kt) / d->st + 1) return MD_ERR_BAD_SHAPE;
  if (d->Ho != (d->Hi + 2 * d->ph - d->kh) / d->sh + 1) return MD_ERR_BAD_SHAPE;
  if (d->Wo != (d->Wi + 2 * d->pw - d->kw) / d->sw + 1) return MD_ERR_BAD_SHAPE;
  if (d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return MD_ERR_BAD_SHAPE;
  if ((int64_t)d->N * d->Ti * d->Hi * d->Wi >= (1ll << 31) / 4) return MD_ERR_UNSUPPORTED;
  if ((int64_t)d->N * d->To * d->Ho * d->Wo >= (1ll << 31) / 4) return MD_ERR_UNSUPPORTED;
  return MD_OK;
}

static Geom geom_fwd(const MdConvDesc* d) {
  Geom g;
  g.Ti = d->Ti; g.Hi = d->Hi; g.Wi = d->Wi; g.Cpi = md_cpad(d->Cin);
  g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.Cpo = md_cpad(d->Cout);
  g.kh = d->kh; g.kw = d->kw; g.khw = d->kh * d->kw;
  g.sn_t = d->st; g.sn_h = d->sh; g.sn_w = d->sw;
  g.sd_t = g.sd_h = g.sd_w = 1;
  g.off_t = -d->pt; g.off_h = -d->ph; g.off_w = -d->pw;
  g.sign = 1;
  g.M = d->N * d->To * d->Ho * d->Wo;
  const int K = d->kt * d->kh * d->kw * g.Cpi;
  g.Kc = K / 4;
  g.Kp = md_round_up(K, KB);
  g.nstages = g.Kp / KB;
  g.N16 = md_round_up(d->Cout, 16);
  return g;
}

static Geom geom_dgrad(const MdConvDesc* d) {
  Geom g;
  g.Ti = d->To; g.Hi = d->Ho; g.Wi = d->Wo; g.Cpi = md_cpad(d->Cout);   // source = dY
  g.To = d->Ti; g.Ho = d->Hi; g.Wo = d->Wi; g.Cpo = md_cpad(d->Cin);    // destination = dX
  g.kh = d->kh; g.kw = d->kw; g.khw = d->kh * d->kw;
  g.sn_t = g.sn_h = g.sn_w = 1;
  g.sd_t = d->st; g.sd_h = d->sh; g.sd_w = d->sw;
  g.off_t = d->pt; g.off_h = d->ph; g.off_w = d->pw;
  g.sign = -1;
  g.M = d->N * d->Ti * d->Hi * d->Wi;
  const int K = d->kt * d->kh * d->kw * g.Cpi;
  g.Kc = K / 4;
  g.Kp = md_round_up(K, KB);
  g.nstages = g.Kp / KB;
  g.N16 = md_round_up(d->Cin, 16);
  return g;
}

static int pick_n_per_blk(int N16) {
  // split the destination channels into equal multiples of 16, each <= 144
  const int nchunks = md_cdiv(N16, NREP_MAX * 16);
  return md_round_up(md_cdiv(N16, nchunks), 16);
}

static size_t conv_gemm_lds_bytes() {
  return (size_t)(BM * PA + NREP_MAX * 16 * PA) * 4 + BM * sizeof(int4) + 2 * MAXC_PROLOGUE * 4;
}

extern "C" size_t md_conv_wpack_fwd_floats(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  if (const PatchPlan* pp = patch_lookup(d, 0)) return patch_wpack_floats(pp);
  Geom g = geom_fwd(d);
  return (size_t)g.N16 * g.Kp;
}
extern "C" size_t md_conv_wpack_dgrad_floats(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  if (const PatchPlan* pp = patch_lookup(d, 1)) return patch_wpack_floats(pp);
  Geom g = geom_dgrad(d);
  return (size_t)g.N16 * g.Kp;
}

extern "C" int md_conv_pack_weights(const MdConvDesc* d, const float* w, float* wf, float* wd, void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!w) return MD_ERR_NULL;
  Geom gf = geom_fwd(d), gd = geom_dgrad(d);
  const int taps = d->kt * d->kh * d->kw;
  // unit-stride geometries use the split-bf16 patch kernel and its own operand layout
  if (wf) if (const PatchPlan* pp = patch_lookup(d, 0)) { rc = patch_pack(d, 0, pp, w, wf, (hipStream_t)stream); if (rc) return rc; wf = nullptr; }
  if (wd) if (const PatchPlan* pp = patch_lookup(d, 1)) { rc = patch_pack(d, 1, pp, w, wd, (hipStream_t)stream); if (rc) return rc; wd = nullptr; }
  const int nF = wf ? gf.N16 * gf.Kp : 0, nD = wd ? gd.N16 * gd.Kp : 0;
  const int n = nF > nD ? nF : nD;
  if (n == 0) return MD_OK;
  MD_KLAUNCH(k_pack_weights, dim3(md_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, d->Cout, d->Cin,
                     taps, wf, gf.Cpi, gf.Kp, gf.N16, wd, gd.Cpi, gd.Kp, gd.N16);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// Many units at once (the composable models pack ~120 operands per training step, one tiny launch each): the patch-format
// operands of all of them go through the batched kernel (<= 64 per launch), the rest one by one.  wf[i] / wd[i] may be NULL.
extern "C" int md_conv_pack_weights_batch(int32_t n, const MdConvDesc* descs, const float* const* w, float* const* wf, float* const* wd,
                                          void* stream) {
  if (n < 0 || (n > 0 && (!descs || !w || !wf || !wd))) return MD_ERR_NULL;
  for (int i = 0; i < n; ++i) { int rc = check_desc(&descs[i]); if (rc != MD_OK) return rc; if (!w[i]) return MD_ERR_NULL; }
  std::vector<const MdConvDesc*> dp(2 * n); std::vector<int> dg(2 * n);
  std::vector<const float*> wsrc(2 * n); std::vector<float*> outs(2 * n); std::vector<unsigned char> handled(2 * n);
  for (int i = 0; i < n; ++i) {
    dp[2 * i] = &descs[i]; dg[2 * i] = 0; wsrc[2 * i] = w[i]; outs[2 * i] = wf[i];
    dp[2 * i + 1] = &descs[i]; dg[2 * i + 1] = 1; wsrc[2 * i + 1] = w[i]; outs[2 * i + 1] = wd[i];
  }
  int rc = patch_pack_batch(2 * n, dp.data(), dg.data(), wsrc.data(), outs.data(), handled.data(), (hipStream_t)stream);
  if (rc) return rc;
  GPackBatch gb; int cnt = 0, maxtot = 0;
  auto flush = [&]() -> int {
    if (!cnt) return MD_OK;
    MD_KLAUNCH(k_pack_weights_many, dim3(md_cdiv(maxtot, 256), cnt), dim3(256), 0, (hipStream_t)stream, gb);
    MD_CHECK_LAUNCH();
    cnt = 0; maxtot = 0;
    return MD_OK;
  };
  for (int i = 0; i < n; ++i) {
    float* f = handled[2 * i] ? nullptr : wf[i];
    float* d = handled[2 * i + 1] ? nullptr : wd[i];
    if (!f && !d) continue;
    const Geom gf = geom_fwd(&descs[i]), gd = geom_dgrad(&descs[i]);
    GPackItem& q = gb.it[cnt++];
    q.w = w[i]; q.wf = f; q.wd = d; q.Cout = descs[i].Cout; q.Cin = descs[i].Cin; q.taps = descs[i].kt * descs[i].kh * descs[i].kw;
    q.CpiF = gf.Cpi; q.KpF = gf.Kp; q.N16F = gf.N16; q.CpoD = gd.Cpi; q.KpD = gd.Kp; q.N16D = gd.N16;
    const int nF = f ? gf.N16 * gf.Kp : 0, nD = d ? gd.N16 * gd.Kp : 0;
    if (nF > maxtot) maxtot = nF;
    if (nD > maxtot) maxtot = nD;
    if (cnt == GPACK_MAX) { rc = flush(); if (rc) return rc; }
  }
  return flush();
}

extern "C" int32_t md_conv_fwd_stat_blocks(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  if (const PatchPlan* pp = patch_lookup(d, 0)) return patch_blocks(pp);
  return md_cdiv(d->N * d->To * d->Ho * d->Wo, BM);
}

static bool is_pure_gemm(const MdConvDesc* d) {       // a Linear over rows: 1x1x1, unit stride, no padding
  return d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0;
}

static int launch_gemm(const Geom& g, const float* src, const float* ps, const float* psh, float slope,
                       const float* wp, float* dst, float* stat, int accumulate, hipStream_t s) {
  if (ps != nullptr && g.Cpi > MAXC_PROLOGUE) return MD_ERR_UNSUPPORTED;
  int npb = pick_n_per_blk(g.N16);
  // 128-row tiles alone may not fill the chip (a Linear over 16 548 tokens is 130 of them): split the destination channels
  // further until there are about two workgroups per CU (the source tile is then read once per split; measured: ViViT cfg3 step 6.27 -> 5.91 ms)
  static const int fill = getenv("MD_GEMM_FILL") ? atoi(getenv("MD_GEMM_FILL")) : 512;
  while (md_cdiv(g.M, BM) * md_cdiv(g.N16, npb) < fill && npb > 32) npb = md_round_up(npb / 2, 16);
  dim3 grid(md_cdiv(g.M, BM), md_cdiv(g.N16, npb));
  static bool attr_set = false;
  const size_t lds = conv_gemm_lds_bytes();
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_conv_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  MD_KLAUNCH(k_conv_gemm, grid, dim3(256), lds, s, g, src, ps, psh, slope, wp, dst, stat, accumulate, npb);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_conv_fwd(const MdConvDesc* d, const MdActView* x, const float* wpack_fwd, float* y_raw,
                           float* stat_partial, void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!x || !x->data || !wpack_fwd || !y_raw) return MD_ERR_NULL;
  if (const PatchPlan* pp = patch_lookup(d, 0))
    return patch_launch(pp, x->data, x->scale, x->shift, x->slope, wpack_fwd, y_raw, stat_partial, 0, (hipStream_t)stream);
  Geom g = geom_fwd(d);
  if (is_pure_gemm(d) && !x->scale && !stat_partial &&
      linear_split_launch(1, x->data, g.M, g.Kc * 4, wpack_fwd, g.Kp, g.N16, y_raw, g.Cpo, 0, (hipStream_t)stream) == MD_OK)
    return MD_OK;
  return launch_gemm(g, x->data, x->scale, x->shift, x->slope, wpack_fwd, y_raw, stat_partial, 0, (hipStream_t)stream);
}

extern "C" int md_conv_dgrad(const MdConvDesc* d, const float* dy_raw, const float* wpack_dgrad, float* dx,
                             int accumulate, void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!dy_raw || !wpack_dgrad || !dx) return MD_ERR_NULL;
  if (const PatchPlan* pp = patch_lookup(d, 1))
    return patch_launch(pp, dy_raw, nullptr, nullptr, 1.f, wpack_dgrad, dx, nullptr, accumulate, (hipStream_t)stream);
  Geom g = geom_dgrad(d);
  if (is_pure_gemm(d) &&
      linear_split_launch(0, dy_raw, g.M, g.Kc * 4, wpack_dgrad, g.Kp, g.N16, dx, g.Cpo, accumulate, (hipStream_t)stream) == MD_OK)
    return MD_OK;
  return launch_gemm(g, dy_raw, nullptr, nullptr, 1.f, wpack_dgrad, dx, nullptr, accumulate, (hipStream_t)stream);
}

extern "C" int32_t md_conv_dgrad_bnred_blocks(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  const PatchPlan* pp = patch_lookup(d, 1);
  return (pp && patch_can_fuse(pp)) ? patch_fused_blocks(pp) : 0;
}
extern "C" int md_conv_dgrad_bnred(const MdConvDesc* d, const float* dy_raw, const float* wpack_dgrad, float* g_out,
                                   int accumulate, const MdActView* y_view, const float* mean, const float* invstd,
                                   float* partial, void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!dy_raw || !wpack_dgrad || !g_out || !y_view || !y_view->data || !y_view->scale || !y_view->shift || !mean || !invstd || !partial)
    return MD_ERR_NULL;
  const PatchPlan* pp = patch_lookup(d, 1);
  if (!pp || !patch_can_fuse(pp)) return MD_ERR_UNSUPPORTED;
  PersBwd bw; bw.yraw = y_view->data; bw.scale = y_view->scale; bw.shift = y_view->shift; bw.mean = mean; bw.invstd = invstd;
  bw.slope = y_view->slope;
  return patch_launch(pp, dy_raw, nullptr, nullptr, 1.f, wpack_dgrad, g_out, partial, accumulate, (hipStream_t)stream, &bw);
}

// ---- gradients in the pre-split bf16 format (written by md_bn_bwd_apply_fmt with split_out): accepted when every kernel
// that reads this unit's d_raw is a patch kernel and the channel pitch is a whole number of 8-channel chunks
extern "C" int md_conv_split_dy_ok(const MdConvDesc* d, int need_dgrad) {
  if (check_desc(d) != MD_OK) return 0;
  if (md_cpad(d->Cout) & 7) return 0;
  // Measured on the BASELINE step (profiles/r02k_*): the consumers gain 2-3 % (their staging is not what bounds them)
  // while the split-writing apply pass is 0.3 ms slower than the fp32 one, so the executor keeps fp32 gradients unless
  // MD_SPLIT_DY=1; the format stays available (and bit-exactly tested) for callers that re-read a gradient many times.
  static const int on = getenv("MD_SPLIT_DY") && atoi(getenv("MD_SPLIT_DY")) == 1;
  if (!on) return 0;
  if (!wgrad_lookup(d)) return 0;
  if (need_dgrad && !patch_lookup(d, 1)) return 0;
  return 1;
}
extern "C" int md_conv_dgrad_fmt(const MdConvDesc* d, const void* dy, int dy_split, const float* wpack_dgrad, float* dx,
                                 int accumulate, const MdActView* y_view, const float* mean, const float* invstd,
                                 float* partial, void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!dy || !wpack_dgrad || !dx) return MD_ERR_NULL;
  if (!dy_split && !y_view) return md_conv_dgrad(d, (const float*)dy, wpack_dgrad, dx, accumulate, stream);
  if (!dy_split) return md_conv_dgrad_bnred(d, (const float*)dy, wpack_dgrad, dx, accumulate, y_view, mean, invstd, partial, stream);
  const PatchPlan* pp = patch_lookup(d, 1);
  if (!pp || (md_cpad(d->Cout) & 7)) return MD_ERR_UNSUPPORTED;
  const int acc = (accumulate ? 1 : 0) | 0x10000;
  if (!y_view) return patch_launch(pp, (const float*)dy, nullptr, nullptr, 1.f, wpack_dgrad, dx, nullptr, acc, (hipStream_t)stream);
  if (!y_view->data || !y_view->scale || !y_view->shift || !mean || !invstd || !partial) return MD_ERR_NULL;
  if (!patch_can_fuse(pp)) return MD_ERR_UNSUPPORTED;
  PersBwd bw; bw.yraw = y_view->data; bw.scale = y_view->scale; bw.shift = y_view->shift; bw.mean = mean; bw.invstd = invstd;
  bw.slope = y_view->slope;
  return patch_launch(pp, (const float*)dy, nullptr, nullptr, 1.f, wpack_dgrad, dx, partial, acc, (hipStream_t)stream, &bw);
}

// Wide Linears (1x1x1, unit stride, more source channels than the LDS-patch weight-gradient kernel stages at once: ViViT's patch
// embedding 768 -> 128 and FeedForward 1024 -> 128): the weight gradient is independent per source channel, so it is computed
// slice by slice of <= 256 channels with the same split-precision kernel reading a channel slice of the rows.
#define WIDE_CHUNK 256
static bool wide_linear_chunks(const MdConvDesc* d) {
  static const int off = getenv("MD_WIDE_WGRAD") && atoi(getenv("MD_WIDE_WGRAD")) == 0;
  return !off && d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 &&
         md_cpad(d->Cin) > 320;      // PMAXC of the patch kernels (exact-fp32 mode: wgrad_lookup declines, the gather kernel runs)
}
// ... or, when the OUTPUT side is narrow (FeedForward 1024 -> 128), in ONE launch with the operands' roles swapped: X' := dY (Cout
// channels), dY' := X (Cin channels) is the weight gradient of a Cout -> Cin Linear, i.e. dW transposed; the slab reduction writes it
// back transposed (conv_wgrad2.hip).  ViViT cfg3: 4 slab kernels + 4 reductions per FeedForward and step become 1 + 1.
static const WgradPlan* wide_swapped_plan(const MdConvDesc* d, MdConvDesc* ds) {
  static const int off = getenv("MD_WIDE_SWAP") && atoi(getenv("MD_WIDE_SWAP")) == 0;
  if (off || md_cpad(d->Cout) > 320 || md_cpad(d->Cout) != d->Cout || md_cpad(d->Cin) != d->Cin) return nullptr;
  *ds = *d; ds->Cin = d->Cout; ds->Cout = d->Cin;
  const WgradPlan* wp = wgrad_lookup(ds, 0, 0, -1);
  return (wp && wgrad_plan_second_form(wp)) ? wp : nullptr;
}
static const WgradPlan* wide_chunk_plan(const MdConvDesc* d, int c0) {
  MdConvDesc dc = *d;
  dc.Cin = d->Cin - c0 < WIDE_CHUNK ? d->Cin - c0 : WIDE_CHUNK;
  return wgrad_lookup(&dc, md_cpad(d->Cin), c0, d->Cin);
}

extern "C" int md_conv_wgrad_fmt2(const MdConvDesc* d, const MdActView* x, int x_split, const void* dy, int dy_split, float* dw,
                                  float* workspace, void* stream) {
  if (!dy_split && !x_split) return md_conv_wgrad(d, x, (const float*)dy, dw, workspace, stream);
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!x || !x->data || !dy || !dw) return MD_ERR_NULL;
  const WgradPlan* wp = wgrad_lookup(d);
  if (!wp || (dy_split && (md_cpad(d->Cout) & 7))) return MD_ERR_UNSUPPORTED;
  if (!workspace) return MD_ERR_WORKSPACE;
  // x_split: x->data is the pre-activated, pre-split copy of the input (md_bn_act_split); scale / shift / slope are not used
  return wgrad_patch_launch(wp, d, x->data, x_split ? nullptr : x->scale, x_split ? nullptr : x->shift, x->slope, (const float*)dy, dw,
                            workspace, (hipStream_t)stream, dy_split ? 1 : 0, x_split ? 1 : 0);
}
extern "C" int md_conv_wgrad_fmt(const MdConvDesc* d, const MdActView* x, const void* dy, int dy_split, float* dw,
                                 float* workspace, void* stream) {
  return md_conv_wgrad_fmt2(d, x, 0, dy, dy_split, dw, workspace, stream);
}
// 1 when the weight gradient of this geometry can read a pre-split X (md_bn_act_split): LDS-patch kernel, not the pixel-pair stem
extern "C" int md_conv_wgrad_xsplit_ok(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  const WgradPlan* wp = wgrad_lookup(d);
  return wp && wgrad_plan_xsplit_ok(wp) ? 1 : 0;
}

extern "C" size_t md_conv_wgrad_workspace_floats(const MdConvDesc* d) {
  if (check_desc(d) != MD_OK) return 0;
  if (const WgradPlan* wp = wgrad_lookup(d)) return wgrad_patch_workspace_floats(wp);
  if (wide_linear_chunks(d)) {
    size_t need = 0;
    MdConvDesc ds;
    if (const WgradPlan* wp = wide_swapped_plan(d, &ds)) need = wgrad_patch_workspace_floats(wp);     // (the slices' need is kept too:
    for (int c0 = 0; c0 < d->Cin; c0 += WIDE_CHUNK) {                                                  //  the form is chosen per call)
      const WgradPlan* wp = wide_chunk_plan(d, c0);
      if (!wp) return 0;
      const size_t n = wgrad_patch_workspace_floats(wp);
      if (n > need) need = n;
    }
    return need;
  }
  return 0;
}

extern "C" int md_conv_wgrad(const MdConvDesc* d, const MdActView* x, const float* dy_raw, float* dw, float* workspace,
                             void* stream) {
  int rc = check_desc(d);
  if (rc != MD_OK) return rc;
  if (!x || !x->data || !dy_raw || !dw) return MD_ERR_NULL;
  if (const WgradPlan* wp = wgrad_lookup(d)) {
    if (!workspace) return MD_ERR_WORKSPACE;
    return wgrad_patch_launch(wp, d, x->data, x->scale, x->shift, x->slope, dy_raw, dw, workspace, (hipStream_t)stream);
  }
  if (wide_linear_chunks(d) && workspace && !x->scale) {
    MdConvDesc ds;
    if (const WgradPlan* wp = wide_swapped_plan(d, &ds)) {
      MdActView dyv; dyv.data = dy_raw; dyv.scale = nullptr; dyv.shift = nullptr; dyv.slope = 1.f;
      rc = wgrad_patch_launch(wp, &ds, dyv.data, nullptr, nullptr, 1.f, x->data, dw, workspace, (hipStream_t)stream);
      if (rc != MD_ERR_UNSUPPORTED) return rc;
    }
  }
  if (wide_linear_chunks(d) && workspace) {
    bool all = true;
    for (int c0 = 0; c0 < d->Cin && all; c0 += WIDE_CHUNK) all = wide_chunk_plan(d, c0) != nullptr;
    if (all) {
      for (int c0 = 0; c0 < d->Cin; c0 += WIDE_CHUNK) {
        MdConvDesc dc = *d;
        dc.Cin = d->Cin - c0 < WIDE_CHUNK ? d->Cin - c0 : WIDE_CHUNK;
        rc = wgrad_patch_launch(wide_chunk_plan(d, c0), &dc, x->data, x->scale, x->shift, x->slope, dy_raw, dw, workspace, (hipStream_t)stream);
        if (rc != MD_OK) return rc;
      }
      return MD_OK;
    }
  }
  Geom g = geom_fwd(d);
  const int taps = d->kt * d->kh * d->kw;
  if (hipMemsetAsync(dw, 0, (size_t)d->Cout * d->Cin * taps * 4, (hipStream_t)stream) != hipSuccess) return MD_ERR_LAUNCH;
  const int ktiles = md_cdiv(g.Kc * 4, WK);
  const int N16 = md_round_up(d->Cout, 16);
  const int npb = pick_n_per_blk(N16);
  const int nchunks = md_cdiv(N16, npb);
  // aim for ~4096 workgroups; each pixel slice a multiple of WR rows
  int slices = md_cdiv(4096, ktiles * nchunks);
  int rows_per_blk = md_round_up(md_cdiv(g.M, slices), WR);
  if (rows_per_blk < 4 * WR) rows_per_blk = 4 * WR;
  slices = md_cdiv(g.M, rows_per_blk);
  if (slices > 65535) return MD_ERR_UNSUPPORTED;
  dim3 grid(ktiles, slices, nchunks);
  const size_t lds = (size_t)(WR * PX + WR * PY) * 4;
  MD_KLAUNCH(k_conv_wgrad, grid, dim3(256), lds, (hipStream_t)stream, g, x->data, x->scale, x->shift,
                     x->slope, dy_raw, dw, d->Cin, d->Cout, taps, rows_per_blk, npb);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
