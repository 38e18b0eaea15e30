// Tubelet / patch embedding of ViViT (reference src/models/ViViT.py:141-148, 175-184):
//   'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)'  ->  Linear(p*p*c, dim)  ->  [space token | patches] + positional table
// as ONE gather-GEMM: the p x p x c patch of every token is gathered straight from the clip while the A tile is staged
// into LDS (no rearranged copy of the clip exists), the products are the same three split-precision MFMAs as everywhere
// else (fp16 hi+lo forward, bf16 hi+lo for the weight gradient), and the epilogue adds the Linear bias and the positional
// row and writes token j of frame (b, t) to row 1 + j of that frame's (n + 1) x dim block; row 0 (space token + its
// positional row) comes from k_patch_embed_token.
//
// K is walked in the order k' = (c, p1, p2) -- 16-byte runs of the clip are contiguous along p2 -- so the weight operand is
// W'[n][k'] = W[n][(p1 p + p2) C + c]; the host passes W' (a 0.4 MB permuted copy made by autograd, which also un-permutes
// the gradient).  The clip is addressed by strides (sb, st, sc floats; rows contiguous), so both (b, t, c, H, W) and the
// permuted view of a (b, c, t, H, W) clip are read in place.
#include "patch_common.h"
#include <cstdlib>
#include "../../include/mi355x_disrupt.h"

#define PE_M 128          // token rows per workgroup
#define PE_K 32           // k' per stage
#define PE_PD 3           // stages of both operands in flight (registers)
#define PE_P 80           // LDS row pitch of a 32-half stage (bytes): 64 + pad, (80/16) % 4 == 1 -> conflict free with lg * 16

struct PEGeom {
  int B, T, C, H, W;          // clip
  int p, lp;                  // patch size (power of two), log2
  int nh, nw, n;              // patches per frame
  int K, Kp;                  // p*p*C, rounded up to PE_K
  int dim, N16;
  long long sb, st, sc;       // clip strides in floats
  int M;                      // B*T*n token rows
};

// element offset of (row m, k') in the clip; k' % 4 == 0 runs of 4 stay inside one image row because p % 4 == 0
__device__ __forceinline__ long long pe_row_base(const PEGeom& g, int m) {
  const int bt = m / g.n, j = m - bt * g.n;
  const int b = bt / g.T, t = bt - b * g.T;
  const int py = j / g.nw, px = j - py * g.nw;
  return (long long)b * g.sb + (long long)t * g.st + (long long)(py << g.lp) * g.W + (px << g.lp);
}
__device__ __forceinline__ long long pe_k_off(const PEGeom& g, int k) {
  const int c = k >> (2 * g.lp), r = k & ((1 << (2 * g.lp)) - 1);
  return (long long)c * g.sc + (long long)(r >> g.lp) * g.W + (r & (g.p - 1));
}

__global__ __launch_bounds__(256) void k_patch_embed_fwd(PEGeom g, const float* __restrict__ x, const float* __restrict__ Wp,
                                                        const float* __restrict__ bias, const float* __restrict__ pos,
                                                        float* __restrict__ out, int npb) {
  __shared__ __attribute__((aligned(16))) char lds[4 * PE_M * PE_P];
  char* aH = lds; char* aL = lds + PE_M * PE_P; char* bH = lds + 2 * PE_M * PE_P; char* bL = lds + 3 * PE_M * PE_P;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * PE_M, n0 = blockIdx.y * npb;
  const int ncols = min(npb, g.N16 - n0), nt = ncols >> 4;
  const int r = t >> 1, h = t & 1;                      // staging: row r, floats 16 h .. 16 h + 15 of the stage
  const bool arow = m0 + r < g.M, brow = r < ncols && n0 + r < g.dim;
  const long long abase = arow ? pe_row_base(g, m0 + r) : 0;
  const float* bp = Wp + (size_t)(n0 + r) * g.K + h * 16;
  // PE_PD stages of both operands in flight, every load issued unconditionally (a position outside the operand reads a valid dummy
  // address and is zeroed afterwards) and the loop body PE_PD whole stages: see k_linear_split (conv_patch.hip)
  float4 ra[PE_PD][4], rb[PE_PD][4];
  auto load = [&](int kb, float4 (&va)[4], float4 (&vb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb * PE_K + h * 16 + i * 4;
      const bool kin = k < g.K;
      const float4 a = *(const float4*)((arow && kin) ? x + abase + pe_k_off(g, k) : x);
      const float4 b = *(const float4*)((brow && kin) ? bp + kb * PE_K + i * 4 : Wp);
      va[i] = (arow && kin) ? a : make_float4(0.f, 0.f, 0.f, 0.f);
      vb[i] = (brow && kin) ? b : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](const float4 (&v)[4], char* hi, char* lo) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float f[8] = {v[2 * q].x, v[2 * q].y, v[2 * q].z, v[2 * q].w, v[2 * q + 1].x, v[2 * q + 1].y, v[2 * q + 1].z, v[2 * q + 1].w};
      uint4 uh, ul;
      split8_f16(f, uh, ul);
      *(uint4*)(hi + r * PE_P + h * 32 + q * 16) = uh;
      *(uint4*)(lo + r * PE_P + h * 32 + q * 16) = ul;
    }
  };
  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nstages = g.Kp / PE_K;
#pragma unroll
  for (int u = 0; u < PE_PD; ++u) load(u, ra[u], rb[u]);
  for (int kb0 = 0; kb0 < nstages; kb0 += PE_PD) {
#pragma unroll
  for (int u = 0; u < PE_PD; ++u) {
    const int kb = kb0 + u;
    __syncthreads();
    store(ra[u], aH, aL);
    store(rb[u], bH, bL);
    __syncthreads();
    load(kb + PE_PD, ra[u], rb[u]);
    uint4 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *(const uint4*)(aH + (wave * 32 + i * 16 + li) * PE_P + lg * 16);
      al[i] = *(const uint4*)(aL + (wave * 32 + i * 16 + li) * PE_P + lg * 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nt) {
        const uint4 bh = *(const uint4*)(bH + (j * 16 + li) * PE_P + lg * 16);
        const uint4 bl = *(const uint4*)(bL + (j * 16 + li) * PE_P + lg * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][j] = mma<true>(ah[i], bl, acc[i][j]);          // smallest terms first
          acc[i][j] = mma<true>(al[i], bh, acc[i][j]);
          acc[i][j] = mma<true>(ah[i], bh, acc[i][j]);
        }
      }
    }
  }
  }
  // epilogue: + Linear bias + positional row; token j of frame bt -> row bt (n + 1) + 1 + j
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + wave * 32 + i * 16 + lg * 4 + q;
      if (m < g.M) {
        const int bt = m / g.n, j = m - bt * g.n;
        const int tt = bt % g.T;
        float* orow = out + ((size_t)bt * (g.n + 1) + 1 + j) * g.dim;
        const float* prow = pos + ((size_t)tt * (g.n + 1) + 1 + j) * g.dim;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int col = n0 + jj * 16 + li;
          if (jj < nt && col < g.dim) orow[col] = acc[i][jj][q] + bias[col] + prow[col];
        }
      }
    }
  }
}

// row 0 of every frame: space token + positional row 0
__global__ __launch_bounds__(256) void k_patch_embed_token(PEGeom g, const float* __restrict__ token, const float* __restrict__ pos,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.B * g.T * g.dim) return;
  const int bt = i / g.dim, col = i - bt * g.dim;
  const int tt = bt % g.T;
  out[(size_t)bt * (g.n + 1) * g.dim + col] = token[col] + pos[(size_t)tt * (g.n + 1) * g.dim + col];
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW'[n][k'] = sum over token rows m of  g_out[row(m)][n] * patch[m][k'] ,  bf16 hi+lo split, three MFMAs per product.
// Workgroup = one 64-wide chunk of k' (wave w owns its 16-row k tile w) x all output columns, over a slice of 128-row boxes.
// Both operands are staged as [row][column] bf16 images and fetched with ds_read_b64_tr_b16 (the reduction axis -- token rows
// -- is the slow axis of both; see k_wgrad_patch).  Partial sums per slice go to a slab, summed in a fixed order by
// k_patch_embed_wreduce.
#define PW_KC 64
#define PW_APITCH 160           // 8 chunks of 16 B + pad: 16 * (4*2 + 2)
typedef short pw_s16x4 __attribute__((ext_vector_type(4)));
typedef short pw_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pw_tr_read2(const char* p0, const char* p1) {
  const pw_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_s16x4 __attribute__((address_space(3)))*)p0);
  const pw_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_s16x4 __attribute__((address_space(3)))*)p1);
  pw_s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int NT8>      // output column tiles (N16 / 16), <= 8
__global__ __launch_bounds__(256) void k_patch_embed_wgrad(PEGeom g, const float* __restrict__ x, const float* __restrict__ gout,
                                                          float* __restrict__ slab, int boxes_per_wg, int ypitch) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* sA = sm;                                   // [128][PW_APITCH] hi | lo
  char* sY = sm + 2 * PE_M * PW_APITCH;            // [128][ypitch] hi | lo
  const int ylo = PE_M * ypitch;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lg = lane >> 4, lq = li >> 2, lp = li & 3;
  const int k0 = blockIdx.x * PW_KC;
  const int nboxes = (g.M + PE_M - 1) / PE_M;
  const int box_beg = blockIdx.y * boxes_per_wg, box_end = min(nboxes, box_beg + boxes_per_wg);
  f32x4 acc[NT8];
#pragma unroll
  for (int j = 0; j < NT8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int NC = g.N16 / 8;                         // 8-column chunks of a g_out row
  // The loads of box n + 1 (both operands, registers) are in flight during the matrix phase of box n; the g_out rows are read as two
  // 16-byte loads per item when the width allows (dim % 8 == 0), element by element otherwise.
  constexpr int NYI = NT8;                          // g_out items (8 floats) per thread: 128 rows x 2 NT8 chunks / 256 threads
  const bool yvec = (g.dim & 7) == 0 && NC == 2 * NT8;
  float4 ra[4][2], ry[NYI][2];
  auto load_box = [&](int box) {
    const int m0 = box * PE_M;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int item = u * 256 + t;
      const int r = item >> 3, c8 = item & 7;
      const int k = k0 + c8 * 8;
      const bool ok = m0 + r < g.M && k < g.K;
      const float* s = ok ? x + pe_row_base(g, m0 + r) + pe_k_off(g, k) : x;      // p >= 8: 8 consecutive k' share an image row
      const float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
      ra[u][0] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f); ra[u][1] = ok ? b : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NYI; ++u) {
      const int item = u * 256 + t;
      const int r = item / NC, c = item - r * NC;
      const int m = m0 + r;
      const bool ok = item < PE_M * NC && m < g.M;
      const int bt = ok ? m / g.n : 0, j = ok ? m - bt * g.n : 0;
      const float* s = ok ? gout + ((size_t)bt * (g.n + 1) + 1 + j) * g.dim + c * 8 : gout;
      if (yvec) {
        const float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
        ry[u][0] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f); ry[u][1] = ok ? b : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (c * 8 + e < g.dim) v[e] = s[e];
        }
        ry[u][0] = make_float4(v[0], v[1], v[2], v[3]); ry[u][1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  };
  if (box_beg < box_end) load_box(box_beg);
  for (int box = box_beg; box < box_end; ++box) {
    __syncthreads();                                // previous box consumed
    // ---- A: 128 rows x 64 k' = 8 chunks of 8 floats per row: 1024 items, 4 per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int item = u * 256 + t;
      const int r = item >> 3, c8 = item & 7;
      const float v[8] = {ra[u][0].x, ra[u][0].y, ra[u][0].z, ra[u][0].w, ra[u][1].x, ra[u][1].y, ra[u][1].z, ra[u][1].w};
      uint4 hi, lo;
      split8(v, hi, lo);
      *(uint4*)(sA + r * PW_APITCH + c8 * 16) = hi;
      *(uint4*)(sA + PE_M * PW_APITCH + r * PW_APITCH + c8 * 16) = lo;
    }
    // ---- g_out rows (frame-block row 1 + j)
#pragma unroll
    for (int u = 0; u < NYI; ++u) {
      const int item = u * 256 + t;
      if (item < PE_M * NC) {
        const int r = item / NC, c = item - r * NC;
        const float v[8] = {ry[u][0].x, ry[u][0].y, ry[u][0].z, ry[u][0].w, ry[u][1].x, ry[u][1].y, ry[u][1].z, ry[u][1].w};
        uint4 hi, lo;
        split8(v, hi, lo);
        *(uint4*)(sY + r * ypitch + c * 16) = hi;
        *(uint4*)(sY + ylo + r * ypitch + c * 16) = lo;
      }
    }
    __syncthreads();
    if (box + 1 < box_end) load_box(box + 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r0 = s * 32 + lg * 4 + lq;
      const int xa = r0 * PW_APITCH + wave * 32 + lp * 8, xb = (r0 + 16) * PW_APITCH + wave * 32 + lp * 8;
      const int ya = r0 * ypitch + lp * 8, yb = (r0 + 16) * ypitch + lp * 8;
      const bf16x8 ah = pw_tr_read2(sA + xa, sA + xb);
      const bf16x8 al = pw_tr_read2(sA + PE_M * PW_APITCH + xa, sA + PE_M * PW_APITCH + xb);
#pragma unroll
      for (int j = 0; j < NT8; ++j) {
        const bf16x8 bh = pw_tr_read2(sY + ya + j * 32, sY + yb + j * 32);
        const bf16x8 bl = pw_tr_read2(sY + ylo + ya + j * 32, sY + ylo + yb + j * 32);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
  }
  // slab[slice][k' of this chunk][N16]: D rows = k' (4 lg + reg), columns = output channel (li)
  float* o = slab + ((size_t)blockIdx.y * g.Kp + k0 + wave * 16) * g.N16;
#pragma unroll
  for (int j = 0; j < NT8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(size_t)(lg * 4 + r) * g.N16 + j * 16 + li] = acc[j][r];
}

__global__ __launch_bounds__(256) void k_patch_embed_wreduce(PEGeom g, const float* __restrict__ slab, int nslices, float* __restrict__ dW) {
  const int idx = blockIdx.x * 256 + threadIdx.x;           // over [k'][N16], column fastest
  if (idx >= g.K * g.N16) return;
  const int k = idx / g.N16, n = idx - k * g.N16;
  if (n >= g.dim) return;
  float s = 0.f;
  for (int i = 0; i < nslices; ++i) s += slab[((size_t)i * g.Kp + k) * g.N16 + n];
  dW[(size_t)n * g.K + k] = s;
}

// ------------------------------------------------------------------------------------------------ host side
static int pe_geom(PEGeom* g, int B, int T, int C, int H, int W, int p, int dim, long long sb, long long st, long long sc) {
  if (B <= 0 || T <= 0 || C <= 0 || H <= 0 || W <= 0 || p < 8 || (p & (p - 1)) || H % p || W % p || dim <= 0) return MD_ERR_BAD_SHAPE;
  if ((W & 3) || (sb & 3) || (st & 3) || (sc & 3)) return MD_ERR_UNSUPPORTED;        // 16-byte runs
  g->B = B; g->T = T; g->C = C; g->H = H; g->W = W; g->p = p; g->lp = 0;
  while ((1 << g->lp) < p) ++g->lp;
  g->nh = H / p; g->nw = W / p; g->n = g->nh * g->nw;
  g->K = p * p * C; g->Kp = md_round_up(g->K, PW_KC);
  g->dim = dim; g->N16 = md_round_up(dim, 16);
  g->sb = sb; g->st = st; g->sc = sc;
  const long long M = (long long)B * T * g->n;
  if (M >= (1ll << 31) / (g->n + 1)) return MD_ERR_UNSUPPORTED;
  g->M = (int)M;
  return MD_OK;
}

extern "C" int md_patch_embed_fwd(const float* x, int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int64_t sb, int64_t st,
                                  int64_t sc, int32_t patch, const float* w_perm, const float* bias, const float* pos,
                                  const float* token, int32_t dim, float* out, void* stream) {
  if (!x || !w_perm || !bias || !pos || !token || !out) return MD_ERR_NULL;
  PEGeom g;
  int rc = pe_geom(&g, B, T, C, H, W, patch, dim, sb, st, sc);
  if (rc) return rc;
  if (md_get_exact_fp32()) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int npb = 128;
  static const int fill = getenv("MD_PE_FILL") ? atoi(getenv("MD_PE_FILL")) : 512;
  while (md_cdiv(g.M, PE_M) * md_cdiv(g.N16, npb) < fill && npb > 32) npb >>= 1;
  MD_KLAUNCH(k_patch_embed_fwd, dim3(md_cdiv(g.M, PE_M), md_cdiv(g.N16, npb)), dim3(256), 0, s, g, x, w_perm, bias, pos, out, npb);
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_patch_embed_token, dim3(md_cdiv(B * T * dim, 256)), dim3(256), 0, s, g, token, pos, out);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

static int pe_slices(const PEGeom& g) {
  const int nboxes = md_cdiv(g.M, PE_M);
  int want = md_cdiv(512, g.Kp / PW_KC);
  if (want > nboxes) want = nboxes;
  if (want < 1) want = 1;
  const int per = md_cdiv(nboxes, want);
  return md_cdiv(nboxes, per);
}

extern "C" size_t md_patch_embed_wgrad_workspace_floats(int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int32_t patch, int32_t dim) {
  PEGeom g;
  if (pe_geom(&g, B, T, C, H, W, patch, dim, 0, 0, 0)) return 0;
  return (size_t)pe_slices(g) * g.Kp * g.N16;
}

extern "C" int md_patch_embed_wgrad(const float* x, int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, int64_t sb, int64_t st,
                                    int64_t sc, int32_t patch, const float* gout, int32_t dim, float* dw_perm, float* workspace,
                                    void* stream) {
  if (!x || !gout || !dw_perm || !workspace) return MD_ERR_NULL;
  PEGeom g;
  int rc = pe_geom(&g, B, T, C, H, W, patch, dim, sb, st, sc);
  if (rc) return rc;
  if (g.N16 > 128 || (dim & 3)) return MD_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nboxes = md_cdiv(g.M, PE_M);
  const int nsl = pe_slices(g);
  const int per = md_cdiv(nboxes, nsl);
  int ypitch = 0; { int c = g.N16 / 8; int u = c; while ((u & 3) != 2) ++u; ypitch = u * 16; }
  const size_t lds = (size_t)2 * PE_M * PW_APITCH + (size_t)2 * PE_M * ypitch;
  const dim3 grid(g.Kp / PW_KC, nsl);
#define PE_WG(NT_)                                                                                                       \
  do {                                                                                                                   \
    if (hipFuncSetAttribute((const void*)k_patch_embed_wgrad<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) \
      return MD_ERR_LAUNCH;                                                                                              \
    MD_KLAUNCH(k_patch_embed_wgrad<NT_>, grid, dim3(256), lds, s, g, x, gout, workspace, per, ypitch);                   \
  } while (0)
  switch (g.N16 / 16) {
    case 1: PE_WG(1); break; case 2: PE_WG(2); break; case 3: PE_WG(3); break; case 4: PE_WG(4); break;
    case 5: PE_WG(5); break; case 6: PE_WG(6); break; case 7: PE_WG(7); break; default: PE_WG(8); break;
  }
  MD_CHECK_LAUNCH();
  MD_KLAUNCH(k_patch_embed_wreduce, dim3(md_cdiv(g.K * g.N16, 256)), dim3(256), 0, s, g, (const float*)workspace, nsl, dw_perm);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
