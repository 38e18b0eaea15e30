// Weight gradient, second form (round 3): the same arithmetic as k_wgrad_patch_pf (conv_wgrad.hip) --
//   dW[k = (tap, cin)][cout] = sum over pixels  X[pixel + tap][cin] * dY[pixel][cout],
// X and dY staged once per box as [pixel][channel] bf16 hi|lo images, both MFMA operands fetched with
// ds_read_b64_tr_b16, three bf16 MFMAs per product, per-slice slabs reduced in a fixed order -- re-tiled so that TWO
// workgroups fit a CU:
//   * measured (profiles/r03a_*): k_wgrad_patch_pf needs 350-410 registers per lane and 70-155 KB of LDS, so one 4-wave
//     workgroup owns a CU and its phases add up (per box 3.4k cycles commit + 4.6k matrix + 1.6k barriers); the layers
//     with a short matrix phase (72 -> 32, k311) wait a whole HBM round trip per box behind a prefetch depth of one box;
//   * here a box has 64 output pixels (two 32-pixel MFMA steps): the staging registers of the box in flight drop from
//     96 to <= 48, the B fragments are held for half of the column tiles at a time, and the kernel is compiled for two
//     waves per SIMD (__launch_bounds__(256, 2)): while one workgroup commits its next box (BatchNorm-on-read, hi/lo
//     split, LDS writes) or waits for loads, the other one's MFMAs run on the same SIMDs, and two boxes are in flight
//     per CU;
//   * output tiles that exceed one workgroup (> 20 k-tiles) are split over k-groups by INPUT CHANNEL instead of by tap:
//     a k-group stages only its 16-channel tiles of the X patch (the tap-major split of the first form staged the whole
//     patch in every group);
//   * every geometry prefetches (the first form fell back to a non-prefetching kernel above 1792 staged items).
// Not handled here (the first form stays): the pixel-pair stem, more than 20 taps, pre-split operand formats, channel slices
// of a wider tensor (wide Linears).
#include "common.h"
#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>
#include <array>
#include "patch_common.h"

typedef short w2_s16x4 __attribute__((ext_vector_type(4)));
typedef short w2_s16x8 __attribute__((ext_vector_type(8)));

struct W2Geom {
  int Ti, Hi, Wi, Cpi;        // X: dims, floats per pixel
  int To, Ho, Wo, Cpo;        // dY
  int kh, kw, khw, taps;
  int org_t, org_h, org_w;
  int st, sh, sw;
  int bt, by, bx, byx, nbt, nby, nbx;
  int pt, py, px, pyx, P;
  int KT, KTg, nkg;           // 16-channel tiles of Cin: total, per k-group, groups
  int nktg;                   // k-tiles of one group = taps * KTg
  int C8i, ppitch;            // X patch image of one group: C8i = 2 * KTg chunks per pixel (hi half; lo half W2_LO bytes behind)
  int NC, ypitch;             // dY image: NC = 2 * nrep chunks per row
  int ktw, nrep, nng;
  int nboxes, boxes_per_wg;
  int N16;
  int pmb;                    // pixels per box (32 * NS)
  unsigned magicC8, magicNC, m_pyx, m_px, m_byx, m_bx;
  int off_y, off_rows, off_scale;
  unsigned x_bytes, y_bytes;
  int xpitch, xc_base;        // floats per X pixel in memory and first channel read (= Cpi, 0 unless X is a channel slice of a wider tensor)
  int dw_cin, dw_c0;          // dW's full Cin and the slice's first channel (= Cin, 0 otherwise)
};

__device__ __forceinline__ bf16x8 w2_tr_read2(const char* p0, const char* p1) {
  const w2_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_s16x4 __attribute__((address_space(3)))*)p0);
  const w2_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w2_s16x4 __attribute__((address_space(3)))*)p1);
  w2_s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// staged 32-byte items per thread: X patch / dY rows.  Large accumulator tiles leave room for fewer.
template <int KTW, int NREP> struct W2Items { static constexpr int NX = (KTW * NREP > 20) ? 3 : 5; static constexpr int NY = 3; };

// lo half of an LDS image = its hi half + W2_LO bytes (a compile-time distance: the lo accesses use the instruction offset on
// the hi address, no second address register)
#define W2_LO 40960
// TEAMS: one 8-wave workgroup per CU whose two 4-wave teams are what the two workgroups of the plain form are -- own boxes, own LDS
// images (team stride W2_TS, lo halves W2_LO_T behind), synchronised only inside a team (LDS counter) -- but they add their
// accumulators through LDS at the end: ONE slab per CU instead of two (the slabs were 0.8 GB written and read back per step).
#define W2_TS 24576
#define W2_LO_T 49152

template <int KTW, int NREP, int NS, bool TEAMS = false>
__global__ __launch_bounds__(TEAMS ? 512 : 256, 2) void k_wgrad2(
    W2Geom g, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const float* __restrict__ dy, float* __restrict__ slab, int dbg) {
  // dbg (MD_DBG2, timing experiments only): 1 no global loads, 2 no matrix phase, 4 no commit
  extern __shared__ __attribute__((aligned(16))) char sm_all[];
  constexpr int LO = TEAMS ? W2_LO_T : W2_LO;
  const int team = TEAMS ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  char* sm = sm_all + team * W2_TS;
  char* sP = sm;
  char* sY = sm + g.off_y;
  int* sRx = (int*)(sm + g.off_rows);           // [32 * NS] X patch byte offset of each output row (box independent)
  float* sScale = (float*)(sm + g.off_scale);   // [C8i * 8] scale | shift of this k-group's channels
  float* sShift = sScale + g.C8i * 8;

  constexpr int NT = 256, PMB = 32 * NS;
  constexpr int NX = W2Items<KTW, NREP>::NX, NY = W2Items<KTW, NREP>::NY;
  constexpr int JH = NREP > 3 ? (NREP + 1) / 2 : NREP;             // column tiles whose B fragments are live together
  const int t = TEAMS ? ((int)threadIdx.x & 255) : (int)threadIdx.x;      // thread within the team
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // synchronisation of the four waves that share the images: the workgroup barrier, or a four-wave barrier on an LDS counter
  int sync_target = 0;                                 // (TEAMS) the teams' counters sit in the last 128 B of the hi regions
  auto sync = [&]() {
    if constexpr (!TEAMS) { __syncthreads(); }
    else {
      int* c = (int*)(sm_all + W2_LO_T - 128) + team * 16;
      sync_target += 4;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < sync_target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  };
  if (TEAMS) {
    if (threadIdx.x < 32) ((int*)(sm_all + W2_LO_T - 128))[threadIdx.x] = 0;
    __syncthreads();
  }
  const int li = lane & 15, lg = lane >> 4;
  const int lq = li >> 2, lp = li & 3;
  const int kg = blockIdx.y / g.nng, ng = blockIdx.y - kg * g.nng;
  const int n0 = ng * g.nrep * 16;              // first dY channel of this workgroup
  const int xc0 = kg * g.KTg * 16;              // first X channel of this k-group
  const int lt0 = wave * KTW;                   // first k-tile (within the group) of this wave
  const bool prologue = pscale != nullptr;
  if (prologue) for (int c = t; c < g.C8i * 8; c += NT) {
    const bool ok = xc0 + c < g.Cpi;            // channel padding: scale = shift = 0 gives leaky(0 * x + 0) = 0
    sScale[c] = ok ? pscale[g.xc_base + xc0 + c] : 0.f; sShift[c] = ok ? pshift[g.xc_base + xc0 + c] : 0.f;
  }
  if (t < PMB) {
    const int rt = mdiv(t, g.m_byx); const int r = t - rt * g.byx;
    const int ry = mdiv(r, g.m_bx); const int rx = r - ry * g.bx;
    sRx[t] = (rt < g.bt) ? ((rt * g.st * g.py + ry * g.sh) * g.px + rx * g.sw) * g.ppitch : 0;
  }

  f32x4 acc[KTW][NREP];
#pragma unroll
  for (int a = 0; a < KTW; ++a)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // byte offset inside the X patch of each of this wave's k-tiles: tap pixel offset + 32 B per 16 channels
  int koff[KTW];
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int lt = lt0 + a;
    int o = 0;
    if (lt < g.nktg) {
      const int tap = lt / g.KTg; const int c16 = lt - tap * g.KTg;
      const int dt = tap / g.khw; const int r = tap - dt * g.khw;
      const int dyy = r / g.kw; const int dxx = r - dyy * g.kw;
      o = ((dt * g.py + dyy) * g.px + dxx) * g.ppitch + c16 * 32;
    }
    koff[a] = o;
  }

  // ---- box-independent decode of this thread's items.  An item lies inside its tensor iff each of its three patch (box)
  // coordinates does, so it carries one bit per coordinate -- bit ppx in word b_x, bits ppy and py + ppt in word b_yt -- and a
  // box provides, as scalars, the sets of valid coordinates: a box then costs an AND and a compare per word and item instead of
  // the coordinate arithmetic (measured, profiles/r03c: the request arithmetic was a third of the vector instructions of a box).
  const int totX = g.P * g.C8i, totY = PMB * g.NC;
  const int xcv4 = max(0, min(g.C8i * 2, (g.Cpi - xc0) >> 2));     // valid float4 units of a patch pixel from xc0
  const int ycv4 = max(0, min(g.NC * 2, (g.Cpo - n0) >> 2));       // valid float4 units of a dY row from n0
  constexpr unsigned NEVER = 0x80000000u;                          // b_yt of an item that is never loaded (no box sets bit 31)
  int xdst[NX], xrel[NX]; unsigned xbx[NX], xbyt[NX];      // xdst: LDS byte offset | c8 << 16 (-1: no item)
  int ydst[NY], yrel[NY]; unsigned ybx[NY], ybyt[NY];
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int item = u * NT + t;
    xdst[u] = -1; xrel[u] = 0; xbx[u] = 0; xbyt[u] = NEVER;
    if (item < totX) {
      const int pixel = mdiv(item, g.magicC8);
      const int c8 = item - pixel * g.C8i;
      const int ppt = mdiv(pixel, g.m_pyx); const int r = pixel - ppt * g.pyx;
      const int ppy = mdiv(r, g.m_px); const int ppx = r - ppy * g.px;
      xdst[u] = (pixel * g.ppitch + c8 * 16) | (c8 << 16);
      xrel[u] = ((ppt * g.Hi + ppy) * g.Wi + ppx) * g.xpitch + g.xc_base + xc0 + c8 * 8;
      xbx[u] = 1u << ppx;
      xbyt[u] = c8 * 2 < xcv4 ? ((1u << ppy) | (1u << (g.py + ppt))) : NEVER;       // (a chunk past the last channel is never loaded: zeros are committed)
    }
  }
#pragma unroll
  for (int u = 0; u < NY; ++u) {
    const int item = u * NT + t;
    ydst[u] = -1; yrel[u] = 0; ybx[u] = 0; ybyt[u] = NEVER;
    if (item < totY) {
      const int row = mdiv(item, g.magicNC);
      const int c = item - row * g.NC;
      const int rt = mdiv(row, g.m_byx); const int r = row - rt * g.byx;
      const int ry = mdiv(r, g.m_bx); const int rx = r - ry * g.bx;
      yrel[u] = ((rt * g.Ho + ry) * g.Wo + rx) * g.Cpo + n0 + c * 8;
      ydst[u] = row * g.ypitch + c * 16;
      ybx[u] = 1u << (rx & 31);
      ybyt[u] = (rt < g.bt && c * 2 < ycv4) ? ((1u << ry) | (1u << (g.by + rt))) : NEVER;      // rows past the box / chunks past the last channel: zeros
    }
  }
  // item slot u exists for this wave (wave-uniform: the whole slot is skipped otherwise)
  auto has_x = [&](int u) { return u * NT + wave * 64 < totX; };
  auto has_y = [&](int u) { return u * NT + wave * 64 < totY; };
  // bits lo .. hi (lo <= hi < 32), or none
  auto bits = [](int lo, int hi) -> unsigned {
    if (lo > hi) return 0u;
    return ((hi >= 31 ? 0u : (2u << hi)) - 1u) & ~((1u << lo) - 1u);
  };

  float4 xa_[NX], xb_[NX], ya_[NY], yb_[NY];
  unsigned xfl = 0;           // bit u: X item u was loaded (inside the tensor) -> BatchNorm-on-read applies
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(src, g.x_bytes), yrs = make_rsrc(dy, g.y_bytes);
  // grid position of the box being requested (wave-uniform, stepped box by box: no divisions in the loop)
  int q_xb, q_yb, q_tb, q_n;
  {
    int b = ((TEAMS ? 2 : 1) * blockIdx.x + team) * g.boxes_per_wg;
    q_xb = b % g.nbx; b /= g.nbx;
    q_yb = b % g.nby; b /= g.nby;
    q_tb = b % g.nbt; q_n = b / g.nbt;
  }
  auto issue = [&]() {
    const int t0 = q_tb * g.bt, y0 = q_yb * g.by, x0 = q_xb * g.bx;
    const int ot = t0 * g.st + g.org_t, oh = y0 * g.sh + g.org_h, ow = x0 * g.sw + g.org_w;
    // valid patch coordinates: 0 <= o + pp < dim; valid box coordinates: 0 + r < dim - 0
    const unsigned sx_x = bits(max(0, -ow), min(g.px - 1, g.Wi - 1 - ow));
    const unsigned sx_yt = bits(max(0, -oh), min(g.py - 1, g.Hi - 1 - oh)) | (bits(max(0, -ot), min(g.pt - 1, g.Ti - 1 - ot)) << g.py);
    const unsigned sy_x = bits(0, min(g.bx, g.Wo - x0) - 1);
    const unsigned sy_yt = bits(0, min(g.by, g.Ho - y0) - 1) | (bits(0, min(g.bt, g.To - t0) - 1) << g.by);
    const int q_xbase = (((q_n * g.Ti + ot) * g.Hi + oh) * g.Wi + ow) * g.xpitch;
    const int q_ybase = (((q_n * g.To + t0) * g.Ho + y0) * g.Wo + x0) * g.Cpo;
    xfl = 0;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      if (has_x(u)) {
        const bool in = (xbx[u] & sx_x) && ((xbyt[u] & sx_yt) == xbyt[u]) && !(dbg & 1);
        const unsigned off = in ? (unsigned)(q_xbase + xrel[u]) * 4u : MD_OOB;
        xa_[u] = buf_load4(xrs, off);
        // (upper half = channel padding when Cpi % 8 == 4: whatever finite values are read there meet a zero scale, and
        // their rows of dW are dropped by the reduction)
        xb_[u] = buf_load4(xrs, off + 16u);
        xfl |= (in ? 1u : 0u) << u;
      }
    }
#pragma unroll
    for (int u = 0; u < NY; ++u) {
      if (has_y(u)) {
        const bool in = (ybx[u] & sy_x) && ((ybyt[u] & sy_yt) == ybyt[u]) && !(dbg & 1);
        const unsigned off = in ? (unsigned)(q_ybase + yrel[u]) * 4u : MD_OOB;
        ya_[u] = buf_load4(yrs, off);
        yb_[u] = buf_load4(yrs, off + 16u);         // (columns past Cout: dropped by the reduction)
      }
    }
    if (++q_xb == g.nbx) { q_xb = 0; if (++q_yb == g.nby) { q_yb = 0; if (++q_tb == g.nbt) { q_tb = 0; ++q_n; } } }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      if (has_x(u) && xdst[u] >= 0) {
        float v[8] = {xa_[u].x, xa_[u].y, xa_[u].z, xa_[u].w, xb_[u].x, xb_[u].y, xb_[u].z, xb_[u].w};
        if (prologue && ((xfl >> u) & 1u)) {
          const int c8 = xdst[u] >> 16;
          const float* sc = sScale + c8 * 8; const float* sh = sShift + c8 * 8;
          bn_leaky8(v, *(const f32x4*)sc, *(const f32x4*)(sc + 4), *(const f32x4*)sh, *(const f32x4*)(sh + 4), pslope);
        }
        uint4 hi, lo;
        split8(v, hi, lo);
        char* d = sP + (xdst[u] & 0xffff);
        *(uint4*)d = hi;
        *(uint4*)(d + LO) = lo;
      }
    }
#pragma unroll
    for (int u = 0; u < NY; ++u) {
      if (has_y(u) && ydst[u] >= 0) {
        const float v[8] = {ya_[u].x, ya_[u].y, ya_[u].z, ya_[u].w, yb_[u].x, yb_[u].y, yb_[u].z, yb_[u].w};
        uint4 hi, lo;
        split8(v, hi, lo);
        char* d = sY + ydst[u];
        *(uint4*)d = hi;
        *(uint4*)(d + LO) = lo;
      }
    }
  };

  const int box_beg = min(g.nboxes, ((TEAMS ? 2 : 1) * (int)blockIdx.x + team) * g.boxes_per_wg);
  const int box_end = min(g.nboxes, box_beg + g.boxes_per_wg);
  // experiment (MD_W2_STAGGER = dbg >> 8, units of 512 cycles): delay the second half of the grid -- the workgroups that join an
  // already occupied CU -- so that the two workgroups of a CU do not run their matrix phases in step
  if (!TEAMS && (dbg >> 8) && blockIdx.x >= gridDim.x / 2) for (int i = 0; i < (dbg >> 8); ++i) __builtin_amdgcn_s_sleep(8);
  if (box_beg < box_end) issue();
  for (int box = box_beg; box < box_end; ++box) {
    sync();                   // previous box fully consumed (first iteration: tables / scale in LDS)
    if (!(dbg & 4)) commit();
    sync();
    if (box + 1 < box_end) issue();      // in flight during the matrix phase
#pragma unroll
    for (int s = 0; s < (((dbg & 0xff) & 2) ? 0 : NS); ++s) {
      // The MFMA's 32 reduction slots of this step are pixels; lane group lg takes pixels {4lg..4lg+3} and
      // {16+4lg..16+4lg+3} of the step (any assignment works as long as A and B agree): a half-wave's first read
      // then covers 8 consecutive pixels, conflict free with the odd-multiple-of-32-byte pixel pitch.
      const int r0 = s * 32 + lg * 4 + lq;
      const char* xa = sP + sRx[r0] + lp * 8; const char* xb2 = sP + sRx[r0 + 16] + lp * 8;
      const char* ya = sY + r0 * g.ypitch + lp * 8; const char* yb2 = sY + (r0 + 16) * g.ypitch + lp * 8;
#pragma unroll
      for (int jb = 0; jb < NREP; jb += JH) {
        bf16x8 bh[JH], bl[JH];
#pragma unroll
        for (int j = 0; j < JH; ++j) {
          if (jb + j < NREP) {
            bh[j] = w2_tr_read2(ya + (jb + j) * 32, yb2 + (jb + j) * 32);
            bl[j] = w2_tr_read2(ya + LO + (jb + j) * 32, yb2 + LO + (jb + j) * 32);
          }
        }
#pragma unroll
        for (int a = 0; a < KTW; ++a) {
          const char* pa = xa + koff[a]; const char* pb = xb2 + koff[a];
          const bf16x8 ah = w2_tr_read2(pa, pb);
          const bf16x8 al = w2_tr_read2(pa + LO, pb + LO);
#pragma unroll
          for (int j = 0; j < JH; ++j) {
            if (jb + j < NREP) {
              acc[a][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[a][jb + j], 0, 0, 0);
              acc[a][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[a][jb + j], 0, 0, 0);
              acc[a][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[a][jb + j], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  if constexpr (TEAMS) {
    // team 1 hands its accumulators to team 0 through LDS (the images are dead): [tile][reg][256 threads], conflict-free
    __syncthreads();
    float* xch = (float*)sm_all;
    if (team == 1) {
#pragma unroll
      for (int a = 0; a < KTW; ++a)
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) xch[((a * NREP + j) * 4 + r) * 256 + t] = acc[a][j][r];
    }
    __syncthreads();
    if (team == 1) return;
#pragma unroll
    for (int a = 0; a < KTW; ++a)
#pragma unroll
      for (int j = 0; j < NREP; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][j][r] += xch[((a * NREP + j) * 4 + r) * 256 + t];
  }
  // ---- slab[slice][(kg * nktg + k-tile) * 16 + row][N16]: D rows = k index (4 * lg + reg), cols = cout (li)
  float* out = slab + (size_t)blockIdx.x * ((size_t)g.nkg * g.nktg * 16 * g.N16);
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int lt = lt0 + a;
    if (lt < g.nktg) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        const int col = n0 + j * 16 + li;
        if (col < g.N16) {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)((kg * g.nktg + lt) * 16 + lg * 4 + r) * g.N16 + col] = acc[a][j][r];
        }
      }
    }
  }
}

// dw[cout][cin][tap] = sum_slices slab[slice][((kg * taps + tap) * KTg + c16) * 16 + cin % 16][cout]   (fixed order)
// Block = 64 outputs x 4 slice groups: slice group q sums slices q, q+4, ... with four independent chains.
__device__ __forceinline__ void wgrad2_reduce_body(const float* __restrict__ slab, int nslices, int rows, int KTg, int taps, int N16,
                                                   int Cout, int Cin, float* __restrict__ dw, int dw_cin, int dw_c0) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;                         // over [rows][N16], cout fastest
  const size_t stride = (size_t)rows * N16;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < rows * N16) {
    int i = q;
    for (; i + 12 < nslices; i += 16) {
      s0 += slab[(size_t)i * stride + idx];
      s1 += slab[(size_t)(i + 4) * stride + idx];
      s2 += slab[(size_t)(i + 8) * stride + idx];
      s3 += slab[(size_t)(i + 12) * stride + idx];
    }
    for (; i < nslices; i += 4) s0 += slab[(size_t)i * stride + idx];
  }
  red[q][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && idx < rows * N16) {
    const float s = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    const int krow = idx / N16, co = idx - krow * N16;
    const int kt = krow >> 4;
    const int nktg = taps * KTg;
    const int kg = kt / nktg; const int lt = kt - kg * nktg;
    const int tap = lt / KTg; const int c = ((kg * KTg) + (lt - tap * KTg)) * 16 + (krow & 15);
    if (co < Cout && c < Cin) {
      if (dw_cin < 0) dw[(size_t)c * Cout + co] = s;       // transposed output (role-swapped Linear, see wgrad2_build): taps == 1
      else dw[((size_t)co * dw_cin + dw_c0 + c) * taps + tap] = s;
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad2_reduce(const float* __restrict__ slab, int nslices, int rows, int KTg, int taps, int N16,
                                                       int Cout, int Cin, float* __restrict__ dw, int dw_cin, int dw_c0) {
  wgrad2_reduce_body(slab, nslices, rows, KTg, taps, N16, Cout, Cin, dw, dw_cin, dw_c0);
}

struct Wgrad2Plan { W2Geom g; size_t lds; int nslices; bool teams; size_t lds_teams; };

static bool wgrad2_build(const MdConvDesc* d_in, W2Geom* out, size_t* lds_bytes, int xpitch = 0, int xc_base = 0, int dw_cin = 0) {
  // A 1x1x1 unit-stride convolution without padding couples no pixels: its N*T*H*W pixels may be walked as any H' x W' grid.  The
  // Linears of the transformer models arrive as one row of `rows` pixels (rows laid along W, models/_unit.py::LinearRowsFunction);
  // they are walked as (rows / w) x w with the divisor w <= 32 that fills the 64-pixel boxes best, so that the box coordinates fit
  // the validity words (and the boxes are not 1 x 64 strips).
  MdConvDesc dd = *d_in;
  if (dd.kt == 1 && dd.kh == 1 && dd.kw == 1 && dd.st == 1 && dd.sh == 1 && dd.sw == 1 && dd.pt == 0 && dd.ph == 0 && dd.pw == 0) {
    const long long rows = (long long)dd.N * dd.Ti * dd.Hi * dd.Wi;
    int best_w = 0, best_fill = 0;
    for (int w = 1; w <= 32; ++w) {
      if (rows % w) continue;
      const int by = 64 / w < 1 ? 1 : 64 / w;
      const long long h = rows / w;
      const int fill = (int)((h < by ? h : by) * w);
      if (fill >= best_fill) { best_fill = fill; best_w = w; }
    }
    if (best_w && rows / best_w < (1ll << 24)) { dd.N = 1; dd.Ti = dd.To = 1; dd.Hi = dd.Ho = (int)(rows / best_w); dd.Wi = dd.Wo = best_w; }
  }
  const MdConvDesc* d = &dd;
  W2Geom g;
  g.st = d->st; g.sh = d->sh; g.sw = d->sw;
  g.Ti = d->Ti; g.Hi = d->Hi; g.Wi = d->Wi; g.Cpi = md_cpad(d->Cin);
  g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.Cpo = md_cpad(d->Cout);
  g.kh = d->kh; g.kw = d->kw; g.khw = d->kh * d->kw; g.taps = d->kt * g.khw;
  g.org_t = -d->pt; g.org_h = -d->ph; g.org_w = -d->pw;
  if (g.taps > 20) return false;
  if (g.Cpi == 4 && d->sw == 2 && (d->Wi & 1) == 0 && d->kw >= 2) return false;      // pixel-pair stem: first form
  g.KT = md_cdiv(d->Cin, 16);
  { int nkg = 1; while (g.taps * md_cdiv(g.KT, nkg) > 20) ++nkg; g.KTg = md_cdiv(g.KT, nkg); g.nkg = md_cdiv(g.KT, g.KTg); }
  g.nktg = g.taps * g.KTg;
  { const int q = md_cdiv(g.nktg, 4); g.ktw = q <= 2 ? 2 : (q > 5 ? 5 : q); }
  g.C8i = 2 * g.KTg;
  g.ppitch = pitch_for(g.C8i);
  g.N16 = md_round_up(d->Cout, 16);
  const int NT = g.N16 / 16;
  g.nng = md_cdiv(NT, 5); g.nrep = md_cdiv(NT, g.nng);
  g.NC = 2 * g.nrep;
  g.ypitch = pitch_for(g.NC);
  g.pmb = 64;
  g.magicC8 = magic_of(g.C8i); g.magicNC = magic_of(g.NC);
  {
    g.xpitch = xpitch ? xpitch : g.Cpi; g.xc_base = xpitch ? xc_base : 0;
    g.dw_cin = xpitch ? dw_cin : d->Cin; g.dw_c0 = xpitch ? xc_base : 0;
    // dw_cin < 0 (whole tensors only): the result is written TRANSPOSED, dw[cin][cout] -- a Linear whose input is wider than the staged
    // channel range (ViViT's FeedForward 1024 -> 128) is computed with the operands' roles swapped (X := dY with 128 channels, dY := X
    // with 1024), which is the [1024][128] transpose of its weight gradient: one launch instead of four channel slices
    if (!xpitch && dw_cin < 0) {
      if (d->kt * d->kh * d->kw != 1) return false;
      g.dw_cin = -1;
    }
    const unsigned long long xb = (unsigned long long)d->N * g.Ti * g.Hi * g.Wi * g.xpitch * 4ull;
    const unsigned long long yb = (unsigned long long)d->N * g.To * g.Ho * g.Wo * g.Cpo * 4ull;
    if (xb >= 0x80000000ull || yb >= 0x80000000ull) return false;      // buffer addressing: 2 GiB per tensor
    g.x_bytes = (unsigned)xb; g.y_bytes = (unsigned)yb;
  }
  const int nx = (g.ktw * g.nrep > 20) ? 3 : 5, ny = 3;
  if (g.pmb * g.NC > ny * 256) return false;
  // LDS of one workgroup (two per CU): [X hi | dY hi | row table | scale, shift] within W2_LO bytes, the lo halves of the two
  // images W2_LO bytes behind their hi halves
  const size_t yimg = (size_t)g.pmb * g.ypitch;
  const size_t fixed = yimg + (size_t)g.pmb * 4 + (size_t)2 * g.C8i * 8 * 4 + 64;
  if (fixed + 1024 > W2_LO) return false;
  long long maxP = (long long)(W2_LO - fixed) / g.ppitch - 1;
  const long long item_cap = (long long)nx * 256 / g.C8i;
  if (maxP > item_cap) maxP = item_cap;
  if (maxP < 1) return false;
  if (!choose_box(g.To, g.Ho, g.Wo, d->kt, d->kh, d->kw, g.st, g.sh, g.sw, 0, (int)maxP, (int)maxP, &g.bt, &g.by, &g.bx, g.pmb))
    return false;
  g.byx = g.by * g.bx;
  g.nbt = md_cdiv(g.To, g.bt); g.nby = md_cdiv(g.Ho, g.by); g.nbx = md_cdiv(g.Wo, g.bx);
  g.pt = (g.bt - 1) * g.st + d->kt; g.py = (g.by - 1) * g.sh + d->kh; g.px = (g.bx - 1) * g.sw + d->kw;
  g.pyx = g.py * g.px; g.P = g.pt * g.pyx;
  if (g.P * g.C8i >= 65536 || g.P * g.ppitch >= 65536) return false;
  if (g.px > 32 || g.py + g.pt > 31 || g.bx > 32 || g.by + g.bt > 31) return false;      // validity bits: one word for x, one for (y, t)
  g.m_pyx = magic_of(g.pyx); g.m_px = magic_of(g.px); g.m_byx = magic_of(g.byx); g.m_bx = magic_of(g.bx);
  g.nboxes = d->N * g.nbt * g.nby * g.nbx;
  g.boxes_per_wg = 1;
  size_t off = ((size_t)g.P * g.ppitch + 15) & ~(size_t)15;
  g.off_y = (int)off; off += yimg;
  g.off_rows = (int)off; off += (size_t)g.pmb * 4;
  off = (off + 15) & ~(size_t)15;
  g.off_scale = (int)off; off += (size_t)2 * g.C8i * 8 * 4;
  if (off > W2_LO) return false;
  off = (size_t)W2_LO + g.off_y + yimg;                                  // end of the dY lo half
  if (getenv("MD_PLAN_PRINT"))
    fprintf(stderr, "wgrad2 %d->%d k%d%d%d s%d%d%d out %dx%dx%d: box %dx%dx%d patch %dx%dx%d=%d C8i=%d KTg=%d nkg=%d nktg=%d ktw=%d nrep=%d nng=%d lds=%zu\n",
            d->Cin, d->Cout, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, g.To, g.Ho, g.Wo, g.bt, g.by, g.bx, g.pt, g.py, g.px, g.P,
            g.C8i, g.KTg, g.nkg, g.nktg, g.ktw, g.nrep, g.nng, off);
  *out = g; *lds_bytes = off;
  return true;
}

const Wgrad2Plan* wgrad2_lookup(const MdConvDesc* d, int beside, int xpitch, int xc0, int dw_cin) {
  static const int dis = getenv("MD_WGRAD2") && atoi(getenv("MD_WGRAD2")) == 0;
  if (dis) return nullptr;
  static std::mutex mu;
  static std::map<std::array<int, 22>, Wgrad2Plan*> cache;
  std::array<int, 22> key = {d->N, d->Ti, d->Hi, d->Wi, d->Cin, d->To, d->Ho, d->Wo, d->Cout, d->kt, d->kh, d->kw,
                             d->st, d->sh, d->sw, d->pt, d->ph, d->pw, beside ? 1 : 0, xpitch, xc0, dw_cin};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  Wgrad2Plan* wp = nullptr;
  W2Geom g; size_t lds = 0;
  if (wgrad2_build(d, &g, &lds, xpitch, xc0, dw_cin)) {
    // one slab of partial sums per workgroup: two resident workgroups per CU, the whole chip whatever the executor's schedule
    // (the plan, and with it the summation order, does not depend on the schedule: both give the same bits)
    static const int fill_env = getenv("MD_WGRAD_FILL") ? atoi(getenv("MD_WGRAD_FILL")) : 0;
    const int fill = fill_env ? fill_env : 256;
    (void)beside;
    int want = md_cdiv(fill * 2, g.nkg * g.nng);
    if (want > g.nboxes) want = g.nboxes;
    if (want < 1) want = 1;
    // Every slice writes a full copy of the unit's dW (read back by the reduction): in the deep layers, where a slice would hold one
    // or two boxes, that traffic exceeds the operands' several times over (64 -> 144 at 16x16: 32 MB of slabs, 10 MB of operands).
    // There a workgroup takes at least `minb` boxes as long as `minw` workgroups remain: time-neutral (profiles/r03_wgrad2.txt),
    // 0.4 GB less HBM traffic per step.
    // (end of round 3: 2 boxes instead of 4 -- together with MD_PATCH_HALF=0 the bench went 1358-1368 -> 1380-1384 clips/s, interleaved)
    static const int minb = getenv("MD_W2_MIN_BOXES") ? atoi(getenv("MD_W2_MIN_BOXES")) : 2;
    static const int minw = getenv("MD_W2_MIN_WGS") ? atoi(getenv("MD_W2_MIN_WGS")) : 128;
    if (minb > 1 && g.nboxes / want < minb) {
      int w2 = g.nboxes / minb;
      const int floor_w = md_cdiv(minw, g.nkg * g.nng);
      if (w2 < floor_w) w2 = floor_w;
      if (w2 < 1) w2 = 1;
      if (w2 < want) want = w2;
    }
    g.boxes_per_wg = md_cdiv(g.nboxes, want);
    wp = new Wgrad2Plan(); wp->g = g; wp->lds = lds; wp->nslices = md_cdiv(g.nboxes, g.boxes_per_wg);
    // two-team form: the hi halves of both teams' images (and the two counters) within W2_LO_T bytes
    static const int teams_env = getenv("MD_W2_TEAMS") ? atoi(getenv("MD_W2_TEAMS")) : 1;
    const size_t hi_part = (size_t)g.off_scale + (size_t)2 * g.C8i * 8 * 4;      // [X hi | dY hi | row table | scale, shift] of one team
    wp->teams = teams_env && hi_part + 256 <= W2_TS && wp->nslices >= 2;
    if (wp->teams) {
      wp->nslices = md_cdiv(wp->nslices, 2);                 // one slab per workgroup = per pair of box ranges
      const size_t xch = (size_t)g.ktw * g.nrep * 4 * 256 * 4;
      wp->lds_teams = (size_t)W2_LO_T + W2_TS + hi_part;
      if (wp->lds_teams < xch) wp->lds_teams = xch;
    }
  }
  cache[key] = wp;
  return wp;
}

size_t wgrad2_workspace_floats(const Wgrad2Plan* p) { return (size_t)p->nslices * p->g.nkg * p->g.nktg * 16 * p->g.N16; }

template <int KTW, int NREP>
static int wgrad2_launch_one(const Wgrad2Plan* p, const float* src, const float* ps, const float* psh, float slope, const float* dy,
                             float* slab, hipStream_t s) {
  const W2Geom& g = p->g;
  if (p->teams) {
    static bool sett_ = false;
    if (!sett_) {
      if (hipFuncSetAttribute((const void*)k_wgrad2<KTW, NREP, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return MD_ERR_LAUNCH;
      sett_ = true;
    }
    static const int dbgt = getenv("MD_DBG2") ? atoi(getenv("MD_DBG2")) : 0;
    MD_KLAUNCH((k_wgrad2<KTW, NREP, 2, true>), dim3(p->nslices, g.nkg * g.nng), dim3(512), p->lds_teams, s, g, src, ps, psh, slope, dy, slab,
               dbgt & 0xff);
    MD_CHECK_LAUNCH();
    return MD_OK;
  }
  static bool set_ = false;
  if (!set_) {
    if (hipFuncSetAttribute((const void*)k_wgrad2<KTW, NREP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return MD_ERR_LAUNCH;
    set_ = true;
  }
  dim3 grid(p->nslices, g.nkg * g.nng);
  static const int dbg = (getenv("MD_DBG2") ? atoi(getenv("MD_DBG2")) : 0) | ((getenv("MD_W2_STAGGER") ? atoi(getenv("MD_W2_STAGGER")) : 0) << 8);
  static const size_t pad = getenv("MD_W2_PAD_KB") ? (size_t)atoi(getenv("MD_W2_PAD_KB")) * 1024 : 0;      // experiments: force fewer workgroups per CU
  MD_KLAUNCH((k_wgrad2<KTW, NREP, 2>), grid, dim3(256), p->lds + pad, s, g, src, ps, psh, slope, dy, slab, dbg);
  MD_CHECK_LAUNCH();
  return MD_OK;
}
template <int KTW>
static int wgrad2_launch_nr(const Wgrad2Plan* p, const float* src, const float* ps, const float* psh, float slope, const float* dy,
                            float* slab, hipStream_t s) {
  switch (p->g.nrep) {
    case 1: return wgrad2_launch_one<KTW, 1>(p, src, ps, psh, slope, dy, slab, s);
    case 2: return wgrad2_launch_one<KTW, 2>(p, src, ps, psh, slope, dy, slab, s);
    case 3: return wgrad2_launch_one<KTW, 3>(p, src, ps, psh, slope, dy, slab, s);
    case 4: return wgrad2_launch_one<KTW, 4>(p, src, ps, psh, slope, dy, slab, s);
    default: return wgrad2_launch_one<KTW, 5>(p, src, ps, psh, slope, dy, slab, s);
  }
}

int wgrad2_launch_partial(const Wgrad2Plan* p, const float* src, const float* ps, const float* psh, float slope, const float* dy,
                          float* slab, hipStream_t s) {
  switch (p->g.ktw) {
    case 2: return wgrad2_launch_nr<2>(p, src, ps, psh, slope, dy, slab, s);
    case 3: return wgrad2_launch_nr<3>(p, src, ps, psh, slope, dy, slab, s);
    case 4: return wgrad2_launch_nr<4>(p, src, ps, psh, slope, dy, slab, s);
    default: return wgrad2_launch_nr<5>(p, src, ps, psh, slope, dy, slab, s);
  }
}

int wgrad2_launch(const Wgrad2Plan* p, const MdConvDesc* d, const float* src, const float* ps, const float* psh, float slope,
                  const float* dy, float* dw, float* slab, hipStream_t s) {
  const W2Geom& g = p->g;
  const int rc = wgrad2_launch_partial(p, src, ps, psh, slope, dy, slab, s);
  if (rc != MD_OK) return rc;
  const int rows = g.nkg * g.nktg * 16;
  MD_KLAUNCH(k_wgrad2_reduce, dim3(md_cdiv(rows * g.N16, 64)), dim3(256), 0, s, slab, p->nslices, rows, g.KTg, g.taps, g.N16, d->Cout,
             d->Cin, dw, g.dw_cin, g.dw_c0);
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// ---- the slab reductions of several weight gradients in one launch (blockIdx.y = which one); same sums, same order
#define W2_RB_MAX 40
struct W2RedItem { const float* slab; float* dw; int nslices, rows, KTg, taps, N16, Cout, Cin, dw_cin, dw_c0, pad; };
struct W2RedBatch { W2RedItem it[W2_RB_MAX]; };
__global__ __launch_bounds__(256) void k_wgrad2_reduce_batch(W2RedBatch b) {
  const W2RedItem& q = b.it[blockIdx.y];
  if ((int)blockIdx.x * 64 >= q.rows * q.N16) return;
  wgrad2_reduce_body(q.slab, q.nslices, q.rows, q.KTg, q.taps, q.N16, q.Cout, q.Cin, q.dw, q.dw_cin, q.dw_c0);
}
int wgrad2_reduce_batch(int n, const WgradPending* items, hipStream_t s) {
  for (int base = 0; base < n; base += W2_RB_MAX) {
    W2RedBatch b; int cnt = 0, maxb = 0;
    for (int i = base; i < n && cnt < W2_RB_MAX; ++i, ++cnt) {
      const W2Geom& g = items[i].p->g;
      W2RedItem& q = b.it[cnt];
      q.slab = items[i].slab; q.dw = items[i].dw; q.nslices = items[i].p->nslices; q.rows = g.nkg * g.nktg * 16; q.KTg = g.KTg;
      q.taps = g.taps; q.N16 = g.N16; q.Cout = items[i].Cout; q.Cin = items[i].Cin; q.dw_cin = g.dw_cin; q.dw_c0 = g.dw_c0; q.pad = 0;
      const int nb = md_cdiv(q.rows * q.N16, 64);
      if (nb > maxb) maxb = nb;
    }
    MD_KLAUNCH(k_wgrad2_reduce_batch, dim3(maxb, cnt), dim3(256), 0, s, b);
    MD_CHECK_LAUNCH();
  }
  return MD_OK;
}
