// Persistent, two-team form of the LDS-patch convolution (forward and data gradient) for the layers whose packed
// weights fit in LDS next to two input patches.  Same arithmetic as k_conv_patch (conv_patch.hip): fp32 operands split
// into two 16-bit halves, three MFMAs per product, fp32 accumulation, the same accumulation order over K per output
// element; only the BatchNorm partial sums are grouped differently (per workgroup instead of per box).
//
// k_conv_patch runs two independent 4-wave workgroups per CU so that one can stage while the other multiplies, and pays
// for it: every workgroup re-streams the packed weights through LDS (two barriers per 64-deep K stage) and rebuilds its
// tables per box; measured on the 64x64-resolution layers, a third of the time is neither loads, MFMAs nor stores.
// Here ONE 8-wave workgroup per CU keeps the weights resident and splits into two TEAMS of four waves that share them:
//   * each team walks its own stream of boxes (team k of workgroup b: boxes 2b + k, + 2 gridDim.x, ...) with its own
//     patch buffer; a team's wave owns 32 pixel rows of the 128-row box and ALL channel tiles (as a k_conv_patch wave);
//   * the teams run the same two-phase loop, one phase apart.  MATRIX phase: request the next box's patch from HBM
//     (registers), then the K loop -- patch and weight fragments from LDS, no barriers inside.  MOVE phase: commit the
//     requested patch (BatchNorm + LeakyReLU on read, hi/lo split, LDS write), then store the finished box.  A
//     workgroup barrier closes every phase, so on each SIMD one wave is always in its MATRIX phase while its partner
//     (the other team's wave) is in its MOVE phase: staging VALU, HBM latency and stores hide behind the MFMAs;
//   * tap offsets, row offsets, per-item patch decodes and BatchNorm constants are computed once per workgroup;
//   * the MFMAs take the WEIGHT fragment as the A operand: the accumulator tile is D[channel][pixel], a lane holds four
//     consecutive channels of one pixel, and the epilogue moves 16 bytes per instruction;
//   * BatchNorm partial sums stay in registers across boxes: one partial row per workgroup (256 rows for the finalize
//     kernel instead of one per box);
//   * data-gradient launches can fuse the BatchNorm-backward REDUCTION of the unit they feed: the epilogue reads that
//     unit's raw output at the pixels it has just produced, stores g = dA * leaky'(bn(y)) instead of dA and
//     accumulates sum(g), sum(g * xhat) per channel (backward of src/models/R2Plus1D.py:53-58; replaces one
//     md_bn_bwd_reduce pass over the tensor).
#include "patch_common.h"
#include <cstdlib>
#include <cstdio>
#include <type_traits>

// 16-byte buffer store / load at a byte offset (out-of-range offset: dropped / zeros)
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ f32x4 buf_load4v(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
// Prefetch load: requested before the matrix loop, consumed a phase later.  (A volatile load -- aux bit 31 -- would pin it
// in place but also makes it system-coherent, sc0 sc1, i.e. the halo re-reads of neighbouring boxes would miss L2.)
__device__ __forceinline__ f32x4 buf_load4_pinned(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// MAXI: 32-byte patch items per thread held in registers between a box's request and its commit (4 or 8)
template <bool F16, bool FUSE, int NREP, int MAXI>
__global__ __launch_bounds__(512) void k_conv_pers(
    PersGeom pg, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const uint4* __restrict__ wp, float* __restrict__ dst, float* __restrict__ stat_partial,
    int accumulate, PersBwd bw) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const PGeom& g = pg.g;
  constexpr int NT = 512, TT = 256;                 // threads per workgroup / per team
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int team = wave >> 2, wr = wave & 3;
  const int tt = t & (TT - 1);
  const int lane = t & 63, li = lane & 15, lg = lane >> 4;
  const bool prologue = pscale != nullptr;
  const bool stats = stat_partial != nullptr;
  const int dbg = (accumulate >> 8) & 0xff;         // timing experiments (MD_DBG): 1 no patch loads, 2 no matrix loop, 4 no stores, 8 no commit
  const bool xcd_order = (accumulate >> 18) & 1;
  const bool early_issue = (accumulate >> 19) & 1;
  const bool presplit = (accumulate >> 16) & 1;      // source = pre-split bf16 gradient [pixel][C8]{hi | lo}: the commit is a plain copy
  accumulate &= 1;

  // ---- tables (once per workgroup)
  int* sK = (int*)(sm + pg.off_k);
  int2* sRow = (int2*)(sm + pg.off_row);
  float* sScale = (float*)(sm + pg.off_sc);
  float* sShift = sScale + pg.sc_stride;
  float* sBn = (float*)(sm + pg.off_bn);            // [scale | shift | mean | invstd][bn_stride] of the fused unit
  const int bns = pg.bn_stride;
  for (int q = t; q < pg.nsteps * 4; q += NT) {
    int ko = 0;                                     // K padding reads patch offset 0 (finite data) against zero weights
    if (q < g.Kc8) {
      const int tap = mdiv(q, g.magicC8); const int c8 = q - tap * g.C8;
      const int dt = mdiv(tap, g.m_khw); const int r = tap - dt * g.khw;
      const int dy = mdiv(r, g.m_kw); const int dx = r - dy * g.kw;
      ko = ((dt * g.py + dy) * g.px + dx) * g.ppitch + c8 * 16;
    }
    sK[q] = ko;
  }
  if (t < 32) ((int*)(sm + pg.off_sync))[t] = 0;
  if (t < PM) {
    const int rt = mdiv(t, g.m_byx); const int r = t - rt * g.byx;
    const int ry = mdiv(r, g.m_bx); const int rx = r - ry * g.bx;
    const bool v = rt < g.bt;
    int2 ri;
    ri.x = v ? ((rt * g.st * g.py + ry * g.sh) * g.px + rx * g.sw) * g.ppitch : 0;
    ri.y = v ? (rt | (ry << 8) | (rx << 16)) : -1;
    sRow[t] = ri;
  }
  // BatchNorm-on-read constants; zero for the channel padding up to whole 8-channel chunks, so that the commit needs no
  // per-element "does this channel exist" test (a padded channel is 0 in memory and stays leaky(0 * 0 + 0) = 0)
  if (prologue) for (int c = t; c < pg.sc_stride; c += NT) {
    const int cs = g.pack2 ? (c & 3) : c;
    sScale[c] = c < g.Cps ? pscale[cs] : 0.f; sShift[c] = c < g.Cps ? pshift[cs] : 0.f;
  }
  if (FUSE) {
    for (int c = t; c < g.N16; c += NT) {
      const int cg = pg.n0 + c;
      const bool ok = cg < g.Cpd;
      sBn[c] = ok ? bw.scale[cg] : 0.f; sBn[bns + c] = ok ? bw.shift[cg] : 0.f;
      sBn[2 * bns + c] = ok ? bw.mean[cg] : 0.f; sBn[3 * bns + c] = ok ? bw.invstd[cg] : 0.f;
    }
  }
  // resident weights: global [stage][hi|lo][N16][8 chunks] -> LDS [hi|lo][N16][bpitch]
  {
    const int total = g.nstages * 2 * g.N16 * 8;
    const int kchunks = pg.nsteps * 4;
    for (int idx = t; idx < total; idx += NT) {
      const int c = idx & 7; const int rest = idx >> 3;
      const int hb = rest / g.N16; const int n = rest - hb * g.N16;
      const int h = hb & 1, kb = hb >> 1;
      const int kc = kb * 8 + c;
      // (column slice: row n0 + n of the N16w packed rows; the whole operand when n0 = 0 and N16w = N16)
      if (kc < kchunks) *(uint4*)(sm + pg.off_bres + (h * g.N16 + n) * pg.bpitch + kc * 16) = wp[((size_t)hb * pg.N16w + pg.n0 + n) * 8 + c];
    }
  }
  __syncthreads();

  // ---- this team
  char* sP = sm + team * pg.patch_bytes;            // the team's patch: hi | lo
  const char* sB = sm + pg.off_bres;

  // box-independent per-thread staging items.  it_lds: LDS byte offset (-1: none); it_rel: element offset from the patch
  // origin; it_pd: ppt | ppy << 6 | ppx << 15 | c8 << 24 | (lower half exists) << 30 | (upper half exists) << 31
  int it_lds[MAXI], it_rel[MAXI]; unsigned it_pd[MAXI];
  const int cvalid4 = g.Cps >> 2;
#pragma unroll
  for (int u = 0; u < MAXI; ++u) {
    const int item = u * TT + tt;
    it_lds[u] = -1; it_rel[u] = 0; it_pd[u] = 0;
    if (u < pg.nit && item < g.P * g.C8) {
      const int pixel = mdiv(item, g.magicC8);
      const int c8 = item - pixel * g.C8;
      const int ppt = mdiv(pixel, g.m_pyx); const int r = pixel - ppt * g.pyx;
      const int ppy = mdiv(r, g.m_px); const int ppx = r - ppy * g.px;
      it_lds[u] = pixel * g.ppitch + c8 * 16;
      it_rel[u] = presplit ? (((ppt * g.Hs + ppy) * g.Ws + ppx) * g.C8 + c8) * 8 : ((ppt * g.Hs + ppy) * g.Ws + ppx) * g.Cps + c8 * 8;
      it_pd[u] = (unsigned)ppt | ((unsigned)ppy << 6) | ((unsigned)ppx << 15) | ((unsigned)c8 << 24) |
                 ((c8 * 2 < cvalid4 ? 1u : 0u) << 30) | ((c8 * 2 + 1 < cvalid4 ? 1u : 0u) << 31);
    }
  }
  // this lane's two pixel rows: patch offset, packed box coordinate, destination pixel offset
  const int rp0 = sRow[wr * 32 + li].x, rp1 = sRow[wr * 32 + 16 + li].x;
  int rdec[2], drel[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int pk = sRow[wr * 32 + a * 16 + li].y;                   // rt | ry << 8 | rx << 16, or -1 outside the box
    rdec[a] = pk;
    const int rt = pk & 0xff, ry = (pk >> 8) & 0xff, rx = (pk >> 16) & 0xff;
    drel[a] = pk < 0 ? 0 : ((rt * g.dmt * g.Hdf) + ry * g.dmh) * g.Wdf + rx * g.dmw;
  }
  const int c0 = lg * 4;                                            // first of this lane's 4 channels in tile 0
  f32x4 s1[NREP], s2[NREP];
#pragma unroll
  for (int j = 0; j < NREP; ++j) { s1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[j] = s1[j]; }

  const __amdgpu_buffer_rsrc_t srs = make_rsrc(src, presplit ? (unsigned)(g.src_bytes / (g.Cps * 4u)) * (unsigned)g.C8 * 32u : g.src_bytes);
  const __amdgpu_buffer_rsrc_t drs = make_rsrc(dst, g.dst_bytes);
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(FUSE ? bw.yraw : dst, g.dst_bytes);

  f32x4 va[MAXI], vb[MAXI];
  unsigned inmask = 0;                        // bit u: item u lies inside the tensor (BatchNorm-on-read applies)
  auto issue = [&](int box) {
    int b = box;
    const int q1 = mdiv(b, pg.m_nbx); const int xb = b - q1 * g.nbx; b = q1;
    const int q2 = mdiv(b, pg.m_nby); const int yb = b - q2 * g.nby; b = q2;
    const int n = mdiv(b, pg.m_nbt); const int tb = b - n * g.nbt;
    const int ot = tb * g.bt * g.st + g.org_t, oh = yb * g.by * g.sh + g.org_h, ow = xb * g.bx * g.sw + g.org_w;
    const int qbase = (((n * g.Ts + ot) * g.Hs + oh) * g.Ws + ow) * (presplit ? g.C8 * 8 : g.Cps);
    inmask = 0;
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      if (u < pg.nit) {
        const unsigned pd = it_pd[u];
        const int st_ = ot + (int)(pd & 63u), sy_ = oh + (int)((pd >> 6) & 511u), sx_ = ow + (int)((pd >> 15) & 511u);
        const bool in = it_lds[u] >= 0 && ((unsigned)st_ < (unsigned)g.Ts) && ((unsigned)sy_ < (unsigned)g.Hs) &&
                        ((unsigned)sx_ < (unsigned)g.Ws) && ((pd >> 30) & 1u) && !(dbg & 1);
        const unsigned off = (unsigned)(qbase + it_rel[u]) * 4u;
        va[u] = buf_load4_pinned(srs, in ? off : MD_OOB);
        vb[u] = buf_load4_pinned(srs, (in && ((pd >> 31) || presplit)) ? off + 16u : MD_OOB);
        inmask |= (in ? 1u : 0u) << u;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
      if (u < pg.nit && it_lds[u] >= 0) {
        float v[8] = {va[u][0], va[u][1], va[u][2], va[u][3], vb[u][0], vb[u][1], vb[u][2], vb[u][3]};
        if (prologue) {
          const int c8 = (int)((it_pd[u] >> 24) & 63u);
          const f32x4 sc0 = *(const f32x4*)(sScale + c8 * 8), sc1 = *(const f32x4*)(sScale + c8 * 8 + 4);
          const f32x4 sh0 = *(const f32x4*)(sShift + c8 * 8), sh1 = *(const f32x4*)(sShift + c8 * 8 + 4);
          bn_leaky8(v, sc0, sc1, sh0, sh1, pslope);
          if (!((inmask >> u) & 1u)) zero8(v);                         // zero padding stays zero after the activation
        }
        uint4 hi, lo;
        if (presplit) { hi = __builtin_bit_cast(uint4, va[u]); lo = __builtin_bit_cast(uint4, vb[u]); }
        else if (F16) split8_f16(v, hi, lo); else split8(v, hi, lo);
        *(uint4*)(sP + it_lds[u]) = hi;
        *(uint4*)(sP + g.lo_off + it_lds[u]) = lo;
      }
    }
  };

  const int blo = g.N16 * pg.bpitch;                                   // lo half of the resident weights
  const char* bbase = sB + li * pg.bpitch + lg * 16;
  auto load_p = [&](int ko, uint4* f) {                                // patch fragments of both pixel slabs: hi0, lo0, hi1, lo1
    f[0] = *(const uint4*)(sP + rp0 + ko); f[1] = *(const uint4*)(sP + g.lo_off + rp0 + ko);
    f[2] = *(const uint4*)(sP + rp1 + ko); f[3] = *(const uint4*)(sP + g.lo_off + rp1 + ko);
  };

  // state carried from a box's MATRIX phase to its MOVE phase
  f32x4 acc[2][NREP];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned goff[2] = {MD_OOB, MD_OOB}; float gw[2] = {0.f, 0.f};

  const int stride = 2 * (int)gridDim.x;
  // XCD-aware order (accumulate bit 18): workgroup w runs on XCD w % 8, each with its own L2.  With the plain order the boxes in
  // flight at any time are spread round-robin over the XCDs, so the halo a box shares with its neighbours is fetched into several
  // L2s; with wid below, XCD x walks the contiguous range of workgroup indices [x G/8, (x+1) G/8).
  const int wid = xcd_order && (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  int cur = 2 * wid + team;                                            // box whose patch is (or is about to be) in LDS
  // boxes of team 0 (>= those of team 1): both teams run that many iterations so that the barriers pair up
  const int first0 = 2 * wid;
  const int niter = first0 < pg.nboxes ? (pg.nboxes - first0 + stride - 1) / stride : 0;
  if (cur < pg.nboxes) { issue(cur); commit(); }
  if (early_issue && cur + stride < pg.nboxes) issue(cur + stride);     // (early_issue: see do_move)
  __syncthreads();                                                     // both teams' first patches complete

  auto do_matrix = [&]() __attribute__((always_inline)) {
    // ================= MATRIX phase of box `cur`
    const int nxt = cur + stride;
    if (!early_issue && nxt < pg.nboxes && !(dbg & 16)) issue(nxt);  // in flight during the matrix loop
    if (cur < pg.nboxes && !(dbg & 64)) {
      // destination pixels of this box (byte offsets of channel c0; pixels outside the tensor: out-of-range offset)
      {
        int b = cur;
        const int q1 = mdiv(b, pg.m_nbx); const int xb = b - q1 * g.nbx; b = q1;
        const int q2 = mdiv(b, pg.m_nby); const int yb = b - q2 * g.nby; b = q2;
        const int n = mdiv(b, pg.m_nbt); const int tb = b - n * g.nbt;
        const int t0 = tb * g.bt, y0 = yb * g.by, x0 = xb * g.bx;
        const int dbase = ((n * g.Tdf + t0 * g.dmt + g.dpt) * g.Hdf + y0 * g.dmh + g.dph) * g.Wdf + x0 * g.dmw + g.dpw;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int pk = rdec[a];
          const bool v = pk >= 0 && (t0 + (pk & 0xff) < g.Td) && (y0 + ((pk >> 8) & 0xff) < g.Hd) && (x0 + ((pk >> 16) & 0xff) < g.Wd);
          goff[a] = v ? (unsigned)((dbase + drel[a]) * g.Cpd + pg.n0 + c0) * 4u : MD_OOB;
          gw[a] = v ? 1.f : 0.f;
        }
      }
      // matrix loop: patch fragments and resident weight fragments from LDS; no barriers.  The first K step starts every
      // accumulator chain from a literal zero C operand instead of 8 * NREP register clears per box (the kernel is bound by
      // vector-instruction issue: 4 cycles per VALU instruction beside the MFMAs, profiles/r03_pers_issue_model.txt).
      uint4 fa[2][4];
      load_p(sK[lg], fa[0]);
      int ko_next = sK[min(1, pg.nsteps - 1) * 4 + lg];            // tap offset of step q + 1 while step q runs
      // weight fragments stream through registers two channel tiles ahead of their use, across step boundaries as well:
      // one tile's six MFMAs (96 cycles) do not cover an LDS read under load, two tiles' do
      auto load_w = [&](const char* bq, int j, uint4* f) __attribute__((always_inline)) {
        const char* bp = bq + j * 16 * pg.bpitch;
        f[0] = *(const uint4*)bp; f[1] = *(const uint4*)(bp + blo);
      };
      uint4 w0[2], w1[2];
      load_w(bbase, 0, w0); load_w(bbase, 1 % NREP, w1);
      auto kstep = [&](auto first_tag, const uint4* fcur, uint4* fnext, int q) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        // patch fragments of the next step (the last step re-reads its own: branch-free), tap offset of the one after
        load_p(ko_next, fnext);
        ko_next = sK[min(q + 2, pg.nsteps - 1) * 4 + lg];
        __builtin_amdgcn_sched_barrier(0);
        const char* bq = bbase + q * 64;
        const char* bqn = bbase + min(q + 1, pg.nsteps - 1) * 64;
        const uint4 ph0 = fcur[0], pl0 = fcur[1], ph1 = fcur[2], pl1 = fcur[3];
        uint4 wc[2] = {w0[0], w0[1]}, wn[2] = {w1[0], w1[1]};
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          uint4 wp2[2];
          if (j + 2 < NREP) load_w(bq, j + 2, wp2); else load_w(bqn, (j + 2 - NREP) % NREP, wp2);
          __builtin_amdgcn_sched_barrier(0);            // keep the reads HERE: the scheduler would sink them to their use
          const uint4 wh = wc[0], wl = wc[1];
          // smallest terms first: lo*hi and hi*lo, then hi*hi.  (weights, patch) operand order: D[channel][pixel]
          acc[0][j] = mma<F16>(wh, pl0, FIRST ? zero : acc[0][j]);
          acc[1][j] = mma<F16>(wh, pl1, FIRST ? zero : acc[1][j]);
          acc[0][j] = mma<F16>(wl, ph0, acc[0][j]);
          acc[1][j] = mma<F16>(wl, ph1, acc[1][j]);
          acc[0][j] = mma<F16>(wh, ph0, acc[0][j]);
          acc[1][j] = mma<F16>(wh, ph1, acc[1][j]);
          __builtin_amdgcn_sched_barrier(0);
          wc[0] = wn[0]; wc[1] = wn[1]; wn[0] = wp2[0]; wn[1] = wp2[1];
        }
        w0[0] = wc[0]; w0[1] = wc[1]; w1[0] = wn[0]; w1[1] = wn[1];
      };
      if (dbg & 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < NREP; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        typedef std::integral_constant<bool, true> First;
        typedef std::integral_constant<bool, false> Later;
        kstep(First(), fa[0], fa[1], 0);
        int q = 1;
        for (; q + 1 < pg.nsteps; q += 2) { kstep(Later(), fa[1], fa[0], q); kstep(Later(), fa[0], fa[1], q + 1); }
        if (q < pg.nsteps) kstep(Later(), fa[1], fa[0], q);
      }
    }
  };
  auto do_move = [&]() __attribute__((always_inline)) {
    // ================= MOVE phase: next box's patch into LDS, then this box out to HBM
    const int nxt = cur + stride;
    // raw output of the differentiated unit at this box's pixels (fused BatchNorm-backward reduction): tiles are requested
    // YD ahead of their use -- the first YD before the commit, the rest as the epilogue walks the tiles -- so that at most
    // YD + 1 tiles of y are live next to the accumulators
    constexpr int YD = NREP <= 3 ? NREP : 2;
    f32x4 yv[FUSE ? NREP : 1][2];
    auto request_y = [&](int j) __attribute__((always_inline)) {
      const bool colok = pg.n0 + c0 + j * 16 < g.Cpd && cur < pg.nboxes;
#pragma unroll
      for (int a = 0; a < 2; ++a) yv[FUSE ? j : 0][a] = buf_load4_pinned(yrs, colok ? goff[a] + j * 64u : MD_OOB);
    };
    if constexpr (FUSE) {
#pragma unroll
      for (int j = 0; j < YD; ++j) request_y(j);
    }
    if (nxt < pg.nboxes && !(dbg & 8)) commit();
    // early_issue (accumulate bit 19): the box after next is requested as soon as the commit has freed the staging registers --
    // in flight during this epilogue, the hand-over and the whole next matrix phase instead of the matrix phase alone
    if (early_issue && nxt + stride < pg.nboxes) issue(nxt + stride);
    if (cur < pg.nboxes && !(dbg & 32)) {
      const bool allin = __all(gw[0] != 0.f && gw[1] != 0.f);
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        if constexpr (FUSE) { if (j + YD < NREP) { request_y(j + YD); __builtin_amdgcn_sched_barrier(0); } }
        const bool colok = pg.n0 + c0 + j * 16 < g.Cpd;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const unsigned off = (colok && !(dbg & 4)) ? goff[a] + j * 64u : MD_OOB;
          f32x4 v = acc[a][j];
          if (accumulate) v += buf_load4v(drs, off);
          if constexpr (FUSE) {
            const f32x4 sc = *(const f32x4*)(sBn + c0 + j * 16), sh = *(const f32x4*)(sBn + bns + c0 + j * 16);
            const f32x4 mu = *(const f32x4*)(sBn + 2 * bns + c0 + j * 16), is = *(const f32x4*)(sBn + 3 * bns + c0 + j * 16);
            const f32x4 y = yv[j][a];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pre = fmaf(y[r], sc[r], sh[r]);
              v[r] *= md_dleaky(pre, bw.slope);                    // g = dA * leaky'(bn(y))
              const float gm = v[r] * gw[a];
              s1[j][r] += gm; s2[j][r] = fmaf(gm, (y[r] - mu[r]) * is[r], s2[j][r]);
            }
          } else if (stats) {
            if (allin) {        // every pixel row of this wave lies inside the tensor (wave-uniform): no row weights
#pragma unroll
              for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] = fmaf(v[r], v[r], s2[j][r]); }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float vm = v[r] * gw[a]; s1[j][r] += vm; s2[j][r] = fmaf(vm, v[r], s2[j][r]); }
            }
          }
          buf_store4(drs, off, v);
        }
      }
    }
  };

#ifdef MD_PERS_LOCKSTEP
  {
    // lock-step schedule: team 1 runs one phase behind team 0; a workgroup barrier closes every phase
    for (int s = 0; s < 2 * niter + 1; ++s) {
      const int ph = s - team;
      if (ph >= 0 && ph < 2 * niter) {
        if ((ph & 1) == 0) do_matrix();
        else { do_move(); cur += stride; }
      }
      __syncthreads();                       // closes the phase: patches written / read by either team are settled
    }
  }
#else
  {
    // decoupled schedule: the weights are read-only and each team owns its patch buffer, so the only hand-offs are inside
    // a team (patch written by all four waves -> read by all four; read by all -> overwritten).  A four-wave barrier on an
    // LDS counter replaces the workgroup barrier; the teams drift freely, so neither waits for the other's longer phase.
    int* ctr = (int*)(sm + pg.off_sync) + team * 16;
    int target = 0;
    auto team_sync = [&]() {
      target += 4;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    while (cur < pg.nboxes) {
      do_matrix();
      team_sync();                           // every wave of the team has read the patch of `cur`
      do_move();
      team_sync();                           // the patch of the next box is complete
      cur += stride;
    }
    __syncthreads();                         // both teams are done with their patches (the reduction scratch aliases them)
  }
#endif

  // ---- one partial row per workgroup: sum over the 16 pixel lanes, then over the eight waves (fixed order)
  if (stats) {
    float* red = (float*)(sm + pg.off_red);                            // [8 waves][2][PNREP*16]; aliases the patches
#pragma unroll
    for (int j = 0; j < NREP; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a1 = s1[j][r], a2 = s2[j][r];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
        if (li == 0) {
          red[(wave * 2 + 0) * (PNREP * 16) + c0 + j * 16 + r] = a1;
          red[(wave * 2 + 1) * (PNREP * 16) + c0 + j * 16 + r] = a2;
        }
      }
    __syncthreads();
    if (t < g.N16 && pg.n0 + t < g.Cpd) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a1 += red[(w * 2 + 0) * (PNREP * 16) + t];
        a2 += red[(w * 2 + 1) * (PNREP * 16) + t];
      }
      float* sp = stat_partial + (size_t)blockIdx.x * 2 * g.Cpd + pg.n0;
      sp[t] = a1;
      sp[g.Cpd + t] = a2;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int pers_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) n = p.multiProcessorCount;
    else { (void)hipGetLastError(); n = 256; }
  }
  return n;
}

// Completes `pg` (pg.g holds a PGeom built by patch_build with the persistent LDS budget) -- LDS layout, grid.
bool pers_finish(PersGeom* pg, size_t* lds_bytes, int* grid) {
  PGeom& g = pg->g;
  pg->nsteps = md_cdiv(g.Kc8, 4);
  { int u = pg->nsteps * 4; while ((u & 3) != 2) ++u; pg->bpitch = u * 16; }
  pg->nboxes = 0;     // set by the caller (depends on N)
  pg->n0 = 0; pg->N16w = g.N16;
  pg->patch_bytes = (g.P * g.ppitch + 15) & ~15;
  size_t off = (size_t)2 * pg->patch_bytes;
  pg->off_red = 0;                               // the end-of-kernel reduction scratch aliases the patches
  if (off < (size_t)8 * 2 * PNREP * 16 * 4) off = (size_t)8 * 2 * PNREP * 16 * 4;
  pg->off_bres = (int)off; off += (size_t)2 * g.N16 * pg->bpitch;
  pg->off_k = (int)off; off += (size_t)(pg->nsteps * 4 + 4) * 4;
  off = (off + 15) & ~(size_t)15;
  pg->off_row = (int)off; off += (size_t)PM * 8;
  pg->sc_stride = g.C8 * 8; pg->bn_stride = g.N16;
  pg->off_sc = (int)off; off += (size_t)2 * pg->sc_stride * 4;
  pg->off_bn = (int)off; off += (size_t)4 * pg->bn_stride * 4;
  off = (off + 15) & ~(size_t)15;
  pg->off_sync = (int)off; off += 128;             // two team counters, 64 B apart
  if (off > (size_t)(160 * 1024)) return false;
  *lds_bytes = off;
  *grid = pers_num_cus();
  return true;
}

size_t pers_bres_bytes(int Kc8, int N16) {
  int u = md_cdiv(Kc8, 4) * 4; while ((u & 3) != 2) ++u;
  return (size_t)2 * N16 * u * 16;
}
size_t pers_fixed_bytes(int Cps, int N16) { return (size_t)PM * 8 + (size_t)(2 * (Cps + 8) + 4 * N16) * 4 + 256 + 144; }

int pers_launch(const PersGeom& pg, size_t lds, int grid, bool f16, const float* src, const float* ps, const float* psh, float slope,
                const float* wp, float* dst, float* stat, int accumulate, const PersBwd& bw, hipStream_t s) {
  const PGeom& g = pg.g;
  const int nrep = g.N16 / 16;
  const int nblk = pers_blocks(pg, grid);
  static const int dbg = getenv("MD_DBG") ? atoi(getenv("MD_DBG")) : 0;
  static const int xcd = getenv("MD_PERS_XCD") ? atoi(getenv("MD_PERS_XCD")) : 0;
  static const int early = getenv("MD_PERS_EARLY") ? atoi(getenv("MD_PERS_EARLY")) : 0;
  accumulate = (accumulate & 0x10001) | ((dbg & 0xff) << 8) | (xcd ? 0x40000 : 0) | (early ? 0x80000 : 0);
#define LAUNCH_PERS(F16_, FUSE_, NR_, MX_)                                                                              \
  do {                                                                                                                  \
    static bool set_ = false;                                                                                           \
    if (!set_) {                                                                                                        \
      if (hipFuncSetAttribute((const void*)k_conv_pers<F16_, FUSE_, NR_, MX_>,                                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MD_ERR_LAUNCH; \
      set_ = true;                                                                                                      \
    }                                                                                                                   \
    MD_KLAUNCH((k_conv_pers<F16_, FUSE_, NR_, MX_>), dim3(nblk), dim3(512), lds, s, pg, src, ps, psh, slope,            \
               (const uint4*)wp, dst, stat, accumulate, bw);                                                            \
  } while (0)
#define LAUNCH_PERS_NR(F16_, FUSE_)                                                                                     \
  if (pg.nit > 4) {                                                                                                     \
    switch (nrep) {                                                                                                     \
      case 2: LAUNCH_PERS(F16_, FUSE_, 2, 8); break;                                                                    \
      case 3: LAUNCH_PERS(F16_, FUSE_, 3, 8); break;                                                                    \
      default: return MD_ERR_UNSUPPORTED;                                                                               \
    }                                                                                                                   \
  } else {                                                                                                              \
    switch (nrep) {                                                                                                     \
      case 2: LAUNCH_PERS(F16_, FUSE_, 2, 4); break;                                                                    \
      case 3: LAUNCH_PERS(F16_, FUSE_, 3, 4); break;                                                                    \
      case 4: LAUNCH_PERS(F16_, FUSE_, 4, 4); break;                                                                    \
      case 5: LAUNCH_PERS(F16_, FUSE_, 5, 4); break;                                                                    \
      case 6: LAUNCH_PERS(F16_, FUSE_, 6, 4); break;                                                                    \
      default: return MD_ERR_UNSUPPORTED;                                                                               \
    }                                                                                                                   \
  }
  if (f16) { if (bw.yraw) return MD_ERR_UNSUPPORTED; LAUNCH_PERS_NR(true, false); }
  else if (bw.yraw) { LAUNCH_PERS_NR(false, true); }
  else { LAUNCH_PERS_NR(false, false); }
  MD_CHECK_LAUNCH();
  return MD_OK;
}
