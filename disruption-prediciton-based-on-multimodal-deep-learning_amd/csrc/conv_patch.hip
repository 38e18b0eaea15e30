// Unit-stride convolution (forward and data-gradient) with an LDS-resident input patch and split-bf16
// ("bf16x3") matrix-core arithmetic, for gfx950.
//
// Why this kernel exists (measured, profiles/r01a_*): the exact-fp32 MFMA runs at 1/16 of the bf16 rate, and the
// generic gather kernel re-stages every input element once per filter tap.  Here
//   * a workgroup owns a BOX of <=128 output pixels of one clip (bt x by x bx) and stages the box's input patch
//     (with halo) into LDS exactly once -- coalesced 32-byte reads per lane, BatchNorm+LeakyReLU of the producer
//     applied on the way in ("BN-on-read"), zero fill outside the tensor;
//   * every fp32 value x is split as x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product a*b is
//     evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The dropped lo*lo
//     term is < 2^-16 relative: measured 4.5e-6 relative error per conv (fp32 MFMA: 7e-7), well inside the
//     1e-3 parity budget, at 16/3 = 5.3x the exact-fp32 matrix rate.  Weights are split once, at pack time;
//   * the K axis is the flat (tap, 8-channel chunk) axis; the A fragment of lane (row i, chunk group g) is ONE
//     ds_read_b128 from the patch at  rowpix[i] + koffs[chunk]  -- no im2col copy exists anywhere.
//
// LDS pixel pitch is 16*(4m+2) bytes so that the 16 lanes of each ds_read_b128 group cover 16 distinct 16-byte
// slots (conflict-free for rows that are consecutive pixels).
#include "common.h"
#include <cstdlib>

#include "patch_common.h"

// Stage `npix` pixels x `C8` 8-channel chunks of a channels-last fp32 tensor into an LDS image
// [pixel][C8 chunks] (pixel pitch `pitch` bytes; hi array at img, lo array at img + lo_off).
// sG[pixel] = global pixel index or -1 (outside the tensor: zeros = the convolution's zero padding).
// Channels c0 .. c0 + 4*cvalid4 are read (cvalid4 = valid float4 units from c0); chunks past that are zero.
// With `prologue`, value = leaky(x*scale[c] + shift[c]) ("BN-on-read") before the bf16 hi/lo split.
template <bool F16, int NT = 256>
__device__ __forceinline__ void stage_image(__amdgpu_buffer_rsrc_t src, int Cpitch, int c0, int cvalid4,
                                            const int* sG, int npix, int C8, unsigned magic, char* img, int pitch,
                                            int lo_off, bool prologue, const float* sScale, const float* sShift,
                                            float pslope, int t, bool presplit = false) {
  // presplit: the source is a gradient tensor already stored as bf16 hi|lo pairs, [pixel][C8 chunks]{hi 8 x bf16, lo 8 x bf16}
  // (32 bytes per 8 channels, written by md_bn_bwd_apply_fmt): staging is a plain copy, no VALU arithmetic at all.
  const int total = npix * C8;
  for (int base = 0; base < total; base += NT * 4) {
    float4 va[4], vb[4];
    int pix[4], c8s[4];      // pix: pixel | 0x20000000 (outside the tensor: zeros, no prologue); -1 = no item
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int item = min(base + u * NT + t, total - 1);
      const int pixel = mdiv(item, magic);
      const int c8 = item - pixel * C8;
      const int gp = sG[pixel];
      const bool in = gp >= 0 && c8 * 2 < cvalid4;
      const unsigned off = in ? (presplit ? (unsigned)(gp * C8 + c8) * 32u : (unsigned)(gp * Cpitch + c0 + c8 * 8) * 4u) : MD_OOB;
      va[u] = buf_load4(src, off);
      vb[u] = buf_load4(src, (in && (presplit || c8 * 2 + 1 < cvalid4)) ? off + 16u : MD_OOB);
      pix[u] = base + u * NT + t < total ? (pixel | (in ? 0 : 0x20000000)) : -1;
      c8s[u] = c8;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pix[u] == -1) continue;
      const int pixel = pix[u] & 0x0fffffff;
      float v[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      if (prologue) {
        // (scale/shift of the channel padding are zero in LDS: a padded channel is 0 in memory and stays leaky(0*0+0) = 0,
        // so no per-element "does this channel exist" test -- it compiled into eight dependent LDS round trips per item)
        const float* sc = sScale + c0 + c8s[u] * 8; const float* sh = sShift + c0 + c8s[u] * 8;
        const f32x4 sc0 = *(const f32x4*)sc, sc1 = *(const f32x4*)(sc + 4), sh0 = *(const f32x4*)sh, sh1 = *(const f32x4*)(sh + 4);
        bn_leaky8(v, sc0, sc1, sh0, sh1, pslope);
        if (pix[u] & 0x20000000) zero8(v);                          // padding stays zero after the activation
      }
      uint4 hi, lo;
      if (presplit) { hi = __builtin_bit_cast(uint4, va[u]); lo = __builtin_bit_cast(uint4, vb[u]); }
      else if (F16) split8_f16(v, hi, lo); else split8(v, hi, lo);
      char* d = img + pixel * pitch + c8s[u] * 16;
      *(uint4*)d = hi;
      *(uint4*)(d + lo_off) = lo;
    }
  }
}

// F16 = true: forward convolution, operands split into fp16 halves (activations/weights are O(1) quantities);
// F16 = false: data gradient, operands split into bf16 halves (gradients need bf16's exponent range).
// W8: eight waves (512 threads) on the same 128-pixel box -- wave = (row pair wr, column half wc); used where LDS allows
// only one workgroup per CU, so that two waves per SIMD can overlap each other's staging, LDS and matrix phases.
// HALF (round 3): four waves on a 64-pixel box -- wave = (row pair wr of two, column half wc): the box and its patch are half the
// size, so the layers whose 128-pixel form fills the LDS of a CU (one workgroup, nothing overlaps its staging) run two workgroups
// per CU, and the deep layers with fewer boxes than CUs spread over twice as many workgroups.
template <bool F16, bool STRIDED, int NREP, bool W8 = false, bool HALF = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void k_conv_patch(
    PGeom g, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const uint4* __restrict__ wp, float* __restrict__ dst, float* __restrict__ stat_partial,
    int accumulate, int n_per_blk, PersBwd bw) {
  // bw.yraw != nullptr (data gradient only): fused BatchNorm-backward reduction of the unit this gradient feeds, as in
  // k_conv_pers -- the epilogue reads that unit's raw output at the pixels it has just produced, stores
  // g = dA * leaky'(bn(y)) instead of dA and writes sum(g), sum(g * xhat) per channel as this box's partial row.
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* sP = sm;                                   // patch hi | lo
  char* sB = sm + g.off_b;                         // B tile hi [n][160 B] | lo
  int* sK = (int*)(sm + g.off_koffs);              // [nstages*8]
  int4* sR = (int4*)(sm + g.off_rows);             // [PM] {rowpix bytes, global dst pixel or -1, packed row numerators, -}
  int* sG = (int*)(sm + g.off_pixg);               // [P] global source pixel or -1
  float* sScale = (float*)(sm + g.off_scale);      // [Cps] scale | shift
  float* sShift = sScale + PMAXC;

  const int t = threadIdx.x;
  constexpr int NT = W8 ? 512 : 256;
  static_assert(!(W8 && HALF), "one wave layout");
  constexpr bool SPLITC = W8 || HALF;                     // the waves of a row pair split the column tiles
  constexpr int RG = HALF ? 2 : 4;                        // row pairs (32 rows each) of the box
  constexpr int NW = SPLITC ? (NREP + 1) / 2 : NREP;      // column tiles per wave
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave & (RG - 1), wc = wave / RG;         // rows 32*wr .. +31; column half (W8 / HALF only)
  const int j0 = wc ? NREP - NW : 0;                      // first column tile of this wave (odd NREP: the halves overlap by one)
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.y * n_per_blk;
  const int ncols = min(n_per_blk, g.N16 - n0);
  const bool prologue = pscale != nullptr;
  const int dbg = (accumulate >> 8) & 0xff;      // timing experiments (MD_DBG): 1 skip patch loads, 2 skip MFMA loop, 4 skip stores
  const bool presplit = (accumulate >> 16) & 1;   // source = pre-split bf16 gradient (data gradient only)
  accumulate &= 1;

  // ---- which box
  int b = blockIdx.x;
  const int xb = b % g.nbx; b /= g.nbx;
  const int yb = b % g.nby; b /= g.nby;
  const int tb = b % g.nbt; const int n = b / g.nbt;
  const int t0 = tb * g.bt, y0 = yb * g.by, x0 = xb * g.bx;

  // strided data gradient: source coordinate o = (i + pad - tap) / stride when divisible.  The patch starts at
  // po = floor((i0 + pad - (k-1)) / stride); base = i0 + pad - po*stride >= k-1 keeps every numerator >= 0.
  int po_t = 0, po_h = 0, po_w = 0, base_t = 0, base_h = 0, base_w = 0;
  if (STRIDED) {
    const int nt_ = t0 + g.padt - (g.kt - 1), nh_ = y0 + g.padh - (g.kh - 1), nw_ = x0 + g.padw - (g.kw - 1);
    po_t = nt_ >= 0 ? nt_ >> g.lt : -((-nt_ + g.dst_ - 1) >> g.lt);
    po_h = nh_ >= 0 ? nh_ >> g.lh : -((-nh_ + g.dsh_ - 1) >> g.lh);
    po_w = nw_ >= 0 ? nw_ >> g.lw : -((-nw_ + g.dsw_ - 1) >> g.lw);
    base_t = t0 + g.padt - po_t * g.dst_; base_h = y0 + g.padh - po_h * g.dsh_; base_w = x0 + g.padw - po_w * g.dsw_;
  }

  // ---- tables
  for (int p = t; p < g.P; p += NT) {
    const int ppt = mdiv(p, g.m_pyx); const int r = p - ppt * g.pyx;
    const int ppy = mdiv(r, g.m_px); const int ppx = r - ppy * g.px;
    int st, sy, sx;
    if (STRIDED) { st = po_t + ppt; sy = po_h + ppy; sx = po_w + ppx; }
    else { st = t0 * g.st + g.org_t + ppt; sy = y0 * g.sh + g.org_h + ppy; sx = x0 * g.sw + g.org_w + ppx; }
    const bool v = ((unsigned)st < (unsigned)g.Ts) && ((unsigned)sy < (unsigned)g.Hs) && ((unsigned)sx < (unsigned)g.Ws);
    sG[p] = v ? ((n * g.Ts + st) * g.Hs + sy) * g.Ws + sx : -1;
  }
  if (t < 32 * RG) {
    const int rt = mdiv(t, g.m_byx); const int r = t - rt * g.byx;
    const int ry = mdiv(r, g.m_bx); const int rx = r - ry * g.bx;
    const bool v = (rt < g.bt) && (t0 + rt < g.Td) && (y0 + ry < g.Hd) && (x0 + rx < g.Wd);
    int4 ri;
    ri.x = v ? ((rt * g.st * g.py + ry * g.sh) * g.px + rx * g.sw) * g.ppitch : 0;
    ri.y = v ? ((n * g.Tdf + (t0 + rt) * g.dmt + g.dpt) * g.Hdf + (y0 + ry) * g.dmh + g.dph) * g.Wdf + (x0 + rx) * g.dmw + g.dpw : -1;
    ri.z = v ? ((base_t + rt) | ((base_h + ry) << 8) | ((base_w + rx) << 16)) : -1;
    ri.w = 0;
    sR[t] = ri;
  }
  for (int q = t; q < g.nstages * 8; q += NT) {
    int ko = STRIDED ? -1 : 0;              // strided: -1 marks a K-padding chunk (reads the zero pixel)
    if (q < g.Kc8) {
      const int tap = mdiv(q, g.magicC8); const int c8 = q - tap * g.C8;
      const int dt = mdiv(tap, g.m_khw); const int r = tap - dt * g.khw;
      const int dy = mdiv(r, g.m_kw); const int dx = r - dy * g.kw;
      if (STRIDED) ko = dt | (dy << 8) | (dx << 16) | (c8 << 24);
      else ko = ((dt * g.py + dy) * g.px + dx) * g.ppitch + c8 * 16;
    }
    sK[q] = ko;
  }
  if (prologue) for (int c = t; c < g.C8 * 8; c += NT) {       // zero for the channel padding up to whole 8-channel chunks
    const int cs = g.pack2 ? (c & 3) : c;
    sScale[c] = c < g.Cps ? pscale[cs] : 0.f; sShift[c] = c < g.Cps ? pshift[cs] : 0.f;
  }

  // ---- B tile prefetch (registers): [stage][hi|lo][N16][8 chunks] uint4, this block's rows n0..n0+ncols.
  // Loads are unconditional (clamped index) so that the prefetch stays a straight run of global loads.
  const int bchunks = ncols * 8;                 // per half
  constexpr int NB = (NREP * 128 + NT - 1) / NT;   // 16-byte chunks per thread per half (NREP*16*8 / threads)
  uint4 rb[2][NB];
  auto load_b = [&](int kb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint4* base = wp + ((size_t)(kb * 2 + h) * g.N16 + n0) * 8;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int c = t + NT * i;
        const uint4 v = base[min(c, bchunks - 1)];     // columns past ncols get copies of a valid one; never stored
        rb[h][i] = v;
      }
    }
  };
  load_b(0);
  __syncthreads();

  // ---- stage the patch: global 32 B per lane -> (BN+act) -> split -> 16 B hi + 16 B lo
  if (!(dbg & 1))
    stage_image<F16, NT>(make_rsrc(src, presplit ? (unsigned)(g.src_bytes / (g.Cps * 4u)) * (unsigned)g.C8 * 32u : g.src_bytes), g.Cps, 0,
                         presplit ? 2 * g.C8 : g.Cps >> 2, sG, g.P, g.C8, g.magicC8, sP, g.ppitch, g.lo_off, prologue, sScale, sShift, pslope, t,
                         presplit);
  if (STRIDED) {
    for (int i = t * 16; i < g.ppitch; i += NT * 16) {
      *(uint4*)(sP + g.zero_off + i) = make_uint4(0, 0, 0, 0);
      *(uint4*)(sP + g.lo_off + g.zero_off + i) = make_uint4(0, 0, 0, 0);
    }
  }

  // per-lane row offsets of this wave's two 16-row slabs
  const int rp0 = sR[wr * 32 + li].x, rp1 = sR[wr * 32 + 16 + li].x;
  const int rc0 = sR[wr * 32 + li].z, rc1 = sR[wr * 32 + 16 + li].z;     // packed numerators (strided dgrad)
  auto strided_off = [&](int rc, int tc) -> int {
    if (rc < 0 || tc < 0) return g.zero_off;
    const int pk = rc - (tc & 0xffffff);
    const int ct = (pk & 0xff) >> g.lt, cy = ((pk >> 8) & 0xff) >> g.lh, cx = ((pk >> 16) & 0xff) >> g.lw;
    const bool ok = ((pk & g.oddmask) == 0) && ct < g.pt && cy < g.py && cx < g.px;
    return ok ? ((ct * g.py + cy) * g.px + cx) * g.ppitch + ((tc >> 24) & 0xff) * 16 : g.zero_off;
  };

  f32x4 acc[2][NW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int blo = n_per_blk * PB_PITCH;    // lo half of the B tile
  // Software pipeline over the k32 steps (two per 64-k stage): the A fragments of step q+1 (they live in the patch,
  // which does not change during the loop) are requested at the start of step q, and inside a stage the B fragments
  // of column tile j+1 are requested before the six MFMAs of tile j, so LDS latency hides behind the matrix pipe.
  const int nsteps = ((dbg & 2) ? 0 : g.nstages) * 2;
  uint4 fa[2][4];                          // [step parity][hi0, lo0, hi1, lo1]
  auto a_offsets = [&](int ko, int& o0, int& o1) {
    o0 = STRIDED ? strided_off(rc0, ko) : rp0 + ko;
    o1 = STRIDED ? strided_off(rc1, ko) : rp1 + ko;
  };
  auto load_a = [&](int o0, int o1, uint4* f) {
    f[0] = *(const uint4*)(sP + o0); f[1] = *(const uint4*)(sP + g.lo_off + o0);
    f[2] = *(const uint4*)(sP + o1); f[3] = *(const uint4*)(sP + g.lo_off + o1);
  };
  const char* bbase = sB + (j0 * 16 + li) * PB_PITCH + lg * 16;
  int ko_next = 0;                         // K offset of step q+1 while step q runs
  if (nsteps) {
    __syncthreads();                       // patch staged
    int o0, o1;
    a_offsets(sK[lg], o0, o1);
    load_a(o0, o1, fa[0]);
    ko_next = sK[4 + lg];                  // nsteps >= 2
  }
  for (int kb = 0; kb * 2 < nsteps; ++kb) {
    __syncthreads();                       // previous B tile consumed
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int c = t + NT * i;
        if (c < n_per_blk * 8 && !((dbg & 16) && kb > 0)) *(uint4*)(sB + h * blo + (c >> 3) * PB_PITCH + (c & 7) * 16) = rb[h][i];
      }
    __syncthreads();
    if (kb + 1 < g.nstages && !(dbg & 32)) load_b(kb + 1);
    uint4 fb[2][2];                        // [buffer][hi, lo]
    fb[0][0] = *(const uint4*)bbase; fb[0][1] = *(const uint4*)(bbase + blo);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      {   // A fragments of the next step (the last step re-reads its own: branch-free), K offset of the one after
        int o0, o1;
        a_offsets(ko_next, o0, o1);
        load_a(o0, o1, fa[s ^ 1]);
        ko_next = sK[min(kb * 2 + s + 2, nsteps - 1) * 4 + lg];
      }
      const uint4 ah0 = fa[s][0], al0 = fa[s][1], ah1 = fa[s][2], al1 = fa[s][3];
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int cur = (s * NW + j) & 1;
        const bool more = j + 1 < NW || s == 0;        // next column tile (of this step, or tile 0 of the second step)
        if (more) {
          const int jn = j + 1 < NW ? j + 1 : 0, sn = j + 1 < NW ? s : 1;
          const char* bp = bbase + jn * 16 * PB_PITCH + sn * 64;
          fb[cur ^ 1][0] = *(const uint4*)bp; fb[cur ^ 1][1] = *(const uint4*)(bp + blo);
        }
        const uint4 bh = fb[cur][0], bl = fb[cur][1];
        // smallest terms first: lo*hi and hi*lo, then hi*hi
        acc[0][j] = mma<F16>(al0, bh, acc[0][j]);
        acc[1][j] = mma<F16>(al1, bh, acc[1][j]);
        acc[0][j] = mma<F16>(ah0, bl, acc[0][j]);
        acc[1][j] = mma<F16>(ah1, bl, acc[1][j]);
        acc[0][j] = mma<F16>(ah0, bh, acc[0][j]);
        acc[1][j] = mma<F16>(ah1, bh, acc[1][j]);
        // keep the issue order: this tile's LDS requests, then its six MFMAs
        if (j == 0 && more) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
        else if (j == 0) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        else if (more) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
    }
  }

  // ---- epilogue: store + BatchNorm partial sums over VALID rows.  Rows outside the tensor get the out-of-range
  // buffer offset (store dropped) and weight 0 in the sums; no branches.
  unsigned goff[2][4];
  float gw[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = sR[wr * 32 + a * 16 + lg * 4 + r].y;
      goff[a][r] = gi >= 0 ? (unsigned)(gi * g.Cpd + n0 + j0 * 16 + li) * 4u : MD_OOB;
      gw[a][r] = gi >= 0 ? 1.f : 0.f;
    }
  __syncthreads();
  float* red = (float*)sP;   // [4 waves][2][PNREP*16]
  const __amdgpu_buffer_rsrc_t drs = make_rsrc(dst, (dbg & 4) ? 0u : g.dst_bytes);
  const bool fuse = !F16 && bw.yraw != nullptr;                  // wave-uniform
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(fuse ? bw.yraw : dst, g.dst_bytes);
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    // (W8, odd NREP: the second half's first tile is also the first half's last -- computed twice, written and counted once)
    const bool dup = SPLITC && (NREP & 1) && wc == 1 && j == 0;
    const bool colok = !dup && n0 + (j0 + j) * 16 + li < g.Cpd;
    float s1 = 0.f, s2 = 0.f;
    float prev[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (accumulate) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          prev[a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, colok ? goff[a][r] : MD_OOB, j * 64, 0));
    }
    if (fuse) {
      const int c = n0 + (j0 + j) * 16 + li;
      const bool cok = c < g.Cpd;
      const float csc = cok ? bw.scale[c] : 0.f, csh = cok ? bw.shift[c] : 0.f, cmu = cok ? bw.mean[c] : 0.f, cis = cok ? bw.invstd[c] : 0.f;
      float yv[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          yv[a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, colok ? goff[a][r] : MD_OOB, j * 64, 0));
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pre = fmaf(yv[a][r], csc, csh);
          const float v = (acc[a][j][r] + prev[a][r]) * md_dleaky(pre, bw.slope);      // g = dA * leaky'(bn(y))
          const float vm = v * gw[a][r];
          s1 += vm; s2 = fmaf(vm, (yv[a][r] - cmu) * cis, s2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), drs, colok ? goff[a][r] : MD_OOB, j * 64, 0);
        }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[a][j][r];
          const float vm = v * gw[a][r];
          s1 += vm; s2 = fmaf(vm, v, s2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + prev[a][r]), drs, colok ? goff[a][r] : MD_OOB, j * 64, 0);
        }
    }
    if (stat_partial != nullptr) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (lg == 0 && !dup) {
        red[(wr * 2 + 0) * (PNREP * 16) + (j0 + j) * 16 + li] = s1;
        red[(wr * 2 + 1) * (PNREP * 16) + (j0 + j) * 16 + li] = s2;
      }
    }
  }
  if (stat_partial != nullptr) {
    __syncthreads();
    if (t < ncols && n0 + t < g.Cpd) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < RG; ++w) {
        s1 += red[(w * 2 + 0) * (PNREP * 16) + t];
        s2 += red[(w * 2 + 1) * (PNREP * 16) + t];
      }
      float* sp = stat_partial + (size_t)blockIdx.x * 2 * g.Cpd;
      sp[n0 + t] = s1;
      sp[g.Cpd + n0 + t] = s2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing for k_conv_patch: [stage][hi|lo][n (N16)][8 chunks][8 bf16], K order = (tap, 8-channel chunk).
// mode 0 (forward): n = cout, channel = cin, tap as is.  mode 1 (data gradient): n = cin, channel = cout,
// tap reversed (the patch walks the flipped filter).
// ------------------------------------------------------------------------------------------------
// mode 3 (pixel-pair reinterpretation, forward): the packed filter has kwp super-taps of 8 channels = 2 pixels x 4;
// super-tap sx', channel 4j+c  <->  real tap dx = 2 sx' + j + shift, channel c.
__device__ __forceinline__ float pack2_weight(const float* __restrict__ w, int n, int tap, int ch, int Cout, int Cin, int kwp,
                                              int kw_real, int shift, int taps_real) {
  const int sx = tap % kwp, row = tap / kwp;            // row = dt*kh + dy
  const int j = ch >> 2, c = ch & 3;
  const int dx = 2 * sx + j + shift;
  if (n < Cout && c < Cin && dx >= 0 && dx < kw_real) return w[((size_t)n * Cin + c) * taps_real + row * kw_real + dx];
  return 0.f;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

// Residue class of a strided data gradient (see patch_classes): destination pixels i = s*j + c per dimension receive
// only the taps tap0, tap0 + s, ... (kc of them); over the class grid j this is a unit-stride correlation.
struct ClassSpec { int c[3], tap0[3], kc[3], e[3]; };

// pers_wgs > 0: geometry for the persistent kernel (conv_pers.hip) at that many workgroups per CU -- the LDS budget then
// holds the whole packed weight operand instead of one streamed stage.
static bool patch_build(const MdConvDesc* d, int dgrad, PGeom* out, size_t* lds_bytes, const ClassSpec* cls = nullptr,
                        int pers_wgs = 0, int pm_rows = PM, int npb_force = 0) {
  PGeom g;
  const bool sdg = !cls && dgrad && (d->st != 1 || d->sh != 1 || d->sw != 1);      // strided data gradient, tap-test form
  g.strided = sdg ? 1 : 0;
  if (sdg && (ilog2_exact(d->st) < 0 || ilog2_exact(d->sh) < 0 || ilog2_exact(d->sw) < 0)) return false;
  g.st = dgrad ? 1 : d->st; g.sh = dgrad ? 1 : d->sh; g.sw = dgrad ? 1 : d->sw;
  g.dst_ = d->st; g.dsh_ = d->sh; g.dsw_ = d->sw;
  g.lt = sdg ? ilog2_exact(d->st) : 0; g.lh = sdg ? ilog2_exact(d->sh) : 0; g.lw = sdg ? ilog2_exact(d->sw) : 0;
  g.oddmask = (g.lt ? ((1 << g.lt) - 1) : 0) | ((g.lh ? ((1 << g.lh) - 1) : 0) << 8) | ((g.lw ? ((1 << g.lw) - 1) : 0) << 16);
  g.kt = d->kt; g.padt = d->pt; g.padh = d->ph; g.padw = d->pw;
  const int cs = dgrad ? d->Cout : d->Cin, cd = dgrad ? d->Cin : d->Cout;
  g.Ts = dgrad ? d->To : d->Ti; g.Hs = dgrad ? d->Ho : d->Hi; g.Ws = dgrad ? d->Wo : d->Wi; g.Cps = md_cpad(cs);
  g.Td = dgrad ? d->Ti : d->To; g.Hd = dgrad ? d->Hi : d->Ho; g.Wd = dgrad ? d->Wi : d->Wo; g.Cpd = md_cpad(cd);
  g.kh = d->kh; g.kw = d->kw; g.khw = d->kh * d->kw; g.taps = d->kt * g.khw;
  if (!dgrad) { g.org_t = -d->pt; g.org_h = -d->ph; g.org_w = -d->pw; }
  else { g.org_t = d->pt - (d->kt - 1); g.org_h = d->ph - (d->kh - 1); g.org_w = d->pw - (d->kw - 1); }
  g.Tdf = g.Td; g.Hdf = g.Hd; g.Wdf = g.Wd; g.dmt = g.dmh = g.dmw = 1; g.dpt = g.dph = g.dpw = 0;
  int kt_eff = d->kt, kh_eff = d->kh;
  if (cls) {
    const int sd[3] = {d->st, d->sh, d->sw};
    int cd_[3];
    const int full[3] = {g.Td, g.Hd, g.Wd};
    for (int i = 0; i < 3; ++i) cd_[i] = full[i] > cls->c[i] ? (full[i] - cls->c[i] + sd[i] - 1) / sd[i] : 0;
    if (cd_[0] < 1 || cd_[1] < 1 || cd_[2] < 1) return false;
    g.Td = cd_[0]; g.Hd = cd_[1]; g.Wd = cd_[2];
    g.dmt = sd[0]; g.dmh = sd[1]; g.dmw = sd[2]; g.dpt = cls->c[0]; g.dph = cls->c[1]; g.dpw = cls->c[2];
    g.kt = cls->kc[0]; g.kh = cls->kc[1]; g.kw = cls->kc[2]; g.khw = g.kh * g.kw; g.taps = g.kt * g.khw;
    g.org_t = cls->e[0] - (cls->kc[0] - 1); g.org_h = cls->e[1] - (cls->kc[1] - 1); g.org_w = cls->e[2] - (cls->kc[2] - 1);
    kt_eff = g.kt; kh_eff = g.kh;
  }
  // Pixel-pair reinterpretation (forward): a <= 4-channel input read with W-stride 2 (the stem) would fill only half
  // of every 8-channel K chunk.  The same memory is a [.., Wi/2][8] tensor of pixel pairs, over which the convolution
  // has unit W-stride and kwp = ceil-ish(kw/2)+1 super-taps: output x reads pairs x + lo .. x + hi, and super-tap sx',
  // channel 4j+c carries real tap dx = 2 sx' + j + shift (zero weight where dx falls outside the filter).
  g.pack2 = 0; g.pk_shift = 0; g.pk_kw = d->kw;
  static const int no_pack2 = getenv("MD_PACK2") && atoi(getenv("MD_PACK2")) == 0;
  int kw_eff = cls ? g.kw : d->kw, sw_eff = g.sw;
  if (!dgrad && !no_pack2 && g.Cps == 4 && d->sw == 2 && (d->Wi & 1) == 0 && d->kw >= 2) {
    const int lo = -((d->pw + 1) / 2);
    const int num = d->kw - 1 - d->pw;
    const int hi = num >= 0 ? num / 2 : -((-num + 1) / 2);
    g.pack2 = 1; g.pk_shift = d->pw + 2 * lo; g.pk_kw = d->kw;
    g.kw = hi - lo + 1; g.khw = g.kh * g.kw; g.taps = d->kt * g.khw;
    g.org_w = lo; g.sw = 1; g.Ws = d->Wi / 2; g.Cps = 8;
    kw_eff = g.kw; sw_eff = 1;
  }
  g.C8 = (g.Cps + 7) / 8;
  g.ppitch = pitch_for(g.C8);
  // persistent form: hi and lo chunks of a pixel share one pitch ([pixel][hi C8 | lo C8 | pad]) -- smaller than two padded arrays
  if (pers_wgs) g.ppitch = pitch_for(2 * g.C8);
  g.Kc8 = g.taps * g.C8;
  g.nstages = md_cdiv(g.Kc8, 8);
  g.N16 = md_round_up(cd, 16);
  g.magicC8 = magic_of(g.C8);
  {
    const unsigned long long sb = (unsigned long long)d->N * g.Ts * g.Hs * g.Ws * g.Cps * 4ull;
    const unsigned long long db = (unsigned long long)d->N * g.Tdf * g.Hdf * g.Wdf * g.Cpd * 4ull;
    if (sb >= 0x80000000ull || db >= 0x80000000ull) return false;     // buffer addressing: 2 GiB per tensor
    g.src_bytes = (unsigned)sb; g.dst_bytes = (unsigned)db;
  }
  if (g.Cps > PMAXC) return false;
  const int nchunks_ = md_cdiv(g.N16, PNREP * 16);
  const int npb_ = npb_force ? npb_force : md_round_up(md_cdiv(g.N16, nchunks_), 16);
  // LDS: everything but the patch
  if (pers_wgs && (sdg || g.N16 > 128 || g.N16 < 32)) return false;
  const int pm = pm_rows;
  const size_t fixed = pers_wgs ? pers_bres_bytes(g.Kc8, g.N16) + pers_fixed_bytes(g.Cps, g.N16) + (size_t)g.nstages * 8 * 4
                                : (size_t)2 * npb_ * PB_PITCH + (size_t)g.nstages * 8 * 4 + (size_t)PM * 16 + 3072 + 2 * PMAXC * 4 + 1024;
  const size_t cap = (size_t)160 * 1024;
  if (fixed + 4096 > cap) return false;
  const long long per_px = pers_wgs ? (long long)2 * g.ppitch : (long long)2 * g.ppitch + 4;     // persistent: two patches (one per team), hi+lo in one pitch
  long long maxP = (long long)(cap - fixed) / per_px - 2;
  static const int soft_kb = getenv("MD_LDS_SOFT_KB") ? atoi(getenv("MD_LDS_SOFT_KB")) : 80;   // target LDS per workgroup
  long long softP = pers_wgs ? maxP : ((long long)soft_kb * 1024 - (long long)fixed) / per_px - 2;
  long long idx_cap = 65535 / g.C8;              // item index must stay below 2^16 for the magic division
  if (pers_wgs && idx_cap > (long long)PERS_MAXI * 256 / g.C8) idx_cap = (long long)PERS_MAXI * 256 / g.C8;   // register-staged items (per team)
  if (maxP > idx_cap) maxP = idx_cap;
  if (softP < 1) softP = 1;
  if (maxP < 1) return false;
  if (!choose_box(g.Td, g.Hd, g.Wd, kt_eff, kh_eff, kw_eff, sdg ? d->st : g.st, sdg ? d->sh : g.sh, sdg ? d->sw : sw_eff, sdg,
                  (int)maxP, (int)softP, &g.bt, &g.by, &g.bx, pm)) return false;
  g.byx = g.by * g.bx;
  g.nbt = md_cdiv(g.Td, g.bt); g.nby = md_cdiv(g.Hd, g.by); g.nbx = md_cdiv(g.Wd, g.bx);
  if (!sdg) {
    g.pt = (g.bt - 1) * g.st + kt_eff; g.py = (g.by - 1) * g.sh + kh_eff; g.px = (g.bx - 1) * g.sw + g.kw;
  } else {
    g.pt = (g.bt + d->kt - 2) / d->st + 2; g.py = (g.by + d->kh - 2) / d->sh + 2; g.px = (g.bx + d->kw - 2) / d->sw + 2;
  }
  g.pyx = g.py * g.px; g.P = g.pt * g.pyx;
  g.m_pyx = magic_of(g.pyx); g.m_px = magic_of(g.px); g.m_byx = magic_of(g.byx); g.m_bx = magic_of(g.bx);
  g.m_khw = magic_of(g.khw); g.m_kw = magic_of(g.kw);
  g.zero_off = g.P * g.ppitch;                          // one all-zero pixel behind the patch (invalid taps read it)
  g.lo_off = ((g.P + 1) * g.ppitch + 15) & ~15;
  if (pers_wgs) g.lo_off = g.C8 * 16;
  size_t off = (size_t)2 * g.lo_off;
  const size_t red = (size_t)4 * 2 * PNREP * 16 * 4;      // epilogue reduction scratch aliases the patch
  if (off < red) off = red;
  g.off_b = (int)off; off += (size_t)2 * npb_ * PB_PITCH;
  g.off_koffs = (int)off; off += (size_t)g.nstages * 8 * 4;
  g.off_rows = (int)off; off += (size_t)PM * 16 + 3072;      // >= 4608 B: also the persistent kernel's reduction scratch
  g.off_pixg = (int)off; off += (size_t)((g.P * 4 + 15) & ~15);
  g.off_scale = (int)off; off += (size_t)2 * PMAXC * 4;
  if (!pers_wgs && off > cap) return false;
  if (pers_wgs && (g.pt >= 64 || g.py >= 512 || g.px >= 512 || g.bt >= 256 || g.by >= 256 || g.bx >= 256)) return false;
  if (getenv("MD_PLAN_PRINT"))
    fprintf(stderr, "patch%s %s %d->%d k%d%d%d s%d%d%d dst %dx%dx%d: box %dx%dx%d patch %dx%dx%d=%d C8=%d stages=%d lds=%zu\n",
            pers_wgs ? "(persistent)" : (pm_rows != PM ? "(half)" : ""), dgrad ? (cls ? "dgrad-class" : "dgrad") : "fwd", d->Cin, d->Cout, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, g.Td, g.Hd,
            g.Wd, g.bt, g.by, g.bx, g.pt, g.py, g.px, g.P, g.C8, g.nstages, off);
  *out = g; *lds_bytes = off;
  return true;
}

#include <atomic>
static std::atomic<int> g_pers_grid{0};     // > 0: test override of the persistent kernels' grid (md_set_pers_grid)
extern "C" int md_set_pers_grid(int n) { return g_pers_grid.exchange(n > 0 ? n : 0); }
struct PersVariant {           // persistent-kernel form of the same launch
  bool on; PersGeom pg; size_t lds; int grid;
  // column slices of <= 3 tiles each (data gradient with more than 3 destination tiles): a FUSED launch runs slice by slice, each
  // with the fused BatchNorm-backward reduction the full-width kernel has no registers for
  int nsl; PersGeom sl[4]; size_t sl_lds[4];
};
struct PatchClass { PGeom g; size_t lds; size_t wp_off; ClassSpec spec; PersVariant pers; };     // wp_off: floats into the packed operand
struct PatchPlan {
  PGeom g; size_t lds; int N; int dgrad; int ncls; PatchClass cls[8]; PersVariant pers;
  bool half; PGeom gh; size_t ldsh; int half_npb;      // 64-pixel boxes, four waves as 2 row pairs x 2 column halves (k_conv_patch HALF)
};

// Persistent form (conv_pers.hip) of one launch, when the geometry qualifies: the packed weights fit in LDS beside the
// patch, the box is nearly full, and there are at least as many boxes as resident workgroups.  The packed-weight format
// (K order, N16) does not depend on the box, so both forms read the same operand.
static void pers_try(const MdConvDesc* d, int dgrad, const ClassSpec* cls, const PGeom& classic, PersVariant* pv) {
  pv->on = false; pv->nsl = 0;
  static const int off = getenv("MD_PERS") && atoi(getenv("MD_PERS")) == 0;
  static const int off_f = getenv("MD_PERS_FWD") && atoi(getenv("MD_PERS_FWD")) == 0;
  static const int off_d = getenv("MD_PERS_DGRAD") && atoi(getenv("MD_PERS_DGRAD")) == 0;
  if (off || (dgrad ? off_d : off_f)) return;
  static const int min_rows = getenv("MD_PERS_MIN_ROWS") ? atoi(getenv("MD_PERS_MIN_ROWS")) : 96;
  const int grid_override = g_pers_grid.load();
  PGeom g; size_t lds = 0;
  if (!patch_build(d, dgrad, &g, &lds, cls, 1)) return;
  if (g.Kc8 != classic.Kc8 || g.N16 != classic.N16 || g.nstages != classic.nstages) return;
  if (!grid_override && g.bt * g.by * g.bx < min_rows) return;
  PersGeom pg; pg.g = g;
  int grid = 0;
  if (!pers_finish(&pg, &lds, &grid)) return;
  if (grid_override) grid = grid_override;
  pg.nboxes = d->N * g.nbt * g.nby * g.nbx;
  if (pg.nboxes >= 65536 || pg.nboxes < 2 * grid) return;       // at least one box per team
  pg.nit = md_cdiv(g.P * g.C8, 256);
  if (pg.nit > PERS_MAXI || (pg.nit > 4 && g.N16 > 48) || g.N16 > 96) return;      // instantiated: <= 6 tiles; 8 items only with <= 3 tiles
  pg.m_nbx = magic_of(g.nbx); pg.m_nby = magic_of(g.nby); pg.m_nbt = magic_of(g.nbt);
  pv->on = true; pv->pg = pg; pv->lds = lds; pv->grid = grid;
  pv->nsl = 0;
  // (measured slower on the 72-channel data gradients of the 64x64 stage, which are bound by their patch loads and stores, not
  // by the matrix work a slice saves: 6.07 against 5.97 ms per step, profiles/r03_fuse_fallback.txt; MD_FUSE_SLICES=1 builds them)
  static const int slices_on = getenv("MD_FUSE_SLICES") && atoi(getenv("MD_FUSE_SLICES")) == 1;
  if (dgrad && slices_on && g.N16 > 48 && pg.nit <= 4) {
    int n0 = 0, k = 0; bool ok = true;
    while (n0 < g.N16 && ok) {
      const int rem = g.N16 - n0;
      const int w = (rem >= 80 || rem == 48) ? 48 : (rem > 48 ? 32 : rem);      // 3-tile slices while at least 2 tiles remain behind them
      if (w < 32 || k >= 4) { ok = false; break; }
      PersGeom sg = pg; sg.g.N16 = w;
      size_t sl = 0; int gr = 0;
      if (!pers_finish(&sg, &sl, &gr)) { ok = false; break; }
      sg.nboxes = pg.nboxes; sg.nit = pg.nit; sg.m_nbx = pg.m_nbx; sg.m_nby = pg.m_nby; sg.m_nbt = pg.m_nbt;
      sg.n0 = n0; sg.N16w = g.N16;
      pv->sl[k] = sg; pv->sl_lds[k] = sl; ++k; n0 += w;
    }
    if (ok && n0 == g.N16) pv->nsl = k;
  }
  if (getenv("MD_PLAN_PRINT")) fprintf(stderr, "  -> persistent: boxes %d grid %d lds %zu nsteps %d nit %d\n", pg.nboxes, grid, lds, pg.nsteps, pg.nit);
}

// Strided data gradient by residue classes.  dX[i] = sum over taps with (i + pad - tap) % s == 0 of
// dY[(i + pad - tap) / s] W[tap].  For i = s*j + c the valid taps are tap0 + s*a, tap0 = (c + pad) % s, a < kc, and the
// source is j + e - a with e = (c + pad - tap0) / s: one dense unit-stride problem per class (s_t*s_h*s_w of them)
// instead of testing every tap for every pixel.  Needs k >= s in every strided dimension (each class has a tap).
static int patch_classes(const MdConvDesc* d, PatchPlan* pp) {
  const int sd[3] = {d->st, d->sh, d->sw}, kd[3] = {d->kt, d->kh, d->kw}, pd[3] = {d->pt, d->ph, d->pw};
  const int full[3] = {d->Ti, d->Hi, d->Wi};
  for (int i = 0; i < 3; ++i) if (sd[i] > 1 && kd[i] < sd[i]) return 0;
  if (sd[0] * sd[1] * sd[2] > 8) return 0;
  // one launch per class: worth it only when each still fills the chip (measured: 704 boxes gain, 176 boxes lose)
  static const long long min_px = getenv("MD_DGRAD_CLASS_MIN_PX") ? atoll(getenv("MD_DGRAD_CLASS_MIN_PX")) : 65536;
  if ((long long)d->N * d->Ti * d->Hi * d->Wi < min_px) return 0;
  int n = 0; size_t off = 0;
  for (int ct = 0; ct < sd[0]; ++ct)
    for (int ch = 0; ch < sd[1]; ++ch)
      for (int cw = 0; cw < sd[2]; ++cw) {
        ClassSpec cs; const int c[3] = {ct, ch, cw};
        bool empty = false;
        for (int i = 0; i < 3; ++i) {
          cs.c[i] = c[i]; cs.tap0[i] = (c[i] + pd[i]) % sd[i];
          cs.kc[i] = (kd[i] - cs.tap0[i] + sd[i] - 1) / sd[i];
          cs.e[i] = (c[i] + pd[i] - cs.tap0[i]) / sd[i];
          if (c[i] >= full[i]) empty = true;
          if (cs.kc[i] < 1) return 0;
        }
        if (empty) continue;                 // no destination pixel in this class
        PatchClass& pc = pp->cls[n];
        if (!patch_build(d, 1, &pc.g, &pc.lds, &cs)) return 0;
        pers_try(d, 1, &cs, pc.g, &pc.pers);
        pc.spec = cs; pc.wp_off = off;
        off += (size_t)pc.g.nstages * 2 * pc.g.N16 * 64 / 2;
        ++n;
      }
  return n;
}

#include <map>
#include <mutex>
#include <array>
#include <atomic>
#include <cstdlib>
static std::atomic<int> g_exact_fp32{0};
extern "C" int md_set_exact_fp32(int on) { int old = g_exact_fp32.exchange(on ? 1 : 0); return old; }
extern "C" int md_get_exact_fp32(void) { return g_exact_fp32.load(); }

const PatchPlan* patch_lookup(const MdConvDesc* d, int dgrad) {
  if (g_exact_fp32.load()) return nullptr;     // exact mode: every convolution on the fp32-MFMA gather kernels
  {   // debugging aid: MD_PATCH_FWD=0 / MD_PATCH_DGRAD=0 route one direction to the generic kernel
    static const int dis_f = getenv("MD_PATCH_FWD") && atoi(getenv("MD_PATCH_FWD")) == 0;
    static const int dis_d = getenv("MD_PATCH_DGRAD") && atoi(getenv("MD_PATCH_DGRAD")) == 0;
    if ((dgrad && dis_d) || (!dgrad && dis_f)) return nullptr;
  }
  {   // A Linear over rows (1x1x1, unit stride, the rows laid along W of a single image: models/_unit.py::LinearRowsFunction) with
      // many rows goes to the K-streaming GEMM (k_linear_split: three stages in flight, no per-box tables) instead of the per-box kernel
      // (whose two-stage K loop at K = 128 is all prologue: 15 of 43 us are box tables and the first weight stage's latency,
      // tools/r03_linear_abl.py).  ViViT cfg3 captured step 2.489 -> 2.426 ms; MD_LINEAR_PREFER=0 restores the per-box kernel.
    static const int prefer = getenv("MD_LINEAR_PREFER") ? atoi(getenv("MD_LINEAR_PREFER")) : 1;
    if (prefer && d->kt == 1 && d->kh == 1 && d->kw == 1 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 0 && d->ph == 0 && d->pw == 0 &&
        d->N == 1 && d->Ti == 1 && d->Hi == 1 && d->Wi >= 4096 && (d->Cin & 3) == 0 && (d->Cout & 3) == 0)
      return nullptr;
  }
  static std::mutex mu;
  static std::map<std::array<int, 20>, PatchPlan*> cache;    // value nullptr = does not qualify
  std::array<int, 20> key = {d->N, d->Ti, d->Hi, d->Wi, d->Cin, d->To, d->Ho, d->Wo, d->Cout, d->kt, d->kh, d->kw,
                             d->st, d->sh, d->sw, d->pt, d->ph, d->pw, dgrad, g_pers_grid.load()};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  PatchPlan* pp = nullptr;
  PGeom g; size_t lds = 0;
  static const int no_cls = getenv("MD_DGRAD_CLASSES") && atoi(getenv("MD_DGRAD_CLASSES")) == 0;
  if (dgrad && !no_cls && (d->st != 1 || d->sh != 1 || d->sw != 1)) {
    PatchPlan* cp = new PatchPlan();
    cp->ncls = patch_classes(d, cp);
    if (cp->ncls > 0) { cp->g = cp->cls[0].g; cp->lds = cp->cls[0].lds; cp->N = d->N; cp->dgrad = 1; cp->pers.on = false; pp = cp; }
    else delete cp;
  }
  if (pp) pp->half = false;
  if (!pp && patch_build(d, dgrad, &g, &lds)) {
    pp = new PatchPlan(); pp->g = g; pp->lds = lds; pp->N = d->N; pp->dgrad = dgrad; pp->ncls = 0;
    pers_try(d, dgrad, nullptr, g, &pp->pers);
    pp->half = false;
    // 64-pixel form: where the 128-pixel workgroup owns the CU's LDS (> 80 KB: nothing overlaps its staging) and the half-size
    // one fits twice, or where the layer has too few boxes to fill the chip.  MD_PATCH_HALF: 0 never, 1 both rules, 2 whenever built,
    // 3 / 4 one rule only.  Default 0 since the end of round 3: with the one-stream schedule and the second-form weight gradients
    // neither rule pays any more (bench, interleaved on one box: 1 -> 1355 / 1361 clips/s, 3 -> 1360 / 1363, 4 -> 1369 / 1366,
    // 0 -> 1372 / 1370; cfg5 656 / 658 -> 659 / 662); the variant stays built and tested (tests/test_patch_half_gpu.py).
    static const int half_env = getenv("MD_PATCH_HALF") ? atoi(getenv("MD_PATCH_HALF")) : 0;
    const bool strided_dg = dgrad && (d->st != 1 || d->sh != 1 || d->sw != 1);
    if (half_env && !pp->pers.on && !strided_dg && g.N16 >= 32) {
      const size_t cap2 = 80 * 1024 - 256;
      for (int nch = 1; nch <= g.N16 / 32 && !pp->half; ++nch) {
        const int npb = md_round_up(md_cdiv(g.N16, nch), 16);
        if (npb > PNREP * 16 || npb < 32) continue;
        PGeom gh; size_t ldsh = 0;
        if (!patch_build(d, dgrad, &gh, &ldsh, nullptr, 0, 64, npb)) continue;
        if (gh.Kc8 != g.Kc8 || gh.N16 != g.N16 || gh.nstages != g.nstages) continue;
        if (gh.bt * gh.by * gh.bx < 48) continue;                       // mostly empty boxes: not worth it
        if (ldsh > cap2 && nch < g.N16 / 32) continue;                  // try a narrower column group first
        const int boxes_full = d->N * g.nbt * g.nby * g.nbx;
        // measured (profiles/r03_patch_half.txt): splitting the columns to get under 80 KB costs more (the patch is staged once per
        // column group) than the second workgroup per CU brings -- 64 -> 144 at 32x32: 63 -> 79 us -- so only the un-split case
        // counts; the layers with a handful of boxes gain from the extra workgroups (128 -> 288 at 8x8: 19 -> 14 us).
        const bool two_per_cu = nch == 1 && lds > cap2 + 256 && ldsh <= cap2;
        const bool few_boxes = boxes_full < 64;
        // MD_PATCH_HALF: 0 never, 1 both rules, 2 whenever built, 3 the few-boxes rule only, 4 the two-per-CU rule only
        const bool take = half_env == 2 || (two_per_cu && (half_env == 1 || half_env == 4)) || (few_boxes && (half_env == 1 || half_env == 3));
        if (take) { pp->half = true; pp->gh = gh; pp->ldsh = ldsh; pp->half_npb = npb; }
        break;
      }
    }
  }
  cache[key] = pp;
  return pp;
}

// ---- batched weight packing: one launch for every patch-format operand of a network (<= 64 items by value)
#define PACK_BATCH 64
struct PackItem {
  const float* w; unsigned short* out;
  int Cout, Cin, taps, mode, C8, nstages, N16, f16, total, kwp, kw_real, shift, taps_real;
  int kc[3], tap0[3], cs[3], kh_real;       // mode 4: residue class of a strided data gradient
};
struct PackBatch { PackItem it[PACK_BATCH]; };
__global__ __launch_bounds__(256) void k_pack_weights_batch(PackBatch pb) {
  const PackItem& q = pb.it[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= q.total) return;
  const int e64 = idx & 63; const int r = idx >> 6;
  const int n = r % q.N16; const int kb = r / q.N16;
  const int ch8 = kb * 8 + (e64 >> 3);
  const int tap = ch8 / q.C8; const int ch = (ch8 - tap * q.C8) * 8 + (e64 & 7);
  float v = 0.f;
  if (tap < q.taps) {
    if (q.mode == 0) { if (n < q.Cout && ch < q.Cin) v = q.w[((size_t)n * q.Cin + ch) * q.taps + tap]; }
    else if (q.mode == 1) { if (n < q.Cin && ch < q.Cout) v = q.w[((size_t)ch * q.Cin + n) * q.taps + (q.taps - 1 - tap)]; }
    else if (q.mode == 2) { if (n < q.Cin && ch < q.Cout) v = q.w[((size_t)ch * q.Cin + n) * q.taps + tap]; }
    else if (q.mode == 3) v = pack2_weight(q.w, n, tap, ch, q.Cout, q.Cin, q.kwp, q.kw_real, q.shift, q.taps_real);
    else {      // mode 4: sub-tap (a't, a'h, a'w) of the class <-> real tap tap0 + s * (kc - 1 - a') per dimension
      const int khw = q.kc[1] * q.kc[2];
      const int at = tap / khw, r2 = tap - at * khw, ah = r2 / q.kc[2], aw = r2 - ah * q.kc[2];
      const int rt = q.tap0[0] + q.cs[0] * (q.kc[0] - 1 - at), rh = q.tap0[1] + q.cs[1] * (q.kc[1] - 1 - ah),
                rw = q.tap0[2] + q.cs[2] * (q.kc[2] - 1 - aw);
      if (n < q.Cin && ch < q.Cout) v = q.w[((size_t)ch * q.Cin + n) * q.taps_real + (rt * q.kh_real + rh) * q.kw_real + rw];
    }
  }
  const size_t o_hi = ((size_t)(kb * 2 + 0) * q.N16 + n) * 64 + e64;
  const size_t o_lo = ((size_t)(kb * 2 + 1) * q.N16 + n) * 64 + e64;
  if (q.f16) {
    const _Float16 hi = (_Float16)v; const _Float16 lo = (_Float16)(v - (float)hi);
    q.out[o_hi] = __builtin_bit_cast(unsigned short, hi); q.out[o_lo] = __builtin_bit_cast(unsigned short, lo);
  } else {
    const __bf16 hi = (__bf16)v; const __bf16 lo = (__bf16)(v - (float)hi);
    q.out[o_hi] = __builtin_bit_cast(unsigned short, hi); q.out[o_lo] = __builtin_bit_cast(unsigned short, lo);
  }
}

// Packs operand i (descs[i], dgrad[i]) into outs[i] when it uses the patch format and sets handled[i]; others are
// left to the caller.  One launch per 64 operands.
int patch_pack_batch(int n, const MdConvDesc* const* descs, const int* dgrad, const float* const* w, float* const* outs,
                     unsigned char* handled, hipStream_t s) {
  PackBatch pb; int cnt = 0, maxtot = 0;
  auto flush = [&]() -> int {
    if (!cnt) return MD_OK;
    MD_KLAUNCH(k_pack_weights_batch, dim3(md_cdiv(maxtot, 256), cnt), dim3(256), 0, s, pb);
    MD_CHECK_LAUNCH();
    cnt = 0; maxtot = 0;
    return MD_OK;
  };
  for (int i = 0; i < n; ++i) {
    handled[i] = 0;
    if (!outs[i]) { handled[i] = 1; continue; }
    const PatchPlan* pp = patch_lookup(descs[i], dgrad[i]);
    if (!pp) continue;
    const int nit = pp->ncls ? pp->ncls : 1;
    for (int c = 0; c < nit; ++c) {
      const PGeom& g = pp->ncls ? pp->cls[c].g : pp->g;
      PackItem& q = pb.it[cnt++];
      q.w = w[i]; q.out = (unsigned short*)(outs[i] + (pp->ncls ? pp->cls[c].wp_off : 0));
      q.Cout = descs[i]->Cout; q.Cin = descs[i]->Cin; q.taps = g.taps;
      q.mode = pp->ncls ? 4 : dgrad[i] ? (g.strided ? 2 : 1) : (g.pack2 ? 3 : 0);
      q.C8 = g.C8; q.nstages = g.nstages; q.N16 = g.N16; q.f16 = dgrad[i] ? 0 : 1;
      q.kwp = g.kw; q.kw_real = pp->ncls ? descs[i]->kw : g.pk_kw; q.shift = g.pk_shift;
      q.taps_real = descs[i]->kt * descs[i]->kh * descs[i]->kw; q.kh_real = descs[i]->kh;
      const int sd[3] = {descs[i]->st, descs[i]->sh, descs[i]->sw};
      for (int k = 0; k < 3; ++k) {
        q.kc[k] = pp->ncls ? pp->cls[c].spec.kc[k] : 1; q.tap0[k] = pp->ncls ? pp->cls[c].spec.tap0[k] : 0; q.cs[k] = sd[k];
      }
      q.total = g.nstages * g.N16 * 64;
      if (q.total > maxtot) maxtot = q.total;
      if (cnt == PACK_BATCH) { int rc = flush(); if (rc) return rc; }
    }
    handled[i] = 1;
  }
  return flush();
}

size_t patch_wpack_floats(const PatchPlan* p) {      // 16-bit element count / 2
  if (!p->ncls) return (size_t)p->g.nstages * 2 * p->g.N16 * 64 / 2;
  const PatchClass& l = p->cls[p->ncls - 1];
  return l.wp_off + (size_t)l.g.nstages * 2 * l.g.N16 * 64 / 2;
}
static int variant_blocks(const PatchPlan* p, const PGeom& g, const PersVariant& pv) {
  if (pv.on) return pers_blocks(pv.pg, pv.grid);
  if (p->half && &g == &p->g) return p->N * p->gh.nbt * p->gh.nby * p->gh.nbx;
  return p->N * g.nbt * g.nby * g.nbx;
}
// rows of the partial-sum buffer written by one launch sequence (all residue classes)
int patch_blocks(const PatchPlan* p) {
  if (!p->ncls) return variant_blocks(p, p->g, p->pers);
  int n = 0;
  for (int c = 0; c < p->ncls; ++c) n += variant_blocks(p, p->cls[c].g, p->cls[c].pers);
  return n;
}

int patch_pack(const MdConvDesc* d, int dgrad, const PatchPlan* p, const float* w, float* out, hipStream_t s) {
  (void)p;
  unsigned char handled = 0;
  const int rc = patch_pack_batch(1, &d, &dgrad, &w, &out, &handled, s);
  return rc ? rc : (handled ? MD_OK : MD_ERR_UNSUPPORTED);
}

// 64-pixel form (HALF): column groups of p->half_npb channels (or narrower when the layer has few boxes)
static bool pers_can_fuse(const PersVariant& pv);
static PersBwd no_bwd() { PersBwd n; n.yraw = nullptr; n.scale = n.shift = n.mean = n.invstd = nullptr; n.slope = 1.f; return n; }
static int patch_launch_half(const PatchPlan* p, const float* src, const float* ps, const float* psh, float slope, const float* wp,
                             float* dst, float* stat, int accumulate, hipStream_t s, const PersBwd* bw) {
  const PGeom& g = p->gh;
  const PersBwd bwv = bw ? *bw : no_bwd();
  if ((accumulate >> 16) & 1) { if (!p->dgrad || g.pack2) return MD_ERR_UNSUPPORTED; }
  const int boxes = p->N * g.nbt * g.nby * g.nbx;
  int npb = p->half_npb;
  {   // few boxes: narrower column groups (>= 2 tiles each) until the chip is covered
    static const int fill = getenv("MD_PATCH_FILL") ? atoi(getenv("MD_PATCH_FILL")) : 128;
    while (npb > 32 && boxes * md_cdiv(g.N16, npb) < 2 * fill) npb = md_round_up(md_cdiv(npb, 2), 16) < 32 ? 32 : md_round_up(md_cdiv(npb, 2), 16);
  }
  dim3 grid(boxes, md_cdiv(g.N16, npb));
  static const int dbg = getenv("MD_DBG") ? atoi(getenv("MD_DBG")) : 0;
  accumulate = (accumulate & 0x10001) | ((dbg & 0xff) << 8);
  const int nrep = npb / 16;
  const size_t lds = p->ldsh;
#define LAUNCH_HALF(F16_, NR_)                                                                                          \
  do {                                                                                                                  \
    static bool set_ = false;                                                                                           \
    if (!set_) {                                                                                                        \
      if (hipFuncSetAttribute((const void*)k_conv_patch<F16_, false, NR_, false, true>,                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                    \
        return MD_ERR_LAUNCH;                                                                                           \
      set_ = true;                                                                                                      \
    }                                                                                                                   \
    MD_KLAUNCH((k_conv_patch<F16_, false, NR_, false, true>), grid, dim3(256), lds, s, g, src, ps, psh, slope,          \
               (const uint4*)wp, dst, stat, accumulate, npb, bwv);                                                      \
  } while (0)
#define LAUNCH_HALF_NR(F16_)                                                                                            \
  switch (nrep) {                                                                                                       \
    case 2: LAUNCH_HALF(F16_, 2); break;                                                                                \
    case 3: LAUNCH_HALF(F16_, 3); break;                                                                                \
    case 4: LAUNCH_HALF(F16_, 4); break;                                                                                \
    case 5: LAUNCH_HALF(F16_, 5); break;                                                                                \
    case 6: LAUNCH_HALF(F16_, 6); break;                                                                                \
    case 7: LAUNCH_HALF(F16_, 7); break;                                                                                \
    case 8: LAUNCH_HALF(F16_, 8); break;                                                                                \
    default: LAUNCH_HALF(F16_, 9); break;                                                                               \
  }
  if (!p->dgrad) { LAUNCH_HALF_NR(true); } else { LAUNCH_HALF_NR(false); }
  MD_CHECK_LAUNCH();
  return MD_OK;
}

static int patch_launch_one(const PatchPlan* p, const PGeom& g, size_t lds, const PersVariant& pv, const float* src, const float* ps,
                            const float* psh, float slope, const float* wp, float* dst, float* stat, int accumulate,
                            const PersBwd* bw, hipStream_t s) {
  // accumulate: bit 0 = add into dst, bit 16 = src is a pre-split bf16 gradient (data gradient, pack2 excluded)
  if ((accumulate >> 16) & 1) { if (!p->dgrad || g.pack2) return MD_ERR_UNSUPPORTED; }
  if (pv.on && (!bw || pers_can_fuse(pv))) {
    PersBwd none; none.yraw = nullptr; none.scale = none.shift = none.mean = none.invstd = nullptr; none.slope = 1.f;
    return pers_launch(pv.pg, pv.lds, pv.grid, !p->dgrad, src, ps, psh, slope, wp, dst, stat, accumulate, bw ? *bw : none, s);
  }
  if (pv.on && bw && pv.nsl > 0) {            // fused, slice by slice (same rows of the partial buffer, disjoint columns)
    for (int k = 0; k < pv.nsl; ++k) {
      const int rc = pers_launch(pv.sl[k], pv.sl_lds[k], pv.grid, false, src, ps, psh, slope, wp, dst, stat, accumulate, *bw, s);
      if (rc != MD_OK) return rc;
    }
    return MD_OK;
  }
  if (bw && !p->dgrad) return MD_ERR_UNSUPPORTED;
  if (p->half && &g == &p->g) return patch_launch_half(p, src, ps, psh, slope, wp, dst, stat, accumulate, s, bw);
  const PersBwd bwv = bw ? *bw : no_bwd();
  int nchunks = md_cdiv(g.N16, PNREP * 16);
  const int boxes = p->N * g.nbt * g.nby * g.nbx;
  {   // few boxes (the deep, small layers): split the destination channels over more workgroups so every CU gets one
    static const int fill = getenv("MD_PATCH_FILL") ? atoi(getenv("MD_PATCH_FILL")) : 128;     // measured: 64 / 128 / 256 / 512
    int wantc = md_cdiv(fill, boxes);
    if (wantc > g.N16 / 16) wantc = g.N16 / 16;
    if (wantc > nchunks) nchunks = wantc;
  }
  const int npb = md_round_up(md_cdiv(g.N16, nchunks), 16);
  dim3 grid(boxes, md_cdiv(g.N16, npb));
  {   // experiment: MD_LDS_PAD_KB forces a lower occupancy (how much do two workgroups per CU buy?)
    static const int padkb = getenv("MD_LDS_PAD_KB") ? atoi(getenv("MD_LDS_PAD_KB")) : 0;
    if (padkb && lds + (size_t)padkb * 1024 <= 160 * 1024) lds += (size_t)padkb * 1024; else if (padkb) lds = 160 * 1024;
  }
  static const int dbg = getenv("MD_DBG") ? atoi(getenv("MD_DBG")) : 0;
  accumulate = (accumulate & 0x10001) | ((dbg & 0xff) << 8);
  const int nrep = npb / 16;
  // eight waves on the box when LDS leaves room for one workgroup per CU only (then nothing else would overlap)
  static const int w8_env = getenv("MD_PATCH_W8") ? atoi(getenv("MD_PATCH_W8")) : 1;
  const bool w8 = w8_env && !g.strided && nrep >= 2 && (lds > 80 * 1024 || w8_env == 2);
#define LAUNCH_PATCH(F16_, STR_, NR_, W8_)                                                                              \
  do {                                                                                                                  \
    static bool set_ = false;                                                                                           \
    if (!set_) {                                                                                                        \
      if (hipFuncSetAttribute((const void*)k_conv_patch<F16_, STR_, NR_, W8_>,                                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                    \
        return MD_ERR_LAUNCH;                                                                                           \
      set_ = true;                                                                                                      \
    }                                                                                                                   \
    MD_KLAUNCH((k_conv_patch<F16_, STR_, NR_, W8_>), grid, dim3(W8_ ? 512 : 256), lds, s, g, src, ps, psh, slope,       \
               (const uint4*)wp, dst, stat, accumulate, npb, bwv);                                                      \
  } while (0)
#define LAUNCH_PATCH_NR(F16_, STR_, W8_)                                                                                \
  switch (nrep) {                                                                                                       \
    case 1: LAUNCH_PATCH(F16_, STR_, 1, false); break;                                                                  \
    case 2: LAUNCH_PATCH(F16_, STR_, 2, W8_); break;                                                                    \
    case 3: LAUNCH_PATCH(F16_, STR_, 3, W8_); break;                                                                    \
    case 4: LAUNCH_PATCH(F16_, STR_, 4, W8_); break;                                                                    \
    case 5: LAUNCH_PATCH(F16_, STR_, 5, W8_); break;                                                                    \
    case 6: LAUNCH_PATCH(F16_, STR_, 6, W8_); break;                                                                    \
    case 7: LAUNCH_PATCH(F16_, STR_, 7, W8_); break;                                                                    \
    case 8: LAUNCH_PATCH(F16_, STR_, 8, W8_); break;                                                                    \
    default: LAUNCH_PATCH(F16_, STR_, 9, W8_); break;                                                                   \
  }
  if (!p->dgrad) { if (w8) { LAUNCH_PATCH_NR(true, false, true); } else { LAUNCH_PATCH_NR(true, false, false); } }
  else if (!g.strided) { if (w8) { LAUNCH_PATCH_NR(false, false, true); } else { LAUNCH_PATCH_NR(false, false, false); } }
  else { LAUNCH_PATCH_NR(false, true, false); }
  MD_CHECK_LAUNCH();
  return MD_OK;
}

// bw != nullptr: data gradient with the fused BatchNorm-backward reduction (every launch of the plan must have the
// persistent form: patch_can_fuse); `stat` then receives patch_blocks(p) partial rows.
// Does the persistent form of this launch carry the fused reduction?  (instantiated without register spills up to 3 channel tiles,
// and then at most 4 staged items per thread with 3)
static bool pers_can_fuse(const PersVariant& pv) {
  static const int maxn = getenv("MD_FUSE_MAXN16") ? atoi(getenv("MD_FUSE_MAXN16")) : 48;      // experiment: larger tile counts spill
  return pv.on && pv.pg.g.N16 <= maxn && !(pv.pg.g.N16 == 48 && pv.pg.nit > 4) && !(pv.pg.g.N16 > 48 && pv.pg.nit > 4);
}
// A FUSED launch of a geometry whose persistent form cannot fuse runs in the per-box form (k_conv_patch: any tile count -- its
// epilogue walks the tiles one at a time): the 72-channel data gradients of the 64x64 stage then save the separate reduction pass
// over their 198 MB tensors.  MD_FUSE_PATCH=0: no fused reduction in the per-box kernels at all.
static bool fuse_per_box(const PersVariant& pv) {
  static const int patch_off = getenv("MD_FUSE_PATCH") && atoi(getenv("MD_FUSE_PATCH")) == 0;
  // (measured slower, profiles/r03_fuse_fallback.txt: 6.05 against 5.97 ms per step -- the per-box kernel loses more on these
  // layers than the saved reduction pass brings -- so the fallback is off unless MD_FUSE_PERS_FALLBACK=1)
  static const int pers_fallback_on = getenv("MD_FUSE_PERS_FALLBACK") && atoi(getenv("MD_FUSE_PERS_FALLBACK")) == 1;
  if (patch_off) return false;
  if (!pv.on) return true;
  return !pers_can_fuse(pv) && pers_fallback_on;
}
bool patch_can_fuse(const PatchPlan* p) {
  if (!p->dgrad) return false;
  auto ok = [](const PersVariant& pv) { return pers_can_fuse(pv) || (pv.on && pv.nsl > 0) || fuse_per_box(pv); };
  if (!p->ncls) return ok(p->pers);
  for (int c = 0; c < p->ncls; ++c) if (!ok(p->cls[c].pers)) return false;
  return true;
}
// rows of the partial-sum buffer written by a FUSED launch sequence
int patch_fused_blocks(const PatchPlan* p) {
  auto nb = [&](const PGeom& g, const PersVariant& pv) {
    if (pers_can_fuse(pv) || (pv.on && pv.nsl > 0)) return pers_blocks(pv.pg, pv.grid);
    if (p->half && &g == &p->g) return p->N * p->gh.nbt * p->gh.nby * p->gh.nbx;
    return p->N * g.nbt * g.nby * g.nbx;
  };
  if (!p->ncls) return nb(p->g, p->pers);
  int n = 0;
  for (int c = 0; c < p->ncls; ++c) n += nb(p->cls[c].g, p->cls[c].pers);
  return n;
}
int patch_launch(const PatchPlan* p, const float* src, const float* ps, const float* psh, float slope, const float* wp,
                 float* dst, float* stat, int accumulate, hipStream_t s, const PersBwd* bw) {
  if (bw && !patch_can_fuse(p)) return MD_ERR_UNSUPPORTED;
  if (!p->ncls) return patch_launch_one(p, p->g, p->lds, p->pers, src, ps, psh, slope, wp, dst, stat, accumulate, bw, s);
  for (int c = 0; c < p->ncls; ++c) {        // residue classes write disjoint pixels of dst
    const int rc = patch_launch_one(p, p->cls[c].g, p->cls[c].lds, p->cls[c].pers, src, ps, psh, slope, wp + p->cls[c].wp_off, dst,
                                    stat, accumulate, bw, s);
    if (rc) return rc;
    if (stat) {
      const PatchClass& pc = p->cls[c];
      const int nb = (bw && !pers_can_fuse(pc.pers) && !(pc.pers.on && pc.pers.nsl > 0)) ? p->N * pc.g.nbt * pc.g.nby * pc.g.nbx
                                                                                        : variant_blocks(p, pc.g, pc.pers);
      stat += (size_t)nb * 2 * pc.g.Cpd;
    }
  }
  return MD_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// K-streaming split-precision GEMM for the Linears whose source width exceeds the patch kernels' LDS budget (more than 320
// channels: ViViT's patch embedding 768 -> 128 and FeedForward 1024 -> 128, the data gradients of qkv / FF1, the 0D Transformer's
// FF): C[M][ldc] (+)= A[M][K] . W[N16][Kp]^T with A, W, C in fp32 and every product as three fp16 (forward) or bf16 (data
// gradient) MFMAs, exactly as in k_conv_patch.  128 rows x npb (32 / 64 / 128) columns per workgroup, 32 k per stage; wave w owns
// rows 32 w .. 32 w + 31 and all columns.  Both operands are split into hi / lo halves while they are staged (registers ->
// LDS, row pitch 80 B so the per-lane 16-byte fragment reads are conflict free); lane (li, lg) feeds the 8 halves at k = 8 lg ..
// 8 lg + 7 of a stage to both sides of v_mfma_f32_16x16x32, so any fixed k permutation inside the instruction cancels.
#define LB_M 128
#define LB_K 32
#define LB_P 80
#define LB_PD 3           // stages of both operands in flight (registers)
// W8: eight waves on the same 128-row tile (two per SIMD, so one wave's staging arithmetic overlaps the other's MFMAs): waves 0-3 take
// the first half of the column tiles, waves 4-7 the second; a thread then stages 8 floats per operand instead of 16.
template <bool F16, bool W8>
__global__ __launch_bounds__(W8 ? 512 : 256) void k_linear_split(const float* __restrict__ A, int M, int K, const float* __restrict__ W, int Kp,
                                                                int N16, float* __restrict__ C, int ldc, int accumulate, int npb) {
  __shared__ __attribute__((aligned(16))) char lds[4 * LB_M * LB_P];
  char* aH = lds; char* aL = lds + LB_M * LB_P; char* bH = lds + 2 * LB_M * LB_P; char* bL = lds + 3 * LB_M * LB_P;
  constexpr int NQ = W8 ? 2 : 4;                        // float4s per operand and thread and stage
  const int t = threadIdx.x, lane = t & 63, wave = (t >> 6) & 3, half = t >> 8, li = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * LB_M, n0 = blockIdx.y * npb;
  const int ncols = min(npb, N16 - n0), nt = ncols >> 4;
  const int r = W8 ? t >> 2 : t >> 1, h = W8 ? t & 3 : t & 1;      // staging: row r, floats 4 NQ h .. 4 NQ (h + 1) - 1 of the stage
  const bool arow = m0 + r < M, brow = r < ncols;
  // raw buffer loads: a row outside the tile gets the out-of-range offset (reads as 0), so the prefetch is a straight run of loads with
  // no branches in it (host side: both operands < 2 GiB)
  const __amdgpu_buffer_rsrc_t ars = make_rsrc(A, (unsigned)((size_t)M * K * 4)), brs = make_rsrc(W, (unsigned)((size_t)N16 * Kp * 4));
  const unsigned aoff = arow ? (unsigned)(((size_t)(m0 + r) * K + h * 4 * NQ) * 4) : MD_OOB;
  const unsigned boff = brow ? (unsigned)(((size_t)(n0 + r) * Kp + h * 4 * NQ) * 4) : MD_OOB;
  // LB_PD stages of both operands are in flight (registers).  A stage is 12-24 MFMAs per wave, far less than one memory latency, and with
  // a single stage ahead every one of the K / 32 stages waited for its loads (FeedForward 1024 -> 128 at 16.5 k rows: 51 us for a 68 MB
  // read).  The loads are branch-free raw buffer loads and the loop body is LB_PD whole stages, so the compiler waits once per body
  // (vmcnt(0) at the loop head) for loads issued up to LB_PD stages earlier.  Measured (tools/r03_linear_abl.py): 3 stages 47.7 / 48.8 us
  // for the two 1024-wide Linears (1: 51.3 / 50.8); 4 stages cost the second workgroup per CU (142 VGPRs: 61.9 / 63.7), and so did 64-k
  // stages with their 73 KB of LDS (66.9 / 64.5): two resident workgroups matter more than either.
  float4 ra[LB_PD][NQ], rb[LB_PD][NQ];
  auto load = [&](int kb, float4 (&va)[NQ], float4 (&vb)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int k = kb * LB_K + h * 4 * NQ + i * 4;
      va[i] = buf_load4(ars, k < K ? aoff + (unsigned)(kb * LB_K + i * 4) * 4u : MD_OOB);      // (K % 4 == 0: a float4 is inside the row or past it)
      vb[i] = buf_load4(brs, (boff == MD_OOB || kb * LB_K >= Kp) ? MD_OOB : boff + (unsigned)(kb * LB_K + i * 4) * 4u);
    }
  };
  auto store = [&](const float4 (&v)[NQ], char* hi, char* lo) {
#pragma unroll
    for (int q = 0; q < NQ / 2; ++q) {
      const float f[8] = {v[2 * q].x, v[2 * q].y, v[2 * q].z, v[2 * q].w, v[2 * q + 1].x, v[2 * q + 1].y, v[2 * q + 1].z, v[2 * q + 1].w};
      uint4 uh, ul;
      if (F16) split8_f16(f, uh, ul); else split8(f, uh, ul);
      *(uint4*)(hi + r * LB_P + h * 8 * NQ + q * 16) = uh;
      *(uint4*)(lo + r * LB_P + h * 8 * NQ + q * 16) = ul;
    }
  };
  constexpr int NJ = W8 ? 4 : 8;
  const int jper = W8 ? (nt + 1) >> 1 : nt;
  const int jlo = half * jper, jn = min(nt, jlo + jper) - jlo;      // this wave's column tiles: jlo .. jlo + jn - 1
  f32x4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nstages = Kp / LB_K;
#pragma unroll
  for (int u = 0; u < LB_PD; ++u) load(u, ra[u], rb[u]);       // (unconditional: past the last stage the offsets are out of range)
  // (the body is LB_PD whole stages with no exit in between -- stages past the last one multiply zeros -- so that the load / wait
  // pattern is the same on every path into the loop header and the waits cover one stage, not everything in flight)
  for (int kb0 = 0; kb0 < nstages; kb0 += LB_PD) {
#pragma unroll
    for (int u = 0; u < LB_PD; ++u) {
      const int kb = kb0 + u;
      {
        __syncthreads();
        store(ra[u], aH, aL);
        store(rb[u], bH, bL);
        __syncthreads();
        load(kb + LB_PD, ra[u], rb[u]);        // always issued (out of range past the end): the wait counts then cover exactly one stage
        uint4 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *(const uint4*)(aH + (wave * 32 + i * 16 + li) * LB_P + lg * 16);
          al[i] = *(const uint4*)(aL + (wave * 32 + i * 16 + li) * LB_P + lg * 16);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (j < jn) {
            const uint4 bh = *(const uint4*)(bH + ((jlo + j) * 16 + li) * LB_P + lg * 16);
            const uint4 bl = *(const uint4*)(bL + ((jlo + j) * 16 + li) * LB_P + lg * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[i][j] = mma<F16>(ah[i], bh, acc[i][j]);
              acc[i][j] = mma<F16>(ah[i], bl, acc[i][j]);
              acc[i][j] = mma<F16>(al[i], bh, acc[i][j]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j < jn) {
        const int col = n0 + (jlo + j) * 16 + li;
        if (col < ldc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = m0 + wave * 32 + i * 16 + lg * 4 + q;
            if (m < M) {
              float* p = C + (size_t)m * ldc + col;
              float v = acc[i][j][q];
              if (accumulate) v += *p;
              *p = v;
            }
          }
        }
      }
    }
  }
}

// nullptr-free entry used by md_conv_fwd / md_conv_dgrad for 1x1x1 unit-stride geometries the patch kernels do not take
int linear_split_launch(int f16, const float* A, int M, int K, const float* W, int Kp, int N16, float* C, int ldc, int accumulate,
                        hipStream_t s) {
  if (g_exact_fp32.load()) return MD_ERR_UNSUPPORTED;
  static const int off = getenv("MD_LINEAR_SPLIT") && atoi(getenv("MD_LINEAR_SPLIT")) == 0;
  if (off || (K & 3) || (Kp % LB_K) || (N16 & 15)) return MD_ERR_UNSUPPORTED;
  if ((size_t)M * K * 4 >= 0x80000000ull || (size_t)N16 * Kp * 4 >= 0x80000000ull) return MD_ERR_UNSUPPORTED;      // buffer addressing
  int npb = 128;
  // column split until this many workgroups exist.  Round 2 (one stage in flight): 512 -> 4.62 ms, 256 -> 4.38, 130 / 64 -> 4.43 for the
  // ViViT cfg3 captured step; round 3 (three stages in flight): 130 -> 2.541, 256 -> 2.579, 512 -> 2.675 -- re-reading the A rows per
  // column block now costs more than the idle half of the chip
  static const int fill = getenv("MD_LINEAR_FILL") ? atoi(getenv("MD_LINEAR_FILL")) : 128;
  while (md_cdiv(M, LB_M) * md_cdiv(N16, npb) < fill && npb > 32) npb >>= 1;
  const dim3 grid(md_cdiv(M, LB_M), md_cdiv(N16, npb));
  static const int w8 = getenv("MD_LINEAR_W8") ? atoi(getenv("MD_LINEAR_W8")) : 1;
  if (w8 && npb >= 32) {
    if (f16) MD_KLAUNCH((k_linear_split<true, true>), grid, dim3(512), 0, s, A, M, K, W, Kp, N16, C, ldc, accumulate, npb);
    else MD_KLAUNCH((k_linear_split<false, true>), grid, dim3(512), 0, s, A, M, K, W, Kp, N16, C, ldc, accumulate, npb);
  } else {
    if (f16) MD_KLAUNCH((k_linear_split<true, false>), grid, dim3(256), 0, s, A, M, K, W, Kp, N16, C, ldc, accumulate, npb);
    else MD_KLAUNCH((k_linear_split<false, false>), grid, dim3(256), 0, s, A, M, K, W, Kp, N16, C, ldc, accumulate, npb);
  }
  MD_CHECK_LAUNCH();
  return MD_OK;
}
