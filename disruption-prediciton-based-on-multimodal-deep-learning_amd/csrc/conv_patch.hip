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
template <bool F16, bool STRIDED, int NREP, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void k_conv_patch(
    PGeom g, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const uint4* __restrict__ wp, float* __restrict__ dst, float* __restrict__ stat_partial,
    int accumulate, int n_per_blk) {
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
  constexpr int NW = W8 ? (NREP + 1) / 2 : NREP;          // column tiles per wave
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave & 3, wc = wave >> 2;                // rows 32*wr .. +31; column half (W8 only)
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
  if (t < PM) {
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
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    // (W8, odd NREP: the second half's first tile is also the first half's last -- computed twice, written and counted once)
    const bool dup = W8 && (NREP & 1) && wc == 1 && j == 0;
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
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[a][j][r];
        const float vm = v * gw[a][r];
        s1 += vm; s2 = fmaf(vm, v, s2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + prev[a][r]), drs, colok ? goff[a][r] : MD_OOB, j * 64, 0);
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
      for (int w = 0; w < 4; ++w) {
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
static int pitch_for(int C8) {          // smallest 16*(4m+2) >= 16*C8
  int u = C8;
  while ((u & 3) != 2) ++u;
  static const int add = getenv("MD_PITCH_ADD") ? atoi(getenv("MD_PITCH_ADD")) : 0;   // experiments only
  return (u + add) * 16;
}

// Choose the output box (bt,by,bx), <= 128 pixels, minimising (boxes) x (MFMA rows + weighted patch pixels).
// Choose the output box (bt,by,bx), <= 128 pixels, minimising (boxes) x (MFMA rows + weighted patch pixels) subject
// to the LDS budget: `maxP` patch pixels at most, and a 25% penalty once the patch exceeds `softP` pixels (the size
// up to which two workgroups still fit on one CU).  dgrad_s > 1: patch of a strided data gradient (source shrinks).
static bool choose_box(int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw, int dgrad, int maxP, int softP,
                       int* bt, int* by, int* bx, int pm = PM) {
  double best = 1e300;
  bool found = false;
  auto pdim = [&](int b, int k, int s) { return dgrad ? (b + k - 2) / s + 2 : (b - 1) * s + k; };
  for (int t = 1; t <= T && t <= 32; ++t) {
    for (int x = 1; x <= pm; x *= 2) {
      const int xe = x >= W ? W : x;     // powers of two, and the full width
      int ymax = pm / (t * xe);
      if (ymax >= 1) {
        if (ymax > H) ymax = H;
        for (int y = ymax; y >= 1; y = (y > 4 ? y / 2 : y - 1)) {
          const double boxes = (double)md_cdiv(T, t) * md_cdiv(H, y) * md_cdiv(W, xe);
          const long long patch = (long long)pdim(t, kt, st) * pdim(y, kh, sh) * pdim(xe, kw, sw);
          if (patch > maxP) continue;
          double cost = boxes * (pm + 0.6 * (double)patch);
          if (patch > softP) cost *= 1.25;
          if (cost < best) { best = cost; *bt = t; *by = y; *bx = xe; found = true; }
        }
      }
      if (x >= W) break;
    }
  }
  return found;
}

static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; }
static int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

// Residue class of a strided data gradient (see patch_classes): destination pixels i = s*j + c per dimension receive
// only the taps tap0, tap0 + s, ... (kc of them); over the class grid j this is a unit-stride correlation.
struct ClassSpec { int c[3], tap0[3], kc[3], e[3]; };

// pers_wgs > 0: geometry for the persistent kernel (conv_pers.hip) at that many workgroups per CU -- the LDS budget then
// holds the whole packed weight operand instead of one streamed stage.
static bool patch_build(const MdConvDesc* d, int dgrad, PGeom* out, size_t* lds_bytes, const ClassSpec* cls = nullptr,
                        int pers_wgs = 0) {
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
  const int npb_ = md_round_up(md_cdiv(g.N16, nchunks_), 16);
  // LDS: everything but the patch
  if (pers_wgs && (sdg || g.N16 > 128 || g.N16 < 32)) return false;
  const int pm = PM;
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
            pers_wgs ? "(persistent)" : "", dgrad ? (cls ? "dgrad-class" : "dgrad") : "fwd", d->Cin, d->Cout, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, g.Td, g.Hd,
            g.Wd, g.bt, g.by, g.bx, g.pt, g.py, g.px, g.P, g.C8, g.nstages, off);
  *out = g; *lds_bytes = off;
  return true;
}

#include <atomic>
static std::atomic<int> g_pers_grid{0};     // > 0: test override of the persistent kernels' grid (md_set_pers_grid)
extern "C" int md_set_pers_grid(int n) { return g_pers_grid.exchange(n > 0 ? n : 0); }
struct PersVariant { bool on; PersGeom pg; size_t lds; int grid; };           // persistent-kernel form of the same launch
struct PatchClass { PGeom g; size_t lds; size_t wp_off; ClassSpec spec; PersVariant pers; };     // wp_off: floats into the packed operand
struct PatchPlan { PGeom g; size_t lds; int N; int dgrad; int ncls; PatchClass cls[8]; PersVariant pers; };

// Persistent form (conv_pers.hip) of one launch, when the geometry qualifies: the packed weights fit in LDS beside the
// patch, the box is nearly full, and there are at least as many boxes as resident workgroups.  The packed-weight format
// (K order, N16) does not depend on the box, so both forms read the same operand.
static void pers_try(const MdConvDesc* d, int dgrad, const ClassSpec* cls, const PGeom& classic, PersVariant* pv) {
  pv->on = false;
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
  if (!pp && patch_build(d, dgrad, &g, &lds)) {
    pp = new PatchPlan(); pp->g = g; pp->lds = lds; pp->N = d->N; pp->dgrad = dgrad; pp->ncls = 0;
    pers_try(d, dgrad, nullptr, g, &pp->pers);
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

static int patch_launch_one(const PatchPlan* p, const PGeom& g, size_t lds, const PersVariant& pv, const float* src, const float* ps,
                            const float* psh, float slope, const float* wp, float* dst, float* stat, int accumulate,
                            const PersBwd* bw, hipStream_t s) {
  // accumulate: bit 0 = add into dst, bit 16 = src is a pre-split bf16 gradient (data gradient, pack2 excluded)
  if ((accumulate >> 16) & 1) { if (!p->dgrad || g.pack2) return MD_ERR_UNSUPPORTED; }
  if (pv.on) {
    PersBwd none; none.yraw = nullptr; none.scale = none.shift = none.mean = none.invstd = nullptr; none.slope = 1.f;
    return pers_launch(pv.pg, pv.lds, pv.grid, !p->dgrad, src, ps, psh, slope, wp, dst, stat, accumulate, bw ? *bw : none, s);
  }
  if (bw) return MD_ERR_UNSUPPORTED;
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
               (const uint4*)wp, dst, stat, accumulate, npb);                                                           \
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
bool patch_can_fuse(const PatchPlan* p) {
  if (!p->dgrad) return false;
  // instantiated without register spills: up to 3 channel tiles (and then at most 4 staged items per thread with 3)
  static const int maxn = getenv("MD_FUSE_MAXN16") ? atoi(getenv("MD_FUSE_MAXN16")) : 48;      // experiment: larger tile counts spill
  auto ok = [](const PersVariant& pv) { return pv.on && pv.pg.g.N16 <= maxn && !(pv.pg.g.N16 == 48 && pv.pg.nit > 4) && !(pv.pg.g.N16 > 48 && pv.pg.nit > 4); };
  if (!p->ncls) return ok(p->pers);
  for (int c = 0; c < p->ncls; ++c) if (!ok(p->cls[c].pers)) return false;
  return true;
}
int patch_launch(const PatchPlan* p, const float* src, const float* ps, const float* psh, float slope, const float* wp,
                 float* dst, float* stat, int accumulate, hipStream_t s, const PersBwd* bw) {
  if (bw && !patch_can_fuse(p)) return MD_ERR_UNSUPPORTED;
  if (!p->ncls) return patch_launch_one(p, p->g, p->lds, p->pers, src, ps, psh, slope, wp, dst, stat, accumulate, bw, s);
  for (int c = 0; c < p->ncls; ++c) {        // residue classes write disjoint pixels of dst
    const int rc = patch_launch_one(p, p->cls[c].g, p->cls[c].lds, p->cls[c].pers, src, ps, psh, slope, wp + p->cls[c].wp_off, dst,
                                    stat, accumulate, bw, s);
    if (rc) return rc;
    if (stat) stat += (size_t)variant_blocks(p, p->cls[c].g, p->cls[c].pers) * 2 * p->cls[c].g.Cpd;
  }
  return MD_OK;
}

// Pointer-based variant (branches around the edge cases) kept for the weight-gradient kernels, where it measured faster.
// Stage `npix` pixels x `C8` 8-channel chunks of a channels-last fp32 tensor into an LDS image
// [pixel][C8 chunks] (pixel pitch `pitch` bytes; hi array at img, lo array at img + lo_off).
// sG[pixel] = global pixel index or -1 (outside the tensor: zeros = the convolution's zero padding).
// Channels c0 .. c0 + 4*cvalid4 are read (cvalid4 = valid float4 units from c0); chunks past that are zero.
// With `prologue`, value = leaky(x*scale[c] + shift[c]) ("BN-on-read") before the bf16 hi/lo split.
template <bool F16>
__device__ __forceinline__ void stage_image_ptr(const float* __restrict__ src, int Cpitch, int c0, int cvalid4,
                                            const int* sG, int npix, int C8, unsigned magic, char* img, int pitch,
                                            int lo_off, bool prologue, const float* sScale, const float* sShift,
                                            float pslope, int t, int presplit_c8 = 0) {
  // presplit_c8 > 0: src is a pre-split bf16 gradient with that many 32-byte chunks per pixel ([pixel][chunk]{hi | lo});
  // c0 (a multiple of 8) selects the first chunk; staging is then a plain copy.
  const int total = npix * C8;
  for (int base = 0; base < total; base += 256 * 4) {
    float4 va[4], vb[4];
    int pix[4], c8s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int item = base + u * 256 + t;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      pix[u] = -1; c8s[u] = 0;
      if (item < total) {
        const int pixel = magic ? (int)__umulhi((unsigned)item, magic) : item;
        const int c8 = item - pixel * C8;
        pix[u] = pixel | 0x20000000; c8s[u] = c8;        // 0x2..: nothing loaded (stays zero, no prologue)
        const int gp = sG[pixel];
        if (gp >= 0 && c8 * 2 < cvalid4) {
          const float* s = presplit_c8 ? src + ((size_t)gp * presplit_c8 + (c0 >> 3) + c8) * 8 : src + (size_t)gp * Cpitch + c0 + c8 * 8;
          va[u] = *(const float4*)s;
          pix[u] = pixel;
          if (presplit_c8 || c8 * 2 + 1 < cvalid4) vb[u] = *(const float4*)(s + 4);
          else pix[u] |= 0x40000000;                       // upper half of the chunk is channel padding
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pix[u] == -1) continue;
      const int pixel = pix[u] & 0x0fffffff;
      const bool inside = !(pix[u] & 0x20000000);
      const bool half = (pix[u] & 0x40000000) != 0;
      float v[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      if (prologue && inside) {
        const float* sc = sScale + c0 + c8s[u] * 8; const float* sh = sShift + c0 + c8s[u] * 8;
        bn_leaky8(v, *(const f32x4*)sc, *(const f32x4*)(sc + 4), *(const f32x4*)sh, *(const f32x4*)(sh + 4), pslope);
        if (half) { v[4] = v[5] = v[6] = v[7] = 0.f; }
      }
      uint4 hi, lo;
      if (presplit_c8) { hi = __builtin_bit_cast(uint4, va[u]); lo = __builtin_bit_cast(uint4, vb[u]); }
      else if (F16) split8_f16(v, hi, lo); else split8(v, hi, lo);
      char* d = img + pixel * pitch + c8s[u] * 16;
      *(uint4*)d = hi;
      *(uint4*)(d + lo_off) = lo;
    }
  }
}


// ================================================================================================
// Weight gradient of a unit-stride convolution, split-bf16 arithmetic, LDS-resident operands.
//   dW[k = (tap, cin)][cout] = sum over pixels  X[pixel + tap][cin] * dY[pixel][cout]
// A workgroup walks a slice of the output boxes.  Per box it stages the X patch (with halo, BN-on-read) and the
// dY box once, both as [pixel][channel] bf16 hi|lo images.  The reduction axis of the MFMA is the pixel axis,
// which is the SLOW axis of both images, so both operands are fetched with ds_read_b64_tr_b16: a 16-lane group
// reads 4 pixels x 16 channels and each lane receives its channel for those 4 pixels -- the transpose is free
// and no second copy of either operand exists.  Wave w owns `ktw` 16-row k-tiles x `nrep` 16-col n-tiles
// (<= 25 accumulator tiles); partial results go to a per-slice slab and k_wgrad_reduce sums the slabs in a fixed
// order straight into the reference's (Cout,Cin,kt,kh,kw) layout (deterministic, no atomics).
// ================================================================================================
#define WKT 5
#define WNR 5
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WGeom {
  int Ti, Hi, Wi, Cpi;        // X
  int To, Ho, Wo, Cpo;        // dY
  int kh, kw, khw, taps;
  int org_t, org_h, org_w;
  int st, sh, sw;
  int bt, by, bx, byx, nbt, nby, nbx;
  int pt, py, px, pyx, P;
  int C8i, ppitch, lo_off;    // X patch image (C8i even: whole 16-channel k-tiles)
  int NC, ypitch, ylo_off;    // dY image: NC = 2*nrep chunks per row
  int KT, nkt;                // 16-channel k-tiles per tap, total
  int ktw, nrep, nkg, nng;
  int nboxes, boxes_per_wg;
  int N16;
  unsigned magicC8, magicNC;
  int off_y, off_rows, off_pixg, off_scale;
  int pack2, pk_shift, pk_kw;   // pixel-pair reinterpretation (see wgrad_build); then kw = k-tiles per filter row
  int tapw;                     // X patch bytes between successive values of the kw index
  unsigned x_bytes, y_bytes;    // tensor sizes for the buffer descriptors (< 2 GiB)
  int xpitch, xc0;              // floats per X pixel in memory and first channel read (a channel slice of a wider tensor: the
                                // chunked weight gradient of wide Linears; = Cpi, 0 otherwise)
  int dw_cin, dw_c0;            // the dW tensor's full Cin and the slice's first channel
};

__device__ __forceinline__ bf16x8 tr_read2(const char* p0, const char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
  s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int KTW, int NREP>
__global__ __launch_bounds__(256) void k_wgrad_patch(
    WGeom g, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const float* __restrict__ dy, float* __restrict__ slab, int dbg) {
  const int ysplit = (dbg >> 16) & 1;            // dY is a pre-split bf16 gradient ([pixel][Cpo/8 chunks]{hi | lo})
  const int xsplit = (dbg >> 17) & 1;            // X is a pre-activated, pre-split bf16 tensor ([pixel][ceil(Cpi/8) chunks]{hi | lo})
  dbg &= 0xffff;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* sP = sm;
  char* sY = sm + g.off_y;
  int2* sR = (int2*)(sm + g.off_rows);          // [PM] {X patch byte offset of the row, dY global pixel or -1}
  int* sG = (int*)(sm + g.off_pixg);
  int* sGY = sG + ((g.P + 3) & ~3);             // [PM] dY pixel table for stage_image
  float* sScale = (float*)(sm + g.off_scale);
  float* sShift = sScale + PMAXC;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int lq = li >> 2, lp = li & 3;
  const int kg = blockIdx.y / g.nng, ng = blockIdx.y - kg * g.nng;
  const int n0 = ng * g.nrep * 16;              // first dY channel of this workgroup
  const int kt0 = (kg * 4 + wave) * g.ktw;      // first k-tile of this wave
  const bool prologue = pscale != nullptr;
  if (prologue) for (int c = t; c < g.Cpi; c += 256) { const int cs = g.pack2 ? (c & 3) : g.xc0 + c; sScale[c] = pscale[cs]; sShift[c] = pshift[cs]; }

  f32x4 acc[KTW][NREP];
#pragma unroll
  for (int a = 0; a < KTW; ++a)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // byte offset inside the X patch of each of this wave's k-tiles: tap pixel offset + 32 B per 16 channels
  int koff[KTW];
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int kt = kt0 + a;
    int o = 0;
    if (kt < g.nkt) {
      const int tap = kt / g.KT; const int c16 = kt - tap * g.KT;
      const int dt = tap / g.khw; const int r = tap - dt * g.khw;
      const int dyy = r / g.kw; const int dxx = r - dyy * g.kw;
      o = (dt * g.py + dyy) * g.px * g.ppitch + dxx * g.tapw + c16 * 32;
    }
    koff[a] = o;
  }

  const int box_beg = blockIdx.x * g.boxes_per_wg;
  const int box_end = min(g.nboxes, box_beg + g.boxes_per_wg);
  const int ycv4 = max(0, min(g.NC * 2, (g.Cpo - n0) >> 2));      // valid float4 units of a dY row from n0
  for (int box = box_beg; box < box_end; ++box) {
    int b = box;
    const int xb = b % g.nbx; b /= g.nbx;
    const int yb = b % g.nby; b /= g.nby;
    const int tb = b % g.nbt; const int n = b / g.nbt;
    const int t0 = tb * g.bt, y0 = yb * g.by, x0 = xb * g.bx;
    __syncthreads();          // previous box fully consumed
    for (int p = t; p < g.P; p += 256) {
      const int ppt = p / g.pyx; const int r = p - ppt * g.pyx;
      const int ppy = r / g.px; const int ppx = r - ppy * g.px;
      const int st = t0 * g.st + g.org_t + ppt, sy = y0 * g.sh + g.org_h + ppy, sx = x0 * g.sw + g.org_w + ppx;
      const bool v = ((unsigned)st < (unsigned)g.Ti) && ((unsigned)sy < (unsigned)g.Hi) && ((unsigned)sx < (unsigned)g.Wi);
      sG[p] = v ? ((n * g.Ti + st) * g.Hi + sy) * g.Wi + sx : -1;
    }
    if (t < PM) {
      const int rt = t / g.byx; const int r = t - rt * g.byx;
      const int ry = r / g.bx; const int rx = r - ry * g.bx;
      const bool v = (rt < g.bt) && (t0 + rt < g.To) && (y0 + ry < g.Ho) && (x0 + rx < g.Wo);
      int2 ri;
      ri.x = v ? ((rt * g.st * g.py + ry * g.sh) * g.px + rx * g.sw) * g.ppitch : 0;
      ri.y = v ? ((n * g.To + t0 + rt) * g.Ho + y0 + ry) * g.Wo + x0 + rx : -1;
      sR[t] = ri;
      sGY[t] = ri.y;
    }
    __syncthreads();
    if (!(dbg & 1)) stage_image_ptr<false>(src, g.xpitch, g.xc0, g.Cpi >> 2, sG, g.P, g.C8i, g.magicC8, sP, g.ppitch, g.lo_off, prologue, sScale - g.xc0,
                sShift - g.xc0, pslope, t, xsplit ? ((g.Cpi + 7) >> 3) : 0);
    if (!(dbg & 2)) stage_image_ptr<false>(dy, g.Cpo, n0, ycv4, sGY, PM, g.NC, g.magicNC, sY, g.ypitch, g.ylo_off, false, nullptr, nullptr, 1.f, t,
                                           ysplit ? (g.Cpo >> 3) : 0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < ((dbg & 4) ? 0 : 4); ++s) {
      // The MFMA's 32 reduction slots of this step are pixels; lane group lg takes pixels {4lg..4lg+3} and
      // {16+4lg..16+4lg+3} of the step (any assignment works as long as A and B agree).  A half-wave's first read
      // then covers 8 CONSECUTIVE pixels: with the odd-multiple-of-32-byte pixel pitch that is conflict free.
      const int r0 = s * 32 + lg * 4 + lq;
      const int xa = sR[r0].x + lp * 8, xb2 = sR[r0 + 16].x + lp * 8;
      const int ya = r0 * g.ypitch + lp * 8, yb2 = (r0 + 16) * g.ypitch + lp * 8;
      bf16x8 bh[NREP], bl[NREP];
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        bh[j] = tr_read2(sY + ya + j * 32, sY + yb2 + j * 32);
        bl[j] = tr_read2(sY + g.ylo_off + ya + j * 32, sY + g.ylo_off + yb2 + j * 32);
      }
#pragma unroll
      for (int a = 0; a < KTW; ++a) {
        const bf16x8 ah = tr_read2(sP + xa + koff[a], sP + xb2 + koff[a]);
        const bf16x8 al = tr_read2(sP + g.lo_off + xa + koff[a], sP + g.lo_off + xb2 + koff[a]);
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[a][j], 0, 0, 0);
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[a][j], 0, 0, 0);
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[a][j], 0, 0, 0);
        }
      }
    }
  }

  // ---- slab[slice][k16 row][N16]: D rows = k index (4*lg + reg), cols = cout (li)
  float* out = slab + (size_t)blockIdx.x * g.nkt * 16 * g.N16;
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int kt = kt0 + a;
    if (kt < g.nkt) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        {
          const int col = n0 + j * 16 + li;
          if (col < g.N16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(kt * 16 + lg * 4 + r) * g.N16 + col] = acc[a][j][r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant: the global loads of box i+1 (X patch and dY rows, up to 6+5 32-byte items per
// thread) are issued into registers BEFORE the MFMA phase of box i and committed (BN-on-read, hi/lo split, LDS
// write) after it, so HBM latency hides behind the matrix work.  One workgroup per CU is all the accumulator
// budget allows for this kernel, so nothing else would overlap the loads.  All per-item index arithmetic that
// does not depend on the box is done once.
// ------------------------------------------------------------------------------------------------
#define WPF_X 7
#define WPF_Y 5
// W8: eight waves (512 threads) per box -- wave = (k-tile group wk, column half wn); two waves per SIMD overlap each
// other's commit (VALU), request and MFMA phases, which one 4-wave workgroup per CU (384-428 VGPRs) cannot.
template <int KTW, int NREP, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void k_wgrad_patch_pf(
    WGeom g, const float* __restrict__ src, const float* __restrict__ pscale, const float* __restrict__ pshift,
    float pslope, const float* __restrict__ dy, float* __restrict__ slab, int fmt) {
  // fmt bit 0 (ysplit): dY is a pre-split bf16 gradient ([pixel][Cpo/8 chunks]{hi 8 x bf16 | lo 8 x bf16}): its commit is a plain copy;
  // bit 1 (xsplit): X is the pre-activated, pre-split bf16 copy of the unit's input ([pixel][ceil(Cpi/8) chunks]{hi | lo}, written
  // by md_bn_act_split during the forward pass): no BatchNorm-on-read, no split, a plain copy as well
  const int ysplit = fmt & 1, xsplit = (fmt >> 1) & 1;
  const int xp = xsplit ? ((g.Cpi + 7) >> 3) * 8 : g.xpitch;         // floats per X pixel in memory
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* sP = sm;
  char* sY = sm + g.off_y;
  int* sRx = (int*)(sm + g.off_rows);           // [PM] X patch byte offset of each output row (box independent)
  float* sScale = (float*)(sm + g.off_scale);
  float* sShift = sScale + PMAXC;

  constexpr int NT = W8 ? 512 : 256;
  constexpr int NX = W8 ? 4 : WPF_X, NY = W8 ? 3 : WPF_Y;    // 32-byte items per thread: X patch, dY rows
  constexpr int NW = W8 ? (NREP + 1) / 2 : NREP;                   // column tiles per wave
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6) & 3;     // k-tile group of this wave
  const int wn = __builtin_amdgcn_readfirstlane(t >> 8);           // column half (W8)
  const int j0 = wn ? NREP - NW : 0;                               // first column tile (odd NREP: the halves overlap by one)
  const int li = lane & 15, lg = lane >> 4;
  const int lq = li >> 2, lp = li & 3;
  const int kg = blockIdx.y / g.nng, ng = blockIdx.y - kg * g.nng;
  const int n0 = ng * g.nrep * 16;
  const int kt0 = (kg * 4 + wave) * g.ktw;
  const bool prologue = pscale != nullptr;
  if (prologue) for (int c = t; c < g.Cpi; c += NT) { const int cs = g.pack2 ? (c & 3) : g.xc0 + c; sScale[c] = pscale[cs]; sShift[c] = pshift[cs]; }
  if (t < PM) {
    const int rt = t / g.byx; const int r = t - rt * g.byx;
    const int ry = r / g.bx; const int rx = r - ry * g.bx;
    sRx[t] = (rt < g.bt) ? ((rt * g.st * g.py + ry * g.sh) * g.px + rx * g.sw) * g.ppitch : 0;
  }

  f32x4 acc[KTW][NW];
#pragma unroll
  for (int a = 0; a < KTW; ++a)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int koff[KTW];
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int kt = kt0 + a;
    int o = 0;
    if (kt < g.nkt) {
      const int tap = kt / g.KT; const int c16 = kt - tap * g.KT;
      const int dt = tap / g.khw; const int r = tap - dt * g.khw;
      const int dyy = r / g.kw; const int dxx = r - dyy * g.kw;
      o = (dt * g.py + dyy) * g.px * g.ppitch + dxx * g.tapw + c16 * 32;
    }
    koff[a] = o;
  }

  // ---- box-independent decode of this thread's items
  const int totX = g.P * g.C8i, totY = PM * g.NC;
  const int xcv4 = g.Cpi >> 2;
  const int ycv4 = max(0, min(g.NC * 2, (g.Cpo - n0) >> 2));
  int xloc[NX], xdst[NX];      // packed (ppt | ppy<<6 | ppx<<15 | c8<<24), LDS byte offset (or -1: no item)
  int yloc[NY], ydst[NY];      // packed (rt | ry<<6 | rx<<15 | c<<24)
  int xrel[NX], yrel[NY];      // element offset of the item relative to the box origin
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int item = u * NT + t;
    xdst[u] = -1; xloc[u] = 0; xrel[u] = 0;
    if (item < totX) {
      const int pixel = g.magicC8 ? (int)__umulhi((unsigned)item, g.magicC8) : item;
      const int c8 = item - pixel * g.C8i;
      const int ppt = pixel / g.pyx; const int r = pixel - ppt * g.pyx;
      const int ppy = r / g.px; const int ppx = r - ppy * g.px;
      xloc[u] = ppt | (ppy << 6) | (ppx << 15) | (c8 << 24);
      xdst[u] = pixel * g.ppitch + c8 * 16;
      xrel[u] = ((ppt * g.Hi + ppy) * g.Wi + ppx) * xp + g.xc0 + c8 * 8;
    }
  }
#pragma unroll
  for (int u = 0; u < NY; ++u) {
    const int item = u * NT + t;
    ydst[u] = -1; yloc[u] = 0; yrel[u] = 0;
    if (item < totY) {
      const int row = g.magicNC ? (int)__umulhi((unsigned)item, g.magicNC) : item;
      const int c = item - row * g.NC;
      const int rt = row / g.byx; const int r = row - rt * g.byx;
      const int ry = r / g.bx; const int rx = r - ry * g.bx;
      yloc[u] = rt | (ry << 6) | (rx << 15) | (c << 24);
      yrel[u] = ((rt * g.Ho + ry) * g.Wo + rx) * g.Cpo + n0 + c * 8;      // (same element offset in both formats: 32 B per 8 channels)
      ydst[u] = (rt < g.bt) ? row * g.ypitch + c * 16 : -2;      // -2: row outside the box -> zeros
    }
  }

  float4 xa_[NX], xb_[NX], ya_[NY], yb_[NY];
  int xfl = 0;                // per item 2 bits: bit0 = loaded (inside the tensor), bit1 = upper half is padding
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(src, xsplit ? (unsigned)((unsigned long long)g.x_bytes / (unsigned)g.xpitch * (unsigned)xp) : g.x_bytes),
                               yrs = make_rsrc(dy, g.y_bytes);
  // Box being requested (scalars): clip index, output-box origin, input-patch origin, element offsets of the origins.
  int q_t0 = 0, q_y0 = 0, q_x0 = 0, q_ot = 0, q_oh = 0, q_ow = 0, q_xbase = 0, q_ybase = 0, q_live = 0;
  auto aim = [&](int box, bool live) {
    int b = box;
    const int xb = b % g.nbx; b /= g.nbx;
    const int yb = b % g.nby; b /= g.nby;
    const int tb = b % g.nbt; const int n = b / g.nbt;
    q_t0 = tb * g.bt; q_y0 = yb * g.by; q_x0 = xb * g.bx;
    q_ot = q_t0 * g.st + g.org_t; q_oh = q_y0 * g.sh + g.org_h; q_ow = q_x0 * g.sw + g.org_w;
    q_xbase = (((n * g.Ti + q_ot) * g.Hi + q_oh) * g.Wi + q_ow) * xp;
    q_ybase = (((n * g.To + q_t0) * g.Ho + q_y0) * g.Wo + q_x0) * g.Cpo;
    q_live = live ? 1 : 0;
  };
  // Branch-free request of one item (out-of-range offset -> zeros, no memory traffic), so that the requests can be
  // spread between the MFMA groups of the box in flight.
  auto issue_x = [&](int u) {
    const int st = q_ot + (xloc[u] & 63), sy = q_oh + ((xloc[u] >> 6) & 511), sx = q_ow + ((xloc[u] >> 15) & 511);
    const int c8 = (xloc[u] >> 24) & 255;
    const bool in = q_live && xdst[u] >= 0 && ((unsigned)st < (unsigned)g.Ti) && ((unsigned)sy < (unsigned)g.Hi) &&
                    ((unsigned)sx < (unsigned)g.Wi) && c8 * 2 < xcv4;
    const bool up = in && (xsplit || c8 * 2 + 1 < xcv4);
    const unsigned off = (unsigned)(q_xbase + xrel[u]) * 4u;
    xa_[u] = buf_load4(xrs, in ? off : MD_OOB);
    xb_[u] = buf_load4(xrs, up ? off + 16u : MD_OOB);
    xfl = (xfl & ~(3 << (2 * u))) | ((in ? 1 : 0) << (2 * u)) | ((in && !up ? 2 : 0) << (2 * u));
  };
  auto issue_y = [&](int u) {
    const int ot_ = q_t0 + (yloc[u] & 63), oy_ = q_y0 + ((yloc[u] >> 6) & 511), ox_ = q_x0 + ((yloc[u] >> 15) & 511);
    const int c = (yloc[u] >> 24) & 255;
    const bool in = q_live && ydst[u] >= 0 && ot_ < g.To && oy_ < g.Ho && ox_ < g.Wo && c * 2 < ycv4;
    const unsigned off = (unsigned)(q_ybase + yrel[u]) * 4u;
    ya_[u] = buf_load4(yrs, in ? off : MD_OOB);
    yb_[u] = buf_load4(yrs, (in && (ysplit || c * 2 + 1 < ycv4)) ? off + 16u : MD_OOB);
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      if (xdst[u] >= 0) {
        float v[8] = {xa_[u].x, xa_[u].y, xa_[u].z, xa_[u].w, xb_[u].x, xb_[u].y, xb_[u].z, xb_[u].w};
        if (prologue && ((xfl >> (2 * u)) & 1)) {
          const int c8 = (xloc[u] >> 24) & 255;
          const float* sc = sScale + c8 * 8; const float* sh = sShift + c8 * 8;
          bn_leaky8(v, *(const f32x4*)sc, *(const f32x4*)(sc + 4), *(const f32x4*)sh, *(const f32x4*)(sh + 4), pslope);
          if ((xfl >> (2 * u)) & 2) { v[4] = v[5] = v[6] = v[7] = 0.f; }
        }
        uint4 hi, lo;
        if (xsplit) { hi = __builtin_bit_cast(uint4, xa_[u]); lo = __builtin_bit_cast(uint4, xb_[u]); }
        else split8(v, hi, lo);
        *(uint4*)(sP + xdst[u]) = hi;
        *(uint4*)(sP + g.lo_off + xdst[u]) = lo;
      }
    }
#pragma unroll
    for (int u = 0; u < NY; ++u) {
      if (ydst[u] != -1) {
        const int item = u * NT + t;
        const int row = g.magicNC ? (int)__umulhi((unsigned)item, g.magicNC) : item;
        const int off = row * g.ypitch + ((yloc[u] >> 24) & 255) * 16;
        const float v[8] = {ya_[u].x, ya_[u].y, ya_[u].z, ya_[u].w, yb_[u].x, yb_[u].y, yb_[u].z, yb_[u].w};
        uint4 hi, lo;
        if (ysplit) { hi = __builtin_bit_cast(uint4, ya_[u]); lo = __builtin_bit_cast(uint4, yb_[u]); }
        else split8(v, hi, lo);
        *(uint4*)(sY + off) = hi;
        *(uint4*)(sY + g.ylo_off + off) = lo;
      }
    }
  };

  const int box_beg = blockIdx.x * g.boxes_per_wg;
  const int box_end = min(g.nboxes, box_beg + g.boxes_per_wg);
#ifdef MD_PHASE_TIMING
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tc0 = clock64(), tc1;
#define PH(i) do { tc1 = clock64(); tph[i] += tc1 - tc0; tc0 = tc1; } while (0)
#else
#define PH(i)
#endif
  if (box_beg < box_end) {
    aim(box_beg, true);
#pragma unroll
    for (int u = 0; u < NX; ++u) issue_x(u);
#pragma unroll
    for (int u = 0; u < NY; ++u) issue_y(u);
  }
  PH(0);
  for (int box = box_beg; box < box_end; ++box) {
    __syncthreads();          // previous box fully consumed (first iteration: tables / scale in LDS)
    PH(1);
    commit();
    PH(2);
    __syncthreads();
    PH(3);
    aim(min(box + 1, box_end - 1), box + 1 < box_end);      // next box: requested item by item between the MFMA groups
    PH(4);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r0 = s * 32 + lg * 4 + lq;
      const int xa = sRx[r0] + lp * 8, xb2 = sRx[r0 + 16] + lp * 8;
      const int ya = r0 * g.ypitch + lp * 8, yb2 = (r0 + 16) * g.ypitch + lp * 8;
      bf16x8 bh[NW], bl[NW];
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        bh[j] = tr_read2(sY + ya + (j0 + j) * 32, sY + yb2 + (j0 + j) * 32);
        bl[j] = tr_read2(sY + g.ylo_off + ya + (j0 + j) * 32, sY + g.ylo_off + yb2 + (j0 + j) * 32);
      }
#pragma unroll
      for (int a = 0; a < KTW; ++a) {
        {
          const int grp = s * KTW + a;          // 4*KTW >= 12 groups for the 7 + 5 items
          if (grp < NX) issue_x(grp);
          else if (grp - NX < NY) issue_y(grp - NX);
        }
        const bf16x8 ah = tr_read2(sP + xa + koff[a], sP + xb2 + koff[a]);
        const bf16x8 al = tr_read2(sP + g.lo_off + xa + koff[a], sP + g.lo_off + xb2 + koff[a]);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[a][j], 0, 0, 0);
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[a][j], 0, 0, 0);
          acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[a][j], 0, 0, 0);
        }
        if (s * KTW + a < NX + NY) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // this group's two requests,
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NW, 0);                                 // then its MFMAs
        // (the scheduler fills groups bottom-up and would otherwise sink the requests to the END of the MFMA phase,
        // where their latency is exposed at the next commit)
        if (s * KTW + a == NX + NY - 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    PH(5);
  }
#ifdef MD_PHASE_TIMING
  if (blockIdx.x == 7 && blockIdx.y == 0 && (t == 0 || t == 192))
    printf("wgrad_pf t=%d boxes=%d issue0=%lld bar1=%lld commit=%lld bar2=%lld issue=%lld mfma=%lld\n", t, box_end - box_beg, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5]);
#endif

  float* out = slab + (size_t)blockIdx.x * g.nkt * 16 * g.N16;
#pragma unroll
  for (int a = 0; a < KTW; ++a) {
    const int kt = kt0 + a;
    if (kt < g.nkt) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int col = n0 + (j0 + j) * 16 + li;
        const bool dup = W8 && (NREP & 1) && wn == 1 && j == 0;      // written by the first column half
        if (col < g.N16 && !dup) {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)(kt * 16 + lg * 4 + r) * g.N16 + col] = acc[a][j][r];
        }
      }
    }
  }
}

// dw[cout][cin][tap] = sum_slices slab[slice][(tap*KT + cin/16)*16 + cin%16][cout]   (fixed order)
// Block = 64 outputs x 4 slice groups: slice group q sums slices q, q+4, ... with four independent chains.
// pack2: k-tile kt = (filter row, txg), row i of the tile = real tap dx = 4 txg + (i >> 2) + shift, channel i & 3.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int nslices, int nkt, int KT, int N16,
                                                      int Cout, int Cin, int taps, float* __restrict__ dw, int pack2,
                                                      int kwt, int kw_real, int shift, int taps_real, int dw_cin, int dw_c0) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;                         // over [k16 rows][N16], cout fastest
  const int rows = nkt * 16;
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
    if (pack2) {
      const int row = kt / kwt, txg = kt - row * kwt, i = krow & 15;
      const int dx = 4 * txg + (i >> 2) + shift, c = i & 3;
      if (co < Cout && c < Cin && dx >= 0 && dx < kw_real) dw[((size_t)co * Cin + c) * taps_real + row * kw_real + dx] = s;
    } else {
      const int tap = kt / KT; const int c = (kt - tap * KT) * 16 + (krow & 15);
      if (co < Cout && c < Cin) dw[((size_t)co * dw_cin + dw_c0 + c) * taps + tap] = s;
    }
  }
}

struct WgradPlan { WGeom g; size_t lds; int nslices; bool w8; };

// xpitch / xc0: X is the channel slice [xc0, xc0 + Cin) of a tensor with xpitch floats per pixel (0: X is the whole tensor);
// dW then goes to columns [xc0, xc0 + Cin) of a (Cout, dw_cin, taps) tensor
static bool wgrad_build(const MdConvDesc* d, WGeom* out, size_t* lds_bytes, int* nslices, int xpitch = 0, int xc0 = 0, int dw_cin = 0) {
  WGeom g;
  g.st = d->st; g.sh = d->sh; g.sw = d->sw;
  g.Ti = d->Ti; g.Hi = d->Hi; g.Wi = d->Wi; g.Cpi = md_cpad(d->Cin);
  g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo; g.Cpo = md_cpad(d->Cout);
  g.kh = d->kh; g.kw = d->kw; g.khw = d->kh * d->kw; g.taps = d->kt * g.khw;
  g.org_t = -d->pt; g.org_h = -d->ph; g.org_w = -d->pw;
  g.KT = md_cdiv(d->Cin, 16); g.nkt = g.taps * g.KT;
  g.C8i = 2 * g.KT;
  g.ppitch = pitch_for(g.C8i);
  g.tapw = g.ppitch;
  // Pixel-pair reinterpretation (see patch_build): the <= 4-channel, W-stride-2 input becomes [.., Wi/2][8] pairs with
  // unit W-stride.  The patch is a dense array of 16-byte pairs, so one 16-channel k-tile spans two neighbouring pairs
  // = 4 real pixels x 4 channels; kw counts k-tiles per filter row and the patch covers 2*kw pairs in x.
  g.pack2 = 0; g.pk_shift = 0; g.pk_kw = d->kw;
  static const int no_pack2 = getenv("MD_PACK2") && atoi(getenv("MD_PACK2")) == 0;
  int kw_patch = d->kw;
  if (!no_pack2 && g.Cpi == 4 && d->sw == 2 && (d->Wi & 1) == 0 && d->kw >= 2) {
    const int lo = -((d->pw + 1) / 2);
    const int num = d->kw - 1 - d->pw;
    const int hi = num >= 0 ? num / 2 : -((-num + 1) / 2);
    g.pack2 = 1; g.pk_shift = d->pw + 2 * lo;
    g.kw = md_cdiv(hi - lo + 1, 2); g.khw = g.kh * g.kw; g.taps = d->kt * g.khw;
    g.KT = 1; g.nkt = g.taps;
    g.org_w = lo; g.sw = 1; g.Wi = d->Wi / 2; g.Cpi = 8; g.C8i = 1;
    g.ppitch = 16; g.tapw = 32;
    kw_patch = 2 * g.kw;
  }
  g.N16 = md_round_up(d->Cout, 16);
  const int NT = g.N16 / 16;
  static const int wnr = getenv("MD_WGRAD_NR") ? atoi(getenv("MD_WGRAD_NR")) : WNR;     // column tiles per workgroup (<= WNR)
  g.nng = md_cdiv(NT, wnr < 1 ? 1 : (wnr > WNR ? WNR : wnr)); g.nrep = md_cdiv(NT, g.nng);
  { const int q = md_cdiv(g.nkt, 4); g.ktw = q <= 3 ? 3 : (q == 4 ? 4 : WKT); }     // instantiated: 3, 4 or 5 k-tiles per wave
  g.nkg = md_cdiv(g.nkt, 4 * g.ktw);
  g.NC = 2 * g.nrep;
  g.ypitch = pitch_for(g.NC);
  g.ylo_off = PM * g.ypitch;
  g.magicC8 = g.C8i == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)g.C8i) + 1u;
  g.magicNC = g.NC == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)g.NC) + 1u;
  if (g.Cpi > PMAXC) return false;
  {
    g.xpitch = xpitch ? xpitch : g.Cpi; g.xc0 = xpitch ? xc0 : 0;
    g.dw_cin = xpitch ? dw_cin : d->Cin; g.dw_c0 = xpitch ? xc0 : 0;
    if (xpitch && g.pack2) return false;
    const unsigned long long xb = (unsigned long long)d->N * g.Ti * g.Hi * g.Wi * g.xpitch * 4ull;
    const unsigned long long yb = (unsigned long long)d->N * g.To * g.Ho * g.Wo * g.Cpo * 4ull;
    g.x_bytes = xb < 0x80000000ull ? (unsigned)xb : 0u; g.y_bytes = yb < 0x80000000ull ? (unsigned)yb : 0u;   // 0: no buffer addressing
  }
  const size_t cap = 160 * 1024;
  const size_t fixed = (size_t)2 * g.ylo_off + (size_t)PM * 8 + (size_t)PM * 4 + 2 * PMAXC * 4 + 1024;
  if (fixed + 4096 > cap) return false;
  const long long per_px = (long long)2 * g.ppitch + 4;
  long long maxP = (long long)(cap - fixed) / per_px - 2;
  long long softP = (long long)(cap / 2 - fixed) / per_px - 2;
  const long long idx_cap = 65535 / g.C8i;
  if (maxP > idx_cap) maxP = idx_cap;
  if (softP < 1) softP = 1;
  if (maxP < 1) return false;
  if (!choose_box(g.To, g.Ho, g.Wo, d->kt, d->kh, kw_patch, g.st, g.sh, g.sw, 0, (int)maxP, (int)softP, &g.bt, &g.by, &g.bx))
    return false;
  g.byx = g.by * g.bx;
  g.nbt = md_cdiv(g.To, g.bt); g.nby = md_cdiv(g.Ho, g.by); g.nbx = md_cdiv(g.Wo, g.bx);
  g.pt = (g.bt - 1) * g.st + d->kt; g.py = (g.by - 1) * g.sh + d->kh; g.px = (g.bx - 1) * g.sw + kw_patch;
  g.pyx = g.py * g.px; g.P = g.pt * g.pyx;
  g.lo_off = (g.P * g.ppitch + 15) & ~15;
  g.nboxes = d->N * g.nbt * g.nby * g.nbx;
  // enough workgroups to fill the chip, few enough that the slabs stay small
  int want = md_cdiv(512, g.nkg * g.nng);
  if (want > g.nboxes) want = g.nboxes;
  if (want < 1) want = 1;
  g.boxes_per_wg = md_cdiv(g.nboxes, want);
  *nslices = md_cdiv(g.nboxes, g.boxes_per_wg);
  size_t off = (size_t)2 * g.lo_off;
  g.off_y = (int)off; off += (size_t)2 * g.ylo_off;
  g.off_rows = (int)off; off += (size_t)PM * 8;
  g.off_pixg = (int)off; off += (size_t)(((g.P + 3) & ~3) + PM) * 4;
  off = (off + 15) & ~(size_t)15;
  g.off_scale = (int)off; off += (size_t)2 * PMAXC * 4;
  if (off > cap) return false;
  if (getenv("MD_PLAN_PRINT"))
    fprintf(stderr, "wgrad %d->%d k%d%d%d s%d%d%d out %dx%dx%d: box %dx%dx%d patch %dx%dx%d=%d C8i=%d nkt=%d ktw=%d nkg=%d nrep=%d nng=%d lds=%zu\n",
            d->Cin, d->Cout, d->kt, d->kh, d->kw, d->st, d->sh, d->sw, g.To, g.Ho, g.Wo, g.bt, g.by, g.bx, g.pt, g.py, g.px, g.P,
            g.C8i, g.nkt, g.ktw, g.nkg, g.nrep, g.nng, off);
  *out = g; *lds_bytes = off;
  return true;
}

static bool wgrad_use_pf(const WGeom& g);
// Few k-tiles (<= 8: a Linear with 128 input channels, a 3-tap convolution over 32) on the non-prefetching kernel: two k-tiles per
// wave instead of three (4 waves x 3 = 12 slots for 8 tiles left a third of the MFMAs multiplying padding).  The prefetching
// kernel interleaves its 12 load groups with 4 x KTW >= 12 MFMA groups and keeps KTW >= 3.
static void wgrad_narrow_k(WGeom* g) {
  static const int off = getenv("MD_WGRAD_KTW2") && atoi(getenv("MD_WGRAD_KTW2")) == 0;
  if (!off && !wgrad_use_pf(*g) && md_cdiv(g->nkt, 4) <= 2) { g->ktw = 2; g->nkg = md_cdiv(g->nkt, 8); }
}

static bool wgrad_use_pf(const WGeom& g) {
  static const int no_pf = getenv("MD_WGRAD_PF") && atoi(getenv("MD_WGRAD_PF")) == 0;
  static const int dbg = getenv("MD_DBG") ? atoi(getenv("MD_DBG")) : 0;
  return !no_pf && !dbg && g.x_bytes && g.y_bytes && g.P * g.C8i <= WPF_X * 256 && PM * g.NC <= WPF_Y * 256 && g.pt < 64 && g.py < 512 && g.px < 512 &&
         g.bt < 64;
}
template <int KT, int NR>
static const void* wgrad_kernel_of(bool pf) { return pf ? (const void*)k_wgrad_patch_pf<KT, NR> : (const void*)k_wgrad_patch<KT, NR>; }
template <int KT>
static const void* wgrad_kernel_nr(int nrep, bool pf) {
  switch (nrep) {
    case 1: return wgrad_kernel_of<KT, 1>(pf);
    case 2: return wgrad_kernel_of<KT, 2>(pf);
    case 3: return wgrad_kernel_of<KT, 3>(pf);
    case 4: return wgrad_kernel_of<KT, 4>(pf);
    default: return wgrad_kernel_of<KT, 5>(pf);
  }
}
// Workgroups of this kernel that fit on one CU (registers and LDS); 2 when the runtime cannot say (no device).
static int wgrad_wgs_per_cu(const WGeom& g, size_t lds) {
  const bool pf = wgrad_use_pf(g);
  const void* k = g.ktw == 2 ? wgrad_kernel_nr<2>(g.nrep, false)
                  : g.ktw == 3 ? wgrad_kernel_nr<3>(g.nrep, pf) : g.ktw == 4 ? wgrad_kernel_nr<4>(g.nrep, pf) : wgrad_kernel_nr<5>(g.nrep, pf);
  int nb = 0;
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds) != hipSuccess || nb < 1) {
    (void)hipGetLastError();
    return 2;
  }
  return nb > 2 ? 2 : nb;
}

thread_local int g_wgrad_beside = 0;
const WgradPlan* wgrad_lookup(const MdConvDesc* d, int xpitch, int xc0, int dw_cin) {
  if (g_exact_fp32.load()) return nullptr;
  static const int dis = getenv("MD_PATCH_WGRAD") && atoi(getenv("MD_PATCH_WGRAD")) == 0;
  if (dis) return nullptr;
  static std::mutex mu;
  static std::map<std::array<int, 22>, WgradPlan*> cache;
  const int beside = g_wgrad_beside > 0 ? 1 : 0;
  std::array<int, 22> key = {d->N, d->Ti, d->Hi, d->Wi, d->Cin, d->To, d->Ho, d->Wo, d->Cout, d->kt, d->kh, d->kw,
                             d->st, d->sh, d->sw, d->pt, d->ph, d->pw, xpitch, xc0, dw_cin, beside};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  WgradPlan* wp = nullptr;
  WGeom g; size_t lds = 0; int ns = 0;
  if (wgrad_build(d, &g, &lds, &ns, xpitch, xc0, dw_cin)) {
    wgrad_narrow_k(&g);
    // one slice (= one slab of partial sums) per resident workgroup: a single full round on the chip, and no more
    // slab traffic than that needs
    // CUs to occupy.  The executor runs weight gradients on a side stream next to the BatchNorm-backward / data-gradient
    // chain (plan.hip); leaving part of the chip to that chain measured best at 160 of 256 (96: 992, 128: 1088,
    // 160: 1097, 192: 1085, 256: 1068 clips/s); with the side stream switched off the kernel takes the whole chip.
    static const int side_off = getenv("MD_WGRAD_STREAM") && atoi(getenv("MD_WGRAD_STREAM")) == 0;
    // Outside the executor (the composable models: a weight gradient runs alone on its stream) the kernel takes the whole chip as
    // well: ViViT cfg3 captured step 4.23 -> 4.14 ms at 256 (512: 4.21, 1024: 4.24).
    static const int fill_env = getenv("MD_WGRAD_FILL") ? atoi(getenv("MD_WGRAD_FILL")) : 0;
    const int fill = fill_env ? fill_env : ((side_off || !beside) ? 256 : 160);
    const int occ = wgrad_wgs_per_cu(g, lds);
    // eight-wave form of the prefetching kernel where only one 4-wave workgroup would fit a CU
    static const int w8_env = getenv("MD_WGRAD_W8") ? atoi(getenv("MD_WGRAD_W8")) : 1;
    const bool w8 = w8_env && occ == 1 && wgrad_use_pf(g) && g.nrep >= 2;
    // the stem's weight gradient (pixel-pair form) is the LAST kernel of the backward pass: nothing runs beside it any more, so it
    // is sized for the whole chip
    static const int tail_full = !(getenv("MD_WGRAD_TAIL_FULL") && atoi(getenv("MD_WGRAD_TAIL_FULL")) == 0);
    const int fill_eff = (g.pack2 && tail_full) ? 256 : fill;
    int want = md_cdiv(fill_eff * occ, g.nkg * g.nng);
    if (want > g.nboxes) want = g.nboxes;
    if (want < 1) want = 1;
    g.boxes_per_wg = md_cdiv(g.nboxes, want);
    ns = md_cdiv(g.nboxes, g.boxes_per_wg);
    wp = new WgradPlan(); wp->g = g; wp->lds = lds; wp->nslices = ns; wp->w8 = w8;
  }
  cache[key] = wp;
  return wp;
}

bool wgrad_plan_xsplit_ok(const WgradPlan* p) { return !p->g.pack2 && p->g.xpitch == p->g.Cpi; }
size_t wgrad_patch_workspace_floats(const WgradPlan* p) { return (size_t)p->nslices * p->g.nkt * 16 * p->g.N16; }

int wgrad_patch_launch(const WgradPlan* p, const MdConvDesc* d, const float* src, const float* ps, const float* psh,
                       float slope, const float* dy, float* dw, float* slab, hipStream_t s, int ysplit, int xsplit) {
  const WGeom& g = p->g;
  if (ysplit && (g.Cpo & 7)) return MD_ERR_UNSUPPORTED;
  if (xsplit && (g.pack2 || g.xpitch != g.Cpi || ps)) return MD_ERR_UNSUPPORTED;      // whole-tensor, already activated X only
  const int fmt = (ysplit ? 1 : 0) | (xsplit ? 2 : 0);
  static const int dbg_env = getenv("MD_DBG") ? atoi(getenv("MD_DBG")) : 0;
  const int dbg = (dbg_env & 0xffff) | (ysplit ? 0x10000 : 0) | (xsplit ? 0x20000 : 0);
  dim3 grid(p->nslices, g.nkg * g.nng);
  const bool pf = wgrad_use_pf(g);
#define LAUNCH_WG(KT_, NR_)                                                                                             \
  do {                                                                                                                  \
    static bool set_ = false;                                                                                           \
    if (!set_) {                                                                                                        \
      if (hipFuncSetAttribute((const void*)k_wgrad_patch<KT_, NR_>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                              160 * 1024) != hipSuccess ||                                                              \
          hipFuncSetAttribute((const void*)k_wgrad_patch_pf<KT_, NR_>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                              160 * 1024) != hipSuccess) return MD_ERR_LAUNCH;                                          \
      set_ = true;                                                                                                      \
    }                                                                                                                   \
    if (pf && p->w8 && NR_ >= 2) {                                                                                      \
      static bool set8_ = false;                                                                                        \
      if (!set8_) {                                                                                                     \
        if (hipFuncSetAttribute((const void*)k_wgrad_patch_pf<KT_, NR_, true>,                                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                  \
          return MD_ERR_LAUNCH;                                                                                         \
        set8_ = true;                                                                                                   \
      }                                                                                                                 \
      MD_KLAUNCH((k_wgrad_patch_pf<KT_, NR_, true>), grid, dim3(512), p->lds, s, g, src, ps, psh, slope, dy, slab, fmt); \
    } else if (pf)                                                                                                      \
      MD_KLAUNCH((k_wgrad_patch_pf<KT_, NR_>), grid, dim3(256), p->lds, s, g, src, ps, psh, slope, dy, slab, fmt); \
    else                                                                                                                \
      MD_KLAUNCH((k_wgrad_patch<KT_, NR_>), grid, dim3(256), p->lds, s, g, src, ps, psh, slope, dy, slab, dbg); \
  } while (0)
#define LAUNCH_WG_NR(KT_)                                                                                               \
  switch (g.nrep) {                                                                                                     \
    case 1: LAUNCH_WG(KT_, 1); break;                                                                                   \
    case 2: LAUNCH_WG(KT_, 2); break;                                                                                   \
    case 3: LAUNCH_WG(KT_, 3); break;                                                                                   \
    case 4: LAUNCH_WG(KT_, 4); break;                                                                                   \
    default: LAUNCH_WG(KT_, 5); break;                                                                                  \
  }
  if (g.ktw == 2) { LAUNCH_WG_NR(2); } else if (g.ktw == 3) { LAUNCH_WG_NR(3); } else if (g.ktw == 4) { LAUNCH_WG_NR(4); } else { LAUNCH_WG_NR(5); }
  MD_CHECK_LAUNCH();
  const int total = g.nkt * 16 * g.N16;
  MD_KLAUNCH(k_wgrad_reduce, dim3(md_cdiv(total, 64)), dim3(256), 0, s, slab, p->nslices, g.nkt, g.KT, g.N16, d->Cout,
             d->Cin, g.taps, dw, g.pack2, g.kw, g.pk_kw, g.pk_shift, d->kt * d->kh * d->kw, g.dw_cin, g.dw_c0);
  MD_CHECK_LAUNCH();
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
  const float* ap = A + (size_t)(m0 + r) * K + h * 4 * NQ;
  const float* bp = W + (size_t)(n0 + r) * Kp + h * 4 * NQ;
  float4 ra[NQ], rb[NQ];
  auto load = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int k = kb * LB_K + h * 4 * NQ + i * 4;
      ra[i] = (arow && k < K) ? *(const float4*)(ap + kb * LB_K + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = brow ? *(const float4*)(bp + kb * LB_K + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
  load(0);
  for (int kb = 0; kb < nstages; ++kb) {
    __syncthreads();
    store(ra, aH, aL);
    store(rb, bH, bL);
    __syncthreads();
    if (kb + 1 < nstages) load(kb + 1);
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
  int npb = 128;
  static const int fill = getenv("MD_LINEAR_FILL") ? atoi(getenv("MD_LINEAR_FILL")) : 256;   // ViViT cfg3 captured step: 512 -> 4.62 ms, 256 -> 4.38, 130 / 64 -> 4.43
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
