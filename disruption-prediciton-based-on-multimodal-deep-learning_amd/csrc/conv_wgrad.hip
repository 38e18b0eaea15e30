// Weight gradient of the LDS-patch convolutions (split-bf16 arithmetic, both operands resident in LDS, transposing LDS
// reads).  Split from conv_patch.hip in round 3 (one translation unit per kernel family keeps the rebuild of one family
// under a minute).
#include "common.h"
#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>
#include <array>
#include <atomic>
#include "patch_common.h"

extern "C" int md_get_exact_fp32(void);

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

struct WgradPlan { WGeom g; size_t lds; int nslices; bool w8; const Wgrad2Plan* v2; bool transposed = false; };

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
  // k-groups first (<= 4 * WKT tiles each), then the tiles per wave from what ONE group holds: 27 tiles are 2 groups of 4 x 4
  // slots (round 2 took WKT = 5 whenever there was more than one group: 40 slots, a third of the MFMAs on padding)
  g.nkg = md_cdiv(g.nkt, 4 * WKT);
  { const int q = md_cdiv(md_cdiv(g.nkt, g.nkg), 4); g.ktw = q <= 3 ? 3 : (q == 4 ? 4 : WKT); }     // instantiated: 3, 4 or 5 k-tiles per wave
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
// Test hook: 1 pins the first form (k_wgrad_patch*) for geometries the second form (conv_wgrad2.hip) would take -- the pre-split
// operand formats exist only in the first form, and their tests compare BITS against the fp32-format launch of the same kernel.
static std::atomic<int> g_wgrad_first_form{0};
extern "C" int md_set_wgrad_form(int first_form_only) { return g_wgrad_first_form.exchange(first_form_only ? 1 : 0); }
const WgradPlan* wgrad_lookup(const MdConvDesc* d, int xpitch, int xc0, int dw_cin) {
  if (md_get_exact_fp32()) return nullptr;
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
    static const int side_off = !(getenv("MD_WGRAD_STREAM") && atoi(getenv("MD_WGRAD_STREAM")) == 1);     // round 3: one stream unless asked
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
    wp->v2 = wgrad2_lookup(d, beside, xpitch, xc0, dw_cin);
    wp->transposed = !xpitch && dw_cin < 0;
  }
  cache[key] = wp;
  return wp;
}

bool wgrad_plan_second_form(const WgradPlan* p) { return p->v2 != nullptr && !g_wgrad_first_form.load(); }
bool wgrad_plan_xsplit_ok(const WgradPlan* p) { return !p->g.pack2 && p->g.xpitch == p->g.Cpi; }
size_t wgrad_patch_workspace_floats(const WgradPlan* p) {
  const size_t a = (size_t)p->nslices * p->g.nkt * 16 * p->g.N16, b = p->v2 ? wgrad2_workspace_floats(p->v2) : 0;
  return a > b ? a : b;
}

int wgrad_partial(const MdConvDesc* d, const float* src, const float* ps, const float* psh, float slope, const float* dy, float* dw,
                  float* slab, hipStream_t s, WgradPending* out) {
  const WgradPlan* p = wgrad_lookup(d);
  if (!p || !p->v2 || g_wgrad_first_form.load()) return MD_ERR_UNSUPPORTED;
  const int rc = wgrad2_launch_partial(p->v2, src, ps, psh, slope, dy, slab, s);
  if (rc != MD_OK) return rc;
  out->p = p->v2; out->Cout = d->Cout; out->Cin = d->Cin; out->slab = slab; out->dw = dw;
  return MD_OK;
}

int wgrad_patch_launch(const WgradPlan* p, const MdConvDesc* d, const float* src, const float* ps, const float* psh,
                       float slope, const float* dy, float* dw, float* slab, hipStream_t s, int ysplit, int xsplit) {
  const WGeom& g = p->g;
  if (p->v2 && !ysplit && !xsplit && !g_wgrad_first_form.load()) return wgrad2_launch(p->v2, d, src, ps, psh, slope, dy, dw, slab, s);
  if (p->transposed) return MD_ERR_UNSUPPORTED;          // only the second form writes the transposed result
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


