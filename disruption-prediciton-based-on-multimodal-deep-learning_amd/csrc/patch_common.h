// Shared device helpers and the geometry record of the LDS-patch convolution kernels (conv_patch.hip: one box per
// workgroup, weights streamed per stage; conv_pers.hip: persistent workgroups, weights resident in LDS).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PM 128            // rows (output pixels) per workgroup
#define PNREP 9           // up to 144 destination channels per workgroup
#define PB_PITCH 160      // B tile row pitch (bytes): 8 chunks of 16 B + pad, (160/16) % 4 == 2
#define PMAXC 320

struct PGeom {
  int Ts, Hs, Ws, Cps;      // source dims, channel pitch (floats)
  int Td, Hd, Wd, Cpd;      // destination dims, channel pitch (floats)
  int kh, kw, khw, taps;
  int org_t, org_h, org_w;  // source coordinate = box origin * stride + org + patch coordinate
  int st, sh, sw;           // stride of the (forward) convolution; 1 for the data gradient
  int strided;              // 1: data gradient of a strided convolution (per-tap divisibility test)
  int dst_, dsh_, dsw_;     // the convolution's stride (strided data gradient)
  int lt, lh, lw, oddmask;  // log2 strides; packed mask of the low bits that must be zero
  int kt, padt, padh, padw;
  int zero_off;             // byte offset of the all-zero pixel
  int bt, by, bx, byx;      // output box
  int nbt, nby, nbx;        // boxes per clip
  int pt, py, px, pyx, P;   // patch dims
  int C8;                   // 8-channel chunks per pixel in LDS
  int ppitch;               // LDS bytes per patch pixel (hi array); lo array follows at lo_off
  int lo_off;
  int Kc8;                  // taps * C8
  int nstages;              // ceil(Kc8 / 8): 64 k per stage
  int N16;
  unsigned magicC8;         // floor(2^32 / C8) + 1 : exact item / C8 for item < 2^16 (0 when the divisor is 1)
  unsigned m_pyx, m_px, m_byx, m_bx, m_khw, m_kw;   // same for the table decodes (all indices < 2^16)
  unsigned src_bytes, dst_bytes;                      // tensor sizes for the buffer descriptors (< 2 GiB)
  int Tdf, Hdf, Wdf;        // full destination dims; destination coordinate = box coordinate * dm + dp
  int dmt, dmh, dmw, dpt, dph, dpw;   // (one residue class of a strided data gradient writes a strided subset)
  int off_b, off_koffs, off_rows, off_pixg, off_scale;   // LDS byte offsets
  int pack2, pk_shift, pk_kw; // pixel-pair reinterpretation of a <=4-channel, W-stride-2 input (see patch_build)
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// fp16 split: hi = fp16(x) (11 significant bits), lo = fp16(x - hi) -> 22 bits, i.e. fp32-level products with
// three MFMAs.  fp16 subnormals are kept (HIP kernels run with float_denorm_mode_16_64 = preserve), so the
// absolute error of hi+lo is <= max(2^-23 |x|, 2^-25); |x| must stay below 65504 (activations and weights do).
__device__ __forceinline__ void split8_f16(const float* v, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pk_f16(v[2 * i], v[2 * i + 1]);
    const f16x2 hv = __builtin_bit_cast(f16x2, h[i]);
    l[i] = pk_f16(v[2 * i] - (float)hv[0], v[2 * i + 1] - (float)hv[1]);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// split 8 floats into packed bf16 hi and lo (4 dwords each)
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
    const float h0 = __builtin_bit_cast(float, h[i] << 16), h1 = __builtin_bit_cast(float, h[i] & 0xffff0000u);
    l[i] = pk_bf16(v[2 * i] - h0, v[2 * i + 1] - h1);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}


// BatchNorm-on-read + LeakyReLU of one 8-channel chunk in packed fp32 math (v_pk_fma_f32, v_pk_mul_f32): per element
// the same operations as  md_leaky(fmaf(v, scale, shift), slope) , so the result is bit-identical.  md_leaky decides
// between max(x, slope x) (0 <= slope <= 1) and the select form per ELEMENT (8 compares + 17 selects per chunk in the
// generated code); here the decision is taken once per call on the wave-uniform slope.
__device__ __forceinline__ void bn_leaky8(float* v, f32x4 sc0, f32x4 sc1, f32x4 sh0, f32x4 sh1, float slope) {
  const f32x2 s[4] = {{sc0[0], sc0[1]}, {sc0[2], sc0[3]}, {sc1[0], sc1[1]}, {sc1[2], sc1[3]}};
  const f32x2 h[4] = {{sh0[0], sh0[1]}, {sh0[2], sh0[3]}, {sh1[0], sh1[1]}, {sh1[2], sh1[3]}};
  const f32x2 sl = {slope, slope};
  if (slope >= 0.f && slope <= 1.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 x = {v[2 * i], v[2 * i + 1]};
      const f32x2 pre = __builtin_elementwise_fma(x, s[i], h[i]);
      const f32x2 r = __builtin_elementwise_max(pre, pre * sl);
      v[2 * i] = r[0]; v[2 * i + 1] = r[1];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 x = {v[2 * i], v[2 * i + 1]};
      const f32x2 pre = __builtin_elementwise_fma(x, s[i], h[i]);
      const f32x2 t = pre * sl;
      v[2 * i] = pre[0] > 0.f ? pre[0] : t[0]; v[2 * i + 1] = pre[1] > 0.f ? pre[1] : t[1];
    }
  }
}
__device__ __forceinline__ void zero8(float* v) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
}

// x / d for 0 <= x, x * d < 2^32, with magic = floor(2^32 / d) + 1 (0 encodes d == 1)
__device__ __forceinline__ int mdiv(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }

// Raw buffer descriptor over a whole tensor (< 2 GiB): loads at an out-of-range offset return zeros and stores there
// are dropped, so edge handling needs no branches and no 64-bit address arithmetic.  MD_OOB = 2 GiB is out of range
// for every tensor we accept and cannot wrap around when an instruction offset is added.
#define MD_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  // (bit_cast of the whole vector: indexing the builtin's result element-wise is miscompiled into one dword load)
  const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(f[0], f[1], f[2], f[3]);
}

template <bool F16>
__device__ __forceinline__ f32x4 mma(uint4 a, uint4 b, f32x4 c) {
  if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}


// ---- persistent form (conv_pers.hip)
#define PERS_MAXI 8       // 32-byte patch items per thread (registers)
struct PersGeom {
  PGeom g;
  int nboxes;
  int nsteps;               // k32 steps = ceil(Kc8 / 4)
  int bpitch;               // resident weight row pitch in bytes, (bpitch / 16) % 4 == 2
  int off_bres;             // LDS byte offsets
  int off_k, off_row, off_sc, off_red, off_bn, off_sync;
  int nit;                  // patch items per thread
  int patch_bytes;          // LDS bytes of one team's patch ([pixel][hi | lo | pad])
  int sc_stride, bn_stride; // floats between the scale | shift (| mean | invstd) rows in LDS
  unsigned m_nbx, m_nby, m_nbt;
  int n0, N16w;             // column slice: this launch computes destination channels n0 .. n0 + g.N16 of N16w packed weight rows
};
// fused BatchNorm-backward reduction (data gradient only); yraw == nullptr: off
struct PersBwd {
  const float* yraw;        // raw output of the unit whose activation this data gradient differentiates
  const float* scale; const float* shift; const float* mean; const float* invstd;
  float slope;
};
bool pers_finish(PersGeom* pg, size_t* lds_bytes, int* grid);
// workgroups launched: each walks two box streams
static inline int pers_blocks(const PersGeom& pg, int grid) { const int need = (pg.nboxes + 1) / 2; return grid < need ? grid : need; }
size_t pers_bres_bytes(int Kc8, int N16);
size_t pers_fixed_bytes(int Cps, int N16);
int pers_launch(const PersGeom& pg, size_t lds, int grid, bool f16, const float* src, const float* ps, const float* psh, float slope,
                const float* wp, float* dst, float* stat, int accumulate, const PersBwd& bw, hipStream_t s);

// ---- host-side helpers shared by conv_patch.hip and conv_wgrad.hip
#include <cstdlib>
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
