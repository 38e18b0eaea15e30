// Shared declarations for the gfx950 kernels of the disruption-predictor hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <tuple>
#include <type_traits>
#include <utility>
#include "../../include/mi355x_disrupt.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MD_ABI_VERSION 1

static inline int md_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t md_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int md_round_up(int a, int b) { return md_cdiv(a, b) * b; }

// Measurement hook (plan.hip::ProfScope, the roofline leg of bench.py): while a scope is open on this thread every launch asks it for a
// (start, stop) event pair and goes through hipExtLaunchKernel, which stamps the two events from the dispatch packet's own completion
// signal.  Bracketing launches with hipEventRecord instead cost the host ~35 us per record (13 ms per sampled step, the GPU queue ran
// dry) and put two barrier packets around every kernel.
struct MdProfHook { void* ctx; int (*acquire)(void* ctx, hipEvent_t* start, hipEvent_t* stop); };
extern thread_local MdProfHook g_md_prof_hook;

template <typename... P, size_t... I>
static inline void md_klaunch_fill(std::tuple<P...>& vals, void** ptrs, std::index_sequence<I...>) {
  ((ptrs[I] = (void*)&std::get<I>(vals)), ...);
}
template <typename... P, typename... A>
static inline void md_klaunch(void (*kernel)(P...), dim3 g, dim3 b, size_t sh, hipStream_t s, A&&... args) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_md_prof_hook.ctx && g_md_prof_hook.acquire(g_md_prof_hook.ctx, &e0, &e1)) {
    std::tuple<std::remove_cv_t<P>...> vals{static_cast<P>(args)...};
    void* ptrs[sizeof...(P) + 1];
    md_klaunch_fill(vals, ptrs, std::index_sequence_for<P...>{});
    (void)hipExtLaunchKernel((const void*)kernel, g, b, ptrs, sh, s, e0, e1, 0);
  } else {
    kernel<<<g, b, sh, s>>>(static_cast<P>(args)...);
  }
}
// hipGetLastError() reports the last error of ANY earlier HIP call on this thread (e.g. a benign failure inside the
// host framework's start-up), so the sticky state is cleared right before every launch and read right after it.
#define MD_KLAUNCH(...)              \
  do {                               \
    (void)hipGetLastError();         \
    md_klaunch(__VA_ARGS__);         \
  } while (0)

#define MD_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return MD_ERR_LAUNCH;           \
  } while (0)

// Geometry of one implicit GEMM: rows = destination pixels, K = taps x source channels.
// A source coordinate along dimension d is  (o*sn + off + sign*tap) / sd  and is valid only
// when the division is exact and the result lies inside the source tensor.  Forward conv:
// sn = stride, sd = 1, off = -pad, sign = +1.  Data gradient: sn = 1, sd = stride, off = +pad,
// sign = -1 (a gather formulation of the transposed conv, so no scatter / atomics are needed).
struct Geom {
  int Ti, Hi, Wi, Cpi;   // source tensor [N][Ti][Hi][Wi][Cpi]
  int To, Ho, Wo, Cpo;   // destination tensor [N][To][Ho][Wo][Cpo]
  int kh, kw, khw;
  int sn_t, sn_h, sn_w;
  int sd_t, sd_h, sd_w;
  int off_t, off_h, off_w;
  int sign;
  int M;        // destination rows
  int Kc;       // K / 4 (16-byte chunks of the flat K axis)
  int Kp;       // row pitch of the packed weights (floats, multiple of 32)
  int nstages;  // Kp / 32
  int N16;      // packed weight rows (destination channels rounded up to 16)
};

// LeakyReLU: v > 0 ? v : v*slope.  For 0 <= slope <= 1 that is max(v, v*slope) (two VALU ops, no compare/select);
// bit-identical including v = +-0 and NaN propagation of the multiply.
__device__ __forceinline__ float md_leaky(float v, float slope) {
  const float s = v * slope;
  return (slope >= 0.f && slope <= 1.f) ? fmaxf(v, s) : (v > 0.f ? v : s);
}
__device__ __forceinline__ float md_dleaky(float pre, float slope) { return pre > 0.f ? 1.f : slope; }

// ---- unit-stride patch kernel (conv_patch.hip); used by the dispatchers in conv_gemm.hip
struct PGeom;
int linear_split_launch(int f16, const float* A, int M, int K, const float* W, int Kp, int N16, float* C, int ldc, int accumulate,
                        hipStream_t s);   // MD_ERR_UNSUPPORTED when the caller should use k_conv_gemm
struct PatchPlan;   // cached PGeom + LDS size for one (descriptor, direction)
const PatchPlan* patch_lookup(const MdConvDesc* d, int dgrad);     // nullptr when the geometry does not qualify
size_t patch_wpack_floats(const PatchPlan* p);
int patch_blocks(const PatchPlan* p);
int patch_pack(const MdConvDesc* d, int dgrad, const PatchPlan* p, const float* w, float* out, hipStream_t s);
struct PersBwd;     // fused BatchNorm-backward reduction of a data gradient (patch_common.h)
int patch_launch(const PatchPlan* p, const float* src, const float* ps, const float* psh, float slope, const float* wp,
                 float* dst, float* stat, int accumulate, hipStream_t s, const PersBwd* bw = nullptr);
bool patch_can_fuse(const PatchPlan* p);
int patch_fused_blocks(const PatchPlan* p);     // partial rows written by a fused data gradient (patch_can_fuse)

struct WgradPlan;
// While one of these is alive on the calling thread, weight-gradient plans are the ones sized to run BESIDE another kernel chain
// (the R(2+1)D executor's side stream: 160 of 256 CUs); otherwise a weight gradient takes the whole chip.
extern thread_local int g_wgrad_beside;
struct WgradBeside { WgradBeside() { ++g_wgrad_beside; } ~WgradBeside() { --g_wgrad_beside; } };
const WgradPlan* wgrad_lookup(const MdConvDesc* d, int xpitch = 0, int xc0 = 0, int dw_cin = 0);   // xpitch != 0: X = channel slice of a wider tensor
size_t wgrad_patch_workspace_floats(const WgradPlan* p);
bool wgrad_plan_xsplit_ok(const WgradPlan* p);
bool wgrad_plan_second_form(const WgradPlan* p);      // this plan's launches go through k_wgrad2 (needed for a transposed result)
int wgrad_patch_launch(const WgradPlan* p, const MdConvDesc* d, const float* src, const float* ps, const float* psh,
                       float slope, const float* dy, float* dw, float* slab, hipStream_t s, int ysplit = 0, int xsplit = 0);
// second form of the weight gradient (conv_wgrad2.hip): two workgroups per CU, 64-pixel boxes, k-groups by input channel
struct Wgrad2Plan;
const Wgrad2Plan* wgrad2_lookup(const MdConvDesc* d, int beside, int xpitch = 0, int xc0 = 0, int dw_cin = 0);     // nullptr: not handled (first form runs)
size_t wgrad2_workspace_floats(const Wgrad2Plan* p);
int wgrad2_launch(const Wgrad2Plan* p, const MdConvDesc* d, const float* src, const float* ps, const float* psh, float slope,
                  const float* dy, float* dw, float* slab, hipStream_t s);
// Deferred reduction of second-form weight gradients (the executor batches the slab reductions of a backward range into one
// launch): wgrad_partial runs only the slab-producing kernel and describes the pending reduction.
struct WgradPending { const Wgrad2Plan* p; int Cout, Cin; const float* slab; float* dw; };
int wgrad_partial(const MdConvDesc* d, const float* src, const float* ps, const float* psh, float slope, const float* dy, float* dw,
                  float* slab, hipStream_t s, WgradPending* out);      // MD_ERR_UNSUPPORTED: not a second-form geometry
int wgrad2_launch_partial(const Wgrad2Plan* p, const float* src, const float* ps, const float* psh, float slope, const float* dy,
                          float* slab, hipStream_t s);
int wgrad2_reduce_batch(int n, const WgradPending* items, hipStream_t s);
int patch_pack_batch(int n, const MdConvDesc* const* descs, const int* dgrad, const float* const* w, float* const* outs,
                     unsigned char* handled, hipStream_t s);
