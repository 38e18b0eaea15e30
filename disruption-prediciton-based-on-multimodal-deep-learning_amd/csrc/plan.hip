// Whole-trunk executor for R2Plus1DNet (src/models/R2Plus1D.py:207-226): builds the unit / block list of
// the network once (host metadata only) and replays forward and backward as a fixed sequence of launches on
// the caller's stream -- no allocation, no synchronisation, so the sequence can be captured in a hipGraph.
//
// Fusion plan (what is materialised in HBM):
//   * every Conv3dBlock writes only its RAW conv output (+ per-block partial BN sums from the epilogue);
//     BatchNorm + LeakyReLU are applied by the CONSUMER while it stages its input ("BN-on-read");
//   * block outputs z = leaky(skip + main) are materialised (two consumers each);
//   * backward keeps 4 gradient scratch buffers and runs BN-backward in place.
#include <vector>
#include <cstdlib>
#include <new>
#include "common.h"

extern "C" int md_bn_bwd_apply_fused(const float* dA, int g_in, const MdActView* main, const MdActView* skip, float alpha,
                                     const float* mean, const float* invstd, const float* partial, int32_t blocks, int64_t count,
                                     float* dgamma, float* dbeta, int64_t rows, int32_t C, float* d_raw, float* dS, void* stream);
extern "C" int md_bn_eval_params(int32_t C, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                 float eps, float* mean, float* invstd, float* scale, float* shift, void* stream);

struct Unit {
  MdConvDesc d;
  int in_unit;      // >=0: input is the raw output of that unit, BN+act applied on read
  int in_z;         // >=0: input is materialised tensor z[in_z]
  float slope;
  int stage;
  int64_t rows;     // output pixels
  size_t raw_off, stat_off, wf_off, wd_off;   // float offsets into the workspace
  int Cp;
  int split;        // d_raw of this unit is kept in the pre-split bf16 format (md_bn_bwd_apply_fmt)
  size_t slab_off;  // this unit's weight-gradient slabs (kept until the batched reduction at the end of a backward range)
  size_t xs_off;    // pre-split bf16 copy of this unit's ACTIVATION (input of the weight gradients of its consumers); 0 = none
  int xsplit;       // this unit's weight gradient reads the pre-split copy of its input
};
struct Block {
  int c1s, c1t, c2s, c2t, dss, dst;   // unit ids (dss/dst = -1 without downsample)
  int in_z, out_z, stage;
};
struct ZT { int C; int64_t rows; size_t off; size_t xs_off = 0; };      // xs_off: pre-split copy (see Unit::xs_off)

// Optional per-kernel-class timing with HIP events on the launch stream (used by bench.py's roofline leg).
enum { KC_FWD = 0, KC_DGRAD = 1, KC_WGRAD = 2, KC_N = 3 };
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> pool;   // pairs (start, stop)
  std::vector<int> cls;           // class of pair i
  std::vector<double> flop;       // algorithmic FLOPs of pair i (on the first launch of an operation)
  std::vector<unsigned char> first;   // pair i is the first launch of its operation (the launch count is per operation)
  size_t used = 0;                // pairs recorded since the last read
};
thread_local MdProfHook g_md_prof_hook = {nullptr, nullptr};

struct MdPlan {
  Prof prof;
  int B, T, H, W;
  float alpha;
  std::vector<Unit> units;
  std::vector<Block> blocks;
  std::vector<ZT> z;
  size_t part_off, part2_off, coef_off, slab_off, g_off[4], gmax, total_floats;
  size_t part_floats;
  int feat_dim;
  int bwd_p;   // gradient buffer currently holding dZ
  // Weight gradients run on a second stream, concurrently with the BatchNorm-backward / data-gradient
  // chain that the next unit waits for: the chain is HBM-bound, the weight gradient MFMA/VALU-bound, and one workgroup
  // of each fits a CU together.  ready = d_raw of the unit is complete (main -> side); done[b] = the last weight
  // gradient reading gradient buffer b has finished (side -> main, awaited before b is overwritten).
  hipStream_t side = nullptr;
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_done[4] = {nullptr, nullptr, nullptr, nullptr}, ev_join = nullptr;
  bool done_pending[4] = {false, false, false, false};
  int ready_ix = 0;
  bool side_used = false;     // work was queued on the side stream since the last join
  int side_state = 0;         // 0 not tried, 1 available, -1 unavailable (then everything stays on the caller's stream)
  bool defer_join = false;    // md_plan_backward_range leaves the join to the caller (md_plan_join)
  bool side_wanted = false;   // md_plan_use_side_stream(1) was called
  // red_blocks[u] > 0: the data gradient of u's consumer has already left g = dA * leaky'(bn(y_u)) in the gradient buffer
  // and that many partial rows of the BatchNorm-backward reduction in the partial buffer (md_conv_dgrad_bnred)
  std::vector<int> red_blocks;
  // second-form weight gradients whose slab reduction is still to be launched (md_plan_backward_range flushes them in ONE launch:
  // measured 32 reductions of 6-10 us, each behind a dependent-launch gap, on the critical chain of the one-stream schedule)
  std::vector<WgradPending> pending;
  bool batch_reduce = true;
};

// Round 3: the default schedule is ONE stream.  With two workgroups per CU the weight-gradient kernels fill the chip on their own,
// and beside the data-gradient chain the two queues only take turns (profiles/r03_schedule_sweep.txt: 6.07 ms serial, 6.09-6.22 with
// all or part of the weight gradients overlapped); the side stream remains for md_plan_use_side_stream(1) / MD_WGRAD_STREAM=1.
static bool side_stream(MdPlan* P) {
  if (P->side_state) return P->side_state > 0;
  static const int on = getenv("MD_WGRAD_STREAM") && atoi(getenv("MD_WGRAD_STREAM")) == 1;
  if (!on && !P->side_wanted) return false;          // (state stays "not tried": an explicit request may still come)
  P->side_state = -1;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); return false; }
  // measured: the side stream at the HIGHEST priority gives 6.36 ms/step, at the lowest 6.42 (the weight gradients are the
  // longer of the two queues under contention, so letting them go first shortens the tail); MD_SIDE_PRIO_HIGH=0 for lowest
  static const int prio_hi = !(getenv("MD_SIDE_PRIO_HIGH") && atoi(getenv("MD_SIDE_PRIO_HIGH")) == 0);
  if (hipStreamCreateWithPriority(&P->side, hipStreamNonBlocking, prio_hi ? hi : lo) != hipSuccess) { (void)hipGetLastError(); return false; }
  bool ok = true;
  for (auto& e : P->ev_ready) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  for (auto& e : P->ev_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); return false; }
  P->side_state = 1;
  return true;
}
// the caller's stream must not overwrite gradient buffer b while a weight gradient on the side stream still reads it
static int await_buffer(MdPlan* P, int b, void* stream) {
  if (b >= 0 && P->side_state > 0 && P->done_pending[b]) {
    if (hipStreamWaitEvent((hipStream_t)stream, P->ev_done[b], 0) != hipSuccess) return MD_ERR_LAUNCH;
    P->done_pending[b] = false;
  }
  return MD_OK;
}
// everything queued on the side stream becomes visible to the caller's stream
static int join_side(MdPlan* P, void* stream) {
  if (P->side_state > 0 && P->side_used) {
    if (hipEventRecord(P->ev_join, P->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, P->ev_join, 0) != hipSuccess)
      return MD_ERR_LAUNCH;
    P->side_used = false;
    for (bool& d : P->done_pending) d = false;
  }
  return MD_OK;
}

static int conv_out(int i, int k, int s, int p) { return (i + 2 * p - k) / s + 1; }

static int add_unit(MdPlan* P, int Ti, int Hi, int Wi, int cin, int cout, int kt, int kh, int kw, int st, int sh, int sw,
                    int pt, int ph, int pw, float slope, int in_unit, int in_z, int stage) {
  Unit u;
  u.d.N = P->B; u.d.Ti = Ti; u.d.Hi = Hi; u.d.Wi = Wi; u.d.Cin = cin; u.d.Cout = cout;
  u.d.kt = kt; u.d.kh = kh; u.d.kw = kw; u.d.st = st; u.d.sh = sh; u.d.sw = sw; u.d.pt = pt; u.d.ph = ph; u.d.pw = pw;
  u.d.To = conv_out(Ti, kt, st, pt); u.d.Ho = conv_out(Hi, kh, sh, ph); u.d.Wo = conv_out(Wi, kw, sw, pw);
  u.in_unit = in_unit; u.in_z = in_z; u.slope = slope; u.stage = stage;
  u.rows = (int64_t)P->B * u.d.To * u.d.Ho * u.d.Wo;
  u.Cp = md_cpad(cout);
  P->units.push_back(u);
  return (int)P->units.size() - 1;
}

// SpatioTemporalConv (R2Plus1D.py:138-157): spatial (1,k,k) unit then temporal (k,1,1) unit.
static void add_st_conv(MdPlan* P, int Ti, int Hi, int Wi, int cin, int cout, int k, int stride, int pad, int in_z, int stage,
                        int* us, int* ut) {
  const int mid = (int)((long long)k * k * k * cin * cout / ((long long)k * k * cin + (long long)k * cout));
  *us = add_unit(P, Ti, Hi, Wi, cin, mid, 1, k, k, 1, stride, stride, 0, pad, pad, 0.01f, -1, in_z, stage);
  const Unit& s = P->units[*us];
  *ut = add_unit(P, s.d.To, s.d.Ho, s.d.Wo, mid, cout, k, 1, 1, stride, 1, 1, pad, 0, 0, 0.01f, *us, -1, stage);
}

extern "C" int md_plan_create(int32_t B, int32_t T, int32_t H, int32_t W, const int32_t* layer_sizes, float alpha,
                              MdPlan** out) {
  WgradBeside beside_hint;          // workspace sizes below are those of the side-stream weight-gradient plans

  if (!layer_sizes || !out) return MD_ERR_NULL;
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return MD_ERR_BAD_SHAPE;
  for (int i = 0; i < 4; ++i) if (layer_sizes[i] < 1) return MD_ERR_BAD_SHAPE;
  MdPlan* P = new (std::nothrow) MdPlan();
  if (!P) return MD_ERR_WORKSPACE;
  P->B = B; P->T = T; P->H = H; P->W = W; P->alpha = alpha; P->bwd_p = 0;
  // z[0] = input clip in channels-last
  P->z.push_back(ZT{3, (int64_t)B * T * H * W, 0});
  // stem (R2Plus1D.py:125-137,210): 7x7/s2 spatial with 45 mid channels, then k3 temporal
  int us = add_unit(P, T, H, W, 3, 45, 1, 7, 7, 1, 2, 2, 0, 3, 3, alpha, -1, 0, 0);
  const Unit s0 = P->units[us];
  int ut = add_unit(P, s0.d.To, s0.d.Ho, s0.d.Wo, 45, 32, 3, 1, 1, 1, 1, 1, 1, 0, 0, alpha, us, -1, 0);
  const Unit s1 = P->units[ut];
  P->z.push_back(ZT{32, s1.rows, 0});
  int cur_z = 1, curT = s1.d.To, curH = s1.d.Ho, curW = s1.d.Wo, curC = 32;
  const int couts[4] = {32, 64, 64, 128};
  for (int st = 0; st < 4; ++st) {
    for (int bi = 0; bi < layer_sizes[st]; ++bi) {
      const bool down = (bi == 0 && st > 0);
      const int cout = couts[st];
      Block b; b.stage = st + 1; b.in_z = cur_z; b.dss = b.dst = -1;
      add_st_conv(P, curT, curH, curW, curC, cout, 3, down ? 2 : 1, 1, cur_z, st + 1, &b.c1s, &b.c1t);
      const Unit t1 = P->units[b.c1t];
      // conv2 takes the (not materialised) output of conv1.temporal
      const int mid2 = (int)((long long)27 * cout * cout / ((long long)9 * cout + 3LL * cout));
      b.c2s = add_unit(P, t1.d.To, t1.d.Ho, t1.d.Wo, cout, mid2, 1, 3, 3, 1, 1, 1, 0, 1, 1, 0.01f, b.c1t, -1, st + 1);
      b.c2t = add_unit(P, t1.d.To, t1.d.Ho, t1.d.Wo, mid2, cout, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0.01f, b.c2s, -1, st + 1);
      if (down) add_st_conv(P, curT, curH, curW, curC, cout, 1, 2, 0, cur_z, st + 1, &b.dss, &b.dst);
      const Unit t2 = P->units[b.c2t];
      if (down) {
        const Unit dt = P->units[b.dst];
        if (dt.d.To != t2.d.To || dt.d.Ho != t2.d.Ho || dt.d.Wo != t2.d.Wo) { delete P; return MD_ERR_BAD_SHAPE; }
      }
      P->z.push_back(ZT{cout, t2.rows, 0});
      b.out_z = (int)P->z.size() - 1;
      P->blocks.push_back(b);
      cur_z = b.out_z; curT = t2.d.To; curH = t2.d.Ho; curW = t2.d.Wo; curC = cout;
    }
  }
  P->feat_dim = curC;
  // workspace layout (floats, every region 64-float aligned)
  size_t off = 0, gmax = 0, pmax = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return o; };
  for (auto& z : P->z) { const size_t n = (size_t)z.rows * md_cpad(z.C); z.off = take(n); if (n > gmax) gmax = n; }
  for (auto& u : P->units) {
    const size_t n = (size_t)u.rows * u.Cp;
    u.raw_off = take(n); if (n > gmax) gmax = n;
    u.stat_off = take((size_t)4 * u.Cp);
    u.wf_off = take(md_conv_wpack_fwd_floats(&u.d));
    u.wd_off = take(md_conv_wpack_dgrad_floats(&u.d));
    size_t pf = (size_t)md_conv_fwd_stat_blocks(&u.d) * 2 * u.Cp;
    const size_t pb = (size_t)md_bn_bwd_blocks(u.rows, u.d.Cout) * 2 * u.Cp;
    if (pb > pf) pf = pb;
    const size_t pr = (size_t)md_conv_dgrad_bnred_blocks(&u.d) * 2 * md_cpad(u.d.Cin);     // fused reduction of the producer
    if (pr > pf) pf = pr;
    if (pf > pmax) pmax = pf;
  }
  P->part_floats = pmax;
  P->part_off = take(pmax);
  {   // statistics partials of the skip-path units (they run on the side stream next to the main path of their block)
    size_t p2 = 0;
    for (const Block& b : P->blocks)
      for (int ui : {b.dss, b.dst})
        if (ui >= 0) { const Unit& u = P->units[ui]; const size_t n = (size_t)md_conv_fwd_stat_blocks(&u.d) * 2 * u.Cp; if (n > p2) p2 = n; }
    P->part2_off = take(p2 ? p2 : 4);
  }
  size_t smax = 0;
  static const int batch_off = getenv("MD_WGRAD_BATCH_REDUCE") && atoi(getenv("MD_WGRAD_BATCH_REDUCE")) == 0;
  P->batch_reduce = !batch_off;
  for (auto& u : P->units) {
    const size_t n = md_conv_wgrad_workspace_floats(&u.d); if (n > smax) smax = n;
    u.slab_off = P->batch_reduce ? take(n) : 0;          // own slabs: they outlive the unit's kernel (0.73 GB in all at the BASELINE shape)
  }
  P->slab_off = take(smax);
  P->coef_off = take(2 * 1024);
  P->gmax = gmax;
  P->red_blocks.assign(P->units.size(), 0);
  for (size_t i = 0; i < P->units.size(); ++i) P->units[i].split = md_conv_split_dy_ok(&P->units[i].d, i != 0 ? 1 : 0);
  // Pre-split copies of the weight gradients' X operands: the activation of a unit (or a materialised block tensor) that feeds
  // a convolution is written once more as bf16 hi|lo pairs during the forward pass -- on the side stream, which the forward
  // leaves idle -- so that the weight-gradient kernels stage X by plain copy (no BatchNorm-on-read, no split: a third of
  // their per-box time).  Not for the stem's pixel-pair form (it reads the clip itself).
  // MEASURED (round 2): bit-identical, and slower -- 6.61 vs 6.21 ms per step: the weight-gradient kernels gain 4 % (3.44 -> 3.30 ms;
  // they wait for their loads, not for the staging arithmetic), the 28 copy launches cost 1.07 ms of side-stream time and slow
  // the concurrent forward convolutions by 0.3 ms.  Off unless MD_SPLIT_X=1.
  static const int split_x = getenv("MD_SPLIT_X") && atoi(getenv("MD_SPLIT_X")) == 1;
  for (auto& u : P->units) { u.xs_off = 0; u.xsplit = 0; }
  for (size_t i = 0; split_x && i < P->units.size(); ++i) {
    Unit& u = P->units[i];
    if (!md_conv_wgrad_xsplit_ok(&u.d) || (u.in_unit < 0 && u.in_z == 0)) continue;
    u.xsplit = 1;
    if (u.in_unit >= 0) {
      Unit& src = P->units[u.in_unit];
      if (!src.xs_off) src.xs_off = take(md_bn_act_split_floats(src.rows, src.d.Cout));
    } else {
      ZT& z = P->z[u.in_z];
      if (!z.xs_off) z.xs_off = take(md_bn_act_split_floats(z.rows, z.C));
    }
  }
  for (int i = 0; i < 4; ++i) P->g_off[i] = take(gmax);
  P->total_floats = off;
  *out = P;
  return MD_OK;
}

extern "C" void md_plan_destroy(MdPlan* p) {
  if (!p) return;
  for (hipEvent_t e : p->prof.pool) (void)hipEventDestroy(e);
  if (p->side) {
    (void)hipStreamSynchronize(p->side);
    for (hipEvent_t e : p->ev_ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_done) if (e) (void)hipEventDestroy(e);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    (void)hipStreamDestroy(p->side);
  }
  delete p;
}
extern "C" int32_t md_plan_num_units(const MdPlan* p) { return p ? (int32_t)p->units.size() : 0; }
extern "C" int md_plan_unit_desc(const MdPlan* p, int32_t i, MdConvDesc* out) {
  if (!p || !out) return MD_ERR_NULL;
  if (i < 0 || i >= (int)p->units.size()) return MD_ERR_BAD_SHAPE;
  *out = p->units[i].d;
  return MD_OK;
}
extern "C" size_t md_plan_workspace_bytes(const MdPlan* p) { return p ? p->total_floats * 4 : 0; }
// Where a forward leaves things in the workspace (float offsets), for diagnostics that compare intermediate tensors
// with the oracle: unit i's raw conv output [rows][Cp] and its [mean | invstd | scale | shift][Cp] statistics block.
extern "C" int md_plan_unit_layout(const MdPlan* p, int32_t i, size_t* raw_off, size_t* stat_off, int64_t* rows, int32_t* Cp) {
  if (!p || !raw_off || !stat_off || !rows || !Cp) return MD_ERR_NULL;
  if (i < 0 || i >= (int)p->units.size()) return MD_ERR_BAD_SHAPE;
  const Unit& u = p->units[i];
  *raw_off = u.raw_off; *stat_off = u.stat_off; *rows = u.rows; *Cp = u.Cp;
  return MD_OK;
}
// Materialised tensor zi (0: the clip in channels-last, 1: the stem output, then one per residual block): [rows][md_cpad(C)].
extern "C" int32_t md_plan_num_z(const MdPlan* p) { return p ? (int32_t)p->z.size() : 0; }
extern "C" int md_plan_z_layout(const MdPlan* p, int32_t zi, size_t* off, int64_t* rows, int32_t* C) {
  if (!p || !off || !rows || !C) return MD_ERR_NULL;
  if (zi < 0 || zi >= (int)p->z.size()) return MD_ERR_BAD_SHAPE;
  *off = p->z[zi].off; *rows = p->z[zi].rows; *C = p->z[zi].C;
  return MD_OK;
}
extern "C" int32_t md_plan_feat_dim(const MdPlan* p) { return p ? p->feat_dim : 0; }

#define RC(x) do { int rc__ = (x); if (rc__ != MD_OK) return rc__; } while (0)

static double unit_flops(const Unit& u) {
  return 2.0 * (double)u.rows * u.d.Cout * u.d.Cin * u.d.kt * u.d.kh * u.d.kw;
}
// time the launches of one convolution operation; a no-op unless profiling is enabled.  Every launch made while the scope is open gets
// its own event pair, stamped by the dispatch itself (common.h::md_klaunch)
struct ProfScope {
  MdPlan* P; int cls; double flop; bool on; bool any;
  static int acquire(void* ctx, hipEvent_t* a, hipEvent_t* b) {
    ProfScope* sc = (ProfScope*)ctx;
    Prof& pr = sc->P->prof;
    if (pr.used * 2 + 2 > pr.pool.size()) {
      hipEvent_t x, y;
      if (hipEventCreate(&x) != hipSuccess) return 0;
      if (hipEventCreate(&y) != hipSuccess) { (void)hipEventDestroy(x); return 0; }
      pr.pool.push_back(x); pr.pool.push_back(y); pr.cls.push_back(0); pr.flop.push_back(0.0); pr.first.push_back(0);
    }
    const size_t idx = pr.used++;
    pr.cls[idx] = sc->cls; pr.flop[idx] = sc->any ? 0.0 : sc->flop; pr.first[idx] = sc->any ? 0 : 1;
    sc->any = true;
    *a = pr.pool[2 * idx]; *b = pr.pool[2 * idx + 1];
    return 1;
  }
  ProfScope(MdPlan* P_, int cls_, double flop_, void*) : P(P_), cls(cls_), flop(flop_), on(P_->prof.on), any(false) {
    if (on) { g_md_prof_hook.ctx = this; g_md_prof_hook.acquire = &ProfScope::acquire; }
  }
  ~ProfScope() { if (on) { g_md_prof_hook.ctx = nullptr; g_md_prof_hook.acquire = nullptr; } }
};

// 0: queue the weight gradients on the caller's stream (serial schedule), 1: use the side stream (default when the
// runtime provides one).  Both schedules launch the same kernels and give bit-identical results.
extern "C" int md_plan_use_side_stream(MdPlan* P, int enable) {
  if (!P) return MD_ERR_NULL;
  if (!enable) {
    P->side_wanted = false;
    if (P->side_state > 0) { if (hipStreamSynchronize(P->side) != hipSuccess) return MD_ERR_LAUNCH; P->side_state = -2; }
    else if (P->side_state == 0) P->side_state = -3;           // never create it
  } else {
    P->side_wanted = true;
    if (P->side_state == -2) P->side_state = 1;                 // re-enable the existing stream
    else if (P->side_state == -3) P->side_state = 0;            // create on next use
  }
  return MD_OK;
}

// Data-parallel training reduces the weight gradients of a stage as soon as they exist.  With the join deferred,
// md_plan_backward_range returns without making the caller's stream wait for the side stream; the caller queues the
// consumer of those gradients behind md_plan_side_stream() instead (the collective then waits for the weight gradients,
// the backward chain does not), and calls md_plan_join once at the end.
extern "C" int md_plan_defer_join(MdPlan* P, int defer) {
  if (!P) return MD_ERR_NULL;
  P->defer_join = defer != 0;
  return MD_OK;
}
extern "C" void* md_plan_side_stream(MdPlan* P) { return (P && side_stream(P)) ? (void*)P->side : nullptr; }
extern "C" int md_plan_join(MdPlan* P, void* stream) { return P ? join_side(P, stream) : MD_ERR_NULL; }

extern "C" int md_plan_profile_enable(MdPlan* P, int enable) {
  if (!P) return MD_ERR_NULL;
  P->prof.on = enable != 0;
  if (enable == 1) P->prof.used = 0;        // 0 (off) and 2 (resume) keep what has been recorded: sampling some steps of a run
  return MD_OK;
}
// Create the event pairs for `records` timed launches ahead of time (hipEventCreate is not free; keep it out of timed regions).
extern "C" int md_plan_profile_reserve(MdPlan* P, int32_t records) {
  if (!P) return MD_ERR_NULL;
  Prof& pr = P->prof;
  while (pr.pool.size() < (size_t)2 * (size_t)(records > 0 ? records : 0)) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return MD_ERR_LAUNCH;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return MD_ERR_LAUNCH; }
    pr.pool.push_back(a); pr.pool.push_back(b); pr.cls.push_back(0); pr.flop.push_back(0.0); pr.first.push_back(0);
  }
  return MD_OK;
}
// Sum of event-measured durations (ms), launch counts and algorithmic FLOPs per class since the last read
// (classes: 0 conv forward, 1 conv data-gradient, 2 conv weight-gradient).  Synchronises on the recorded events.
extern "C" int md_plan_profile_read(MdPlan* P, double* ms, int64_t* launches, double* flops) {
  if (!P || !ms || !launches || !flops) return MD_ERR_NULL;
  for (int c = 0; c < KC_N; ++c) { ms[c] = 0.0; launches[c] = 0; flops[c] = 0.0; }
  Prof& pr = P->prof;
  for (size_t i = 0; i < pr.used; ++i) {
    if (hipEventSynchronize(pr.pool[2 * i + 1]) != hipSuccess) return MD_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, pr.pool[2 * i], pr.pool[2 * i + 1]) != hipSuccess) return MD_ERR_LAUNCH;
    ms[pr.cls[i]] += t; launches[pr.cls[i]] += pr.first[i] ? 1 : 0; flops[pr.cls[i]] += pr.flop[i];
  }
  pr.used = 0;
  return MD_OK;
}

static MdActView unit_out_view(const MdPlan* P, float* ws, int ui) {
  const Unit& u = P->units[ui];
  MdActView v; v.data = ws + u.raw_off; v.scale = ws + u.stat_off + 2 * u.Cp; v.shift = ws + u.stat_off + 3 * u.Cp;
  v.slope = u.slope; return v;
}
static MdActView z_view(const MdPlan* P, float* ws, int zi) {
  MdActView v; v.data = ws + P->z[zi].off; v.scale = nullptr; v.shift = nullptr; v.slope = 1.f; return v;
}
static MdActView unit_in_view(const MdPlan* P, float* ws, int ui) {
  const Unit& u = P->units[ui];
  return u.in_unit >= 0 ? unit_out_view(P, ws, u.in_unit) : z_view(P, ws, u.in_z);
}

extern "C" int md_plan_forward(MdPlan* P, const float* x, const float* const* w, const float* const* gamma,
                               const float* const* beta, float* const* rmean, float* const* rvar, int training,
                               float* feat, void* workspace, void* stream) {
  if (!P || !x || !w || !gamma || !beta || !feat || !workspace) return MD_ERR_NULL;
  if (!training && (!rmean || !rvar)) return MD_ERR_NULL;
  float* ws = (float*)workspace;
  RC(md_nchw_to_cl(x, P->B, 3, (int64_t)P->T * P->H * P->W, ws + P->z[0].off, stream));
  // pack every GEMM operand up front: one batched launch for the patch-format ones, per-unit for the rest
  {
    const int n = (int)P->units.size();
    std::vector<const MdConvDesc*> descs(2 * n); std::vector<int> dg(2 * n);
    std::vector<const float*> wsrc(2 * n); std::vector<float*> outs(2 * n); std::vector<unsigned char> handled(2 * n);
    for (int i = 0; i < n; ++i) {
      const Unit& u = P->units[i];
      descs[2 * i] = &u.d; dg[2 * i] = 0; wsrc[2 * i] = w[i]; outs[2 * i] = ws + u.wf_off;
      descs[2 * i + 1] = &u.d; dg[2 * i + 1] = 1; wsrc[2 * i + 1] = w[i];
      outs[2 * i + 1] = (training && i != 0) ? ws + u.wd_off : nullptr;      // unit 0 (stem) needs no data gradient
    }
    RC(patch_pack_batch(2 * n, descs.data(), dg.data(), wsrc.data(), outs.data(), handled.data(), (hipStream_t)stream));
    for (int i = 0; i < n; ++i) {
      const Unit& u = P->units[i];
      float* wf = handled[2 * i] ? nullptr : ws + u.wf_off;
      float* wd = handled[2 * i + 1] ? nullptr : outs[2 * i + 1];
      if (wf || wd) RC(md_conv_pack_weights(&u.d, w[i], wf, wd, stream));
    }
  }
  size_t next_block = 0;
  // conv + BatchNorm statistics of one unit in training mode, on stream `s` with partial-sum scratch `part`
  auto train_unit = [&](size_t i, float* part, void* s) -> int {
    const Unit& u = P->units[i];
    MdActView in = unit_in_view(P, ws, (int)i);
    float* st = ws + u.stat_off;
    { ProfScope ps(P, KC_FWD, unit_flops(u), s);
      RC(md_conv_fwd(&u.d, &in, ws + u.wf_off, ws + u.raw_off, part, s)); }
    return md_bn_finalize(part, md_conv_fwd_stat_blocks(&u.d), u.d.Cout, u.rows, gamma[i], beta[i], 1e-5f, 0.1f,
                          rmean ? rmean[i] : nullptr, rvar ? rvar[i] : nullptr, st, st + u.Cp, st + 2 * u.Cp, st + 3 * u.Cp, s);
  };
  // pre-split copy of a unit's activation / a block tensor for the weight gradients (see md_plan_create): on the side stream,
  // behind an event of the stream that has just completed the statistics (`from` == the side stream itself: in order, no event)
  auto split_copy = [&](const MdActView& v, int64_t rows, int C, size_t xs_off, void* from) -> int {
    if (!training || !xs_off) return MD_OK;
    void* to = from;
    if (side_stream(P)) {
      to = P->side;
      if (from != (void*)P->side) {
        hipEvent_t ready = P->ev_ready[P->ready_ix]; P->ready_ix ^= 1;
        if (hipEventRecord(ready, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent(P->side, ready, 0) != hipSuccess) return MD_ERR_LAUNCH;
      }
      P->side_used = true;
    }
    return md_bn_act_split(&v, rows, C, ws + xs_off, to);
  };
  auto split_unit = [&](size_t i, void* from) -> int {
    const Unit& u = P->units[i];
    MdActView v = unit_out_view(P, ws, (int)i);
    return split_copy(v, u.rows, u.d.Cout, u.xs_off, from);
  };
  auto split_z = [&](int zi, void* from) -> int {
    const ZT& z = P->z[zi];
    MdActView v = z_view(P, ws, zi);
    return split_copy(v, z.rows, z.C, z.xs_off, from);
  };
  bool skip_on_side = false;       // the current block's skip path (dss, dst) has been queued on the side stream
  for (size_t i = 0; i < P->units.size(); ++i) {
    const Unit& u = P->units[i];
    MdActView in = unit_in_view(P, ws, (int)i);
    float* st = ws + u.stat_off;
    if (training && next_block < P->blocks.size()) {
      const Block& b = P->blocks[next_block];
      if ((int)i == b.c1s && b.dst >= 0 && side_stream(P)) {
        // downsampling block: its 1x1x1 skip convolutions depend only on the block input -- run them on the side stream
        // while the four main-path units run here; the residual close below waits for them
        hipEvent_t ready = P->ev_ready[P->ready_ix]; P->ready_ix ^= 1;
        if (hipEventRecord(ready, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(P->side, ready, 0) != hipSuccess)
          return MD_ERR_LAUNCH;
        RC(train_unit((size_t)b.dss, ws + P->part2_off, P->side));
        RC(split_unit((size_t)b.dss, P->side));
        RC(train_unit((size_t)b.dst, ws + P->part2_off, P->side));
        P->side_used = true; skip_on_side = true;
      }
      if (skip_on_side && ((int)i == b.dss || (int)i == b.dst)) {
        if ((int)i == b.dst) {
          RC(join_side(P, stream));
          skip_on_side = false;
          MdActView mainv = unit_out_view(P, ws, b.c2t);
          MdActView skipv = unit_out_view(P, ws, b.dst);
          const Unit& t2 = P->units[b.c2t];
          RC(md_residual_fwd(&skipv, &mainv, P->alpha, t2.rows, t2.d.Cout, ws + P->z[b.out_z].off, stream));
          RC(split_z(b.out_z, stream));
          ++next_block;
        }
        continue;
      }
    }
    if (training) {
      RC(train_unit(i, ws + P->part_off, stream));
      RC(split_unit(i, stream));
    } else {
      RC(md_conv_fwd(&u.d, &in, ws + u.wf_off, ws + u.raw_off, nullptr, stream));
      RC(md_bn_eval_params(u.d.Cout, gamma[i], beta[i], rmean[i], rvar[i], 1e-5f, st, st + u.Cp, st + 2 * u.Cp, st + 3 * u.Cp,
                           stream));
    }
    if (i == 1) {   // stem output feeds a conv AND an identity skip: materialise it
      MdActView v = unit_out_view(P, ws, 1);
      RC(md_bn_act(&v, u.rows, u.d.Cout, ws + P->z[1].off, stream));
      RC(split_z(1, stream));
    }
    // close the residual block whose last unit this is
    if (next_block < P->blocks.size()) {
      const Block& b = P->blocks[next_block];
      const int last = b.dst >= 0 ? b.dst : b.c2t;
      if ((int)i == last) {
        MdActView mainv = unit_out_view(P, ws, b.c2t);
        MdActView skipv = b.dst >= 0 ? unit_out_view(P, ws, b.dst) : z_view(P, ws, b.in_z);
        const Unit& t2 = P->units[b.c2t];
        RC(md_residual_fwd(&skipv, &mainv, P->alpha, t2.rows, t2.d.Cout, ws + P->z[b.out_z].off, stream));
        RC(split_z(b.out_z, stream));
        ++next_block;
      }
    }
  }
  const ZT& zl = P->z.back();
  RC(md_avgpool_fwd(ws + zl.off, P->B, zl.C, zl.rows / P->B, feat, stream));
  return MD_OK;
}

// md_bn_bwd_apply_fused (the apply pass sums the partial rows itself, no finalize launch) is available but OFF in the executor:
// measured slower (profiles/r03_bn_fused_finalize.txt: 6.31 ms per step with the prologue in each of the apply pass's 4096 workgroups,
// 6.15 ms with 256 workgroups of 1024 threads, 5.99 ms with the separate 5 us finalize launch).  MD_BN_FUSED_FIN=1 switches it on.
static bool bn_fused_fin() {
  static const int on = getenv("MD_BN_FUSED_FIN") && atoi(getenv("MD_BN_FUSED_FIN")) == 1;
  return on;
}

// Weight gradient of unit ui on stream s: second-form geometries only produce their slabs here (own slab region) and queue the
// reduction for flush_pending; everything else (the stem's pixel-pair form, pre-split operands) reduces at once.
static int unit_wgrad(MdPlan* P, float* ws, int ui, const MdActView& in, float* G, float* dw, void* s) {
  const Unit& u = P->units[ui];
  if (P->batch_reduce && !u.xsplit && !u.split) {
    WgradPending pd;
    const int rc = wgrad_partial(&u.d, in.data, in.scale, in.shift, in.slope, G, dw, ws + u.slab_off, (hipStream_t)s, &pd);
    if (rc == MD_OK) { P->pending.push_back(pd); return MD_OK; }
    if (rc != MD_ERR_UNSUPPORTED) return rc;
  }
  return md_conv_wgrad_fmt2(&u.d, &in, u.xsplit, G, u.split, dw, ws + P->slab_off, s);
}
static int flush_pending(MdPlan* P, void* s) {
  if (P->pending.empty()) return MD_OK;
  const int rc = wgrad2_reduce_batch((int)P->pending.size(), P->pending.data(), (hipStream_t)s);
  P->pending.clear();
  return rc;
}

// BN-backward of unit ui given dA in G[gb] (overwritten with d_raw), then wgrad and (optionally) dgrad.
static int unit_backward(MdPlan* P, float* ws, int ui, int gb, int dxb, int accumulate, bool bn_done,
                         const float* const* w, float* const* dw, float* const* dgamma, float* const* dbeta, void* stream) {
  const Unit& u = P->units[ui];
  float* G = ws + P->g_off[gb];
  float* st = ws + u.stat_off;
  (void)w;
  if (!bn_done) {
    MdActView mainv = unit_out_view(P, ws, ui);
    // (fused_fin: the apply pass sums the <= 256 partial rows itself -- no finalize launch between the two passes)
    const int rb = P->red_blocks[ui];
    const bool fused_fin = bn_fused_fin() && !u.split;
    if (rb > 0) {
      // the consumer's data gradient already reduced: G holds g, the partial buffer its sums
      if (fused_fin) {
        RC(md_bn_bwd_apply_fused(G, 1, &mainv, nullptr, 1.f, st, st + u.Cp, ws + P->part_off, rb, u.rows, dgamma[ui], dbeta[ui], u.rows,
                                 u.d.Cout, G, nullptr, stream));
      } else {
        RC(md_bn_bwd_finalize(ws + P->part_off, rb, u.d.Cout, u.rows, dgamma[ui], dbeta[ui], ws + P->coef_off, stream));
        RC(md_bn_bwd_apply_fmt(G, 1, &mainv, nullptr, 1.f, st, st + u.Cp, ws + P->coef_off, u.rows, u.d.Cout, G, u.split, nullptr, stream));
      }
      P->red_blocks[ui] = 0;
    } else {
      const int nb = md_bn_bwd_blocks(u.rows, u.d.Cout);
      RC(md_bn_bwd_reduce(G, &mainv, nullptr, 1.f, st, st + u.Cp, u.rows, u.d.Cout, ws + P->part_off, stream));
      if (fused_fin) {
        RC(md_bn_bwd_apply_fused(G, 0, &mainv, nullptr, 1.f, st, st + u.Cp, ws + P->part_off, nb, u.rows, dgamma[ui], dbeta[ui], u.rows,
                                 u.d.Cout, G, nullptr, stream));
      } else {
        RC(md_bn_bwd_finalize(ws + P->part_off, nb, u.d.Cout, u.rows, dgamma[ui], dbeta[ui], ws + P->coef_off, stream));
        RC(md_bn_bwd_apply_fmt(G, 0, &mainv, nullptr, 1.f, st, st + u.Cp, ws + P->coef_off, u.rows, u.d.Cout, G, u.split, nullptr, stream));
      }
    }
  }
  MdActView in = unit_in_view(P, ws, ui);
  if (u.xsplit) {          // the pre-split copy of the input written during the forward pass
    in.data = ws + (u.in_unit >= 0 ? P->units[u.in_unit].xs_off : P->z[u.in_z].xs_off);
    in.scale = in.shift = nullptr;
  }
  // timing experiments only (wrong gradients): MD_DBG_SKIP_WGRAD=1 leaves the weight gradients out of the step
  static const int skip_wgrad = getenv("MD_DBG_SKIP_WGRAD") && atoi(getenv("MD_DBG_SKIP_WGRAD")) == 1;
  if (skip_wgrad) {
  } else if (side_stream(P)) {
    // d_raw (G) is complete on the caller's stream here; the weight gradient reads it from the side stream
    hipEvent_t ready = P->ev_ready[P->ready_ix]; P->ready_ix ^= 1;
    if (hipEventRecord(ready, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(P->side, ready, 0) != hipSuccess)
      return MD_ERR_LAUNCH;
    { ProfScope ps(P, KC_WGRAD, unit_flops(u), P->side);
      RC(unit_wgrad(P, ws, ui, in, G, dw[ui], P->side)); }
    if (hipEventRecord(P->ev_done[gb], P->side) != hipSuccess) return MD_ERR_LAUNCH;
    P->done_pending[gb] = true; P->side_used = true;
  } else {
    ProfScope ps(P, KC_WGRAD, unit_flops(u), stream);
    RC(unit_wgrad(P, ws, ui, in, G, dw[ui], stream));
  }
  if (dxb >= 0) {
    RC(await_buffer(P, dxb, stream));
    // Which unit's BatchNorm-backward consumes what this data gradient writes?  The producer of the input when the input is
    // a raw unit output (BN-on-read), or the stem's second unit when the input is its materialised activation z[1] and this
    // launch completes dZ1 (accumulate).  Block outputs z[k>1] close with the residual form: not fused.
    static const int fuse_on = !(getenv("MD_FUSE_BNRED") && atoi(getenv("MD_FUSE_BNRED")) == 0);
    const int target = u.in_unit >= 0 ? u.in_unit : ((u.in_z == 1 && accumulate) ? 1 : -1);
    const int nbr = (fuse_on && target >= 0) ? md_conv_dgrad_bnred_blocks(&u.d) : 0;
    ProfScope ps(P, KC_DGRAD, unit_flops(u), stream);
    if (nbr > 0) {
      const Unit& tu = P->units[target];
      MdActView tv = unit_out_view(P, ws, target);
      float* tst = ws + tu.stat_off;
      RC(md_conv_dgrad_fmt(&u.d, G, u.split, ws + u.wd_off, ws + P->g_off[dxb], accumulate, &tv, tst, tst + tu.Cp, ws + P->part_off, stream));
      P->red_blocks[target] = nbr;
    } else {
      RC(md_conv_dgrad_fmt(&u.d, G, u.split, ws + u.wd_off, ws + P->g_off[dxb], accumulate, nullptr, nullptr, nullptr, nullptr, stream));
    }
  }
  return MD_OK;
}

static int block_backward(MdPlan* P, float* ws, const Block& b, const float* const* w, float* const* dw,
                          float* const* dgamma, float* const* dbeta, void* stream) {
  const int p = P->bwd_p;
  int fr[3], n = 0;
  for (int i = 0; i < 4; ++i) if (i != p) fr[n++] = i;
  const int a = fr[0], bb = fr[1], c = fr[2];
  const Unit& t2 = P->units[b.c2t];
  float* st = ws + t2.stat_off;
  MdActView mainv = unit_out_view(P, ws, b.c2t);
  MdActView skipv = b.dst >= 0 ? unit_out_view(P, ws, b.dst) : z_view(P, ws, b.in_z);
  const int nb = md_bn_bwd_blocks(t2.rows, t2.d.Cout);
  float* Gp = ws + P->g_off[p];
  RC(md_bn_bwd_reduce(Gp, &mainv, &skipv, P->alpha, st, st + t2.Cp, t2.rows, t2.d.Cout, ws + P->part_off, stream));
  const bool fused_fin = bn_fused_fin() && !t2.split;
  if (!fused_fin) RC(md_bn_bwd_finalize(ws + P->part_off, nb, t2.d.Cout, t2.rows, dgamma[b.c2t], dbeta[b.c2t], ws + P->coef_off, stream));
  RC(await_buffer(P, a, stream));
  RC(await_buffer(P, p, stream));
  if (fused_fin) {
    RC(md_bn_bwd_apply_fused(Gp, 0, &mainv, &skipv, P->alpha, st, st + t2.Cp, ws + P->part_off, nb, t2.rows, dgamma[b.c2t], dbeta[b.c2t],
                             t2.rows, t2.d.Cout, ws + P->g_off[a], Gp, stream));
  } else {
    RC(md_bn_bwd_apply_fmt(Gp, 0, &mainv, &skipv, P->alpha, st, st + t2.Cp, ws + P->coef_off, t2.rows, t2.d.Cout,
                           ws + P->g_off[a], t2.split, Gp, stream));
  }
  RC(unit_backward(P, ws, b.c2t, a, bb, 0, true, w, dw, dgamma, dbeta, stream));
  RC(unit_backward(P, ws, b.c2s, bb, a, 0, false, w, dw, dgamma, dbeta, stream));
  RC(unit_backward(P, ws, b.c1t, a, bb, 0, false, w, dw, dgamma, dbeta, stream));
  if (b.dst < 0) {
    RC(unit_backward(P, ws, b.c1s, bb, p, 1, false, w, dw, dgamma, dbeta, stream));   // dX accumulates onto dS
  } else {
    RC(unit_backward(P, ws, b.c1s, bb, c, 0, false, w, dw, dgamma, dbeta, stream));
    RC(unit_backward(P, ws, b.dst, p, a, 0, false, w, dw, dgamma, dbeta, stream));    // dS is dA of the skip path
    RC(unit_backward(P, ws, b.dss, a, c, 1, false, w, dw, dgamma, dbeta, stream));
    P->bwd_p = c;
  }
  return MD_OK;
}

extern "C" int md_plan_backward_range(MdPlan* P, const float* dfeat, const float* const* w, const float* const* gamma,
                                      float* const* dw, float* const* dgamma, float* const* dbeta, void* workspace,
                                      int32_t stage_hi, int32_t stage_lo, void* stream) {
  WgradBeside beside_hint;

  if (!P || !w || !dw || !dgamma || !dbeta || !workspace) return MD_ERR_NULL;
  (void)gamma;
  if (stage_hi > 4 || stage_lo < 0 || stage_lo > stage_hi) return MD_ERR_BAD_SHAPE;
  float* ws = (float*)workspace;
  if (stage_hi == 4) {
    if (!dfeat) return MD_ERR_NULL;
    P->bwd_p = 0;
    for (int& r : P->red_blocks) r = 0;
    const ZT& zl = P->z.back();
    RC(await_buffer(P, 0, stream));
    RC(md_avgpool_bwd(dfeat, P->B, zl.C, zl.rows / P->B, ws + P->g_off[0], stream));
  }
  for (int bi = (int)P->blocks.size() - 1; bi >= 0; --bi) {
    const Block& b = P->blocks[bi];
    if (b.stage > stage_hi || b.stage < stage_lo) continue;
    RC(block_backward(P, ws, b, w, dw, dgamma, dbeta, stream));
  }
  if (stage_lo == 0) {
    const int p = P->bwd_p, q = (p + 1) & 3;
    RC(unit_backward(P, ws, 1, p, q, 0, false, w, dw, dgamma, dbeta, stream));
    RC(unit_backward(P, ws, 0, q, -1, 0, false, w, dw, dgamma, dbeta, stream));
  }
  RC(flush_pending(P, side_stream(P) ? (void*)P->side : stream));      // one launch reduces the slabs of every unit of this range
  if (P->defer_join) return MD_OK;    // the caller orders its consumers itself (md_plan_side_stream / md_plan_join)
  return join_side(P, stream);        // the weight gradients of this range are ordered before whatever follows
}

extern "C" int md_plan_backward(MdPlan* P, const float* dfeat, const float* const* w, const float* const* gamma,
                                float* const* dw, float* const* dgamma, float* const* dbeta, void* workspace, void* stream) {
  WgradBeside beside_hint;

  return md_plan_backward_range(P, dfeat, w, gamma, dw, dgamma, dbeta, workspace, 4, 0, stream);
}

extern "C" int md_version(const char** arch_out) {
  if (arch_out) *arch_out = "gfx950";
  return MD_ABI_VERSION;
}
