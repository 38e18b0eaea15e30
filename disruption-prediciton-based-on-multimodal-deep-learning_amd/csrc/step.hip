// One training step of the R(2+1)D classifier from ONE C call: trunk forward (md_plan_forward) -> fused head -> fused softmax loss
// -> head backward -> trunk backward -> gradient clipping + AdamW, all queued on the caller's stream.
// Replaces, for this model, the per-step Python sequence of the reference's loop (src/train.py:40-66: zero_grad, forward, loss,
// finite check, backward, clip_grad_norm_, optimizer.step) -- the finite check is a device flag that gates the update instead of
// a host read (the flag is what the host inspects afterwards).  Same kernels in the same order as the composed path
// (src/models/R2Plus1D.py + src/loss.py + src/optim.py): bit-identical parameters (tests/test_fused_step_gpu.py).
#include "common.h"

namespace {
__global__ void k_step_flag(const float* __restrict__ loss, float* __restrict__ ok) {
  const float l = loss[0];
  ok[0] = (l == l && fabsf(l) != INFINITY) ? 1.f : 0.f;
}
__global__ void k_bump_counters(int64_t* const* __restrict__ counters, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && counters[i]) counters[i][0] += 1;
}
}  // namespace

#define RCS(x) do { int rc_ = (x); if (rc_ != MD_OK) return rc_; } while (0)

extern "C" int md_plan_train_step(MdPlan* plan, const MdTrainStepArgs* a, void* stream) {
  if (!plan || !a) return MD_ERR_NULL;
  if (!a->x || !a->target || !a->w || !a->gamma || !a->beta || !a->dw || !a->dgamma || !a->dbeta || !a->workspace) return MD_ERR_NULL;
  if (!a->feat || !a->dfeat || !a->logits || !a->dlogits || !a->head_save || !a->loss || !a->pred || !a->ok_flag) return MD_ERR_NULL;
  if (!a->w0 || !a->b0 || !a->hgamma || !a->hbeta || !a->w1 || !a->b1 || !a->dw0 || !a->db0 || !a->dhgamma || !a->dhbeta || !a->dw1 ||
      !a->db1)
    return MD_ERR_NULL;
  const int B = a->B, D = md_plan_feat_dim(plan);
  if (B <= 0 || a->Hd <= 0 || a->K <= 0 || D <= 0) return MD_ERR_BAD_SHAPE;
  if ((size_t)2 * B * a->Hd * 4 > 60000) return MD_ERR_UNSUPPORTED;      // the fused head keeps two (B, Hd) tiles in LDS
  hipStream_t s = (hipStream_t)stream;
  RCS(md_plan_forward(plan, a->x, a->w, a->gamma, a->beta, a->rmean, a->rvar, 1, a->feat, a->workspace, stream));
  RCS(md_head_fwd(a->feat, B, D, a->Hd, a->K, a->w0, a->b0, a->hgamma, a->hbeta, a->w1, a->b1, a->head_alpha, a->head_eps,
                  a->head_momentum, 1, a->hrmean, a->hrvar, a->logits, a->head_save, stream));
  if (a->counters && a->ncounters > 0) {
    MD_KLAUNCH(k_bump_counters, dim3(md_cdiv(a->ncounters, 64)), dim3(64), 0, s, a->counters, a->ncounters);
    MD_CHECK_LAUNCH();
  }
  RCS(md_softmax_loss(a->loss_kind, a->logits, a->target, B, a->K, a->class_weight, a->margins, a->gamma_or_s, a->loss, a->dlogits,
                      a->pred, stream));
  MD_KLAUNCH(k_step_flag, dim3(1), dim3(1), 0, s, a->loss, a->ok_flag);
  MD_CHECK_LAUNCH();
  RCS(md_head_bwd(a->dlogits, a->feat, B, D, a->Hd, a->K, a->w0, a->hgamma, a->w1, a->head_alpha, a->head_save, a->dfeat, a->dw0,
                  a->db0, a->dhgamma, a->dhbeta, a->dw1, a->db1, stream));
  RCS(md_plan_backward(plan, a->dfeat, a->w, a->gamma, a->dw, a->dgamma, a->dbeta, a->workspace, stream));
  if (a->opt_nchunks > 0) {
    if (!a->opt_tensors || !a->opt_chunks || !a->opt_partial || a->opt_step < 1) return MD_ERR_NULL;
    const float* coef = nullptr;
    if (a->max_norm > 0.f) {
      // opt_partial = [norm, coef | one sum per chunk] (src/optim.py)
      RCS(md_opt_grad_norm(a->opt_tensors, a->opt_chunks, a->opt_nchunks, a->max_norm, a->opt_partial + 2, a->opt_partial, stream));
      coef = a->opt_partial;
    }
    RCS(md_opt_adamw_step_if(a->opt_tensors, a->opt_chunks, a->opt_nchunks, coef, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay,
                             a->opt_step, a->ok_flag, stream));
  }
  return MD_OK;
}
