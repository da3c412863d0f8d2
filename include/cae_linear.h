/* cae_linear.h — C ABI of the LinearModel path of libcae_hip.so (gfx950 / MI355X).
 *
 * Replaces, for `--method linear` (reference files under src/cae_tools/models/):
 *   linear.py:19-35            Linear: Flatten -> nn.Linear(C1*y1*x1, C2*y2*x2) -> Unflatten     -> lin_score / the forward half of a step
 *   linear_model.py:142-159    LinearModel.__train_epoch (MSELoss :241, Adam(lr, weight_decay) :247) -> lin_train_step
 *   linear_model.py:161-175    __test_epoch                                                          -> lin_eval_step
 *   linear_model.py:177-184    score                                                                 -> lin_score
 * The parameter arena is [weight (nout, nin) row-major, bias (nout)] = the state_dict entries linear.1.weight / linear.1.bias.
 * Conventions as in cae_hip.h.  The GEMMs run on the MFMA tile engine of kernels_unet_mfma.h (the 256 x 65536 weight of
 * the 16x16 -> 256x256 configuration makes every pass weight-bandwidth-bound).
 */
#ifndef CAE_LINEAR_H
#define CAE_LINEAR_H

#include <stdint.h>

#include "cae_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lin_engine lin_engine;

int lin_engine_create(int64_t n_in, int64_t n_out, int max_batch, lin_engine** out);
void lin_engine_destroy(lin_engine* e);
int64_t lin_param_count(const lin_engine* e);       /* n_out * n_in + n_out (bias offset = n_out * n_in) */
int64_t lin_workspace_bytes(const lin_engine* e);
int lin_bind(lin_engine* e, float* params_dev, float* exp_avg_dev, float* exp_avg_sq_dev, void* workspace_dev,
             int64_t workspace_bytes);
int lin_set_stream(lin_engine* e, void* hip_stream);
int lin_set_hyper(lin_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay);
int lin_set_step(lin_engine* e, int64_t completed_steps);
int lin_set_dataset(lin_engine* e, int which, const float* x_dev, const float* target_dev, int64_t n);
/* one iteration of __train_epoch on samples perm[start .. start+batch): forward, MSE, backward, Adam; loss -> slot */
int lin_train_step(lin_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
/* the same without the optimiser step; fp32 gradient -> grads_dev (lin_param_count floats) */
int lin_forward_backward(lin_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot,
                          float* grads_dev, double grad_scale);
/* Data parallelism: every rank calls lin_forward_backward with grad_scale = local batch / global batch, the ranks SUM-all-reduce
 * grads_dev (torch.distributed on the same stream), then each applies the Adam step to the reduced gradient. */
int lin_apply_gradients(lin_engine* e, const float* grads_dev);
int lin_eval_step(lin_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
int lin_score(lin_engine* e, const float* x_dev, int batch, float* y_dev);
int lin_loss_slots(const lin_engine* e);
int lin_read_losses(lin_engine* e, int first_slot, int count, double* out_host);
int lin_sync(lin_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* CAE_LINEAR_H */
