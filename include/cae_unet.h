/* cae_unet.h — C ABI of the UNET path of libcae_hip.so (gfx950 / MI355X).
 *
 * Replaces, for `--method unet`, these reference sites (file:line under src/cae_tools/models/unet.py):
 *
 *   :73-112   Encoder.__init__/forward  (Conv2d+BN+ReLU+Dropout stack, skip list, Linear-BN1d-ReLU-Dropout-Linear-ReLU-Dropout)
 *   :114-163  Decoder.__init__/forward  (Linear stack, ConvTranspose2d, ChannelAttention gate, skip concat, BN, ReLU, Dropout, sigmoid)
 *   :23-39    ChannelAttention
 *   :295-337  UNET.__train_epoch         -> unet_train_step per batch + unet_read_losses
 *   :339-370  UNET.__test_epoch          -> unet_eval_step per batch + unet_read_losses
 *   :373-382  UNET.score                 -> unet_score
 *   :457      torch.optim.AdamW(lr, weight_decay); :459 cosine schedule with eta_min == lr (constant rate)
 *   :635-639  masked_mse_loss, :641-678 pearson_corr_torch, combined at :316-321
 *
 * Layer geometry is cae_layer_spec (cae_hip.h); as in the reference's UNET modules the `output_padding` field
 * is the PADDING of both Conv2d (:82) and ConvTranspose2d (:140).  The model must have as many decoder as encoder
 * layers; decoder layer j < n-1 is followed by attention + concat with the encoder's ReLU output of layer n-2-j.
 *
 * Same conventions as cae_hip.h: plain pointers and sizes, *_dev = caller-owned device memory, every call returns
 * 0 or a negative cae_status with the message in cae_last_error(), work is enqueued on the stream given to
 * unet_set_stream.  All arithmetic fp32, reductions fp64.
 *
 * Dropout masks are a pure function of (seed, step, site, element index) (kernels_unet.h: pcg hash), not of
 * torch's generator: with dropout_rate 0 a step is the reference's arithmetic exactly.
 */
#ifndef CAE_UNET_H
#define CAE_UNET_H

#include <stdint.h>

#include "cae_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct unet_engine unet_engine;

int unet_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                       int latent_size, int max_batch, unet_engine** out);
void unet_engine_destroy(unet_engine* e);

int64_t unet_param_count(const unet_engine* e);   /* floats in the parameter arena */
int64_t unet_buffer_count(const unet_engine* e);  /* floats in the running-statistics arena */
int unet_tensor_count(const unet_engine* e);
/* names are the reference's state_dict keys with an "enc/" or "dec/" prefix, e.g. "dec/attention_layers.0.fc1.weight" */
int unet_tensor_info(const unet_engine* e, int index, cae_tensor_info_t* out);
int64_t unet_workspace_bytes(const unet_engine* e);

/* params / exp_avg / exp_avg_sq: unet_param_count floats each; buffers: unet_buffer_count floats;
 * workspace: unet_workspace_bytes bytes, 256-B aligned, zero-filled by the caller before the first use. */
int unet_bind(unet_engine* e, float* params_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* buffers_dev,
              void* workspace_dev, int64_t workspace_bytes);
int unet_set_stream(unet_engine* e, void* hip_stream);
/* 1 (default): MFMA kernels where a layer is eligible (4x4, stride 2, padding 1); 0: shape-generic kernels */
int unet_set_kernel_mode(unet_engine* e, int specialised);

/* unet.py:201-204 hyper-parameters; dropout_seed keys the dropout hash */
int unet_set_hyper(unet_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay,
                   double dropout_rate, double lambda_pearson, uint32_t dropout_seed);
/* completed optimiser steps (AdamW bias correction uses step+1; the dropout hash uses step) */
int unet_set_step(unet_engine* e, int64_t step);

/* Resident data set `which` (0 train, 1 test): x (n, Cin, H, W), target (n, Cout, H, W), mask (n, mask_channels,
 * H, W) with mask_channels 1 or Cout, or mask_dev NULL (= ones; ds_dataset.py:152-156). */
int unet_set_dataset(unet_engine* e, int which, const float* x_dev, const float* target_dev, const float* mask_dev,
                     int mask_channels, int64_t n);

/* One iteration of __train_epoch (:307-325) on samples perm[start .. start+batch) (perm_dev NULL: start..):
 * forward (train mode), masked MSE + lambda*(1 - mean Pearson), backward, AdamW.  {mse, pearson loss} -> loss slot. */
int unet_train_step(unet_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
/* The same without the optimiser step: the fp32 gradient of the loss is written to grads_dev (unet_param_count). */
int unet_forward_backward(unet_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot,
                          float* grads_dev, double grad_scale);
/* Data parallelism: every rank calls unet_forward_backward with grad_scale = local batch / global batch, the ranks SUM-all-reduce
 * grads_dev (torch.distributed on the same stream), then each applies the AdamW step to the reduced gradient. */
int unet_apply_gradients(unet_engine* e, const float* grads_dev);
/* One iteration of __test_epoch (:347-361): eval-mode forward + the two losses -> loss slot. */
int unet_eval_step(unet_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
/* UNET.score (:373-382): eval-mode forward of x (batch, Cin, H, W) -> y (batch, Cout, H, W). */
int unet_score(unet_engine* e, const float* x_dev, int batch, float* y_dev);
int unet_loss_slots(const unet_engine* e);
/* blocking: out[2*i] = mse, out[2*i+1] = 1 - mean Pearson of slot first+i */
int unet_read_losses(unet_engine* e, int first_slot, int count, double* out_host);
int unet_sync(unet_engine* e);
/* blocking debug read of an internal activation ("enc_z0", "enc_s0", "dec_u1", "dec_cat0", "att0", "y" ...) */
int unet_debug_read(unet_engine* e, const char* what, float* out_host, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* CAE_UNET_H */
