/* cae_vae.h — C ABI of the 'var' (variational autoencoder + MS-SSIM) path of libcae_hip.so (gfx950 / MI355X).
 *
 * The reference has NO source for this path: `--method var` is the default of its train_cae CLI (cli/train_cae.py:42)
 * and model_evaluator.py:35 imports cae_tools.models.var_ae_model, but that file is missing from the repository; only
 * the flags --lambda-mse / --lambda-kl / --lambda-ssim (cli/train_cae.py:32-36) and README.md:29 (pytorch_msssim)
 * survive.  The model computed here is the build's own published definition (oracle/vae_oracle.py):
 *   encoder  = the ConvAE encoder stack (encoder.py:40-46) + Linear(F, fc) + ReLU + two heads Linear(fc, latent): mu, logvar
 *   z        = mu + eps * exp(logvar / 2) in training (eps from a counter-based hash), mu in eval / scoring
 *   decoder  = the ConvAE decoder (decoder.py:31-50,73-78)
 *   loss     = lambda_mse * MSE + lambda_kl * KL + lambda_ssim * (1 - MS-SSIM)   (5 scales, 11-tap gaussian, data range 1)
 *   Adam with L2 weight decay (conv_ae_model.py:310)
 * Layer geometry is cae_layer_spec exactly as for the ConvAE engine (no padding; output_padding on the decoder).
 * MS-SSIM needs output height and width that are multiples of 16 and at least 176.
 * Conventions as in cae_hip.h.
 */
#ifndef CAE_VAE_H
#define CAE_VAE_H

#include <stdint.h>

#include "cae_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vae_engine vae_engine;

int vae_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                      int latent_size, int max_batch, vae_engine** out);
void vae_engine_destroy(vae_engine* e);
int64_t vae_param_count(const vae_engine* e);
int64_t vae_buffer_count(const vae_engine* e);
int vae_tensor_count(const vae_engine* e);
/* "enc/encoder_cnn.0.weight", ..., "enc/encoder_lin.0.*", "enc/encoder_mu.*", "enc/encoder_logvar.*", "dec/decoder_lin.{0,2}.*",
 * "dec/decoder_conv.{3j}.*", "dec/decoder_conv.{3j+1}.*" */
int vae_tensor_info(const vae_engine* e, int index, cae_tensor_info_t* out);
int64_t vae_workspace_bytes(const vae_engine* e);
int vae_bind(vae_engine* e, float* params_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* buffers_dev,
             void* workspace_dev, int64_t workspace_bytes);
int vae_set_stream(vae_engine* e, void* hip_stream);
int vae_set_hyper(vae_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay, double lambda_mse,
                  double lambda_kl, double lambda_ssim, uint32_t noise_seed);
/* 1 (default): the MS-SSIM passes run the row-streaming kernels (a wave walks a strip of 64 columns, DPP neighbours, register
 * ring); 0: the LDS tile kernels they replaced.  Same arithmetic in the same order: results equal to fp32 rounding (kept
 * selectable so that the parity tests can say so). */
int vae_set_kernel_mode(vae_engine* e, int mode);
int vae_set_step(vae_engine* e, int64_t step);
int vae_set_dataset(vae_engine* e, int which, const float* x_dev, const float* target_dev, int64_t n);
/* forward (train mode) + loss + backward + Adam on samples perm[start .. start+batch); the loss slot receives
 * {mse, kl, 1 - ms_ssim, total} */
int vae_train_step(vae_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
/* the same without the optimiser step: fp32 gradient of the total loss -> grads_dev (vae_param_count floats) */
int vae_forward_backward(vae_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot,
                          float* grads_dev, double grad_scale);
/* Data parallelism: every rank calls vae_forward_backward with grad_scale = local batch / global batch, the ranks SUM-all-reduce
 * grads_dev (torch.distributed on the same stream), then each applies the Adam step to the reduced gradient. */
int vae_apply_gradients(vae_engine* e, const float* grads_dev);
int vae_eval_step(vae_engine* e, int which, const int32_t* perm_dev, int64_t start, int batch, int loss_slot);
int vae_score(vae_engine* e, const float* x_dev, int batch, float* y_dev);
int vae_loss_slots(const vae_engine* e);
int vae_read_losses(vae_engine* e, int first_slot, int count, double* out_host);   /* 4 doubles per slot */
int vae_sync(vae_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* CAE_VAE_H */
