// trunk_api.h — the ConvAE engine as the convolutional TRUNK of another model in the same library (internal C++ interface
// between engine.hip and vae_engine.hip; not part of the C ABI).
//
// The 'var' model (include/cae_vae.h) is the ConvAE encoder / decoder stack with two changes: the encoder's second Linear
// layer is a pair of heads (mu, logvar) followed by the reparameterisation, and the loss is computed outside the trunk
// (MSE + KL + MS-SSIM), so the last decoder layer hands out its RAW output and takes the gradient with respect to it.
// A trunk engine is a cae_engine created in that mode: it runs every convolution, BatchNorm, Linear layer and the optimiser
// with the ConvAE path's kernels (kernels_ctlds.h, kernels_rows.h, kernels_s2.h, kernels_igemm.h, kernels_gemm.h, k_adam)
// and calls back for the two places where the models differ.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cae_hip.h"

namespace cae_internal {

struct TrunkHooks {
    // heads (B, 2*latent) rows [mu | logvar] -> z (B, latent), on stream s (train: sampled; eval: z = mu)
    void (*reparam)(void* user, hipStream_t s, const float* heads, int B, int latent, int train, float* z);
    // dL/dz (B, latent) + heads -> dL/dheads (B, 2*latent) rows [dmu | dlogvar] (the KL term's gradient is added here)
    void (*reparam_bwd)(void* user, hipStream_t s, const float* gz, const float* heads, int B, int latent, float* gheads);
    void* user;
};

// Tensor table: enc convs, "enc/encoder_lin.0", "enc/encoder_mu", "enc/encoder_logvar", "dec/decoder_lin.{0,2}", dec convs.
// The two heads' weights are adjacent in the parameter arena (one (2*latent, fc) matrix), so are their biases.
int trunk_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size, int latent_size,
                 int max_batch, cae_engine** out);
void trunk_set_hooks(cae_engine* e, const TrunkHooks& hooks);

// Train-mode or eval-mode forward of an explicit input batch x (B, in_c, in_h, in_w).  external_loss: the last decoder
// layer's raw output is left in trunk_raw_output(); otherwise its sigmoid goes to yhat (scoring).
int trunk_forward(cae_engine* e, const float* x_dev, int batch, bool train, bool external_loss, float* yhat_dev);
float* trunk_raw_output(cae_engine* e);          // (B, out_c, out_h, out_w)
float* trunk_output_gradient(cae_engine* e);     // where the caller leaves dL/d(raw output) before trunk_backward
double* trunk_output_bias_acc(cae_engine* e);    // fp64 accumulator of the last layer's bias gradient (the caller adds sum dL/d(raw))
// backward of the last trunk_forward(train) through every layer into the fp64 accumulators
int trunk_backward(cae_engine* e, const float* x_dev, int batch);
// Adam (L2 decay) from the accumulators, which it clears (cae_set_hyper / cae_set_adam_step configure it)
int trunk_adam(cae_engine* e);
// accumulators -> fp32 gradient * scale in out_dev (n_param floats), cleared; trunk_adam_from applies Adam from such a buffer
int trunk_gradients(cae_engine* e, float* out_dev, double scale);
int trunk_adam_from(cae_engine* e, const float* grads_dev);

}  // namespace cae_internal
