// vae_engine.hip — host side of the 'var' (variational autoencoder + MS-SSIM) path: the extern "C" ABI of include/cae_vae.h.
// Every convolution, BatchNorm, Linear layer and the optimiser run in a ConvAE engine created in trunk mode (trunk_api.h:
// the LDS-staged / row-streaming / MFMA kernels of the ConvAE path, BatchNorm folded into producers and consumers); this file
// adds what the two models differ in: the reparameterisation between the heads and the decoder (kernels_vae.h, called back
// from the trunk) and the loss - sigmoid, MSE, KL, MS-SSIM and its gradient - between the trunk's forward and backward.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cae_vae.h"
#include "kernels_unet.h"
#include "kernels_vae.h"
#include "trunk_api.h"

void cae_detail_set_error(const char* msg);

using namespace unet;

namespace {

int vfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cae_detail_set_error(buf);
    return code;
}

#define VHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return vfail(CAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int kLossSlots = 4096;

struct VData {
    const float* x = nullptr;
    const float* t = nullptr;
    int64_t n = 0;
};

}  // namespace

struct vae_engine {
    cae_engine* trunk = nullptr;   // convolutions, BatchNorm, Linear layers, Adam: the ConvAE engine in trunk mode
    int fc_size = 0, latent = 0, max_batch = 0;
    int in_c = 0, in_h = 0, in_w = 0, out_c = 0, out_h = 0, out_w = 0;
    int64_t n_params = 0, n_buffers = 0, ws_bytes = 0, trunk_ws = 0;
    int64_t off_trunk = 0, off_grads = 0, off_losses = 0, off_f32 = 0;
    int64_t xb = 0, eps = 0, kappa = 0;
    int64_t sx[vae::kScales] = {0}, sy[vae::kScales] = {0}, sA[vae::kScales] = {0}, sB[vae::kScales] = {0},
            sC[vae::kScales] = {0}, sG[vae::kScales] = {0};
    int64_t off_ssum = 0, off_part = 0;   // doubles: [scale][BC][2]; parts {mse, kl, ssim}
    char* ws = nullptr;
    hipStream_t stream = nullptr;
    double l_mse = 1, l_kl = 1, l_ssim = 1;
    double grad_scale = 1.0;     // data-parallel half-step: every loss gradient is scaled by local / global batch at its source
    uint32_t seed = 0;
    int64_t step = 0;
    bool row_kernels = true;     // MS-SSIM passes: row-streaming kernels (vae_set_kernel_mode 1) or the LDS tile kernels (0)
    VData ds[2];
    vae::Gauss gauss;

    float* f(int64_t off) const { return reinterpret_cast<float*>(ws + off_f32) + off; }
    double* parts() const { return reinterpret_cast<double*>(ws + off_part); }
    ~vae_engine() {
        if (trunk) cae_engine_destroy(trunk);
    }
};

namespace {

int blocks_for(long long n, int cap = 65536) {
    long long b = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

uint32_t noise_key(const vae_engine* e) {
    return pcg(pcg(e->seed + 0x9E3779B9u * 977u) ^ (uint32_t)(e->step & 0xFFFFFFFF));
}

// ---- the two call-backs of the trunk (trunk_api.h) ----------------------------------------------------------------------
void hook_reparam(void* user, hipStream_t s, const float* heads, int B, int latent, int train, float* z) {
    vae_engine* e = static_cast<vae_engine*>(user);
    hipLaunchKernelGGL(vae::k_reparam, dim3(1), dim3(256), 0, s, heads, B, latent, noise_key(e), train, z, e->f(e->eps), e->parts() + 1);
}

void hook_reparam_bwd(void* user, hipStream_t s, const float* gz, const float* heads, int B, int latent, float* gheads) {
    vae_engine* e = static_cast<vae_engine*>(user);
    hipLaunchKernelGGL(vae::k_reparam_bwd, dim3(blocks_for((long long)B * latent)), dim3(256), 0, s, gz, heads, e->f(e->eps), B, latent,
                       (float)(e->l_kl * e->grad_scale), gheads);
}

// y, target batch, MS-SSIM + MSE; with want_grad the gradient wrt the last layer's raw output goes to dec.back().gz
int loss(vae_engine* e, int which, const int32_t* perm, int64_t start, int B, int slot, bool want_grad, double* parts) {
    const int C = e->out_c, H = e->out_h, W = e->out_w, BC = B * C;
    const long long E = (long long)C * H * W, n = (long long)B * E;
    if (!e->row_kernels)   // (the row-kernel path forms the finest level inside its pyramid launch)
        hipLaunchKernelGGL(vae::k_sigmoid_gather, dim3(blocks_for(n)), dim3(256), 0, e->stream, cae_internal::trunk_raw_output(e->trunk), e->ds[which].t, perm,
                           (long long)start, B, E, e->f(e->sx[0]), e->f(e->sy[0]));
    double* ssum = reinterpret_cast<double*>(e->ws + e->off_ssum);   // (cleared by step_common together with the loss parts)
    if (e->row_kernels) {
        // the pooling pyramid of both maps in one launch, the finest scale's walk, then the four coarse scales' walks in one launch
        vae::Pyramid pm;
        for (int s = 0; s < vae::kScales; s++) pm.x[s] = e->f(e->sx[s]), pm.y[s] = e->f(e->sy[s]);
        pm.u = cae_internal::trunk_raw_output(e->trunk), pm.target = e->ds[which].t, pm.perm = perm, pm.start = start, pm.C = C;
        hipLaunchKernelGGL(vae::k_pool_pyramid, dim3(W / 16, H / 16, BC), dim3(256), 0, e->stream, pm, H, W);
        vae::SsimScales set;
        memset(&set, 0, sizeof set);
        set.BC = BC;
        dim3 coarse_grid(1, 1, 1);
        for (int s = 0, h = H, w = W; s < vae::kScales; s++, h /= 2, w /= 2) {
            const int rb = vae::ssim_band_rows(h - vae::kHalo, true), bands = (h - vae::kHalo + rb - 1) / rb;
            const dim3 grid((w - vae::kHalo + vae::kSsimCols - 1) / vae::kSsimCols, (bands + 3) / 4, BC);
            double* ss = ssum + (size_t)s * BC * 2;
            const int lastf = s == vae::kScales - 1 ? 1 : 0;
            if (s == 0) {
                hipLaunchKernelGGL(vae::k_ssim_fwd_rows, grid, dim3(256), 0, e->stream, e->f(e->sx[s]), e->f(e->sy[s]), h, w, rb, e->gauss,
                                   1e-4f, 9e-4f, lastf, ss, e->f(e->sA[s]), e->f(e->sB[s]), e->f(e->sC[s]));
            } else {
                set.s[set.n++] = vae::SsimScale{e->f(e->sx[s]), e->f(e->sy[s]), h, w, rb, lastf, ss, e->f(e->sA[s]), e->f(e->sB[s]), e->f(e->sC[s])};
                coarse_grid.x = std::max(coarse_grid.x, grid.x);
                coarse_grid.y = std::max(coarse_grid.y, grid.y);
            }
        }
        coarse_grid.z = (unsigned)(BC * set.n);
        hipLaunchKernelGGL(vae::k_ssim_fwd_rows_multi, coarse_grid, dim3(256), 0, e->stream, set, e->gauss, 1e-4f, 9e-4f);
    } else {
        int h = H, w = W;
        for (int s = 0; s < vae::kScales; s++) {
            if (s > 0) {
                hipLaunchKernelGGL(vae::k_pool2_pair, dim3(blocks_for((long long)BC * h * w / 4)), dim3(256), 0, e->stream, e->f(e->sx[s - 1]),
                                   e->f(e->sy[s - 1]), BC, h, w, e->f(e->sx[s]), e->f(e->sy[s]));
                h /= 2, w /= 2;
            }
            const dim3 grid((w - vae::kHalo + vae::kTile - 1) / vae::kTile, (h - vae::kHalo + vae::kTile - 1) / vae::kTile, BC);
            hipLaunchKernelGGL(vae::k_ssim_fwd, grid, dim3(256), 0, e->stream, e->f(e->sx[s]), e->f(e->sy[s]), h, w, e->gauss, 1e-4f,
                               9e-4f, s == vae::kScales - 1 ? 1 : 0, ssum + (size_t)s * BC * 2, e->f(e->sA[s]), e->f(e->sB[s]),
                               e->f(e->sC[s]));
        }
    }
    vae::ScaleCounts nv;
    for (int s = 0, hh = H, ww = W; s < vae::kScales; s++, hh /= 2, ww /= 2) nv.v[s] = (hh - vae::kHalo) * (ww - vae::kHalo);
    hipLaunchKernelGGL(vae::k_msssim_finalize, dim3(1), dim3(256), 0, e->stream, ssum, BC, nv,
                       (float)e->l_ssim, parts + 2, e->f(e->kappa));
    if (want_grad && e->row_kernels) {
        // the four coarse scales' own terms in one launch, their pooling chain folded into scale 1's map, then the finest scale
        vae::SsimBwdScales set;
        memset(&set, 0, sizeof set);
        set.BC = BC;
        dim3 cg(1, 1, 1);
        for (int s = 1; s < vae::kScales; s++) {
            const int hs = H >> s, wsz = W >> s;
            const int rb = vae::ssim_band_rows(hs, false), bands = (hs + rb - 1) / rb;
            set.s[set.n++] = vae::SsimBwdScale{e->f(e->sx[s]), e->f(e->sy[s]), hs, wsz, rb, s, e->f(e->sA[s]), e->f(e->sB[s]), e->f(e->sC[s]),
                                               e->f(e->sG[s])};
            cg.x = std::max<unsigned>(cg.x, (wsz + vae::kSsimCols - 1) / vae::kSsimCols);
            cg.y = std::max<unsigned>(cg.y, (bands + 3) / 4);
        }
        cg.z = (unsigned)(BC * set.n);
        hipLaunchKernelGGL(vae::k_ssim_bwd_rows_multi, cg, dim3(256), 0, e->stream, set, e->gauss, e->f(e->kappa));
        hipLaunchKernelGGL(vae::k_ssim_combine, dim3(blocks_for((long long)BC * (H / 2) * (W / 2))), dim3(256), 0, e->stream, e->f(e->sG[1]),
                           e->f(e->sG[2]), e->f(e->sG[3]), e->f(e->sG[4]), BC, H / 2, W / 2);
        {
            const int rb = vae::ssim_band_rows(H, false), bands = (H + rb - 1) / rb;
            const dim3 grid((W + vae::kSsimCols - 1) / vae::kSsimCols, (bands + 3) / 4, BC);
            hipLaunchKernelGGL(vae::k_ssim_bwd_rows, grid, dim3(256), 0, e->stream, e->f(e->sx[0]), e->f(e->sy[0]), H, W, rb, e->gauss,
                               e->f(e->sA[0]), e->f(e->sB[0]), e->f(e->sC[0]), e->f(e->kappa), 0, e->f(e->sG[1]), e->f(e->sG[0]));
        }
    } else if (want_grad) {
        for (int s = vae::kScales - 1; s >= 0; s--) {
            const int hs = H >> s, wsz = W >> s;
            const float* coarse = s == vae::kScales - 1 ? (const float*)nullptr : e->f(e->sG[s + 1]);
            const dim3 grid((wsz + vae::kTile - 1) / vae::kTile, (hs + vae::kTile - 1) / vae::kTile, BC);
            hipLaunchKernelGGL(vae::k_ssim_bwd, grid, dim3(256), 0, e->stream, e->f(e->sx[s]), e->f(e->sy[s]), hs, wsz, e->gauss,
                               e->f(e->sA[s]), e->f(e->sB[s]), e->f(e->sC[s]), e->f(e->kappa), s, coarse, e->f(e->sG[s]));
        }
    }
    hipLaunchKernelGGL(vae::k_vae_loss_grad, dim3(blocks_for(n, 1024)), dim3(256), 0, e->stream, e->f(e->sx[0]), e->f(e->sy[0]),
                       want_grad ? e->f(e->sG[0]) : (const float*)nullptr, n, (float)e->l_mse, (float)e->grad_scale,
                       want_grad ? cae_internal::trunk_output_gradient(e->trunk) : (float*)nullptr, parts + 0,
                       want_grad && C == 1 ? cae_internal::trunk_output_bias_acc(e->trunk) : (double*)nullptr);
    (void)slot;
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int check_batch(vae_engine* e, int which, int64_t start, int batch, int slot, bool need_target) {
    if (!e || !e->ws) return vfail(CAE_ERR_STATE, "vae: engine is not bound");
    if (which < 0 || which > 1 || !e->ds[which].x) return vfail(CAE_ERR_STATE, "vae: data set %d is not set", which);
    if (need_target && !e->ds[which].t) return vfail(CAE_ERR_STATE, "vae: data set %d has no target", which);
    if (batch < 1 || batch > e->max_batch) return vfail(CAE_ERR_ARG, "vae: batch %d outside 1..%d", batch, e->max_batch);
    if (start < 0 || start + batch > e->ds[which].n)
        return vfail(CAE_ERR_ARG, "vae: samples %lld..%lld outside the data set (%lld)", (long long)start,
                     (long long)(start + batch), (long long)e->ds[which].n);
    if (slot < 0 || slot >= kLossSlots) return vfail(CAE_ERR_ARG, "vae: loss slot %d outside 0..%d", slot, kLossSlots - 1);
    return CAE_OK;
}

int step_common(vae_engine* e, int which, const int32_t* perm, int64_t start, int batch, int slot, bool train, float* grads_out,
                bool optimise, double grad_scale = 1.0) {
    int rc = check_batch(e, which, start, batch, slot, true);
    if (rc) return rc;
    e->grad_scale = grad_scale;
    const long long E = (long long)e->in_c * e->in_h * e->in_w;
    hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)batch * E)), dim3(256), 0, e->stream, e->ds[which].x, perm, (long long)start,
                       batch, E, e->f(e->xb));
    double* parts = e->parts();
    // the MS-SSIM sums and the loss parts lie next to each other: one fill
    VHIP_TRY(hipMemsetAsync(e->ws + e->off_ssum, 0, (size_t)(e->off_part + 4 * sizeof(double) - e->off_ssum), e->stream));
    // trunk forward: encoder stack, Linear + heads, z (call-back), decoder stack; the last layer leaves its raw output
    if ((rc = cae_internal::trunk_forward(e->trunk, e->f(e->xb), batch, train, true, nullptr))) return rc;
    // KL is a mean over B*latent
    if ((rc = loss(e, which, perm, start, batch, slot, train, parts))) return rc;
    hipLaunchKernelGGL(vae::k_loss_slot, dim3(1), dim3(1), 0, e->stream, parts, (double)batch * e->latent, e->l_mse, e->l_kl, e->l_ssim,
                       reinterpret_cast<double*>(e->ws + e->off_losses) + 4 * (size_t)slot);
    if (train) {
        // the last layer's bias gradient = sum of dL/d(raw output) per channel (the ConvAE path gets it from its fused loss
        // epilogue; here the loss is ours)
        const int HW = e->out_h * e->out_w;
        const int chunks = (int)std::max<long long>(1, std::min<long long>(((long long)batch * HW + 2047) / 2048, 64));
        if (e->out_c > 1)     // (a single-channel output's sum rides in k_vae_loss_grad)
            hipLaunchKernelGGL(k_chan_sums, dim3(chunks, e->out_c), dim3(256), 0, e->stream, cae_internal::trunk_output_gradient(e->trunk),
                           (long long)e->out_c * HW, batch, HW, cae_internal::trunk_output_bias_acc(e->trunk), 1, 0);
        if ((rc = cae_internal::trunk_backward(e->trunk, e->f(e->xb), batch))) return rc;
        if (grads_out) {
            if ((rc = cae_internal::trunk_gradients(e->trunk, grads_out, 1.0))) return rc;   // already scaled at the sources
            if (optimise) return vfail(CAE_ERR_ARG, "vae: gradients are either handed out or applied");
        } else if (optimise) {
            e->step += 1;     // (on the device the trunk's first training kernel counted it)
            if ((rc = cae_internal::trunk_adam(e->trunk))) return rc;
        }
    }
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // namespace

extern "C" {

int vae_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size, int latent_size,
                      int max_batch, vae_engine** out) {
    if (!enc || !dec || !out || n_enc < 1 || n_dec < 1 || fc_size < 1 || latent_size < 1 || max_batch < 1)
        return vfail(CAE_ERR_ARG, "vae_engine_create: bad argument");
    vae_engine* e = new vae_engine();
    e->fc_size = fc_size, e->latent = latent_size, e->max_batch = max_batch;
    e->gauss = vae::make_gauss();
    e->in_c = enc[0].in_c, e->in_h = enc[0].in_h, e->in_w = enc[0].in_w;
    e->out_c = dec[n_dec - 1].out_c, e->out_h = dec[n_dec - 1].out_h, e->out_w = dec[n_dec - 1].out_w;
    if (e->out_h % 16 || e->out_w % 16 || e->out_h < 176 || e->out_w < 176) {
        delete e;
        return vfail(CAE_ERR_ARG, "vae_engine_create: layer %d: the MS-SSIM loss needs an output height and width that are multiples "
                                  "of 16 and at least 176", n_dec - 1);
    }
    // geometry checks, tensor table and parameter arena: the trunk's (its message is the one cae_last_error() returns)
    if (int rc = cae_internal::trunk_create(enc, n_enc, dec, n_dec, fc_size, latent_size, max_batch, &e->trunk)) {
        delete e;
        return rc;
    }
    cae_internal::trunk_set_hooks(e->trunk, cae_internal::TrunkHooks{hook_reparam, hook_reparam_bwd, e});
    e->n_params = cae_param_count(e->trunk);
    e->n_buffers = cae_buffer_count(e->trunk);
    e->trunk_ws = cae_workspace_bytes(e->trunk);
    // ---- workspace: [trunk][fp32 gradient arena of the trunk][losses][MS-SSIM sums][parts][fp32 maps] ----
    const int64_t B = max_batch;
    int64_t nf = 0;
    auto F32 = [&](int64_t n) {
        const int64_t o = nf;
        nf += (n + 63) & ~int64_t(63);
        return o;
    };
    e->xb = F32(B * e->in_c * e->in_h * e->in_w);
    e->eps = F32(B * latent_size);
    const int64_t BC = B * e->out_c;
    for (int s = 0, h = e->out_h, w = e->out_w; s < vae::kScales; s++, h /= 2, w /= 2) {
        e->sx[s] = F32(BC * h * w), e->sy[s] = F32(BC * h * w), e->sG[s] = F32(BC * h * w);
        const int64_t nv = BC * (h - vae::kHalo) * (w - vae::kHalo);
        e->sA[s] = F32(nv), e->sB[s] = F32(nv), e->sC[s] = F32(nv);
    }
    e->kappa = F32(BC * vae::kScales);
    int64_t off = 0;
    auto bytes = [&](int64_t n) {
        const int64_t o = off;
        off += (n + 255) & ~int64_t(255);
        return o;
    };
    e->off_trunk = bytes(e->trunk_ws);
    e->off_grads = bytes(e->n_params * 4);
    e->off_losses = bytes((int64_t)kLossSlots * 4 * 8);
    e->off_ssum = bytes((int64_t)vae::kScales * BC * 2 * 8);
    e->off_part = bytes(4 * 8);
    e->off_f32 = bytes(nf * 4);
    e->ws_bytes = off;
    *out = e;
    return CAE_OK;
}

void vae_engine_destroy(vae_engine* e) { delete e; }
int64_t vae_param_count(const vae_engine* e) { return e ? e->n_params : 0; }
int64_t vae_buffer_count(const vae_engine* e) { return e ? e->n_buffers : 0; }
int vae_tensor_count(const vae_engine* e) { return e ? cae_tensor_count(e->trunk) : 0; }
int vae_tensor_info(const vae_engine* e, int index, cae_tensor_info_t* out) {
    if (!e || !out) return vfail(CAE_ERR_ARG, "vae_tensor_info: bad argument");
    return cae_tensor_info(e->trunk, index, out);
}
int64_t vae_workspace_bytes(const vae_engine* e) { return e ? e->ws_bytes : 0; }

int vae_bind(vae_engine* e, float* params, float* m, float* v, float* buffers, void* workspace, int64_t workspace_bytes) {
    if (!e || !params || !m || !v || !buffers || !workspace) return vfail(CAE_ERR_ARG, "vae_bind: null pointer");
    if (workspace_bytes < e->ws_bytes || ((uintptr_t)workspace & 255)) return vfail(CAE_ERR_ARG, "vae_bind: workspace too small or misaligned");
    e->ws = (char*)workspace;
    return cae_bind(e->trunk, params, reinterpret_cast<float*>(e->ws + e->off_grads), m, v, buffers, e->ws + e->off_trunk, e->trunk_ws);
}
int vae_set_stream(vae_engine* e, void* hip_stream) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_set_stream: null engine");
    e->stream = (hipStream_t)hip_stream;
    if (int rc = cae_set_stream(e->trunk, hip_stream)) return rc;
    return cae_set_graph_mode(e->trunk, 0);   // the trunk runs as plain launches between this file's own
}
int vae_set_hyper(vae_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay, double lambda_mse,
                  double lambda_kl, double lambda_ssim, uint32_t noise_seed) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_set_hyper: null engine");
    e->l_mse = lambda_mse, e->l_kl = lambda_kl, e->l_ssim = lambda_ssim, e->seed = noise_seed;
    return cae_set_hyper(e->trunk, lr, beta1, beta2, eps, weight_decay);
}
int vae_set_kernel_mode(vae_engine* e, int mode) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_set_kernel_mode: null engine");
    e->row_kernels = (mode & 1) != 0;
    return CAE_OK;
}
int vae_set_step(vae_engine* e, int64_t step) {
    if (!e || step < 0) return vfail(CAE_ERR_ARG, "vae_set_step: bad argument");
    e->step = step;
    return e->ws ? cae_set_adam_step(e->trunk, (int)step) : CAE_OK;
}
int vae_set_dataset(vae_engine* e, int which, const float* x, const float* target, int64_t n) {
    if (!e || which < 0 || which > 1 || !x || n < 1) return vfail(CAE_ERR_ARG, "vae_set_dataset: bad argument");
    e->ds[which] = VData{x, target, n};
    return CAE_OK;
}
int vae_train_step(vae_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    return step_common(e, which, perm, start, batch, loss_slot, true, nullptr, true);
}
int vae_forward_backward(vae_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot, float* grads,
                         double grad_scale) {
    if (!grads) return vfail(CAE_ERR_ARG, "vae_forward_backward: null gradient buffer");
    return step_common(e, which, perm, start, batch, loss_slot, true, grads, false, grad_scale);
}
int vae_apply_gradients(vae_engine* e, const float* grads) {
    if (!e || !e->ws || !grads) return vfail(CAE_ERR_ARG, "vae_apply_gradients: bad argument");
    e->step += 1;     // (on the device vae_forward_backward's first kernel counted it)
    return cae_internal::trunk_adam_from(e->trunk, grads);
}
int vae_eval_step(vae_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    return step_common(e, which, perm, start, batch, loss_slot, false, nullptr, false);
}
int vae_score(vae_engine* e, const float* x, int batch, float* y) {
    if (!e || !e->ws) return vfail(CAE_ERR_STATE, "vae_score: engine is not bound");
    if (!x || !y || batch < 1 || batch > e->max_batch) return vfail(CAE_ERR_ARG, "vae_score: bad argument");
    VHIP_TRY(hipMemsetAsync(e->parts(), 0, 4 * sizeof(double), e->stream));
    return cae_internal::trunk_forward(e->trunk, x, batch, false, false, y);
}
int vae_loss_slots(const vae_engine* e) { return e ? kLossSlots : 0; }
int vae_read_losses(vae_engine* e, int first_slot, int count, double* out) {
    if (!e || !e->ws || !out || first_slot < 0 || count < 0 || first_slot + count > kLossSlots)
        return vfail(CAE_ERR_ARG, "vae_read_losses: bad argument");
    VHIP_TRY(hipMemcpyAsync(out, reinterpret_cast<double*>(e->ws + e->off_losses) + 4 * (size_t)first_slot,
                            (size_t)count * 4 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    VHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}
int vae_sync(vae_engine* e) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_sync: null engine");
    VHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

}  // extern "C"
