// linear_engine.hip — the LinearModel path (include/cae_linear.h): one nn.Linear between the flattened input and the
// flattened output, MSE, Adam.  Three GEMMs per step on the MFMA tile engine, all of them bound by the weight matrix.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "cae_linear.h"
#include "kernels_unet.h"
#include "kernels_unet_mfma.h"
#include "kernels_vae.h"

void cae_detail_set_error(const char* msg);

using namespace unet;

namespace {

int lfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cae_detail_set_error(buf);
    return code;
}

#define LHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return lfail(CAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int kLossSlots = 4096;

// g = 2 (y - t) / n (optional);  loss_out += sum (y - t)^2 / n;  t gathered through perm
__global__ void __launch_bounds__(256) k_mse(const float* __restrict__ y, const float* __restrict__ target, const int* __restrict__ perm,
                                             long long start, int B, long long E, float* __restrict__ g, double* __restrict__ loss_out) {
    __shared__ double red[4];
    const long long n = (long long)B * E;
    const float k = 2.f / (float)n;
    double s = 0;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
        const long long b = o / E, i = o - b * E;
        const long long smp = perm ? (long long)perm[start + b] : start + b;
        const float d = y[o] - target[smp * E + i];
        s += (double)d * (double)d;
        if (g) g[o] = k * d;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(loss_out, t / (double)n);
}

int blocks_for(long long n, int cap = 65536) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap)); }

}  // namespace

struct lin_engine {
    int64_t nin = 0, nout = 0, n_params = 0, ws_bytes = 0;
    int max_batch = 0;
    int64_t off_gacc = 0, off_losses = 0, off_gscratch = 0, off_xb = 0, off_y = 0, off_g = 0;
    char* ws = nullptr;
    float *params = nullptr, *m = nullptr, *v = nullptr;
    hipStream_t stream = nullptr;
    vae::AdamHyper hyper{1e-3, 0.9, 0.999, 1e-8, 1e-5};
    int64_t step = 0;
    bool gacc_clean = false;
    const float* dx[2] = {nullptr, nullptr};
    const float* dt[2] = {nullptr, nullptr};
    int64_t dn[2] = {0, 0};
    double* gacc() const { return reinterpret_cast<double*>(ws + off_gacc); }
    float* xb() const { return reinterpret_cast<float*>(ws + off_xb); }
    float* y() const { return reinterpret_cast<float*>(ws + off_y); }
    float* g() const { return reinterpret_cast<float*>(ws + off_g); }
    double* slot(int s) const { return reinterpret_cast<double*>(ws + off_losses) + s; }
};

namespace {

int forward(lin_engine* e, const float* x, int B, float* y) {
    // y[b][o] = bias[o] + sum_i x[b][i] W[o][i]
    GemmDesc d{(int)e->nout, B, (int)e->nin, e->params, e->nin, 1, x, 1, e->nin, e->params + e->nout * e->nin, y, nullptr, 1, e->nout, 0};
    gemm_launch(d, reinterpret_cast<double*>(e->ws + e->off_gscratch), e->stream);
    LHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int step_common(lin_engine* e, int which, const int32_t* perm, int64_t start, int batch, int slot, bool train, float* grads_out,
                bool optimise, double grad_scale = 1.0) {
    if (!e || !e->ws) return lfail(CAE_ERR_STATE, "linear: engine is not bound");
    if (which < 0 || which > 1 || !e->dx[which] || !e->dt[which]) return lfail(CAE_ERR_STATE, "linear: data set %d is not set", which);
    if (batch < 1 || batch > e->max_batch) return lfail(CAE_ERR_ARG, "linear: batch %d outside 1..%d", batch, e->max_batch);
    if (start < 0 || start + batch > e->dn[which]) return lfail(CAE_ERR_ARG, "linear: samples outside the data set");
    if (slot < 0 || slot >= kLossSlots) return lfail(CAE_ERR_ARG, "linear: loss slot out of range");
    hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)batch * e->nin)), dim3(256), 0, e->stream, e->dx[which], perm, (long long)start,
                       batch, (long long)e->nin, e->xb());
    int rc = forward(e, e->xb(), batch, e->y());
    if (rc) return rc;
    LHIP_TRY(hipMemsetAsync(e->slot(slot), 0, sizeof(double), e->stream));
    hipLaunchKernelGGL(k_mse, dim3(blocks_for((long long)batch * e->nout, 1024)), dim3(256), 0, e->stream, e->y(), e->dt[which], perm,
                       (long long)start, batch, (long long)e->nout, train ? e->g() : (float*)nullptr, e->slot(slot));
    if (train) {
        if (!e->gacc_clean) LHIP_TRY(hipMemsetAsync(e->gacc(), 0, (size_t)e->n_params * sizeof(double), e->stream));
        e->gacc_clean = false;
        // dW[o][i] = sum_b g[b][o] x[b][i];  db[o] = sum_b g[b][o]
        GemmDesc w{(int)e->nout, (int)e->nin, batch, e->g(), 1, e->nout, e->xb(), e->nin, 1, nullptr, nullptr, e->gacc(), e->nin, 1, 2};
        gemm_launch(w, nullptr, e->stream);
        hipLaunchKernelGGL(k_col_sums, dim3((unsigned)((e->nout + 255) / 256)), dim3(256), 0, e->stream, batch, (int)e->nout, e->g(),
                           e->gacc() + e->nout * e->nin);
        if (grads_out)
            hipLaunchKernelGGL(k_acc_to_f32, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->gacc(),
                               grads_out, grad_scale, F32Ranges{{0, 0, 0, 0}, {0, 0, 0, 0}, 0});
        if (optimise) {
            e->step += 1;
            hipLaunchKernelGGL(vae::k_adam_l2, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->params,
                               e->gacc(), e->m, e->v, e->hyper, (int)e->step);
            e->gacc_clean = true;
        }
    }
    LHIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // namespace

extern "C" {

int lin_engine_create(int64_t n_in, int64_t n_out, int max_batch, lin_engine** out) {
    if (!out || n_in < 1 || n_out < 1 || max_batch < 1 || n_in > (1ll << 30) || n_out > (1ll << 30) || n_in * n_out > (1ll << 31) - 64)
        return lfail(CAE_ERR_ARG, "lin_engine_create: bad argument (sizes must keep n_in * n_out below 2^31)");
    lin_engine* e = new lin_engine();
    e->nin = n_in, e->nout = n_out, e->max_batch = max_batch;
    e->n_params = n_out * n_in + n_out;
    int64_t off = 0;
    auto bytes = [&](int64_t n) {
        const int64_t o = off;
        off += (n + 255) & ~int64_t(255);
        return o;
    };
    e->off_gacc = bytes(e->n_params * 8);
    e->off_losses = bytes((int64_t)kLossSlots * 8);
    e->off_gscratch = bytes(std::max(n_in, n_out) * max_batch * 8);
    e->off_xb = bytes((int64_t)max_batch * n_in * 4);
    e->off_y = bytes((int64_t)max_batch * n_out * 4);
    e->off_g = bytes((int64_t)max_batch * n_out * 4);
    e->ws_bytes = off;
    *out = e;
    return CAE_OK;
}
void lin_engine_destroy(lin_engine* e) { delete e; }
int64_t lin_param_count(const lin_engine* e) { return e ? e->n_params : 0; }
int64_t lin_workspace_bytes(const lin_engine* e) { return e ? e->ws_bytes : 0; }
int lin_bind(lin_engine* e, float* params, float* m, float* v, void* workspace, int64_t workspace_bytes) {
    if (!e || !params || !m || !v || !workspace) return lfail(CAE_ERR_ARG, "lin_bind: null pointer");
    if (workspace_bytes < e->ws_bytes || ((uintptr_t)workspace & 255)) return lfail(CAE_ERR_ARG, "lin_bind: workspace too small or misaligned");
    e->params = params, e->m = m, e->v = v, e->ws = (char*)workspace;
    return CAE_OK;
}
int lin_set_stream(lin_engine* e, void* hip_stream) {
    if (!e) return lfail(CAE_ERR_ARG, "lin_set_stream: null engine");
    e->stream = (hipStream_t)hip_stream;
    return CAE_OK;
}
int lin_set_hyper(lin_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay) {
    if (!e) return lfail(CAE_ERR_ARG, "lin_set_hyper: null engine");
    e->hyper = vae::AdamHyper{lr, beta1, beta2, eps, weight_decay};
    return CAE_OK;
}
int lin_set_step(lin_engine* e, int64_t completed_steps) {
    if (!e || completed_steps < 0) return lfail(CAE_ERR_ARG, "lin_set_step: bad argument");
    e->step = completed_steps;
    return CAE_OK;
}
int lin_set_dataset(lin_engine* e, int which, const float* x, const float* target, int64_t n) {
    if (!e || which < 0 || which > 1 || !x || n < 1) return lfail(CAE_ERR_ARG, "lin_set_dataset: bad argument");
    e->dx[which] = x, e->dt[which] = target, e->dn[which] = n;
    return CAE_OK;
}
int lin_train_step(lin_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    return step_common(e, which, perm, start, batch, loss_slot, true, nullptr, true);
}
int lin_forward_backward(lin_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot, float* grads,
                         double grad_scale) {
    if (!grads) return lfail(CAE_ERR_ARG, "lin_forward_backward: null gradient buffer");
    return step_common(e, which, perm, start, batch, loss_slot, true, grads, false, grad_scale);
}
int lin_apply_gradients(lin_engine* e, const float* grads) {
    if (!e || !e->ws || !grads) return lfail(CAE_ERR_ARG, "lin_apply_gradients: bad argument");
    hipLaunchKernelGGL(k_f32_to_acc, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, grads, e->gacc());
    e->step += 1;
    hipLaunchKernelGGL(vae::k_adam_l2, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->params,
                       e->gacc(), e->m, e->v, e->hyper, (int)e->step);
    e->gacc_clean = true;
    LHIP_TRY(hipGetLastError());
    return CAE_OK;
}
int lin_eval_step(lin_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    return step_common(e, which, perm, start, batch, loss_slot, false, nullptr, false);
}
int lin_score(lin_engine* e, const float* x, int batch, float* y) {
    if (!e || !e->ws) return lfail(CAE_ERR_STATE, "lin_score: engine is not bound");
    if (!x || !y || batch < 1 || batch > e->max_batch) return lfail(CAE_ERR_ARG, "lin_score: bad argument");
    return forward(e, x, batch, y);
}
int lin_loss_slots(const lin_engine* e) { return e ? kLossSlots : 0; }
int lin_read_losses(lin_engine* e, int first_slot, int count, double* out) {
    if (!e || !e->ws || !out || first_slot < 0 || count < 0 || first_slot + count > kLossSlots)
        return lfail(CAE_ERR_ARG, "lin_read_losses: bad argument");
    LHIP_TRY(hipMemcpyAsync(out, e->slot(first_slot), (size_t)count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    LHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}
int lin_sync(lin_engine* e) {
    if (!e) return lfail(CAE_ERR_ARG, "lin_sync: null engine");
    LHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

}  // extern "C"
