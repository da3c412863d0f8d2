// vae_engine.hip — host side of the 'var' (variational autoencoder + MS-SSIM) path: tensor table, workspace, launch
// sequence, extern "C" ABI of include/cae_vae.h.  Convolutions, BatchNorm, Linear and reductions are the kernels of
// kernels_unet.h / kernels_unet_mfma.h; this path adds kernels_vae.h.
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
#include "kernels_unet_mfma.h"
#include "kernels_vae.h"

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

constexpr float kEps = 1e-5f, kMomentum = 0.1f;
constexpr int kLossSlots = 4096;

struct VBn {
    int C = 0;
    int64_t gamma = 0, beta = 0, rmean = 0, rvar = 0, saved = 0, sums = 0, bsums = 0;
};

struct VConv {
    Geom g;
    int64_t w = 0, b = 0;
    bool has_bn = false;
    VBn bn;
    int64_t z = 0, a = 0, gz = 0, ga = 0;   // raw output, activated output, grad wrt raw (in place of BN-bwd), grad wrt input
};

struct VFc {
    int nin = 0, nout = 0;
    int64_t w = 0, b = 0;
    int64_t h = 0, a = 0, gh = 0, gin = 0;
};

struct VData {
    const float* x = nullptr;
    const float* t = nullptr;
    int64_t n = 0;
};

}  // namespace

struct vae_engine {
    std::vector<VConv> enc, dec;
    VFc fc0, mu, lv, d0, d1;   // enc_lin.0, heads, decoder_lin.0, decoder_lin.2
    int fc_size = 0, latent = 0, max_batch = 0;
    int in_c = 0, in_h = 0, in_w = 0, out_c = 0, out_h = 0, out_w = 0;
    std::vector<cae_tensor_info_t> tensors;
    int64_t n_params = 0, n_buffers = 0, ws_bytes = 0;
    int64_t off_gacc = 0, off_dsum = 0, n_dsum = 0, off_losses = 0, off_gscratch = 0, off_f32 = 0;
    int64_t xb = 0, z = 0, eps = 0, gz = 0, gmu = 0, glv = 0, y = 0, tb = 0, kappa = 0, nvalid = 0;
    int64_t sx[vae::kScales] = {0}, sy[vae::kScales] = {0}, sA[vae::kScales] = {0}, sB[vae::kScales] = {0},
            sC[vae::kScales] = {0}, sG[vae::kScales] = {0};
    int64_t off_ssum = 0, off_part = 0;   // doubles: [scale][BC][2]; parts {mse, kl, ssim}
    char* ws = nullptr;
    float *params = nullptr, *m = nullptr, *v = nullptr, *buffers = nullptr;
    hipStream_t stream = nullptr;
    vae::AdamHyper hyper{1e-3, 0.9, 0.999, 1e-8, 1e-5};
    double l_mse = 1, l_kl = 1, l_ssim = 1;
    uint32_t seed = 0;
    int64_t step = 0;
    bool gacc_clean = false;
    VData ds[2];
    vae::Gauss gauss;

    float* f(int64_t off) const { return reinterpret_cast<float*>(ws + off_f32) + off; }
    double* dsum(int64_t off) const { return reinterpret_cast<double*>(ws + off_dsum) + off; }
    double* gacc(int64_t off) const { return reinterpret_cast<double*>(ws + off_gacc) + off; }
    float* P(int64_t off) const { return params + off; }
    float* Bf(int64_t off) const { return buffers + off; }
};

namespace {

int64_t add_tensor(vae_engine* e, const std::string& name, int arena, std::vector<int64_t> shape) {
    cae_tensor_info_t t;
    memset(&t, 0, sizeof t);
    snprintf(t.name, sizeof t.name, "%s", name.c_str());
    t.arena = arena;
    t.ndim = (int)shape.size();
    int64_t n = 1;
    for (size_t i = 0; i < shape.size(); i++) t.shape[i] = shape[i], n *= shape[i];
    int64_t& top = arena == 0 ? e->n_params : e->n_buffers;
    top = (top + 3) & ~int64_t(3);
    t.offset = top;
    t.numel = n;
    top += n;
    e->tensors.push_back(t);
    return t.offset;
}

void add_bn(vae_engine* e, const std::string& key, int C, VBn& bn) {
    bn.C = C;
    bn.gamma = add_tensor(e, key + ".weight", 0, {C});
    bn.beta = add_tensor(e, key + ".bias", 0, {C});
    bn.rmean = add_tensor(e, key + ".running_mean", 1, {C});
    bn.rvar = add_tensor(e, key + ".running_var", 1, {C});
}

int blocks_for(long long n, int cap = 65536) {
    long long b = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

dim3 ew_grid(int B, int C, int HW) {
    int chunks = (int)(((long long)B * HW + 256 * 4 - 1) / (256 * 4));
    chunks = std::max(1, std::min(chunks, 128));
    return dim3(chunks, C);
}

const Drop kNoDrop{0, 0, 1.f, 0};

void conv_down(vae_engine* e, const Geom& g, const float* L, const float* w, const float* bias, float* S) {
    hipLaunchKernelGGL(k_down, dim3(blocks_for((long long)g.B * g.Cs * g.Hs * g.Ws)), dim3(256), 0, e->stream, g, L, w, bias, S);
}
void conv_up(vae_engine* e, const Geom& g, const float* S, const float* w, const float* bias, float* L) {
    hipLaunchKernelGGL(k_up, dim3(blocks_for((long long)g.B * g.Cl * g.Hl * g.Wl)), dim3(256), 0, e->stream, g, S, w, bias, L);
}
void conv_wgrad(vae_engine* e, const Geom& g, const float* S, const float* L, double* acc) {
    const long long per = (long long)g.B * g.Hs * g.Ws;
    const int split = (int)std::max<long long>(1, std::min<long long>((per + 4095) / 4096, 16));
    hipLaunchKernelGGL(k_wgrad, dim3(g.Cs * g.Cl * g.kh * g.kw, split), dim3(256), 0, e->stream, g, S, L, acc);
}
void chan_sums(vae_engine* e, const float* x, long long bs, int B, int C, int HW, double* sums, int sstride, int want_sq) {
    const int chunks = (int)std::max<long long>(1, std::min<long long>(((long long)B * HW + 2047) / 2048, 64));
    hipLaunchKernelGGL(k_chan_sums, dim3(chunks, C), dim3(256), 0, e->stream, x, bs, B, HW, sums, sstride, want_sq);
}

// conv output z (B,C,HW) -> a = relu(bn(z)); batch statistics when training
void bn_relu(vae_engine* e, VBn& bn, const float* z, int B, int HW, bool train, float* a) {
    if (train) chan_sums(e, z, (long long)bn.C * HW, B, bn.C, HW, e->dsum(bn.sums), 2, 1);
    hipLaunchKernelGGL(k_bn_act, ew_grid(B, bn.C, HW), dim3(256), 0, e->stream, z, (long long)bn.C * HW, B, bn.C, HW, e->f(bn.saved),
                       e->Bf(bn.rmean), e->Bf(bn.rvar), kEps, train ? 2 : 1, e->dsum(bn.sums), (double)B * HW, kMomentum,
                       e->P(bn.gamma), e->P(bn.beta), kNoDrop, (float*)nullptr, a);
}

// g_io: in = grad wrt a, out = grad wrt z
void bn_relu_bwd(vae_engine* e, VBn& bn, const float* ga, const float* z, int B, int HW, float* gz) {
    hipLaunchKernelGGL(k_bn_bwd_reduce, ew_grid(B, bn.C, HW), dim3(256), 0, e->stream, ga, (long long)bn.C * HW,
                       (const float*)nullptr, 0ll, z, (long long)bn.C * HW, B, bn.C, HW, e->f(bn.saved), e->P(bn.gamma),
                       e->P(bn.beta), kNoDrop, gz, e->dsum(bn.bsums));
    hipLaunchKernelGGL(k_bn_bwd_apply, ew_grid(B, bn.C, HW), dim3(256), 0, e->stream, gz, z, (long long)bn.C * HW, B, bn.C, HW,
                       e->f(bn.saved), e->P(bn.gamma), e->dsum(bn.bsums), (double)B * HW, e->gacc(bn.gamma), e->gacc(bn.beta));
}

void lin_fwd(vae_engine* e, const VFc& L, int B, const float* in, float* out) {
    GemmDesc d{L.nout, B, L.nin, e->P(L.w), L.nin, 1, in, 1, L.nin, e->P(L.b), out, nullptr, 1, L.nout, 0};
    gemm_launch(d, reinterpret_cast<double*>(e->ws + e->off_gscratch), e->stream);
}
void lin_bwd(vae_engine* e, const VFc& L, int B, const float* in, const float* gout, float* gin) {
    GemmDesc w{L.nout, L.nin, B, gout, 1, L.nout, in, L.nin, 1, nullptr, nullptr, e->gacc(L.w), L.nin, 1, 2};
    gemm_launch(w, nullptr, e->stream);
    hipLaunchKernelGGL(k_col_sums, dim3((L.nout + 255) / 256), dim3(256), 0, e->stream, B, L.nout, gout, e->gacc(L.b));
    if (gin) {
        GemmDesc d{L.nin, B, L.nout, e->P(L.w), 1, L.nin, gout, 1, L.nout, nullptr, gin, nullptr, 1, L.nin, 0};
        gemm_launch(d, reinterpret_cast<double*>(e->ws + e->off_gscratch), e->stream);
    }
}

uint32_t noise_key(const vae_engine* e) {
    return pcg(pcg(e->seed + 0x9E3779B9u * 977u) ^ (uint32_t)(e->step & 0xFFFFFFFF));
}

int forward(vae_engine* e, const float* x, int B, bool train, double* parts) {
    if (train) VHIP_TRY(hipMemsetAsync(e->dsum(0), 0, (size_t)e->n_dsum * sizeof(double), e->stream));
    const float* cur = x;
    for (auto& L : e->enc) {
        Geom g = L.g;
        g.B = B;
        conv_down(e, g, cur, e->P(L.w), e->P(L.b), e->f(L.z));
        bn_relu(e, L.bn, e->f(L.z), B, g.Hs * g.Ws, train, e->f(L.a));
        cur = e->f(L.a);
    }
    lin_fwd(e, e->fc0, B, cur, e->f(e->fc0.h));
    hipLaunchKernelGGL(k_relu_drop, dim3(blocks_for((long long)B * e->fc_size)), dim3(256), 0, e->stream, e->f(e->fc0.h),
                       (long long)B * e->fc_size, kNoDrop, e->f(e->fc0.a));
    lin_fwd(e, e->mu, B, e->f(e->fc0.a), e->f(e->mu.h));
    lin_fwd(e, e->lv, B, e->f(e->fc0.a), e->f(e->lv.h));
    hipLaunchKernelGGL(vae::k_reparam, dim3(1), dim3(256), 0, e->stream, e->f(e->mu.h), e->f(e->lv.h), B * e->latent,
                       noise_key(e), train ? 1 : 0, e->f(e->z), e->f(e->eps), parts + 1);
    lin_fwd(e, e->d0, B, e->f(e->z), e->f(e->d0.h));
    hipLaunchKernelGGL(k_relu_drop, dim3(blocks_for((long long)B * e->fc_size)), dim3(256), 0, e->stream, e->f(e->d0.h),
                       (long long)B * e->fc_size, kNoDrop, e->f(e->d0.a));
    lin_fwd(e, e->d1, B, e->f(e->d0.a), e->f(e->d1.h));
    cur = e->f(e->d1.h);
    for (auto& L : e->dec) {
        Geom g = L.g;
        g.B = B;
        conv_up(e, g, cur, e->P(L.w), e->P(L.b), e->f(L.z));
        if (!L.has_bn) break;
        bn_relu(e, L.bn, e->f(L.z), B, g.Hl * g.Wl, train, e->f(L.a));
        cur = e->f(L.a);
    }
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
}

// y, target batch, MS-SSIM + MSE; with want_grad the gradient wrt the last layer's raw output goes to dec.back().gz
int loss(vae_engine* e, int which, const int32_t* perm, int64_t start, int B, int slot, bool want_grad, double* parts) {
    VConv& L = e->dec.back();
    const int C = e->out_c, H = e->out_h, W = e->out_w, BC = B * C;
    const long long E = (long long)C * H * W, n = (long long)B * E;
    hipLaunchKernelGGL(vae::k_sigmoid_gather, dim3(blocks_for(n)), dim3(256), 0, e->stream, e->f(L.z), e->ds[which].t, perm,
                       (long long)start, B, E, e->f(e->sx[0]), e->f(e->sy[0]));
    double* ssum = reinterpret_cast<double*>(e->ws + e->off_ssum);
    VHIP_TRY(hipMemsetAsync(ssum, 0, (size_t)vae::kScales * BC * 2 * sizeof(double), e->stream));
    int h = H, w = W;
    for (int s = 0; s < vae::kScales; s++) {
        if (s > 0) {
            hipLaunchKernelGGL(vae::k_pool2, dim3(blocks_for((long long)BC * h * w / 4)), dim3(256), 0, e->stream, e->f(e->sx[s - 1]),
                               BC, h, w, e->f(e->sx[s]));
            hipLaunchKernelGGL(vae::k_pool2, dim3(blocks_for((long long)BC * h * w / 4)), dim3(256), 0, e->stream, e->f(e->sy[s - 1]),
                               BC, h, w, e->f(e->sy[s]));
            h /= 2, w /= 2;
        }
        const dim3 grid((w - vae::kHalo + vae::kTile - 1) / vae::kTile, (h - vae::kHalo + vae::kTile - 1) / vae::kTile, BC);
        hipLaunchKernelGGL(vae::k_ssim_fwd, grid, dim3(256), 0, e->stream, e->f(e->sx[s]), e->f(e->sy[s]), h, w, e->gauss, 1e-4f,
                           9e-4f, s == vae::kScales - 1 ? 1 : 0, ssum + (size_t)s * BC * 2, e->f(e->sA[s]), e->f(e->sB[s]),
                           e->f(e->sC[s]));
    }
    int nv[vae::kScales];
    for (int s = 0, hh = H, ww = W; s < vae::kScales; s++, hh /= 2, ww /= 2) nv[s] = (hh - vae::kHalo) * (ww - vae::kHalo);
    VHIP_TRY(hipMemcpyAsync(e->f(e->nvalid), nv, sizeof nv, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(vae::k_msssim_finalize, dim3(1), dim3(256), 0, e->stream, ssum, BC, reinterpret_cast<const int*>(e->f(e->nvalid)),
                       (float)e->l_ssim, parts + 2, e->f(e->kappa));
    if (want_grad) {
        for (int s = vae::kScales - 1; s >= 0; s--) {
            const int hs = H >> s, wsz = W >> s;
            const dim3 grid((wsz + vae::kTile - 1) / vae::kTile, (hs + vae::kTile - 1) / vae::kTile, BC);
            hipLaunchKernelGGL(vae::k_ssim_bwd, grid, dim3(256), 0, e->stream, e->f(e->sx[s]), e->f(e->sy[s]), hs, wsz, e->gauss,
                               e->f(e->sA[s]), e->f(e->sB[s]), e->f(e->sC[s]), e->f(e->kappa), s,
                               s == vae::kScales - 1 ? (const float*)nullptr : e->f(e->sG[s + 1]), e->f(e->sG[s]));
        }
    }
    hipLaunchKernelGGL(vae::k_vae_loss_grad, dim3(blocks_for(n, 1024)), dim3(256), 0, e->stream, e->f(e->sx[0]), e->f(e->sy[0]),
                       want_grad ? e->f(e->sG[0]) : (const float*)nullptr, n, (float)e->l_mse,
                       want_grad ? e->f(L.gz) : (float*)nullptr, parts + 0);
    (void)slot;
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int backward(vae_engine* e, const float* x, int B) {
    if (!e->gacc_clean) VHIP_TRY(hipMemsetAsync(e->gacc(0), 0, (size_t)e->n_params * sizeof(double), e->stream));
    e->gacc_clean = false;
    const int nd = (int)e->dec.size(), ne = (int)e->enc.size();
    for (int j = nd - 1; j >= 0; j--) {
        VConv& L = e->dec[j];
        Geom g = L.g;
        g.B = B;
        const int HW = g.Hl * g.Wl;
        if (L.has_bn) bn_relu_bwd(e, L.bn, e->f(e->dec[j + 1].ga), e->f(L.z), B, HW, e->f(L.gz));
        const float* in = j == 0 ? e->f(e->d1.h) : e->f(e->dec[j - 1].a);
        conv_wgrad(e, g, in, e->f(L.gz), e->gacc(L.w));
        // a bias in front of a BatchNorm has no gradient; the last layer's does
        if (!L.has_bn) chan_sums(e, e->f(L.gz), (long long)g.Cl * HW, B, g.Cl, HW, e->gacc(L.b), 1, 0);
        conv_down(e, g, e->f(L.gz), e->P(L.w), nullptr, e->f(L.ga));
    }
    // decoder_lin.2 (no activation), decoder_lin.0 (ReLU)
    lin_bwd(e, e->d1, B, e->f(e->d0.a), e->f(e->dec[0].ga), e->f(e->d1.gin));
    hipLaunchKernelGGL(k_relu_drop_bwd, dim3(blocks_for((long long)B * e->fc_size)), dim3(256), 0, e->stream, e->f(e->d1.gin),
                       e->f(e->d0.h), (long long)B * e->fc_size, kNoDrop);
    lin_bwd(e, e->d0, B, e->f(e->z), e->f(e->d1.gin), e->f(e->gz));
    hipLaunchKernelGGL(vae::k_reparam_bwd, dim3(blocks_for((long long)B * e->latent)), dim3(256), 0, e->stream, e->f(e->gz),
                       e->f(e->mu.h), e->f(e->lv.h), e->f(e->eps), B * e->latent, (float)e->l_kl, e->f(e->gmu), e->f(e->glv));
    // heads: both read fc0.a; their input gradients add
    lin_bwd(e, e->mu, B, e->f(e->fc0.a), e->f(e->gmu), e->f(e->mu.gin));
    lin_bwd(e, e->lv, B, e->f(e->fc0.a), e->f(e->glv), e->f(e->lv.gin));
    hipLaunchKernelGGL(vae::k_add_relu_bwd, dim3(blocks_for((long long)B * e->fc_size)), dim3(256), 0, e->stream, e->f(e->mu.gin),
                       e->f(e->lv.gin), e->f(e->fc0.h), (long long)B * e->fc_size);
    lin_bwd(e, e->fc0, B, e->f(e->enc.back().a), e->f(e->mu.gin), e->f(e->fc0.gin));
    const float* ga = e->f(e->fc0.gin);
    for (int i = ne - 1; i >= 0; i--) {
        VConv& L = e->enc[i];
        Geom g = L.g;
        g.B = B;
        const int HW = g.Hs * g.Ws;
        bn_relu_bwd(e, L.bn, ga, e->f(L.z), B, HW, e->f(L.gz));
        const float* in = i == 0 ? x : e->f(e->enc[i - 1].a);
        conv_wgrad(e, g, e->f(L.gz), in, e->gacc(L.w));
        if (i > 0) {
            conv_up(e, g, e->f(L.gz), e->P(L.w), nullptr, e->f(L.ga));
            ga = e->f(L.ga);
        }
    }
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
    const long long E = (long long)e->in_c * e->in_h * e->in_w;
    hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)batch * E)), dim3(256), 0, e->stream, e->ds[which].x, perm, (long long)start,
                       batch, E, e->f(e->xb));
    double* parts = reinterpret_cast<double*>(e->ws + e->off_part);
    VHIP_TRY(hipMemsetAsync(parts, 0, 4 * sizeof(double), e->stream));
    if ((rc = forward(e, e->f(e->xb), batch, train, parts))) return rc;
    // KL is a mean over B*latent
    if ((rc = loss(e, which, perm, start, batch, slot, train, parts))) return rc;
    hipLaunchKernelGGL(vae::k_loss_slot, dim3(1), dim3(1), 0, e->stream, parts, (double)batch * e->latent, e->l_mse, e->l_kl, e->l_ssim,
                       reinterpret_cast<double*>(e->ws + e->off_losses) + 4 * (size_t)slot);
    if (train) {
        if ((rc = backward(e, e->f(e->xb), batch))) return rc;
        if (grads_out)
            hipLaunchKernelGGL(k_acc_to_f32, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->gacc(0),
                               grads_out, grad_scale);
        if (optimise) {
            e->step += 1;
            hipLaunchKernelGGL(vae::k_adam_l2, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->params,
                               e->gacc(0), e->m, e->v, e->hyper, (int)e->step);
            e->gacc_clean = true;
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
    auto bad = [&](const char* msg, int i) {
        const int rc = vfail(CAE_ERR_ARG, "vae_engine_create: layer %d: %s", i, msg);
        delete e;
        return rc;
    };
    for (int i = 0; i < n_enc; i++) {
        const cae_layer_spec& l = enc[i];
        if (l.stride < 1 || l.k_h < 1 || l.k_w < 1) return bad("bad kernel / stride", i);
        if ((l.in_h - l.k_h) / l.stride + 1 != l.out_h || (l.in_w - l.k_w) / l.stride + 1 != l.out_w)
            return bad("encoder output size does not follow from Conv2d(kernel, stride)", i);
        if (i > 0 && (enc[i - 1].out_c != l.in_c || enc[i - 1].out_h != l.in_h || enc[i - 1].out_w != l.in_w))
            return bad("encoder input does not match the previous layer's output", i);
        VConv L;
        L.g = Geom{0, l.out_c, l.out_h, l.out_w, l.in_c, l.in_h, l.in_w, l.k_h, l.k_w, l.stride, 0};
        L.has_bn = true;
        e->enc.push_back(L);
    }
    for (int j = 0; j < n_dec; j++) {
        const cae_layer_spec& l = dec[j];
        if (l.stride < 1 || l.k_h < 1 || l.k_w < 1 || l.output_padding < 0) return bad("bad kernel / stride / output_padding", j);
        if ((l.in_h - 1) * l.stride + l.k_h + l.output_padding != l.out_h || (l.in_w - 1) * l.stride + l.k_w + l.output_padding != l.out_w)
            return bad("decoder output size does not follow from ConvTranspose2d(kernel, stride, output_padding)", j);
        if (j > 0 && (dec[j - 1].out_c != l.in_c || dec[j - 1].out_h != l.in_h || dec[j - 1].out_w != l.in_w))
            return bad("decoder input does not match the previous layer's output", j);
        VConv L;
        L.g = Geom{0, l.in_c, l.in_h, l.in_w, l.out_c, l.out_h, l.out_w, l.k_h, l.k_w, l.stride, 0};
        L.has_bn = j != n_dec - 1;
        e->dec.push_back(L);
    }
    e->in_c = enc[0].in_c, e->in_h = enc[0].in_h, e->in_w = enc[0].in_w;
    e->out_c = dec[n_dec - 1].out_c, e->out_h = dec[n_dec - 1].out_h, e->out_w = dec[n_dec - 1].out_w;
    if (e->out_h % 16 || e->out_w % 16 || e->out_h < 176 || e->out_w < 176)
        return bad("the MS-SSIM loss needs an output height and width that are multiples of 16 and at least 176", n_dec - 1);
    const int F = enc[n_enc - 1].out_c * enc[n_enc - 1].out_h * enc[n_enc - 1].out_w;
    const int G = dec[0].in_c * dec[0].in_h * dec[0].in_w;
    for (int i = 0; i < n_enc; i++) {
        VConv& L = e->enc[i];
        const std::string c = "enc/encoder_cnn." + std::to_string(3 * i), b = "enc/encoder_cnn." + std::to_string(3 * i + 1);
        L.w = add_tensor(e, c + ".weight", 0, {L.g.Cs, L.g.Cl, L.g.kh, L.g.kw});
        L.b = add_tensor(e, c + ".bias", 0, {L.g.Cs});
        add_bn(e, b, L.g.Cs, L.bn);
    }
    auto add_fc = [&](VFc& L, const std::string& key, int nin, int nout) {
        L.nin = nin, L.nout = nout;
        L.w = add_tensor(e, key + ".weight", 0, {nout, nin});
        L.b = add_tensor(e, key + ".bias", 0, {nout});
    };
    add_fc(e->fc0, "enc/encoder_lin.0", F, fc_size);
    add_fc(e->mu, "enc/encoder_mu", fc_size, latent_size);
    add_fc(e->lv, "enc/encoder_logvar", fc_size, latent_size);
    add_fc(e->d0, "dec/decoder_lin.0", latent_size, fc_size);
    add_fc(e->d1, "dec/decoder_lin.2", fc_size, G);
    for (int j = 0; j < n_dec; j++) {
        VConv& L = e->dec[j];
        const std::string c = "dec/decoder_conv." + std::to_string(3 * j), b = "dec/decoder_conv." + std::to_string(3 * j + 1);
        L.w = add_tensor(e, c + ".weight", 0, {L.g.Cs, L.g.Cl, L.g.kh, L.g.kw});
        L.b = add_tensor(e, c + ".bias", 0, {L.g.Cl});
        if (L.has_bn) add_bn(e, b, L.g.Cl, L.bn);
    }
    e->n_params = (e->n_params + 3) & ~int64_t(3);
    e->n_buffers = (e->n_buffers + 3) & ~int64_t(3);
    // ---- workspace ----
    const int64_t B = max_batch;
    int64_t nd = 0;
    auto carve_bn = [&](VBn& bn) { bn.sums = nd, nd += 2 * bn.C, bn.bsums = nd, nd += 2 * bn.C; };
    for (auto& L : e->enc) carve_bn(L.bn);
    for (auto& L : e->dec)
        if (L.has_bn) carve_bn(L.bn);
    e->n_dsum = nd;
    int64_t nf = 0;
    auto F32 = [&](int64_t n) {
        const int64_t o = nf;
        nf += (n + 63) & ~int64_t(63);
        return o;
    };
    e->xb = F32(B * e->in_c * e->in_h * e->in_w);
    for (auto& L : e->enc) {
        const int64_t n = B * L.g.Cs * L.g.Hs * L.g.Ws, ni = B * L.g.Cl * L.g.Hl * L.g.Wl;
        L.z = F32(n), L.a = F32(n), L.gz = F32(n), L.ga = F32(ni);
        L.bn.saved = F32(2 * L.bn.C);
    }
    auto carve_fc = [&](VFc& L) { L.h = F32(B * L.nout), L.a = F32(B * L.nout), L.gh = F32(B * L.nout), L.gin = F32(B * L.nin); };
    carve_fc(e->fc0), carve_fc(e->mu), carve_fc(e->lv), carve_fc(e->d0), carve_fc(e->d1);
    e->z = F32(B * latent_size), e->eps = F32(B * latent_size), e->gz = F32(B * latent_size);
    e->gmu = F32(B * latent_size), e->glv = F32(B * latent_size);
    for (auto& L : e->dec) {
        const int64_t nl = B * L.g.Cl * L.g.Hl * L.g.Wl, ns = B * L.g.Cs * L.g.Hs * L.g.Ws;
        L.z = F32(nl), L.gz = F32(nl), L.ga = F32(ns);
        if (L.has_bn) L.a = F32(nl), L.bn.saved = F32(2 * L.bn.C);
    }
    const int64_t BC = B * e->out_c;
    for (int s = 0, h = e->out_h, w = e->out_w; s < vae::kScales; s++, h /= 2, w /= 2) {
        e->sx[s] = F32(BC * h * w), e->sy[s] = F32(BC * h * w), e->sG[s] = F32(BC * h * w);
        const int64_t nv = BC * (h - vae::kHalo) * (w - vae::kHalo);
        e->sA[s] = F32(nv), e->sB[s] = F32(nv), e->sC[s] = F32(nv);
    }
    e->kappa = F32(BC * vae::kScales), e->nvalid = F32(16);
    int64_t off = 0;
    auto bytes = [&](int64_t n) {
        const int64_t o = off;
        off += (n + 255) & ~int64_t(255);
        return o;
    };
    e->off_gacc = bytes(e->n_params * 8);
    e->off_dsum = bytes(nd * 8);
    e->off_losses = bytes((int64_t)kLossSlots * 4 * 8);
    e->off_ssum = bytes((int64_t)vae::kScales * BC * 2 * 8);
    e->off_part = bytes(4 * 8);
    e->off_gscratch = bytes((int64_t)std::max({F, G, fc_size, latent_size}) * B * 8);
    e->off_f32 = bytes(nf * 4);
    e->ws_bytes = off;
    *out = e;
    return CAE_OK;
}

void vae_engine_destroy(vae_engine* e) { delete e; }
int64_t vae_param_count(const vae_engine* e) { return e ? e->n_params : 0; }
int64_t vae_buffer_count(const vae_engine* e) { return e ? e->n_buffers : 0; }
int vae_tensor_count(const vae_engine* e) { return e ? (int)e->tensors.size() : 0; }
int vae_tensor_info(const vae_engine* e, int index, cae_tensor_info_t* out) {
    if (!e || !out || index < 0 || index >= (int)e->tensors.size()) return vfail(CAE_ERR_ARG, "vae_tensor_info: bad argument");
    *out = e->tensors[index];
    return CAE_OK;
}
int64_t vae_workspace_bytes(const vae_engine* e) { return e ? e->ws_bytes : 0; }

int vae_bind(vae_engine* e, float* params, float* m, float* v, float* buffers, void* workspace, int64_t workspace_bytes) {
    if (!e || !params || !m || !v || !buffers || !workspace) return vfail(CAE_ERR_ARG, "vae_bind: null pointer");
    if (workspace_bytes < e->ws_bytes || ((uintptr_t)workspace & 255)) return vfail(CAE_ERR_ARG, "vae_bind: workspace too small or misaligned");
    e->params = params, e->m = m, e->v = v, e->buffers = buffers, e->ws = (char*)workspace;
    return CAE_OK;
}
int vae_set_stream(vae_engine* e, void* hip_stream) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_set_stream: null engine");
    e->stream = (hipStream_t)hip_stream;
    return CAE_OK;
}
int vae_set_hyper(vae_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay, double lambda_mse,
                  double lambda_kl, double lambda_ssim, uint32_t noise_seed) {
    if (!e) return vfail(CAE_ERR_ARG, "vae_set_hyper: null engine");
    e->hyper = vae::AdamHyper{lr, beta1, beta2, eps, weight_decay};
    e->l_mse = lambda_mse, e->l_kl = lambda_kl, e->l_ssim = lambda_ssim, e->seed = noise_seed;
    return CAE_OK;
}
int vae_set_step(vae_engine* e, int64_t step) {
    if (!e || step < 0) return vfail(CAE_ERR_ARG, "vae_set_step: bad argument");
    e->step = step;
    return CAE_OK;
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
    hipLaunchKernelGGL(k_f32_to_acc, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, grads, e->gacc(0));
    e->step += 1;
    hipLaunchKernelGGL(vae::k_adam_l2, dim3(blocks_for(e->n_params)), dim3(256), 0, e->stream, (long long)e->n_params, e->params,
                       e->gacc(0), e->m, e->v, e->hyper, (int)e->step);
    e->gacc_clean = true;
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
}
int vae_eval_step(vae_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    return step_common(e, which, perm, start, batch, loss_slot, false, nullptr, false);
}
int vae_score(vae_engine* e, const float* x, int batch, float* y) {
    if (!e || !e->ws) return vfail(CAE_ERR_STATE, "vae_score: engine is not bound");
    if (!x || !y || batch < 1 || batch > e->max_batch) return vfail(CAE_ERR_ARG, "vae_score: bad argument");
    double* parts = reinterpret_cast<double*>(e->ws + e->off_part);
    int rc = forward(e, x, batch, false, parts);
    if (rc) return rc;
    const long long n = (long long)batch * e->out_c * e->out_h * e->out_w;
    hipLaunchKernelGGL(k_sigmoid, dim3(blocks_for(n)), dim3(256), 0, e->stream, e->f(e->dec.back().z), n, y);
    VHIP_TRY(hipGetLastError());
    return CAE_OK;
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
