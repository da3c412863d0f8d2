// kernels_vae.h — kernels specific to the 'var' (variational autoencoder) path: reparameterisation + KL, and the
// MS-SSIM loss with its gradient.  The definition is the build's own (oracle/vae_oracle.py: the reference has no source
// for this path); the convolutions, BatchNorm, Linear layers and Adam run in the ConvAE engine as a trunk (trunk_api.h).
//
// MS-SSIM (Wang et al. 2003, pytorch_msssim conventions): per scale s, with G = 11-tap gaussian (sigma 1.5), valid:
//   mu_x = G*x, mu_y = G*y, s_xx = G*x^2 - mu_x^2, s_yy = G*y^2 - mu_y^2, s_xy = G*xy - mu_x mu_y
//   cs = (2 s_xy + C2) / (s_xx + s_yy + C2),  l = (2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1),  ssim = l * cs
//   f_s = mean(cs) for s < 4, mean(ssim) for s = 4;  M = prod_s relu(f_s)^w_s;  loss = 1 - mean_(b,c) M
// Forward stores, per valid pixel, the partial derivatives of the per-pixel f with respect to (G*x, G*x^2, G*xy):
//   A = df/d(G*x), Bm = df/d(G*x^2), Cm = df/d(G*xy)
// so that  d mean(f) / d x(q) = (1/N) * [ (G^T A)(q) + 2 x(q) (G^T Bm)(q) + y(q) (G^T Cm)(q) ].
// Both filters are separable and run through LDS: a 16x16 tile needs a 26x26 halo.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_unet.h"

namespace vae {
namespace {   // internal linkage: this header is compiled into more than one translation unit

using unet::block_sum;
using unet::pcg;

constexpr int kWin = 11, kHalo = 10, kTile = 16, kIn = kTile + kHalo;   // 26
constexpr int kScales = 5;
__constant__ float c_ms_weights[kScales] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};

struct Gauss {
    float g[kWin];
};

inline Gauss make_gauss() {
    Gauss w;
    float sum = 0.f;
    for (int i = 0; i < kWin; i++) {
        const float c = (float)(i - kWin / 2);
        w.g[i] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        sum += w.g[i];
    }
    for (int i = 0; i < kWin; i++) w.g[i] /= sum;
    return w;
}

// ---------------------------------------------------------------------------------------------
// reparameterisation and KL
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float normal_hash(uint32_t key, uint32_t idx) {
    const uint32_t h1 = pcg((idx * 2u) ^ key), h2 = pcg((idx * 2u + 1u) ^ key);
    const double u1 = ((double)h1 + 1.0) / 4294967296.0, u2 = (double)h2 / 4294967296.0;
    return (float)(sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2));
}

// heads: (B, 2 * L) rows [mu | logvar] - the two heads are ONE Linear layer of the trunk (trunk_api.h).  i = b * L + j is the
// element's index in the (B, L) latent array (what the noise hash and the oracle count in).
// z = mu + eps * exp(logvar / 2) (train) or mu (eval); kl_out += -0.5 * sum(1 + lv - mu^2 - exp(lv)).  one block per launch chunk
__global__ void __launch_bounds__(256) k_reparam(const float* __restrict__ heads, int B, int L, uint32_t key, int train,
                                                 float* __restrict__ z, float* __restrict__ eps, double* __restrict__ kl_out) {
    __shared__ double red[4];
    double s = 0;
    const int n = B * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / L, j = i - b * L;
        const float m = heads[(size_t)b * 2 * L + j], l = heads[(size_t)b * 2 * L + L + j];
        float e = 0.f;
        if (train) e = normal_hash(key, (uint32_t)i);
        eps[i] = e;
        z[i] = m + e * expf(0.5f * l);
        s += (double)(1.f + l - m * m - expf(l));
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(kl_out, -0.5 * t);
}

// dmu = dz + lambda_kl * mu / n;  dlv = dz * eps * 0.5 * exp(lv/2) + lambda_kl * 0.5 * (exp(lv) - 1) / n;  rows [dmu | dlv]
__global__ void __launch_bounds__(256) k_reparam_bwd(const float* __restrict__ dz, const float* __restrict__ heads,
                                                     const float* __restrict__ eps, int B, int L, float lambda_kl,
                                                     float* __restrict__ dheads) {
    const int n = B * L;
    const float k = lambda_kl / (float)n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / L, j = i - b * L;
        const size_t o = (size_t)b * 2 * L + j;
        const float m = heads[o], l = heads[o + L];
        dheads[o] = dz[i] + k * m;
        dheads[o + L] = dz[i] * eps[i] * 0.5f * expf(0.5f * l) + k * 0.5f * (expf(l) - 1.f);
    }
}

// ---------------------------------------------------------------------------------------------
// MS-SSIM
// ---------------------------------------------------------------------------------------------

// 2x2 average pooling of (BC, H, W) -> (BC, H/2, W/2)  (H, W even)
__global__ void __launch_bounds__(256) k_pool2(const float* __restrict__ x, int BC, int H, int W, float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)BC * Ho * Wo;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int xo = (int)(o % Wo);
        const long long r = o / Wo;
        const int yo = (int)(r % Ho);
        const long long bc = r / Ho;
        const float* p = x + (bc * H + 2 * yo) * W + 2 * xo;
        out[o] = 0.25f * (p[0] + p[1] + p[W] + p[W + 1]);
    }
}

// one 16x16 tile of valid outputs of image bc.  sums[bc*2] += sum ssim, [bc*2+1] += sum cs; the partials of f (cs, or
// ssim when `last`) go to A / Bm / Cm (BC, H-10, W-10).  grid (tiles_x, tiles_y, BC), block 256
__global__ void __launch_bounds__(256) k_ssim_fwd(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                  Gauss gw, float C1, float C2, int last, double* __restrict__ sums,
                                                  float* __restrict__ A, float* __restrict__ Bm, float* __restrict__ Cm) {
    __shared__ float sx[kIn][kIn + 1], sy[kIn][kIn + 1];
    __shared__ float rowf[5][kIn][kTile + 1];
    __shared__ double red[4];
    const int Hv = H - kHalo, Wv = W - kHalo;
    const int bc = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
    const float* xp = X + (size_t)bc * H * W;
    const float* yp = Y + (size_t)bc * H * W;
    for (int e = threadIdx.x; e < kIn * kIn; e += 256) {
        const int r = e / kIn, c = e - r * kIn;
        const int yy = ty0 + r, xx = tx0 + c;
        const bool ok = yy < H && xx < W;
        sx[r][c] = ok ? xp[(size_t)yy * W + xx] : 0.f;
        sy[r][c] = ok ? yp[(size_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    // rows: 26 rows x 16 columns x 5 quantities
    for (int e = threadIdx.x; e < kIn * kTile; e += 256) {
        const int r = e / kTile, c = e - r * kTile;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int t = 0; t < kWin; t++) {
            const float xv = sx[r][c + t], yv = sy[r][c + t], g = gw.g[t];
            a0 = fmaf(g, xv, a0);
            a1 = fmaf(g, yv, a1);
            a2 = fmaf(g, xv * xv, a2);
            a3 = fmaf(g, yv * yv, a3);
            a4 = fmaf(g, xv * yv, a4);
        }
        rowf[0][r][c] = a0, rowf[1][r][c] = a1, rowf[2][r][c] = a2, rowf[3][r][c] = a3, rowf[4][r][c] = a4;
    }
    __syncthreads();
    const int r = threadIdx.x / kTile, c = threadIdx.x - r * kTile;
    float m[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < kWin; t++) {
        const float g = gw.g[t];
#pragma unroll
        for (int k = 0; k < 5; k++) m[k] = fmaf(g, rowf[k][r + t][c], m[k]);
    }
    const int oy = ty0 + r, ox = tx0 + c;
    double s_ssim = 0, s_cs = 0;
    if (oy < Hv && ox < Wv) {
        const float mux = m[0], muy = m[1];
        const float sxx = m[2] - mux * mux, syy = m[3] - muy * muy, sxy = m[4] - mux * muy;
        const float D2 = sxx + syy + C2, D1 = mux * mux + muy * muy + C1;
        const float cs = (2.f * sxy + C2) / D2, l = (2.f * mux * muy + C1) / D1;
        s_ssim = (double)(l * cs);
        s_cs = (double)cs;
        const size_t o = ((size_t)bc * Hv + oy) * Wv + ox;
        const float dcs_dgx = (2.f * mux * cs - 2.f * muy) / D2;       // through s_xx and s_xy
        if (last) {
            const float dl_dmux = 2.f * (muy - l * mux) / D1;
            A[o] = cs * dl_dmux + l * dcs_dgx;
            Bm[o] = -l * cs / D2;
            Cm[o] = l * 2.f / D2;
        } else {
            A[o] = dcs_dgx;
            Bm[o] = -cs / D2;
            Cm[o] = 2.f / D2;
        }
    }
    const double t1 = block_sum(s_ssim, red);
    const double t2 = block_sum(s_cs, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * bc], t1);
        atomicAdd(&sums[2 * bc + 1], t2);
    }
}

// per (b,c): f_s = relu(mean), M = prod f_s^w_s, loss += (1 - M)/BC; kappa[bc][s] = -lambda_ssim/BC * w_s * M / f_s / N_s
// sums: [scale][BC][2]; nvalid[s] = (H_s-10)*(W_s-10).  one block
__global__ void __launch_bounds__(256) k_msssim_finalize(const double* __restrict__ sums, int BC, const int* __restrict__ nvalid,
                                                         float lambda_ssim, double* __restrict__ loss_out, float* __restrict__ kappa) {
    __shared__ double red[4];
    double acc = 0;
    for (int bc = threadIdx.x; bc < BC; bc += 256) {
        double f[kScales], M = 1.0;
        for (int s = 0; s < kScales; s++) {
            const double v = sums[((size_t)s * BC + bc) * 2 + (s == kScales - 1 ? 0 : 1)] / (double)nvalid[s];
            f[s] = v > 0.0 ? v : 0.0;
            M *= pow(f[s], (double)c_ms_weights[s]);
        }
        acc += 1.0 - M;
        for (int s = 0; s < kScales; s++)
            kappa[bc * kScales + s] =
                f[s] > 0.0 ? (float)(-(double)lambda_ssim / BC * c_ms_weights[s] * M / f[s] / (double)nvalid[s]) : 0.f;
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) *loss_out = t / (double)BC;
}

// gradient of the scale's term with respect to its input x: G(q) = kappa * [ (G^T A)(q) + 2 x(q) (G^T Bm)(q) + y(q) (G^T Cm)(q) ]
// plus, when coarse != nullptr, a quarter of the coarser scale's gradient at (q/2) (backward of the 2x2 average pooling).
// grid (tiles_x, tiles_y, BC) over the H x W input, block 256
__global__ void __launch_bounds__(256) k_ssim_bwd(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, Gauss gw,
                                                  const float* __restrict__ A, const float* __restrict__ Bm,
                                                  const float* __restrict__ Cm, const float* __restrict__ kappa, int scale,
                                                  const float* __restrict__ coarse, float* __restrict__ Gout) {
    __shared__ float sm[3][kIn][kIn + 1];
    __shared__ float rowf[3][kIn][kTile + 1];
    const int Hv = H - kHalo, Wv = W - kHalo;
    const int bc = blockIdx.z, ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
    // input pixel q receives from valid outputs p in [q-10, q]: stage maps at rows ty0-10 .. ty0+15
    for (int e = threadIdx.x; e < kIn * kIn; e += 256) {
        const int r = e / kIn, c = e - r * kIn;
        const int py = ty0 - kHalo + r, px = tx0 - kHalo + c;
        const bool ok = py >= 0 && py < Hv && px >= 0 && px < Wv;
        const size_t o = ((size_t)bc * Hv + (ok ? py : 0)) * Wv + (ok ? px : 0);
        sm[0][r][c] = ok ? A[o] : 0.f;
        sm[1][r][c] = ok ? Bm[o] : 0.f;
        sm[2][r][c] = ok ? Cm[o] : 0.f;
    }
    __syncthreads();
    // q = p + t  =>  G^T M (q) = sum_t g[t] M(q - t): staged column index of (qx - t) is c + 10 - t
    for (int e = threadIdx.x; e < kIn * kTile; e += 256) {
        const int r = e / kTile, c = e - r * kTile;
        float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int t = 0; t < kWin; t++) {
            const float g = gw.g[t];
            a0 = fmaf(g, sm[0][r][c + kHalo - t], a0);
            a1 = fmaf(g, sm[1][r][c + kHalo - t], a1);
            a2 = fmaf(g, sm[2][r][c + kHalo - t], a2);
        }
        rowf[0][r][c] = a0, rowf[1][r][c] = a1, rowf[2][r][c] = a2;
    }
    __syncthreads();
    const int r = threadIdx.x / kTile, c = threadIdx.x - r * kTile;
    const int qy = ty0 + r, qx = tx0 + c;
    if (qy >= H || qx >= W) return;
    float m0 = 0, m1 = 0, m2 = 0;
#pragma unroll
    for (int t = 0; t < kWin; t++) {
        const float g = gw.g[t];
        m0 = fmaf(g, rowf[0][r + kHalo - t][c], m0);
        m1 = fmaf(g, rowf[1][r + kHalo - t][c], m1);
        m2 = fmaf(g, rowf[2][r + kHalo - t][c], m2);
    }
    const size_t o = ((size_t)bc * H + qy) * W + qx;
    float gq = kappa[bc * kScales + scale] * (m0 + 2.f * X[o] * m1 + Y[o] * m2);
    if (coarse) gq += 0.25f * coarse[((size_t)bc * (H / 2) + qy / 2) * (W / 2) + qx / 2];
    Gout[o] = gq;
}

// y = sigmoid(u); yb = y, tb = gathered target
__global__ void __launch_bounds__(256) k_sigmoid_gather(const float* __restrict__ u, const float* __restrict__ target,
                                                        const int* __restrict__ perm, long long start, int B, long long E,
                                                        float* __restrict__ y, float* __restrict__ tb) {
    const long long total = (long long)B * E;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        y[o] = 1.f / (1.f + expf(-u[o]));
        if (tb) {
            const long long b = o / E, i = o - b * E;
            const long long s = perm ? (long long)perm[start + b] : start + b;
            tb[o] = target[s * E + i];
        }
    }
}

// du = scale * (lambda_mse * 2 (y - t) / n + gssim) * y (1 - y);  mse_out += sum (y-t)^2 / n
// (scale: the data-parallel weight local / global batch of this rank's gradient, 1 on a single device)
__global__ void __launch_bounds__(256) k_vae_loss_grad(const float* __restrict__ y, const float* __restrict__ t,
                                                       const float* __restrict__ gssim, long long n, float lambda_mse, float scale,
                                                       float* __restrict__ du, double* __restrict__ mse_out) {
    __shared__ double red[4];
    double s = 0;
    const float k = 2.f * lambda_mse / (float)n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float yv = y[i], d = yv - t[i];
        s += (double)d * (double)d;
        if (du) du[i] = scale * ((k * d + (gssim ? gssim[i] : 0.f)) * yv * (1.f - yv));
    }
    const double tt = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(mse_out, tt / (double)n);
}

// g1 = (g1 + g2) * [h > 0]   (two heads read the same ReLU output)
__global__ void __launch_bounds__(256) k_add_relu_bwd(float* __restrict__ g1, const float* __restrict__ g2,
                                                      const float* __restrict__ h, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        g1[i] = h[i] > 0.f ? g1[i] + g2[i] : 0.f;
}

// parts = {mse, kl sum, 1 - ms_ssim}: slot = {mse, kl mean, ssim loss, weighted total}
__global__ void k_loss_slot(const double* __restrict__ parts, double kl_count, double l_mse, double l_kl, double l_ssim,
                            double* __restrict__ slot) {
    const double mse = parts[0], kl = parts[1] / kl_count, sl = parts[2];
    slot[0] = mse, slot[1] = kl, slot[2] = sl, slot[3] = l_mse * mse + l_kl * kl + l_ssim * sl;
}

// Adam with L2 weight decay (torch.optim.Adam single-tensor formula): consumes and clears the fp64 accumulator
struct AdamHyper {
    double lr, beta1, beta2, eps, wd;
};
__global__ void __launch_bounds__(256) k_adam_l2(long long n, float* __restrict__ p, double* __restrict__ gacc,
                                                 float* __restrict__ m, float* __restrict__ v, AdamHyper h, int step) {
    __shared__ float corr[2];   // one lane per workgroup pays for the two fp64 pow() of the bias corrections
    if (threadIdx.x == 0) {
        const double bc1 = 1.0 - pow(h.beta1, (double)step), bc2 = 1.0 - pow(h.beta2, (double)step);
        corr[0] = (float)(h.lr / bc1);
        corr[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const float step_size = corr[0], bc2_sqrt = corr[1];
    const float b1 = (float)h.beta1, b2 = (float)h.beta2, eps = (float)h.eps, wd = (float)h.wd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float w = p[i];
        const float g = fmaf(wd, w, (float)gacc[i]);      // grad = grad.add(param, alpha=weight_decay)
        gacc[i] = 0.0;
        const float mi = m[i] + (g - m[i]) * (1.f - b1);
        const float vi = fmaf(g * g, 1.f - b2, v[i] * b2);
        p[i] = w - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace
}  // namespace vae
