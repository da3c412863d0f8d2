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

#include <utility>

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

// ---- row-streaming forms of the two MS-SSIM passes (round 3) --------------------------------------------------------------
// The tile kernels above stage a 26x26 halo in LDS for 16x16 outputs and read it ~90 times per output pixel through the LDS
// pipe: 278 us for the first scale at the benchmark size, 60 % of what was left of the 'var' step once its trunk ran on the
// ConvAE kernels.  Both filters are separable, so here a WAVE owns a strip of 64 input columns (54 outputs) and walks down the
// rows of a band: a lane loads its column's pixel of the row (one coalesced 256-byte run per map), the 11-tap horizontal
// filter takes its neighbours' values by ten one-lane DPP shifts (no LDS), and the vertical filter reads the last eleven
// rows' horizontal results from a register ring whose indices are compile-time constants (the row loop is unrolled by
// eleven).  ~215 vector instructions per row of 54 outputs, no barrier, no LDS.  The arithmetic - every product, every fma and
// their order - is the tile kernels' (but for two reciprocals in place of six divisions per pixel); the results agree to fp32
// rounding (tests/test_vae_hip_parity.py runs both).
constexpr int kSsimCols = 64 - kHalo;   // output columns of a wave
// output rows of a wave (its band) by the map's height: a band of R rows filters R + 10 (tall bands repeat less work, short
// ones give more waves and a shorter serial walk; measured at the benchmark size, first scale: forward 80 us with 32 rows,
// 110 with 16; the small scales are bound by the length of the walk, not by its arithmetic)
inline int ssim_band_rows(int rows, bool forward) { return rows >= 400 ? (forward ? 32 : 16) : (rows >= 200 ? 16 : 8); }
constexpr int kSsimAhead = 5;           // rows requested ahead of the one being filtered (a wave has one or two peers on its
                                        // SIMD: with a single row in flight a row cost a memory round trip, 0.66 us)

template <class F, int... I>
__device__ __forceinline__ void ssim_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ssim_static_for(F&& f) {
    ssim_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// lane i <- lane i+1 (lane 63: 0) / lane i <- lane i-1 (lane 0: 0)
__device__ __forceinline__ float ssim_from_right(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ float ssim_from_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ double ssim_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one scale's forward walk of image bc: wave w of the workgroup takes band blockIdx.y * 4 + w of strip blockIdx.x
__device__ __forceinline__ void ssim_fwd_rows_body(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, int rb,
                                                   const Gauss& gw, float C1, float C2, int last, double* __restrict__ sums,
                                                   float* __restrict__ A, float* __restrict__ Bm, float* __restrict__ Cm, int bc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Hv = H - kHalo, Wv = W - kHalo;
    const int x0 = blockIdx.x * kSsimCols, y0 = (blockIdx.y * 4 + wv) * rb;
    if (y0 >= Hv || x0 >= Wv) return;   // (the kernel has no barrier)
    const int n_out = min(rb, Hv - y0), n_in = n_out + kHalo;
    const bool in_ok = x0 + lane < W;
    const bool out_ok = lane < kSsimCols && x0 + lane < Wv;
    const size_t in0 = ((size_t)bc * H + y0) * W + (in_ok ? x0 + lane : 0);
    const float* xp = X + in0;
    const float* yp = Y + in0;
    const size_t out0 = ((size_t)bc * Hv + y0) * Wv + (out_ok ? x0 + lane : 0);
    float ring[kWin][5];
    float xq[kWin], yq[kWin];              // rows in flight: row r lives in slot r % 11
    double acc_ssim = 0.0, acc_cs = 0.0;   // (as the tile kernel: every pixel's term enters an fp64 sum)
#pragma unroll
    for (int d = 0; d < kSsimAhead; d++) {
        const bool ok = in_ok && d < n_in;
        xq[d] = ok ? xp[(size_t)d * W] : 0.f;
        yq[d] = ok ? yp[(size_t)d * W] : 0.f;
    }
    for (int base = 0; base < n_in; base += kWin) {
        ssim_static_for<kWin>([&](auto S) {
            constexpr int s = decltype(S)::value;
            const int r = base + s;            // input row y0 + r; r % 11 == s
            if (r < n_in) {
                float xs = xq[s], ys = yq[s];
                {
                    const bool ok = in_ok && r + kSsimAhead < n_in;
                    xq[(s + kSsimAhead) % kWin] = ok ? xp[(size_t)(r + kSsimAhead) * W] : 0.f;
                    yq[(s + kSsimAhead) % kWin] = ok ? yp[(size_t)(r + kSsimAhead) * W] : 0.f;
                }
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
                for (int t = 0; t < kWin; t++) {
                    const float g = gw.g[t];
                    a0 = fmaf(g, xs, a0);
                    a1 = fmaf(g, ys, a1);
                    a2 = fmaf(g, xs * xs, a2);
                    a3 = fmaf(g, ys * ys, a3);
                    a4 = fmaf(g, xs * ys, a4);
                    if (t + 1 < kWin) {
                        xs = ssim_from_right(xs);
                        ys = ssim_from_right(ys);
                    }
                }
                ring[s][0] = a0, ring[s][1] = a1, ring[s][2] = a2, ring[s][3] = a3, ring[s][4] = a4;
                if (r >= kHalo) {              // output row y0 + r - 10: input rows r-10 .. r = ring[(s + 1 + t) % 11]
                    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < kWin; t++) {
                        const float g = gw.g[t];
#pragma unroll
                        for (int k = 0; k < 5; k++) m[k] = fmaf(g, ring[(s + 1 + t) % kWin][k], m[k]);
                    }
                    const float mux = m[0], muy = m[1];
                    const float sxx = m[2] - mux * mux, syy = m[3] - muy * muy, sxy = m[4] - mux * muy;
                    const float D2 = sxx + syy + C2, D1 = mux * mux + muy * muy + C1;
                    // two divisions per pixel instead of the tile kernel's six (a correctly rounded fp32 division is ~10
                    // instructions of this VALU-bound walk): the quotients by D2 and D1 become products with 1 / D2, 1 / D1
                    const float rD2 = 1.f / D2, rD1 = 1.f / D1;
                    const float cs = (2.f * sxy + C2) * rD2, l = (2.f * mux * muy + C1) * rD1;
                    if (out_ok) {
                        acc_ssim += (double)(l * cs);
                        acc_cs += (double)cs;
                        const size_t o = out0 + (size_t)(r - kHalo) * Wv;
                        const float dcs_dgx = (2.f * mux * cs - 2.f * muy) * rD2;       // through s_xx and s_xy
                        if (last) {
                            const float dl_dmux = 2.f * (muy - l * mux) * rD1;
                            A[o] = cs * dl_dmux + l * dcs_dgx;
                            Bm[o] = -l * cs * rD2;
                            Cm[o] = l * 2.f * rD2;
                        } else {
                            A[o] = dcs_dgx;
                            Bm[o] = -cs * rD2;
                            Cm[o] = 2.f * rD2;
                        }
                    }
                }
            }
        });
    }
    const double t1 = ssim_wave_sum(acc_ssim), t2 = ssim_wave_sum(acc_cs);
    if (lane == 0) {
        atomicAdd(&sums[2 * bc], t1);
        atomicAdd(&sums[2 * bc + 1], t2);
    }
}

// grid (ceil(Wv / 54), ceil(bands / 4), BC), block 256
__global__ void __launch_bounds__(256) k_ssim_fwd_rows(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                       int rb, Gauss gw, float C1, float C2, int last, double* __restrict__ sums,
                                                       float* __restrict__ A, float* __restrict__ Bm, float* __restrict__ Cm) {
    ssim_fwd_rows_body(X, Y, H, W, rb, gw, C1, C2, last, sums, A, Bm, Cm, blockIdx.z);
}

// The coarse scales in ONE launch: they are independent of one another once the pooled maps exist, and each is bound by the
// length of its serial walk (10-16 us per launch for 16 images of 256 x 256 and smaller), not by its arithmetic.
// grid (strips of the largest scale, band groups of the largest scale, BC * n): blockIdx.z = scale slot * BC + bc; workgroups
// outside a smaller scale's extent leave at once.
struct SsimScale {
    const float* X;
    const float* Y;
    int H, W, rb, last;
    double* sums;
    float *A, *Bm, *Cm;
};
struct SsimScales {
    SsimScale s[kScales];
    int n, BC;
};
__global__ void __launch_bounds__(256) k_ssim_fwd_rows_multi(SsimScales set, Gauss gw, float C1, float C2) {
    const int slot = blockIdx.z / set.BC, bc = blockIdx.z - slot * set.BC;
    const SsimScale& p = set.s[slot];
    ssim_fwd_rows_body(p.X, p.Y, p.H, p.W, p.rb, gw, C1, C2, p.last, p.sums, p.A, p.Bm, p.Cm, bc);
}

// The whole pooling pyramid of both maps in one launch: a workgroup takes a 16 x 16 block of the finest map and writes its
// 8 x 8, 4 x 4, 2 x 2 and 1 x 1 averages (each level the 2 x 2 average of the level above, as k_pool2 computes them one launch
// per level).  H, W multiples of 16.  grid (W / 16, H / 16, BC), block 256
struct Pyramid {
    float* x[kScales];   // [1..4] written; [0] read, or written too when the finest level is formed here (below)
    float* y[kScales];
    // finest level formed in this pass: x[0] = sigmoid(u), y[0] = the target rows of samples perm[start + b] (what
    // k_sigmoid_gather does in a launch of its own); u == nullptr: x[0], y[0] are inputs.  E = C * H * W elements per sample.
    const float* u;
    const float* target;
    const int* perm;
    long long start;
    int C;
};
__global__ void __launch_bounds__(256) k_pool_pyramid(Pyramid pm, int H, int W) {
    __shared__ float sx[2][16][17], sy[2][16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int bc = blockIdx.z;
    const size_t in_img = (size_t)(blockIdx.y * 16 + ty) * W + blockIdx.x * 16 + tx;
    const size_t o0 = (size_t)bc * H * W + in_img;
    if (pm.u) {
        const int b = bc / pm.C, c = bc - b * pm.C;
        const long long smp = pm.perm ? (long long)pm.perm[pm.start + b] : pm.start + b;
        const float yv = 1.f / (1.f + expf(-pm.u[o0]));
        const float tv = pm.target[((size_t)smp * pm.C + c) * H * W + in_img];
        pm.x[0][o0] = yv;
        pm.y[0][o0] = tv;
        sx[0][ty][tx] = yv;
        sy[0][ty][tx] = tv;
    } else {
        sx[0][ty][tx] = pm.x[0][o0];
        sy[0][ty][tx] = pm.y[0][o0];
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int lvl = 1; lvl < kScales; lvl++) {
        const int n = 16 >> lvl;                       // side of this level's block
        if (tx < n && ty < n) {
            const float vx = 0.25f * (sx[cur][2 * ty][2 * tx] + sx[cur][2 * ty][2 * tx + 1] + sx[cur][2 * ty + 1][2 * tx] + sx[cur][2 * ty + 1][2 * tx + 1]);
            const float vy = 0.25f * (sy[cur][2 * ty][2 * tx] + sy[cur][2 * ty][2 * tx + 1] + sy[cur][2 * ty + 1][2 * tx] + sy[cur][2 * ty + 1][2 * tx + 1]);
            sx[cur ^ 1][ty][tx] = vx;
            sy[cur ^ 1][ty][tx] = vy;
            const int Hl = H >> lvl, Wl = W >> lvl;
            const size_t o = ((size_t)bc * Hl + blockIdx.y * n + ty) * Wl + blockIdx.x * n + tx;
            pm.x[lvl][o] = vx;
            pm.y[lvl][o] = vy;
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Backward of one scale, row-streaming: G(q) = kappa [ (G^T A)(q) + 2 x(q) (G^T Bm)(q) + y(q) (G^T Cm)(q) ] (+ coarse / 4).
// (G^T M)(q) = sum_t g[t] M(q - t) in both directions: lane l holds map column x0 + l (x0 = 54 * strip - 10; zero outside the
// valid region), takes its LEFT neighbours by DPP and owns output column x0 + l when l >= 10; rows likewise from a ring.
// The epilogue's operands (x, y, the coarser scale's gradient) are requested as far ahead as the maps' rows.
__device__ __forceinline__ void ssim_bwd_rows_body(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, int rb,
                                                   const Gauss& gw, const float* __restrict__ A, const float* __restrict__ Bm,
                                                   const float* __restrict__ Cm, const float* __restrict__ kappa, int scale,
                                                   const float* __restrict__ coarse, float* __restrict__ Gout, int bc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Hv = H - kHalo, Wv = W - kHalo;
    const int x0 = blockIdx.x * kSsimCols - kHalo, y0 = (blockIdx.y * 4 + wv) * rb;   // first map column / first output row
    if (y0 >= H || x0 + kHalo >= W) return;
    const int n_out = min(rb, H - y0), n_in = n_out + kHalo;    // map rows y0 - 10 .. y0 + n_out - 1
    const int px = x0 + lane;
    const bool in_col = px >= 0 && px < Wv;
    const bool out_ok = lane >= kHalo && px < W;
    const float kap = kappa[bc * kScales + scale];
    const size_t mbase = (size_t)bc * Hv * Wv + (in_col ? px : 0);
    const size_t obase = (size_t)bc * H * W + (out_ok ? px : 0);
    const size_t cbase = (size_t)bc * (H / 2) * (W / 2) + (out_ok ? px / 2 : 0);
    float ring[kWin][3];
    float aq[kWin], bq[kWin], cq[kWin];    // map rows in flight (row r in slot r % 11)
    float xe[kWin], ye[kWin], ce[kWin];    // epilogue operands of the output row emitted at step r
    auto fetch = [&](int r, int slot) {    // map row y0 - 10 + r and, once r >= 10, output row y0 + r - 10
        const int py = y0 - kHalo + r;
        const bool ok = r < n_in && in_col && py >= 0 && py < Hv;
        const size_t o = mbase + (size_t)(ok ? py : 0) * Wv;
        aq[slot] = ok ? A[o] : 0.f;
        bq[slot] = ok ? Bm[o] : 0.f;
        cq[slot] = ok ? Cm[o] : 0.f;
        const bool eo = r < n_in && r >= kHalo && out_ok;
        const size_t q = obase + (size_t)(eo ? py : 0) * W;
        xe[slot] = eo ? X[q] : 0.f;
        ye[slot] = eo ? Y[q] : 0.f;
        ce[slot] = eo && coarse ? coarse[cbase + (size_t)(py / 2) * (W / 2)] : 0.f;
    };
#pragma unroll
    for (int d = 0; d < kSsimAhead; d++) fetch(d, d);
    for (int base = 0; base < n_in; base += kWin) {
        ssim_static_for<kWin>([&](auto S) {
            constexpr int s = decltype(S)::value;
            const int r = base + s;
            if (r < n_in) {
                float as = aq[s], bs = bq[s], cs = cq[s];
                const float xv = xe[s], yv = ye[s], cv = ce[s];
                fetch(r + kSsimAhead, (s + kSsimAhead) % kWin);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int t = 0; t < kWin; t++) {
                    const float g = gw.g[t];
                    a0 = fmaf(g, as, a0);
                    a1 = fmaf(g, bs, a1);
                    a2 = fmaf(g, cs, a2);
                    if (t + 1 < kWin) {
                        as = ssim_from_left(as);
                        bs = ssim_from_left(bs);
                        cs = ssim_from_left(cs);
                    }
                }
                ring[s][0] = a0, ring[s][1] = a1, ring[s][2] = a2;
                if (r >= kHalo) {              // output row qy = y0 + r - 10 = the newest map row: rows qy - t = ring[(s - t) mod 11]
                    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int t = 0; t < kWin; t++) {
                        const float g = gw.g[t];
                        m0 = fmaf(g, ring[(s + kWin - t) % kWin][0], m0);
                        m1 = fmaf(g, ring[(s + kWin - t) % kWin][1], m1);
                        m2 = fmaf(g, ring[(s + kWin - t) % kWin][2], m2);
                    }
                    if (out_ok) {
                        float gq = kap * (m0 + 2.f * xv * m1 + yv * m2);
                        if (coarse) gq += 0.25f * cv;
                        Gout[obase + (size_t)(y0 + r - kHalo) * W] = gq;
                    }
                }
            }
        });
    }
}

// grid (ceil(W / 54), ceil(bands / 4), BC), block 256
__global__ void __launch_bounds__(256) k_ssim_bwd_rows(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                       int rb, Gauss gw, const float* __restrict__ A, const float* __restrict__ Bm,
                                                       const float* __restrict__ Cm, const float* __restrict__ kappa, int scale,
                                                       const float* __restrict__ coarse, float* __restrict__ Gout) {
    ssim_bwd_rows_body(X, Y, H, W, rb, gw, A, Bm, Cm, kappa, scale, coarse, Gout, blockIdx.z);
}

// The coarse scales' OWN terms kappa_s (G^T A + 2 x G^T Bm + y G^T Cm) in one launch (they do not depend on one another; what
// chains the scales is only the pooling term, added by k_ssim_combine).  grid as k_ssim_fwd_rows_multi.
struct SsimBwdScale {
    const float* X;
    const float* Y;
    int H, W, rb, scale;
    const float *A, *Bm, *Cm;
    float* G;
};
struct SsimBwdScales {
    SsimBwdScale s[kScales];
    int n, BC;
};
__global__ void __launch_bounds__(256) k_ssim_bwd_rows_multi(SsimBwdScales set, Gauss gw, const float* __restrict__ kappa) {
    const int slot = blockIdx.z / set.BC, bc = blockIdx.z - slot * set.BC;
    const SsimBwdScale& p = set.s[slot];
    ssim_bwd_rows_body(p.X, p.Y, p.H, p.W, p.rb, gw, p.A, p.Bm, p.Cm, kappa, p.scale, nullptr, p.G, bc);
}

// G1(q) = own1(q) + 1/4 (own2(q/2) + 1/4 (own3(q/4) + 1/4 own4(q/8))): the pooling chain of scales 4 -> 1, nested exactly as the
// per-scale launches added it (G_s = own_s + G_{s+1} / 4), in place on own1.  (BC, H1, W1) with H1, W1 multiples of 8.
__global__ void __launch_bounds__(256) k_ssim_combine(float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3,
                                                      const float* __restrict__ g4, int BC, int H1, int W1) {
    const long long total = (long long)BC * H1 * W1;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int x = (int)(o % W1);
        const long long r = o / W1;
        const int y = (int)(r % H1);
        const long long bc = r / H1;
        float g3v = g3[(bc * (H1 / 4) + y / 4) * (W1 / 4) + x / 4];
        g3v += 0.25f * g4[(bc * (H1 / 8) + y / 8) * (W1 / 8) + x / 8];
        float g2v = g2[(bc * (H1 / 2) + y / 2) * (W1 / 2) + x / 2];
        g2v += 0.25f * g3v;
        g1[o] += 0.25f * g2v;
    }
}

// per (b,c): f_s = relu(mean), M = prod f_s^w_s, loss += (1 - M)/BC; kappa[bc][s] = -lambda_ssim/BC * w_s * M / f_s / N_s
// sums: [scale][BC][2]; nvalid.v[s] = (H_s-10)*(W_s-10), a kernel argument (no per-step host-to-device copy).  one block
struct ScaleCounts {
    int v[kScales];
};
__global__ void __launch_bounds__(256) k_msssim_finalize(const double* __restrict__ sums, int BC, ScaleCounts nvalid,
                                                         float lambda_ssim, double* __restrict__ loss_out, float* __restrict__ kappa) {
    __shared__ double red[4];
    double acc = 0;
    for (int bc = threadIdx.x; bc < BC; bc += 256) {
        double f[kScales], M = 1.0;
        for (int s = 0; s < kScales; s++) {
            const double v = sums[((size_t)s * BC + bc) * 2 + (s == kScales - 1 ? 0 : 1)] / (double)nvalid.v[s];
            f[s] = v > 0.0 ? v : 0.0;
            M *= pow(f[s], (double)c_ms_weights[s]);
        }
        acc += 1.0 - M;
        for (int s = 0; s < kScales; s++)
            kappa[bc * kScales + s] =
                f[s] > 0.0 ? (float)(-(double)lambda_ssim / BC * c_ms_weights[s] * M / f[s] / (double)nvalid.v[s]) : 0.f;
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
// du_sum (single-channel outputs): += sum du = the last layer's bias gradient, in the pass that writes du
__global__ void __launch_bounds__(256) k_vae_loss_grad(const float* __restrict__ y, const float* __restrict__ t,
                                                       const float* __restrict__ gssim, long long n, float lambda_mse, float scale,
                                                       float* __restrict__ du, double* __restrict__ mse_out,
                                                       double* __restrict__ du_sum) {
    __shared__ double red[4];
    double s = 0, sb = 0;
    const float k = 2.f * lambda_mse / (float)n;
    auto one = [&](float yv, float tv, float gs) {
        const float d = yv - tv;
        s += (double)d * (double)d;
        const float g = scale * ((k * d + gs) * yv * (1.f - yv));
        sb += (double)g;
        return g;
    };
    if ((n & 3) == 0) {      // 16-byte accesses (the maps start on 256-byte boundaries)
        for (long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x); i < n; i += 4 * (long long)gridDim.x * 256) {
            const float4 yv = *reinterpret_cast<const float4*>(y + i), tv = *reinterpret_cast<const float4*>(t + i);
            const float4 gs = (du && gssim) ? *reinterpret_cast<const float4*>(gssim + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 g = make_float4(one(yv.x, tv.x, gs.x), one(yv.y, tv.y, gs.y), one(yv.z, tv.z, gs.z), one(yv.w, tv.w, gs.w));
            if (du) *reinterpret_cast<float4*>(du + i) = g;
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const float g = one(y[i], t[i], (du && gssim) ? gssim[i] : 0.f);
            if (du) du[i] = g;
        }
    }
    const double tt = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(mse_out, tt / (double)n);
    if (du && du_sum) {
        __syncthreads();
        const double tb = block_sum(sb, red);
        if (threadIdx.x == 0) atomicAdd(du_sum, tb);
    }
}

// the 2x2 average pooling of both maps of a scale in one launch
__global__ void __launch_bounds__(256) k_pool2_pair(const float* __restrict__ x, const float* __restrict__ y, int BC, int H, int W,
                                                    float* __restrict__ xo, float* __restrict__ yo) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)BC * Ho * Wo;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int xq = (int)(o % Wo);
        const long long r = o / Wo;
        const int yq = (int)(r % Ho);
        const long long bc = r / Ho;
        const size_t i = (size_t)(bc * H + 2 * yq) * W + 2 * xq;
        const float2 x0 = *reinterpret_cast<const float2*>(x + i), x1 = *reinterpret_cast<const float2*>(x + i + W);
        const float2 y0 = *reinterpret_cast<const float2*>(y + i), y1 = *reinterpret_cast<const float2*>(y + i + W);
        xo[o] = 0.25f * (x0.x + x0.y + x1.x + x1.y);
        yo[o] = 0.25f * (y0.x + y0.y + y1.x + y1.y);
    }
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
