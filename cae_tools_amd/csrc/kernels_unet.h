// kernels_unet.h — gfx950 kernels of the UNET path (reference: src/cae_tools/models/unet.py).
//
// Unlike the ConvAE path (3.8 FLOP/B, everything fused into a handful of streaming kernels) the UNET
// layers are wide (32..256 channels, 4x4 kernels): the convolutions are GEMMs that dominate the step,
// and the element-wise work between them (BatchNorm, ReLU, dropout, channel attention, skip concat)
// is a few passes over tensors the GEMMs have to read anyway.  So the convolutions here are PURE
// (plain tensors in, plain tensors out, bias in the epilogue) and the glue is a set of small
// element-wise / reduction kernels.
//
// This file holds the shape-generic kernels (any kernel size, stride, padding; one thread per output)
// and all of the glue; kernels_unet_mfma.h holds the LDS-tiled MFMA kernels for the 4x4 / stride 2 /
// pad 1 layers that carry the FLOPs.  Both are checked against the CPU oracle.
//
// Conv primitives, on a small map S (B,Cs,Hs,Ws), a big map L (B,Cl,Hl,Wl), weights w[cs][cl][ky][kx]
// (PyTorch's layout for Conv2d (Cout,Cin,kh,kw) with S = output and for ConvTranspose2d
// (Cin,Cout,kh,kw) with S = input), stride s and padding p:
//   down : S[cs][y][x]        = bias + sum_cl,ky,kx L[cl][y*s+ky-p][x*s+kx-p] * w     (Conv2d fwd, ConvT dgrad)
//   up   : L[cl][Y][X]        = bias + sum_cs,ky,kx S[cs][(Y+p-ky)/s][(X+p-kx)/s] * w (ConvT fwd, Conv2d dgrad)
//   wgrad: dw[cs][cl][ky][kx] = sum_b,y,x S[cs][y][x] * L[cl][y*s+ky-p][x*s+kx-p]
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

namespace unet {
namespace {   // internal linkage: this header is compiled into more than one translation unit

// sum of v over the block (any multiple of 64 threads), valid in thread 0.  red: LDS scratch of blockDim/64 doubles.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_d(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double block_sum(double v, double* red) {
    // fp64 wave sum on the VALU (DPP moves of both halves; the ds_bpermute butterfly costs ~100 cycles a step): lane 63
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    v += dpp_d<0x142, 0xA>(v);
    v += dpp_d<0x143, 0xC>(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) red[wv] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); i++) t += red[i];
    return t;
}

struct Geom {
    int B, Cs, Hs, Ws, Cl, Hl, Wl, kh, kw, s, p;
};

// ---------------------------------------------------------------------------------------------
// dropout: keep(idx) is a pure function of (key, idx); key = pcg(pcg(seed + 0x9E3779B9*site) ^ step)
// is formed on the host.  oracle/unet_oracle.py:dropout_keep is the same function in numpy.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t pcg(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

struct Drop {
    uint32_t key;
    uint32_t thr;   // keep iff hash >= thr
    float scale;    // 1 / (1 - p)
    int on;
};

__device__ __forceinline__ float drop_factor(const Drop& d, unsigned long long idx) {
    if (!d.on) return 1.f;
    const uint32_t hi = pcg(d.key + (uint32_t)(idx >> 32));
    const uint32_t r = pcg((uint32_t)idx ^ hi);
    return r >= d.thr ? d.scale : 0.f;
}


// (b, c, y, x) of flat index idx in a (B, C, H, W) tensor.  Four 64-bit div/mod pairs cost ~400 instructions per output
// element; three fp32 products do the same while idx < 2^22 * ... each quotient is exact (see plane_loop); otherwise 32- or
// 64-bit integer division.
struct Nchw {
    int b, c, y, x;
};
__device__ __forceinline__ Nchw nchw_of(long long idx, int C, int H, int W) {
    Nchw r;
    if (idx < (1LL << 31)) {
        const unsigned i = (unsigned)idx, w = (unsigned)W, h = (unsigned)H, c = (unsigned)C;
        const unsigned q1 = i / w;
        r.x = (int)(i - q1 * w);
        const unsigned q2 = q1 / h;
        r.y = (int)(q1 - q2 * h);
        const unsigned q3 = q2 / c;
        r.c = (int)(q2 - q3 * c);
        r.b = (int)q3;
    } else {
        r.x = (int)(idx % W);
        long long t = idx / W;
        r.y = (int)(t % H);
        t /= H;
        r.c = (int)(t % C);
        r.b = (int)(t / C);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// generic convolutions (fallback for any geometry)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_down(Geom g, const float* __restrict__ L, const float* __restrict__ w,
                                              const float* __restrict__ bias, float* __restrict__ S) {
    const long long total = (long long)g.B * g.Cs * g.Hs * g.Ws;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const Nchw o = nchw_of(idx, g.Cs, g.Hs, g.Ws);
        const int x = o.x, y = o.y, cs = o.c, b = o.b;
        float acc = bias ? bias[cs] : 0.f;
        const float* wp = w + (size_t)cs * g.Cl * g.kh * g.kw;
        for (int cl = 0; cl < g.Cl; cl++) {
            const float* lp = L + ((size_t)b * g.Cl + cl) * g.Hl * g.Wl;
            for (int ky = 0; ky < g.kh; ky++) {
                const int Y = y * g.s + ky - g.p;
                if (Y < 0 || Y >= g.Hl) continue;
                for (int kx = 0; kx < g.kw; kx++) {
                    const int X = x * g.s + kx - g.p;
                    if (X < 0 || X >= g.Wl) continue;
                    acc = fmaf(lp[(size_t)Y * g.Wl + X], wp[(cl * g.kh + ky) * g.kw + kx], acc);
                }
            }
        }
        S[idx] = acc;
    }
}

__global__ void __launch_bounds__(256) k_up(Geom g, const float* __restrict__ S, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ L) {
    const long long total = (long long)g.B * g.Cl * g.Hl * g.Wl;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const Nchw o = nchw_of(idx, g.Cl, g.Hl, g.Wl);
        const int X = o.x, Y = o.y, cl = o.c, b = o.b;
        float acc = bias ? bias[cl] : 0.f;
        // only the taps of this output's parity: ky = (Y + p) mod s, + s, ... reading row y = (Y + p) / s, - 1, ... (one
        // division per dimension per output instead of a modulo and a division per tap)
        const int y0 = (Y + g.p) / g.s, x0 = (X + g.p) / g.s;
        const int ky0 = Y + g.p - y0 * g.s, kx0 = X + g.p - x0 * g.s;
        for (int ky = ky0, y = y0; ky < g.kh && y >= 0; ky += g.s, y--) {
            if (y >= g.Hs) continue;
            for (int kx = kx0, x = x0; kx < g.kw && x >= 0; kx += g.s, x--) {
                if (x >= g.Ws) continue;
                const float* sp = S + (size_t)b * g.Cs * g.Hs * g.Ws + (size_t)y * g.Ws + x;
                const float* wp = w + ((size_t)cl * g.kh + ky) * g.kw + kx;
                for (int cs = 0; cs < g.Cs; cs++)
                    acc = fmaf(sp[(size_t)cs * g.Hs * g.Ws], wp[(size_t)cs * g.Cl * g.kh * g.kw], acc);
            }
        }
        L[idx] = acc;
    }
}

// grid (Cs*Cl*kh*kw, nsplit): block = one weight, a slice of the batch; fp64 atomics into acc
__global__ void __launch_bounds__(256) k_wgrad(Geom g, const float* __restrict__ S, const float* __restrict__ L,
                                               double* __restrict__ acc) {
    __shared__ double red[4];
    int wi = blockIdx.x;
    const int kx = wi % g.kw;
    wi /= g.kw;
    const int ky = wi % g.kh;
    wi /= g.kh;
    const int cl = wi % g.Cl;
    const int cs = wi / g.Cl;
    const int per = g.Hs * g.Ws;
    const long long total = (long long)g.B * per;
    double sum = 0.0;
    const bool fast = total < (1LL << 22) && per < 8000;   // exact fp32-product quotients (see plane_loop)
    const float inv_per = 1.0f / (float)per, inv_w = 1.0f / (float)g.Ws;
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long long)gridDim.y * 256) {
        int b, y;
        if (fast) {
            b = (int)(((float)(int)e + 0.5f) * inv_per);
            y = (int)(((float)((int)e - b * per) + 0.5f) * inv_w);
        } else {
            b = (int)(e / per);
            y = (int)(e - (long long)b * per) / g.Ws;
        }
        const int r = (int)(e - (long long)b * per);
        const int x = r - y * g.Ws;
        const int Y = y * g.s + ky - g.p, X = x * g.s + kx - g.p;
        if (Y < 0 || Y >= g.Hl || X < 0 || X >= g.Wl) continue;
        sum += (double)(S[((size_t)b * g.Cs + cs) * per + r] * L[(((size_t)b * g.Cl + cl) * g.Hl + Y) * g.Wl + X]);
    }
    const double t = block_sum(sum, red);
    if (threadIdx.x == 0 && t != 0.0) atomicAdd(&acc[blockIdx.x], t);
}


// ---------------------------------------------------------------------------------------------
// 4x4 / stride 2 / pad 1 transposed convolution with a handful of output channels (the last decoder layer: Cl = 3).
// A GEMM tile would be 3 rows tall; this is a streaming problem instead: one thread per input position (b,q,r) forms
// the 2x2 output quad of every output channel from the 3x3 neighbourhood of each input channel; the weights of the
// channel in flight are wave-uniform (scalar loads, SGPR operands).  grid-stride over B*Hs*Ws
// ---------------------------------------------------------------------------------------------
template <int CL>
__global__ void __launch_bounds__(256) k_up_thin(Geom g, const float* __restrict__ S, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ L) {
    const int HW = g.Hs * g.Ws;
    const long long total = (long long)g.B * HW;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / HW), p = (int)(idx - (long long)b * HW);
        const int q = p / g.Ws, r = p - q * g.Ws;
        float acc[CL][2][2];
#pragma unroll
        for (int cl = 0; cl < CL; cl++) {
            const float bv = bias ? bias[cl] : 0.f;
            acc[cl][0][0] = acc[cl][0][1] = acc[cl][1][0] = acc[cl][1][1] = bv;
        }
        const bool up = q > 0, dn = q + 1 < g.Hs, lf = r > 0, rt = r + 1 < g.Ws;
        const float* sp = S + (size_t)b * g.Cs * HW + p;
        for (int cs = 0; cs < g.Cs; cs++, sp += HW) {
            float v[3][3];
            v[0][0] = up && lf ? sp[-g.Ws - 1] : 0.f;
            v[0][1] = up ? sp[-g.Ws] : 0.f;
            v[0][2] = up && rt ? sp[-g.Ws + 1] : 0.f;
            v[1][0] = lf ? sp[-1] : 0.f;
            v[1][1] = sp[0];
            v[1][2] = rt ? sp[1] : 0.f;
            v[2][0] = dn && lf ? sp[g.Ws - 1] : 0.f;
            v[2][1] = dn ? sp[g.Ws] : 0.f;
            v[2][2] = dn && rt ? sp[g.Ws + 1] : 0.f;
            const float* wp = w + (size_t)cs * CL * 16;
#pragma unroll
            for (int cl = 0; cl < CL; cl++)
#pragma unroll
                for (int py = 0; py < 2; py++)
#pragma unroll
                    for (int px = 0; px < 2; px++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int i = 0; i < 2; i++)   // output (2q+py, 2r+px) sees S[q+py-j][r+px-i] through w[1-py+2j][1-px+2i]
                                acc[cl][py][px] = fmaf(v[1 + py - j][1 + px - i], wp[cl * 16 + (1 - py + 2 * j) * 4 + (1 - px + 2 * i)],
                                                       acc[cl][py][px]);
        }
#pragma unroll
        for (int cl = 0; cl < CL; cl++)
#pragma unroll
            for (int py = 0; py < 2; py++) {
                float2* dst = reinterpret_cast<float2*>(L + (((size_t)b * CL + cl) * g.Hl + 2 * q + py) * g.Wl + 2 * r);
                *dst = make_float2(acc[cl][py][0], acc[cl][py][1]);
            }
    }
}


// The (b, i) plane of one channel of a (B, C, HW) tensor, walked by a grid (chunks, ...) of 256-thread workgroups: f(b, i)
// with i a multiple of V, V consecutive elements per visit.  The batch index comes from a 3-instruction fp32 product
// instead of a 64-bit division where the plane is small enough for that to be exact ((e + 0.5) / d is at least 0.5 / d away
// from an integer; the product's rounding error is below that while B * HW / V < 2^22).
template <int V, class F>
__device__ __forceinline__ void plane_loop(int B, int HW, F f) {
    const int HWV = HW / V;
    const long long total = (long long)B * HWV;
    if (total < (1LL << 22)) {
        const float inv = 1.0f / (float)HWV;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < (int)total; e += gridDim.x * 256) {
            const int b = (int)(((float)e + 0.5f) * inv);
            f(b, (e - b * HWV) * V);
        }
    } else {
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            const long long b = e / HWV;
            f((int)b, (int)(e - b * HWV) * V);
        }
    }
}
template <int V>
struct VecF;
template <>
struct VecF<1> {
    float v[1];
    __device__ __forceinline__ static VecF ld(const float* p) { return VecF{{*p}}; }
    __device__ __forceinline__ void st(float* p) const { *p = v[0]; }
};
template <>
struct VecF<4> {
    float v[4];
    __device__ __forceinline__ static VecF ld(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        return VecF{{t.x, t.y, t.z, t.w}};
    }
    __device__ __forceinline__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
__device__ __forceinline__ bool vec4_ok(int HW, long long s0 = 0, long long s1 = 0, long long s2 = 0) {
    return ((HW | s0 | s1 | s2) & 3) == 0;
}

// The decoder's concatenated tensor (unet.py:157-160: channels 0..Ch-1 = the gated ConvTranspose2d output u * att[b][c], channels
// Ch..2Ch-1 = the encoder skip) is never materialised: the BatchNorm kernels that would read it take the two sources instead.
// u == nullptr: an ordinary tensor, z[b * zbs + c * HW + i].
struct ZCat {
    const float* u;
    const float* att;
    const float* skip;
    int Ch;
};
// plane (b, c) of a BatchNorm input and the factor its values are to be multiplied by (1 for everything but the gated half)
__device__ __forceinline__ const float* z_plane(const float* z, long long zbs, const ZCat& zc, int b, int c, int HW, float& k) {
    k = 1.f;
    if (!zc.u) return z + b * zbs + (long long)c * HW;
    if (c < zc.Ch) {
        k = zc.att[(long long)b * zc.Ch + c];
        return zc.u + ((long long)b * zc.Ch + c) * HW;
    }
    return zc.skip + ((long long)b * zc.Ch + (c - zc.Ch)) * HW;
}

// ---------------------------------------------------------------------------------------------
// per-channel reductions and BatchNorm (2-d: (B,C,HW); 1-d: HW = 1)
// ---------------------------------------------------------------------------------------------

// sums[c*sstride + 0..1] += sum x, sum x^2 over (b, i) of x[b*bs + c*HW + i] (the squares only if want_sq).
// grid (chunks, C)
template <int V>
__device__ __forceinline__ void chan_sums_body(const float* __restrict__ x, long long bs, int B, int HW, double* __restrict__ sums,
                                               int sstride, int want_sq, double* red) {
    const int c = blockIdx.y;
    double s1 = 0, s2 = 0;
    plane_loop<V>(B, HW, [&](int b, int i) {
        const VecF<V> t = VecF<V>::ld(x + b * bs + (long long)c * HW + i);
#pragma unroll
        for (int j = 0; j < V; j++) {
            const double v = (double)t.v[j];
            s1 += v;
            s2 += v * v;
        }
    });
    const double t1 = block_sum(s1, red);
    if (threadIdx.x == 0) atomicAdd(&sums[(size_t)c * sstride], t1);
    if (want_sq) {
        const double t2 = block_sum(s2, red);
        if (threadIdx.x == 0) atomicAdd(&sums[(size_t)c * sstride + 1], t2);
    }
}
__global__ void __launch_bounds__(256) k_chan_sums(const float* __restrict__ x, long long bs, int B, int HW,
                                                   double* __restrict__ sums, int sstride, int want_sq) {
    __shared__ double red[4];
    if (vec4_ok(HW, bs)) chan_sums_body<4>(x, bs, B, HW, sums, sstride, want_sq, red);
    else chan_sums_body<1>(x, bs, B, HW, sums, sstride, want_sq, red);
}

// a = dropout(relu(bn(z))), s = relu(bn(z)) (optional).  stat_mode 0: mean / invstd from `saved`; 1: running statistics
// (eval); 2: batch statistics straight from the fp64 sums {sum, sum of squares} over N values per channel - every
// workgroup derives its channel's constants itself and the first one of each channel also writes `saved` and updates the
// running statistics as torch does (unbiased running_var), so that no separate finalize launch is needed.
// z is read with batch stride zbs; s and a are contiguous (B,C,HW).  grid (chunks, C)
__global__ void __launch_bounds__(256) k_bn_act(const float* __restrict__ z, long long zbs, int B, int C, int HW,
                                                float* __restrict__ saved, float* __restrict__ rmean, float* __restrict__ rvar,
                                                float eps, int stat_mode, const double* __restrict__ sums, double N,
                                                float momentum, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, Drop d, float* __restrict__ s_out,
                                                float* __restrict__ a_out, ZCat zc, double* __restrict__ skip_sums) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    double q1 = 0, q2 = 0;       // sum s, sum s^2 (skip_sums: the decoder's BatchNorm statistics of this skip, [c][2])
    float mean, invstd;
    if (stat_mode == 2) {
        const double m = sums[2 * c] / N;
        double var = sums[2 * c + 1] / N - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            saved[2 * c] = mean;
            saved[2 * c + 1] = invstd;
            const double unbiased = N > 1 ? var * N / (N - 1) : var;
            rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + (double)momentum * m);
            rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + (double)momentum * unbiased);
        }
    } else if (stat_mode == 1) {
        mean = rmean[c];
        invstd = 1.f / sqrtf(rvar[c] + eps);
    } else {
        mean = saved[2 * c];
        invstd = saved[2 * c + 1];
    }
    const float sc = invstd * gamma[c], sh = beta[c];
    auto body = [&](auto VT) {
        constexpr int V = decltype(VT)::value;
        plane_loop<V>(B, HW, [&](int b, int i) {
            float zk;
            VecF<V> t = VecF<V>::ld(z_plane(z, zbs, zc, b, c, HW, zk) + i);
            const unsigned long long o = ((unsigned long long)b * C + c) * HW + i;
            VecF<V> a;
#pragma unroll
            for (int j = 0; j < V; j++) {
                t.v[j] = fmaxf(fmaf(__fmul_rn(t.v[j], zk) - mean, sc, sh), 0.f);      // (the product rounded by itself, as when it was stored)
                a.v[j] = t.v[j] * drop_factor(d, o + j);
                if (skip_sums) q1 += (double)t.v[j], q2 += (double)t.v[j] * (double)t.v[j];
            }
            if (s_out) t.st(s_out + o);
            if (a_out) a.st(a_out + o);
        });
    };
    if (vec4_ok(HW, zc.u ? 0 : zbs)) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 1>{});
    if (skip_sums) {
        const double t1 = block_sum(q1, red);
        const double t2 = block_sum(q2, red);
        if (threadIdx.x == 0) {
            atomicAdd(&skip_sums[2 * c], t1);
            atomicAdd(&skip_sums[2 * c + 1], t2);
        }
    }
}

// backward, pass 1: g = (gA * dropmask + gB) * [bn(z) > 0]; accumulates sums[c][0..1] += sum g, sum g*xhat and, when g_out is
// given, writes g (contiguous) - since round 3 it is not: pass 2 forms g again from the same operands instead of reading back
// a map this pass wrote (one tensor-sized write less per BatchNorm layer).  gA / gB may be null; each has its own batch
// stride.  grid (chunks, C)
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(const float* __restrict__ gA, long long gAbs,
                                                       const float* __restrict__ gB, long long gBbs,
                                                       const float* __restrict__ z, long long zbs, int B, int C, int HW,
                                                       const float* __restrict__ saved, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, Drop d, float* __restrict__ g_out,
                                                       double* __restrict__ sums, ZCat zc) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    const float mean = saved[2 * c], invstd = saved[2 * c + 1];
    const float ga = gamma[c], be = beta[c];
    const float sc = invstd * ga;
    double s1 = 0, s2 = 0;
    auto body = [&](auto VT) {
        constexpr int V = decltype(VT)::value;
        plane_loop<V>(B, HW, [&](int b, int i) {
            const long long ci = (long long)c * HW + i;
            float zk;
            VecF<V> zt = VecF<V>::ld(z_plane(z, zbs, zc, b, c, HW, zk) + i);
#pragma unroll
            for (int j = 0; j < V; j++) zt.v[j] = __fmul_rn(zt.v[j], zk);
            const unsigned long long o = ((unsigned long long)b * C + c) * HW + i;
            VecF<V> g, ta, tb;
            if (gA) ta = VecF<V>::ld(gA + b * gAbs + ci);
            if (gB) tb = VecF<V>::ld(gB + b * gBbs + ci);
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (zt.v[j] - mean) * invstd;
                float gv = 0.f;
                if (gA) gv = ta.v[j] * drop_factor(d, o + j);
                if (gB) gv += tb.v[j];
                if (!(fmaf(zt.v[j] - mean, sc, be) > 0.f)) gv = 0.f;      // the forward's own expression: the same decision, bit for bit
                g.v[j] = gv;
                s1 += (double)gv;
                s2 += (double)gv * (double)xh;
            }
            if (g_out) g.st(g_out + o);
        });
    };
    if (vec4_ok(HW, zc.u ? 0 : zbs, gA ? gAbs : 0, gB ? gBbs : 0)) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 1>{});
    const double t1 = block_sum(s1, red);
    const double t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * c], t1);
        atomicAdd(&sums[2 * c + 1], t2);
    }
}

// backward, pass 2 from the operands of pass 1: dz = gamma*invstd*(g - sum_g/N - xhat*sum_gx/N) with g formed as there;
// block x == 0 also adds dgamma = sum g*xhat, dbeta = sum g to the gradient accumulator.  grid (chunks, C)
__global__ void __launch_bounds__(256) k_bn_bwd_apply2(const float* __restrict__ gA, long long gAbs, const float* __restrict__ gB,
                                                       long long gBbs, const float* __restrict__ z, long long zbs, int B, int C,
                                                       int HW, const float* __restrict__ saved, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, Drop d, const double* __restrict__ sums,
                                                       double N, double* __restrict__ acc_gamma, double* __restrict__ acc_beta,
                                                       float* __restrict__ dz, ZCat zc) {
    const int c = blockIdx.y;
    const float mean = saved[2 * c], invstd = saved[2 * c + 1];
    const float ga = gamma[c], be = beta[c];
    const float k1 = ga * invstd, sc = invstd * ga;
    const float m1 = (float)(sums[2 * c] / N), m2 = (float)(sums[2 * c + 1] / N);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        acc_gamma[c] += sums[2 * c + 1];
        acc_beta[c] += sums[2 * c];
    }
    auto body = [&](auto VT) {
        constexpr int V = decltype(VT)::value;
        plane_loop<V>(B, HW, [&](int b, int i) {
            const long long ci = (long long)c * HW + i;
            float zk;
            VecF<V> zt = VecF<V>::ld(z_plane(z, zbs, zc, b, c, HW, zk) + i);
#pragma unroll
            for (int j = 0; j < V; j++) zt.v[j] = __fmul_rn(zt.v[j], zk);
            const unsigned long long o = ((unsigned long long)b * C + c) * HW + i;
            VecF<V> g, ta, tb;
            if (gA) ta = VecF<V>::ld(gA + b * gAbs + ci);
            if (gB) tb = VecF<V>::ld(gB + b * gBbs + ci);
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (zt.v[j] - mean) * invstd;
                float gv = 0.f;
                if (gA) gv = ta.v[j] * drop_factor(d, o + j);
                if (gB) gv += tb.v[j];
                if (!(fmaf(zt.v[j] - mean, sc, be) > 0.f)) gv = 0.f;      // the forward's own expression: the same decision, bit for bit
                g.v[j] = k1 * (gv - m1 - xh * m2);
            }
            g.st(dz + o);
        });
    };
    if (vec4_ok(HW, zc.u ? 0 : zbs, gA ? gAbs : 0, gB ? gBbs : 0)) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 1>{});
}

// pass 2 in the decoder's form: one workgroup per (b, c) plane of the (never written) concatenated tensor - same arithmetic per
// element as k_bn_bwd_apply2 - which lets the gated half leave da[b][c] = sum_i dz[b][c][i] * u[b][c][i] on the way (the gate's
// gradient; it took k_att_da another pass over dz and u).  No gB here.  grid (1, C = 2 Ch, B)
__global__ void __launch_bounds__(256) k_bn_bwd_apply2_planes(const float* __restrict__ gA, long long gAbs, int C, int HW,
                                                              const float* __restrict__ saved, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, Drop d, const double* __restrict__ sums,
                                                              double N, double* __restrict__ acc_gamma, double* __restrict__ acc_beta,
                                                              float* __restrict__ dz, ZCat zc, float* __restrict__ da) {
    __shared__ double red[4];
    const int c = blockIdx.y, b = blockIdx.z;
    const float mean = saved[2 * c], invstd = saved[2 * c + 1];
    const float ga = gamma[c], be = beta[c];
    const float k1 = ga * invstd, sc = invstd * ga;
    const float m1 = (float)(sums[2 * c] / N), m2 = (float)(sums[2 * c + 1] / N);
    if (b == 0 && threadIdx.x == 0) {
        acc_gamma[c] += sums[2 * c + 1];
        acc_beta[c] += sums[2 * c];
    }
    const bool gated = c < zc.Ch;
    const float zk = gated ? zc.att[(long long)b * zc.Ch + c] : 1.f;
    const float* zp = gated ? zc.u + ((long long)b * zc.Ch + c) * HW : zc.skip + ((long long)b * zc.Ch + (c - zc.Ch)) * HW;
    const float* gp = gA + b * gAbs + (long long)c * HW;
    const unsigned long long o0 = ((unsigned long long)b * C + c) * HW;
    float* op = dz + o0;
    double s = 0.0;
    auto body = [&](auto VT) {
        constexpr int V = decltype(VT)::value;
        for (int i = threadIdx.x * V; i < HW; i += 256 * V) {
            const VecF<V> ut = VecF<V>::ld(zp + i), ta = VecF<V>::ld(gp + i);
            VecF<V> g;
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float zv = __fmul_rn(ut.v[j], zk);
                const float xh = (zv - mean) * invstd;
                float gv = ta.v[j] * drop_factor(d, o0 + i + j);
                if (!(fmaf(zv - mean, sc, be) > 0.f)) gv = 0.f;
                g.v[j] = k1 * (gv - m1 - xh * m2);
                if (gated) s += (double)(g.v[j] * ut.v[j]);
            }
            g.st(op + i);
        }
    };
    if (vec4_ok(HW, gAbs)) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 1>{});
    if (gated) {
        const double t = block_sum(s, red);
        if (threadIdx.x == 0) da[(size_t)b * zc.Ch + c] = (float)t;
    }
}

// g = gin * dropmask * [h > 0]  (ReLU + dropout backward where there is no BatchNorm: the second Linear of each stack)
__global__ void __launch_bounds__(256) k_relu_drop_bwd(float* __restrict__ g, const float* __restrict__ gin, const float* __restrict__ h,
                                                       long long n, Drop d) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        g[i] = h[i] > 0.f ? gin[i] * drop_factor(d, (unsigned long long)i) : 0.f;
}

// a = dropout(relu(h))  (forward of the same)
__global__ void __launch_bounds__(256) k_relu_drop(const float* __restrict__ h, long long n, Drop d, float* __restrict__ a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        a[i] = fmaxf(h[i], 0.f) * drop_factor(d, (unsigned long long)i);
}

// ---------------------------------------------------------------------------------------------
// channel attention (unet.py:23-39) and the skip concat (unet.py:157-160)
// ---------------------------------------------------------------------------------------------

// pool[b*C+c] = {mean, max, (float)argmax} of u[b][c][:]; psum[b*C+c] = {sum u, sum u^2} (training: k_att_fwd turns them into the
// BatchNorm sums of the gated half of the concatenated tensor).  grid (B*C)
__global__ void __launch_bounds__(256) k_pool(const float* __restrict__ u, int HW, float* __restrict__ pool,
                                              double* __restrict__ psum) {
    __shared__ double red[4];
    __shared__ float smax[256];
    __shared__ int sarg[256];
    const float* p = u + (size_t)blockIdx.x * HW;
    double s = 0, s2 = 0;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    if ((HW & 3) == 0) {   // 16-byte loads; elements visited in increasing index per thread, so the first maximum is kept
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
            const float4 t4 = p4[i];
            const float v[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                s += (double)v[j];
                s2 += (double)v[j] * (double)v[j];
                if (v[j] > mx) {
                    mx = v[j];
                    am = 4 * i + j;
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const float v = p[i];
            s += (double)v;
            s2 += (double)v * (double)v;
            if (v > mx) {
                mx = v;
                am = i;
            }
        }
    }
    smax[threadIdx.x] = mx;
    sarg[threadIdx.x] = am;
    const double t = block_sum(s, red);
    const double t2 = psum ? block_sum(s2, red) : 0.0;      // (uniform)
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const float o = smax[threadIdx.x + off];
            const int oa = sarg[threadIdx.x + off];
            if (o > smax[threadIdx.x] || (o == smax[threadIdx.x] && oa < sarg[threadIdx.x])) {
                smax[threadIdx.x] = o;
                sarg[threadIdx.x] = oa;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pool[3 * blockIdx.x + 0] = (float)(t / HW);
        pool[3 * blockIdx.x + 1] = smax[0];
        pool[3 * blockIdx.x + 2] = __int_as_float(sarg[0]);
        if (psum) psum[2 * blockIdx.x] = t, psum[2 * blockIdx.x + 1] = t2;
    }
}

// att[b][c] = sigmoid(sum_r W2[c][r] * (relu(W1[r]·avg) + relu(W1[r]·max)));  hid[b][0..R) = W1·avg, [R..2R) = W1·max;
// sums != nullptr: sums[c][0..1] += att * sum u, att^2 * sum u^2 (the gated half's BatchNorm statistics without another pass over u)
// grid (B), block 256, dynamic LDS (2C + 2R) floats
__global__ void __launch_bounds__(256) k_att_fwd(const float* __restrict__ pool, int C, int R, const float* __restrict__ W1,
                                                 const float* __restrict__ W2, float* __restrict__ att,
                                                 float* __restrict__ hid, const double* __restrict__ psum,
                                                 double* __restrict__ sums) {
    extern __shared__ float lds_f[];
    float* avg = lds_f;
    float* mx = lds_f + C;
    float* h = lds_f + 2 * C;   // 2R pre-activations
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        avg[c] = pool[3 * ((size_t)b * C + c)];
        mx[c] = pool[3 * ((size_t)b * C + c) + 1];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < 2 * R; r += 256) {
        const float* src = r < R ? avg : mx;
        const float* wr = W1 + (size_t)(r < R ? r : r - R) * C;
        float acc = 0.f;
        for (int c = 0; c < C; c++) acc = fmaf(wr[c], src[c], acc);
        h[r] = acc;
        hid[(size_t)b * 2 * R + r] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float oa = 0.f, om = 0.f;
        for (int r = 0; r < R; r++) {
            oa = fmaf(W2[(size_t)c * R + r], fmaxf(h[r], 0.f), oa);
            om = fmaf(W2[(size_t)c * R + r], fmaxf(h[R + r], 0.f), om);
        }
        const float a = 1.f / (1.f + expf(-(oa + om)));
        att[(size_t)b * C + c] = a;
        if (sums) {   // channel c of the concatenated tensor is a * u: its sums over this sample's map
            atomicAdd(&sums[2 * c], (double)a * psum[2 * ((size_t)b * C + c)]);
            atomicAdd(&sums[2 * c + 1], (double)a * (double)a * psum[2 * ((size_t)b * C + c) + 1]);
        }
    }
}

// backward of k_att_fwd for one sample per block: dpool[b*C+c] = {davg, dmax}; weight gradients by fp64 atomics.
// grid (B), block 256, dynamic LDS (3C + 4R) floats
__global__ void __launch_bounds__(256) k_att_bwd(const float* __restrict__ pool, const float* __restrict__ att,
                                                 const float* __restrict__ hid, const float* __restrict__ da, int C, int R,
                                                 const float* __restrict__ W1, const float* __restrict__ W2,
                                                 double* __restrict__ accW1, double* __restrict__ accW2,
                                                 float* __restrict__ dpool) {
    extern __shared__ float lds_f[];
    float* avg = lds_f;
    float* mx = lds_f + C;
    float* dout = lds_f + 2 * C;        // dL/d(pre-sigmoid)
    float* h = lds_f + 3 * C;           // 2R pre-activations
    float* dh = lds_f + 3 * C + 2 * R;  // 2R gradients wrt pre-activations
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        avg[c] = pool[3 * ((size_t)b * C + c)];
        mx[c] = pool[3 * ((size_t)b * C + c) + 1];
        const float a = att[(size_t)b * C + c];
        dout[c] = da[(size_t)b * C + c] * a * (1.f - a);
    }
    for (int r = threadIdx.x; r < 2 * R; r += 256) h[r] = hid[(size_t)b * 2 * R + r];
    __syncthreads();
    for (int r = threadIdx.x; r < 2 * R; r += 256) {
        const int rr = r < R ? r : r - R;
        float acc = 0.f;
        for (int c = 0; c < C; c++) acc = fmaf(W2[(size_t)c * R + rr], dout[c], acc);
        dh[r] = h[r] > 0.f ? acc : 0.f;
    }
    for (int e = threadIdx.x; e < C * R; e += 256) {
        const int c = e / R, r = e - c * R;
        const float v = dout[c] * (fmaxf(h[r], 0.f) + fmaxf(h[R + r], 0.f));
        if (v != 0.f) atomicAdd(&accW2[e], (double)v);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < R * C; e += 256) {
        const int r = e / C, c = e - r * C;
        const float v = dh[r] * avg[c] + dh[R + r] * mx[c];
        if (v != 0.f) atomicAdd(&accW1[e], (double)v);
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float ga = 0.f, gm = 0.f;
        for (int r = 0; r < R; r++) {
            ga = fmaf(W1[(size_t)r * C + c], dh[r], ga);
            gm = fmaf(W1[(size_t)r * C + c], dh[R + r], gm);
        }
        dpool[2 * ((size_t)b * C + c)] = ga;
        dpool[2 * ((size_t)b * C + c) + 1] = gm;
    }
}

// du[b][c][i] = dcat[b][c][i]*att + davg/HW + [i == argmax]*dmax;  accb[c] += sum du (the ConvTranspose2d bias gradient)
// grid (chunks, C)
__global__ void __launch_bounds__(256) k_scale_bwd(const float* __restrict__ dcat, const float* __restrict__ att,
                                                   const float* __restrict__ pool, const float* __restrict__ dpool, int B,
                                                   int C, int HW, float* __restrict__ du, double* __restrict__ accb) {
    __shared__ double red[4];
    const int c = blockIdx.y;
    const float inv = 1.f / (float)HW;
    double s = 0;
    auto body = [&](auto VT) {
        constexpr int V = decltype(VT)::value;
        plane_loop<V>(B, HW, [&](int b, int i) {
            const long long bc = (long long)b * C + c;
            VecF<V> t = VecF<V>::ld(dcat + ((long long)b * 2 * C + c) * HW + i);
            const float ka = att[bc], kp = dpool[2 * bc] * inv;
            const int imax = __float_as_int(pool[3 * bc + 2]);
#pragma unroll
            for (int j = 0; j < V; j++) {
                float v = t.v[j] * ka + kp;
                if (i + j == imax) v += dpool[2 * bc + 1];
                t.v[j] = v;
                s += (double)v;
            }
            t.st(du + bc * HW + i);
        });
    };
    if (vec4_ok(HW)) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 1>{});
    const double t = block_sum(s, red);
    if (threadIdx.x == 0 && accb) atomicAdd(&accb[c], t);
}

// ---------------------------------------------------------------------------------------------
// Linear layers, W[out][in]
// ---------------------------------------------------------------------------------------------

// out[b][o] = bias[o] + sum_i in[b][i] * W[o][i].  grid (nout, ceil(B/8)), block 256
__global__ void __launch_bounds__(256) k_lin_fwd(int B, int nin, int nout, const float* __restrict__ in,
                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                 float* __restrict__ out) {
    __shared__ double red[4];
    const int o = blockIdx.x, b0 = blockIdx.y * 8;
    const int nb = min(8, B - b0);
    const float* wp = W + (size_t)o * nin;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nin; i += 256) {
        const float wv = wp[i];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < nb) acc[j] = fmaf(in[(size_t)(b0 + j) * nin + i], wv, acc[j]);
    }
    for (int j = 0; j < nb; j++) {
        const double t = block_sum((double)acc[j], red);
        if (threadIdx.x == 0) out[(size_t)(b0 + j) * nout + o] = (float)(t + (double)bias[o]);
    }
}

// gin[b][i] = sum_o gout[b][o] * W[o][i]
__global__ void __launch_bounds__(256) k_lin_dgrad(int B, int nin, int nout, const float* __restrict__ gout,
                                                   const float* __restrict__ W, float* __restrict__ gin) {
    const long long total = (long long)B * nin;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long b = idx / nin, i = idx - b * nin;
        const float* gp = gout + b * nout;
        float acc = 0.f;
        for (int o = 0; o < nout; o++) acc = fmaf(gp[o], W[(size_t)o * nin + i], acc);
        gin[idx] = acc;
    }
}

// accW[o][i] += sum_b gout[b][o] * in[b][i];  accB[o] += sum_b gout[b][o]
__global__ void __launch_bounds__(256) k_lin_wgrad(int B, int nin, int nout, const float* __restrict__ gout,
                                                   const float* __restrict__ in, double* __restrict__ accW,
                                                   double* __restrict__ accB) {
    const long long total = (long long)nout * nin;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long o = idx / nin, i = idx - o * nin;
        float sum = 0.f, bsum = 0.f;
        for (int b = 0; b < B; b++) {
            const float gv = gout[(size_t)b * nout + o];
            sum = fmaf(gv, in[(size_t)b * nin + i], sum);
            bsum += gv;
        }
        accW[idx] += (double)sum;
        if (i == 0) accB[o] += (double)bsum;
    }
}

// ---------------------------------------------------------------------------------------------
// loss: masked MSE + lambda * (1 - mean masked Pearson r)  (unet.py:316-321,635-678)
// ---------------------------------------------------------------------------------------------

// dst[b][:] = src[perm ? perm[start+b] : start+b][:]
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ src, const int* __restrict__ perm, long long start,
                                                int B, long long E, float* __restrict__ dst) {
    if ((E & 3) == 0) {   // rows are whole float4s: 16-byte copies, one 64-bit division per four elements
        const long long E4 = E >> 2, total4 = (long long)B * E4;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total4; o += (long long)gridDim.x * 256) {
            const long long b = o / E4, i = o - b * E4;
            const long long s = perm ? (long long)perm[start + b] : start + b;
            d4[o] = s4[s * E4 + i];
        }
        return;
    }
    const long long total = (long long)B * E;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const long long b = o / E, i = o - b * E;
        const long long s = perm ? (long long)perm[start + b] : start + b;
        dst[o] = src[s * E + i];
    }
}

struct LossSrc {
    const float* target;  // dataset (N, C, HW)
    const float* mask;    // dataset (N, Cm, HW) or nullptr (all ones, counted with C channels)
    const int* perm;
    long long start;
    int Cm;               // 1 or C
};

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ls[(b*C+c)*8 + k] += {sum m, sum m p, sum m t, sum m p^2, sum m t^2, sum m p t, sum (m (p-t))^2}; p = sigmoid(u) or u
// grid (chunks, B*C)
__global__ void __launch_bounds__(256) k_loss_sums(const float* __restrict__ u, int apply_sigmoid, LossSrc ls_src, int C,
                                                   int HW, double* __restrict__ ls) {
    __shared__ double red[4];
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const long long smp = ls_src.perm ? (long long)ls_src.perm[ls_src.start + b] : ls_src.start + b;
    const float* tp = ls_src.target + (smp * C + c) * HW;
    const float* mp = ls_src.mask ? ls_src.mask + (smp * ls_src.Cm + (ls_src.Cm == 1 ? 0 : c)) * HW : nullptr;
    const float* up = u + (size_t)bc * HW;
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    auto one = [&](float uv, float tv, float mv) {
        const double m = (double)mv;
        const double p = (double)(apply_sigmoid ? sigmoidf(uv) : uv);
        const double t = (double)tv;
        const float df = ((float)p - (float)t) * (float)m;   // the reference forms (pred - target) * mask in fp32
        a[0] += m;
        a[1] += m * p;
        a[2] += m * t;
        a[3] += m * p * p;
        a[4] += m * t * t;
        a[5] += m * p * t;
        a[6] += (double)df * (double)df;
    };
    if ((HW & 3) == 0) {      // 16-byte loads (the planes start on multiples of HW floats)
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 1024) {
            const float4 uv = *reinterpret_cast<const float4*>(up + i), tv = *reinterpret_cast<const float4*>(tp + i);
            const float4 mv = mp ? *reinterpret_cast<const float4*>(mp + i) : make_float4(1.f, 1.f, 1.f, 1.f);
            one(uv.x, tv.x, mv.x), one(uv.y, tv.y, mv.y), one(uv.z, tv.z, mv.z), one(uv.w, tv.w, mv.w);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) one(up[i], tp[i], mp ? mp[i] : 1.f);
    }
    for (int k = 0; k < 7; k++) {
        const double t = block_sum(a[k], red);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(&ls[(size_t)bc * 8 + k], t);
    }
}

// one block: loss values out2 = {mse, 1 - mean r} and the gradient coefficients coef[bc] = {k0, k1, k2, k3}:
//   dL/dp = m*(k0 + k1*t + k2*p) + k3*m^2*(p - t)
__global__ void __launch_bounds__(256) k_loss_finalize(const double* __restrict__ ls, int B, int C, int Cm, int has_mask,
                                                       double lambda_p, double* __restrict__ out2,
                                                       float* __restrict__ coef) {
    __shared__ double red[4];
    const int n_bc = B * C;
    double mtot = 0, sq = 0, rsum = 0;
    for (int bc = threadIdx.x; bc < n_bc; bc += 256) {
        const double* s = ls + (size_t)bc * 8;
        const int c = bc % C;
        if (!has_mask || Cm == C || c == 0) mtot += s[0];   // torch.sum(mask) counts the mask tensor as stored
        sq += s[6];
    }
    mtot = block_sum(mtot, red);
    __shared__ double sh_m;
    if (threadIdx.x == 0) sh_m = mtot;
    __syncthreads();
    mtot = sh_m;
    sq = block_sum(sq, red);
    for (int bc = threadIdx.x; bc < n_bc; bc += 256) {
        const double* s = ls + (size_t)bc * 8;
        const double n = s[0], np = n + 1e-8;
        const double mup = s[1] / np, mut = s[2] / np;
        const double vp = (s[3] - 2 * mup * s[1] + mup * mup * n) / np + 1e-8;
        const double vt = (s[4] - 2 * mut * s[2] + mut * mut * n) / np + 1e-8;
        const double sp = sqrt(vp), st = sqrt(vt);
        const double A = s[5] - mup * s[2] - mut * s[1] + mup * mut * n;
        const double r = A / (sp * st) / n;
        rsum += r;
        if (coef) {
            const double kappa = -lambda_p / (double)n_bc;
            const double q = 1.0 / (sp * st * n);
            const double Sp = s[1] - mup * n, St = s[2] - mut * n;   // sum m (p - mup), sum m (t - mut)
            const double e1 = q;
            const double e2 = -q * A / (vp * np);
            const double e0 = q * (-mut - St / np) + e2 * (-mup - Sp / np);
            coef[4 * bc + 0] = (float)(kappa * e0);
            coef[4 * bc + 1] = (float)(kappa * e1);
            coef[4 * bc + 2] = (float)(kappa * e2);
            coef[4 * bc + 3] = (float)(2.0 / mtot);
        }
    }
    rsum = block_sum(rsum, red);
    if (threadIdx.x == 0) {
        out2[0] = sq / mtot;
        out2[1] = 1.0 - rsum / (double)n_bc;
    }
}

// du = dL/dp * p * (1 - p),  p = sigmoid(u), and the last layer's bias gradient bias_acc[c] += sum du (nullptr: not wanted).
// grid (chunks, B*C): a workgroup walks a slice of one (b, c) plane, 16-byte accesses when HW % 4 == 0
__global__ void __launch_bounds__(256) k_loss_grad(const float* __restrict__ u, LossSrc ls_src, int B, int C, int HW,
                                                   const float* __restrict__ coef, float* __restrict__ du,
                                                   double* __restrict__ bias_acc) {
    __shared__ double red[4];
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const long long smp = ls_src.perm ? (long long)ls_src.perm[ls_src.start + b] : ls_src.start + b;
    const float* tp = ls_src.target + (smp * C + c) * HW;
    const float* mp = ls_src.mask ? ls_src.mask + (smp * ls_src.Cm + (ls_src.Cm == 1 ? 0 : c)) * HW : nullptr;
    const float* up = u + (long long)bc * HW;
    float* dp_out = du + (long long)bc * HW;
    const float4 k = reinterpret_cast<const float4*>(coef)[bc];
    double s = 0.0;
    auto one = [&](float uv, float t, float m) {
        const float p = sigmoidf(uv);
        const float dp = m * (k.x + k.y * t + k.z * p) + k.w * m * m * (p - t);
        const float d = dp * p * (1.f - p);
        s += (double)d;
        return d;
    };
    if ((HW & 3) == 0) {
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += gridDim.x * 1024) {
            const float4 uv = *reinterpret_cast<const float4*>(up + i), t = *reinterpret_cast<const float4*>(tp + i);
            const float4 m = mp ? *reinterpret_cast<const float4*>(mp + i) : make_float4(1.f, 1.f, 1.f, 1.f);
            float4 d;
            d.x = one(uv.x, t.x, m.x), d.y = one(uv.y, t.y, m.y), d.z = one(uv.z, t.z, m.z), d.w = one(uv.w, t.w, m.w);
            *reinterpret_cast<float4*>(dp_out + i) = d;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) dp_out[i] = one(up[i], tp[i], mp ? mp[i] : 1.f);
    }
    if (bias_acc) {
        const double t = block_sum(s, red);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(&bias_acc[c], t);
    }
}

__global__ void __launch_bounds__(256) k_sigmoid(const float* __restrict__ u, long long n, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = sigmoidf(u[i]);
}

// ---------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW single-tensor formula, unet.py:457): decoupled weight decay, then Adam.
// Consumes and clears the fp64 gradient accumulator; optionally exports the fp32 gradient.
// ---------------------------------------------------------------------------------------------
struct Hyper {
    double lr, beta1, beta2, eps, wd;
};

// parameter ranges whose gradient was stored as fp32 (one writer per element, the whole sum in one store: the big Linear
// layers, kernels_unet_lin.h) in the first half of their fp64 accumulator slots: element i of [lo, hi) is float number i - lo
// counted from (float*)(gacc + lo).  Such a range is not cleared by its consumer: its producer overwrites it.
struct F32Ranges {
    long long lo[4], hi[4];
    int n;
};
__device__ __forceinline__ bool f32_grad(const F32Ranges& fr, long long i, const double* gacc, float& g) {
    for (int r = 0; r < fr.n; r++)
        if (i >= fr.lo[r] && i < fr.hi[r]) {
            g = reinterpret_cast<const float*>(gacc + fr.lo[r])[i - fr.lo[r]];
            return true;
        }
    return false;
}

// step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t), decay = 1 - lr * wd: formed on the host in fp64 (the step number is
// a host-side count here), so that no workgroup starts with two fp64 pow() calls in front of its loads
struct AdamwConsts {
    float step_size, bc2_sqrt, decay, b1, b2, eps;
};
__device__ __forceinline__ void adamw_one(float g, float& w, float& mi, float& vi, const AdamwConsts& c) {
    w *= c.decay;
    mi = mi + (g - mi) * (1.f - c.b1);              // exp_avg.lerp_(grad, 1 - beta1)
    vi = fmaf(g * g, 1.f - c.b2, vi * c.b2);        // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
    w -= c.step_size * (mi / denom);
}

// four parameters per thread and trip (16-byte accesses; the arenas are 16-byte aligned); a group of four that straddles the
// edge of an fp32 range takes the element-wise path.  grid: a few workgroups per CU, grid-stride
__global__ void __launch_bounds__(256) k_adamw(long long n, float* __restrict__ p, double* __restrict__ gacc,
                                               float* __restrict__ m, float* __restrict__ v, AdamwConsts c, F32Ranges fr) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; 4 * t < n; t += (long long)gridDim.x * 256) {
        const long long i0 = 4 * t;
        int r = -1;
        bool whole = i0 + 3 < n;
        for (int k = 0; k < fr.n; k++) {
            const bool in0 = i0 >= fr.lo[k] && i0 < fr.hi[k], in3 = i0 + 3 >= fr.lo[k] && i0 + 3 < fr.hi[k];
            if (in0 && in3) r = k;
            else if (in0 || in3 || (fr.lo[k] > i0 && fr.lo[k] <= i0 + 3)) whole = false;
        }
        if (whole) {
            float4 P = *reinterpret_cast<const float4*>(p + i0), M = *reinterpret_cast<const float4*>(m + i0);
            float4 V = *reinterpret_cast<const float4*>(v + i0);
            float g[4];
            if (r >= 0) {
                const float* gf = reinterpret_cast<const float*>(gacc + fr.lo[r]) + (i0 - fr.lo[r]);
                if ((fr.lo[r] & 3) == 0) {
                    const float4 G = *reinterpret_cast<const float4*>(gf);
                    g[0] = G.x, g[1] = G.y, g[2] = G.z, g[3] = G.w;
                } else {
                    g[0] = gf[0], g[1] = gf[1], g[2] = gf[2], g[3] = gf[3];
                }
            } else {
                double2* gd = reinterpret_cast<double2*>(gacc + i0);
                const double2 a = gd[0], b = gd[1];
                g[0] = (float)a.x, g[1] = (float)a.y, g[2] = (float)b.x, g[3] = (float)b.y;
                gd[0] = make_double2(0.0, 0.0), gd[1] = make_double2(0.0, 0.0);
            }
            adamw_one(g[0], P.x, M.x, V.x, c);
            adamw_one(g[1], P.y, M.y, V.y, c);
            adamw_one(g[2], P.z, M.z, V.z, c);
            adamw_one(g[3], P.w, M.w, V.w, c);
            *reinterpret_cast<float4*>(p + i0) = P;
            *reinterpret_cast<float4*>(m + i0) = M;
            *reinterpret_cast<float4*>(v + i0) = V;
            continue;
        }
        for (long long i = i0; i < i0 + 4 && i < n; i++) {
            float g;
            if (!f32_grad(fr, i, gacc, g)) {
                g = (float)gacc[i];
                gacc[i] = 0.0;
            }
            float w = p[i], mi = m[i], vi = v[i];
            adamw_one(g, w, mi, vi, c);
            p[i] = w, m[i] = mi, v[i] = vi;
        }
    }
}

__global__ void __launch_bounds__(256) k_f32_to_acc(long long n, const float* __restrict__ g32, double* __restrict__ gacc) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) gacc[i] = (double)g32[i];
}

__global__ void __launch_bounds__(256) k_acc_to_f32(long long n, const double* __restrict__ gacc, float* __restrict__ g32,
                                                    double scale, F32Ranges fr) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float g;
        g32[i] = f32_grad(fr, i, gacc, g) ? (float)((double)g * scale) : (float)(gacc[i] * scale);
    }
}

}  // namespace
}  // namespace unet
