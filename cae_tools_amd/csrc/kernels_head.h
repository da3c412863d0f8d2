// kernels_head.h — the narrow middle of the ConvAE step as two launches instead of thirteen.
//
// At the benchmark geometry (16x16 -> 256x256, batch 64) the encoder (encoder.py:40-46: 1->2->4 channels, 7x7 and
// 3x3 maps), the four Linear layers (encoder.py:54-58, decoder.py:31-35: 36->128->32->128->576) and their backward
// passes are ~0.2 % of the step's arithmetic but were 6 + 7 of its 26 launches, each a 5-12 us link in a serial chain.
// All of it fits in one CU's LDS, so:
//
//   k_head_fwd   every workgroup recomputes the WHOLE encoder for the whole batch in LDS (BatchNorm needs batch-wide
//                sums; ~0.1 MFLOP), then Linear 0..2 for its 16 batch rows, then its strip of Linear 3's columns.
//                Workgroup (0,0) is the one that publishes the encoder's raw outputs, the saved / running BatchNorm
//                statistics and the optimiser step bump; workgroups (*,0) publish the Linear activations.
//   k_tail_bwd   Linear 2..0 input gradients for the whole batch (redundantly in every workgroup), the three weight
//                gradients split over workgroups 1.., and in workgroup 0 the encoder's backward pass.
//
// Same arithmetic as the kernels they replace (kernels_generic.h k_down / k_up / k_wgrad, kernels_gemm.h k_gemm16):
// fp32 fmaf chains in the same order for the convolutions, v_mfma_f32_16x16x4_f32 for the Linear layers, fp64
// BatchNorm sums (here reduced in a fixed order, so every workgroup derives bit-identical statistics).
#pragma once
#include <type_traits>

#include "kernels_gemm.h"

namespace cae {

constexpr int kHeadThreads = 1024;
constexpr int kHeadWaves = kHeadThreads / 64;
constexpr int kHeadMaxEnc = 4;
constexpr int kHeadMaxC = 64;   // encoder channels the LDS tables are sized for

// How a stage's tiles are shared by the 16 waves (stage_split): 2^lg waves split the k-steps of one tile, `per` k-steps
// each; 16 >> lg tiles are in flight per pass.
struct StageSplit {
    int lg, per;
};

constexpr int kHeadW4 = 8;      // float4s of Linear weights a thread carries from kernel start to the LDS copy
struct HeadW {
    const float* src;   // row-major [rows][4 * n4row]
    int start4, n4row, ldw, lds_off, strip_floats;   // strip_floats: src advance per blockIdx.y
};

constexpr int kHeadMaxSeg = 12;
struct HeadSeg {
    const float* src;
    int count, lds_off, strip;   // strip: src advances by 16 * tiles_per_wg floats per blockIdx.y (Linear 3's bias)
};

struct HeadConv {
    int cin, hin, win, cout, hout, wout, kh, kw, s;
    const float *w, *bias, *gamma, *beta;
    float *rmean, *rvar, *saved;
    float* y;           // raw conv output [B][cout][hout][wout] (kept for the backward pass)
    float* g;           // tail: masked gradient wrt the BatchNorm output, same shape
    double *w_acc, *gamma_acc, *beta_acc;   // tail: fp64 gradient slots
    double inv_count, unbias;   // 1 / (B * hout * wout) and count / (count - 1): BatchNorm statistics without fp64 divisions
    int o_y, o_c;       // LDS offsets (floats): output map; float4 BatchNorm constants
    int o_w, o_b, o_gamma, o_beta, o_rm, o_rv;   // LDS copies of the parameters (two segments: parameter block, running stats)
};

struct HeadFc {
    int nin, nout, relu;
    const float *w, *bias;
    float* act;         // [B][nout] output activation
    float* grad;        // tail: gradient wrt the pre-activation of the output
    double *w_acc, *b_acc;
};

struct HeadArgs {
    int B, n_enc, train, bump_adam;
    HeadConv enc[kHeadMaxEnc];
    HeadFc fc[4];
    const float* x;     // data set (gathered through perm / the cursor) or an explicit batch
    const int* perm;
    int use_cursor;
    float momentum, eps;
    const StepState* st;
    int tiles_per_wg;   // head: Linear-3 column tiles (16 wide) per workgroup
    int n_wg;           // tail: workgroups in the launch
    int o_h[2], ld_h;   // LDS: two [rows][ld_h] activation / gradient panels
    int o_part, o_red, o_perm;
    int o_x;            // head: the gathered input batch [B][cin][hin][win]
    int o_bias[4];      // head: Linear biases (Linear 3: this workgroup's strip)
    StageSplit fc_split[4];
    HeadW wmat[4];      // head: Linear weights (Linear 3: this workgroup's strip of rows) copied to LDS after the encoder
    int w_total4;       // float4s over the four matrices
    int n_seg;          // head: small read-only vectors copied to LDS in one burst (<= 1024 floats each)
    HeadSeg seg[kHeadMaxSeg];
    long long* dbg;     // diagnostics (tools/head_phases.py): per workgroup 16 wall-clock stamps, or nullptr
};

// C[m][n] = sum_k A(m,k) * B(k,n) over the 16x16 tiles (tm0.., tn0..) x (TM, TN), k in [0,K): the workgroup's 16 waves
// share the tiles; with fewer tiles than waves the waves split K and the partial tiles are summed through LDS in a fixed
// order.  fa / fb fetch one operand element (they are only called for in-range indices), fe(m, n, value) consumes C.
// All threads of the workgroup must call mfma_stage (it synchronises).
// n / d for 0 <= n < 2^22 and d < 8000 with inv_d = 1.0f / d: (n + 0.5) / d is at least 0.5 / d away from an integer,
// far more than the rounding error of the fp32 product (3 instructions instead of the ~40 of an integer division)
__device__ __forceinline__ int div_small(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

inline StageSplit stage_split(int tiles, int K) {
    const int S = (K + 3) / 4;
    int lg = 0;
    while ((2 << lg) * tiles <= kHeadWaves && (2 << lg) <= S) lg++;
    StageSplit r;
    r.lg = lg;
    r.per = (S + (1 << lg) - 1) >> lg;
    return r;
}
struct StageCoords {
    int split, group, per, tl, kc, S;
};
__device__ __forceinline__ StageCoords stage_coords(StageSplit sp, int K) {
    StageCoords c;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.S = (K + 3) >> 2;
    c.split = 1 << sp.lg;
    c.group = kHeadWaves >> sp.lg;
    c.per = sp.per;
    c.tl = wv >> sp.lg;
    c.kc = wv & (c.split - 1);
    return c;
}

template <bool APAD, bool BFAST, class FA, class FB, class FE>
__device__ __forceinline__ void mfma_stage(int M, int N, int K, int tm0, int TM, int tn0, int TN, StageSplit sp, float* part,
                                           FA fa, FB fb, FE fe) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int tiles = TM * TN;
    const StageCoords c = stage_coords(sp, K);
    const float inv_tn = 1.0f / (float)TN;
    for (int p0 = 0; p0 < tiles; p0 += c.group) {
        const int t = p0 + c.tl;
        if (t < tiles) {
            const int tmr = TM == 1 ? 0 : div_small(t, inv_tn);
            const int tm = tm0 + tmr, tn = tn0 + t - tmr * TN;
            const int am = tm * 16 + r, bn = tn * 16 + r;
            const bool am_ok = am < M, bn_ok = bn < N;
            const int s0 = c.kc * c.per, s1 = min(c.S, s0 + c.per);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int st = s0; st < s1; st += 8) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = (st + u) * 4 + q;
                    const bool ok = (st + u) < s1 && k < K;
                    // APAD: the A panel is zero-padded to whole tiles and k-batches (and readable a k-batch past the end)
                    a[u] = APAD ? fa(am, k) : ((ok && am_ok) ? fa(am, k) : 0.f);
                    // BFAST: B is an LDS image with K a multiple of 4, readable past its rows; only the wave's own k-range
                    // is masked (columns >= N come out as garbage and are dropped by the epilogue's n < N)
                    if (BFAST) {
                        const float t = fb(k, bn);
                        b[u] = (st + u) < s1 ? t : 0.f;
                    } else {
                        b[u] = (ok && bn_ok) ? fb(k, bn) : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) part[wv * 256 + j * 64 + lane] = acc[j];
        }
        __syncthreads();
        const int n_pass = min(c.group, tiles - p0);
        for (int e = threadIdx.x; e < n_pass * 256; e += kHeadThreads) {
            const int tl2 = e >> 8, idx = e & 255, j = idx >> 6, ln = idx & 63;
            float v = 0.f;
            for (int i = 0; i < c.split; i++) v += part[((tl2 << sp.lg) + i) * 256 + idx];
            const int t2 = p0 + tl2;
            const int tmr = TM == 1 ? 0 : div_small(t2, inv_tn);
            const int m = (tm0 + tmr) * 16 + 4 * (ln >> 4) + j, n = (tn0 + t2 - tmr * TN) * 16 + (ln & 15);
            if (m < M && n < N) fe(m, n, v);
        }
        __syncthreads();
    }
}

// sum over taps of in[ky][kx] * w[ky][kx] added to acc (ky outer, as k_down); KS > 0: square kernel known at compile time
template <int KS>
__device__ __forceinline__ float head_taps(const float* __restrict__ in, int row_stride, const float* __restrict__ w, int kh,
                                           int kw, float acc) {
    if constexpr (KS > 0) {
        float v[KS * KS];
#pragma unroll
        for (int ky = 0; ky < KS; ky++)
#pragma unroll
            for (int kx = 0; kx < KS; kx++) v[ky * KS + kx] = in[ky * row_stride + kx];
#pragma unroll
        for (int i = 0; i < KS * KS; i++) acc = fmaf(v[i], w[i], acc);
    } else {
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++) acc = fmaf(in[ky * row_stride + kx], w[ky * kw + kx], acc);
    }
    return acc;
}

// fp64 wave sum on the VALU: both halves of the double travel by DPP (the ds_bpermute butterfly costs ~100 cycles a step)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_d(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// valid in lane 63
__device__ __forceinline__ double head_wave_sum(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    v += dpp_d<0x142, 0xA>(v);
    v += dpp_d<0x143, 0xC>(v);
    return v;
}

// Everything small the kernel reads from global memory (encoder parameters and running statistics, Linear biases), copied
// to LDS in ONE burst of loads: one element per thread per segment, all loads issued before the first store.  The sample
// gather (cursor -> permutation -> input rows: three dependent hops) is started first and lands in LDS last.
__device__ __forceinline__ void head_stage_inputs(const HeadArgs& a, float* lds) {
    const int tid = threadIdx.x;
    int* sample = reinterpret_cast<int*>(lds + a.o_perm);
    float v[kHeadMaxSeg];
#pragma unroll
    for (int g = 0; g < kHeadMaxSeg; g++) {
        v[g] = 0.f;
        if (g < a.n_seg) {
            const float* src = a.seg[g].src + (a.seg[g].strip ? blockIdx.y * 16 * a.tiles_per_wg : 0);
            v[g] = src[min(tid, a.seg[g].count - 1)];
        }
    }
    int smp = 0;
    if (tid < a.B) smp = (int)sample_of(a.perm, a.use_cursor, a.st, tid);
    for (int b = tid + kHeadThreads; b < a.B; b += kHeadThreads) sample[b] = (int)sample_of(a.perm, a.use_cursor, a.st, b);
    if (tid < a.B) sample[tid] = smp;
#pragma unroll
    for (int g = 0; g < kHeadMaxSeg; g++)
        if (g < a.n_seg && tid < a.seg[g].count) lds[a.seg[g].lds_off + tid] = v[g];
    __syncthreads();
    const HeadConv& L0 = a.enc[0];
    const int per = L0.cin * L0.hin * L0.win;
    float* xs = lds + a.o_x;
    if ((per & 3) == 0) {
        const int n4 = per >> 2, total = a.B * n4;
        const float inv_n4 = 1.0f / (float)n4;
        f32x4* xs4 = reinterpret_cast<f32x4*>(xs);
        auto fetch = [&](int i) {   // clamped: unconditional loads, so all four are in flight together
            i = min(i, total - 1);
            const int b = div_small(i, inv_n4);
            return reinterpret_cast<const f32x4*>(a.x + (size_t)sample[b] * per)[i - b * n4];
        };
        for (int i0 = tid; i0 < total; i0 += 4 * kHeadThreads) {
            f32x4 v0 = fetch(i0), v1 = fetch(i0 + kHeadThreads), v2 = fetch(i0 + 2 * kHeadThreads), v3 = fetch(i0 + 3 * kHeadThreads);
            // keeps the compiler from sinking the loads into the conditional stores below
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            xs4[i0] = v0;
            if (i0 + kHeadThreads < total) xs4[i0 + kHeadThreads] = v1;
            if (i0 + 2 * kHeadThreads < total) xs4[i0 + 2 * kHeadThreads] = v2;
            if (i0 + 3 * kHeadThreads < total) xs4[i0 + 3 * kHeadThreads] = v3;
        }
    } else {
        for (int i = tid; i < a.B * per; i += kHeadThreads) {
            const int b = i / per;
            xs[i] = a.x[(size_t)sample[b] * per + (i - b * per)];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void head_stamp(const HeadArgs& a, int phase) {
    if (a.dbg && threadIdx.x == 0) a.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + phase] = wall_clock64();
}

// The encoder for the whole batch, in LDS: raw outputs at o_y, BatchNorm constants {mean, gamma*invstd, beta, invstd} at
// o_c.  `publish`: this workgroup writes the raw outputs, saved and running statistics to global memory.
__device__ __forceinline__ void head_encoder(const HeadArgs& a, float* lds, bool publish) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int B = a.B;
    double* red = reinterpret_cast<double*>(lds + a.o_red);   // [channels][kHeadWaves][2]
    for (int l = 0; l < a.n_enc; l++) {
        const HeadConv& L = a.enc[l];
        const float* wl = lds + L.o_w;
        const int taps = L.kh * L.kw;
        const float* yin = l ? lds + a.enc[l - 1].o_y : lds + a.o_x;
        float* yout = lds + L.o_y;
        const int hw = L.hout * L.wout, npos = B * hw, ihw = L.hin * L.win;
        const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)L.wout;
        const bool k3 = L.kh == 3 && L.kw == 3;
        const float4* kin = l ? reinterpret_cast<const float4*>(lds + a.enc[l - 1].o_c) : nullptr;
        // four output channels at a time share the position arithmetic and the (BatchNorm + ReLU'd) input taps; the
        // weights are wave-uniform reads of the parameter arena (scalar loads)
        for (int c0 = 0; c0 < L.cout; c0 += 4) {
            int cj[4];
#pragma unroll
            for (int j = 0; j < 4; j++) cj[j] = min(c0 + j, L.cout - 1);
            double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
            for (int idx = tid; idx < npos; idx += kHeadThreads) {
                const int b = div_small(idx, inv_hw), r = idx - b * hw;
                const int y = div_small(r, inv_w), x = r - y * L.wout;
                const int o_in = (y * L.s) * L.win + x * L.s;
                float acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = L.bias[cj[j]];
                for (int cl = 0; cl < L.cin; cl++) {
                    const float* ip = yin + (b * L.cin + cl) * ihw + o_in;
                    const float4 kb = l ? kin[cl] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k3) {
                        float v[9];
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) v[ky * 3 + kx] = ip[ky * L.win + kx];
                        if (l) {
#pragma unroll
                            for (int i = 0; i < 9; i++) v[i] = fmaxf(0.f, fmaf(v[i] - kb.x, kb.y, kb.z));
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float* __restrict__ wp = L.w + (cj[j] * L.cin + cl) * 9;
#pragma unroll
                            for (int i = 0; i < 9; i++) acc[j] = fmaf(v[i], wp[i], acc[j]);
                        }
                    } else {
                        for (int ky = 0; ky < L.kh; ky++)
                            for (int kx = 0; kx < L.kw; kx++) {
                                float v = ip[ky * L.win + kx];
                                if (l) v = fmaxf(0.f, fmaf(v - kb.x, kb.y, kb.z));
#pragma unroll
                                for (int j = 0; j < 4; j++) acc[j] = fmaf(v, L.w[(cj[j] * L.cin + cl) * taps + ky * L.kw + kx], acc[j]);
                            }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c0 + j < L.cout) {
                        yout[(b * L.cout + c0 + j) * hw + r] = acc[j];
                        s1[j] += (double)acc[j];
                        s2[j] += (double)acc[j] * (double)acc[j];
                    }
            }
            if (a.train) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c0 + j < L.cout) {
                        const double t1 = head_wave_sum(s1[j]), t2 = head_wave_sum(s2[j]);
                        if (lane == 63) {
                            red[((c0 + j) * kHeadWaves + wv) * 2] = t1;
                            red[((c0 + j) * kHeadWaves + wv) * 2 + 1] = t2;
                        }
                    }
            }
        }
        if (l < 2) head_stamp(a, 9 + 3 * l);
        __syncthreads();
        float4* kout = reinterpret_cast<float4*>(lds + L.o_c);
        if (tid < L.cout) {
            const int c = tid;
            const float gamma = lds[L.o_gamma + c], beta = lds[L.o_beta + c], rm = lds[L.o_rm + c], rv = lds[L.o_rv + c];
            float mean, invstd;
            if (a.train) {
                double t1 = 0.0, t2 = 0.0;
                for (int w = 0; w < kHeadWaves; w++) {
                    t1 += red[(c * kHeadWaves + w) * 2];
                    t2 += red[(c * kHeadWaves + w) * 2 + 1];
                }
                const double inv_count = L.inv_count;   // host-computed 1 / (B * hout * wout): no fp64 divisions here
                const double m = t1 * inv_count;
                double var = t2 * inv_count - m * m;
                var = var < 0.0 ? 0.0 : var;
                mean = (float)m;
                invstd = 1.0f / sqrtf((float)(var + (double)a.eps));
                if (publish) {
                    L.saved[2 * c] = mean;
                    L.saved[2 * c + 1] = invstd;
                    const double unb = var * L.unbias;   // count / (count - 1), or 1 for a single element
                    L.rmean[c] = (1.f - a.momentum) * rm + a.momentum * mean;
                    L.rvar[c] = (1.f - a.momentum) * rv + a.momentum * (float)unb;
                }
            } else {
                mean = rm;
                invstd = 1.0f / sqrtf(rv + a.eps);
            }
            kout[c] = make_float4(mean, gamma * invstd, beta, invstd);
        }
        if (publish && a.train)
            for (int i = tid; i < npos * L.cout; i += kHeadThreads) L.y[i] = yout[i];
        __syncthreads();
        if (l < 2) head_stamp(a, 10 + 3 * l);
    }
}


// grid (ceil(B/16), ceil(ceil(fc[3].nout/16) / tiles_per_wg)), block 1024, dynamic LDS laid out by the host
__global__ void __launch_bounds__(kHeadThreads) k_head_fwd(HeadArgs a) {
    extern __shared__ double lds_d[];
    float* lds = reinterpret_cast<float*>(lds_d);
    const int tid = threadIdx.x;
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    // no kernel of the forward/backward pass reads adam_step, so bumping it here cannot race
    if (first && tid == 0 && a.bump_adam) const_cast<StepState*>(a.st)->adam_step += 1;

    head_stamp(a, 0);
    const int T3 = (a.fc[3].nout + 15) >> 4;
    const int tn3 = blockIdx.y * a.tiles_per_wg;
    const int TN3 = min(a.tiles_per_wg, T3 - tn3);
    // The Linear weights, as coalesced 16-byte loads that stay in flight while the encoder runs (fetching them in the MFMA
    // operand layout instead costs 16 cache lines per wave instruction: measured 10 us for the four layers).
    f32x4 wreg[kHeadW4];
#pragma unroll
    for (int j = 0; j < kHeadW4; j++) {
        const int idx = min(tid + j * kHeadThreads, a.w_total4 - 1);
        const int m = (idx >= a.wmat[1].start4) + (idx >= a.wmat[2].start4) + (idx >= a.wmat[3].start4);
        const float* src = a.wmat[m].src + (size_t)blockIdx.y * a.wmat[m].strip_floats;
        wreg[j] = reinterpret_cast<const f32x4*>(src)[idx - a.wmat[m].start4];
    }
    for (int i = tid; i < 32 * a.ld_h + 32; i += kHeadThreads) lds[a.o_h[0] + i] = 0.f;   // both panels (adjacent), padding and guard included
    head_stamp(a, 1);
    head_stage_inputs(a, lds);
    head_stamp(a, 2);
    head_encoder(a, lds, first);
    head_stamp(a, 3);

    const int row0 = blockIdx.x * 16;
    const int rows = min(16, a.B - row0);
    const int ld = a.ld_h;
    float* part = lds + a.o_part;
    float* hin = lds + a.o_h[0];
    float* hout = lds + a.o_h[1];
    {   // the flattened encoder output through BatchNorm + ReLU (encoder.py:46-47, :52)
        const HeadConv& P = a.enc[a.n_enc - 1];
        const float4* kp = reinterpret_cast<const float4*>(lds + P.o_c);
        const float* yp = lds + P.o_y;
        const int hw = P.hout * P.wout, nin = a.fc[0].nin;
        const float inv_nin = 1.0f / (float)nin, inv_hw = 1.0f / (float)hw;
        for (int e = tid; e < rows * nin; e += kHeadThreads) {
            const int m = div_small(e, inv_nin), k = e - m * nin;
            const float4 c4 = kp[div_small(k, inv_hw)];
            hin[m * ld + k] = fmaxf(0.f, fmaf(yp[(row0 + m) * nin + k] - c4.x, c4.y, c4.z));
        }
    }
    // the encoder's input and inner maps are dead: the weights take their place
#pragma unroll
    for (int j = 0; j < kHeadW4; j++) {
        const int idx = tid + j * kHeadThreads;
        if (idx < a.w_total4) {
            const int m = (idx >= a.wmat[1].start4) + (idx >= a.wmat[2].start4) + (idx >= a.wmat[3].start4);
            const int local = idx - a.wmat[m].start4;
            const int row = div_small(local, 1.0f / (float)a.wmat[m].n4row);
            *reinterpret_cast<f32x4*>(lds + a.wmat[m].lds_off + row * a.wmat[m].ldw + 4 * (local - row * a.wmat[m].n4row)) = wreg[j];
        }
    }
    __syncthreads();
    head_stamp(a, 4);
    const bool store_h = blockIdx.y == 0 && a.train;
    auto run_fc = [&](auto I) {
        constexpr int i = decltype(I)::value;
        const HeadFc& F = a.fc[i];
        const float* W = lds + a.wmat[i].lds_off;
        const int ldw = a.wmat[i].ldw;
        const float* bias = lds + a.o_bias[i];
        float* act = F.act;
        const int nin = F.nin, nout = F.nout, relu = F.relu;
        const int tn0 = i == 3 ? tn3 : 0, TN = i == 3 ? TN3 : (nout + 15) >> 4;
        const int nb0 = i == 3 ? tn3 * 16 : 0;
        const bool to_lds = i < 3, to_mem = i == 3 || store_h;
        float* hi = hin;
        float* ho = hout;
        mfma_stage<true, true>(rows, nout, nin, 0, 1, tn0, TN, a.fc_split[i], part,
                         [=](int m, int k) { return hi[m * ld + k]; },
                         [=](int k, int n) { return W[(n - nb0) * ldw + k]; },
                         [=](int m, int n, float v) {
                             v += bias[n - nb0];
                             if (relu) v = fmaxf(v, 0.f);
                             if (to_lds) ho[m * ld + n] = v;
                             if (to_mem) act[(size_t)(row0 + m) * nout + n] = v;
                         });
        hin = ho;
        hout = hi;
        head_stamp(a, 5 + i);
    };
    run_fc(std::integral_constant<int, 0>{});
    run_fc(std::integral_constant<int, 1>{});
    run_fc(std::integral_constant<int, 2>{});
    run_fc(std::integral_constant<int, 3>{});
}

}  // namespace cae
