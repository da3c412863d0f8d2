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
//   k_tail_bwd   the backward pass of Linear 2..0: workgroups = (16-row groups of the batch) x (4 tasks); each walks the
//                input-gradient chain for its rows as far as its task needs, then adds its rows' share of one weight
//                gradient, or (task 0) writes the masked gradient for the encoder and its BatchNorm sums.
//
// Measured rules these follow (tools/head_phases.py prints per-phase wall-clock stamps of both kernels): every global read
// is issued in one burst at kernel start; weights travel as coalesced 16-byte loads and reach the MFMA operand layout through
// LDS (gathering them from global memory in that layout costs 16 cache lines per wave instruction); a phase on one CU is
// bound by latency and instruction issue, not FLOPs, so work is spread over workgroups where no batch-wide sum forbids it.
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
constexpr int kHeadMaxC = 64;   // encoder channels (one thread per channel derives the BatchNorm constants)

// How a stage's tiles are shared by the 16 waves (stage_split): 2^lg waves split the k-steps of one tile, `per` k-steps
// each; 16 >> lg tiles are in flight per pass.
struct StageSplit {
    int lg, per;
};

constexpr int kHeadEarly = 4;   // of which requested at kernel start (k_head_fwd); 5: +0.4 us per step, 6: spills
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
    int o_h[2], ld_h;   // LDS: two [rows][ld_h] activation / gradient panels
    int o_part, o_red, o_perm;
    int o_x;            // head: the gathered input batch [B][cin][hin][win]
    int o_bias[4];      // head: Linear biases (Linear 3: this workgroup's strip)
    StageSplit fc_split[4];
    HeadW wmat[4];      // head: Linear weights (Linear 3: this workgroup's strip of rows) copied to LDS after the encoder
    int w_total4;       // float4s over the four matrices
    int piece_m[kHeadW4];   // matrix that holds ALL of piece j (float4s [j, j + 1) * kHeadThreads), or -1 when it straddles two
    int n_seg;          // head: small read-only vectors copied to LDS in one burst (<= 1024 floats each)
    HeadSeg seg[kHeadMaxSeg];
    float* xbatch;      // head, training: the first workgroup leaves the staged input batch here, contiguous (k_adam's fused conv-0 weight gradient)
    double* clear0;     // ... and clears these clear0_n doubles (the first encoder layer's BatchNorm sum table, which k_adam then leaves alone)
    int clear0_n;
    long long* dbg;     // diagnostics (tools/head_phases.py): per workgroup 16 wall-clock stamps, or nullptr
};

// C[m][n] = sum_k A(m,k) * B(k,n) over the 16x16 tiles (tm0.., tn0..) x (TM, TN), k in [0,K): the workgroup's 16 waves
// share the tiles; with fewer tiles than waves the waves split K and the partial tiles are summed through LDS in a fixed
// order.  fa / fb fetch one operand element (they are only called for in-range indices), fe(m, n, value) consumes C.
// All threads of the workgroup must call mfma_stage (it synchronises).
inline StageSplit stage_split(int tiles, int K) {
    const int S = (K + 3) / 4;
    int lg = 0;
    // splitting K costs an LDS round trip of the partial tiles and two barriers (~700 cycles, about 20 MFMA k-steps): only
    // worth it for long contractions on few tiles; otherwise each wave keeps a whole tile and consumes it from registers
    if (S >= 24 && tiles <= 4)
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

template <bool APAD, bool BFAST, int DEPTH, class FA, class FB, class FE>
__device__ __forceinline__ void mfma_stage_d(int M, int N, int K, int tm0, int TM, int tn0, int TN, StageSplit sp, float* part,
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
            for (int st = s0; st < s1; st += DEPTH) {
                float a[DEPTH], b[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
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
                for (int u = 0; u < DEPTH; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
            if (sp.lg == 0) {
                // the wave owns the whole K range of its tile: consume it straight from the accumulator registers
                // (C/D layout of the 16x16 MFMA: register j of lane l is C[row 4*(l>>4) + j][col l&15])
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int m = tm * 16 + 4 * q + j, n = tn * 16 + r;
                    if (m < M && n < N) fe(m, n, acc[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) part[wv * 256 + j * 64 + lane] = acc[j];
            }
        }
        if (sp.lg == 0) continue;
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
    if (sp.lg == 0) __syncthreads();   // what fe wrote to LDS is visible to the next stage
}

// k-batches of 8 MFMAs, or of 4 where a wave's share of K is that short (half the operand fetches and MFMAs would be padding)
template <bool APAD, bool BFAST, class FA, class FB, class FE>
__device__ __forceinline__ void mfma_stage(int M, int N, int K, int tm0, int TM, int tn0, int TN, StageSplit sp, float* part,
                                           FA fa, FB fb, FE fe) {
    if (sp.per <= 4) mfma_stage_d<APAD, BFAST, 4>(M, N, K, tm0, TM, tn0, TN, sp, part, fa, fb, fe);
    else mfma_stage_d<APAD, BFAST, 8>(M, N, K, tm0, TM, tn0, TN, sp, part, fa, fb, fe);
}

// fp64 wave sum, valid in lane 63 (kernels_generic.h)
__device__ __forceinline__ double head_wave_sum(double v) { return wave_sum_lane63(v); }

// Everything small the kernel reads from global memory (encoder parameters and running statistics, Linear biases), copied
// to LDS in ONE burst of loads: one element per thread per segment, all loads issued before the first store.  The sample
// gather (cursor -> permutation -> input rows: three dependent hops) is started first and lands in LDS last.
// `bs`: the cursor (StepState.batch_start), requested by the caller at the top of the kernel so that its round trip hides
// behind the weight loads' issue.  Without a permutation (batches laid out contiguously once: ConvAEModel.train, bench.py) the
// batch is ONE contiguous block starting at row bs: its loads are issued first and need no sample table.
__device__ __forceinline__ void head_stage_inputs(const HeadArgs& a, float* lds, long long bs) {
    const int tid = threadIdx.x;
    int* sample = reinterpret_cast<int*>(lds + a.o_perm);
    const HeadConv& L0 = a.enc[0];
    const int per = L0.cin * L0.hin * L0.win;
    float* xs = lds + a.o_x;
    const bool direct = a.perm == nullptr && (per & 3) == 0 && a.B * (per >> 2) <= 4 * kHeadThreads;
    f32x4 xv[4];
    if (direct) {   // the first use of the cursor: a counted wait (it was the first load), the Linear weights stay in flight
        const int total = a.B * (per >> 2);
        const f32x4* src = reinterpret_cast<const f32x4*>(a.x + (size_t)bs * per);
#pragma unroll
        for (int u = 0; u < 4; u++) xv[u] = src[min(tid + u * kHeadThreads, total - 1)];
    }
    float v[kHeadMaxSeg];
#pragma unroll
    for (int g = 0; g < kHeadMaxSeg; g++) {
        v[g] = 0.f;
        if (g < a.n_seg) {
            const float* src = a.seg[g].src + (a.seg[g].strip ? blockIdx.y * 16 * a.tiles_per_wg : 0);
            v[g] = src[min(tid, a.seg[g].count - 1)];
        }
    }
    if (!direct) {
        int smp = 0;
        if (tid < a.B) smp = a.perm ? a.perm[bs + tid] : (int)(bs + tid);
        for (int b = tid + kHeadThreads; b < a.B; b += kHeadThreads) sample[b] = a.perm ? a.perm[bs + b] : (int)(bs + b);
        if (tid < a.B) sample[tid] = smp;
    }
#pragma unroll
    for (int g = 0; g < kHeadMaxSeg; g++)
        if (g < a.n_seg && tid < a.seg[g].count) lds[a.seg[g].lds_off + tid] = v[g];
    if (direct) {
        const int total = a.B * (per >> 2);
        f32x4* xs4 = reinterpret_cast<f32x4*>(xs);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (tid + u * kHeadThreads < total) xs4[tid + u * kHeadThreads] = xv[u];
        __syncthreads();
        return;
    }
    __syncthreads();
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
        // weights and biases are broadcast reads of their LDS copy (head_stage_inputs).  Read from the parameter arena they
        // are vector loads - the compiler cannot prove the arena unchanged across this kernel's stores, so no scalar loads -
        // each waited for with vmcnt(0) inside the channel loop: a trip to L2 per iteration, and the first one drains the
        // Linear weights still in flight.  (The phase itself is bound by instruction issue - one workgroup computes the whole
        // batch on one CU - and did not get shorter: 8.0 -> 8.7 us with the weight loads no longer waited for ahead of it.)
        const float* bl = lds + L.o_b;
        for (int c0 = 0; c0 < L.cout; c0 += 4) {
            int cj[4];
#pragma unroll
            for (int j = 0; j < 4; j++) cj[j] = min(c0 + j, L.cout - 1);
            double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
            if (k3 && L.cin == 1) {
                // the first layer (one input channel, several positions per thread): the 36 weights of the four output
                // channels are read once, ahead of the position loop - per position their LDS reads outnumbered the taps'
                // four to one - and the sums stay in fp32 for four positions at a time before they are folded into the
                // fp64 running sums (this phase is bound by instruction issue: one workgroup computes the whole batch)
                float w[4][9];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 9; i++) w[j][i] = wl[cj[j] * 9 + i];
                float bj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) bj[j] = bl[cj[j]];
                float p1[4] = {0.f, 0.f, 0.f, 0.f}, p2[4] = {0.f, 0.f, 0.f, 0.f};
                int pass = 0;
                for (int idx = tid; idx < npos; idx += kHeadThreads, pass++) {
                    const int b = div_small(idx, inv_hw), r = idx - b * hw;
                    const int y = div_small(r, inv_w), x = r - y * L.wout;
                    const float* ip = yin + b * ihw + (y * L.s) * L.win + x * L.s;
                    float v[9];
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) v[ky * 3 + kx] = ip[ky * L.win + kx];
                    if (l) {
                        const float4 kb = kin[0];
#pragma unroll
                        for (int i = 0; i < 9; i++) v[i] = fmaxf(0.f, fmaf(v[i] - kb.x, kb.y, kb.z));
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float acc = bj[j];
#pragma unroll
                        for (int i = 0; i < 9; i++) acc = fmaf(v[i], w[j][i], acc);
                        if (c0 + j < L.cout) {
                            yout[(b * L.cout + c0 + j) * hw + r] = acc;
                            p1[j] += acc;
                            p2[j] = fmaf(acc, acc, p2[j]);
                        }
                    }
                    if ((pass & 3) == 3) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            s1[j] += (double)p1[j];
                            s2[j] += (double)p2[j];
                            p1[j] = p2[j] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    s1[j] += (double)p1[j];
                    s2[j] += (double)p2[j];
                }
            } else
            for (int idx = tid; idx < npos; idx += kHeadThreads) {
                const int b = div_small(idx, inv_hw), r = idx - b * hw;
                const int y = div_small(r, inv_w), x = r - y * L.wout;
                const int o_in = (y * L.s) * L.win + x * L.s;
                float acc[4];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = bl[cj[j]];
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
                            const float* wp = wl + (cj[j] * L.cin + cl) * 9;
#pragma unroll
                            for (int i = 0; i < 9; i++) acc[j] = fmaf(v[i], wp[i], acc[j]);
                        }
                    } else {
                        for (int ky = 0; ky < L.kh; ky++)
                            for (int kx = 0; kx < L.kw; kx++) {
                                float v = ip[ky * L.win + kx];
                                if (l) v = fmaxf(0.f, fmaf(v - kb.x, kb.y, kb.z));
#pragma unroll
                                for (int j = 0; j < 4; j++) acc[j] = fmaf(v, wl[(cj[j] * L.cin + cl) * taps + ky * L.kw + kx], acc[j]);
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
        // a row of 16 lanes per channel: each lane takes one wave's partial sums, the row adds them up with DPP (one thread
        // per channel walking the 16 partials was 32 dependent LDS reads: a microsecond per layer)
        static_assert(kHeadWaves == 16, "one DPP row per channel");
        if (tid < L.cout * 16) {
            const int c = tid >> 4;
            const float gamma = lds[L.o_gamma + c], beta = lds[L.o_beta + c], rm = lds[L.o_rm + c], rv = lds[L.o_rv + c];
            float mean, invstd;
            if (a.train) {
                double t1 = red[tid * 2], t2 = red[tid * 2 + 1];   // [(c * kHeadWaves + wave) * 2]
                t1 += dpp_d<0xB1>(t1);  t2 += dpp_d<0xB1>(t2);
                t1 += dpp_d<0x4E>(t1);  t2 += dpp_d<0x4E>(t2);
                t1 += dpp_d<0x141>(t1); t2 += dpp_d<0x141>(t2);
                t1 += dpp_d<0x140>(t1); t2 += dpp_d<0x140>(t2);   // every lane of the row holds the row's sum
                const double inv_count = L.inv_count;   // host-computed 1 / (B * hout * wout): no fp64 divisions here
                const double m = t1 * inv_count;
                double var = t2 * inv_count - m * m;
                var = var < 0.0 ? 0.0 : var;
                mean = (float)m;
                invstd = 1.0f / sqrtf((float)(var + (double)a.eps));
                if (publish && (tid & 15) == 0) {
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
            if ((tid & 15) == 0) kout[c] = make_float4(mean, gamma * invstd, beta, invstd);
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
    kernarg_warm<sizeof(HeadArgs)>();
    const bool first = blockIdx.x == 0 && blockIdx.y == 0;
    // no kernel of the forward/backward pass reads adam_step, so bumping it here cannot race
    if (first && tid == 0 && a.bump_adam) const_cast<StepState*>(a.st)->adam_step += 1;

    head_stamp(a, 0);
    // The cursor first: the input batch hangs off it (head_stage_inputs).  Read through an index the compiler cannot see is
    // zero: a load it knows to be wave-uniform is moved to a scalar register at once, i.e. waited for here, with nothing
    // else in flight yet.
    long long bs = 0;
    if (a.perm || a.use_cursor) {
        int z = 0;
        asm volatile("" : "+v"(z));
        bs = (&a.st->batch_start)[z];
    }
    const int T3 = (a.fc[3].nout + 15) >> 4;
    const int tn3 = blockIdx.y * a.tiles_per_wg;
    const int TN3 = min(a.tiles_per_wg, T3 - tn3);
    // The Linear weights, as coalesced 16-byte loads that stay in flight while the encoder runs (fetching them in the MFMA
    // operand layout instead costs 16 cache lines per wave instruction: measured 10 us for the four layers).
    // (the matrix a 16-byte piece belongs to is picked with selects between wave-uniform values: indexing a.wmat[] with a
    // per-lane index makes the compiler fetch the descriptor fields from the kernel argument segment with per-lane loads,
    // a dependent trip to memory ahead of every weight load)
    // Only the first kHeadEarly pieces per thread are carried across the encoder: with all of them the register allocator
    // (128 VGPRs at 1024 threads) spills two right where they are loaded, which waits for every load at the top of the
    // kernel.  The rest - the tail of the piece order, i.e. the last Linear layer's - is requested after the encoder and
    // lands behind the first Linear layer.
    const int st1 = a.wmat[1].start4, st2 = a.wmat[2].start4, st3 = a.wmat[3].start4;
    // (a piece that lies inside one matrix - the host says which - takes that matrix's descriptor through scalar registers;
    // only a straddling piece pays for the per-lane selects)
    auto load_piece = [&](int j) {
        const int idx = min(tid + j * kHeadThreads, a.w_total4 - 1);
        const int pm = a.piece_m[j];
        if (pm >= 0) {
            const HeadW& w = a.wmat[pm];
            return (reinterpret_cast<const f32x4*>(w.src + (size_t)blockIdx.y * w.strip_floats) - w.start4)[idx];
        }
        const f32x4* wsrc[4];
#pragma unroll
        for (int m = 0; m < 4; m++)
            wsrc[m] = reinterpret_cast<const f32x4*>(a.wmat[m].src + (size_t)blockIdx.y * a.wmat[m].strip_floats) - a.wmat[m].start4;
        const f32x4* src = idx >= st3 ? wsrc[3] : idx >= st2 ? wsrc[2] : idx >= st1 ? wsrc[1] : wsrc[0];
        return src[idx];
    };
    auto store_piece = [&](int j, const f32x4& val) {
        const int idx = tid + j * kHeadThreads;
        const int pm = a.piece_m[j];
        if (pm >= 0) {
            const HeadW& w = a.wmat[pm];
            if (idx < a.w_total4) {
                const int local = idx - w.start4;
                const int row = div_small(local, 1.0f / (float)w.n4row);
                *reinterpret_cast<f32x4*>(lds + w.lds_off + row * w.ldw + 4 * (local - row * w.n4row)) = val;
            }
            return;
        }
        if (idx < a.w_total4) {
            const bool m3 = idx >= st3, m2 = idx >= st2, m1 = idx >= st1;
            auto pick = [&](int v0, int v1, int v2, int v3) { return m3 ? v3 : m2 ? v2 : m1 ? v1 : v0; };
            const int local = idx - pick(0, st1, st2, st3);
            const int n4row = pick(a.wmat[0].n4row, a.wmat[1].n4row, a.wmat[2].n4row, a.wmat[3].n4row);
            const int ldw = pick(a.wmat[0].ldw, a.wmat[1].ldw, a.wmat[2].ldw, a.wmat[3].ldw);
            const int off = pick(a.wmat[0].lds_off, a.wmat[1].lds_off, a.wmat[2].lds_off, a.wmat[3].lds_off);
            const int row = div_small(local, 1.0f / (float)n4row);
            *reinterpret_cast<f32x4*>(lds + off + row * ldw + 4 * (local - row * n4row)) = val;
        }
    };
    f32x4 wreg[kHeadEarly];
#pragma unroll
    for (int j = 0; j < kHeadEarly; j++) wreg[j] = load_piece(j);
    for (int i = tid; i < 32 * a.ld_h + 32; i += kHeadThreads) lds[a.o_h[0] + i] = 0.f;   // both panels (adjacent), padding and guard included
    head_stamp(a, 1);
    head_stage_inputs(a, lds, bs);
    // (the LAST workgroup: the first one already publishes the encoder's outputs and statistics)
    if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) {   // uniform per workgroup
        if (a.xbatch) {
            const HeadConv& L0 = a.enc[0];
            const int nx = a.B * L0.cin * L0.hin * L0.win;
            for (int i = tid; i < nx; i += kHeadThreads) a.xbatch[i] = lds[a.o_x + i];
            // ... and the layer's BatchNorm gamma as it is NOW, behind the batch: the optimiser launch that takes this layer's
            // weight gradient rewrites gamma while its other workgroups still need the old value (BN-backward constants)
            if (tid < L0.cout) a.xbatch[nx + tid] = lds[L0.o_gamma + tid];
        }
        for (int i = tid; i < a.clear0_n; i += kHeadThreads) a.clear0[i] = 0.0;
    }
    head_stamp(a, 2);
    head_encoder(a, lds, first);
    f32x4 wlate[kHeadW4 - kHeadEarly];
    const bool late = a.w_total4 > kHeadEarly * kHeadThreads;   // uniform
    if (late) {
#pragma unroll
        for (int j = kHeadEarly; j < kHeadW4; j++) wlate[j - kHeadEarly] = load_piece(j);
    }
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
    for (int j = 0; j < kHeadEarly; j++) store_piece(j, wreg[j]);
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
    if (late) {   // their LDS rows are not read before the barriers of the next stage
#pragma unroll
        for (int j = kHeadEarly; j < kHeadW4; j++) store_piece(j, wlate[j - kHeadEarly]);
    }
    run_fc(std::integral_constant<int, 1>{});
    run_fc(std::integral_constant<int, 2>{});
    run_fc(std::integral_constant<int, 3>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// k_tail_bwd: the backward pass of Linear 2, 1, 0 in one launch (was three k_gemm16_pair launches).
//
// grid (row groups of 16 samples, 4 tasks).  Every workgroup walks the input-gradient chain for ITS 16 rows as far as its
// task needs (a link is ~1.5 us on 16 rows; handing it over through memory would cost more), then adds its rows' share of
// one weight gradient to the fp64 accumulators (atomics: one add per element per row group):
//   task 0   g1 = g2 W2, g0 = relu'(h0)(g1 W1), gx = bnrelu'(y)(g0 W0) -> global, + this row group's BatchNorm sums
//   task 1   g1;      dW1 += g1^T h0
//   task 2   g1, g0;  dW0 += g0^T bnrelu(y)
//   task 3            dW2 += g2^T h1
// g_i = gradient wrt the pre-activation of Linear i's output, [B][nout_i].  One fp32 MFMA pipe per CU (256 FLOP/clk) is what
// bounds a link, so the rows are spread over workgroups rather than recomputed for the whole batch.  Everything a task
// reads from global memory is fetched in one burst at its start.
// ---------------------------------------------------------------------------------------------------------------------
struct TailArgs {
    int B;
    HeadFc fc[3];           // Linear 0..2
    const float* y_last;    // encoder's last raw conv output [B][C*hw]
    float* g_last;          // out: masked gradient wrt its BatchNorm output
    const float *gamma, *beta, *saved;
    double* stats;          // [kStatShards][C][4]: slots 2, 3 (zero at step start); row group r adds into shard r % kStatShards
    int C, hw;
    int ld2, ld1, ld0;      // row strides of the 16-row g2 / g1 / g0 panels
    int o_g2, o_g1, o_g0, o_y, o_gx, o_w, o_part, o_c, o_red, zero4;   // LDS offsets (floats); zero4 = float4s from o_g2 to clear
    int ldw[3];             // LDS row strides of W0, W1, W2
    int w4[3];              // float4s of W0, W1, W2
    StageSplit sp_d[3];     // chain stages producing g1, g0, gx
    StageSplit sp_w[3];     // dW2, dW1, dW0
    long long* dbg;         // diagnostics (tools/head_phases.py): 16 wall-clock stamps per workgroup, or nullptr
};

// one row-major block [rows][4 * n4row] as coalesced 16-byte loads, at most two per thread
struct TailRegs {
    f32x4 v[2];
};
__device__ __forceinline__ TailRegs tail_fetch(const float* src, int total4, bool wanted) {
    TailRegs r;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int idx = min((int)threadIdx.x + j * kHeadThreads, total4 - 1);
        r.v[j] = wanted ? reinterpret_cast<const f32x4*>(src)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return r;
}
__device__ __forceinline__ void tail_put(const TailRegs& r, float* dst, int total4, int n4row, int ld) {
    const float inv = 1.0f / (float)n4row;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int idx = (int)threadIdx.x + j * kHeadThreads;
        if (idx < total4) {
            const int row = div_small(idx, inv);
            *reinterpret_cast<f32x4*>(dst + row * ld + 4 * (idx - row * n4row)) = r.v[j];
        }
    }
}

#define TAIL_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)

// dW[o][i] += sum over this workgroup's rows of g[b][o] * in(b, i); column nin of `in` is the constant 1 (bias gradient)
template <class FIN>
__device__ __forceinline__ void tail_wgrad(const float* g, int ldg, int nin, int nout, StageSplit sp, float* part, FIN in,
                                           double* accW, double* accB) {
    // K = the whole 16-row panel: rows past the batch are zero in g, so nothing here needs a predicate
    mfma_stage<true, true>(nout, nin + 1, 16, 0, (nout + 15) >> 4, 0, (nin + 16) >> 4, sp, part,
                             [=](int m, int k) { return g[k * ldg + m]; },
                             [=](int k, int n) { return n == nin ? 1.0f : in(k, n); },
                             [=](int m, int n, float v) {
                                 if (n == nin) acc_add<ACC_GRAD>(&accB[m], (double)v);
                                 else acc_add<ACC_GRAD>(&accW[(size_t)m * nin + n], (double)v);
                             });
}

// grid (ceil(B / 16), 4), block 1024
__global__ void __launch_bounds__(kHeadThreads) k_tail_bwd(TailArgs a) {
    kernarg_warm<sizeof(TailArgs)>();
    extern __shared__ double lds_d[];
    float* lds = reinterpret_cast<float*>(lds_d);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int task = blockIdx.y;
    const bool need1 = task != 3, need2 = task == 0 || task == 2, need3 = task == 0;
    const int row0 = blockIdx.x * 16, rows = min(16, a.B - row0);
    const HeadFc F0 = a.fc[0], F1 = a.fc[1], F2 = a.fc[2];

    TAIL_STAMP(0);
    // one burst of coalesced loads: this row group's g2; h1 / h0 / y rows, parked in the panel whose masked result they gate
    // (or, for the task whose weight gradient multiplies them, simply kept there); the weights of the links this task walks
    const int g4 = rows * (F2.nout >> 2), h14 = rows * (F1.nout >> 2), h04 = rows * (F0.nout >> 2), y4 = rows * (F0.nin >> 2);
    const bool want_h1 = task == 3 || (need1 && F1.relu), want_h0 = task == 1 || (need2 && F0.relu), want_y = task == 0 || task == 2;
    const TailRegs rg = tail_fetch(F2.grad + (size_t)row0 * F2.nout, g4, true);
    const TailRegs rh1 = tail_fetch(F1.act + (size_t)row0 * F1.nout, h14, want_h1);
    const TailRegs rh0 = tail_fetch(F0.act + (size_t)row0 * F0.nout, h04, want_h0);
    const TailRegs ry = tail_fetch(a.y_last + (size_t)row0 * F0.nin, y4, want_y);
    const TailRegs rw2 = tail_fetch(F2.w, a.w4[2], need1);
    const TailRegs rw1 = tail_fetch(F1.w, a.w4[1], need2);
    const TailRegs rw0 = tail_fetch(F0.w, a.w4[0], need3);
    for (int i = tid; i < a.zero4; i += kHeadThreads) reinterpret_cast<f32x4*>(lds + a.o_g2)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4* kc = reinterpret_cast<float4*>(lds + a.o_c);
    if (tid < a.C) {
        const float mean = a.saved[2 * tid], invstd = a.saved[2 * tid + 1];
        kc[tid] = make_float4(mean, a.gamma[tid] * invstd, a.beta[tid], invstd);
    }
    __syncthreads();
    TAIL_STAMP(1);
    float* g2 = lds + a.o_g2;
    float* g1 = lds + a.o_g1;
    float* g0 = lds + a.o_g0;
    float* ys = lds + a.o_y;   // [16][fc0.nin], dense
    float* Wb = lds + a.o_w;
    float* part = lds + a.o_part;
    const int ld2 = a.ld2, ld1 = a.ld1, ld0 = a.ld0;
    tail_put(rg, g2, g4, F2.nout >> 2, ld2);
    if (need1) tail_put(rw2, Wb, a.w4[2], F2.nin >> 2, a.ldw[2]);
    if (want_h1) tail_put(rh1, g1, h14, F1.nout >> 2, ld1);
    if (want_h0) tail_put(rh0, g0, h04, F0.nout >> 2, ld0);
    if (want_y) tail_put(ry, ys, y4, F0.nin >> 2, F0.nin);
    __syncthreads();
    TAIL_STAMP(2);

    if (need1)
    {   // g1[b][i] = relu'(h1) sum_o g2[b][o] W2[o][i]      (decoder.py:31-35 backwards)
        const int ldw = a.ldw[2], nin = F2.nin, relu = F1.relu;
        mfma_stage<true, true>(rows, nin, F2.nout, 0, 1, 0, (nin + 15) >> 4, a.sp_d[0], part,
                               [=](int m, int k) { return g2[m * ld2 + k]; },
                               [=](int k, int n) { return Wb[k * ldw + n]; },
                               [=](int m, int n, float v) {
                                   if (relu) v = g1[m * ld1 + n] > 0.f ? v : 0.f;   // the panel holds h1 until now
                                   g1[m * ld1 + n] = v;
                               });
    }
    TAIL_STAMP(3);
    if (need2) {   // g0[b][i] = relu'(h0) sum_o g1[b][o] W1[o][i]      (encoder.py:54-58 backwards)
        tail_put(rw1, Wb, a.w4[1], F1.nin >> 2, a.ldw[1]);
        __syncthreads();
        const int ldw = a.ldw[1], nin = F1.nin, relu = F0.relu;
        mfma_stage<true, true>(rows, nin, F1.nout, 0, 1, 0, (nin + 15) >> 4, a.sp_d[1], part,
                               [=](int m, int k) { return g1[m * ld1 + k]; },
                               [=](int k, int n) { return Wb[k * ldw + n]; },
                               [=](int m, int n, float v) {
                                   if (relu) v = g0[m * ld0 + n] > 0.f ? v : 0.f;   // the panel holds h0 until now
                                   g0[m * ld0 + n] = v;
                               });
    }
    TAIL_STAMP(4);
    if (need3) {   // gx[b][i] = bnrelu'(y) sum_o g0[b][o] W0[o][i] -> global, and a dense LDS copy for the BatchNorm sums
        tail_put(rw0, Wb, a.w4[0], F0.nin >> 2, a.ldw[0]);
        __syncthreads();
        const int ldw = a.ldw[0], nin = F0.nin, hw = a.hw;
        const float inv_hw = 1.0f / (float)hw;
        float* gl = a.g_last + (size_t)row0 * nin;
        float* gxs = lds + a.o_gx;
        mfma_stage<true, true>(rows, nin, F0.nout, 0, 1, 0, (nin + 15) >> 4, a.sp_d[2], part,
                               [=](int m, int k) { return g0[m * ld0 + k]; },
                               [=](int k, int n) { return Wb[k * ldw + n]; },
                               [=](int m, int n, float v) {
                                   const float4 c4 = kc[div_small(n, inv_hw)];
                                   const float yv = ys[m * nin + n];
                                   const float gm = fmaf(yv - c4.x, c4.y, c4.z) > 0.f ? v : 0.f;
                                   gl[(size_t)m * nin + n] = gm;
                                   gxs[m * nin + n] = gm;
                               });
        TAIL_STAMP(5);
        // this row group's share of dbeta = sum g and dgamma = sum g * xhat per channel (fp64, fixed order): up to 16 channels
        // at a time, 16 / channels waves each
        double* red = reinterpret_cast<double*>(lds + a.o_red);
        const int npos = rows * hw;
        for (int c0 = 0; c0 < a.C; c0 += kHeadWaves) {
            const int nc = min(kHeadWaves, a.C - c0);
            int wpc = 1;
            while (2 * wpc * nc <= kHeadWaves) wpc *= 2;
            const int cw = wv / wpc, sub = wv - cw * wpc;   // channel slot and share of this wave
            double s1 = 0.0, s2 = 0.0;
            if (cw < nc) {
                const int c = c0 + cw;
                const float4 c4 = kc[c];
                for (int idx = sub * 64 + lane; idx < npos; idx += wpc * 64) {
                    const int b = div_small(idx, inv_hw), r = idx - b * hw;
                    const int off = b * nin + c * hw + r;
                    const float gm = gxs[off];
                    const float d = ys[off] - c4.x;
                    s1 += (double)gm;
                    s2 += (double)gm * (double)(d * c4.w);
                }
            }
            s1 = head_wave_sum(s1);
            s2 = head_wave_sum(s2);
            if (lane == 63) {
                red[wv * 2] = s1;
                red[wv * 2 + 1] = s2;
            }
            __syncthreads();
            if (tid < nc) {
                double t1 = 0.0, t2 = 0.0;
                for (int w = 0; w < wpc; w++) {
                    t1 += red[(tid * wpc + w) * 2];
                    t2 += red[(tid * wpc + w) * 2 + 1];
                }
                double* row = a.stats + ((size_t)(blockIdx.x & (kStatShards - 1)) * a.C + c0 + tid) * 4;
                if (gridDim.x <= kStatShards) {
                    row[2] = t1;
                    row[3] = t2;
                } else {
                    acc_add<ACC_GRAD>(&row[2], t1);
                    acc_add<ACC_GRAD>(&row[3], t2);
                }
            }
            __syncthreads();
        }
        TAIL_STAMP(6);
    }
    // this row group's share of one weight gradient; the multiplier rows are already in LDS
    if (task == 3) {
        tail_wgrad(g2, ld2, F2.nin, F2.nout, a.sp_w[0], part, [=](int k, int n) { return g1[k * ld1 + n]; }, F2.w_acc, F2.b_acc);
    } else if (task == 1) {
        tail_wgrad(g1, ld1, F1.nin, F1.nout, a.sp_w[1], part, [=](int k, int n) { return g0[k * ld0 + n]; }, F1.w_acc, F1.b_acc);
    } else if (task == 2) {
        const int nin = F0.nin;
        const float inv_hw = 1.0f / (float)a.hw;
        tail_wgrad(g0, ld0, nin, F0.nout, a.sp_w[2], part,
                   [=](int k, int n) {
                       const float4 c4 = kc[div_small(n, inv_hw)];
                       return fmaxf(0.f, fmaf(ys[k * nin + n] - c4.x, c4.y, c4.z));
                   },
                   F0.w_acc, F0.b_acc);
    }
    TAIL_STAMP(7);
}

}  // namespace cae
