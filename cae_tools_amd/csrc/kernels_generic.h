// kernels_generic.h — shape-generic gfx950 kernels for every op of the ConvAE step.
//
// These are the fallback / reference-on-device kernels: any kernel size, stride, channel
// count, output_padding.  One thread per output element, BatchNorm + ReLU folded into the
// operand load of the consumer, all cross-block reductions through fp64 atomics.  The
// specialised kernels in kernels_s2.h replace them for the stride-2 layers that carry the
// traffic; both are checked against the CPU oracle.
//
// Every conv-like op is one of three primitives on a "small" map S (B,Cs,Hs,Ws), a "big" map
// L (B,Cl,Hl,Wl) and weights w[cs][cl][ky][kx]  (that IS PyTorch's layout for both
// Conv2d (Cout,Cin,kh,kw) with S = output and ConvTranspose2d (Cin,Cout,kh,kw) with S = input):
//   down : S[cs][y][x]   = sum_cl,ky,kx L[cl][y*s+ky][x*s+kx] * w      (Conv2d fwd, ConvT dgrad)
//   up   : L[cl][Y][X]   = sum_cs,ky,kx S[cs][(Y-ky)/s][(X-kx)/s] * w  (ConvT fwd, Conv2d dgrad)
//   wgrad: dw[cs][cl][ky][kx] = sum_b,y,x S[cs][y][x] * L[cl][y*s+ky][x*s+kx]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cae {

// BatchNorm sum accumulators are kept in kStatShards copies, [shard][C][4]; producers add into the
// copy picked by their block index (same-address fp64 atomics serialise at the memory side),
// consumers add the copies up in bn_consts.
constexpr int kStatShards = 8;

// ---- order-independent accumulation -----------------------------------------------------------------------------------------
// Every sum over workgroups (and the few over waves that go through LDS atomics) is a set of fp64 atomic adds whose ORDER
// differs from run to run.  fp64 addition is not associative, so such a sum is reproducible to ~1e-16 but not bitwise - and
// one last bit of an fp64 sum is, every few thousand values, one last bit of the fp32 number derived from it.  Here every
// addend is first rounded to a fixed absolute grid q (a power of two per kind of quantity).  A sum of multiples of q is EXACT
// in fp64 as long as it stays below 2^53 q, exact addition is associative, and the result therefore does not depend on the
// order the atomics arrive in: the training step is bitwise reproducible from run to run (tests/test_reproducible_gpu.py),
// for two fp64 instructions per atomic.  The grids:
//   ACC_STAT  sum y, sum y^2 of a BatchNorm layer (table slots 0, 1): q = 2^-28 (3.7e-9), exact while |sum| < 2^25 = 3.4e7
//             (N E[y^2] of a layer: 2e6 values of order 1 at the benchmark size);
//   ACC_GRAD  everything gradient-sized - sum g and sum g x_hat of BatchNorm backward (slots 2, 3), weight and bias gradients,
//             the loss: q = 2^-50 (8.9e-16), exact while |sum| < 8.
// A sum that leaves its window is not wrong, it merely rounds as fp64 sums always did (reproducible to ~1e-16 again).  The
// resolution costs at most 0.5 q per addend, and an addend covers >= 16 values, so E[y^2] and the mean carry an absolute error
// of at most 1.2e-10: on the variance of a layer with outputs of order 1 that is 1e-10 relative; in the worst case (a layer
// whose variance is far below BatchNorm's eps = 1e-5) 1 / sqrt(var + eps) moves by 6e-6.  Gradient sums: 4e-16 per addend.
enum AccKind : int { ACC_STAT = 0, ACC_GRAD = 1 };
template <int KIND>
__device__ __forceinline__ double acc_grid(double v) {
    constexpr double S = KIND == ACC_STAT ? 268435456.0 : 1125899906842624.0;   // 2^28, 2^50
    return __dmul_rn(rint(__dmul_rn(v, S)), 1.0 / S);
}
template <int KIND>
__device__ __forceinline__ void acc_add(double* p, double v) {
    atomicAdd(p, acc_grid<KIND>(v));
}

// Gradient accumulators that many workgroups hit at once (weights of the thin stride-2 layers,
// the last layer's bias) live in a sharded side table [kStatShards][n]; Adam adds the shards up.
struct ShardSeg {
    long long param_off;  // first parameter of the segment in the flat arena
    int count;
    int sh_off;           // offset inside one shard
};
struct ShardSegs {
    int nseg;
    int n;                // doubles per shard
    const double* base;   // [kStatShards][n]
    ShardSeg seg[12];
};

__device__ __forceinline__ double sharded_grad(const ShardSegs& ss, long long i) {
    double g = 0.0;
    for (int s = 0; s < ss.nseg; s++) {
        const long long d = i - ss.seg[s].param_off;
        if (d >= 0 && d < ss.seg[s].count) {
            for (int sh = 0; sh < kStatShards; sh++) g += ss.base[(size_t)sh * ss.n + ss.seg[s].sh_off + d];
        }
    }
    return g;
}

struct StepState {
    long long batch_start;  // first position in the permutation of the current batch
    int loss_slot;          // where this step's loss is accumulated
    int adam_step;          // completed optimiser steps
};

// How a tensor that is READ relates to BatchNorm.
enum BnMode : int {
    BN_NONE = 0,     // identity
    BN_BATCH = 1,    // activation a = relu((y-mean)*scale+beta), mean/var from this step's sums
    BN_RUNNING = 2,  // same with running statistics (eval)
    BN_SAVED = 3,    // same with the mean/invstd the forward pass saved (backward reads)
    BN_BWD = 4       // gradient wrt the raw conv output: k1*g - k2 - (y-mean)*k3
};

struct BnDesc {
    int mode;
    int C;
    const double* stats;  // [kStatShards][C][4]: sum y, sum y^2, sum g, sum g*xhat
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    float* saved;  // [C][2]: mean, invstd
    double count;  // elements per channel (global count under SyncBN)
    double inv_count, unbias;   // 1 / count and count / (count - 1) (1 for a single element): no fp64 divisions on the device
    float momentum;
    float eps;
    int update;    // BN_BATCH: this consumer also updates running stats and `saved`
};

// A tensor read through an optional per-channel transform.
struct Src {
    const float* p;    // data (for BN_BWD: the masked upstream gradient g)
    const float* q;    // BN_BWD only: the raw forward output y
    const int* perm;   // dataset gather: sample = perm[batch_start + b]   (nullptr: sample = b)
    int use_cursor;    // add StepState.batch_start even when perm == nullptr
    int C, H, W;
    int bump_adam;     // first kernel of a training step: its designated block starts optimiser step t+1
};

enum EpiKind : int { EPI_PLAIN = 0, EPI_STATS = 1, EPI_MASKSTATS = 2, EPI_SIGMSE = 3, EPI_SIGOUT = 4 };

struct Epi {
    int kind;
    float* out;           // PLAIN/STATS: raw output; MASKSTATS: masked gradient; SIGMSE: dL/d(pre-sigmoid)
    double* stats;        // STATS: [shards][C][4] slots 0,1; MASKSTATS: slots 2,3
    int stats_C;          // channel count of that table
    const float* yprev;   // MASKSTATS: raw forward output at the same element
    const float* target;  // SIGMSE / SIGOUT(optional)
    const int* perm;
    int use_cursor;
    double* losses;       // SIGMSE / SIGOUT: per-slot loss accumulators
    float inv_count;      // 1 / (global_batch * C * H * W)
    double* bias_acc;     // SIGMSE: gradient accumulator of the last layer's bias
    float* yhat;          // SIGOUT: sigmoid output (may be nullptr when only the loss is wanted)
};

struct ConvGeom {
    int B, Cs, Hs, Ws, Cl, Hl, Wl, kh, kw, s;
};

// ---------------------------------------------------------------------------------------------

// Requests every 64-byte line of the kernel's argument block at once, as the kernel's first instructions.  The compiler
// fetches arguments where they are first used: a kernel with a few hundred bytes of them starts with a chain of scalar loads
// from lines nobody has touched yet (the block is rewritten for every launch), each a trip to memory with little else in
// flight - measured on k_head_fwd (1.7 KB of arguments): its first phase 1.7 -> 1.0 us, 1.8 us off the step.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if defined(__HIP_DEVICE_COMPILE__)
    const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int kLines = (BYTES + 63) / 64;
    int w[kLines];
#pragma unroll
    for (int i = 0; i < kLines; i++) w[i] = ka[16 * i];
#pragma unroll
    for (int i = 0; i < kLines; i++) asm volatile("" :: "s"(w[i]));
#endif
}

// n / d for 0 <= n < 2^22 and d < 8000 with inv_d = 1.0f / d: (n + 0.5) / d is at least 0.5 / d away from an integer,
// far more than the rounding error of the fp32 product (3 instructions instead of the ~40 of an integer division)
__device__ __forceinline__ int div_small(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }
constexpr int kDivSmallMaxN = 1 << 22, kDivSmallMaxD = 8000;

// Wave-wide sums with DPP (VALU cross-lane moves) instead of __shfl (ds_bpermute, an LDS-pipe
// instruction with ~50 cycles of latency per step): quad swaps, row mirrors, then the GFX9
// row-broadcasts.  The total lands in lane 63 and is broadcast back with readlane.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}

__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror
    v += dpp_f<0x140>(v);       // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// fp64 wave sum on the VALU: both halves of the double travel by DPP (the ds_bpermute butterfly costs ~100 cycles a step)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_d(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// valid in lane 63 only
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v += dpp_d<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);       // row_half_mirror
    v += dpp_d<0x140>(v);       // row_mirror
    v += dpp_d<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_d<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
// valid in every lane
__device__ __forceinline__ double wave_sum(double v) {
    const long long bits = __builtin_bit_cast(long long, wave_sum_lane63(v));
    const int lo = __builtin_amdgcn_readlane((int)bits, 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// sum of v over the block, valid in thread 0.  red: LDS scratch of >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) t += red[i];
    }
    return t;
}

// Per-channel constants for a BnDesc into LDS.
//   activation modes: {mean, gamma*invstd, beta, invstd}
//   BN_BWD:           {mean, k1, k2, k3}  with  gy = k1*g - k2 - (y-mean)*k3
// BN_BATCH constants of channel c in two halves, so that a prologue can request the sums before its other loads and
// finish behind them: request = every global read, finish = arithmetic, the LDS entry and (upd) the running statistics.
struct BnBatchReq {
    double2 t[kStatShards];
    float gamma, beta, rm, rv;
};
__device__ __forceinline__ void bn_batch_request(const BnDesc& d, int c, bool upd, BnBatchReq& r) {
    r.gamma = d.gamma[c];
#pragma unroll
    for (int sh = 0; sh < kStatShards; sh++) r.t[sh] = *reinterpret_cast<const double2*>(d.stats + ((size_t)sh * d.C + c) * 4);
    r.beta = d.beta[c];
    r.rm = 0.f;
    r.rv = 0.f;
    if (upd) {
        r.rm = d.rmean[c];
        r.rv = d.rvar[c];
    }
}
__device__ __forceinline__ void bn_batch_finish(const BnDesc& d, int c, bool upd, const BnBatchReq& r, float4* out) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int sh = 0; sh < kStatShards; sh++) {
        s1 += r.t[sh].x;
        s2 += r.t[sh].y;
    }
    // multiplications by host-computed reciprocals and an fp32 square root instead of three fp64 divisions and an fp64
    // square root
    const double m = s1 * d.inv_count;
    double var = s2 * d.inv_count - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)m;
    const float invstd = 1.0f / sqrtf((float)(var + (double)d.eps));
    out[c] = make_float4(mean, r.gamma * invstd, r.beta, invstd);
    if (upd) {
        d.saved[2 * c] = mean;
        d.saved[2 * c + 1] = invstd;
        const double unb = var * d.unbias;
        d.rmean[c] = (1.f - d.momentum) * r.rm + d.momentum * mean;
        d.rvar[c] = (1.f - d.momentum) * r.rv + d.momentum * (float)unb;
    }
}

// `first`: the thread that takes channel 0.  A kernel with two descriptors gives the second one to its second wave
// (first = 64): on one wave the two would be two trips to memory one after the other, the loads of the second behind the
// wait of the first.
__device__ __forceinline__ void bn_consts(const BnDesc& d, float4* out, bool designated, int first = 0) {
    if (d.mode == BN_NONE) return;
    // Every consumer's prologue runs this on its critical path, so within a mode every global read is requested before the
    // first wait and before the first store (a store to memory the reads might alias pins the later reads behind it: the
    // sums, then gamma and beta, then the running statistics used to be three to four trips to memory, one after the other).
    if (first >= (int)blockDim.x) first = 0;
    int c_first = (int)threadIdx.x - first;
    if (c_first < 0) c_first += blockDim.x;
    for (int c = c_first; c < d.C; c += blockDim.x) {
        float mean, invstd;
        if (d.mode == BN_BATCH) {
            const bool upd = designated && d.update;
            BnBatchReq rq;
            bn_batch_request(d, c, upd, rq);
            bn_batch_finish(d, c, upd, rq, out);
            continue;
        }
        const float gamma = d.gamma[c];
        if (d.mode == BN_RUNNING) {
            const float beta = d.beta[c];
            mean = d.rmean[c];
            invstd = 1.0f / sqrtf(d.rvar[c] + d.eps);
            out[c] = make_float4(mean, gamma * invstd, beta, invstd);
        } else if (d.mode == BN_BWD) {
            double2 t[kStatShards];
#pragma unroll
            for (int sh = 0; sh < kStatShards; sh++) t[sh] = *reinterpret_cast<const double2*>(d.stats + ((size_t)sh * d.C + c) * 4 + 2);
            mean = d.saved[2 * c];
            invstd = d.saved[2 * c + 1];
            double dbeta = 0.0, dgamma = 0.0;
#pragma unroll
            for (int sh = 0; sh < kStatShards; sh++) {
                dbeta += t[sh].x;
                dgamma += t[sh].y;
            }
            const float scale = gamma * invstd;
            const float k2 = (float)((double)scale * dbeta * d.inv_count);
            const float k3 = (float)((double)scale * (double)invstd * dgamma * d.inv_count);
            out[c] = make_float4(mean, scale, k2, k3);
        } else {   // saved statistics of this step (activation recomputed in the backward pass)
            const float beta = d.beta[c];
            mean = d.saved[2 * c];
            invstd = d.saved[2 * c + 1];
            out[c] = make_float4(mean, gamma * invstd, beta, invstd);
        }
    }
}

__device__ __forceinline__ float bn_apply(int mode, const float4 k, float v, float yraw) {
    if (mode == BN_NONE) return v;
    if (mode == BN_BWD) return k.y * v - k.z - (yraw - k.x) * k.w;
    return fmaxf(0.f, fmaf(v - k.x, k.y, k.z));
}

__device__ __forceinline__ size_t sample_of(const int* perm, int use_cursor, const StepState* st, int b) {
    if (perm) return (size_t)perm[st->batch_start + b];
    if (use_cursor) return (size_t)(st->batch_start + b);
    return (size_t)b;
}

// ---------------------------------------------------------------------------------------------
// shared epilogue: thread-level part returns the two values to be block-reduced
// ---------------------------------------------------------------------------------------------
// `ypre` (with have_ypre): the caller already holds e.yprev[o] (EPI_MASKSTATS), requested ahead of its main loop
__device__ __forceinline__ void epi_element(const Epi& e, const float4* ce, int c, size_t o, size_t tgt_off,
                                             float acc, double& r1, double& r2, bool have_ypre = false, float ypre = 0.f) {
    switch (e.kind) {
        case EPI_PLAIN:
            e.out[o] = acc;
            break;
        case EPI_STATS:
            e.out[o] = acc;
            r1 = (double)acc;
            r2 = (double)acc * (double)acc;
            break;
        case EPI_MASKSTATS: {
            const float4 k = ce[c];
            const float yv = have_ypre ? ypre : e.yprev[o];
            const float d = yv - k.x;
            const float a = fmaf(d, k.y, k.z);
            const float g = a > 0.f ? acc : 0.f;
            e.out[o] = g;
            r1 = (double)g;
            r2 = (double)g * (double)(d * k.w);
            break;
        }
        case EPI_SIGMSE: {
            const float yh = 1.0f / (1.0f + expf(-acc));
            const float d = yh - e.target[tgt_off];
            const float g = (2.0f * d * e.inv_count) * (yh * (1.0f - yh));
            e.out[o] = g;
            r1 = (double)d * (double)d * (double)e.inv_count;
            r2 = (double)g;
            break;
        }
        case EPI_SIGOUT: {
            const float yh = 1.0f / (1.0f + expf(-acc));
            if (e.yhat) e.yhat[o] = yh;
            if (e.target) {
                const float d = yh - e.target[tgt_off];
                r1 = (double)d * (double)d * (double)e.inv_count;
            }
            break;
        }
    }
}

__device__ __forceinline__ void epi_block(const Epi& e, int c, const StepState* st, double r1, double r2,
                                          double* red, int bx) {
    if (e.kind == EPI_PLAIN) return;
    if (e.kind == EPI_SIGOUT && !e.target) return;
    const double t1 = block_sum(r1, red);
    const double t2 = (e.kind == EPI_SIGOUT) ? 0.0 : block_sum(r2, red);
    if (threadIdx.x == 0) {
        const size_t row = ((size_t)(bx & (kStatShards - 1)) * e.stats_C + c) * 4;
        if (e.kind == EPI_STATS) {
            acc_add<ACC_STAT>(&e.stats[row + 0], t1);
            acc_add<ACC_STAT>(&e.stats[row + 1], t2);
        } else if (e.kind == EPI_MASKSTATS) {
            acc_add<ACC_GRAD>(&e.stats[row + 2], t1);
            acc_add<ACC_GRAD>(&e.stats[row + 3], t2);
        } else if (e.kind == EPI_SIGMSE) {
            acc_add<ACC_GRAD>(&e.losses[(size_t)st->loss_slot * kStatShards + (bx & (kStatShards - 1))], t1);
            acc_add<ACC_GRAD>(&e.bias_acc[c], t2);
        } else {
            acc_add<ACC_GRAD>(&e.losses[(size_t)st->loss_slot * kStatShards + (bx & (kStatShards - 1))], t1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// down: S[b][cs][y][x] = bias[cs] + sum_cl,ky,kx T(L[b][cl][y*s+ky][x*s+kx]) * w[cs][cl][ky][kx]
// grid (ceil(B*Hs*Ws/256), Cs), block 256, dynamic LDS (Cl + Cs) float4 + 4 doubles
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_down(ConvGeom g, Src big, BnDesc bnb, const float* __restrict__ w,
                                               const float* __restrict__ bias, Epi e, BnDesc bne,
                                               const StepState* __restrict__ st) {
    extern __shared__ double lds_d[];
    double* red = lds_d;
    float4* cb = reinterpret_cast<float4*>(lds_d + 4);
    float4* ce = cb + g.Cl;
    const bool designated = blockIdx.x == 0 && blockIdx.y == 0;
    bn_consts(bnb, cb, designated);
    bn_consts(bne, ce, false, 64);
    // no kernel of the forward/backward pass reads adam_step, so bumping it here cannot race
    if (designated && threadIdx.x == 0 && big.bump_adam) const_cast<StepState*>(st)->adam_step += 1;
    __syncthreads();

    const int cs = blockIdx.y;
    const int hw = g.Hs * g.Ws;
    const int n = g.B * hw;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    double r1 = 0, r2 = 0;
    if (idx < n) {
        const int b = idx / hw, r = idx - b * hw;
        const int y = r / g.Ws, x = r - y * g.Ws;
        const size_t sb = sample_of(big.perm, big.use_cursor, st, b);
        float acc = bias ? bias[cs] : 0.f;
        for (int cl = 0; cl < g.Cl; cl++) {
            const float* wp = w + (size_t)(cs * g.Cl + cl) * g.kh * g.kw;
            const size_t base = (sb * g.Cl + cl) * (size_t)g.Hl * g.Wl;
            const float4 k = bnb.mode ? cb[cl] : make_float4(0, 0, 0, 0);
            for (int ky = 0; ky < g.kh; ky++) {
                const size_t row = base + (size_t)(y * g.s + ky) * g.Wl + (size_t)x * g.s;
                for (int kx = 0; kx < g.kw; kx++) {
                    const float v = big.p[row + kx];
                    const float yr = bnb.mode == BN_BWD ? big.q[row + kx] : 0.f;
                    acc = fmaf(bn_apply(bnb.mode, k, v, yr), wp[ky * g.kw + kx], acc);
                }
            }
        }
        const size_t o = ((size_t)(b * g.Cs + cs) * g.Hs + y) * g.Ws + x;
        epi_element(e, ce, cs, o, 0, acc, r1, r2);
    }
    epi_block(e, cs, st, r1, r2, red, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// up: L[b][cl][Y][X] = bias[cl] + sum_cs sum_{ky = Y mod s (s) , kx = X mod s (s)}
//                                   T(S[b][cs][(Y-ky)/s][(X-kx)/s]) * w[cs][cl][ky][kx]
// grid (ceil(B*Hl*Wl/256), Cl)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void up_body(const ConvGeom& g, const Src& small, const BnDesc& bns, const float* __restrict__ w,
                                        const float* __restrict__ bias, const Epi& e, const BnDesc& bne,
                                        const StepState* __restrict__ st, double* lds_d, const int bx, const int by) {
    double* red = lds_d;
    float4* cs4 = reinterpret_cast<float4*>(lds_d + 4);
    float4* ce = cs4 + g.Cs;
    const bool designated = bx == 0 && by == 0;

    const int cl = by;
    const int hw = g.Hl * g.Wl;
    const int n = g.B * hw;
    const int idx = bx * 256 + threadIdx.x;
    const bool mine = idx < n;
    const int ic = mine ? idx : 0;
    const int b = ic / hw, r = ic - b * hw;
    const int Y = r / g.Wl, X = r - Y * g.Wl;
    const size_t o = ((size_t)(b * g.Cl + cl) * g.Hl + Y) * g.Wl + X;
    // taps of output (Y, X): ky = Y mod s, + s, ... reading row y = Y / s, - 1, ...: one division per dimension per
    // thread instead of one per (channel, tap)
    const int y0 = Y / g.s, x0 = X / g.s;
    const int ky0 = Y - y0 * g.s, kx0 = X - x0 * g.s;
    // At most 2 x 2 taps per output (3x3 or 4x4 kernels at stride 2).  The general loop further down issues one load, waits,
    // multiplies, per (channel, tap): ~16 dependent memory round trips per thread.  Here four channels' taps, their raw
    // outputs (BatchNorm backward) and weights are fetched together from clamped addresses, then consumed in the same
    // (channel, ky, kx) order.
    const bool four_taps = g.kh <= 2 * g.s && g.kw <= 2 * g.s;
    int yy[2], xx[2], kyy[2], kxx[2];
    bool vy[2], vx[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int ky = ky0 + t * g.s, y = y0 - t, kx = kx0 + t * g.s, x = x0 - t;
        vy[t] = ky < g.kh && y >= 0 && y < g.Hs;
        vx[t] = kx < g.kw && x >= 0 && x < g.Ws;
        yy[t] = min(max(y, 0), g.Hs - 1);
        xx[t] = min(max(x, 0), g.Ws - 1);
        kyy[t] = min(ky, g.kh - 1);
        kxx[t] = min(kx, g.kw - 1);
    }
    const int khw = g.kh * g.kw;
    size_t sb = 0;
    auto load_group = [&](int c0, float (&gv)[4][4], float (&qv)[4][4], float (&wv)[4][4]) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int cs = min(c0 + c, g.Cs - 1);
            const float* wp = w + (size_t)(cs * g.Cl + cl) * khw;
            const size_t base = (sb * g.Cs + cs) * (size_t)g.Hs * g.Ws;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const size_t off = base + (size_t)yy[t >> 1] * g.Ws + xx[t & 1];
                gv[c][t] = small.p[off];
                qv[c][t] = bns.mode == BN_BWD ? small.q[off] : 0.f;
                wv[c][t] = wp[kyy[t >> 1] * g.kw + kxx[t & 1]];
            }
        }
    };
    // The first group of four channels and the epilogue's mask input are requested before the BatchNorm constants are worked
    // out (their sums, the barrier): one trip to memory for all of it instead of three in a row.
    float gv0[4][4], qv0[4][4], wv0[4][4];
    float ypre = 0.f;
    const bool pre_mask = e.kind == EPI_MASKSTATS;
    if (mine) {
        sb = sample_of(small.perm, small.use_cursor, st, b);
        if (four_taps) load_group(0, gv0, qv0, wv0);
        if (pre_mask) ypre = e.yprev[o];
    }

    bn_consts(bns, cs4, designated);
    bn_consts(bne, ce, false, 64);
    __syncthreads();

    double r1 = 0, r2 = 0;
    if (mine) {
        float acc = bias ? bias[cl] : 0.f;
        if (four_taps) {
            for (int c0 = 0; c0 < g.Cs; c0 += 4) {
                float gv[4][4], qv[4][4], wv[4][4];
                if (c0 == 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            gv[c][t] = gv0[c][t];
                            qv[c][t] = qv0[c][t];
                            wv[c][t] = wv0[c][t];
                        }
                } else {
                    load_group(c0, gv, qv, wv);
                }
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int t = 0; t < 4; t++) asm volatile("" : "+v"(gv[c][t]), "+v"(qv[c][t]), "+v"(wv[c][t]));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (c0 + c >= g.Cs) break;
                    const float4 k = bns.mode ? cs4[c0 + c] : make_float4(0, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if (vy[t >> 1] && vx[t & 1]) acc = fmaf(bn_apply(bns.mode, k, gv[c][t], qv[c][t]), wv[c][t], acc);
                }
            }
        } else
        for (int cs = 0; cs < g.Cs; cs++) {
            const float* wp = w + (size_t)(cs * g.Cl + cl) * g.kh * g.kw;
            const size_t base = (sb * g.Cs + cs) * (size_t)g.Hs * g.Ws;
            const float4 k = bns.mode ? cs4[cs] : make_float4(0, 0, 0, 0);
            for (int ky = ky0, y = y0; ky < g.kh && y >= 0; ky += g.s, y--) {
                if (y >= g.Hs) continue;
                for (int kx = kx0, x = x0; kx < g.kw && x >= 0; kx += g.s, x--) {
                    if (x >= g.Ws) continue;
                    const size_t off = base + (size_t)y * g.Ws + x;
                    const float v = small.p[off];
                    const float yr = bns.mode == BN_BWD ? small.q[off] : 0.f;
                    acc = fmaf(bn_apply(bns.mode, k, v, yr), wp[ky * g.kw + kx], acc);
                }
            }
        }
        size_t tgt = 0;
        if (e.kind >= EPI_SIGMSE && e.target) {
            const size_t tb = sample_of(e.perm, e.use_cursor, st, b);
            tgt = ((tb * g.Cl + cl) * (size_t)g.Hl + Y) * g.Wl + X;
        }
        epi_element(e, ce, cl, o, tgt, acc, r1, r2, pre_mask, ypre);
    }
    epi_block(e, cl, st, r1, r2, red, bx);
}

__global__ void __launch_bounds__(256) k_up(ConvGeom g, Src small, BnDesc bns, const float* __restrict__ w,
                                             const float* __restrict__ bias, Epi e, BnDesc bne,
                                             const StepState* __restrict__ st) {
    extern __shared__ double lds_d[];
    up_body(g, small, bns, w, bias, e, bne, st, lds_d, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// wgrad: acc[cs][cl][ky][kx] += sum_{b,y,x} Ts(S[b][cs][y][x]) * Tl(L[b][cl][y*s+ky][x*s+kx])
// grid (Cs*Cl*kh*kw, nsplit); each block covers `ppb` positions of (b,y,x).
// The designated block also publishes the BatchNorm parameter gradients of the layer whose
// backward this is (dgamma = sum g*xhat, dbeta = sum g), taken from the fp64 stat sums.
// ---------------------------------------------------------------------------------------------
struct BnGradOut {
    const double* stats;  // [shards][C][4] of the BN that follows this layer, or nullptr
    double* gamma_acc;
    double* beta_acc;
    int C;
    double scale;  // 1/nranks under SyncBN (the sums are already global), else 1
};

__device__ __forceinline__ void wgrad_body(const ConvGeom& g, const Src& small, const BnDesc& bns, const Src& big,
                                           const BnDesc& bnb, double* __restrict__ acc, int ppb, const BnGradOut& bg,
                                           const StepState* __restrict__ st, double* lds_d, const int bx, const int by) {
    double* red = lds_d;
    float4* cs4 = reinterpret_cast<float4*>(lds_d + 4);
    float4* cb = cs4 + g.Cs;

    int widx = bx;
    const int kx = widx % g.kw;
    widx /= g.kw;
    const int ky = widx % g.kh;
    widx /= g.kh;
    const int cl = widx % g.Cl;
    const int cs = widx / g.Cl;
    const int hw = g.Hs * g.Ws;
    const long long n = (long long)g.B * hw;
    const long long p0 = (long long)by * ppb;
    long long p1 = p0 + ppb;
    if (p1 > n) p1 = n;
    const bool fastdiv = n < kDivSmallMaxN && hw < kDivSmallMaxD;   // 3-instruction index arithmetic instead of 64-bit divisions
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)g.Ws;
    // raw operands of position p (the BatchNorm transforms are applied by the caller)
    auto fetch = [&](long long p, float& sp, float& sq, float& lp, float& lq) {
        int b, y;
        if (fastdiv) {
            b = div_small((int)p, inv_hw);
            y = div_small((int)p - b * hw, inv_w);
        } else {
            b = (int)(p / hw);
            y = (int)(p - (long long)b * hw) / g.Ws;
        }
        const int r = (int)(p - (long long)b * hw), x = r - y * g.Ws;
        const size_t ss = sample_of(small.perm, small.use_cursor, st, b);
        const size_t sl = sample_of(big.perm, big.use_cursor, st, b);
        const size_t so = ((ss * g.Cs + cs) * (size_t)g.Hs + y) * g.Ws + x;
        const size_t lo = ((sl * g.Cl + cl) * (size_t)g.Hl + (size_t)(y * g.s + ky)) * g.Wl + (size_t)x * g.s + kx;
        sp = small.p[so];
        sq = bns.mode == BN_BWD ? small.q[so] : 0.f;
        lp = big.p[lo];
        lq = bnb.mode == BN_BWD ? big.q[lo] : 0.f;
    };
    // The first position's operands are requested before the BatchNorm constants are worked out: with the usual one
    // position per thread (wgrad_ppb) the kernel is then one trip to memory deep instead of two (cursor -> operands used to
    // start behind the constants' sums and the barrier).
    const long long pf = p0 + threadIdx.x;
    float f_sp = 0.f, f_sq = 0.f, f_lp = 0.f, f_lq = 0.f;
    if (pf < p1) fetch(pf, f_sp, f_sq, f_lp, f_lq);

    bn_consts(bns, cs4, false);
    bn_consts(bnb, cb, false, 64);
    if (bx == 0 && by == 0 && bg.stats) {
        for (int c = threadIdx.x; c < bg.C; c += blockDim.x) {
            double sb = 0.0, sg = 0.0;
            for (int sh = 0; sh < kStatShards; sh++) {
                sb += bg.stats[((size_t)sh * bg.C + c) * 4 + 2];
                sg += bg.stats[((size_t)sh * bg.C + c) * 4 + 3];
            }
            bg.beta_acc[c] = sb * bg.scale;
            bg.gamma_acc[c] = sg * bg.scale;
        }
    }
    __syncthreads();
    const float4 ks = bns.mode ? cs4[cs] : make_float4(0, 0, 0, 0);
    const float4 kb = bnb.mode ? cb[cl] : make_float4(0, 0, 0, 0);

    float sum = 0.f;
    if (pf < p1) sum = bn_apply(bns.mode, ks, f_sp, f_sq) * bn_apply(bnb.mode, kb, f_lp, f_lq);
    for (long long p = pf + 256; p < p1; p += 256) {
        float sp, sq, lp, lq;
        fetch(p, sp, sq, lp, lq);
        sum = fmaf(bn_apply(bns.mode, ks, sp, sq), bn_apply(bnb.mode, kb, lp, lq), sum);
    }
    const double t = block_sum((double)sum, red);
    if (threadIdx.x == 0) acc_add<ACC_GRAD>(&acc[bx], t);
}

__global__ void __launch_bounds__(256) k_wgrad(ConvGeom g, Src small, BnDesc bns, Src big, BnDesc bnb,
                                                double* __restrict__ acc, int ppb, BnGradOut bg,
                                                const StepState* __restrict__ st) {
    kernarg_warm<sizeof(ConvGeom) + 2 * sizeof(Src) + 2 * sizeof(BnDesc) + sizeof(BnGradOut) + 32>();
    extern __shared__ double lds_d[];
    wgrad_body(g, small, bns, big, bnb, acc, ppb, bg, st, lds_d, blockIdx.x, blockIdx.y);
}

// One launch for a Conv2d layer's weight gradient AND its input gradient (they share only their inputs): workgroups
// [0, nwx * nwy) are k_wgrad's grid, the rest k_up's (ux wide).
struct WgradArgs {
    ConvGeom g;
    Src small;
    BnDesc bns;
    Src big;
    BnDesc bnb;
    double* acc;
    int ppb;
    BnGradOut bg;
};
struct UpArgs {
    ConvGeom g;
    Src small;
    BnDesc bns;
    const float* w;
    const float* bias;
    Epi e;
    BnDesc bne;
};
__global__ void __launch_bounds__(256) k_conv_bwd_pair(WgradArgs wa, UpArgs ua, int nwx, int nwy, int ux,
                                                        const StepState* __restrict__ st) {
    kernarg_warm<sizeof(WgradArgs) + sizeof(UpArgs) + 24>();
    extern __shared__ double lds_d[];
    const int i = blockIdx.x, nw = nwx * nwy;
    if (i < nw) {
        const int by = i / nwx;
        wgrad_body(wa.g, wa.small, wa.bns, wa.big, wa.bnb, wa.acc, wa.ppb, wa.bg, st, lds_d, i - by * nwx, by);
    } else {
        const int j = i - nw, by = j / ux;
        up_body(ua.g, ua.small, ua.bns, ua.w, ua.bias, ua.e, ua.bne, st, lds_d, j - by * ux, by);
    }
}

// ---------------------------------------------------------------------------------------------
// Linear layers (encoder.py:54-58, decoder.py:31-35), weight W[out][in]
// ---------------------------------------------------------------------------------------------

// out[b][o] = act(bias[o] + sum_i T(in[b][i]) * W[o][i]);  T = BN+ReLU per channel i/hw or identity
__global__ void __launch_bounds__(256) k_lin_fwd(int B, int nin, int nout, const float* __restrict__ in,
                                                  BnDesc bni, int hw, const float* __restrict__ W,
                                                  const float* __restrict__ bias, int relu,
                                                  float* __restrict__ out) {
    extern __shared__ double lds_d[];
    float4* ci = reinterpret_cast<float4*>(lds_d + 4);
    bn_consts(bni, ci, blockIdx.x == 0);
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * nout) return;
    const int b = idx / nout, o = idx - b * nout;
    const float* ip = in + (size_t)b * nin;
    const float* wp = W + (size_t)o * nin;
    float acc = bias[o];
    if (bni.mode) {
        for (int i = 0; i < nin; i++) acc = fmaf(bn_apply(bni.mode, ci[i / hw], ip[i], 0.f), wp[i], acc);
    } else {
        for (int i = 0; i < nin; i++) acc = fmaf(ip[i], wp[i], acc);
    }
    out[idx] = relu ? fmaxf(acc, 0.f) : acc;
}

// gin[b][i] = M( sum_o gout[b][o] * W[o][i] )
//   epi 0: M = identity; 1: M = mask by hin[b][i] > 0 (ReLU of the producing Linear);
//   epi 2: M = mask by BN+ReLU of yprev + D1/D2 sums (grid.y = channel, i = c*hw + r)
__global__ void __launch_bounds__(256) k_lin_dgrad(int B, int nin, int nout, const float* __restrict__ gout,
                                                    const float* __restrict__ W, int epi,
                                                    const float* __restrict__ hin, BnDesc bne, int hw,
                                                    double* stats, float* __restrict__ gin) {
    extern __shared__ double lds_d[];
    double* red = lds_d;
    float4* ce = reinterpret_cast<float4*>(lds_d + 4);
    bn_consts(bne, ce, false);
    __syncthreads();
    double r1 = 0, r2 = 0;
    if (epi != 2) {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx < B * nin) {
            const int b = idx / nin, i = idx - b * nin;
            const float* gp = gout + (size_t)b * nout;
            float acc = 0.f;
            for (int o = 0; o < nout; o++) acc = fmaf(gp[o], W[(size_t)o * nin + i], acc);
            if (epi == 1) acc = hin[idx] > 0.f ? acc : 0.f;
            gin[idx] = acc;
        }
        return;
    }
    const int c = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < B * hw) {
        const int b = idx / hw, r = idx - b * hw;
        const int i = c * hw + r;
        const float* gp = gout + (size_t)b * nout;
        float acc = 0.f;
        for (int o = 0; o < nout; o++) acc = fmaf(gp[o], W[(size_t)o * nin + i], acc);
        const size_t off = (size_t)b * nin + i;
        const float4 k = ce[c];
        const float d = hin[off] - k.x;
        const float a = fmaf(d, k.y, k.z);
        const float gm = a > 0.f ? acc : 0.f;
        gin[off] = gm;
        r1 = (double)gm;
        r2 = (double)gm * (double)(d * k.w);
    }
    const double t1 = block_sum(r1, red);
    const double t2 = block_sum(r2, red);
    if (threadIdx.x == 0) {
        const size_t row = ((size_t)(blockIdx.x & (kStatShards - 1)) * bne.C + c) * 4;
        acc_add<ACC_GRAD>(&stats[row + 2], t1);
        acc_add<ACC_GRAD>(&stats[row + 3], t2);
    }
}

// accW[o][i] = sum_b gout[b][o] * T(in[b][i]);  accB[o] = sum_b gout[b][o]
__global__ void __launch_bounds__(256) k_lin_wgrad(int B, int nin, int nout, const float* __restrict__ gout,
                                                    const float* __restrict__ in, BnDesc bni, int hw,
                                                    double* __restrict__ accW, double* __restrict__ accB) {
    extern __shared__ double lds_d[];
    float4* ci = reinterpret_cast<float4*>(lds_d + 4);
    bn_consts(bni, ci, false);
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nout * nin) return;
    const int o = idx / nin, i = idx - o * nin;
    const float4 k = bni.mode ? ci[i / hw] : make_float4(0, 0, 0, 0);
    float sum = 0.f, bsum = 0.f;
    for (int b = 0; b < B; b++) {
        const float gv = gout[(size_t)b * nout + o];
        sum = fmaf(gv, bn_apply(bni.mode, k, in[(size_t)b * nin + i], 0.f), sum);
        bsum += gv;
    }
    accW[idx] = (double)sum;
    if (i == 0) accB[o] = (double)bsum;
}

// ---------------------------------------------------------------------------------------------
// optimiser + bookkeeping
// ---------------------------------------------------------------------------------------------
struct Hyper {
    double lr, beta1, beta2, eps, wd;
};

// What the last kernel of a step does besides its own work: zero the accumulators for the next step
// (each fp64 gradient slot is cleared by the thread that just consumed it; the BatchNorm sum tables,
// which nobody reads any more, by a grid-stride sweep) and move the cursor (nobody in this kernel
// reads batch_start / loss_slot).
struct StepTail {
    double* zero_extra;       // BatchNorm sum tables (and whatever else sits before the gradient accumulator)
    long long zero_extra_n;   // doubles
    double* acc_rw;           // fp64 gradient accumulator, writable view (or nullptr)
    double* shard_rw;         // sharded side table, writable view (or nullptr)
    StepState* st;
    int batch_inc, slot_inc;
};

__device__ __forceinline__ float consume_grad(const StepTail& tl, const ShardSegs& ss, long long i) {
    double g = tl.acc_rw[i];
    tl.acc_rw[i] = 0.0;
    for (int s = 0; s < ss.nseg; s++) {
        const long long d = i - ss.seg[s].param_off;
        if (d >= 0 && d < ss.seg[s].count) {
            // all eight shard reads in flight before the first clearing store (a load/store per shard in turn is eight
            // dependent round trips for the threads that own sharded parameters: the tail of this kernel)
            double* p = tl.shard_rw + ss.seg[s].sh_off + d;
            double v[kStatShards];
#pragma unroll
            for (int sh = 0; sh < kStatShards; sh++) v[sh] = p[(size_t)sh * ss.n];
#pragma unroll
            for (int sh = 0; sh < kStatShards; sh++) {
                g += v[sh];
                p[(size_t)sh * ss.n] = 0.0;
            }
        }
    }
    return (float)g;
}

__device__ __forceinline__ void step_tail(const StepTail& tl, long long gid, long long nthreads) {
    for (long long j = gid; j < tl.zero_extra_n; j += nthreads) tl.zero_extra[j] = 0.0;
    if (gid == 0 && tl.st) {
        tl.st->batch_start += tl.batch_inc;
        tl.st->loss_slot += tl.slot_inc;
    }
}

// torch.optim.Adam single-tensor update (L2 weight decay added to the gradient), element-wise
// over the flat arena.  Gradient source: fp64 accumulators (fused path; consumed and cleared) or
// the fp32 arena (data-parallel path, after the all-reduce).  st->adam_step is the number of the
// step being taken (bumped by the first kernel of the step / by cae_adam_step's own bump flag).
// The first encoder layer's weight gradient, folded into the optimiser launch (k_adam): it is the last kernel of backward, a
// reduction of B*Hs*Ws products per weight that nothing but the optimiser waits for - as its own launch 5.6 us, 3 % of the
// step.  One extra workgroup per weight: it reduces the whole batch itself (no cross-workgroup sum, so no ordering between
// the reduction and the update), then updates that one parameter.  The inputs come from the contiguous copy of the batch
// that k_head_fwd leaves behind (the cursor has moved on by now: this kernel moves it).  The BatchNorm parameter gradients
// of that layer, which the weight-gradient launch used to turn from sums into accumulator entries, are taken from the sums
// directly by the threads that own gamma and beta; the layer's sum table is left for the next step's first kernel to clear.
struct AdamConv0 {
    int on, nw, C, n_regular;   // n_regular: workgroups of the element-wise update (then one per weight)
    long long w_off, gamma_off, beta_off;
    ConvGeom g;
    Src small;                  // p = masked gradient of the layer's output, q = its raw output (BN_BWD)
    BnDesc bns;
    const float* xb;            // (B, Cl, Hl, Wl) the batch's inputs, contiguous
    const double* stats;        // the layer's BatchNorm table [shards][C][4]
    double scale;               // 1 / world (1 here: the fused form is single-device only)
};
constexpr int kAdamConv0Pos = 16;   // positions per thread requested in one burst

__device__ __forceinline__ void adam_conv0_body(const AdamConv0& c0, float* __restrict__ p, float* __restrict__ m,
                                                float* __restrict__ v, const Hyper& h, const StepState* __restrict__ st,
                                                int t_add, double ln_b1, double ln_b2, float* corr, double* red, float4* cs4) {
    const ConvGeom& g = c0.g;
    const int tid = threadIdx.x;
    int widx = blockIdx.x - c0.n_regular;
    const int kx = widx % g.kw;
    widx /= g.kw;
    const int ky = widx % g.kh;
    widx /= g.kh;
    const int cl = widx % g.Cl;
    const int cs = widx / g.Cl;
    const long long pi = c0.w_off + (blockIdx.x - c0.n_regular);
    int t = 0;
    if (tid < 2) t = st->adam_step + t_add;
    float w = 0.f, mi = 0.f, vi = 0.f;
    if (tid == 0) {
        w = p[pi];
        mi = m[pi];
        vi = v[pi];
    }
    const int hw = g.Hs * g.Ws, n = g.B * hw;
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)g.Ws;
    auto fetch = [&](int pp, float& sp, float& sq, float& lp) {
        const int b = div_small(pp, inv_hw), r = pp - b * hw;
        const int y = div_small(r, inv_w), x = r - y * g.Ws;
        const size_t so = ((size_t)(b * g.Cs + cs) * g.Hs + y) * g.Ws + x;
        sp = c0.small.p[so];
        sq = c0.small.q[so];
        lp = c0.xb[((size_t)(b * g.Cl + cl) * g.Hl + (y * g.s + ky)) * g.Wl + x * g.s + kx];
    };
    float sp[kAdamConv0Pos], sq[kAdamConv0Pos], lp[kAdamConv0Pos];
#pragma unroll
    for (int u = 0; u < kAdamConv0Pos; u++) fetch(min(tid + u * 256, n - 1), sp[u], sq[u], lp[u]);
#pragma unroll
    for (int u = 0; u < kAdamConv0Pos; u++) asm volatile("" : "+v"(sp[u]), "+v"(sq[u]), "+v"(lp[u]));
    bn_consts(c0.bns, cs4, false);
    if (tid < 2) {
        const double bc = -expm1((double)t * (tid ? ln_b2 : ln_b1));
        corr[tid] = tid ? (float)sqrt(bc) : (float)(h.lr / bc);
    }
    __syncthreads();
    const float4 ks = cs4[cs];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < kAdamConv0Pos; u++)
        if (tid + u * 256 < n) sum = fmaf(bn_apply(BN_BWD, ks, sp[u], sq[u]), lp[u], sum);
    for (int pp = tid + kAdamConv0Pos * 256; pp < n; pp += 256) {
        float a, b, c;
        fetch(pp, a, b, c);
        sum = fmaf(bn_apply(BN_BWD, ks, a, b), c, sum);
    }
    const double tot = block_sum((double)sum, red);
    if (tid == 0) {
        float gr = (float)(tot * c0.scale);
        if (h.wd != 0.0) gr = fmaf((float)h.wd, w, gr);
        mi = mi + (gr - mi) * (float)(1.0 - h.beta1);
        vi = vi * (float)h.beta2 + ((float)(1.0 - h.beta2) * gr) * gr;
        const float denom = sqrtf(vi) / corr[1] + (float)h.eps;
        p[pi] = w - corr[0] * (mi / denom);
        m[pi] = mi;
        v[pi] = vi;
    }
}

// `ln_b1`, `ln_b2`: natural logarithms of the betas (host).
// Order matters, this being the one kernel every step ends with: every thread requests its parameter, moments and gradient
// first; the bias corrections 1 - beta^t = -expm1(t ln beta) are worked out meanwhile by two lanes, one each (the two fp64
// pow() calls on one lane they replace were a dependent chain of a microsecond ahead of the barrier, with the loads not
// yet requested behind it).
__global__ void __launch_bounds__(256) k_adam(long long n, float* __restrict__ p, const float* __restrict__ g32,
                                               float* __restrict__ m, float* __restrict__ v, Hyper h,
                                               const StepState* __restrict__ st, ShardSegs ss, StepTail tl, int t_add,
                                               double ln_b1, double ln_b2, AdamConv0 c0) {
    kernarg_warm<sizeof(Hyper) + sizeof(ShardSegs) + sizeof(StepTail) + 72 + sizeof(AdamConv0)>();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    __shared__ float corr[2];
    if (c0.on && (int)blockIdx.x >= c0.n_regular) {   // uniform per workgroup
        __shared__ double red0[8];
        __shared__ float4 cs40[64];
        adam_conv0_body(c0, p, m, v, h, st, t_add, ln_b1, ln_b2, corr, red0, cs40);
        return;
    }
    // The step tail (step_tail's work) first.  No kernel of this step reads the BatchNorm sum tables any more, and nobody in
    // this kernel reads the cursor: cleared / moved up here, the stores complete in the shadow of the loads below.  Behind
    // the parameter update they cost a wait of their own: registers are reused there, and the compiler then waits for the
    // stores ahead of them (the memory counter retires in order).
    for (long long j = i; j < tl.zero_extra_n; j += (long long)(c0.on ? c0.n_regular : (int)gridDim.x) * 256) tl.zero_extra[j] = 0.0;
    int t = 0;
    if (threadIdx.x < 2) t = st->adam_step + t_add;
    const bool moves_cursor = i == 0 && tl.st != nullptr;
    long long cursor = 0;
    int slot = 0;
    if (moves_cursor) {
        cursor = tl.st->batch_start;
        slot = tl.st->loss_slot;
    }
    const bool mine = i < n && !(c0.on && i >= c0.w_off && i < c0.w_off + c0.nw);   // those weights: adam_conv0_body
    float g = 0.f, w = 0.f, mi = 0.f, vi = 0.f;
    // the fp64 gradient: its accumulator plus, for a parameter that is accumulated in shards, the eight shard copies.
    // Requested here, added up behind the barrier, cleared by the kernel's last stores (consume_grad's clearing stores sit
    // between its loads and their use, and where its paths meet the compiler waits for all but one of them to complete)
    double gacc = 0.0, sv[kStatShards], bv[kStatShards];
    double* shp = nullptr;
    const double* bnp = nullptr;
    if (mine) {
        w = p[i];
        mi = m[i];
        vi = v[i];
        if (g32) {
            g = g32[i];
        } else {
            gacc = tl.acc_rw[i];
            for (int sg = 0; sg < ss.nseg; sg++) {
                const long long d = i - ss.seg[sg].param_off;
                if (d >= 0 && d < ss.seg[sg].count) shp = tl.shard_rw + ss.seg[sg].sh_off + d;
            }
            if (shp) {
#pragma unroll
                for (int sh = 0; sh < kStatShards; sh++) sv[sh] = shp[(size_t)sh * ss.n];
            }
            if (c0.on) {   // gamma / beta of the fused layer: straight from its BatchNorm-backward sums (slots 3 / 2)
                const long long dg = i - c0.gamma_off, db = i - c0.beta_off;
                if (dg >= 0 && dg < c0.C) bnp = c0.stats + dg * 4 + 3;
                if (db >= 0 && db < c0.C) bnp = c0.stats + db * 4 + 2;
                if (bnp) {
#pragma unroll
                    for (int sh = 0; sh < kStatShards; sh++) bv[sh] = bnp[(size_t)sh * c0.C * 4];
                }
            }
        }
    }
    if (moves_cursor) {
        tl.st->batch_start = cursor + tl.batch_inc;
        tl.st->loss_slot = slot + tl.slot_inc;
    }
    if (threadIdx.x < 2) {
        const double bc = -expm1((double)t * (threadIdx.x ? ln_b2 : ln_b1));
        corr[threadIdx.x] = threadIdx.x ? (float)sqrt(bc) : (float)(h.lr / bc);
    }
    __syncthreads();
    if (mine) {
        const float step_size = corr[0];
        const float bc2_sqrt = corr[1];
        const float b2 = (float)h.beta2;
        if (!g32) {
            if (shp) {
#pragma unroll
                for (int sh = 0; sh < kStatShards; sh++) gacc += sv[sh];
            }
            if (bnp) {
                double t2 = 0.0;
#pragma unroll
                for (int sh = 0; sh < kStatShards; sh++) t2 += bv[sh];
                gacc += t2 * c0.scale;
            }
            g = (float)gacc;
        }
        if (h.wd != 0.0) g = fmaf((float)h.wd, w, g);
        mi = mi + (g - mi) * (float)(1.0 - h.beta1);
        vi = vi * b2 + ((float)(1.0 - h.beta2) * g) * g;
        const float denom = sqrtf(vi) / bc2_sqrt + (float)h.eps;
        p[i] = w - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
        if (!g32) {
            tl.acc_rw[i] = 0.0;
            if (shp) {
#pragma unroll
                for (int sh = 0; sh < kStatShards; sh++) shp[(size_t)sh * ss.n] = 0.0;
            }
        }
    }
}

__global__ void k_acc_to_f32(long long n, float* __restrict__ g, ShardSegs ss, StepTail tl) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    step_tail(tl, i, (long long)gridDim.x * 256);
    if (i < n) g[i] = consume_grad(tl, ss, i);
}

// data-parallel path: the fp64 accumulators of parameters [lo, hi) -> the fp32 gradient arena (consumed and cleared), one
// gradient bucket per launch; the launch for the last bucket also carries the step tail
__global__ void k_narrow_range(long long lo, long long hi, float* __restrict__ g, ShardSegs ss, StepTail tl) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tl.zero_extra || tl.st) step_tail(tl, t, (long long)gridDim.x * 256);
    const long long i = lo + t;
    if (i < hi) g[i] = consume_grad(tl, ss, i);
}

__global__ void k_fill_f32(float* __restrict__ p, long long n, float v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_scale_f32(float* __restrict__ p, long long n, float f) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] *= f;
}

__global__ void k_scale_f64(double* __restrict__ p, long long n, double f) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] *= f;
}

// the optimiser step counter for the data-parallel path, where Adam runs as its own op
__global__ void k_bump_adam(StepState* st) { st->adam_step += 1; }

__global__ void k_set_state(StepState* st, long long batch_start, int loss_slot, int set_cursor, int adam_step,
                            int set_adam) {
    if (set_cursor) {
        st->batch_start = batch_start;
        st->loss_slot = loss_slot;
    }
    if (set_adam) st->adam_step = adam_step;
}

__global__ void k_advance(StepState* st, int batch, int slot_inc, int adam_inc) {
    st->batch_start += batch;
    st->loss_slot += slot_inc;
    st->adam_step += adam_inc;
}

// ---------------------------------------------------------------------------------------------
// loader kernels (ds_dataset.py)
// ---------------------------------------------------------------------------------------------

// per-block partial {nan count, min, max}; out [gridDim.x][3] doubles
__global__ void __launch_bounds__(256) k_scan(const float* __restrict__ x, long long n, double* __restrict__ out) {
    __shared__ double red[3][4];
    double cnt = 0;
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        if (v != v) {
            cnt += 1;
        } else {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        lo = fminf(lo, __shfl_down(lo, off, 64));
        hi = fmaxf(hi, __shfl_down(hi, off, 64));
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wv] = cnt;
        red[1][wv] = lo;
        red[2][wv] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0, l = INFINITY, h = -INFINITY;
        for (int i = 0; i < 4; i++) {
            c += red[0][i];
            l = fmin(l, red[1][i]);
            h = fmax(h, red[2][i]);
        }
        out[3 * blockIdx.x + 0] = c;
        out[3 * blockIdx.x + 1] = l;
        out[3 * blockIdx.x + 2] = h;
    }
}

// dst[row(i)][c_off + c][:] = (src[i][c][:] - vmin) / range   (two correctly rounded fp32 ops, as numpy)
// row(i) = dst_row[i] when a table is given (the frozen shuffle's inverse: normalisation writes the samples in batch order
// in the pass it makes anyway - the reference's DataLoader + collate stacking, conv_ae_model.py:291-292,315-325), else i.
__global__ void __launch_bounds__(256) k_normalise_pack(const float* __restrict__ src, long long total, int c_src,
                                                         long long hw, float* __restrict__ dst, int c_dst, int c_off,
                                                         float vmin, float range, int enable,
                                                         const int* __restrict__ dst_row) {
    const long long per = (long long)c_src * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long s = i / per, r = i - s * per;
        float v = src[i];
        if (enable) v = (range == 0.f) ? 0.f : __fdiv_rn(__fsub_rn(v, vmin), range);
        const long long d = dst_row ? (long long)dst_row[s] : s;
        dst[(d * c_dst + c_off) * hw + r] = v;
    }
}

// inv[perm[i]] = i: the destination-row table of k_normalise_pack from a frozen sample order
__global__ void __launch_bounds__(256) k_invert_perm(const int* __restrict__ perm, long long n, int* __restrict__ inv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int p = perm[i];
        if (p >= 0 && p < n) inv[p] = (int)i;
    }
}

__global__ void __launch_bounds__(256) k_denorm_f64(const float* __restrict__ y, long long n, double vmin, double range,
                                                     double* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = __dadd_rn(vmin, __dmul_rn((double)y[i], range));
}

// model_metric.py:47-71 per instance: sums[inst][8] += {n, Σa', Σe', Σa'², Σe'², Σa'e', Σ|a-e|, Σ(a-e)²} over the
// pixels whose mask is non-zero, with e = vmin + (double)y*range (the denormalised score of base_model.py:90) and
// a' = a - vmin, e' = e - vmin (shifted so the second moments do not cancel).  grid (chunks, n_inst), block 256.
__global__ void __launch_bounds__(256) k_metric_sums(const float* __restrict__ y, const float* __restrict__ a,
                                                      const float* __restrict__ mask, long long elems, double vmin,
                                                      double range, double* __restrict__ sums) {
    __shared__ double red[4];
    const long long base = (long long)blockIdx.y * elems;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < elems; i += (long long)gridDim.x * 256) {
        if (mask && mask[base + i] == 0.f) continue;
        const double e = __dadd_rn(vmin, __dmul_rn((double)y[base + i], range));
        const double av = (double)a[base + i];
        const double d = av - e, as = av - vmin, es = e - vmin;
        acc[0] += 1.0;
        acc[1] += as;
        acc[2] += es;
        acc[3] += as * as;
        acc[4] += es * es;
        acc[5] += as * es;
        acc[6] += fabs(d);
        acc[7] += d * d;
    }
    for (int k = 0; k < 8; k++) {
        const double t = block_sum(acc[k], red);
        if (threadIdx.x == 0 && t != 0.0) atomicAdd(&sums[(size_t)blockIdx.y * 8 + k], t);
    }
}

// NetCDF-3 stores big-endian words: swap n 32-bit words in place (16 bytes per lane per trip)
__global__ void __launch_bounds__(256) k_bswap32(unsigned* __restrict__ x, long long n) {
    const long long n4 = n >> 2;
    uint4* x4 = reinterpret_cast<uint4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        uint4 v = x4[i];
        v.x = __builtin_bswap32(v.x);
        v.y = __builtin_bswap32(v.y);
        v.z = __builtin_bswap32(v.z);
        v.w = __builtin_bswap32(v.w);
        x4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] = __builtin_bswap32(x[(n4 << 2) + threadIdx.x]);
}

}  // namespace cae
