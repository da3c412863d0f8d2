// kernels_ctlds.h — LDS-staged implicit GEMM on the fp32 matrix cores for the channel-rich stride-2 ConvTranspose2d
// layers at the head of the decoder (decoder.py:44-48; 64->32, 32->16, 16->8 channels on 3x3 .. 31x31 maps at the benchmark
// geometry), forward pass.  Successor of k_ig_fwd_s2 (kernels_igemm.h), which gathered every MFMA operand per lane from
// global memory and re-applied BatchNorm+ReLU per use: 53 vector instructions per MFMA, bound by instruction issue.
//
// Here a workgroup owns one image (or a band of its output) and one block of 16 output channels:
//   1. coalesced global reads -> LDS:  the image, BatchNorm+ReLU of the producer applied ONCE per element, inside a zero
//      border (so a tap outside the map needs no predicate), and the weight slice [Cin][16 co][kh*kw] as stored;
//   2. v_mfma_f32_16x16x4_f32 with both operands read from LDS by one ds_read_b32 each at loop-invariant lane offsets
//      (no address arithmetic in the loop: the channel stride is an immediate offset);
//   3. bias, stores and the BatchNorm sums of THIS layer in the epilogue.
// GEMM view (sub-pixel decomposition): output pixel (2m+py, 2n+px) only sees inputs (m-j, n-i), j,i in {0,1}, through taps
// (py+2j, px+2i) < (kh, kw).  Per output parity (py,px):  out[(m,n)][co] = sum_{ci,(j,i)} a[ci][m-j][n-i] W[ci][co][py+2j][px+2i]
// with n_p = nj*ni in {1,2,4} valid taps, so one MFMA k-step (4 k) covers 4/n_p input channels x n_p taps and a lane's tap
// and channel offset are loop invariants.  Rows of a tile = 16 consecutive quads (m,n) of one image, columns = 16 output
// channels; a wave keeps the four parities' accumulators and walks its share of the input channels (K split over ks waves,
// combined through LDS).  MFMAs per tile: Cin * (sum of n_p) / 4 = 2.25 Cin for a 3x3 kernel.
#pragma once
#include "kernels_gemm.h"

namespace cae {

struct CtFwd {
    int B, Cin, H, W, Cout, OH, OW, QH, QW;
    int PW, plane;   // padded image in LDS: rows -1 .. QH-1 ((QH+1) rows) of PW = QW+1 columns; plane = odd size of one channel
    int tiles;       // row tiles (16 quads) per image
    int rt;          // row tiles per workgroup
    int ks;          // waves that share a row tile and split the input channels; blockDim = 64 * rt * ks
    int tg;          // tile groups per image = ceil(tiles / rt)
    const float* in;
    BnDesc bn_in;
    const float* w;
    const float* bias;
    float* out;
    double* stats;   // [shards][Cout][4] or nullptr (eval)
    long long* dbg;  // diagnostics (tools/ct_phases.py): 8 wall-clock stamps per workgroup (first 384), or nullptr
};

template <int KH, int KW>
struct CtShape {
    static constexpr int KK = KH * KW;
    static constexpr int KKp = KK | 1;       // odd stride between output channels: their taps land on distinct banks
    static constexpr int WS = 16 * KKp;      // floats per input channel of the staged weight slice
};

// one parity's operand loads for 4 input channels starting at channel c (relative to the lane's base pointers)
template <int NP>
__device__ __forceinline__ void ct_load4(const float* __restrict__ ap, const float* __restrict__ bp, int c, int plane, int ws,
                                         float* av, float* bv) {
    constexpr int CSTEP = 4 / NP;   // channels per MFMA
#pragma unroll
    for (int u = 0; u < NP; u++) {
        av[u] = ap[(c + u * CSTEP) * plane];
        bv[u] = bp[(c + u * CSTEP) * ws];
    }
}

// grid (B * tg, ceil(Cout / 16)), block 64 * rt * ks, dynamic LDS = ct_fwd_lds_bytes(...)
constexpr int kCtImgRegs = 12;   // image elements a thread holds between the global loads and the LDS scatter

template <int KH, int KW>
__global__ void __launch_bounds__(512) k_ct_fwd_lds(CtFwd a) {
    using S = CtShape<KH, KW>;
    constexpr int KK = S::KK, KKp = S::KKp, WS = S::WS;
    // valid taps per parity: ky = py + 2j < KH
    constexpr int NJ0 = (KH + 1) / 2, NJ1 = KH / 2, NI0 = (KW + 1) / 2, NI1 = KW / 2;
    constexpr int NP00 = NJ0 * NI0, NP01 = NJ0 * NI1, NP10 = NJ1 * NI0, NP11 = NJ1 * NI1;
    static_assert(KH >= 2 && KH <= 4 && KW >= 2 && KW <= 4, "stride-2 kernels of 2..4 taps per axis");

#define CT_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 384) a.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
    kernarg_warm<sizeof(CtFwd)>();
    extern __shared__ double lds_d[];
    CT_STAMP(0);
    double* lstat = lds_d;                                          // [16 channels][2]
    float4* cin4 = reinterpret_cast<float4*>(lstat + 32);           // [Cin]
    float* wl = reinterpret_cast<float*>(cin4 + a.Cin);             // [Cin][16][KKp]
    float* img = wl + a.Cin * WS;                                   // [Cin][plane]
    float* part = img + ((a.Cin * a.plane + 3) & ~3);               // [waves][16 regs][64 lanes]
    // (not blockDim.x: the compiler fetches that from the hidden kernel arguments with a per-lane load, and every address
    // below waits for it)
    const int tid = threadIdx.x, nthr = 64 * a.rt * a.ks;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), and the tg tile groups and
    // ceil(Cout / 16) channel blocks of an image all stage that image.  With the ids in plain (image, group) order the groups of
    // one image sat on tg different XCDs and every L2 fetched it (the 16->8 layer at the benchmark size: 6.1 MB of traffic for
    // 2.9 algorithmic MB).  Dealt so that id % 8 == image % 8, an image is fetched into ONE L2 and its other stagings hit there.
    int b, g;
    if ((a.B & 7) == 0) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int kb = k / a.tg;
        b = xcd + 8 * kb;
        g = k - kb * a.tg;
    } else {
        b = blockIdx.x / a.tg;
        g = blockIdx.x - b * a.tg;
    }
    const int cb = blockIdx.y;

    // Every global read of the prologue is issued before anything waits: the image (in source order: coalesced, no
    // predicates), then the weight slice, then the BatchNorm sums inside bn_consts.  A load under a predicate is a branch
    // around the load plus a wait at its consumer - one memory round trip per loop iteration (3.6 us for nine of them).
    const int HW = a.H * a.W, n_img = a.Cin * HW;
    const float* src = a.in + (size_t)b * n_img;
    float iv[kCtImgRegs];
#pragma unroll
    for (int u = 0; u < kCtImgRegs; u++) iv[u] = src[min(tid + u * nthr, n_img - 1)];

    // the producer's BatchNorm sums (the usual mode: batch statistics, one thread per input channel) are requested here, with
    // the image and ahead of the weights, and turned into constants behind the weight stores: one wait covers all three
    const bool bn_designated = blockIdx.x == 0 && blockIdx.y == 0;
    const bool bn_split = a.bn_in.mode == BN_BATCH && a.bn_in.C <= nthr;   // uniform
    const bool bn_mine = bn_split && tid < a.bn_in.C, bn_upd = bn_designated && a.bn_in.update;
    BnBatchReq bnrq;
    if (bn_mine) bn_batch_request(a.bn_in, tid, bn_upd, bnrq);

    const int ncol = min(16, a.Cout - cb * 16);
    const float* wsrc = a.w + (size_t)cb * 16 * KK;
    const int wstride = a.Cout * KK;
    const bool vec = KKp == KK && ((ncol * KK) & 3) == 0 && (wstride & 3) == 0;
    if (vec) {
        // [ci][co][tap] rows are contiguous runs of ncol * KK floats, 16-byte aligned on both sides
        const int n4 = ncol * KK / 4, total4 = a.Cin * n4;
        const float inv_n4 = 1.0f / (float)n4;
        for (int i0 = tid; i0 < total4; i0 += 6 * nthr) {
            f32x4 wv4[6];
            int dst[6];
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int idx = min(i0 + u * nthr, total4 - 1);
                const int ci = div_small(idx, inv_n4), jq = idx - ci * n4;
                wv4[u] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)ci * wstride + 4 * jq);
                dst[u] = ci * WS + 4 * jq;
            }
            // keeps the compiler from sinking each load into its conditional store below (load, wait, store, six times over:
            // six trips to memory one after the other)
            asm volatile("" : "+v"(wv4[0]), "+v"(wv4[1]), "+v"(wv4[2]), "+v"(wv4[3]), "+v"(wv4[4]), "+v"(wv4[5]));
#pragma unroll
            for (int u = 0; u < 6; u++)
                if (i0 + u * nthr < total4) *reinterpret_cast<f32x4*>(wl + dst[u]) = wv4[u];
        }
        if (ncol < 16) {   // columns past Cout: zero B operands
            const int tail = WS - ncol * KK;
            for (int idx = tid; idx < a.Cin * tail; idx += nthr) {
                const int ci = idx / tail;
                wl[ci * WS + ncol * KK + (idx - ci * tail)] = 0.f;
            }
        }
    }
    if (bn_split) {
        if (bn_mine) bn_batch_finish(a.bn_in, tid, bn_upd, bnrq, cin4);
    } else {
        bn_consts(a.bn_in, cin4, bn_designated);
    }
    // (the general weight path comes after everything the common one issues: placed ahead of it, its loads - never
    // executed then - still count as pending where the paths meet, and the common path waits for the image before it
    // requests the weights)
    if (!vec) {
        for (int i0 = tid; i0 < a.Cin * WS; i0 += 8 * nthr) {
            float wv1[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = min(i0 + u * nthr, a.Cin * WS - 1);
                const int ci = idx / WS, rem = idx - ci * WS;
                const int col = rem / KKp, tap = rem - col * KKp;
                ok[u] = col < ncol && tap < KK;
                wv1[u] = wsrc[ok[u] ? ci * wstride + col * KK + tap : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * nthr < a.Cin * WS) wl[i0 + u * nthr] = ok[u] ? wv1[u] : 0.f;
        }
    }
    if (tid < 32) lstat[tid] = 0.0;
    {   // zero the padded image (its border stays zero: a tap outside the map then needs no predicate)
        const int n4 = (a.Cin * a.plane + 3) >> 2;
        for (int idx = tid; idx < n4; idx += nthr) reinterpret_cast<float4*>(img)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    CT_STAMP(1);
    {   // the image: BatchNorm + ReLU once per element, scattered into the zero border
        const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.W;
        const bool bn = a.bn_in.mode != BN_NONE;
#pragma unroll
        for (int u = 0; u < kCtImgRegs; u++) {
            const int idx = tid + u * nthr;
            if (idx < n_img) {
                const int ci = div_small(idx, inv_hw), pos = idx - ci * HW;
                const int iy = div_small(pos, inv_w), ix = pos - iy * a.W;
                float v = iv[u];
                if (bn) {
                    const float4 k = cin4[ci];
                    v = fmaxf(0.f, fmaf(v - k.x, k.y, k.z));
                }
                img[ci * a.plane + (iy + 1) * a.PW + ix + 1] = v;
            }
        }
        for (int idx = tid + kCtImgRegs * nthr; idx < n_img; idx += nthr) {   // images beyond the register batch (rare)
            const int ci = div_small(idx, inv_hw), pos = idx - ci * HW;
            const int iy = div_small(pos, inv_w), ix = pos - iy * a.W;
            float v = src[idx];
            if (bn) {
                const float4 k = cin4[ci];
                v = fmaxf(0.f, fmaf(v - k.x, k.y, k.z));
            }
            img[ci * a.plane + (iy + 1) * a.PW + ix + 1] = v;
        }
    }
    __syncthreads();
    CT_STAMP(2);

    const int wv = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int tl = wv / a.ks, kslot = wv - tl * a.ks;
    const int tile = g * a.rt + tl;
    const bool tile_ok = tile < a.tiles;                         // uniform per wave
    const int Q = a.QH * a.QW;
    const float inv_qw = 1.0f / (float)a.QW;
    const int qi = tile * 16 + r;
    const int qic = (tile_ok && qi < Q) ? qi : 0;                // idle rows read quad 0 (never stored)
    const int qy = div_small(qic, inv_qw), qx = qic - qy * a.QW;
    const int cper = a.Cin / a.ks, c0 = kslot * cper;            // host: cper % 4 == 0
    const int co = cb * 16 + r;
    float bias = a.bias[min(co, a.Cout - 1)];                    // requested before the loop, used after it

    f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
    if (tile_ok) {
        // lane invariants per parity: k-slot q -> (channel offset dci, tap (j,i)); operand pointers at channel c0
#define CT_PTRS(NP, NI, PY, PX, AP, BP)                                                              \
        const float* AP;                                                                             \
        const float* BP;                                                                             \
        {                                                                                            \
            const int t = q % (NP), dci = q / (NP), j = t / (NI), i = t - j * (NI);                   \
            AP = img + (c0 + dci) * a.plane + (qy - j + 1) * a.PW + (qx - i + 1);                    \
            BP = wl + (c0 + dci) * WS + r * KKp + ((PY) + 2 * j) * KW + (PX) + 2 * i;                \
        }
        CT_PTRS(NP00, NI0, 0, 0, ap00, bp00)
        CT_PTRS(NP01, NI1, 0, 1, ap01, bp01)
        CT_PTRS(NP10, NI0, 1, 0, ap10, bp10)
        CT_PTRS(NP11, NI1, 1, 1, ap11, bp11)
#undef CT_PTRS
        for (int c = 0; c < cper; c += 4) {
            float a00[NP00], b00[NP00], a01[NP01], b01[NP01], a10[NP10], b10[NP10], a11[NP11], b11[NP11];
            ct_load4<NP00>(ap00, bp00, c, a.plane, WS, a00, b00);
            ct_load4<NP01>(ap01, bp01, c, a.plane, WS, a01, b01);
            ct_load4<NP10>(ap10, bp10, c, a.plane, WS, a10, b10);
            ct_load4<NP11>(ap11, bp11, c, a.plane, WS, a11, b11);
            // round-robin over the four accumulators: a dependent 16x16x4 MFMA waits 40 cycles, an independent one 32
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u < NP00) acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[u], b00[u], acc00, 0, 0, 0);
                if (u < NP01) acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[u], b01[u], acc01, 0, 0, 0);
                if (u < NP10) acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10[u], b10[u], acc10, 0, 0, 0);
                if (u < NP11) acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[u], b11[u], acc11, 0, 0, 0);
            }
        }
    }
    CT_STAMP(3);
    // The bias is waited for HERE, once, outside the predicated stores below.  Left to its first use - inside a predicated
    // block, which the lanes that skip it leave with the load still formally pending - the compiler repeats the wait in every
    // one of the sixteen blocks, and from the second on that wait is for the previous block's STORE to complete (the memory
    // counter retires in order): sixteen store round trips one after the other.
    asm volatile("" : "+v"(bias));

    // Epilogue.  Register `reg` = 4 * parity + jj of lane (q, r) is quad row 4q + jj of the tile, output channel r.
    float s1 = 0.f, s2 = 0.f;
    float* obase = a.out + (size_t)(b * a.Cout + min(co, a.Cout - 1)) * a.OH * a.OW;
    const bool col_ok = tile_ok && co < a.Cout;
    auto emit = [&](int reg, float acc) {
        const int p = reg >> 2, jj = reg & 3;
        const int qo = tile * 16 + q * 4 + jj;
        const int oqy = div_small(min(qo, Q - 1), inv_qw), oqx = min(qo, Q - 1) - oqy * a.QW;
        const int oy = 2 * oqy + (p >> 1), ox = 2 * oqx + (p & 1);
        if (col_ok && qo < Q && oy < a.OH && ox < a.OW) {
            const float v = acc + bias;
            obase[oy * a.OW + ox] = v;
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    };
    if (a.ks > 1) {
        // The ks waves of a row tile each hold a partial of all 16 registers.  Every wave parks its partials in LDS and then
        // owns 16 / ks of the registers: it sums those over the ks slices and stores them - combine, stores and statistics
        // spread over all waves instead of queueing behind slice 0.
        float* pw = part + wv * 1024 + lane;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            pw[(0 + jj) * 64] = acc00[jj];
            pw[(4 + jj) * 64] = acc01[jj];
            pw[(8 + jj) * 64] = acc10[jj];
            pw[(12 + jj) * 64] = acc11[jj];
        }
        __syncthreads();
        CT_STAMP(4);
        const int rpw = 16 / a.ks;
        const float* pt = part + (tl * a.ks) * 1024 + lane;
        for (int rl = 0; rl < rpw; rl++) {
            const int reg = kslot * rpw + rl;
            float v = 0.f;
            for (int o = 0; o < a.ks; o++) v += pt[o * 1024 + reg * 64];
            emit(reg, v);
        }
    } else {
        CT_STAMP(4);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            emit(0 + jj, acc00[jj]);
            emit(4 + jj, acc01[jj]);
            emit(8 + jj, acc10[jj]);
            emit(12 + jj, acc11[jj]);
        }
    }
    CT_STAMP(5);
    if (a.stats) {
        // a lane's fp32 sums cover at most 16 values; from there on fp64: lanes r, r+16, r+32, r+48 hold the same channel
        // (fold), waves meet in LDS, then one fp64 atomic per value and workgroup
        double d1 = (double)s1, d2 = (double)s2;
        d1 += __shfl_xor(d1, 16, 64); d2 += __shfl_xor(d2, 16, 64);
        d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
        if (q == 0 && col_ok) {
            atomicAdd(&lstat[2 * r], acc_grid<ACC_STAT>(d1));       // (on the grid: exact, hence order-independent, adds)
            atomicAdd(&lstat[2 * r + 1], acc_grid<ACC_STAT>(d2));
        }
        __syncthreads();
        if (tid < 32) {
            const int c = cb * 16 + (tid >> 1);
            if (c < a.Cout) {
                const int shard = (blockIdx.x + blockIdx.y) & (kStatShards - 1);
                acc_add<ACC_STAT>(&a.stats[((size_t)shard * a.Cout + c) * 4 + (tid & 1)], lstat[tid]);
            }
        }
    }
    CT_STAMP(6);
#undef CT_STAMP
}

inline size_t ct_fwd_lds_bytes(int Cin, int plane, int KH, int KW, int waves, int ks) {
    const int KKp = (KH * KW) | 1;
    size_t floats = 64 + 4 * (size_t)Cin + (size_t)Cin * 16 * KKp + (((size_t)Cin * plane + 3) & ~(size_t)3);
    if (ks > 1) floats += (size_t)waves * 1024;
    return floats * sizeof(float);
}

}  // namespace cae
