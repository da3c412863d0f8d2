// kernels_last.h — the LAST decoder layer of a training step as ONE launch: ConvTranspose2d (stride 2) forward, sigmoid
// (decoder.py:77), MSELoss (conv_ae_model.py:303) and the layer's whole backward (input gradient masked by the producer's
// ReLU + the producer's BatchNorm-backward sums, weight gradient, bias gradient).
//
// Why one launch: there is no BatchNorm behind the last layer, so nothing batch-wide separates its forward from its backward.
// As two launches (k_s2_fwd2 + k_s2_bwd2) the gradient map dL/d(pre-sigmoid) — as large as the output, 16.8 MB at the
// benchmark geometry — was written by the first and read back by the second: 75 MB of traffic and two launches for a layer
// whose inputs are 8.3 MB of activations and 16.8 MB of targets.  Here that map only ever exists in registers.
//
// Sub-pixel view (kernels_s2.h): output quad (m, n) = the 2x2 outputs (2m+py, 2n+px); it sees the inputs a[m-j][n-i], j,i in
// {0,1}, through the taps (py+2j, px+2i).  The SAME sixteen (input, output) pairs of a quad are all of its backward:
//     dW[ci][py+2j][px+2i] += a[ci][m-j][n-i] * g[py][px]                                  (weight gradient)
//     P[ci][j][i]           = sum_{py,px} g[py][px] * W[ci][py+2j][px+2i]  ->  pixel (m-j, n-i) (input-gradient share)
// and the input gradient of pixel (y, x) is P[0][0](y,x) + P[0][1](y,x+1) + P[1][0](y+1,x) + P[1][1](y+1,x+1).
//
// Mapping: a wave owns a strip of 128 quad columns (lane l: quads 2l, 2l+1 = four output columns = one 16-byte target load
// per row) and walks down a band of HB quad rows.  The previous row's inputs and the j = 0 shares travel to the next row in
// registers; the i = 1 shares of a lane's first quad go to its left neighbour by one DPP wave shift.  No LDS, no barrier, no
// atomics inside the walk.  Only the last pixel column of a strip and the last pixel row of a band need a quad of the
// neighbouring strip / band: strips overlap by one quad column (127 pixel columns per strip: exactly the 127-wide input of
// the benchmark geometry, so no overlap there) and bands by one quad row, recomputed without being counted (loss, bias and
// weight gradients are counted by the strip / band that owns the quad).
// Every global load of a band is issued before its first use (HB+1 rows x 12 registers) and the rows are consumed as they
// arrive: the whole launch's reads are in flight within the first microsecond.
#pragma once
#include <utility>
#include "kernels_s2.h"

namespace cae {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a row loop whose index is a constant at the SOURCE level.
// (#pragma unroll leaves a loop with an early exit around DPP operations - convergent - rolled until after the last
// scalar-replacement pass, and the register arrays it indexes then live in scratch memory.)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct S2Last {
    int B, H, W, OH, OW;
    int QH, QW;              // quad rows / columns of the output
    int strips, bands;       // per image
    int total;               // B * strips * bands wave work items
    const float* in;         // (B, CIN, H, W) raw output of the producer
    BnDesc bn_in;            // BN_BATCH of the producer (the designated workgroup also saves mean/invstd + running stats), or BN_NONE
    const float* w;          // (CIN, COUT, KH, KW)
    const float* bias;       // (COUT)
    const float* target;     // dataset targets (N, COUT, OH, OW)
    const int* perm;
    int use_cursor;
    const StepState* st;
    double* losses;
    float inv_count;         // 1 / (global_batch * COUT * OH * OW)
    double* bias_acc;        // sharded [kStatShards][acc_stride]
    double* wacc;            // sharded [kStatShards][acc_stride]
    int acc_stride;
    float* gin;              // (B, CIN, H, W) gradient wrt the producer's raw output side (masked by its ReLU)
    double* stats_in;        // producer's [kStatShards][CIN][4] sums (slots 2, 3) or nullptr
    long long* dbg;          // diagnostics (tools/last_phases.py): 8 wall-clock stamps per workgroup (first 384), or nullptr
};

constexpr int kLastStripPx = 127;   // pixel columns a strip owns (it computes 128 quad columns)

__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// lane i <- lane i-1 (lane 0 keeps `edge`) / lane i <- lane i+1 (lane 63 keeps `edge`)
__device__ __forceinline__ float from_left(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v, float edge) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// BN: the input carries the producer's BatchNorm + ReLU (always, unless the decoder has a single layer)
// Sums over the wave of N per-lane values, written to out[0..N) (LDS).  Through LDS in chunks of 16 values instead of six
// DPP steps per value: the lanes write a chunk as [value][lane], then lane 4v + q reads a quarter of row v (sixteen floats,
// four 16-byte reads) and two DPP steps join the quarters.  Per value ~2.5 instructions instead of ~8 (+ the DPP hazard
// no-ops): the DPP form cost the row-streaming kernels 4.3 us of a 10.8 us workgroup lifetime (76 values per lane).
// A wave's LDS operations execute in order, so the write -> read -> write sequence on `scr` needs no barrier; `scr` is this
// wave's own kWsumScratch floats.
constexpr int kWsumLd = 68;                       // row stride: 16-byte aligned, rows 4 banks apart
constexpr int kWsumScratch = 16 * kWsumLd;
template <int N>
__device__ __forceinline__ void wave_sums_lds(const float (&v)[N], float* scr, float* out, int lane) {
    const int row = lane >> 2, qd = lane & 3;
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (c0 + i < N) scr[i * kWsumLd + lane] = v[c0 + i];
        const float4* p = reinterpret_cast<const float4*>(scr + row * kWsumLd + 4 * qd);   // floats 4 qd + 16 k + {0..3}
        const float4 a0 = p[0], a1 = p[4], a2 = p[8], a3 = p[12];
        float t = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) + ((a2.x + a2.y) + (a2.z + a2.w)) +
                  ((a3.x + a3.y) + (a3.z + a3.w));
        t += dpp_f<0xB1>(t);        // quad_perm [1,0,3,2]
        t += dpp_f<0x4E>(t);        // quad_perm [2,3,0,1]
        if (qd == 0 && c0 + row < N) out[c0 + row] = t;
    }
}

// VEC4: output rows are 16-byte aligned and there is one strip (OW % 4 == 0, strips == 1): 16-byte target loads
template <int CIN, int COUT, int KH, int KW, int HB, bool VEC4, bool BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_s2_last_fused(S2Last a) {
    kernarg_warm<sizeof(S2Last)>();
    constexpr int NACC = CIN * COUT * KH * KW;
    constexpr int NRED = NACC + 2 * CIN;
    constexpr int NR = HB + 1;            // quad rows a wave computes: its band + the first row of the next one
    static_assert(KH >= 3 && KH <= 4 && KW >= 3 && KW <= 4, "3- or 4-tap stride-2 kernels");
    __shared__ float redf[4 * NRED];
    __shared__ double redd[4 * 2 * COUT];
    __shared__ __attribute__((aligned(16))) float wscr[4 * kWsumScratch];
    __shared__ float xch[4 * CIN * 2 * 64];      // first-row shares of the four bands of a workgroup

#define LF_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.x < 384) a.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    LF_STAMP(0);
    const unsigned HW = a.H * a.W, OHW = a.OH * a.OW;

    // ---- this wave's band
    const int item = blockIdx.x * 4 + wv;
    const bool work = item < a.total;
    const int it = work ? item : 0;
    const int b = it / (a.strips * a.bands);
    const int rem = it - b * (a.strips * a.bands);
    const int band = rem / a.strips, strip = rem - band * a.strips;
    const int m0 = band * HB;
    const bool last_band = band == a.bands - 1, last_strip = strip == a.strips - 1;
    const int n0 = strip * kLastStripPx + 2 * lane;      // first of the lane's two quad columns = its first pixel column
    const bool multi = a.strips > 1;
    // With one strip per image the four waves of a workgroup are consecutive bands: a band whose neighbour below is in the
    // same workgroup (and the same image) gets that neighbour's first-row shares through LDS instead of recomputing its row
    const bool has_below = !multi && work && wv < 3 && band + 1 < a.bands;
    const int nrows = has_below ? HB : NR;
    const bool all_cols = VEC4 && 256 <= a.OW;           // uniform: every lane's four output columns are inside the map

    // ---- loads, in the order their consumers can use them.  The wave's memory counter retires in issue order, so whatever a
    // value's consumer waits for, it waits for everything issued before it: the small dependent reads (cursor -> permutation
    // -> target row; BatchNorm sums; weights) go first and the band's bulk behind them, never the other way round.
    // (zero offsets hidden from the compiler in a vector register: a load it knows to be wave-uniform is moved to a scalar
    // register - and waited for - where it is issued; the cursor would cost a trip to memory with nothing else in flight)
    int lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    long long bs_v = 0;
    const bool indirect = a.perm != nullptr || a.use_cursor != 0;
    if (indirect) bs_v = (&a.st->batch_start)[lane_zero];
    // (the loss slot rides along: read where the loss atomic is issued it is one more trip to memory at the very end of the
    // workgroup's last lane)
    const int loss_slot_v = (&a.st->loss_slot)[lane_zero];
    // BatchNorm sums of the producer: lane (8 c + shard) reads that shard's {sum y, sum y^2} of channel c (one 16-byte load)
    double sa = 0.0, sb = 0.0;
    float gam[CIN], bet[CIN], rmn[CIN], rvr[CIN];
    if constexpr (BN) {
        static_assert(CIN <= 8, "one lane per (channel, shard)");
        if (lane < 8 * CIN) {
            const double2 t = *reinterpret_cast<const double2*>(a.bn_in.stats + ((size_t)(lane & 7) * CIN + (lane >> 3)) * 4);
            sa = t.x;
            sb = t.y;
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            gam[ci] = a.bn_in.gamma[ci];
            bet[ci] = a.bn_in.beta[ci];
            rmn[ci] = a.bn_in.rmean[ci];
            rvr[ci] = a.bn_in.rvar[ci];
        }
    }
    // weights and biases: wave-uniform; moved to scalar registers below (the compiler will not use scalar loads for memory the
    // kernel's own stores might alias; left to itself it keeps them in vector registers)
    float wk[NACC], bk[COUT];
#pragma unroll
    for (int i = 0; i < NACC; i++) wk[i] = a.w[i];
#pragma unroll
    for (int i = 0; i < COUT; i++) bk[i] = a.bias[i];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the permutation entry is requested here and waited for behind the input loads (branch-free: without a permutation the
    // load reads a dummy word of the step state)
    // (uniform branch: only this path waits for the cursor here.  Left unset otherwise, on purpose: writing the register in
    // the other path makes the compiler drain every load in flight first, since a load may be pending on it)
    int perm_val_v;
    if (a.perm) perm_val_v = a.perm[bs_v + b + lane_zero];
    // Inputs: rows m0-1 .. m0+HB, columns n0, n0+1 (+ column n0-1 for lane 0 of a later strip); clamped addresses, validity
    // by select.  Targets: output rows 2m, 2m+1 of every quad row, four columns from 2 n0.
    constexpr int NLEFT = VEC4 ? 1 : NR + 1;   // one strip: nothing to the left of lane 0
    float rin[NR + 1][CIN][2], rleft[NLEFT][CIN];
    float4 tgt[NR][COUT][2];
    {
        const float* inb = a.in + (size_t)b * CIN * HW;
        const int c0 = min(n0, a.W - 1), c1 = min(n0 + 1, a.W - 1), cl = min(max(n0 - 1, 0), a.W - 1);
#pragma unroll
        for (int r = 0; r <= NR; r++) {
            const unsigned ro = (unsigned)min(max(m0 - 1 + r, 0), a.H - 1) * a.W;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                rin[r][ci][0] = inb[ci * HW + ro + c0];
                rin[r][ci][1] = inb[ci * HW + ro + c1];
                if constexpr (!VEC4) {
                    rleft[r][ci] = 0.f;
                    if (multi && lane == 0) rleft[r][ci] = inb[ci * HW + ro + cl];
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // (the cursor was the first load: a counted wait, the inputs stay in flight)
        const long long bs = ((long long)__builtin_amdgcn_readfirstlane((int)(bs_v >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)bs_v);
        const long long tb = a.perm ? (long long)__builtin_amdgcn_readfirstlane(perm_val_v) : bs + b;   // bs = 0 without a cursor
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (r >= nrows) break;       // uniform: the band below reads that row (straight-line code: no DPP in this loop)
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                const float* plane = a.target + ((size_t)tb * COUT + co) * (size_t)OHW;
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    const unsigned ro = (unsigned)min(2 * (m0 + r) + py, a.OH - 1) * a.OW;
                    const int ox = 2 * n0;
                    if constexpr (VEC4) {
                        tgt[r][co][py] = *reinterpret_cast<const float4*>(plane + ro + min(ox, a.OW - 4));
                    } else {
                        tgt[r][co][py].x = plane[ro + min(ox, a.OW - 1)];
                        tgt[r][co][py].y = plane[ro + min(ox + 1, a.OW - 1)];
                        tgt[r][co][py].z = plane[ro + min(ox + 2, a.OW - 1)];
                        tgt[r][co][py].w = plane[ro + min(ox + 3, a.OW - 1)];
                    }
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    LF_STAMP(1);
#pragma unroll
    for (int i = 0; i < NACC; i++) wk[i] = uniform_f(wk[i]);
#pragma unroll
    for (int i = 0; i < COUT; i++) bk[i] = uniform_f(bk[i]);
    // BatchNorm constants {mean, gamma * invstd, beta, invstd} inside the wave (bn_consts' arithmetic): the eight shards of a
    // channel sit in eight neighbouring lanes - three DPP steps - and lane 8 c hands the sums to everybody
    float4 kc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ci++) kc[ci] = make_float4(0.f, 1.f, 0.f, 1.f);
    if constexpr (BN) {
        sa += dpp_d<0xB1>(sa); sb += dpp_d<0xB1>(sb);       // quad_perm [1,0,3,2]
        sa += dpp_d<0x4E>(sa); sb += dpp_d<0x4E>(sb);       // quad_perm [2,3,0,1]
        sa += dpp_d<0x141>(sa); sb += dpp_d<0x141>(sb);     // row_half_mirror: every lane of an 8-lane group holds the group's sum
        const long long ba = __builtin_bit_cast(long long, sa), bb = __builtin_bit_cast(long long, sb);
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            const int lo1 = __builtin_amdgcn_readlane((int)ba, 8 * ci), hi1 = __builtin_amdgcn_readlane((int)(ba >> 32), 8 * ci);
            const int lo2 = __builtin_amdgcn_readlane((int)bb, 8 * ci), hi2 = __builtin_amdgcn_readlane((int)(bb >> 32), 8 * ci);
            const double s1 = __builtin_bit_cast(double, ((long long)hi1 << 32) | (unsigned)lo1);
            const double s2 = __builtin_bit_cast(double, ((long long)hi2 << 32) | (unsigned)lo2);
            const double mu = s1 * a.bn_in.inv_count;
            double var = s2 * a.bn_in.inv_count - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            const float mean = (float)mu;
            const float invstd = 1.0f / sqrtf((float)(var + (double)a.bn_in.eps));
            kc[ci] = make_float4(mean, uniform_f(gam[ci]) * invstd, uniform_f(bet[ci]), invstd);
            if (blockIdx.x == 0 && tid == 0 && a.bn_in.update) {   // the designated consumer of this BatchNorm's forward pass
                a.bn_in.saved[2 * ci] = mean;
                a.bn_in.saved[2 * ci + 1] = invstd;
                a.bn_in.rmean[ci] = (1.f - a.bn_in.momentum) * rmn[ci] + a.bn_in.momentum * mean;
                a.bn_in.rvar[ci] = (1.f - a.bn_in.momentum) * rvr[ci] + a.bn_in.momentum * (float)(var * a.bn_in.unbias);
            }
        }
    }

    LF_STAMP(2);
    float dw[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) dw[i] = 0.f;
    float d1[CIN], d2[CIN];
#pragma unroll
    for (int i = 0; i < CIN; i++) d1[i] = d2[i] = 0.f;
    float ls[COUT], gs[COUT];
#pragma unroll
    for (int i = 0; i < COUT; i++) ls[i] = gs[i] = 0.f;

    // activation of input row `r` (register row index): {column n0-1, n0, n0+1}, zero outside the map
    auto activate = [&](int r, int ci, float (&act)[3]) {
        const int y = m0 - 1 + r;
        const bool rowok = y >= 0 && y < a.H;
        float v0 = rin[r][ci][0], v1 = rin[r][ci][1], vl = VEC4 ? 0.f : rleft[VEC4 ? 0 : r][ci];
        if constexpr (BN) {
            v0 = fmaxf(0.f, fmaf(v0 - kc[ci].x, kc[ci].y, kc[ci].z));
            v1 = fmaxf(0.f, fmaf(v1 - kc[ci].x, kc[ci].y, kc[ci].z));
            vl = fmaxf(0.f, fmaf(vl - kc[ci].x, kc[ci].y, kc[ci].z));
        }
        act[1] = (rowok && n0 < a.W) ? v0 : 0.f;
        act[2] = (rowok && n0 + 1 < a.W) ? v1 : 0.f;
        const float edge = (!VEC4 && rowok && n0 >= 1) ? vl : 0.f;       // only lane 0 keeps it: column n0-1 of a later strip (or nothing)
        act[0] = from_left(act[2], edge);
    };

    float actP[CIN][3];          // the previous input row (j = 1 of the current quad row)
    float carry[CIN][2];         // j = 0 shares of the previous quad row: the rest of pixel row m-1's input gradient
#pragma unroll
    for (int ci = 0; ci < CIN; ci++) {
        activate(0, ci, actP[ci]);
        carry[ci][0] = carry[ci][1] = 0.f;
    }

    float xsave[CIN][2];         // j = 1 shares of this band's first row: they complete the last pixel row of the band above
#pragma unroll
    for (int ci = 0; ci < CIN; ci++) xsave[ci][0] = xsave[ci][1] = 0.f;
    {
        static_for<NR>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int m = m0 + r;                       // quad row; its inputs are rows m (register row r+1) and m-1 (r)
            if (work && r < nrows && m < a.QH) {        // uniform per wave (the barrier below is outside)
            const bool own_row = r < HB || last_band;   // uniform: loss / bias / weight gradients counted here
            float actN[CIN][3];
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) activate(r + 1, ci, actN[ci]);

            // ---- forward of the lane's two quads, sigmoid, loss, gradient
            float g[COUT][2][2][2];                     // [co][quad][py][px]
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                float acc[2][2][2];
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int py = 0; py < 2; py++)
#pragma unroll
                        for (int px = 0; px < 2; px++) acc[q][py][px] = bk[co];
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) {
                    const float* wc = &wk[(ci * COUT + co) * KH * KW];
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                const float av = j ? actP[ci][1 + q - i] : actN[ci][1 + q - i];
#pragma unroll
                                for (int py = 0; py < 2; py++)
#pragma unroll
                                    for (int px = 0; px < 2; px++)
                                        if (py + 2 * j < KH && px + 2 * i < KW)
                                            acc[q][py][px] = fmaf(av, wc[(py + 2 * j) * KW + px + 2 * i], acc[q][py][px]);
                            }
                }
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    const float tv[4] = {tgt[r][co][py].x, tgt[r][co][py].y, tgt[r][co][py].z, tgt[r][co][py].w};
                    const bool rowok = 2 * m + py < a.OH;
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int px = 0; px < 2; px++) {
                            const int ox = 2 * (n0 + q) + px;
                            const bool valid = rowok && (all_cols || ox < a.OW);
                            const float yh = sigmoid_fast(acc[q][py][px]);
                            // an output outside the map has no error: d = 0 makes its loss term and its gradient vanish
                            const float d = valid ? yh - tv[2 * q + px] : 0.f;
                            const float gv = (2.0f * d * a.inv_count) * (yh * (1.0f - yh));
                            g[co][q][py][px] = gv;
                            // a quad column is counted by the strip that owns it: the last lane's second quad is the next strip's
                            // (selects, not branches: a wave runs alone on its SIMD and every exec-mask branch is a bubble);
                            // with one strip (VEC4) ownership is the row's, a uniform branch
                            if constexpr (VEC4) {
                                if (own_row) { ls[co] = fmaf(d, d, ls[co]); gs[co] += gv; }
                            } else {
                                const bool own = own_row && (q == 0 || lane < 63 || last_strip);
                                const float dl = own ? d : 0.f;
                                ls[co] = fmaf(dl, dl, ls[co]);
                                gs[co] += own ? gv : 0.f;
                            }
                        }
                }
            }
            // ---- weight gradient (owned quads only) and input-gradient shares
            float P[CIN][2][2][2];                      // [ci][quad][j][i]
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int i = 0; i < 2; i++) P[ci][q][j][i] = 0.f;
#pragma unroll
                for (int co = 0; co < COUT; co++) {
                    const float* wc = &wk[(ci * COUT + co) * KH * KW];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const bool ownq = VEC4 || q == 0 || lane < 63 || last_strip;
#pragma unroll
                        for (int j = 0; j < 2; j++)
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                const float av = j ? actP[ci][1 + q - i] : actN[ci][1 + q - i];
                                const float avw = ownq ? av : 0.f;
#pragma unroll
                                for (int py = 0; py < 2; py++)
#pragma unroll
                                    for (int px = 0; px < 2; px++)
                                        if (py + 2 * j < KH && px + 2 * i < KW) {
                                            const int t = (py + 2 * j) * KW + px + 2 * i;
                                            P[ci][q][j][i] = fmaf(g[co][q][py][px], wc[t], P[ci][q][j][i]);
                                            if (own_row) dw[(ci * COUT + co) * KH * KW + t] = fmaf(avw, g[co][q][py][px], dw[(ci * COUT + co) * KH * KW + t]);
                                        }
                            }
                    }
                }
            }
            // ---- pixel row y = m-1: complete, mask, store, BatchNorm-backward sums; then the new carry
            const int y = m - 1;
            const bool emit = r >= 1 && y < a.H;        // uniform; r == 0 completes a row of the band above (it does so itself)
            float tj1[CIN][2], tj0[CIN][2];
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                // pixel n0: quad 0's i = 0 shares + quad 1's i = 1 shares; pixel n0+1: quad 1's i = 0 + the right neighbour's quad 0, i = 1
                tj1[ci][0] = P[ci][0][1][0] + P[ci][1][1][1];
                tj0[ci][0] = P[ci][0][0][0] + P[ci][1][0][1];
                tj1[ci][1] = P[ci][1][1][0] + from_right(P[ci][0][1][1], 0.f);
                tj0[ci][1] = P[ci][1][0][0] + from_right(P[ci][0][0][1], 0.f);
                if (r == 0) { xsave[ci][0] = tj1[ci][0]; xsave[ci][1] = tj1[ci][1]; }
            }
            if (emit) {
                const bool p0 = n0 < a.W, p1 = n0 + 1 < a.W && lane < 63;
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) {
                    float gv0 = carry[ci][0] + tj1[ci][0], gv1 = carry[ci][1] + tj1[ci][1];
                    if constexpr (BN) {
                        const float e0 = rin[r][ci][0] - kc[ci].x, e1 = rin[r][ci][1] - kc[ci].x;
                        gv0 = fmaf(e0, kc[ci].y, kc[ci].z) > 0.f ? gv0 : 0.f;
                        gv1 = fmaf(e1, kc[ci].y, kc[ci].z) > 0.f ? gv1 : 0.f;
                        const float s0 = p0 ? gv0 : 0.f, s1 = p1 ? gv1 : 0.f;
                        d1[ci] += s0 + s1;
                        d2[ci] = fmaf(s0, e0 * kc[ci].w, fmaf(s1, e1 * kc[ci].w, d2[ci]));
                    }
                    // one exec-mask branch per channel: the second store of a lane without a second pixel repeats its first
                    float* gp = a.gin + (size_t)(b * CIN + ci) * HW + (unsigned)y * a.W + n0;
                    if (p0) {
                        gp[0] = gv0;
                        gp[p1 ? 1 : 0] = p1 ? gv1 : gv0;
                    }
                }
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                carry[ci][0] = tj0[ci][0];
                carry[ci][1] = tj0[ci][1];
#pragma unroll
                for (int k = 0; k < 3; k++) actP[ci][k] = actN[ci][k];
            }
            }
            if constexpr (r == 0) {
                LF_STAMP(3);
                // every wave, working or not: the first row's j = 1 shares go up one band through LDS
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) {
                    xch[((wv * CIN + ci) * 2 + 0) * 64 + lane] = xsave[ci][0];
                    xch[((wv * CIN + ci) * 2 + 1) * 64 + lane] = xsave[ci][1];
                }
                __syncthreads();
            }
        });
    }
    if (has_below) {
        // HB rows computed: the last pixel row (m0 + HB - 1) is completed by the first-row shares of the band below
        const int y = m0 + HB - 1;
        if (y < a.H) {
            const bool p0 = n0 < a.W, p1 = n0 + 1 < a.W && lane < 63;
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                float gv0 = carry[ci][0] + xch[(((wv + 1) * CIN + ci) * 2 + 0) * 64 + lane];
                float gv1 = carry[ci][1] + xch[(((wv + 1) * CIN + ci) * 2 + 1) * 64 + lane];
                if constexpr (BN) {
                    const float e0 = rin[HB][ci][0] - kc[ci].x, e1 = rin[HB][ci][1] - kc[ci].x;
                    gv0 = fmaf(e0, kc[ci].y, kc[ci].z) > 0.f ? gv0 : 0.f;
                    gv1 = fmaf(e1, kc[ci].y, kc[ci].z) > 0.f ? gv1 : 0.f;
                    const float s0 = p0 ? gv0 : 0.f, s1 = p1 ? gv1 : 0.f;
                    d1[ci] += s0 + s1;
                    d2[ci] = fmaf(s0, e0 * kc[ci].w, fmaf(s1, e1 * kc[ci].w, d2[ci]));
                }
                float* gp = a.gin + (size_t)(b * CIN + ci) * HW + (unsigned)y * a.W + n0;
                if (p0) {
                    gp[0] = gv0;
                    gp[p1 ? 1 : 0] = p1 ? gv1 : gv0;
                }
            }
        }
    }

    LF_STAMP(4);
    // ---- reductions: wave (through LDS), workgroup (LDS), then one fp64 atomic per value
    {
        float red[NRED];
#pragma unroll
        for (int i = 0; i < NACC; i++) red[i] = dw[i];
#pragma unroll
        for (int c = 0; c < CIN; c++) {
            red[NACC + 2 * c] = d1[c];
            red[NACC + 2 * c + 1] = d2[c];
        }
        wave_sums_lds<NRED>(red, wscr + wv * kWsumScratch, redf + wv * NRED, lane);
    }
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        // a lane's partial loss covers at most 8 (HB+1) terms in [0,1]: fp32; across lanes and waves fp64
        const double t1 = wave_sum_lane63((double)ls[co] * (double)a.inv_count), t2 = wave_sum_lane63((double)gs[co]);
        if (lane == 63) {
            redd[(wv * COUT + co) * 2] = t1;
            redd[(wv * COUT + co) * 2 + 1] = t2;
        }
    }
    __syncthreads();
    const int shard = blockIdx.x & (kStatShards - 1);
    for (int j = tid; j < NRED + 2 * COUT; j += 256) {
        if (j < NRED) {
            const double s = (double)redf[j] + (double)redf[NRED + j] + (double)redf[2 * NRED + j] + (double)redf[3 * NRED + j];
            if (j < NACC) {
                acc_add<ACC_GRAD>(&a.wacc[(size_t)shard * a.acc_stride + j], s);
            } else if (a.stats_in) {
                const int jj = j - NACC;
                acc_add<ACC_GRAD>(&a.stats_in[((size_t)shard * CIN + (jj >> 1)) * 4 + 2 + (jj & 1)], s);
            }
        } else {
            const int q = j - NRED, co = q >> 1, which = q & 1;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < 4; w++) s += redd[(w * COUT + co) * 2 + which];
            if (which == 0) acc_add<ACC_GRAD>(&a.losses[(size_t)loss_slot_v * kStatShards + shard], s);
            else acc_add<ACC_GRAD>(&a.bias_acc[(size_t)shard * a.acc_stride + co], s);
        }
    }
    LF_STAMP(5);
    // keeps the permutation value's register its own to the end: were it reused, every write to it would wait for the
    // load that may be pending on it (and, the counter being in order, for the BatchNorm sums and weights ahead of it)
    asm volatile("" :: "v"(perm_val_v));
#undef LF_STAMP
}

}  // namespace cae
