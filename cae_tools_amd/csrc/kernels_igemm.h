// kernels_igemm.h — implicit-GEMM kernels on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) for
// the channel-rich ConvTranspose2d layers at the head of the decoder (64->32, 32->16, 16->8 ...):
// there the contraction is long enough (K = 64..576) to be GEMM-shaped, while the maps are tiny
// (3x3 .. 31x31), so the operands are gathered straight from NCHW global memory (L2-resident)
// into the MFMA lane layout — im2col exists only as address arithmetic.
//
//   forward, stride 2 (k_ig_fwd_s2): one GEMM per output parity class (py,px):
//       out[(b,m,n)][co] = sum_{ci,(j,i)} a[b][ci][m-j][n-i] * W[ci][co][py+2j][px+2i]
//       M = B*QH*QW quads, N = Cout, K = 4*Cin.  A k-step of the 16x16x4 MFMA is exactly one input
//       channel's 2x2 neighbourhood, so a lane's tap (j,i) and its validity are loop invariants.
//   input gradient (k_ig_dgrad):  ga[(b,y,x)][ci] = sum_{(co,ky,kx)} gy[b][co][s*y+ky][s*x+kx] * W[ci][(co,ky,kx)]
//       M = B*H*W, N = Cin, K = Cout*kh*kw
//   weight gradient (k_ig_wgrad): dW[ci][(co,ky,kx)] = sum_{(b,y,x)} a[b][ci][y][x] * gy[b][co][s*y+ky][s*x+kx]
//       M = Cin, N = Cout*kh*kw, K = B*H*W, split over workgroups (fp64 atomics) and over the 4 waves.
// BatchNorm+ReLU of the producer / BatchNorm-backward of the gradient are applied to the operands
// as they are loaded; bias, BatchNorm sums and ReLU masking ride in the epilogues.
#pragma once
#include "kernels_gemm.h"

namespace cae {


// 4-byte aligned 8-, 12- and 16-byte loads (global memory takes them unaligned)
struct __attribute__((packed, aligned(4))) F2 {
    float x, y;
};
struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
};
struct __attribute__((packed, aligned(4))) F4 {
    float x, y, z, w;
};

// ---------------------------------------------------------------------------------------------
struct IgFwd {
    int B, Cin, H, W, Cout, OH, OW, KH, KW, QH, QW;
    int tiles_per_wave;
    int ksplit;            // 1, 2 or 4: waves of a workgroup that share one M-tile and split the channels
    const float* in;
    BnDesc bn_in;
    const float* w;
    const float* bias;
    float* out;
    double* stats;  // [shards][Cout][4] or nullptr (eval)
    long long* dbg; // diagnostics (tools/head_phases.py ig): 8 wall-clock stamps per workgroup of parity 0, or nullptr
};

// grid (ceil(Mtiles / (4*tiles_per_wave)), 4 parities, ceil(Cout/16)), block 256
__global__ void __launch_bounds__(256) k_ig_fwd_s2(IgFwd a) {
#define IG_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) a.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    extern __shared__ double lds_d[];
    IG_STAMP(0);
    double* lstat = lds_d;                           // [16 channels][2], fp64: see the statistics epilogue
    float* part = reinterpret_cast<float*>(lds_d + 32);   // [4 waves][256] split-K partial tiles
    float4* cin4 = reinterpret_cast<float4*>(part + 1024);
    bn_consts(a.bn_in, cin4, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
    if (threadIdx.x < 32) lstat[threadIdx.x] = 0.0;
    __syncthreads();
    IG_STAMP(1);

    const int py = blockIdx.y >> 1, px = blockIdx.y & 1;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int j = q >> 1, i = q & 1;                 // this lane's tap inside the 2x2 neighbourhood
    const int QQ = a.QH * a.QW;
    const int M = a.B * QQ;
    const int HW = a.H * a.W;
    const bool fastdiv = M + 16 < kDivSmallMaxN && QQ < kDivSmallMaxD;
    const float inv_qq = 1.0f / (float)QQ, inv_qw = 1.0f / (float)a.QW;
    const int co = blockIdx.z * 16 + r;              // B column / C column of this lane
    const int ky = py + 2 * j, kx = px + 2 * i;
    const bool b_ok = co < a.Cout && ky < a.KH && kx < a.KW;
    const size_t b_step = (size_t)a.Cout * a.KH * a.KW;
    const float bias = co < a.Cout ? a.bias[co] : 0.f;
    float s1 = 0.f, s2 = 0.f;

    const int KS = a.ksplit;
    const int mslot = wv / KS, kslot = wv - mslot * KS;       // which M-tile of the workgroup, which K slice
    const int cper = ((a.Cin + KS - 1) / KS + 7) & ~7;        // channels per K slice (multiple of the unroll)
    const int cbeg = kslot * cper, cend = min(a.Cin, cbeg + cper);
    for (int tt = 0; tt < a.tiles_per_wave; tt++) {
        const int tile = (blockIdx.x * (4 / KS) + mslot) * a.tiles_per_wave + tt;
        const bool tile_ok = tile * 16 < M;                   // uniform per wave; no early exit: barriers below
        // A row of this lane: quad m -> (b, qm, qn)
        const int m = tile * 16 + r;
        bool a_ok = tile_ok && m < M;
        int b, qm;   // 3-instruction fp32 products instead of ~40-instruction integer divisions (ten of them per tile)
        if (fastdiv) {
            b = div_small(m, inv_qq);
            qm = div_small(m - b * QQ, inv_qw);
        } else {
            b = m / QQ;
            qm = (m - b * QQ) / a.QW;
        }
        const int rem = m - b * QQ, qn = rem - qm * a.QW;
        const int iy = qm - j, ix = qn - i;
        a_ok = a_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        // Lean operand fetch: this phase is bound by instruction issue (16 waves per CU run ~700 instructions each around
        // 16 MFMAs: tools/head_phases.py ig), so a load is `wave-uniform channel base + loop-invariant 32-bit lane offset`
        // (one instruction, no 64-bit lane arithmetic), issued unconditionally from an in-range address, and invalid lanes /
        // channels are zeroed by one select after the BatchNorm transform.
        const unsigned a_off = a_ok ? (unsigned)((b * a.Cin) * HW + iy * a.W + ix) : 0u;
        const unsigned b_off = b_ok ? (unsigned)((co * a.KH + ky) * a.KW + kx) : 0u;

        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // Paired-tap variant (the vector-memory instruction count bounds this phase, see ig_dgrad_body): lane group q takes
        // row tap jp = q & 1 and BOTH column taps of every second channel (parity q >> 1): one 8-byte load of two adjacent
        // inputs and one 12-byte load of the weight row per channel instead of two dword loads each; slot u = 2t + i of the
        // MFMA holds k = (channel c0 + 2t + (q >> 1), row tap jp, column tap i) on both operands.
        const bool paired = a.W >= 2;
        const int jp = q & 1, hp = q >> 1;
        const int iyp = qm - jp;
        const bool rowp_ok = tile_ok && m < M && iyp >= 0 && iyp < a.H;
        const int cst = min(max(qn - 1, 0), a.W - 2);                 // first column of the loaded pair
        const int d0 = qn - cst, d1 = qn - 1 - cst;                   // where column taps i = 0 / 1 sit in the pair (0, 1: valid)
        const bool v0 = rowp_ok && qn < a.W && (unsigned)d0 < 2u, v1 = rowp_ok && qn >= 1 && (unsigned)d1 < 2u;
        const unsigned ap_off = rowp_ok ? (unsigned)((b * a.Cin) * HW + iyp * a.W + cst) : 0u;
        const int kyp = py + 2 * jp;
        const bool wrow_ok = co < a.Cout && kyp < a.KH;
        const bool w0_ok = wrow_ok && px < a.KW, w1_ok = wrow_ok && px + 2 < a.KW;
        const unsigned bp_off = wrow_ok ? (unsigned)((co * a.KH + kyp) * a.KW + px) : 0u;
        for (int c0 = cbeg; paired && c0 < cend; c0 += 8) {
            float av[8], bv[8];
            F2 x2[4];
            F3 w3[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int cc = min(c0 + 2 * t + hp, a.Cin - 1);
                x2[t] = *reinterpret_cast<const F2*>(a.in + (size_t)cc * HW + ap_off);
                w3[t] = *reinterpret_cast<const F3*>(a.w + (size_t)cc * b_step + bp_off);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int ci = c0 + 2 * t + hp;
                const bool c_ok = ci < cend;
                float e0 = d0 == 0 ? x2[t].x : x2[t].y, e1 = d1 == 0 ? x2[t].x : x2[t].y;
                if (a.bn_in.mode != BN_NONE) {
                    const float4 k = cin4[min(ci, a.Cin - 1)];
                    e0 = fmaxf(0.f, fmaf(e0 - k.x, k.y, k.z));
                    e1 = fmaxf(0.f, fmaf(e1 - k.x, k.y, k.z));
                }
                av[2 * t] = (v0 && c_ok) ? e0 : 0.f;
                av[2 * t + 1] = (v1 && c_ok) ? e1 : 0.f;
                bv[2 * t] = (w0_ok && c_ok) ? w3[t].x : 0.f;
                bv[2 * t + 1] = (w1_ok && c_ok) ? w3[t].z : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        for (int c0 = cbeg; !paired && c0 < cend; c0 += 8) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int cc = min(c0 + u, a.Cin - 1);   // wave-uniform
                av[u] = (a.in + (size_t)cc * HW)[a_off];
                bv[u] = (a.w + (size_t)cc * b_step)[b_off];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ci = c0 + u;
                const bool c_ok = ci < cend;
                if (a.bn_in.mode != BN_NONE) {
                    const float4 k = cin4[min(ci, a.Cin - 1)];
                    av[u] = fmaxf(0.f, fmaf(av[u] - k.x, k.y, k.z));
                }
                av[u] = (a_ok && c_ok) ? av[u] : 0.f;
                bv[u] = (b_ok && c_ok) ? bv[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        IG_STAMP(2);
        if (KS > 1) {   // combine the K slices of this M-tile in the slice-0 wave
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < 4; jj++) part[wv * 256 + jj * 64 + lane] = acc[jj];
            __syncthreads();
            if (kslot == 0) {
                for (int o = 1; o < KS; o++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) acc[jj] += part[(wv + o) * 256 + jj * 64 + lane];
            }
        }
        // epilogue: rows 4q..4q+3 of the tile, column co
        if (co < a.Cout && kslot == 0 && tile_ok) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int cm = tile * 16 + q * 4 + jj;
                if (cm >= M) continue;
                int cb, cqm;
                if (fastdiv) {
                    cb = div_small(cm, inv_qq);
                    cqm = div_small(cm - cb * QQ, inv_qw);
                } else {
                    cb = cm / QQ;
                    cqm = (cm - cb * QQ) / a.QW;
                }
                const int crem = cm - cb * QQ, cqn = crem - cqm * a.QW;
                const int oy = 2 * cqm + py, ox = 2 * cqn + px;
                if (oy >= a.OH || ox >= a.OW) continue;
                const float v = acc[jj] + bias;
                a.out[((size_t)(cb * a.Cout + co) * a.OH + oy) * a.OW + ox] = v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
    }
    IG_STAMP(3);
    if (a.stats) {
        // A lane's fp32 sums cover 4 * tiles_per_wave values; from there on fp64 (as k_ct_fwd_lds does).  Until round 3 the
        // fold over lane groups and waves ran in fp32 (64 * tiles_per_wave values per channel and workgroup): var = E[y^2] -
        // mean^2 amplifies that rounding by mean^2 / var, and at the benchmark batch this kernel's gradients sat 6.8e-4 from
        // the oracle's where the LDS-staged forward's sit within 2e-4 (tests/test_full_size_gpu.py).
        // lanes r, r+16, r+32, r+48 hold the same channel: fold, then LDS, then one fp64 atomic per value
        double d1 = (double)s1, d2 = (double)s2;
        d1 += __shfl_xor(d1, 16, 64); d2 += __shfl_xor(d2, 16, 64);
        d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
        if (q == 0 && co < a.Cout) {
            atomicAdd(&lstat[2 * r], acc_grid<ACC_STAT>(d1));       // (on the grid: exact, hence order-independent, adds)
            atomicAdd(&lstat[2 * r + 1], acc_grid<ACC_STAT>(d2));
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int c = blockIdx.z * 16 + (threadIdx.x >> 1);
            if (c < a.Cout) {
                const int shard = (blockIdx.x + blockIdx.y) & (kStatShards - 1);
                acc_add<ACC_STAT>(&a.stats[((size_t)shard * a.Cout + c) * 4 + (threadIdx.x & 1)], lstat[threadIdx.x]);
            }
        }
        IG_STAMP(4);
}
}

// ---------------------------------------------------------------------------------------------
struct IgDgrad {
    int B, Cin, H, W, Cout, OH, OW, KH, KW, S;
    int tiles_per_wave;
    int ksplit;            // 1, 2 or 4 waves share one M-tile and split K = Cout*kh*kw
    const float* g;        // (B,Cout,OH,OW) masked gradient (or dL/dy of the last layer)
    const float* yout;     // raw forward output of this layer (BN_BWD) or nullptr
    BnDesc bn_out;
    const float* w;        // (Cin, Cout*KH*KW)
    float* gin;            // (B,Cin,H,W)
    const float* yprev;    // raw output of the producer (mask) or nullptr: plain store
    BnDesc bn_prev;        // BN_SAVED of the producer or BN_NONE
    double* stats_prev;    // [shards][Cin][4] slots 2,3
    long long* dbg;        // diagnostics: 4 stamps per workgroup at dbg[(512 + bx) * 4 ..], or nullptr
};

// grid (ceil(Mtiles / (4*tiles_per_wave)), ceil(Cin/16)), block 256; LDS: see host
constexpr int kIgdSteps = 9;

__device__ __forceinline__ void ig_dgrad_body(const IgDgrad& a, const int bx, const int by, double* lds_d) {
#define IGD_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && by == 0 && bx < 256) a.dbg[(512 + bx) * 4 + (i)] = wall_clock64(); } while (0)
    IGD_STAMP(0);
    const int K = a.Cout * a.KH * a.KW;
    float* lstat = reinterpret_cast<float*>(lds_d);               // [4 waves][16][2]: one writer per slot, folded in wave order
    float* part = lstat + 128;                                    // [4][256] split-K partial tiles
    float4* cout4 = reinterpret_cast<float4*>(part + 1024);       // [Cout]
    float4* cprev4 = cout4 + a.Cout;                              // [Cin]
    int2* koff = reinterpret_cast<int2*>(cprev4 + a.Cin);         // [K] {offset of tap k inside one image of gy, its channel}
    bn_consts(a.bn_out, cout4, false);
    bn_consts(a.bn_prev, cprev4, false, 64);
    for (int k = threadIdx.x; k < K && !(a.KH == 3 && a.KW == 3); k += 256) {   // the 3x3 path below needs no tap table
        const int co = k / (a.KH * a.KW), t = k - co * (a.KH * a.KW);
        const int ky = t / a.KW, kx = t - ky * a.KW;
        koff[k] = make_int2((co * a.OH + ky) * a.OW + kx, co);
    }
    if (threadIdx.x < 128) lstat[threadIdx.x] = 0.f;
    __syncthreads();
    IGD_STAMP(1);

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int HW = a.H * a.W;
    const int M = a.B * HW;
    const int ci = by * 16 + r;
    const bool b_ok = ci < a.Cin;
    const unsigned w_lane = (unsigned)(min(ci, a.Cin - 1) * K);
    const bool k33 = a.KH == 3 && a.KW == 3 && kIgdSteps == 9;

    const int KS = a.ksplit;
    const int mslot = wv / KS, kslot = wv - mslot * KS;
    // k per slice, a multiple of one batch of kIgdSteps MFMA k-steps: the benchmark layers' slices (Cout * 9 / ksplit = 36, 36
    // and 72 k) are whole batches of 9, so no batch carries padding steps (12-deep batches were a quarter padding there)
    const int kper = (((K + 3) / 4 + KS - 1) / KS + kIgdSteps - 1) / kIgdSteps * (4 * kIgdSteps);
    const int kbeg = kslot * kper, kend = min(K, kbeg + kper);
    for (int tt = 0; tt < a.tiles_per_wave; tt++) {
        const int tile = (bx * (4 / KS) + mslot) * a.tiles_per_wave + tt;
        const bool tile_ok = tile * 16 < M;
        const int m = tile * 16 + r;
        const bool a_ok = tile_ok && m < M;
        const int mc = a_ok ? m : 0;
        const int b = mc / HW, rem = mc - b * HW;
        const int y = rem / a.W, x = rem - y * a.W;
        // lean operand fetch (see k_ig_fwd_s2): 32-bit offsets, always in range (clamped row and k), one select afterwards
        const unsigned abase = (unsigned)(b * a.Cout * a.OH * a.OW + (a.S * y) * a.OW + a.S * x);

        // epilogue inputs of this lane (position m, channels by * 16 + 4q + jj), requested before the k loop so that their
        // latency hides behind it
        unsigned eoff[4];
        float yp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            eoff[jj] = (unsigned)((b * a.Cin + min(by * 16 + q * 4 + jj, a.Cin - 1)) * HW + rem);
            if (a.bn_prev.mode != BN_NONE && kslot == 0) yp[jj] = a.yprev[eoff[jj]];
        }

        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = kbeg; k0 < kend; k0 += 4 * kIgdSteps) {
            float av[kIgdSteps], bv[kIgdSteps], yv[kIgdSteps];
            if (k33) {
                // 3x3 kernels: a batch is 36 k = four output channels' 3x3 taps.  The vector-memory pipe, not bytes or FLOPs,
                // bounds this phase (each 64-lane load instruction costs it ~16 cycles whatever it hits; with every load
                // removed the phase is 2.1 us instead of 6.6): so lane group q takes ALL nine taps of channel k0/9 + q —
                // three 12-byte row loads each of g and y and three loads of its nine contiguous weights instead of 27
                // dword loads — and slot (u, q) of the MFMA holds k = k0 + 9q + u on both operands.
                const int co = k0 / 9 + q;
                const bool c_ok = co * 9 < kend;
                const int coc = min(co, a.Cout - 1);
                const unsigned base = abase + (unsigned)(coc * a.OH * a.OW);
                const float* wp = a.w + w_lane + coc * 9;
                F3 g3[3], y3[3];
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    g3[ky] = *reinterpret_cast<const F3*>(a.g + base + ky * a.OW);
                    if (a.bn_out.mode == BN_BWD) y3[ky] = *reinterpret_cast<const F3*>(a.yout + base + ky * a.OW);
                }
                const F4 w0 = *reinterpret_cast<const F4*>(wp), w1 = *reinterpret_cast<const F4*>(wp + 4);
                const float w8 = wp[8];
                const float wv9[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w8};
                const float4 c4 = a.bn_out.mode == BN_BWD ? cout4[coc] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 9; u++) {
                    const F3 gr = g3[u / 3], yr = y3[u / 3];
                    float v = u % 3 == 0 ? gr.x : (u % 3 == 1 ? gr.y : gr.z);
                    if (a.bn_out.mode == BN_BWD) {
                        const float yy = u % 3 == 0 ? yr.x : (u % 3 == 1 ? yr.y : yr.z);
                        v = c4.y * v - c4.z - (yy - c4.x) * c4.w;
                    }
                    av[u] = (a_ok && c_ok) ? v : 0.f;
                    bv[u] = (b_ok && c_ok) ? wv9[u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 9; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
                continue;
            }
            int ch[kIgdSteps];
#pragma unroll
            for (int u = 0; u < kIgdSteps; u++) {
                const int kc = min(k0 + 4 * u + q, kend - 1);
                const int2 kk = koff[kc];
                const unsigned off = abase + (unsigned)kk.x;
                ch[u] = kk.y;
                av[u] = a.g[off];
                yv[u] = a.bn_out.mode == BN_BWD ? a.yout[off] : 0.f;
                bv[u] = a.w[w_lane + (unsigned)kc];
            }
#pragma unroll
            for (int u = 0; u < kIgdSteps; u++) {
                const bool k_ok = k0 + 4 * u + q < kend;
                float v = av[u];
                if (a.bn_out.mode == BN_BWD) {
                    const float4 c4 = cout4[ch[u]];
                    v = c4.y * v - c4.z - (yv[u] - c4.x) * c4.w;
                }
                av[u] = (a_ok && k_ok) ? v : 0.f;
                bv[u] = (b_ok && k_ok) ? bv[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kIgdSteps; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        IGD_STAMP(2);
        // the epilogue inputs are waited for here, once, outside the predicated stores below (left to their first use inside
        // a predicated block, the wait is repeated per block and, the memory counter retiring in order, becomes a wait for
        // the previous block's store to complete: four store round trips one after the other)
        asm volatile("" : "+v"(yp[0]), "+v"(yp[1]), "+v"(yp[2]), "+v"(yp[3]));
        if (KS > 1) {
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < 4; jj++) part[wv * 256 + jj * 64 + lane] = acc[jj];
            __syncthreads();
            if (kslot == 0) {
                for (int o = 1; o < KS; o++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) acc[jj] += part[(wv + o) * 256 + jj * 64 + lane];
            }
        }
        if (kslot == 0 && tile_ok) {
            // The MFMA leaves C as (row = position, column = channel) with the channel along the lanes, but the tensors are
            // channel-major: stored from that layout every store / yprev load touches 16 lines.  Transposed through the wave's
            // own LDS tile, a lane owns one position and four channels: 16 consecutive positions per instruction and channel.
            float* tl = part + wv * 256;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) tl[(q * 4 + jj) * 16 + r] = acc[jj];
            const bool m_ok = a_ok;                     // this lane's position is m; channels by * 16 + q * 4 + jj
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int cc = by * 16 + q * 4 + jj;
                const bool ok = m_ok && cc < a.Cin;
                const unsigned off = eoff[jj];
                float v = tl[r * 16 + q * 4 + jj];
                float e1 = 0.f, e2 = 0.f;
                if (a.bn_prev.mode != BN_NONE) {
                    const float4 c4 = cprev4[min(cc, a.Cin - 1)];
                    const float d = yp[jj] - c4.x;
                    v = fmaf(d, c4.y, c4.z) > 0.f ? v : 0.f;
                    e1 = ok ? v : 0.f;
                    e2 = e1 * (d * c4.w);
                }
                if (ok) a.gin[off] = v;
                if (a.stats_prev) {
                    // channel cc's sums over the 16 positions of this row of lanes (DPP row reduction: lane 15 of the row)
                    e1 += dpp_f<0xB1>(e1); e2 += dpp_f<0xB1>(e2);
                    e1 += dpp_f<0x4E>(e1); e2 += dpp_f<0x4E>(e2);
                    e1 += dpp_f<0x141>(e1); e2 += dpp_f<0x141>(e2);
                    e1 += dpp_f<0x140>(e1); e2 += dpp_f<0x140>(e2);
                    if (r == 15 && cc < a.Cin) {   // the only lane of this wave that owns slot (q, jj): no atomics, a fixed order
                        lstat[wv * 32 + 2 * (q * 4 + jj)] += e1;
                        lstat[wv * 32 + 2 * (q * 4 + jj) + 1] += e2;
                    }
                }
            }
        }
    }
    if (a.stats_prev) {
        __syncthreads();
        if (threadIdx.x < 32) {
            const int c = by * 16 + (threadIdx.x >> 1);
            if (c < a.Cin) {
                const int shard = bx & (kStatShards - 1);
                const double t = (double)lstat[threadIdx.x] + (double)lstat[32 + threadIdx.x] + (double)lstat[64 + threadIdx.x] +
                                 (double)lstat[96 + threadIdx.x];
                acc_add<ACC_GRAD>(&a.stats_prev[((size_t)shard * a.Cin + c) * 4 + 2 + (threadIdx.x & 1)], t);
            }
        }
    }
    IGD_STAMP(3);
}

// ---------------------------------------------------------------------------------------------
struct IgWgrad {
    int B, Cin, H, W, Cout, OH, OW, KH, KW, S;
    int ksteps_per_block;  // MFMA k-steps (of 4 positions) per workgroup
    const float* ain;      // (B,Cin,H,W) raw output of the producer (or plain activations)
    BnDesc bn_in;          // BN_SAVED or BN_NONE
    const float* g;
    const float* yout;
    BnDesc bn_out;         // BN_BWD or BN_NONE
    double* wacc;          // (Cin, Cout*KH*KW) fp64 accumulator
    BnGradOut bg;
    long long* dbg;        // diagnostics: 4 stamps per workgroup at dbg[(by * tiles + bx) * 4 ..] for the first 512, or nullptr
};

// grid (Mtiles*Ntiles, K chunks), block 256: the 4 waves split the chunk, LDS combine, fp64 atomics
__device__ __forceinline__ void ig_wgrad_body(const IgWgrad& a, const int bx, const int by, double* lds_d) {
#define IGW_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && wslot < 512) a.dbg[wslot * 4 + (i)] = wall_clock64(); } while (0)
    const int wslot = by * (((a.Cin + 15) >> 4) * ((a.Cout * a.KH * a.KW + 15) >> 4)) + bx;
    IGW_STAMP(0);
    float* part = reinterpret_cast<float*>(lds_d);            // [4][256]
    float4* cin4 = reinterpret_cast<float4*>(part + 1024);    // [Cin]
    float4* cout4 = cin4 + a.Cin;                             // [Cout]
    bn_consts(a.bn_in, cin4, false);
    bn_consts(a.bn_out, cout4, false, 64);
    if (bx == 0 && by == 0 && a.bg.stats) {
        for (int c = threadIdx.x; c < a.bg.C; c += 256) {
            double sb = 0.0, sg = 0.0;
            for (int sh = 0; sh < kStatShards; sh++) {
                sb += a.bg.stats[((size_t)sh * a.bg.C + c) * 4 + 2];
                sg += a.bg.stats[((size_t)sh * a.bg.C + c) * 4 + 3];
            }
            a.bg.beta_acc[c] = sb * a.bg.scale;
            a.bg.gamma_acc[c] = sg * a.bg.scale;
        }
    }
    __syncthreads();
    IGW_STAMP(1);

    const int khw = a.KH * a.KW;
    const int N = a.Cout * khw;
    const int HW = a.H * a.W;
    const int K = a.B * HW;
    const int tiles_n = (N + 15) >> 4;
    const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int ci = tm * 16 + r;                      // A row of this lane
    const int n = tn * 16 + r;                       // B column of this lane: (co, ky, kx)
    const bool a_ok = ci < a.Cin, b_ok = n < N;
    const int co = b_ok ? n / khw : 0;
    const int t = b_ok ? n - co * khw : 0;
    const int ky = t / a.KW, kx = t - ky * a.KW;
    const float4 ka = (a.bn_in.mode != BN_NONE && a_ok) ? cin4[ci] : make_float4(0, 0, 0, 0);
    const float4 kb = (a.bn_out.mode == BN_BWD && b_ok) ? cout4[co] : make_float4(0, 0, 0, 0);

    const int steps = (K + 3) >> 2;
    const int s_begin = by * a.ksteps_per_block;
    const int s_end = min(steps, s_begin + a.ksteps_per_block);
    const int per = (s_end - s_begin + 3) >> 2;
    const int s0 = s_begin + wv * per, s1 = min(s_end, s0 + per);

    // Lean operand fetch (see k_ig_fwd_s2): 32-bit offsets from the tensor bases (one load instruction each, no 64-bit lane
    // arithmetic), always from an in-range address (clamped position / row / column), zeroed by one select afterwards.
    const bool fastdiv = K < kDivSmallMaxN && HW < kDivSmallMaxD;
    const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.W;
    const unsigned a_lane = (unsigned)(min(ci, a.Cin - 1) * HW);                          // + (b * Cin) * HW + pos
    const unsigned b_lane = (unsigned)((co * a.OH + ky) * a.OW + kx);                     // + b * Cout*OH*OW + S*y*OW + S*x
    const unsigned img_a = (unsigned)(a.Cin * HW), img_b = (unsigned)(a.Cout * a.OH * a.OW);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int st = s0; st < s1; st += 8) {
        float av[8], bv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = min((st + u) * 4 + q, K - 1);
            int b, y;
            if (fastdiv) {
                b = div_small(k, inv_hw);
                y = div_small(k - b * HW, inv_w);
            } else {
                b = k / HW;
                y = (k - b * HW) / a.W;
            }
            const int pos = k - b * HW, x = pos - y * a.W;
            const unsigned oa = (unsigned)b * img_a + (unsigned)pos + a_lane;
            const unsigned ob = (unsigned)b * img_b + (unsigned)((a.S * y) * a.OW + a.S * x) + b_lane;
            av[u] = a.ain[oa];
            bv[u] = a.g[ob];
            yv[u] = a.bn_out.mode == BN_BWD ? a.yout[ob] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool k_ok = (st + u) < s1 && (st + u) * 4 + q < K;
            float va = av[u], vb = bv[u];
            if (a.bn_in.mode != BN_NONE) va = fmaxf(0.f, fmaf(va - ka.x, ka.y, ka.z));
            if (a.bn_out.mode == BN_BWD) vb = kb.y * vb - kb.z - (yv[u] - kb.x) * kb.w;
            av[u] = (a_ok && k_ok) ? va : 0.f;
            bv[u] = (b_ok && k_ok) ? vb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
    IGW_STAMP(2);
#pragma unroll
    for (int j = 0; j < 4; j++) part[wv * 256 + j * 64 + lane] = acc[j];
    __syncthreads();
    if (wv != 0) return;
    const int cn = tn * 16 + r;
    if (cn >= N) {
        IGW_STAMP(3);
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cm = tm * 16 + q * 4 + j;
        if (cm >= a.Cin) continue;
        const float v = part[j * 64 + lane] + part[256 + j * 64 + lane] + part[512 + j * 64 + lane] + part[768 + j * 64 + lane];
        acc_add<ACC_GRAD>(&a.wacc[(size_t)cm * N + cn], (double)v);
    }
    IGW_STAMP(3);
}

__global__ void __launch_bounds__(256) k_ig_dgrad(IgDgrad a) {
    extern __shared__ double lds_d[];
    ig_dgrad_body(a, blockIdx.x, blockIdx.y, lds_d);
}

__global__ void __launch_bounds__(256) k_ig_wgrad(IgWgrad a) {
    extern __shared__ double lds_d[];
    ig_wgrad_body(a, blockIdx.x, blockIdx.y, lds_d);
}

// weight gradient and input gradient of one layer in a single launch: workgroups [0, nw) are the
// (tile, K-chunk) grid of the weight gradient, the rest the (M-tile, Cin-tile) grid of the input gradient
//
// XCD-aware block order.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2, and both halves read the same
// gradient map: a weight-gradient K chunk (w.ksteps_per_block * 4 input positions, all its tiles) and the input-gradient
// blocks of the same positions (d_group of them, all Cin tiles) are given ids with the same id % 8, so one XCD's L2 fetches
// a slice of the gradient / activation maps once instead of every L2 fetching most of it.
// grid = ig_pair_blocks(...): w_n8 = ceil(chunks / 8), d_n8 = ceil(ceil(d_gx / d_group) / 8).
__global__ void __launch_bounds__(256) k_ig_bwd_pair(IgWgrad w, IgDgrad d, int w_tiles, int w_chunks, int w_n8, int d_gx,
                                                      int d_gy, int d_group) {
    kernarg_warm<sizeof(IgWgrad) + sizeof(IgDgrad) + 24>();
    extern __shared__ double lds_d[];
    const int nw = 8 * w_tiles * w_n8;
    const int id = blockIdx.x, x = id & 7;
    if (id < nw) {
        const int j = id >> 3, cx = j / w_tiles;
        const int chunk = cx * 8 + x;
        if (chunk < w_chunks) ig_wgrad_body(w, j - cx * w_tiles, chunk, lds_d);
    } else {
        const int j = (id - nw) >> 3;   // nw is a multiple of 8: (id - nw) & 7 == x
        const int by = j % d_gy, t = j / d_gy;
        const int sub = t % d_group, cx = t / d_group;
        const int bx = (cx * 8 + x) * d_group + sub;
        if (bx < d_gx) ig_dgrad_body(d, bx, by, lds_d);
    }
}

}  // namespace cae
