// kernels_s2.h — specialised gfx950 kernels for the stride-2 ConvTranspose2d layers with few
// channels (the last decoder layers: 8->4, 4->2, 2->1 at 63..256 px), which carry >80 % of the
// bytes of a training step.  These layers are pure HBM streaming (<= 18 FLOP per output), so the
// design goal is: every activation byte crosses the memory system once per pass, coalesced, with
// BatchNorm/ReLU/sigmoid/MSE/masking folded into the load or the store of the neighbouring conv.
//
// Forward (k_s2_fwd): sub-pixel decomposition.  With stride 2 the output pixel (2m+py, 2n+px)
// only sees inputs (m-j, n-i), j,i in {0,1}, through taps (py+2j, px+2i):
//     out[co][2m+py][2n+px] = b[co] + sum_ci sum_j,i a[ci][m-j][n-i] * W[ci][co][py+2j][px+2i]
// One thread owns the "quad" (m,n): it loads the 2x2 input neighbourhood of every input channel
// once (BN+ReLU applied in registers), produces the 2x2 output block of every output channel,
// and stores pairs of adjacent pixels.  Lanes run along n, so loads and stores are contiguous.
//
// Backward (k_s2_bwd): one pass that computes, from the gradient map gy of the layer's output,
//   (a) the gradient wrt the layer's input  ga[ci][y][x] = sum_co,k gy[co][2y+ky][2x+kx] W[ci][co][k]
//       masked by the producer's ReLU and reduced into the producer's BatchNorm sums, and
//   (b) the weight gradient dW[ci][co][k] = sum_pixels a[ci][y][x] * gy[co][2y+ky][2x+kx],
// with the BatchNorm-backward transform of gy applied once per element while the tile is staged
// into LDS (NCHW -> LDS patch tile).  Accumulators stay in registers across all tiles a
// workgroup walks; one wave-shuffle + LDS reduction and one fp64 atomic per value at the end.
#pragma once
#include "kernels_generic.h"

namespace cae {

// One patch row (KW = 3 or 4 contiguous floats) as ONE 12- / 16-byte load: 4-byte alignment is all global memory needs, and the
// count of vector-memory instructions — not their bytes — is what the backward kernels' patch fetch costs (each 64-lane load
// occupies the CU's address pipe for ~16 cycles; a thread of the 8->4 layer's backward fetched 72 dwords per pixel).
struct __attribute__((packed, aligned(4))) S2Row3 {
    float v[3];
};
struct __attribute__((packed, aligned(4))) S2Row4 {
    float v[4];
};
template <int KW>
__device__ __forceinline__ void s2_load_row(const float* __restrict__ p, float (&dst)[KW]) {
    if constexpr (KW == 3) {
        const S2Row3 r = *reinterpret_cast<const S2Row3*>(p);
        dst[0] = r.v[0]; dst[1] = r.v[1]; dst[2] = r.v[2];
    } else if constexpr (KW == 4) {
        const S2Row4 r = *reinterpret_cast<const S2Row4*>(p);
        dst[0] = r.v[0]; dst[1] = r.v[1]; dst[2] = r.v[2]; dst[3] = r.v[3];
    } else {
#pragma unroll
        for (int kx = 0; kx < KW; kx++) dst[kx] = p[kx];
    }
}


struct S2Fwd {
    int B, H, W, OH, OW;     // input map H x W, output map OH x OW (per channel)
    int tiles_x, tiles_y;    // quad tiles per image
    int total_tiles;
    const float* in;         // (B, CIN, H, W) raw output of the producer (or plain activations)
    const float* w;          // (CIN, COUT, KH, KW)
    const float* bias;       // (COUT)
    float* out;              // (B, COUT, OH, OW): raw output | dL/d(pre-sigmoid) | sigmoid output
    BnDesc bn_in;            // BatchNorm(+ReLU) sitting on `in` (BN_NONE: identity)
    double* stats;           // EPI 0: [COUT][4] sums of the BatchNorm that follows
    // EPI 1/2 (last layer):
    const float* target;     // dataset targets (N, COUT, OH, OW) or nullptr
    const int* perm;
    int use_cursor;
    double* losses;
    float inv_count;
    double* bias_acc;        // sharded: [kStatShards][bias_stride], this layer's bias at bias_acc[shard*stride + co]
    int bias_stride;
    const StepState* st;
    int epi;                 // S2_RAW_STATS | S2_RAW | S2_SIGMSE | S2_SIGOUT
};

enum S2Epi : int { S2_RAW_STATS = 0, S2_SIGMSE = 1, S2_SIGOUT = 2, S2_RAW = 3 };

__device__ __forceinline__ float wave_sum_f(float v) { return wave_sum_dpp(v); }

// store two horizontally adjacent floats; vectorised when the address allows (wave-uniform test:
// all lanes of a wave share the row, and columns are even)
__device__ __forceinline__ void store_pair(float* p, float v0, float v1, bool both) {
    if (both) {
        if ((reinterpret_cast<uintptr_t>(p) & 7) == 0) {
            *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
        } else {
            p[0] = v0;
            p[1] = v1;
        }
    } else {
        p[0] = v0;
    }
}

// a.epi (uniform): S2_RAW_STATS = raw output + BatchNorm sums, S2_RAW = raw output only (eval),
// S2_SIGMSE = sigmoid + MSE + gradient (train, last layer), S2_SIGOUT = sigmoid output (+ optional
// loss) (eval, last layer)
template <int CIN, int COUT, int KH, int KW, int TW>
__global__ void __launch_bounds__(256) k_s2_fwd(S2Fwd a) {
    const int EPI = a.epi;
    constexpr int TH = 256 / TW;
    __shared__ float4 cin4[CIN];
    __shared__ double red[4 * 2 * COUT];
    const bool designated = blockIdx.x == 0;
    {
        BnDesc d = a.bn_in;
        bn_consts(d, cin4, designated);
    }
    __syncthreads();

    const int tiles = a.tiles_x * a.tiles_y;
    double r[2 * COUT];
#pragma unroll
    for (int i = 0; i < 2 * COUT; i++) r[i] = 0.0;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int b = tile / tiles;
    const int t = tile - b * tiles;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int n = tx * TW + (threadIdx.x % TW);
    const int m = ty * TH + (threadIdx.x / TW);
    const bool active = (2 * m < a.OH) && (2 * n < a.OW);

    float acc[COUT][2][2];
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        const float bv = a.bias[co];
        acc[co][0][0] = acc[co][0][1] = acc[co][1][0] = acc[co][1][1] = bv;
    }
    {
        const bool r0 = active && m < a.H, r1 = active && m >= 1 && m - 1 < a.H;   // rows m (j=0) and m-1 (j=1) exist
        const bool c0 = n < a.W, c1 = n >= 1 && n - 1 < a.W;                       // cols n (i=0) and n-1 (i=1) exist
        // Every load of the quad's neighbourhood - all input channels - is issued before the first use, from clamped
        // (always valid) addresses; validity is one select after the BatchNorm transform.  A load under a predicate is a
        // branch plus a wait at its consumer: with the channel loop not unrolled that was one memory round trip per
        // input channel (eight for the 8->4 layer).
        const unsigned HWu = (unsigned)(a.H * a.W);
        const unsigned ro0 = (unsigned)min(m, a.H - 1) * a.W, ro1 = (unsigned)min(max(m - 1, 0), a.H - 1) * a.W;
        const unsigned co0 = (unsigned)min(n, a.W - 1), co1 = (unsigned)min(max(n - 1, 0), a.W - 1);
        const float* base = a.in + (size_t)b * CIN * HWu;
        float vin[CIN][2][2];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            const float* p = base + ci * HWu;
            vin[ci][0][0] = p[ro0 + co0];
            vin[ci][0][1] = p[ro0 + co1];
            vin[ci][1][0] = p[ro1 + co0];
            vin[ci][1][1] = p[ro1 + co1];
        }
        // weights are wave-uniform (scalar loads, 36 per input channel for a 4-channel 3x3 layer); the channel loop is
        // unrolled so that the register array is indexed statically
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            float v[2][2];
            v[0][0] = vin[ci][0][0]; v[0][1] = vin[ci][0][1]; v[1][0] = vin[ci][1][0]; v[1][1] = vin[ci][1][1];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[ci];
                v[0][0] = fmaxf(0.f, fmaf(v[0][0] - k.x, k.y, k.z));
                v[0][1] = fmaxf(0.f, fmaf(v[0][1] - k.x, k.y, k.z));
                v[1][0] = fmaxf(0.f, fmaf(v[1][0] - k.x, k.y, k.z));
                v[1][1] = fmaxf(0.f, fmaf(v[1][1] - k.x, k.y, k.z));
            }
            v[0][0] = (r0 && c0) ? v[0][0] : 0.f;
            v[0][1] = (r0 && c1) ? v[0][1] : 0.f;
            v[1][0] = (r1 && c0) ? v[1][0] : 0.f;
            v[1][1] = (r1 && c1) ? v[1][1] : 0.f;
            const float* wc = a.w + (size_t)ci * COUT * KH * KW;
#pragma unroll
            for (int co = 0; co < COUT; co++) {
#pragma unroll
                for (int py = 0; py < 2; py++) {
#pragma unroll
                    for (int px = 0; px < 2; px++) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                if (py + 2 * j < KH && px + 2 * i < KW)
                                    acc[co][py][px] = fmaf(v[j][i], wc[(co * KH + py + 2 * j) * KW + px + 2 * i],
                                                           acc[co][py][px]);
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue
    const int oy = 2 * m, ox = 2 * n;
    const bool row1 = active && (oy + 1 < a.OH);
    const bool col1 = active && (ox + 1 < a.OW);
    if (EPI == S2_RAW_STATS || EPI == S2_RAW) {
        if (active) {
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                float* o = a.out + ((size_t)(b * COUT + co) * a.OH + oy) * a.OW + ox;
                store_pair(o, acc[co][0][0], acc[co][0][1], col1);
                if (row1) store_pair(o + a.OW, acc[co][1][0], acc[co][1][1], col1);
                if (EPI == S2_RAW_STATS) {
                    float s1 = acc[co][0][0], s2 = acc[co][0][0] * acc[co][0][0];
                    if (col1) { s1 += acc[co][0][1]; s2 = fmaf(acc[co][0][1], acc[co][0][1], s2); }
                    if (row1) {
                        s1 += acc[co][1][0]; s2 = fmaf(acc[co][1][0], acc[co][1][0], s2);
                        if (col1) { s1 += acc[co][1][1]; s2 = fmaf(acc[co][1][1], acc[co][1][1], s2); }
                    }
                    r[2 * co] += (double)s1;
                    r[2 * co + 1] += (double)s2;
                }
            }
        }
    } else {
        if (active) {
            size_t tb = 0;
            if (a.target) tb = sample_of(a.perm, a.use_cursor, a.st, b);
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                float* o = a.out ? a.out + ((size_t)(b * COUT + co) * a.OH + oy) * a.OW + ox : nullptr;
                const float* tp = a.target ? a.target + ((tb * COUT + co) * (size_t)a.OH + oy) * a.OW + ox : nullptr;
                float lsum = 0.f, gsum = 0.f;
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    if (py == 1 && !row1) break;
                    float res[2];
#pragma unroll
                    for (int px = 0; px < 2; px++) {
                        const bool ok = px == 0 || col1;
                        const float yh = 1.0f / (1.0f + expf(-acc[co][py][px]));
                        float outv = yh;
                        if (tp && ok) {
                            const float d = yh - tp[py * a.OW + px];
                            lsum = fmaf(d, d, lsum);
                            if (EPI == S2_SIGMSE) {
                                outv = (2.0f * d * a.inv_count) * (yh * (1.0f - yh));
                                gsum += outv;
                            }
                        }
                        res[px] = outv;
                    }
                    if (o) store_pair(o + py * a.OW, res[0], res[1], col1);
                }
                r[2 * co] += (double)lsum * (double)a.inv_count;
                r[2 * co + 1] += (double)gsum;
            }
        }
    }

    }  // tile loop

    if (EPI != S2_RAW) {
        if (EPI == S2_SIGOUT && !a.target) return;
        // block reduction of the 2*COUT partials, then one fp64 atomic each
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 2 * COUT; i++) {
            const double s = wave_sum(r[i]);
            if (lane == 0) red[wv * 2 * COUT + i] = s;
        }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            const int i = threadIdx.x;
            const double s = red[i] + red[2 * COUT + i] + red[4 * COUT + i] + red[6 * COUT + i];
            const int co = i >> 1;
            if (EPI == S2_RAW_STATS) {
                acc_add<ACC_STAT>(&a.stats[((size_t)(blockIdx.x & (kStatShards - 1)) * COUT + co) * 4 + (i & 1)], s);
            } else {
                const int shard = blockIdx.x & (kStatShards - 1);
                if ((i & 1) == 0) {
                    acc_add<ACC_GRAD>(&a.losses[(size_t)a.st->loss_slot * kStatShards + shard], s);
                } else if (EPI == S2_SIGMSE) {
                    acc_add<ACC_GRAD>(&a.bias_acc[(size_t)shard * a.bias_stride + co], s);
                }
            }
        }
    }
}

// ---- forward with the output channels split over the waves ----------------------------------------------------------
// For the intermediate thin layers (8->4, 4->2; raw output + BatchNorm sums).  Wave group g of the workgroup computes
// output channel g for the SAME quads as the other groups: a thread then needs only CIN*KH*KW wave-uniform weights (72 for
// 8->4 at 3x3: they fit the scalar registers, where all 288 did not), keeps CIN*4 inputs + 4 accumulators in vector
// registers (occupancy 8 instead of 1-4), and all of its loads are in flight at once.  The COUT-fold re-read of the
// inputs is served by the CU's L1 (the inputs are the small side of these layers).
template <int CIN, int COUT, int KH, int KW, int TW>
__global__ void __launch_bounds__(256) k_s2_fwd_cs(S2Fwd a) {
    constexpr int PIX = 256 / COUT;      // quads per tile
    constexpr int TH = PIX / TW;
    constexpr int WPG = PIX / 64;        // waves per channel group
    static_assert(PIX % 64 == 0 && TH >= 1, "a wave must not straddle channel groups");
    __shared__ float4 cin4[CIN];
    __shared__ double red[4 * 2];
    bn_consts(a.bn_in, cin4, blockIdx.x == 0);
    __syncthreads();

    const int co = __builtin_amdgcn_readfirstlane(threadIdx.x / PIX);   // wave-uniform: weights stay scalar
    const int pix = threadIdx.x - co * PIX;
    const int tiles = a.tiles_x * a.tiles_y;
    const unsigned HWu = (unsigned)(a.H * a.W);
    const float bv = a.bias[co];
    const bool stats = a.epi == S2_RAW_STATS;
    double r1 = 0.0, r2 = 0.0;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int b = tile / tiles;
        const int t = tile - b * tiles;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int n = tx * TW + (pix % TW);
        const int m = ty * TH + (pix / TW);
        const bool active = (2 * m < a.OH) && (2 * n < a.OW);
        const bool r0 = active && m < a.H, r1ok = active && m >= 1 && m - 1 < a.H;
        const bool c0 = n < a.W, c1 = n >= 1 && n - 1 < a.W;
        const unsigned ro0 = (unsigned)min(m, a.H - 1) * a.W, ro1 = (unsigned)min(max(m - 1, 0), a.H - 1) * a.W;
        const unsigned co0 = (unsigned)min(n, a.W - 1), co1 = (unsigned)min(max(n - 1, 0), a.W - 1);
        const float* base = a.in + (size_t)b * CIN * HWu;
        float vin[CIN][2][2];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {      // clamped, unconditional: every load in flight before the first use
            const float* p = base + ci * HWu;
            vin[ci][0][0] = p[ro0 + co0];
            vin[ci][0][1] = p[ro0 + co1];
            vin[ci][1][0] = p[ro1 + co0];
            vin[ci][1][1] = p[ro1 + co1];
        }
        float acc[2][2] = {{bv, bv}, {bv, bv}};
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            float v[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    float x = vin[ci][j][i];
                    if (a.bn_in.mode != BN_NONE) {
                        const float4 k = cin4[ci];
                        x = fmaxf(0.f, fmaf(x - k.x, k.y, k.z));
                    }
                    v[j][i] = ((j ? r1ok : r0) && (i ? c1 : c0)) ? x : 0.f;
                }
            const float* wc = a.w + ((size_t)ci * COUT + co) * KH * KW;
#pragma unroll
            for (int py = 0; py < 2; py++)
#pragma unroll
                for (int px = 0; px < 2; px++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int i = 0; i < 2; i++)
                            if (py + 2 * j < KH && px + 2 * i < KW)
                                acc[py][px] = fmaf(v[j][i], wc[(py + 2 * j) * KW + px + 2 * i], acc[py][px]);
        }
        if (active) {
            const int oy = 2 * m, ox = 2 * n;
            const bool row1 = oy + 1 < a.OH, col1 = ox + 1 < a.OW;
            float* o = a.out + ((size_t)(b * COUT + co) * a.OH + oy) * a.OW + ox;
            store_pair(o, acc[0][0], acc[0][1], col1);
            if (row1) store_pair(o + a.OW, acc[1][0], acc[1][1], col1);
            if (stats) {
                float s1 = acc[0][0], s2 = acc[0][0] * acc[0][0];
                if (col1) { s1 += acc[0][1]; s2 = fmaf(acc[0][1], acc[0][1], s2); }
                if (row1) {
                    s1 += acc[1][0]; s2 = fmaf(acc[1][0], acc[1][0], s2);
                    if (col1) { s1 += acc[1][1]; s2 = fmaf(acc[1][1], acc[1][1], s2); }
                }
                r1 += (double)s1;
                r2 += (double)s2;
            }
        }
    }
    if (!stats) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        const double t1 = wave_sum_lane63(r1), t2 = wave_sum_lane63(r2);
        if (lane == 63) {
            red[wv * 2] = t1;
            red[wv * 2 + 1] = t2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * COUT) {
        const int c = threadIdx.x >> 1, i = threadIdx.x & 1;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < WPG; w++) sum += red[(c * WPG + w) * 2 + i];
        acc_add<ACC_STAT>(&a.stats[((size_t)(blockIdx.x & (kStatShards - 1)) * COUT + c) * 4 + i], sum);
    }
}

// =================================================================================================
// fused backward of a thin stride-2 ConvTranspose2d layer
// =================================================================================================
struct S2Bwd {
    int B, H, W, OH, OW;     // layer input map H x W (the "small" map), output map OH x OW
    int tiles_x, tiles_y, total_tiles;
    const float* g;          // (B, COUT, OH, OW) masked upstream gradient (or dL/dy for the last layer)
    const float* yout;       // (B, COUT, OH, OW) raw forward output of this layer (BN_BWD transform), or nullptr
    BnDesc bn_out;           // BN_BWD descriptor of this layer's BatchNorm, or BN_NONE
    const float* ain;        // (B, CIN, H, W) raw output of the producer layer (this layer's input before BN+ReLU)
    BnDesc bn_in;            // BN_SAVED descriptor of the producer's BatchNorm, or BN_NONE (plain input)
    const float* w;          // (CIN, COUT, KH, KW)
    float* gin;              // (B, CIN, H, W) gradient wrt the producer's raw output side: masked by its ReLU
    double* stats_in;        // producer's [CIN][4] sums (slots 2,3), or nullptr when bn_in is BN_NONE
    double* wacc;            // sharded fp64 accumulator of dW: [kStatShards][wacc_stride]
    int wacc_stride;
    BnGradOut bg;            // BatchNorm parameter gradients of this layer (published by block 0)
};

// Tile: TPY x TPX input pixels per pass, one pixel per thread per ci-group.
//   threads = TPY*TPX*CG, CG = CIN / CT ci-groups (each thread owns CT input channels)
template <int CIN, int CT, int COUT, int KH, int KW, int TPX, int TPY>
__global__ void __launch_bounds__(256) k_s2_bwd(S2Bwd a) {
    constexpr int CG = CIN / CT;
    static_assert(TPX * TPY * CG == 256, "256 threads");
    constexpr int LW = 2 * TPX + KW - 2;   // staged gy tile width
    constexpr int LH = 2 * TPY + KH - 2;
    constexpr int LWH = (LW + 1) / 2;      // even and odd columns are kept in separate half-rows, so the
    constexpr int LWP = 2 * LWH + 1;       // stride-2 patch reads of a wave are contiguous (no bank conflicts)
    constexpr int NACC = CT * COUT * KH * KW;
    __shared__ float tile[COUT * LH * LWP];
    __shared__ float4 cout4[COUT];
    __shared__ float4 cin4[CIN];
    __shared__ float redf[4 * (NACC + 2 * CT)];
    __shared__ __attribute__((aligned(16))) float wl[COUT * KH * KW * CIN];  // [co][ky][kx][ci]

    for (int e = threadIdx.x; e < COUT * KH * KW * CIN; e += 256) {
        const int ci = e % CIN, tap = e / CIN;  // tap = (co*KH + ky)*KW + kx
        wl[e] = a.w[(size_t)ci * COUT * KH * KW + tap];
    }
    bn_consts(a.bn_out, cout4, false);
    bn_consts(a.bn_in, cin4, false, 64);
    if (blockIdx.x == 0 && a.bg.stats) {
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

    // ci-group of this thread; wave-uniform (a wave never straddles groups), so weights stay scalar
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / (TPX * TPY));
    const int pix = threadIdx.x - grp * (TPX * TPY);
    const int ly = pix / TPX, lx = pix - ly * TPX;
    const int tiles = a.tiles_x * a.tiles_y;

    float dw[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) dw[i] = 0.f;
    float d1[CT], d2[CT];
#pragma unroll
    for (int i = 0; i < CT; i++) d1[i] = d2[i] = 0.f;

    for (int t = blockIdx.x; t < a.total_tiles; t += gridDim.x) {
        const int b = t / tiles;
        const int tt = t - b * tiles;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int y0 = ty * TPY, x0 = tx * TPX;
        __syncthreads();  // previous tile fully consumed (also orders the bn_consts writes)
        // ---- stage gy[co][2*y0 .. +LH)[2*x0 .. +LW) with the BatchNorm-backward transform
        for (int e = threadIdx.x; e < COUT * LH * LW; e += 256) {
            const int co = e / (LH * LW);
            const int r = e - co * (LH * LW);
            const int yy = r / LW, xx = r - yy * LW;
            const int gy_ = 2 * y0 + yy, gx_ = 2 * x0 + xx;
            float v = 0.f;
            if (gy_ < a.OH && gx_ < a.OW) {
                const size_t off = ((size_t)(b * COUT + co) * a.OH + gy_) * a.OW + gx_;
                v = a.g[off];
                if (a.bn_out.mode == BN_BWD) {
                    const float4 k = cout4[co];
                    v = k.y * v - k.z - (a.yout[off] - k.x) * k.w;
                }
            }
            tile[(co * LH + yy) * LWP + (xx & 1) * LWH + (xx >> 1)] = v;
        }
        __syncthreads();
        const int y = y0 + ly, x = x0 + lx;
        if (y < a.H && x < a.W) {
            // this thread's input activations (BN + ReLU of the producer's raw output)
            float av[CT], yraw[CT];
#pragma unroll
            for (int c = 0; c < CT; c++) {
                const int ci = grp * CT + c;
                const size_t off = ((size_t)(b * CIN + ci) * a.H + y) * a.W + x;
                yraw[c] = a.ain[off];
                av[c] = yraw[c];
                if (a.bn_in.mode != BN_NONE) {
                    const float4 k = cin4[ci];
                    av[c] = fmaxf(0.f, fmaf(yraw[c] - k.x, k.y, k.z));
                }
            }
            float ga[CT];
#pragma unroll
            for (int c = 0; c < CT; c++) ga[c] = 0.f;
#pragma unroll
            for (int co = 0; co < COUT; co++) {
#pragma unroll
                for (int ky = 0; ky < KH; ky++) {
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) {
                        const float p = tile[(co * LH + 2 * ly + ky) * LWP + (kx & 1) * LWH + lx + (kx >> 1)];
                        const float* wp = &wl[((co * KH + ky) * KW + kx) * CIN + grp * CT];
#pragma unroll
                        for (int c = 0; c < CT; c++) {
                            const float wv = wp[c];
                            ga[c] = fmaf(p, wv, ga[c]);
                            dw[((c * COUT + co) * KH + ky) * KW + kx] =
                                fmaf(av[c], p, dw[((c * COUT + co) * KH + ky) * KW + kx]);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CT; c++) {
                const int ci = grp * CT + c;
                const size_t off = ((size_t)(b * CIN + ci) * a.H + y) * a.W + x;
                float gv = ga[c];
                if (a.bn_in.mode != BN_NONE) {
                    const float4 k = cin4[ci];
                    const float d = yraw[c] - k.x;
                    gv = fmaf(d, k.y, k.z) > 0.f ? gv : 0.f;
                    d1[c] += gv;
                    d2[c] = fmaf(gv, d * k.w, d2[c]);
                }
                a.gin[off] = gv;
            }
        }
    }

    // ---- reduce accumulators over the block: wave shuffle, then the 4 waves through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NRED = NACC + 2 * CT;
    // all lanes of a wave belong to one ci-group when TPX*TPY >= 64
    static_assert((TPX * TPY) % 64 == 0, "a wave must not straddle ci-groups");
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        const float s = wave_sum_f(dw[i]);
        if (lane == 0) redf[wv * NRED + i] = s;
    }
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const float s1 = wave_sum_f(d1[c]), s2 = wave_sum_f(d2[c]);
        if (lane == 0) {
            redf[wv * NRED + NACC + 2 * c] = s1;
            redf[wv * NRED + NACC + 2 * c + 1] = s2;
        }
    }
    __syncthreads();
    constexpr int WPG = (TPX * TPY) / 64;  // waves per ci-group
    for (int i = threadIdx.x; i < CG * NRED; i += 256) {
        const int gsel = i / NRED, j = i - gsel * NRED;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < WPG; q++) s += (double)redf[(gsel * WPG + q) * NRED + j];
        if (j < NACC) {
            // j = ((c*COUT + co)*KH + ky)*KW + kx  ->  global weight index with ci = gsel*CT + c
            const int c = j / (COUT * KH * KW), rest = j - c * (COUT * KH * KW);
            const int ci = gsel * CT + c;
            acc_add<ACC_GRAD>(&a.wacc[(size_t)(blockIdx.x & (kStatShards - 1)) * a.wacc_stride + (size_t)ci * COUT * KH * KW + rest], s);
        } else if (a.stats_in) {
            const int jj = j - NACC;
            const int ci = gsel * CT + (jj >> 1);
            acc_add<ACC_GRAD>(&a.stats_in[((size_t)(blockIdx.x & (kStatShards - 1)) * CIN + ci) * 4 + 2 + (jj & 1)], s);
        }
    }
}


// =================================================================================================
// Second-generation kernels for the thinnest layers: instruction count per output, not bytes, was
// the limit of k_s2_fwd / k_s2_bwd (150 instructions per output pixel, a barrier pair per tile).
//   * k_s2_fwd2: each thread owns 2x2 quads = a 4x4 output block per channel, from a 3x3 input
//     neighbourhood per input channel; no branches (clamped addresses + selects), 32-bit offsets,
//     16-byte stores / target loads when the row pitch allows.
//   * k_s2_bwd2: no LDS staging and no barriers in the loop: each thread owns one input pixel and
//     reads its kh x kw gradient patch straight from global memory (the 2-4x overlap between
//     neighbouring patches is served by L1/L2), so latency is covered by occupancy alone.
// =================================================================================================

__device__ __forceinline__ float sigmoid_fast(float x) {
    // 1 / (1 + 2^(-x log2 e)): v_exp_f32 and v_rcp_f32 are good to 1 ulp and the product's rounding error scales with |x|
    // (~1e-7 |x| relative on e^-x), ample for a [0,1] output compared at 1e-5; saturates cleanly (2^-inf = 0, 1/inf = 0).
    // The library expf costs ~15 instructions of range reduction per call, 16 calls per thread in the last layer.
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

template <int CIN, int COUT, int KH, int KW, int TWL>
__global__ void __launch_bounds__(256) k_s2_fwd2(S2Fwd a) {
    constexpr int TROWS = 256 / TWL;  // thread rows per block; each thread covers 2 quad rows x 2 quad cols
    const int EPI = a.epi;
    __shared__ float4 cin4[CIN];
    __shared__ double red[4 * 2 * COUT];
    bn_consts(a.bn_in, cin4, blockIdx.x == 0);
    __syncthreads();

    const int tiles = a.tiles_x * a.tiles_y;
    const unsigned HW = a.H * a.W;
    double r[2 * COUT];
#pragma unroll
    for (int i = 0; i < 2 * COUT; i++) r[i] = 0.0;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int b = tile / tiles;
        const int t = tile - b * tiles;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int n0 = 2 * (tx * TWL + (threadIdx.x % TWL));   // first quad column
        const int m0 = 2 * (ty * TROWS + (threadIdx.x / TWL)); // first quad row
        const int oy0 = 2 * m0, ox0 = 2 * n0;
        const bool active = oy0 < a.OH && ox0 < a.OW;

        float acc[COUT][4][4];
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            const float bv = a.bias[co];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) acc[co][y][x] = bv;
        }
        // input rows m0-1, m0, m0+1 and columns n0-1, n0, n0+1: clamped offsets + validity masks
        unsigned roff[3], coff[3];
        bool rok[3], cok[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int yy = m0 - 1 + d, xx = n0 - 1 + d;
            rok[d] = active && yy >= 0 && yy < a.H;
            cok[d] = xx >= 0 && xx < a.W;
            roff[d] = (unsigned)min(max(yy, 0), a.H - 1) * (unsigned)a.W;
            coff[d] = (unsigned)min(max(xx, 0), a.W - 1);
        }
        const float* inb = a.in + (size_t)b * CIN * HW;
#pragma unroll(CIN * COUT * KH * KW <= 80 ? CIN : 1)
        for (int ci = 0; ci < CIN; ci++) {
            const float* p = inb + ci * HW;
            float v[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) v[dy][dx] = p[roff[dy] + coff[dx]];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[ci];
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) v[dy][dx] = fmaxf(0.f, fmaf(v[dy][dx] - k.x, k.y, k.z));
            }
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) v[dy][dx] = (rok[dy] && cok[dx]) ? v[dy][dx] : 0.f;
            const float* wc = a.w + (size_t)ci * COUT * KH * KW;
#pragma unroll
            for (int co = 0; co < COUT; co++) {
#pragma unroll
                for (int y = 0; y < 4; y++) {        // output row 2*m0 + y: quad row y>>1, parity y&1
#pragma unroll
                    for (int x = 0; x < 4; x++) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {
#pragma unroll
                            for (int i = 0; i < 2; i++) {
                                const int ky = (y & 1) + 2 * j, kx = (x & 1) + 2 * i;
                                if (ky < KH && kx < KW)
                                    acc[co][y][x] = fmaf(v[(y >> 1) + 1 - j][(x >> 1) + 1 - i],
                                                         wc[(co * KH + ky) * KW + kx], acc[co][y][x]);
                            }
                        }
                    }
                }
            }
        }
        if (!active) continue;
        // ---- epilogue: 4 rows x 4 columns per output channel
        const bool vec4 = (a.OW & 3) == 0;   // rows are 16-byte aligned and ox0 is a multiple of 4
        size_t tb = 0;
        if (a.target) tb = sample_of(a.perm, a.use_cursor, a.st, b);
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int y = 0; y < 4; y++) {
                const int oy = oy0 + y;
                if (oy >= a.OH) break;
                const size_t orow = ((size_t)(b * COUT + co) * a.OH + oy) * a.OW + ox0;
                float res[4];
                if (EPI == S2_RAW_STATS || EPI == S2_RAW) {
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        res[x] = acc[co][y][x];
                        if (ox0 + x < a.OW) { s1 += res[x]; s2 = fmaf(res[x], res[x], s2); }
                    }
                } else {
                    float tv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.target) {
                        const float* tp = a.target + ((tb * COUT + co) * (size_t)a.OH + oy) * a.OW + ox0;
                        if (vec4) {
                            const float4 t4 = *reinterpret_cast<const float4*>(tp);
                            tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
                        } else {
#pragma unroll
                            for (int x = 0; x < 4; x++) tv[x] = (ox0 + x < a.OW) ? tp[x] : 0.f;
                        }
                    }
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const float yh = sigmoid_fast(acc[co][y][x]);
                        res[x] = yh;
                        if (a.target && ox0 + x < a.OW) {
                            const float d = yh - tv[x];
                            s1 = fmaf(d, d, s1);
                            if (EPI == S2_SIGMSE) {
                                res[x] = (2.0f * d * a.inv_count) * (yh * (1.0f - yh));
                                s2 += res[x];
                            }
                        }
                    }
                }
                if (a.out) {
                    float* o = a.out + orow;
                    if (vec4) {
                        *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
                    } else {
#pragma unroll
                        for (int x = 0; x < 4; x++)
                            if (ox0 + x < a.OW) o[x] = res[x];
                    }
                }
            }
            if (EPI == S2_RAW_STATS) {
                r[2 * co] += (double)s1;
                r[2 * co + 1] += (double)s2;
            } else if (EPI != S2_RAW) {
                r[2 * co] += (double)s1 * (double)a.inv_count;
                r[2 * co + 1] += (double)s2;
            }
        }
    }

    if (EPI == S2_RAW) return;
    if (EPI == S2_SIGOUT && !a.target) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2 * COUT; i++) {
        const double s = wave_sum(r[i]);
        if (lane == 0) red[wv * 2 * COUT + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2 * COUT) {
        const int i = threadIdx.x;
        const double s = red[i] + red[2 * COUT + i] + red[4 * COUT + i] + red[6 * COUT + i];
        const int co = i >> 1;
        const int shard = blockIdx.x & (kStatShards - 1);
        if (EPI == S2_RAW_STATS) {
            acc_add<ACC_STAT>(&a.stats[((size_t)shard * COUT + co) * 4 + (i & 1)], s);
        } else if ((i & 1) == 0) {
            acc_add<ACC_GRAD>(&a.losses[(size_t)a.st->loss_slot * kStatShards + shard], s);
        } else if (EPI == S2_SIGMSE) {
            acc_add<ACC_GRAD>(&a.bias_acc[(size_t)shard * a.bias_stride + co], s);
        }
    }
}

// ---- direct (LDS-free) fused backward for CIN <= 4 ------------------------------------------------
template <int CIN, int COUT, int KH, int KW, int TWL>
__global__ void __launch_bounds__(256) k_s2_bwd2(S2Bwd a) {
    constexpr int TROWS = 256 / TWL;
    constexpr int NACC = CIN * COUT * KH * KW;
    constexpr int NRED = NACC + 2 * CIN;
    __shared__ float4 cout4[COUT];
    __shared__ float4 cin4[CIN];
    __shared__ float redf[4 * NRED];

    bn_consts(a.bn_out, cout4, false);
    bn_consts(a.bn_in, cin4, false, 64);
    if (blockIdx.x == 0 && a.bg.stats) {
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

    const int tiles = a.tiles_x * a.tiles_y;
    const unsigned HW = a.H * a.W, OHW = a.OH * a.OW;
    float dw[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) dw[i] = 0.f;
    float d1[CIN], d2[CIN];
#pragma unroll
    for (int i = 0; i < CIN; i++) d1[i] = d2[i] = 0.f;

    for (int t = blockIdx.x; t < a.total_tiles; t += gridDim.x) {
        const int b = t / tiles;
        const int tt = t - b * tiles;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int x = tx * TWL + (threadIdx.x % TWL);
        const int y = ty * TROWS + (threadIdx.x / TWL);
        if (y >= a.H || x >= a.W) continue;
        const unsigned poff = (unsigned)y * a.W + x;
        float av[CIN], yraw[CIN], ga[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            yraw[ci] = a.ain[(size_t)(b * CIN + ci) * HW + poff];
            av[ci] = yraw[ci];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[ci];
                av[ci] = fmaxf(0.f, fmaf(yraw[ci] - k.x, k.y, k.z));
            }
            ga[ci] = 0.f;
        }
        const unsigned pbase = (unsigned)(2 * y) * a.OW + 2 * x;
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            const float* gp = a.g + (size_t)(b * COUT + co) * OHW + pbase;
            const float* yp = a.yout ? a.yout + (size_t)(b * COUT + co) * OHW + pbase : nullptr;
            float p[KH][KW];
#pragma unroll
            for (int ky = 0; ky < KH; ky++) s2_load_row<KW>(gp + ky * a.OW, p[ky]);
            if (a.bn_out.mode == BN_BWD) {
                const float4 k = cout4[co];
                float yr[KH][KW];
#pragma unroll
                for (int ky = 0; ky < KH; ky++) s2_load_row<KW>(yp + ky * a.OW, yr[ky]);
#pragma unroll
                for (int ky = 0; ky < KH; ky++)
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) p[ky][kx] = k.y * p[ky][kx] - k.z - (yr[ky][kx] - k.x) * k.w;
            }
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                const float* wc = a.w + ((size_t)ci * COUT + co) * KH * KW;
#pragma unroll
                for (int ky = 0; ky < KH; ky++)
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) {
                        ga[ci] = fmaf(p[ky][kx], wc[ky * KW + kx], ga[ci]);
                        dw[((ci * COUT + co) * KH + ky) * KW + kx] =
                            fmaf(av[ci], p[ky][kx], dw[((ci * COUT + co) * KH + ky) * KW + kx]);
                    }
            }
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
            float gv = ga[ci];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[ci];
                const float d = yraw[ci] - k.x;
                gv = fmaf(d, k.y, k.z) > 0.f ? gv : 0.f;
                d1[ci] += gv;
                d2[ci] = fmaf(gv, d * k.w, d2[ci]);
            }
            a.gin[(size_t)(b * CIN + ci) * HW + poff] = gv;
        }
    }

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        const float s = wave_sum_f(dw[i]);
        if (lane == 0) redf[wv * NRED + i] = s;
    }
#pragma unroll
    for (int c = 0; c < CIN; c++) {
        const float s1 = wave_sum_f(d1[c]), s2 = wave_sum_f(d2[c]);
        if (lane == 0) {
            redf[wv * NRED + NACC + 2 * c] = s1;
            redf[wv * NRED + NACC + 2 * c + 1] = s2;
        }
    }
    __syncthreads();
    const int shard = blockIdx.x & (kStatShards - 1);
    for (int j = threadIdx.x; j < NRED; j += 256) {
        const double s = (double)redf[j] + (double)redf[NRED + j] + (double)redf[2 * NRED + j] + (double)redf[3 * NRED + j];
        if (j < NACC) {
            acc_add<ACC_GRAD>(&a.wacc[(size_t)shard * a.wacc_stride + j], s);
        } else if (a.stats_in) {
            const int jj = j - NACC;
            acc_add<ACC_GRAD>(&a.stats_in[((size_t)shard * CIN + (jj >> 1)) * 4 + 2 + (jj & 1)], s);
        }
    }
}


// ---- direct (LDS-free) fused backward with the input channels split over the waves ---------------
// For Cin*Cout*kh*kw too large for one thread's registers (8->4, k3: 288 weight-gradient
// accumulators): wave w of the workgroup owns input channels [CT*g, CT*(g+1)), g = w mod CG, of the
// SAME pixels as the other waves, so each thread keeps CT*Cout*kh*kw accumulators.  The gradient patch
// is read once per channel group (4x for 8->4; the repeats hit L1/L2).  Replaces the LDS-staged
// k_s2_bwd (204 VGPRs, two barriers per 64-pixel tile) for these shapes.
template <int CIN, int CT, int COUT, int KH, int KW, int TWL>
__global__ void __launch_bounds__(256) k_s2_bwd_split(S2Bwd a) {
    constexpr int CG = CIN / CT;                 // channel groups = waves that share a pixel tile
    constexpr int PIX = 256 / CG;                // pixels per tile
    constexpr int TROWS = PIX / TWL;
    static_assert(PIX % 64 == 0, "a wave must not straddle channel groups");
    constexpr int NACC = CT * COUT * KH * KW;
    constexpr int NRED = NACC + 2 * CT;
    constexpr int WPG = PIX / 64;                // waves per group
    __shared__ float4 cout4[COUT];
    __shared__ float4 cin4[CIN];
    __shared__ float redf[4 * NRED];

    bn_consts(a.bn_out, cout4, false);
    bn_consts(a.bn_in, cin4, false, 64);
    if (blockIdx.x == 0 && a.bg.stats) {
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

    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / PIX);   // wave-uniform: weights stay scalar
    const int pix = threadIdx.x - grp * PIX;
    const int c0 = grp * CT;
    const int tiles = a.tiles_x * a.tiles_y;
    const unsigned HW = a.H * a.W, OHW = a.OH * a.OW;
    float dw[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) dw[i] = 0.f;
    float d1[CT], d2[CT];
#pragma unroll
    for (int i = 0; i < CT; i++) d1[i] = d2[i] = 0.f;

    for (int t = blockIdx.x; t < a.total_tiles; t += gridDim.x) {
        const int b = t / tiles;
        const int tt = t - b * tiles;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int x = tx * TWL + (pix % TWL);
        const int y = ty * TROWS + (pix / TWL);
        if (y >= a.H || x >= a.W) continue;
        const unsigned poff = (unsigned)y * a.W + x;
        float av[CT], yraw[CT], ga[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            yraw[c] = a.ain[(size_t)(b * CIN + c0 + c) * HW + poff];
            av[c] = yraw[c];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[c0 + c];
                av[c] = fmaxf(0.f, fmaf(yraw[c] - k.x, k.y, k.z));
            }
            ga[c] = 0.f;
        }
        const unsigned pbase = (unsigned)(2 * y) * a.OW + 2 * x;
        // every load of the pixel's patch is issued before the first use: one memory round trip per
        // pixel instead of one per output channel
        float pg[COUT][KH][KW], py[COUT][KH][KW];
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            const float* gp = a.g + (size_t)(b * COUT + co) * OHW + pbase;
#pragma unroll
            for (int ky = 0; ky < KH; ky++) s2_load_row<KW>(gp + ky * a.OW, pg[co][ky]);
        }
        if (a.bn_out.mode == BN_BWD) {
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                const float* yp = a.yout + (size_t)(b * COUT + co) * OHW + pbase;
#pragma unroll
                for (int ky = 0; ky < KH; ky++) s2_load_row<KW>(yp + ky * a.OW, py[co][ky]);
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            float p[KH][KW];
            const float4 k = a.bn_out.mode == BN_BWD ? cout4[co] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < KH; ky++)
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    float gvv = pg[co][ky][kx];
                    if (a.bn_out.mode == BN_BWD) gvv = k.y * gvv - k.z - (py[co][ky][kx] - k.x) * k.w;
                    p[ky][kx] = gvv;
                }
#pragma unroll
            for (int c = 0; c < CT; c++) {
                const float* wc = a.w + ((size_t)(c0 + c) * COUT + co) * KH * KW;
#pragma unroll
                for (int ky = 0; ky < KH; ky++)
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) {
                        ga[c] = fmaf(p[ky][kx], wc[ky * KW + kx], ga[c]);
                        dw[((c * COUT + co) * KH + ky) * KW + kx] =
                            fmaf(av[c], p[ky][kx], dw[((c * COUT + co) * KH + ky) * KW + kx]);
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; c++) {
            float gv = ga[c];
            if (a.bn_in.mode != BN_NONE) {
                const float4 k = cin4[c0 + c];
                const float d = yraw[c] - k.x;
                gv = fmaf(d, k.y, k.z) > 0.f ? gv : 0.f;
                d1[c] += gv;
                d2[c] = fmaf(gv, d * k.w, d2[c]);
            }
            a.gin[(size_t)(b * CIN + c0 + c) * HW + poff] = gv;
        }
    }

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        const float s = wave_sum_f(dw[i]);
        if (lane == 0) redf[wv * NRED + i] = s;
    }
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const float s1 = wave_sum_f(d1[c]), s2 = wave_sum_f(d2[c]);
        if (lane == 0) {
            redf[wv * NRED + NACC + 2 * c] = s1;
            redf[wv * NRED + NACC + 2 * c + 1] = s2;
        }
    }
    __syncthreads();
    const int shard = blockIdx.x & (kStatShards - 1);
    for (int i = threadIdx.x; i < CG * NRED; i += 256) {
        const int gsel = i / NRED, j = i - gsel * NRED;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WPG; w++) s += (double)redf[(gsel * WPG + w) * NRED + j];
        if (j < NACC) {
            const int c = j / (COUT * KH * KW), rest = j - c * (COUT * KH * KW);
            acc_add<ACC_GRAD>(&a.wacc[(size_t)shard * a.wacc_stride + (size_t)(gsel * CT + c) * COUT * KH * KW + rest], s);
        } else if (a.stats_in) {
            const int jj = j - NACC;
            acc_add<ACC_GRAD>(&a.stats_in[((size_t)shard * CIN + gsel * CT + (jj >> 1)) * 4 + 2 + (jj & 1)], s);
        }
    }
}

}  // namespace cae
