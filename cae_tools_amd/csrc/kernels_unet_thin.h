// kernels_unet_thin.h — the 4x4 / stride 2 / padding 1 layers of the UNET path with a handful of channels on the LARGE side
// (the 3-channel image end: Conv2d 3 -> 32 of the encoder, ConvTranspose2d 64 -> 3 of the decoder; unet.py:84-90,131-160).
// Their GEMM is 48 columns wide: in the tile engine (kernels_unet_mfma.h) they move 90-160 MB per launch at 1 TB/s, every
// im2col value a 4-byte gather.  Here the large-side map is staged ONCE per tile as a zero-bordered patch in LDS and the
// MFMA operands are read from it in place, conflict-free:
//
//   k_thin_wgrad   dW[cs][cl][ky][kx] = sum_{b,y,x} S[b][cs][y][x] * L[b][cl][2y-1+ky][2x-1+kx]
//                  (S: the small-side map, 32 or 64 channels; L: the large-side map, <= 4 channels)
//
// as D[cs][t] = sum_p S[cs][p] * patch(t, p), t = (cl, ky, kx): rows = channels, columns = taps, K = pixels.  A tile is 128
// consecutive pixels of S (128 / Ws rows); the four waves take 32 pixels each, a workgroup walks `tpw` tiles with the next
// tile's loads in flight and the accumulators in registers, then the waves' partial sums are added in wave order through LDS
// and the workgroup's Cs x 16 Cl tile goes to scratch; k_thin_fold adds the workgroups' tiles in index order (fp64): no atomics,
// the result does not depend on timing.
//
// LDS images.  S tile [cs][128 + 4] floats: a lane's ds_read_b128 gives it four k steps; sixteen rows start four banks apart.
// Patch [cl][2R + 2 rows][4 + Wl] floats: index 4 + c holds column c, indices 0-3 are zero (3 = column -1; the row's right
// border, column Wl, is index 0 of the next row), rows outside the map are zero.  An MFMA step multiplies the pixel pair
// {p, p + 16}: lanes 0-31 (32 taps of pixel p) read 32 consecutive banks - kx, then ky (row pitch = 4 mod 64), then cl (plane
// pitch = 16 mod 64) - and lanes 32-63 the other 32 (16 pixels = 32 floats further).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_unet.h"

namespace unet {
namespace {

typedef float thin_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThinPx = 128;          // pixels of S per tile
constexpr int kThinPA = kThinPx + 4;  // pitch of the S tile

struct ThinShape {
    int R;          // rows of S per tile (128 / Ws)
    int rowp;       // patch row pitch: Wl + 4
    int planep;     // patch plane pitch: (2R + 2) * rowp rounded up to 16 mod 64
    int tiles;      // tiles per image: Hs / R
};
inline bool thin_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws && g.Cl >= 1 && g.Cl <= 4 &&
           g.Ws % 32 == 0 && g.Ws <= kThinPx && g.Hs % (kThinPx / g.Ws) == 0 && g.Cs >= 8 && g.Cs <= 64 &&
           (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31) && (long long)g.B * g.Cl * g.Hl * g.Wl < (1ll << 31);
}
__host__ __device__ inline ThinShape thin_shape(const Geom& g) {
    ThinShape s;
    s.R = kThinPx / g.Ws;
    s.rowp = g.Wl + 4;
    const int raw = (2 * s.R + 2) * s.rowp;
    s.planep = raw + ((16 - raw % 64) + 64) % 64;
    s.tiles = g.Hs / s.R;
    return s;
}
inline size_t thin_lds_bytes(const Geom& g, int rbn) {
    const ThinShape s = thin_shape(g);
    return (size_t)(32 * rbn * kThinPA + g.Cl * s.planep + 4) * sizeof(float);
}

// grid: B * ceil(tiles / tpw) workgroups, block 256, dynamic LDS thin_lds_bytes.  RBN = row blocks of 32 channels (Cs <= 32 * RBN)
template <int RBN>
__global__ void __launch_bounds__(256) k_thin_wgrad(Geom g, const float* __restrict__ S, const float* __restrict__ L,
                                                    double* __restrict__ acc, float* __restrict__ part, int tpw) {
    extern __shared__ float4 thin_lds4[];
    float* Ss = reinterpret_cast<float*>(thin_lds4);          // [32 * RBN][132]
    float* Lp = Ss + 32 * RBN * kThinPA;                      // [Cl][planep] + 4
    const ThinShape sh = thin_shape(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int groups = (sh.tiles + tpw - 1) / tpw;
    const int b = blockIdx.x / groups, tg = blockIdx.x - b * groups;
    const int t_begin = tg * tpw, t_end = min(sh.tiles, t_begin + tpw);
    const int W4 = g.Wl >> 2, prow = 2 * sh.R + 2;            // float4s per patch row, patch rows
    const int lunits = g.Cl * prow * W4;

    for (int i = tid; i < g.Cl * sh.planep + 4; i += 256) Lp[i] = 0.f;   // pads and borders stay zero

    // the patch float4s this thread moves: LDS offset, row inside the patch, map offset of (cl, row 0, its columns); -1: none
    int l_lds[4], l_row[4];
    long long l_map[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i;
        const int c4 = u % W4, rr = u / W4, lr = rr % prow, cl = rr / prow;
        l_row[i] = u < lunits ? lr : -1;
        l_lds[i] = cl * sh.planep + lr * sh.rowp + 4 + 4 * c4;
        l_map[i] = ((long long)b * g.Cl + cl) * g.Hl * g.Wl + 4 * c4;
    }
    float4 sreg[4 * RBN], lreg[4];
    auto fetch = [&](int ti) {
        const int y0 = ti * sh.R;
#pragma unroll
        for (int i = 0; i < 4 * RBN; i++) {
            const int u = tid + 256 * i, cs = u >> 5, c4 = u & 31;
            sreg[i] = cs < g.Cs ? *reinterpret_cast<const float4*>(S + (((long long)b * g.Cs + cs) * g.Hs + y0) * g.Ws + 4 * c4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int Y = 2 * y0 - 1 + l_row[i];
            lreg[i] = (l_row[i] >= 0 && Y >= 0 && Y < g.Hl) ? *reinterpret_cast<const float4*>(L + l_map[i] + (long long)Y * g.Wl)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4 * RBN; i++) {
            const int u = tid + 256 * i;
            *reinterpret_cast<float4*>(Ss + (u >> 5) * kThinPA + 4 * (u & 31)) = sreg[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (l_row[i] >= 0) *reinterpret_cast<float4*>(Lp + l_lds[i]) = lreg[i];
    };

    thin_f32x16 a_[RBN][2];
#pragma unroll
    for (int rb = 0; rb < RBN; rb++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) a_[rb][cb][r] = 0.f;

    // this wave's 32 pixels: tile row yb, columns xb .. xb + 31; lanes 32-63 take the pixel 16 further
    const int yb = (32 * wave) / g.Ws, xb = (32 * wave) % g.Ws;
    const float* arow = Ss + (lane & 31) * kThinPA + 32 * wave + 16 * h;
    const float* bcol[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        const int t = (lane & 31) + 32 * cb;
        const int cl = min(t >> 4, g.Cl - 1), ky = (t >> 2) & 3, kx = t & 3;
        bcol[cb] = Lp + cl * sh.planep + (2 * yb + ky) * sh.rowp + kx + 3 + 2 * (xb + 16 * h);
    }

    if (t_begin < t_end) fetch(t_begin);
    for (int ti = t_begin; ti < t_end; ti++) {
        __syncthreads();            // the previous tile has been read (first trip: the patch has been cleared)
        commit();
        __syncthreads();
        if (ti + 1 < t_end) fetch(ti + 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 a4[RBN];
#pragma unroll
            for (int rb = 0; rb < RBN; rb++) a4[rb] = *reinterpret_cast<const float4*>(arow + 32 * rb * kThinPA + 4 * q);
            float bv[2][4];
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int j = 0; j < 4; j++) bv[cb][j] = bcol[cb][8 * q + 2 * j];
#pragma unroll
            for (int rb = 0; rb < RBN; rb++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++) {
                    a_[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rb].x, bv[cb][0], a_[rb][cb], 0, 0, 0);
                    a_[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rb].y, bv[cb][1], a_[rb][cb], 0, 0, 0);
                    a_[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rb].z, bv[cb][2], a_[rb][cb], 0, 0, 0);
                    a_[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rb].w, bv[cb][3], a_[rb][cb], 0, 0, 0);
                }
        }
    }

    // the four waves' partial sums, added in wave order (the S tile's LDS is free now): red[block][register][lane]
    float* red = Ss;
    for (int w = 0; w < 4; w++) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int rb = 0; rb < RBN; rb++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float* p = red + ((rb * 2 + cb) * 16 + r) * 64 + lane;
                        *p = w == 0 ? a_[rb][cb][r] : *p + a_[rb][cb][r];
                    }
        }
    }
    __syncthreads();
    const int ncol = 16 * g.Cl;
    for (int idx = tid; idx < RBN * 2 * 1024; idx += 256) {
        const int l = idx & 63, r = (idx >> 6) & 15, blk = idx >> 10;
        const int rb = blk >> 1, cb = blk & 1;
        const int cs = 32 * rb + (r >> 2) * 8 + (l >> 5) * 4 + (r & 3), t = 32 * cb + (l & 31);
        const float v = red[idx];
        if (cs >= g.Cs || t >= ncol) continue;
        if (part) part[(size_t)blockIdx.x * g.Cs * ncol + (size_t)cs * ncol + t] = v;     // k_thin_fold adds the workgroups' tiles
        else if (v != 0.f) atomicAdd(&acc[(size_t)cs * ncol + t], (double)v);
    }
}

// acc[e] += sum over workgroups of part[wg * E + e], fp64, workgroups q, q + 16, ... per thread and the sixteen partial sums in
// index order (no atomics: the result does not depend on timing).  grid ceil(E / 16), block 256
__global__ void __launch_bounds__(256) k_thin_fold(const float* __restrict__ part, int nwg, int E, double* __restrict__ acc) {
    __shared__ double red[16][17];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    double s = 0.0;
    if (e < E) {
        int z = q;
        for (; z + 48 < nwg; z += 64) {     // four loads in flight
            const float p0 = part[(size_t)z * E + e], p1 = part[(size_t)(z + 16) * E + e];
            const float p2 = part[(size_t)(z + 32) * E + e], p3 = part[(size_t)(z + 48) * E + e];
            s += (double)p0;
            s += (double)p1;
            s += (double)p2;
            s += (double)p3;
        }
        for (; z < nwg; z += 16) s += (double)part[(size_t)z * E + e];
    }
    red[q][el] = s;
    __syncthreads();
    if (q == 0 && e < E) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) t += red[i][el];
        acc[e] += t;
    }
}

inline int thin_wgrad_tpw(const Geom& g) {
    static const int tpw_env = getenv("CAE_THIN_TPW") ? atoi(getenv("CAE_THIN_TPW")) : 0;   // env: tuning runs only
    const int tiles = thin_shape(g).tiles;
    const int tpw = tpw_env > 0 ? tpw_env : 8;
    return tpw > tiles ? tiles : tpw;
}
inline int thin_wgrad_groups(const Geom& g) {
    const int tpw = thin_wgrad_tpw(g);
    return (thin_shape(g).tiles + tpw - 1) / tpw;
}
// bytes of partial tiles a launch writes (one Cs x 16 Cl tile per workgroup)
inline size_t thin_wgrad_part_bytes(const Geom& g) { return (size_t)g.B * thin_wgrad_groups(g) * g.Cs * 16 * g.Cl * sizeof(float); }

// part: room for thin_wgrad_part_bytes(g), or nullptr (fp64 atomics from every workgroup instead)
inline void thin_wgrad_launch(const Geom& g, const float* S, const float* L, double* acc, float* part, hipStream_t s) {
    const int tpw = thin_wgrad_tpw(g), nwg = g.B * thin_wgrad_groups(g);
    if (g.Cs <= 32) hipLaunchKernelGGL(k_thin_wgrad<1>, dim3(nwg), dim3(256), thin_lds_bytes(g, 1), s, g, S, L, acc, part, tpw);
    else hipLaunchKernelGGL(k_thin_wgrad<2>, dim3(nwg), dim3(256), thin_lds_bytes(g, 2), s, g, S, L, acc, part, tpw);
    if (part) {
        const int E = g.Cs * 16 * g.Cl;
        hipLaunchKernelGGL(k_thin_fold, dim3((E + 15) / 16), dim3(256), 0, s, part, nwg, E, acc);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
//   k_thin_down    S[b][cs][y][x] = bias[cs] + sum_{cl,ky,kx} w[cs][cl][ky][kx] * L[b][cl][2y-1+ky][2x-1+kx]
//                  (Conv2d forward of the 3 -> 32 layer; input gradient of the 64 -> 3 ConvTranspose2d)
// as D[cs][p] = sum_t W[cs][t] * patch(t, p): rows = channels, columns = pixels, K = 16 Cl taps.  The weights are the MFMA's A
// operand and never change: a lane keeps its 8 Cl values per row block in registers.  An MFMA step multiplies the tap pair
// (kx, kx + 1): lanes 0-31 (32 consecutive pixels, two floats apart in the patch) read the 32 even banks, lanes 32-63 the odd
// ones.  Only the patch is staged (no S tile): a tile is 128 pixels = 16-32 KB of output written as 128-byte runs.
// grid: B * ceil(tiles / tpw), block 256, dynamic LDS max(Cl * planep + 4, 32 RBN * 16 Cl) floats
template <int RBN, int CL>
__global__ void __launch_bounds__(256) k_thin_down(Geom g, const float* __restrict__ L, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ S, int tpw) {
    constexpr int KS = 8 * CL;      // MFMA steps (tap pairs)
    extern __shared__ float4 thin_lds4[];
    float* Lp = reinterpret_cast<float*>(thin_lds4);          // [CL][planep] + 4
    const ThinShape sh = thin_shape(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int groups = (sh.tiles + tpw - 1) / tpw;
    const int b = blockIdx.x / groups, tg = blockIdx.x - b * groups;
    const int t_begin = tg * tpw, t_end = min(sh.tiles, t_begin + tpw);
    const int W4 = g.Wl >> 2, prow = 2 * sh.R + 2;
    const int lunits = CL * prow * W4;

    // the weights pass through the (not yet used) patch area once, in 16-byte runs, into the lanes' registers:
    // wr[rb][s] = W[cs = 32 rb + lane % 32][t = 2 s + h];  one more step carries the bias (against a column of ones)
    const int wn = g.Cs * CL * 16;
    for (int i = tid; i < 32 * RBN * CL * 16; i += 256) Lp[i] = i < wn ? w[i] : 0.f;
    __syncthreads();
    float wr[RBN][KS + 1];
#pragma unroll
    for (int rb = 0; rb < RBN; rb++) {
        const int cs = 32 * rb + (lane & 31);
#pragma unroll
        for (int s = 0; s < KS; s++) wr[rb][s] = Lp[cs * (CL * 16) + 2 * s + h];
        wr[rb][KS] = (bias && h == 0 && cs < g.Cs) ? bias[cs] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < CL * sh.planep + 4; i += 256) Lp[i] = 0.f;      // pads and borders stay zero

    int l_lds[4], l_row[4];
    long long l_map[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i;
        const int c4 = u % W4, rr = u / W4, lr = rr % prow, cl = rr / prow;
        l_row[i] = u < lunits ? lr : -1;
        l_lds[i] = cl * sh.planep + lr * sh.rowp + 4 + 4 * c4;
        l_map[i] = ((long long)b * CL + cl) * g.Hl * g.Wl + 4 * c4;
    }
    float4 lreg[4];
    auto fetch = [&](int ti) {
        const int y0 = ti * sh.R;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int Y = 2 * y0 - 1 + l_row[i];
            lreg[i] = (l_row[i] >= 0 && Y >= 0 && Y < g.Hl) ? *reinterpret_cast<const float4*>(L + l_map[i] + (long long)Y * g.Wl)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (l_row[i] >= 0) *reinterpret_cast<float4*>(Lp + l_lds[i]) = lreg[i];
    };

    const int yb = (32 * wave) / g.Ws, xb = (32 * wave) % g.Ws;      // this wave's 32 pixels inside the tile
    const float* bcol = Lp + 2 * yb * sh.rowp + 3 + 2 * (xb + (lane & 31)) + h;
    // output address of accumulator register 0 (row 4 h of block 0); register r is (r >> 2) * 8 + (r & 3) rows further
    const long long plane = (long long)g.Hs * g.Ws;
    float* out0 = S + ((long long)b * g.Cs + 4 * h) * plane + (long long)yb * g.Ws + xb + (lane & 31);

    if (t_begin < t_end) fetch(t_begin);
    for (int ti = t_begin; ti < t_end; ti++) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ti + 1 < t_end) fetch(ti + 1);
        thin_f32x16 acc[RBN];
#pragma unroll
        for (int rb = 0; rb < RBN; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rb][r] = 0.f;
#pragma unroll
        for (int rb = 0; rb < RBN; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[rb][KS], 1.f, acc[rb], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int cl = s >> 3, ky = (s >> 1) & 3, kx = 2 * (s & 1);
            const float bv = bcol[cl * sh.planep + ky * sh.rowp + kx];
#pragma unroll
            for (int rb = 0; rb < RBN; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[rb][s], bv, acc[rb], 0, 0, 0);
        }
        float* out = out0 + (long long)ti * sh.R * g.Ws;
#pragma unroll
        for (int rb = 0; rb < RBN; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 32 * rb + (r >> 2) * 8 + (r & 3);       // + 4 h: in out0
                if (row + 4 * h < g.Cs) out[(long long)row * plane] = acc[rb][r];
            }
    }
}

inline int thin_down_tpw(const Geom& g) {
    static const int tpw_env = getenv("CAE_THIN_DOWN_TPW") ? atoi(getenv("CAE_THIN_DOWN_TPW")) : 0;   // env: tuning runs only
    const int tiles = thin_shape(g).tiles;
    const int tpw = tpw_env > 0 ? tpw_env : 8;
    return tpw > tiles ? tiles : tpw;
}
template <int RBN>
inline void thin_down_launch_cl(const Geom& g, const float* L, const float* w, const float* bias, float* S, hipStream_t s) {
    const ThinShape sh = thin_shape(g);
    const int tpw = thin_down_tpw(g), nwg = g.B * ((sh.tiles + tpw - 1) / tpw);
    // the patch; before the first tile the same area holds the weights once
    const size_t lds = (size_t)std::max(g.Cl * sh.planep + 4, 32 * RBN * g.Cl * 16) * sizeof(float);
    switch (g.Cl) {
        case 1: hipLaunchKernelGGL((k_thin_down<RBN, 1>), dim3(nwg), dim3(256), lds, s, g, L, w, bias, S, tpw); break;
        case 2: hipLaunchKernelGGL((k_thin_down<RBN, 2>), dim3(nwg), dim3(256), lds, s, g, L, w, bias, S, tpw); break;
        case 3: hipLaunchKernelGGL((k_thin_down<RBN, 3>), dim3(nwg), dim3(256), lds, s, g, L, w, bias, S, tpw); break;
        default: hipLaunchKernelGGL((k_thin_down<RBN, 4>), dim3(nwg), dim3(256), lds, s, g, L, w, bias, S, tpw); break;
    }
}
inline void thin_down_launch(const Geom& g, const float* L, const float* w, const float* bias, float* S, hipStream_t s) {
    if (g.Cs <= 32) thin_down_launch_cl<1>(g, L, w, bias, S, s);
    else thin_down_launch_cl<2>(g, L, w, bias, S, s);
}


// ---------------------------------------------------------------------------------------------------------------------------
//   k_thin_up      L[b][cl][2q+ky-1][2r+kx-1] += S[b][cs][q][r] * w[cs][cl][ky][kx]  (+ bias)      ConvTranspose2d forward,
//                  CL <= 4 output channels (the 64 -> 3 layer), maps 128 columns wide
// This one is not a GEMM (three useful rows of 32): it is 96 multiply-adds per input value on the vector pipes.  A wave spans a
// whole input row - lane l owns columns 2l and 2l + 1, the left / right neighbours come from the adjacent lanes by one-lane DPP
// shifts (the row's ends are the map's edges: zeros) - and walks a band of T input rows per input channel with the band's
// 2T x 4 x CL outputs in registers: one 8-byte load per lane, row and channel instead of the nine 4-byte gathers per position
// of k_up_thin, the next channel's rows in flight.  The four output columns of a lane are two float pairs so that the
// multiply-adds are packed (v_pk_fma_f32): pair0 += (b, b) * (w1, w2) + (a, c) * (w3, w0), pair1 += (c, c) * (w1, w2) + (b, d) *
// (w3, w0) for the inputs a b c d = columns 2l-1 .. 2l+2; wq holds the taps in that order (k_pack_up_weights, thin entry).
// grid B * Hs / T waves (4 per workgroup), block 256
typedef float thin_f2 __attribute__((ext_vector_type(2)));

// (wq[((cs * CL + cl) * 4 + ky) * 4 + {0,1,2,3}] = w[cs][cl][ky][{1,2,3,0}]: written by k_pack_up_weights, kernels_unet_mfma.h)

__device__ __forceinline__ float thin_from_right(float v) {      // lane i <- lane i + 1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ float thin_from_left(float v) {       // lane i <- lane i - 1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

template <int CL, int T>
__global__ void __launch_bounds__(256) k_thin_up(Geom g, const float* __restrict__ S, const float* __restrict__ wq,
                                                 const float* __restrict__ bias, float* __restrict__ L) {
    const int lane = threadIdx.x & 63;
    const int bands = g.Hs / T;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));     // (image, band): wave-uniform
    if (wid >= g.B * bands) return;
    const int b = wid / bands, q0 = (wid - b * bands) * T;
    thin_f2 acc[2 * T][CL][2];
#pragma unroll
    for (int o = 0; o < 2 * T; o++)
#pragma unroll
        for (int cl = 0; cl < CL; cl++) {
            const float bv = bias ? bias[cl] : 0.f;
            acc[o][cl][0] = thin_f2{bv, bv}, acc[o][cl][1] = thin_f2{bv, bv};
        }
    const long long plane = (long long)g.Hs * g.Ws;
    const float* sp = S + (long long)b * g.Cs * plane + 2 * lane;
    thin_f2 cur[T + 2], nxt[T + 2], nx2[T + 2];       // the rows of channel cs, cs + 1, cs + 2
    auto load = [&](thin_f2 (&r)[T + 2], int cs) {
#pragma unroll
        for (int d = 0; d < T + 2; d++) {
            const int q = q0 - 1 + d;
            r[d] = (cs < g.Cs && q >= 0 && q < g.Hs) ? *reinterpret_cast<const thin_f2*>(sp + cs * plane + (long long)q * g.Ws) : thin_f2{0.f, 0.f};
        }
    };
    load(cur, 0);
    load(nxt, 1);
    for (int cs = 0; cs < g.Cs; cs++) {
        load(nx2, cs + 2);
        const float4* wp = reinterpret_cast<const float4*>(wq) + (size_t)cs * CL * 4;      // wave-uniform: scalar loads
#pragma unroll
        for (int d = 0; d < T + 2; d++) {
            const float bb = cur[d].x, cc = cur[d].y;
            const float aa = thin_from_left(cc), dd = thin_from_right(bb);
            const thin_f2 vb{bb, bb}, vc{cc, cc}, vac{aa, cc}, vbd{bb, dd};
#pragma unroll
            for (int ky = 0; ky < 4; ky++) {
                const int o = 2 * (d - 1) + ky - 1;          // output row inside the band: input row q0 - 1 + d, tap ky
                if (o < 0 || o >= 2 * T) continue;
#pragma unroll
                for (int cl = 0; cl < CL; cl++) {
                    const float4 wv = wp[cl * 4 + ky];
                    const thin_f2 w12{wv.x, wv.y}, w30{wv.z, wv.w};
                    acc[o][cl][0] = __builtin_elementwise_fma(vb, w12, acc[o][cl][0]);
                    acc[o][cl][0] = __builtin_elementwise_fma(vac, w30, acc[o][cl][0]);
                    acc[o][cl][1] = __builtin_elementwise_fma(vc, w12, acc[o][cl][1]);
                    acc[o][cl][1] = __builtin_elementwise_fma(vbd, w30, acc[o][cl][1]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < T + 2; d++) cur[d] = nxt[d], nxt[d] = nx2[d];
    }
#pragma unroll
    for (int o = 0; o < 2 * T; o++)
#pragma unroll
        for (int cl = 0; cl < CL; cl++)
            *reinterpret_cast<float4*>(L + (((long long)b * CL + cl) * g.Hl + 2 * q0 + o) * g.Wl + 4 * lane) =
                make_float4(acc[o][cl][0].x, acc[o][cl][0].y, acc[o][cl][1].x, acc[o][cl][1].y);
}

inline bool thin_up_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws && g.Cl >= 1 && g.Cl <= 4 &&
           g.Ws == 128 && g.Hs % 2 == 0 && (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31);
}
// wq: the layer's weights in k_thin_up's order (k_pack_up_weights, thin entry)
inline void thin_up_launch(const Geom& g, const float* S, const float* wq, const float* bias, float* L, hipStream_t s) {
    static const int t_env = getenv("CAE_THIN_UP_T") ? atoi(getenv("CAE_THIN_UP_T")) : 2;   // env: tuning runs only
    const int T = (t_env == 4 && g.Hs % 4 == 0) ? 4 : 2;
    const int waves = g.B * (g.Hs / T);
    const dim3 grid((waves + 3) / 4);
#define THIN_UP_CASE(CL_)                                                                                       \
    if (T == 4) hipLaunchKernelGGL((k_thin_up<CL_, 4>), grid, dim3(256), 0, s, g, S, wq, bias, L);              \
    else hipLaunchKernelGGL((k_thin_up<CL_, 2>), grid, dim3(256), 0, s, g, S, wq, bias, L)
    switch (g.Cl) {
        case 1: THIN_UP_CASE(1); break;
        case 2: THIN_UP_CASE(2); break;
        case 3: THIN_UP_CASE(3); break;
        default: THIN_UP_CASE(4); break;
    }
#undef THIN_UP_CASE
}

}  // namespace
}  // namespace unet
