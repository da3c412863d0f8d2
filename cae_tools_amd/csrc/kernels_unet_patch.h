// kernels_unet_patch.h — the WIDE 4x4 / stride 2 / padding 1 layers of the UNET path with the im2col operand read in place from
// an LDS patch (the idea of kernels_unet_thin.h carried to the layers whose K is too long for registers).
//
// The tile engine (kernels_unet_mfma.h) builds the pixel-side MFMA operand by gathering 16 values per output pixel and input
// channel with 4-byte loads (a 256-pixel tile x 4 channels = 16,384 load instructions' worth of lanes) and writes them to an
// im2col image in LDS; with those gathers compiled out its convolutions run 20 % faster (ablations).  Here a
// workgroup stages, per group of four input channels, the 2R + 2 map rows its 128 output pixels touch - ONCE, as 16-byte runs,
// 792 float4s instead of 16,384 gathers - into a zero-bordered patch, and an MFMA step's two k values are the taps (kx, kx + 1)
// of one (channel, ky): lanes 0-31 (32 consecutive output pixels, two floats apart in the patch) read the even banks, lanes
// 32-63 the odd ones.
//
//   k_pdown   S[b][cs][y][x] = bias[cs] + sum_{cl,ky,kx} w[cs][cl][ky][kx] * L[b][cl][2y-1+ky][2x-1+kx]
//             (Conv2d forward of the encoder; input gradient of the decoder's ConvTranspose2d layers)
//
// as D[cs][p] = sum_k W[cs][k] * patch(k, p): rows = output channels (32 RBN per workgroup), columns = 128 pixels (R = 128 / Ws map
// rows; a wave owns 32 of them and all the rows), K walked in chunks of 64 = four input channels.  The weight chunk is a
// [rows][64 + 4] LDS tile whose taps are stored (kx0, kx2, kx1, kx3), so that one ds_read_b64 per (row block, channel, ky) gives
// a lane its two k values; the patch is read one ds_read_b32 per step.  The next chunk's weights and patch rows are in registers while this one is
// multiplied.  Deep layers with few pixel tiles slice K over blockIdx.z onto a zeroed output (fp32 atomics), as the tile engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_unet.h"
#include "kernels_unet_mfma.h"

namespace unet {
namespace {

typedef float patch_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPatchPx = 128;     // output pixels per workgroup
#ifndef PATCH_CG
#define PATCH_CG 4
#endif
constexpr int kPatchCG = PATCH_CG;       // input channels per K chunk
constexpr int kPatchWP = 16 * kPatchCG + 4;  // pitch of the weight chunk [row][16 CG]

struct PatchShape {
    int R;        // map rows of S per tile
    int rowp;     // patch row pitch
    int planep;   // patch plane (channel) pitch
    int tiles;    // tiles per image
};
inline bool patch_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws &&
           (g.Ws == 16 || g.Ws == 32 || g.Ws == 64 || g.Ws == 128) && g.Hs % (kPatchPx / g.Ws) == 0 && g.Cl % kPatchCG == 0 && g.Cl >= 8 &&
           g.Cs >= 32 && (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31) && (long long)g.B * g.Cl * g.Hl * g.Wl < (1ll << 31);
}
__host__ __device__ inline PatchShape patch_shape(const Geom& g) {
    PatchShape s;
    s.R = kPatchPx / g.Ws;
    // a 32-pixel block of a 16-wide map spans two map rows: 2 * rowp = 32 mod 64 keeps its lanes on distinct banks
    s.rowp = g.Ws == 16 ? 48 : g.Wl + 4;
    s.planep = (2 * s.R + 2) * s.rowp;
    s.tiles = g.Hs / s.R;
    return s;
}
inline size_t patch_lds_bytes(const Geom& g, int rbn) {
    return (size_t)(32 * rbn * kPatchWP + kPatchCG * patch_shape(g).planep + 4) * sizeof(float);
}

// grid (B * tiles, ceil(Cs / (32 RBN)), K slices), block 256, dynamic LDS patch_lds_bytes.  gper: channel groups per K slice
template <int RBN>
__global__ void __launch_bounds__(256) k_pdown(Geom g, const float* __restrict__ L, const float* __restrict__ w,
                                               const float* __restrict__ bias, float* __restrict__ S, int gper, int nsplit) {
    constexpr int TN = 32 * RBN;
    extern __shared__ float4 patch_lds4[];
    float* Wt = reinterpret_cast<float*>(patch_lds4);        // [TN][68]
    float* Lp = Wt + TN * kPatchWP;                          // [4][planep] + 4
    const PatchShape sh = patch_shape(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int b = blockIdx.x / sh.tiles, y0 = (blockIdx.x - b * sh.tiles) * sh.R;
    const int n0 = blockIdx.y * TN;
    const int groups = g.Cl / kPatchCG;
    const int g_begin = blockIdx.z * gper, g_end = min(groups, g_begin + gper);
    const int W4 = g.Wl >> 2, prow = 2 * sh.R + 2;
    const int lunits = kPatchCG * prow * W4;                 // <= 1024 float4s

    for (int i = tid; i < kPatchCG * sh.planep + 4; i += 256) Lp[i] = 0.f;     // pads and borders stay zero

    // the patch float4s this thread moves: LDS offset, map offset of (channel-in-group, patch row, its columns); -1: none / outside
    int l_lds[4];
    long long l_map[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i;
        const int c4 = u % W4, rr = u / W4, lr = rr % prow, cl = rr / prow;
        const int Y = 2 * y0 - 1 + lr;
        const bool ok = u < lunits && Y >= 0 && Y < g.Hl;
        l_lds[i] = u < lunits ? cl * sh.planep + lr * sh.rowp + 4 + 4 * c4 : -1;
        l_map[i] = ok ? (((long long)b * g.Cl + cl) * g.Hl + Y) * g.Wl + 4 * c4 : -1;
    }
    const long long cplane = (long long)g.Hl * g.Wl;
    constexpr int QW = 4 * kPatchCG, RPP = 256 / QW, NWQ = 32 * RBN / RPP;   // float4s per weight row, rows per pass, passes
    const int wc4 = tid % QW, wr0 = tid / QW;                // weights: column quad wc4 of rows wr0, wr0 + RPP, ...
    float4 wreg[NWQ], lreg[4];
    auto fetch = [&](int cg) {
#pragma unroll
        for (int i = 0; i < NWQ; i++) {
            const int n = n0 + wr0 + RPP * i;
            wreg[i] = n < g.Cs ? *reinterpret_cast<const float4*>(w + ((size_t)n * g.Cl + cg * kPatchCG) * 16 + 4 * wc4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            lreg[i] = l_map[i] >= 0 ? *reinterpret_cast<const float4*>(L + l_map[i] + (long long)cg * kPatchCG * cplane)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NWQ; i++)      // taps stored (kx0, kx2, kx1, kx3): a lane's two k values of a (channel, ky) are adjacent
            *reinterpret_cast<float4*>(Wt + (wr0 + RPP * i) * kPatchWP + 4 * wc4) = make_float4(wreg[i].x, wreg[i].z, wreg[i].y, wreg[i].w);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (l_lds[i] >= 0) *reinterpret_cast<float4*>(Lp + l_lds[i]) = lreg[i];
    };

    patch_f32x16 acc[RBN];
#pragma unroll
    for (int rb = 0; rb < RBN; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[rb][r] = 0.f;

    // this lane's output pixel inside the tile, and where its taps start in the patch
    const int p = 32 * wave + (lane & 31), yl = p / g.Ws, xl = p - yl * g.Ws;
    const float* bcol = Lp + 2 * yl * sh.rowp + 3 + 2 * xl + h;
    const float* arow = Wt + (lane & 31) * kPatchWP + 2 * h;

    if (g_begin < g_end) fetch(g_begin);
    for (int cg = g_begin; cg < g_end; cg++) {
        __syncthreads();            // the previous chunk has been read (first trip: the patch has been cleared)
        commit();
        __syncthreads();
        if (cg + 1 < g_end) fetch(cg + 1);
#pragma unroll
        for (int c = 0; c < kPatchCG; c++)
#pragma unroll
            for (int ky = 0; ky < 4; ky++) {
                float2 a2[RBN];
#pragma unroll
                for (int rb = 0; rb < RBN; rb++) a2[rb] = *reinterpret_cast<const float2*>(arow + 32 * rb * kPatchWP + c * 16 + ky * 4);
                const float b0 = bcol[c * sh.planep + ky * sh.rowp], b1 = bcol[c * sh.planep + ky * sh.rowp + 2];
#pragma unroll
                for (int rb = 0; rb < RBN; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[rb].x, b0, acc[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RBN; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[rb].y, b1, acc[rb], 0, 0, 0);
            }
    }
    float* out = S + (((long long)b * g.Cs + n0 + 4 * h) * g.Hs + y0 + yl) * g.Ws + xl;
    const long long plane = (long long)g.Hs * g.Ws;
#pragma unroll
    for (int rb = 0; rb < RBN; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = 32 * rb + (r >> 2) * 8 + (r & 3);       // + 4 h: in `out`
            const int n = n0 + row + 4 * h;
            if (n >= g.Cs) continue;
            float v = acc[rb][r];
            if (bias && blockIdx.z == 0) v += bias[n];
            if (nsplit > 1) atomicAdd(out + (long long)row * plane, v);
            else out[(long long)row * plane] = v;
        }
}

inline void pdown_launch(const Geom& g, const float* L, const float* w, const float* bias, float* S, hipStream_t s) {
    const PatchShape sh = patch_shape(g);
    // 64-row tiles where there are few pixel tiles (the deep layers): twice the workgroups without slicing K twice as fine -
    // half the atomics, or none (101 -> 77 us and 102 -> 84 us for the encoder's 32 x 32 and 16 x 16 layers at batch 32)
    const int rbn = (g.Cs <= 64 || (long long)g.B * sh.tiles <= 256) ? 2 : 4;
    const int TN = 32 * rbn, rt = (g.Cs + TN - 1) / TN;
    const long long tiles = (long long)g.B * sh.tiles * rt;
    const int groups = g.Cl / kPatchCG;
    int nsplit = 1;     // few tiles and a long K: slice it so that every CU has work
    while (tiles * nsplit < 384 && groups / (nsplit * 2) >= 16 / kPatchCG && nsplit < 8) nsplit *= 2;
    const int gper = (groups + nsplit - 1) / nsplit;
    if (nsplit > 1) (void)hipMemsetAsync(S, 0, (size_t)g.B * g.Cs * g.Hs * g.Ws * sizeof(float), s);
    const dim3 grid((unsigned)(g.B * sh.tiles), rt, nsplit);
    const size_t lds = patch_lds_bytes(g, rbn);
    if (rbn == 2) hipLaunchKernelGGL(k_pdown<2>, grid, dim3(256), lds, s, g, L, w, bias, S, gper, nsplit);
    else hipLaunchKernelGGL(k_pdown<4>, grid, dim3(256), lds, s, g, L, w, bias, S, gper, nsplit);
}


// ---------------------------------------------------------------------------------------------------------------------------
//   k_pup     L[b][cl][2q+py][2r+px] = bias[cl] + sum_{cs,j,i} S[b][cs][q+py-j][r+px-i] * w[cs][cl][1-py+2j][1-px+2i]
//             (ConvTranspose2d forward of the decoder; input gradient of the encoder's Conv2d layers)
// per output-row parity py (blockIdx.z) and for BOTH column parities at once: D_px[cl][p] = sum_k Wz[cl][k] * patch(k, p), k = (cs,
// j, i), with the weights repacked per parity as the tile engine has them (k_pack_up_weights: wp[z][cl][cs*4 + j*2 + i]).  The
// patch is the R + 2 rows of S around the tile's 128 input pixels, unit stride: an MFMA step's two k values are the taps j = 0
// and j = 1 of one (channel, i), one patch row apart (row pitch = 32 mod 64: lanes 0-31 on 32 consecutive banks, lanes 32-63 on
// the other 32); the three column shifts r-1, r, r+1 serve both column parities, and a lane's two results (px = 0, 1) are
// adjacent in memory: 8-byte stores, a wave's run is 512 contiguous bytes.
constexpr int kPupCG = 8;              // input channels per K chunk (K = 32)
constexpr int kPupWP = 4 * kPupCG + 4; // pitch of a weight chunk [row][32]

inline bool pup_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws &&
           (g.Ws == 16 || g.Ws == 32 || g.Ws == 64 || g.Ws == 128) && g.Hs % (kPatchPx / g.Ws) == 0 && g.Cs % kPupCG == 0 && g.Cs >= 16 &&
           g.Cl >= 32 && (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31) && (long long)g.B * g.Cl * g.Hl * g.Wl < (1ll << 31);
}
__host__ __device__ inline PatchShape pup_shape(const Geom& g) {
    PatchShape s;
    s.R = kPatchPx / g.Ws;
    s.rowp = g.Ws == 16 ? 48 : g.Ws == 128 ? 160 : 96;      // >= Ws + 5, = 32 mod 64 (48 for the two-row blocks of a 16-wide map)
    s.planep = (s.R + 2) * s.rowp;
    s.tiles = g.Hs / s.R;
    return s;
}
inline size_t pup_lds_bytes(const Geom& g, int rbn) {
    return (size_t)(2 * 32 * rbn * kPupWP + kPupCG * pup_shape(g).planep) * sizeof(float);
}

// grid (B * tiles, ceil(Cl / (32 RBN)), 2 = py), block 256, dynamic LDS pup_lds_bytes
template <int RBN>
__global__ void __launch_bounds__(256) k_pup(Geom g, const float* __restrict__ S, const float* __restrict__ wp,
                                             const float* __restrict__ bias, float* __restrict__ L) {
    constexpr int TN = 32 * RBN;
    extern __shared__ float4 patch_lds4[];
    float* Wt = reinterpret_cast<float*>(patch_lds4);        // [2 = px][TN][36]
    float* Sp = Wt + 2 * TN * kPupWP;                        // [8][planep]
    const PatchShape sh = pup_shape(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    // the row parities and the row tiles of one pixel tile read the same patch rows: dealt to the same XCD (linear id mod 8) one
    // right after the other, so that the later ones find them in that L2
    int bx = blockIdx.x, by = blockIdx.y, py = blockIdx.z;
    if (RBN == 1 && (gridDim.x & 7) == 0) {      // (measured: -8 % for the 32-row tiles, +9 % for the 64-row ones, which keep launch order)
        const int i = bx + gridDim.x * (by + gridDim.y * py), j = i >> 3, nc = 2 * gridDim.y, c = j % nc;
        bx = (j / nc) * 8 + (i & 7);
        py = c & 1;
        by = c >> 1;
    }
    const int b = bx / sh.tiles, q0 = (bx - b * sh.tiles) * sh.R;
    const int n0 = by * TN;
    const int W4 = g.Ws >> 2, prow = sh.R + 2;
    const int lunits = kPupCG * prow * W4;                   // <= 768 float4s

    for (int i = tid; i < kPupCG * sh.planep; i += 256) Sp[i] = 0.f;           // pads and borders stay zero

    int l_lds[3];
    long long l_map[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int u = tid + 256 * i;
        const int c4 = u % W4, rr = u / W4, lr = rr % prow, c = rr / prow;
        const int q = q0 - 1 + lr;
        const bool ok = u < lunits && q >= 0 && q < g.Hs;
        l_lds[i] = u < lunits ? c * sh.planep + lr * sh.rowp + 4 + 4 * c4 : -1;
        l_map[i] = ok ? (((long long)b * g.Cs + c) * g.Hs + q) * g.Ws + 4 * c4 : -1;
    }
    const long long cplane = (long long)g.Hs * g.Ws;
    const int kw = g.Cs * 4;                                  // k length of a packed weight row
    // weights: 2 (px) x TN rows x 8 float4s per chunk
    const int wc4 = tid & 7, wr0 = tid >> 3;                 // column quad wc4 of rows wr0, wr0 + 32, ... of px 0, then of px 1
    float4 wreg[2 * RBN], lreg[3];
    auto fetch = [&](int cg) {
#pragma unroll
        for (int i = 0; i < 2 * RBN; i++) {
            const int px = i / RBN, n = n0 + wr0 + 32 * (i % RBN);
            wreg[i] = n < g.Cl ? *reinterpret_cast<const float4*>(wp + ((size_t)(2 * py + px) * g.Cl + n) * kw + cg * (4 * kPupCG) + 4 * wc4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
            lreg[i] = l_map[i] >= 0 ? *reinterpret_cast<const float4*>(S + l_map[i] + (long long)cg * kPupCG * cplane)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * RBN; i++)
            *reinterpret_cast<float4*>(Wt + ((i / RBN) * TN + wr0 + 32 * (i % RBN)) * kPupWP + 4 * wc4) = wreg[i];
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (l_lds[i] >= 0) *reinterpret_cast<float4*>(Sp + l_lds[i]) = lreg[i];
    };

    patch_f32x16 acc[2][RBN];
#pragma unroll
    for (int px = 0; px < 2; px++)
#pragma unroll
        for (int rb = 0; rb < RBN; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[px][rb][r] = 0.f;

    // this lane's input pixel inside the tile; its taps: patch row ql + py + 1 - j (j = h), columns r - 1, r, r + 1
    const int p = 32 * wave + (lane & 31), ql = p / g.Ws, r = p - ql * g.Ws;
    const float* bcol = Sp + (ql + py + 1 - h) * sh.rowp + 4 + r;
    const float* arow = Wt + (lane & 31) * kPupWP + 2 * h;

    const int chunks = g.Cs / kPupCG;
    fetch(0);
    for (int cg = 0; cg < chunks; cg++) {
        __syncthreads();
        commit();
        __syncthreads();
        if (cg + 1 < chunks) fetch(cg + 1);
#pragma unroll
        for (int c = 0; c < kPupCG; c++) {
            float2 a2[2][RBN];
#pragma unroll
            for (int px = 0; px < 2; px++)
#pragma unroll
                for (int rb = 0; rb < RBN; rb++)
                    a2[px][rb] = *reinterpret_cast<const float2*>(arow + (px * TN + 32 * rb) * kPupWP + c * 4);
            const float bm = bcol[c * sh.planep - 1], b0 = bcol[c * sh.planep], bp = bcol[c * sh.planep + 1];
#pragma unroll
            for (int rb = 0; rb < RBN; rb++) {
                acc[0][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][rb].x, b0, acc[0][rb], 0, 0, 0);      // px 0, i 0: column r
                acc[1][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][rb].x, bp, acc[1][rb], 0, 0, 0);      // px 1, i 0: column r + 1
            }
#pragma unroll
            for (int rb = 0; rb < RBN; rb++) {
                acc[0][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][rb].y, bm, acc[0][rb], 0, 0, 0);      // px 0, i 1: column r - 1
                acc[1][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][rb].y, b0, acc[1][rb], 0, 0, 0);      // px 1, i 1: column r
            }
        }
    }
    const long long plane = (long long)g.Hl * g.Wl;
    float* out = L + ((long long)b * g.Cl + n0 + 4 * h) * plane + (long long)(2 * (q0 + ql) + py) * g.Wl + 2 * r;
#pragma unroll
    for (int rb = 0; rb < RBN; rb++)
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
            const int row = 32 * rb + (rr >> 2) * 8 + (rr & 3);       // + 4 h: in `out`
            const int n = n0 + row + 4 * h;
            if (n >= g.Cl) continue;
            const float bv = bias ? bias[n] : 0.f;
            *reinterpret_cast<float2*>(out + (long long)row * plane) = make_float2(acc[0][rb][rr] + bv, acc[1][rb][rr] + bv);
        }
}

// wp: the layer's weights repacked per parity by k_pack_up_weights
inline void pup_launch(const Geom& g, const float* S, const float* wp, const float* bias, float* L, hipStream_t s) {
    const PatchShape sh = pup_shape(g);
    // 32-row tiles where there are few pixel tiles (the 16 x 16 layers at batch 32: 256 -> 512 workgroups, 82 -> 77 us)
    const int rbn = (g.Cl <= 32 || (long long)g.B * sh.tiles <= 64) ? 1 : 2;
    const int TN = 32 * rbn;
    const dim3 grid((unsigned)(g.B * sh.tiles), (g.Cl + TN - 1) / TN, 2);
    const size_t lds = pup_lds_bytes(g, rbn);
    if (rbn == 1) hipLaunchKernelGGL(k_pup<1>, grid, dim3(256), lds, s, g, S, wp, bias, L);
    else hipLaunchKernelGGL(k_pup<2>, grid, dim3(256), lds, s, g, S, wp, bias, L);
}


// ---------------------------------------------------------------------------------------------------------------------------
//   k_pwgrad  dW[cs][cl][ky][kx] = sum_{b,y,x} S[b][cs][y][x] * L[b][cl][2y-1+ky][2x-1+kx]      (weight gradient of either layer kind)
// as D[cs][t] = sum_p S[cs][p] * patch(t, p), t = (cl, ky, kx): rows = 32 WR channels of S, columns = 128 taps = eight channels of L,
// K = pixels in tiles of 64 (64 / Ws map rows).  The S tile is a [rows][64 + 4] LDS image (one ds_read_b128 = four k steps of a
// row); the patch holds the eight channels' 2R + 2 rows, zero-bordered as in k_pdown, and is read in place: an MFMA step
// multiplies the pixel pair {p, p + 16}; lanes 0-31 are 32 taps of pixel p - kx, then ky (row pitch = 4 mod 64), then the
// channel (plane pitch = 16 mod 64): 32 consecutive banks - lanes 32-63 the same taps 16 pixels on (32 floats: the other 32
// banks; 16-wide maps put that pixel two patch rows down and pay a two-way conflict).  A workgroup walks `tpw` pixel tiles with
// the next tile's loads in flight and leaves its 32 WR x 128 tile at part[(slice * Cs + n) * 16 Cl + t] for k_wgrad_fold.
// WR x WC = 4 waves: a wave owns 32 rows x 128 / WC columns.   grid (Cl / 8, ceil(Cs / (32 WR)), slices), block 256
constexpr int kPwPx = 64;            // pixels per K tile
constexpr int kPwSP = kPwPx + 4;     // pitch of the S tile

inline bool pwgrad_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws &&
           (g.Ws == 16 || g.Ws == 32 || g.Ws == 64) && g.Hs % (kPwPx / g.Ws) == 0 && g.Cl % 8 == 0 && g.Cs >= 64 &&
           (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31) && (long long)g.B * g.Cl * g.Hl * g.Wl < (1ll << 31);
}
__host__ __device__ inline PatchShape pwgrad_shape(const Geom& g) {
    PatchShape s;
    s.R = kPwPx / g.Ws;
    s.rowp = g.Wl + 4;
    const int raw = (2 * s.R + 2) * s.rowp;
    s.planep = raw + ((16 - raw % 64) + 64) % 64;
    s.tiles = g.Hs / s.R;            // per image
    return s;
}
inline size_t pwgrad_lds_bytes(const Geom& g, int wr) {
    return (size_t)(32 * wr * kPwSP + 8 * pwgrad_shape(g).planep + 4) * sizeof(float);
}

template <int WR, int WC>
__global__ void __launch_bounds__(256) k_pwgrad(Geom g, const float* __restrict__ S, const float* __restrict__ L,
                                                float* __restrict__ part, int tpw) {
    constexpr int TN = 32 * WR, NCB = 4 / WC;                // rows per workgroup, column blocks per wave
    extern __shared__ float4 patch_lds4[];
    float* Ss = reinterpret_cast<float*>(patch_lds4);        // [TN][68]
    float* Lp = Ss + TN * kPwSP;                             // [8][planep] + 4
    const PatchShape sh = pwgrad_shape(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wr = wave / WC, wc = wave - wr * WC;
    // workgroups are dealt to the 8 XCDs by linear id mod 8: the output tiles of ONE K slice (they read the same pixel tiles of
    // both maps) go to one XCD, so that its L2 fetches them once
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if ((gridDim.z & 7) == 0) {
        const int T = gridDim.x * gridDim.y, i = bx + gridDim.x * (by + gridDim.y * bz), j = i >> 3, t = j % T;
        bz = (j / T) * 8 + (i & 7);
        bx = t % gridDim.x;
        by = t / gridDim.x;
    }
    const int cl0 = bx * 8, n0 = by * TN;
    const int total = g.B * sh.tiles;                        // pixel tiles in the batch
    const int t_begin = bz * tpw, t_end = min(total, t_begin + tpw);
    const int W4 = g.Wl >> 2, prow = 2 * sh.R + 2;
    const int lunits = 8 * prow * W4;                        // <= 1024 float4s

    for (int i = tid; i < 8 * sh.planep + 4; i += 256) Lp[i] = 0.f;            // pads and borders stay zero

    int l_lds[4], l_row[4], l_off[4];                        // patch float4s of this thread: LDS offset, patch row, offset in the map
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int u = tid + 256 * i;
        const int c4 = u % W4, rr = u / W4, lr = rr % prow, c = rr / prow;
        l_row[i] = u < lunits ? lr : -1;
        l_lds[i] = c * sh.planep + lr * sh.rowp + 4 + 4 * c4;
        l_off[i] = (cl0 + c) * g.Hl * g.Wl + 4 * c4;         // + (b * Cl * Hl + Y) * Wl
    }
    const int sc4 = tid & 15, sr0 = tid >> 4;                // S tile: column quad sc4 of rows sr0, sr0 + 16, ...
    float4 sreg[2 * WR], lreg[4];
    auto fetch = [&](int ti) {
        const int b = ti / sh.tiles, y0 = (ti - b * sh.tiles) * sh.R;
#pragma unroll
        for (int i = 0; i < 2 * WR; i++) {
            const int n = n0 + sr0 + 16 * i;
            sreg[i] = n < g.Cs ? *reinterpret_cast<const float4*>(S + (((long long)b * g.Cs + n) * g.Hs + y0) * g.Ws + 4 * sc4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const long long lb = (long long)b * g.Cl * g.Hl * g.Wl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int Y = 2 * y0 - 1 + l_row[i];
            lreg[i] = (l_row[i] >= 0 && Y >= 0 && Y < g.Hl) ? *reinterpret_cast<const float4*>(L + lb + l_off[i] + (long long)Y * g.Wl)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 2 * WR; i++) *reinterpret_cast<float4*>(Ss + (sr0 + 16 * i) * kPwSP + 4 * sc4) = sreg[i];
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (l_row[i] >= 0) *reinterpret_cast<float4*>(Lp + l_lds[i]) = lreg[i];
    };

    patch_f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[cb][r] = 0.f;

    // A: this wave's 32 rows, pixel 16 h + ...; B: this lane's tap of column block cb, pixel offsets added per step
    const float* arow = Ss + (32 * wr + (lane & 31)) * kPwSP + 16 * h;
    const int yh = (16 * h) / g.Ws, xh = (16 * h) - yh * g.Ws;
    const float* bcol[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int t = lane & 31, c = 2 * (wc * NCB + cb) + (t >> 4), ky = (t >> 2) & 3, kx = t & 3;
        bcol[cb] = Lp + c * sh.planep + (ky + 2 * yh) * sh.rowp + kx + 3 + 2 * xh;
    }
    int boff[2];                                             // the two 32-pixel blocks of a tile: first pixel's (row, column) in the patch
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const int yb = (32 * blk) / g.Ws, xb = 32 * blk - yb * g.Ws;
        boff[blk] = 2 * yb * sh.rowp + 2 * xb;
    }

    if (t_begin < t_end) fetch(t_begin);
    for (int ti = t_begin; ti < t_end; ti++) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ti + 1 < t_end) fetch(ti + 1);
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + 32 * blk + 4 * q);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bcol[cb][boff[blk] + 2 * (4 * q + j)], acc[cb], 0, 0, 0);
            }
    }
    const int ncol = g.Cl * 16;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const int col = cl0 * 16 + 32 * (wc * NCB + cb) + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = n0 + 32 * wr + (r >> 2) * 8 + 4 * h + (r & 3);
            if (n < g.Cs) part[((size_t)bz * g.Cs + n) * ncol + col] = acc[cb][r];
        }
    }
}

// K slices and bytes of partial tiles of a launch
inline int pwgrad_tpw(const Geom& g) {
    const int rows = g.Cs >= 128 ? 128 : 64;
    const long long tiles_out = (long long)(g.Cl / 8) * ((g.Cs + rows - 1) / rows);
    const int total = g.B * pwgrad_shape(g).tiles;
    static const int target = getenv("CAE_PWGRAD_WGS") ? atoi(getenv("CAE_PWGRAD_WGS")) : 512;   // env: tuning runs only
    long long want = (target + tiles_out - 1) / tiles_out;    // slices to aim at: two resident workgroups per CU
    if (want < 1) want = 1;
    int tpw = (int)((total + want - 1) / want);
    if (tpw < 2) tpw = 2;
    return tpw;
}
inline int pwgrad_slices(const Geom& g) {
    const int total = g.B * pwgrad_shape(g).tiles, tpw = pwgrad_tpw(g);
    return (total + tpw - 1) / tpw;
}
inline size_t pwgrad_part_bytes(const Geom& g) { return (size_t)pwgrad_slices(g) * g.Cs * g.Cl * 16 * sizeof(float); }

// part: room for pwgrad_part_bytes(g)
inline void pwgrad_launch(const Geom& g, const float* S, const float* L, double* acc, float* part, hipStream_t s) {
    const int tpw = pwgrad_tpw(g), slices = pwgrad_slices(g);
    if (g.Cs >= 128) {
        const dim3 grid(g.Cl / 8, (g.Cs + 127) / 128, slices);
        hipLaunchKernelGGL((k_pwgrad<4, 1>), grid, dim3(256), pwgrad_lds_bytes(g, 4), s, g, S, L, part, tpw);
    } else {
        const dim3 grid(g.Cl / 8, (g.Cs + 63) / 64, slices);
        hipLaunchKernelGGL((k_pwgrad<2, 2>), grid, dim3(256), pwgrad_lds_bytes(g, 2), s, g, S, L, part, tpw);
    }
    const long long E = (long long)g.Cs * g.Cl * 16;
    hipLaunchKernelGGL(k_wgrad_fold, dim3((unsigned)std::min<long long>((E + 255) / 256, 4096)), dim3(256), 0, s, part, slices, E, acc);
}

}  // namespace
}  // namespace unet
