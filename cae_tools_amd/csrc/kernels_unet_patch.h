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

namespace unet {
namespace {

typedef float patch_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPatchPx = 128;     // output pixels per workgroup
constexpr int kPatchCG = 4;       // input channels per K chunk
constexpr int kPatchWP = 64 + 4;  // pitch of the weight chunk [row][64]

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
    const int wc4 = tid & 15, wr0 = tid >> 4;                // weights: column quad wc4 of rows wr0, wr0 + 16, ...
    float4 wreg[2 * RBN], lreg[4];
    auto fetch = [&](int cg) {
#pragma unroll
        for (int i = 0; i < 2 * RBN; i++) {
            const int n = n0 + wr0 + 16 * i;
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
        for (int i = 0; i < 2 * RBN; i++)      // taps stored (kx0, kx2, kx1, kx3): a lane's two k values of a (channel, ky) are adjacent
            *reinterpret_cast<float4*>(Wt + (wr0 + 16 * i) * kPatchWP + 4 * wc4) = make_float4(wreg[i].x, wreg[i].z, wreg[i].y, wreg[i].w);
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
    const int rbn = g.Cs <= 64 ? 2 : 4;
    const int TN = 32 * rbn, rt = (g.Cs + TN - 1) / TN;
    const long long tiles = (long long)g.B * sh.tiles * rt;
    const int groups = g.Cl / kPatchCG;
    int nsplit = 1;     // few tiles and a long K: slice it so that every CU has work
    while (tiles * nsplit < 384 && groups / (nsplit * 2) >= 4 && nsplit < 8) nsplit *= 2;
    const int gper = (groups + nsplit - 1) / nsplit;
    if (nsplit > 1) (void)hipMemsetAsync(S, 0, (size_t)g.B * g.Cs * g.Hs * g.Ws * sizeof(float), s);
    const dim3 grid((unsigned)(g.B * sh.tiles), rt, nsplit);
    const size_t lds = patch_lds_bytes(g, rbn);
    if (rbn == 2) hipLaunchKernelGGL(k_pdown<2>, grid, dim3(256), lds, s, g, L, w, bias, S, gper, nsplit);
    else hipLaunchKernelGGL(k_pdown<4>, grid, dim3(256), lds, s, g, L, w, bias, S, gper, nsplit);
}

}  // namespace
}  // namespace unet
