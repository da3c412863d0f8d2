// kernels_unet_skinny.h — the UNET's two big Linear layers (unet.py:92-100,121-129: flat -> fc and fc -> flat; 65536 <-> 128 at
// BASELINE cfg3) in the backward pass, where a batch of <= 64 rows meets a 33.5 MB weight matrix: GEMMs that are one pass over
// the weights (or over an 8.4 M-element fp64 gradient) and nothing else.  The convolution tile engine (kernels_unet_mfma.h,
// OpGemm) runs them through 16-wide K chunks with a barrier per chunk: 29-58 us each, 0.6-1.1 TB/s.  Here:
//
//   k_skinny_kj   Y[b][j] = sum_k X[b][k] * W[k][j]   with W stored [K][J], j contiguous - the input gradients
//                 (fc0: K = 128 outputs, J = 65536 inputs; fc3: K = 65536, J = 128, K sliced over workgroups).
//                 A thread owns BT batch rows x 4 columns; per k one 16-byte load of W (a wave reads a contiguous 1 KB run of
//                 the row), BT values of X from an LDS tile transposed to [k][b] (broadcast reads), BT * 4 FMAs.
//   k_skinny_fold the K slices' partial tiles summed in slice order (no atomics: the result does not depend on timing)
//   k_skinny_outer dW[o][i] = sum_b G[b][o] * X[b][i] (fp64 accumulator, plain stores: one writer per element) and
//                 db[o] = sum_b G[b][o] - the weight gradients; 64 x 64 output tiles, a thread owns 4 x 4 of it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {
namespace {

constexpr int kSkKs = 128;   // k rows of W a workgroup walks (its X^T tile: 128 x 64 floats at most)

// grid (ceil(J / (4 * JG)), K slices), block 256 = bgs batch-row groups (a power of two >= ceil(B / BT)) x JG = 256 / bgs column
// groups of 4: for B = 32, BT = 8 that is 4 x 64 (256 columns per workgroup), BT = 4 gives 8 x 32 (128 columns).
// part: nullptr -> Y[b * J + j] (one slice), else part[(slice * B + b) * J + j].
template <int BT>
__global__ void __launch_bounds__(256) k_skinny_kj(const float* __restrict__ X, long long x_sb, const float* __restrict__ W, int B,
                                                   int K, int J, int bgs, float* __restrict__ Y, float* __restrict__ part) {
    __shared__ float xt[kSkKs][64 + 4];   // [k][b]; 16-byte aligned rows (the batch-row values of a k are read as float4s)
    const int BG = (B + BT - 1) / BT;             // batch groups in use (<= bgs)
    const int JG = 256 / bgs;                     // column groups
    const int tid = threadIdx.x;
    const int jg = tid % JG, bg = tid / JG;
    const int j0 = (blockIdx.x * JG + jg) * 4;
    const int k0 = blockIdx.y * kSkKs, kn = min(kSkKs, K - k0);
    // X tile, transposed: X[b][k0 + k] -> xt[k][b]; rows b >= B are zero
    for (int e = tid; e < kSkKs * 64; e += 256) {
        const int b = e / kSkKs, k = e - b * kSkKs;      // a wave reads 64 consecutive k of one row: coalesced
        xt[k][b] = (b < B && k < kn) ? X[(size_t)b * x_sb + k0 + k] : 0.f;
    }
    __syncthreads();
    if (bg >= BG) return;
    float acc[BT][4];
#pragma unroll
    for (int r = 0; r < BT; r++) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
    const bool full = j0 + 3 < J;
    const float* wp = W + (size_t)k0 * J + (full ? j0 : 0);
    auto step = [&](int k, float4 w) {
        float xr[BT];
#pragma unroll
        for (int q = 0; q < BT / 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(&xt[k][bg * BT + 4 * q]);
            xr[4 * q] = v.x, xr[4 * q + 1] = v.y, xr[4 * q + 2] = v.z, xr[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int r = 0; r < BT; r++) {
            const float x = xr[r];
            acc[r][0] = fmaf(x, w.x, acc[r][0]);
            acc[r][1] = fmaf(x, w.y, acc[r][1]);
            acc[r][2] = fmaf(x, w.z, acc[r][2]);
            acc[r][3] = fmaf(x, w.w, acc[r][3]);
        }
    };
    if (full && (J & 3) == 0) {
        // eight rows of W in flight while the previous eight are multiplied: a wave alone on its SIMD would otherwise
        // alternate between a memory round trip and 0.5 us of arithmetic (29 us for the 33.5 MB of the 128 x 65536 matrix)
        int k = 0;
        float4 wn[8];
#pragma unroll
        for (int u = 0; u < 8; u++) wn[u] = u < kn ? *reinterpret_cast<const float4*>(wp + (size_t)u * J) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (; k + 8 <= kn; k += 8) {
            float4 w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = wn[u];
#pragma unroll
            for (int u = 0; u < 8; u++)
                wn[u] = k + 8 + u < kn ? *reinterpret_cast<const float4*>(wp + (size_t)(k + 8 + u) * J) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; u++) step(k + u, w[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k + u < kn) step(k + u, wn[u]);
    } else {
        for (int k = 0; k < kn; k++) {
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* p = W + (size_t)(k0 + k) * J;
            if (j0 < J) w.x = p[j0];
            if (j0 + 1 < J) w.y = p[j0 + 1];
            if (j0 + 2 < J) w.z = p[j0 + 2];
            if (j0 + 3 < J) w.w = p[j0 + 3];
            step(k, w);
        }
    }
    float* dst = part ? part + (size_t)blockIdx.y * B * J : Y;
#pragma unroll
    for (int r = 0; r < BT; r++) {
        const int b = bg * BT + r;
        if (b >= B) continue;
        float* o = dst + (size_t)b * J + j0;
        if (full && (J & 3) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (j0 + c < J) o[c] = acc[r][c];
        }
    }
}

// Y[e] = sum over slices of part[slice * n + e], in a fixed order: thread (e, q) adds the slices q, q + 4, ... (64 consecutive e
// per wave: coalesced), the four partial sums meet in LDS.  The partials are cleared behind the read: they live in the engine's
// split-K scratch, which the tile engine's own split GEMMs expect to find zero.   grid ceil(n / 64), block 256
__global__ void __launch_bounds__(256) k_skinny_fold(float* __restrict__ part, int slices, long long n, float* __restrict__ Y) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (e < n) {
        int z = q;
        for (; z + 12 < slices; z += 16) {       // four loads in flight per thread
            float* p = part + (size_t)z * n + e;
            const float a = p[0], b = p[(size_t)4 * n], c = p[(size_t)8 * n], d = p[(size_t)12 * n];
            p[0] = 0.f, p[(size_t)4 * n] = 0.f, p[(size_t)8 * n] = 0.f, p[(size_t)12 * n] = 0.f;
            s += a;
            s += b;
            s += c;
            s += d;
        }
        for (; z < slices; z += 4) {
            s += part[(size_t)z * n + e];
            part[(size_t)z * n + e] = 0.f;
        }
    }
    red[q][el] = s;
    __syncthreads();
    if (q == 0 && e < n) Y[e] = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
}

// dW[o][i] = sum_b G[b][o] X[b][i],  db[o] = sum_b G[b][o]   (G: B x O, X: B x I, row-major; accW: O x I fp64, accB: O fp64)
// grid (ceil(I / 64), ceil(O / 64)), block 256 = 16 (o groups of 4) x 16 (i groups of 4)
__global__ void __launch_bounds__(256) k_skinny_outer(const float* __restrict__ G, const float* __restrict__ X, int B, int O, int I,
                                                      double* __restrict__ accW, double* __restrict__ accB) {
    // dynamic LDS: [B][68] floats twice (2 * B * 272 bytes: 17 KB at B = 32, so that eight workgroups share a CU - a workgroup
    // is a load, 0.1 us of arithmetic and a 32 KB store one after the other, and only its neighbours hide that)
    extern __shared__ float sk_lds[];
    float (*gs)[64 + 4] = reinterpret_cast<float (*)[64 + 4]>(sk_lds);            // [b][o]
    float (*xs)[64 + 4] = reinterpret_cast<float (*)[64 + 4]>(sk_lds + B * 68);   // [b][i]
    const int tid = threadIdx.x;
    const int o0 = blockIdx.y * 64, i0 = blockIdx.x * 64;
    for (int e = tid; e < B * 64; e += 256) {
        const int b = e >> 6, c = e & 63;
        gs[b][c] = o0 + c < O ? G[(size_t)b * O + o0 + c] : 0.f;
        xs[b][c] = i0 + c < I ? X[(size_t)b * I + i0 + c] : 0.f;
    }
    __syncthreads();
    const int ig = tid & 15, og = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; b++) {
        const float4 g = *reinterpret_cast<const float4*>(&gs[b][og * 4]);
        const float4 x = *reinterpret_cast<const float4*>(&xs[b][ig * 4]);
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            acc[r][0] = fmaf(gv[r], x.x, acc[r][0]);
            acc[r][1] = fmaf(gv[r], x.y, acc[r][1]);
            acc[r][2] = fmaf(gv[r], x.z, acc[r][2]);
            acc[r][3] = fmaf(gv[r], x.w, acc[r][3]);
            bsum[r] += gv[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int o = o0 + og * 4 + r;
        if (o >= O) continue;
        double* row = accW + (size_t)o * I + i0 + ig * 4;
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (i0 + ig * 4 + c < I) row[c] = (double)acc[r][c];
        if (accB && blockIdx.x == 0 && ig == 0) accB[o] = (double)bsum[r];
    }
}

}  // namespace
}  // namespace unet
