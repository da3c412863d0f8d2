// kernels_unet_lin.h — the UNET's two big Linear layers (unet.py:92-100,121-129: flat -> fc and fc -> flat; 65536 <-> 128 at
// BASELINE cfg3) at batch <= 64: six GEMMs per step that are ONE pass over a 33.5 MB weight matrix (or over its gradient)
// and a few hundred MFMAs per workgroup.  The convolution tile engine walks them in 16-wide K chunks with a barrier per
// chunk (36-41 us per forward layer, 0.9 TB/s); here a workgroup stages a 128 x 64 weight tile with 256/512-byte coalesced
// runs, multiplies it against the batch rows on the matrix pipes (v_mfma_f32_32x32x2_f32: the batch on the 32 result rows,
// 32 output columns per wave, so a result register is a 128-byte run of one output row) and is gone; the next tile of its
// K range is in registers while the current one is multiplied.
//
//   k_lin_nt     Y[b][n] = sum_k X[b][k] W[n][k]      W rows are k-contiguous   (forward: W[out][in])
//   k_lin_nn     Y[b][n] = sum_k X[b][k] W[k][n]      W rows are n-contiguous   (input gradient: W[out = k][in = n])
//   k_lin_outer  dW[n][k] = sum_b G[b][n] X[b][k], db[n] = sum_b G[b][n]   fp32 plain stores, one writer per element
//   k_lin_fold   the K slices' partial tiles summed in fp64, slice order fixed, + bias
//
// Operand images in LDS, and why their pitches: an MFMA step multiplies the k pair {8G + j, 8G + 4 + j} (lanes 0-31 the
// first, lanes 32-63 the second), so that a k-contiguous image gives a lane four steps' operands in one ds_read_b128
// (pitch 68 floats: 16 consecutive rows start 4 banks apart and tile the 64 banks); an n-contiguous image is read one
// ds_read_b32 per step, lanes 0-31 on 32 consecutive banks and lanes 32-63 four rows further on the other 32 (pitch 136:
// 4 * 136 = 32 mod 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unet {
namespace {

typedef float lin_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLinKT = 64;      // k per staged tile
constexpr int kLinNT = 128;     // output columns per workgroup (4 waves x 32)
constexpr int kLinPK = 68;      // pitch of a k-contiguous image (floats)
constexpr int kLinPN = 136;     // pitch of an n-contiguous image

struct LinArgs {
    const float* X;     // (B, K) rows, x_ld apart
    long long x_ld;
    const float* W;     // nt: W[n * w_ld + k]; nn: W[k * w_ld + n]
    long long w_ld;
    const float* bias;  // per n or nullptr (direct store only; the fold adds it otherwise)
    float* Y;           // part == nullptr: Y[b * y_ld + n]
    long long y_ld;
    float* part;        // K sliced over blockIdx.y: part[((long long)slice * B + b) * N + n]
    int B, N, K;
    int kiters;         // tiles of kLinKT per workgroup
};

__device__ __forceinline__ float4 lin_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 lin_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// accumulator register r of lane l is row (r / 4) * 8 + (l / 32) * 4 + r % 4, column l % 32 of the 32 x 32 result
template <int RB>
__device__ __forceinline__ void lin_store(const LinArgs& a, const lin_f32x16 (&acc)[RB], int n, int lane) {
    if (n >= a.N) return;
    const float bias = (!a.part && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int b = 32 * rb + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
            if (b >= a.B) continue;
            if (a.part) a.part[((long long)blockIdx.y * a.B + b) * a.N + n] = acc[rb][r];
            else a.Y[(long long)b * a.y_ld + n] = acc[rb][r] + bias;
        }
}

// grid (ceil(N / 128), K slices), block 256, dynamic LDS (32 * RB + 128) * 68 floats.   K % 4 == 0, rows 16-byte aligned
template <int RB>
__global__ void __launch_bounds__(256) k_lin_nt(LinArgs a) {
    extern __shared__ float4 lin_lds4[];
    float* Xs = reinterpret_cast<float*>(lin_lds4);          // [32 * RB][68]
    float* Ws = Xs + 32 * RB * kLinPK;                       // [128][68]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * kLinNT;
    const int kbeg = blockIdx.y * a.kiters * kLinKT;
    const int c4 = tid & 15, r0 = tid >> 4;                  // a thread moves column quad c4 of rows r0, r0 + 16, ...
    float4 wreg[8], xreg[2 * RB];
    auto fetch = [&](int k0) {
        const int k = k0 + 4 * c4;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int n = n0 + r0 + 16 * i;
            wreg[i] = (n < a.N && k < a.K) ? lin_ld4(a.W + (long long)n * a.w_ld + k) : lin_zero4();
        }
#pragma unroll
        for (int i = 0; i < 2 * RB; i++) {
            const int b = r0 + 16 * i;
            xreg[i] = (b < a.B && k < a.K) ? lin_ld4(a.X + (long long)b * a.x_ld + k) : lin_zero4();
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<float4*>(Ws + (r0 + 16 * i) * kLinPK + 4 * c4) = wreg[i];
#pragma unroll
        for (int i = 0; i < 2 * RB; i++) *reinterpret_cast<float4*>(Xs + (r0 + 16 * i) * kLinPK + 4 * c4) = xreg[i];
    };
    lin_f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[rb][r] = 0.f;
    fetch(kbeg);
    const float* xrow = Xs + (lane & 31) * kLinPK + 4 * (lane >> 5);
    const float* wrow = Ws + (32 * wave + (lane & 31)) * kLinPK + 4 * (lane >> 5);
    for (int it = 0; it < a.kiters; it++) {
        if (it) __syncthreads();        // the previous tile has been read
        commit();
        __syncthreads();
        if (it + 1 < a.kiters) fetch(kbeg + (it + 1) * kLinKT);
#pragma unroll
        for (int G = 0; G < kLinKT / 8; G++) {
            const float4 b4 = lin_ld4(wrow + 8 * G);
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const float4 a4 = lin_ld4(xrow + 32 * rb * kLinPK + 8 * G);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[rb], 0, 0, 0);
            }
        }
    }
    lin_store<RB>(a, acc, n0 + 32 * wave + (lane & 31), lane);
}

// grid (ceil(N / 128), K slices), block 256, dynamic LDS 32 * RB * 68 + 64 * 136 floats.   N % 4 == 0, K % 4 == 0
template <int RB>
__global__ void __launch_bounds__(256) k_lin_nn(LinArgs a) {
    extern __shared__ float4 lin_lds4[];
    float* Xs = reinterpret_cast<float*>(lin_lds4);          // [32 * RB][68]   (b, k)
    float* Ws = Xs + 32 * RB * kLinPK;                       // [64][136]       (k, n)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * kLinNT;
    const int kbeg = blockIdx.y * a.kiters * kLinKT;
    const int wc4 = tid & 31, wr0 = tid >> 5;                // W: column quad wc4 (n) of rows wr0, wr0 + 8, ... (k)
    const int c4 = tid & 15, r0 = tid >> 4;                  // X: as in k_lin_nt
    float4 wreg[8], xreg[2 * RB];
    auto fetch = [&](int k0) {
        const int n = n0 + 4 * wc4;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int k = k0 + wr0 + 8 * i;
            wreg[i] = (n < a.N && k < a.K) ? lin_ld4(a.W + (long long)k * a.w_ld + n) : lin_zero4();
        }
        const int kx = k0 + 4 * c4;
#pragma unroll
        for (int i = 0; i < 2 * RB; i++) {
            const int b = r0 + 16 * i;
            xreg[i] = (b < a.B && kx < a.K) ? lin_ld4(a.X + (long long)b * a.x_ld + kx) : lin_zero4();
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 8; i++) *reinterpret_cast<float4*>(Ws + (wr0 + 8 * i) * kLinPN + 4 * wc4) = wreg[i];
#pragma unroll
        for (int i = 0; i < 2 * RB; i++) *reinterpret_cast<float4*>(Xs + (r0 + 16 * i) * kLinPK + 4 * c4) = xreg[i];
    };
    lin_f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[rb][r] = 0.f;
    fetch(kbeg);
    const float* xrow = Xs + (lane & 31) * kLinPK + 4 * (lane >> 5);
    const float* wcol = Ws + 4 * (lane >> 5) * kLinPN + 32 * wave + (lane & 31);
    for (int it = 0; it < a.kiters; it++) {
        if (it) __syncthreads();
        commit();
        __syncthreads();
        if (it + 1 < a.kiters) fetch(kbeg + (it + 1) * kLinKT);
#pragma unroll
        for (int G = 0; G < kLinKT / 8; G++) {
            float bv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) bv[j] = wcol[(8 * G + j) * kLinPN];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const float4 a4 = lin_ld4(xrow + 32 * rb * kLinPK + 8 * G);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bv[0], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bv[1], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bv[2], acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bv[3], acc[rb], 0, 0, 0);
            }
        }
    }
    lin_store<RB>(a, acc, n0 + 32 * wave + (lane & 31), lane);
}

// Y[b * y_ld + n] = bias[n] + sum over slices of part[(slice * B + b) * N + n]: fp64, slices q, q + 16, ... per thread and
// the sixteen partial sums in index order (no atomics: the result does not depend on timing).  grid ceil(B * N / 16), block 256
__global__ void __launch_bounds__(256) k_lin_fold(const float* __restrict__ part, int slices, int B, int N,
                                                  const float* __restrict__ bias, float* __restrict__ Y, long long y_ld) {
    __shared__ double red[16][17];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long long E = (long long)B * N, e = (long long)blockIdx.x * 16 + el;
    double s = 0.0;
    if (e < E) {
        int z = q;
        for (; z + 48 < slices; z += 64) {     // four loads in flight
            const float p0 = part[(long long)z * E + e], p1 = part[(long long)(z + 16) * E + e];
            const float p2 = part[(long long)(z + 32) * E + e], p3 = part[(long long)(z + 48) * E + e];
            s += (double)p0;
            s += (double)p1;
            s += (double)p2;
            s += (double)p3;
        }
        for (; z < slices; z += 16) s += (double)part[(long long)z * E + e];
    }
    red[q][el] = s;
    __syncthreads();
    if (q == 0 && e < E) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) t += red[i][el];
        const int b = (int)(e / N), n = (int)(e - (long long)b * N);
        Y[(long long)b * y_ld + n] = (float)(t + (bias ? (double)bias[n] : 0.0));
    }
}

// dW[n][k] = sum_b G[b][n] X[b][k] (fp32, plain stores: every element has one writer and the batch is the whole sum),
// db[n] = sum_b G[b][n] (fp64).  grid (ceil(K / 128), ceil(N / 128)), block 256, dynamic LDS 2 * B8 * 136 floats, B8 = B rounded
// up to 8.   N % 4 == 0, K % 4 == 0.  A wave owns 32 rows (n) x 128 columns (k): four accumulators.
__global__ void __launch_bounds__(256) k_lin_outer(const float* __restrict__ G, long long g_ld, const float* __restrict__ X,
                                                   long long x_ld, int B, int N, int K, float* __restrict__ dW,
                                                   double* __restrict__ db) {
    extern __shared__ float4 lin_lds4[];
    const int B8 = (B + 7) & ~7;
    float* Gs = reinterpret_cast<float*>(lin_lds4);          // [B8][136]   (b, n)
    float* Xs = Gs + B8 * kLinPN;                            // [B8][136]   (b, k)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int c4 = tid & 31, r0 = tid >> 5;
    for (int b = r0; b < B8; b += 8) {
        const bool ok = b < B;
        const int n = n0 + 4 * c4, k = k0 + 4 * c4;
        *reinterpret_cast<float4*>(Gs + b * kLinPN + 4 * c4) = (ok && n < N) ? lin_ld4(G + (long long)b * g_ld + n) : lin_zero4();
        *reinterpret_cast<float4*>(Xs + b * kLinPN + 4 * c4) = (ok && k < K) ? lin_ld4(X + (long long)b * x_ld + k) : lin_zero4();
    }
    __syncthreads();
    lin_f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const float* gcol = Gs + 4 * (lane >> 5) * kLinPN + 32 * wave + (lane & 31);
    const float* xcol = Xs + 4 * (lane >> 5) * kLinPN + (lane & 31);
    for (int g8 = 0; g8 < B8; g8 += 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float av = gcol[(g8 + j) * kLinPN];
#pragma unroll
            for (int c = 0; c < 4; c++)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xcol[(g8 + j) * kLinPN + 32 * c], acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int k = k0 + 32 * c + (lane & 31);
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = n0 + 32 * wave + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
            if (n < N) dW[(long long)n * K + k] = acc[c][r];
        }
    }
    if (db && blockIdx.x == 0 && tid < 128 && n0 + tid < N) {
        double s = 0.0;
        for (int b = 0; b < B; b++) s += (double)Gs[b * kLinPN + tid];
        db[n0 + tid] = s;
    }
}

}  // namespace
}  // namespace unet
