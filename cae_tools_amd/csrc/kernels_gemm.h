// kernels_gemm.h — small fp32 GEMM on the CDNA4 matrix cores for the Linear layers
// (encoder.py:54-58, decoder.py:31-35): forward, input gradient and weight gradient are all
// C[M,N] = A[M,K] * B[K,N] with different strides.
//
// The matrices are tiny (M = batch 64, N,K <= 576) and L2-resident, so the kernel is built for
// latency, not for operand reuse: one wave owns one 16x16 tile of C and walks K with
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate), operands fetched straight from
// global memory into the MFMA lane layout (lane l: A[row l&15][k0 + (l>>4)], B[k0 + (l>>4)][col l&15]).
// 4 waves per workgroup; grid = ceil(tiles / 4).
#pragma once
#include "kernels_generic.h"

namespace cae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum GemmEpi : int {
    GE_STORE = 0,        // C = acc (+bias[n]) (relu)
    GE_RELU_MASK = 1,    // C = H[m][n] > 0 ? acc : 0        (gradient through the producer's ReLU)
    GE_BN_MASK = 2,      // C = bnrelu(Y[m][n]) > 0 ? acc : 0, D1/D2 sums of channel n / hw
    GE_ACC64 = 3,        // double accumulator: accW[m][n] = acc, column N-1 (ones) -> accB[m]
    GE_ACC64_T = 4       // transposed: accW[n][m] = acc, row M-1 of A is the constant 1 -> accB[n]
};

struct GemmArgs {
    int M, N, K;
    const float* A; long long sa_m, sa_k;
    const float* B; long long sb_k, sb_n;
    float* C; long long sc_m, sc_n;
    int epi;
    const float* bias;     // GE_STORE: per n, or nullptr
    int relu;              // GE_STORE
    const float* H;        // GE_RELU_MASK / GE_BN_MASK: same indexing as C
    BnDesc bn_a;           // BatchNorm+ReLU applied to A elements, channel = k / hw_a (BN_NONE: identity)
    int hw_a;
    int bn_a_by_row;       // 1: the BatchNorm channel of A[m][k] is m / hw_a (default: k / hw_a)
    BnDesc bn_c;           // GE_BN_MASK: BN_SAVED descriptor, channel = n / hw_c
    int hw_c;
    double* stats_c;       // GE_BN_MASK: [C][4] sums (slots 2,3)
    double* accW; double* accB;  // GE_ACC64; B's last column (n == N-1) is the constant 1 when ones_col
    int ones_col;
    int ones_row;                // GE_ACC64_T: A's last row (m == M-1) is the constant 1
};

// One workgroup = 4 waves = one 16x16 tile of C; the waves split K into 4 contiguous chunks
// (split-K inside the workgroup, combined through LDS), so even a 64x128x576 product keeps
// every k-chain short.  Loads are issued 8 k-steps at a time ahead of their MFMAs.
__device__ __forceinline__ void gemm16_body(const GemmArgs& g, const int bx, double* lds_d) {
    float* part = reinterpret_cast<float*>(lds_d);              // [4 waves][256] partial tiles
    double* lstat = reinterpret_cast<double*>(part + 4 * 256);  // [<=64 channels][2] GE_BN_MASK sums: fp64 on the accumulation grid
    float4* ca = reinterpret_cast<float4*>(part + 4 * 256 + 256);
    float4* cc = ca + (g.bn_a.mode ? g.bn_a.C : 0);
    bn_consts(g.bn_a, ca, bx == 0);
    bn_consts(g.bn_c, cc, false, 64);
    if (threadIdx.x < 128) lstat[threadIdx.x] = 0.0;
    __syncthreads();

    const int tiles_n = (g.N + 15) >> 4;
    const int tile = bx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int am = tm * 16 + r;   // A row fetched by this lane
    const int bn = tn * 16 + r;   // B column fetched by this lane
    const bool am_ok = am < g.M, bn_ok = bn < g.N;
    const bool b_ones = g.ones_col && bn == g.N - 1;
    const bool a_ones = g.ones_row && am == g.M - 1;
    const float* ap = g.A + (long long)am * g.sa_m;
    const float* bp = g.B + (long long)bn * g.sb_n;

    // this wave's K chunk, in whole k-steps of 4
    const int steps = (g.K + 3) >> 2;
    const int per = (steps + 3) >> 2;
    const int s0 = wv * per, s1 = min(steps, s0 + per);

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int st = s0; st < s1; st += 8) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = (st + u) * 4 + q;
            const bool k_ok = (st + u) < s1 && k < g.K;
            a[u] = (am_ok && k_ok) ? (a_ones ? 1.0f : ap[(long long)k * g.sa_k]) : 0.f;
            b[u] = (bn_ok && k_ok) ? (b_ones ? 1.0f : bp[(long long)k * g.sb_k]) : 0.f;
            if (g.bn_a.mode != BN_NONE && am_ok && k_ok && !a_ones) {
                // channel of an A element: k / hw_a, or m / hw_a when the transform runs along the rows
                const float4 c4 = ca[(g.bn_a_by_row ? am : k) / g.hw_a];
                a[u] = fmaxf(0.f, fmaf(a[u] - c4.x, c4.y, c4.z));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) part[wv * 256 + j * 64 + lane] = acc[j];
    __syncthreads();
    if (wv != 0) {
        if (g.epi != GE_BN_MASK) return;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            acc[j] = part[j * 64 + lane] + part[256 + j * 64 + lane] + part[512 + j * 64 + lane] + part[768 + j * 64 + lane];
    }

    // C/D layout of the 16x16 MFMA: register j of lane l is C[row 4*(l>>4) + j][col l&15]
    const int cn = tn * 16 + r;
    if (wv == 0 && cn < g.N) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int cm = tm * 16 + q * 4 + j;
            if (cm >= g.M) continue;
            float v = acc[j];
            const long long off = (long long)cm * g.sc_m + (long long)cn * g.sc_n;
            if (g.epi == GE_STORE) {
                if (g.bias) v += g.bias[cn];
                if (g.relu) v = fmaxf(v, 0.f);
                g.C[off] = v;
            } else if (g.epi == GE_RELU_MASK) {
                g.C[off] = g.H[off] > 0.f ? v : 0.f;
            } else if (g.epi == GE_BN_MASK) {
                const int ch = cn / g.hw_c;
                const float4 c4 = cc[ch];
                const float d = g.H[off] - c4.x;
                const float gm = fmaf(d, c4.y, c4.z) > 0.f ? v : 0.f;
                g.C[off] = gm;
                // LDS atomics (a tile spans <= 16 channels), many lanes per slot: addends on the accumulation grid, so that the
                // adds are exact and their order does not matter
                atomicAdd(&lstat[2 * ch], acc_grid<ACC_GRAD>((double)gm));
                atomicAdd(&lstat[2 * ch + 1], acc_grid<ACC_GRAD>((double)(gm * (d * c4.w))));
            } else if (g.epi == GE_ACC64_T) {
                if (g.ones_row && cm == g.M - 1) {
                    g.accB[cn] = (double)v;
                } else {
                    g.accW[(long long)cn * (g.M - (g.ones_row ? 1 : 0)) + cm] = (double)v;
                }
            } else {  // GE_ACC64
                if (g.ones_col && cn == g.N - 1) {
                    g.accB[cm] = (double)v;
                } else {
                    g.accW[(long long)cm * (g.N - (g.ones_col ? 1 : 0)) + cn] = (double)v;
                }
            }
        }
    }
    if (g.epi == GE_BN_MASK) {
        __syncthreads();
        const int c0 = (tn * 16) / g.hw_c, c1 = min(g.bn_c.C - 1, (tn * 16 + 15) / g.hw_c);
        const int i = threadIdx.x;
        if (i < 2 * (c1 - c0 + 1)) {
            const int ch = c0 + (i >> 1);
            acc_add<ACC_GRAD>(&g.stats_c[((size_t)(bx & (kStatShards - 1)) * g.bn_c.C + ch) * 4 + 2 + (i & 1)],
                      lstat[2 * ch + (i & 1)]);
        }
    }
}

// The same tile for a long contraction with A row-major (sa_k == 1), B[k][n] contiguous in n (sb_n == 1), K a multiple of
// 4 and at most 4 * kBurstSteps k-steps: every operand read is issued in one burst at kernel start — this wave's B operands
// into registers (4 cache lines per wave load), the tile's 16 A rows as coalesced 16-byte loads into LDS (fetched in the MFMA
// lane layout they cost 16 lines per wave load), the ReLU-mask inputs of the epilogue — so the k loop runs out of LDS and
// registers instead of paying a dependent memory round trip per batch of eight k-steps.  Epilogues: GE_STORE, GE_RELU_MASK.
constexpr int kBurstSteps = 40;
inline size_t gemm16_burst_lds(int K) { return (size_t)(4 * 256 + 16 * (K + 4)) * sizeof(float); }

__device__ __forceinline__ void gemm16_burst_body(const GemmArgs& g, const int bx, double* lds_d) {
    float* part = reinterpret_cast<float*>(lds_d);   // [4][256]
    float* ap = part + 4 * 256;                      // [16][lda]
    const int lda = g.K + 4;
    const int tiles_n = (g.N + 15) >> 4;
    const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int steps = g.K >> 2, per = (steps + 3) >> 2;
    const int s0 = wv * per, s1 = min(steps, s0 + per);

    const int bn = tn * 16 + r;
    const float* bp = g.B + (long long)min(bn, g.N - 1) * g.sb_n;
    // ---- one burst: B operands, epilogue inputs, the tile's A rows; nothing is consumed before all of it is in flight ----
    float b[kBurstSteps];
#pragma unroll
    for (int u = 0; u < kBurstSteps; u++) b[u] = bp[(long long)min((s0 + u) * 4 + q, g.K - 1) * g.sb_k];
    const int cn = tn * 16 + r;
    float hmask[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.epi == GE_RELU_MASK) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            hmask[j] = g.H[(long long)min(tm * 16 + q * 4 + j, g.M - 1) * g.sc_m + (long long)min(cn, g.N - 1) * g.sc_n];
    }
    const float bias = (g.epi == GE_STORE && g.bias) ? g.bias[min(cn, g.N - 1)] : 0.f;
    // float4s of the A panel per thread: 16 rows * (K / 4) / 256 threads <= 16 * (4 * kBurstSteps) / 256
    static_assert(16 * (4 * kBurstSteps) / 256 == 10, "A panel registers");
    const int k4 = g.K >> 2, n4 = 16 * k4;
    const float inv_k4 = 1.0f / (float)k4;
    f32x4 av[10];
#pragma unroll
    for (int u = 0; u < 10; u++) {
        const int i = min(tid + u * 256, n4 - 1);
        const int row = div_small(i, inv_k4);
        av[u] = *reinterpret_cast<const f32x4*>(g.A + (long long)min(tm * 16 + row, g.M - 1) * g.sa_m + 4 * (i - row * k4));
    }
    // The loads above are unconditional (clamped addresses) and must stay so: left to itself the compiler moves each one
    // under the predicate of its consumer and waits for it there — forty dependent round trips instead of one burst.
#pragma unroll
    for (int u = 0; u < kBurstSteps; u++) asm volatile("" : "+v"(b[u]));
    asm volatile("" : "+v"(hmask[0]), "+v"(hmask[1]), "+v"(hmask[2]), "+v"(hmask[3]));
#pragma unroll
    for (int u = 0; u < 10; u++) asm volatile("" : "+v"(av[u]));
#pragma unroll
    for (int u = 0; u < kBurstSteps; u++) b[u] = ((s0 + u) < s1 && bn < g.N) ? b[u] : 0.f;
#pragma unroll
    for (int u = 0; u < 10; u++) {
        const int i = tid + u * 256;
        if (i < n4) {
            const int row = div_small(i, inv_k4);
            *reinterpret_cast<f32x4*>(ap + row * lda + 4 * (i - row * k4)) = tm * 16 + row < g.M ? av[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* arow = ap + r * lda + q;
#pragma unroll
    for (int u = 0; u < kBurstSteps; u++) {
        const int k4 = min(s0 + u, steps - 1);   // past this wave's range b[u] is zero; keep the read inside the panel
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * k4], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) part[wv * 256 + j * 64 + lane] = acc[j];
    __syncthreads();
    if (wv != 0 || cn >= g.N) return;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int cm = tm * 16 + q * 4 + j;
        if (cm >= g.M) continue;
        float v = part[j * 64 + lane] + part[256 + j * 64 + lane] + part[512 + j * 64 + lane] + part[768 + j * 64 + lane];
        const long long off = (long long)cm * g.sc_m + (long long)cn * g.sc_n;
        if (g.epi == GE_STORE) {
            v += bias;
            if (g.relu) v = fmaxf(v, 0.f);
        } else {
            v = hmask[j] > 0.f ? v : 0.f;
        }
        g.C[off] = v;
    }
}

__global__ void __launch_bounds__(256) k_gemm16(GemmArgs g) {
    extern __shared__ double lds_d[];
    gemm16_body(g, blockIdx.x, lds_d);
}

// two independent GEMMs in one launch (a Linear layer's weight gradient beside its input gradient):
// workgroups [0, na) run `a`, the rest run `b`
__global__ void __launch_bounds__(256) k_gemm16_pair(GemmArgs a, GemmArgs b, int na, int b_burst) {
    kernarg_warm<2 * sizeof(GemmArgs) + 8>();
    extern __shared__ double lds_d[];
    if ((int)blockIdx.x < na) gemm16_body(a, blockIdx.x, lds_d);
    else if (b_burst) gemm16_burst_body(b, blockIdx.x - na, lds_d);
    else gemm16_body(b, blockIdx.x - na, lds_d);
}

}  // namespace cae
