// kernels_unet_mfma.h — LDS-tiled fp32 MFMA implicit GEMMs for the 4x4 / stride 2 / padding 1 layers of the UNET path.
//
// All three conv primitives are computed as D[n][m] = sum_k W[n][k] * X[k][m] with the CHANNEL-like index on the
// rows (MFMA A operand) and the PIXEL-like index on the columns (MFMA B operand), so that the 32 lanes of a
// v_mfma_f32_32x32x2_f32 result column run along contiguous memory when the tile is stored:
//
//   down  (Conv2d fwd, ConvT dgrad): n = cs,  m = (b,y,x) of S,  k = (cl,ky,kx)          K = Cl*16
//   up    (ConvT fwd, Conv2d dgrad): n = cl,  m = (b,q,r) of S,  k = (cs,j,i) per output parity (py,px), K = Cs*4
//                                    output pixel (2q+py, 2r+px) sees S[q+py-j][r+px-i] through w[..][1-py+2j][1-px+2i]
//   wgrad                          : n = cs,  m = (cl,ky,kx),     k = (b,y,x) of S,       K = B*Hs*Ws, split over blocks
//
// One workgroup = 4 waves = a (32*WN) x (128*WM) tile, WN*WM = 4; each wave owns 32 rows x 128 columns = four
// 32x32 accumulators (64 VGPRs).  K is walked in chunks of 16 through a double-buffered LDS image
// (W[16][TN+1], X[16][TM+1]); the gathers of chunk c+1 are in flight while chunk c is multiplied.  Each primitive
// chooses the thread -> element mapping of its gathers for global-memory coalescing (lanes along x), independent of
// the MFMA operand layout, which the LDS image provides.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_unet.h"

namespace unet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IG_KC = 16;

// ---------------------------------------------------------------------------------------------
// policies
// ---------------------------------------------------------------------------------------------
template <int WM>
struct OpDown {
    Geom g;
    const float* __restrict__ L;
    const float* __restrict__ w;
    const float* __restrict__ bias;
    float* __restrict__ S;
    int nsplit, ksplit;   // K slices (blockIdx.z) of ksplit chunks; nsplit > 1: S is pre-zeroed and added to atomically
    struct Ctx {
        long long xbase[WM];
        unsigned vmask[WM];
    };
    __device__ int rows() const { return g.Cs; }
    __device__ long long cols() const { return (long long)g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(g.Cl * 16, (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx& c, int tid, long long m0, int) const {
        const int HW = g.Hs * g.Ws;
#pragma unroll
        for (int j = 0; j < WM; j++) {
            const long long m = m0 + (tid & 127) + 128 * j;
            c.vmask[j] = 0;
            c.xbase[j] = 0;
            if (m < cols()) {
                const int b = (int)(m / HW), p = (int)(m - (long long)b * HW);
                const int y = p / g.Ws, x = p - y * g.Ws;
                const int Y0 = 2 * y - 1, X0 = 2 * x - 1;
                c.xbase[j] = (long long)b * g.Cl * g.Hl * g.Wl + (long long)Y0 * g.Wl + X0;
                unsigned mk = 0;
                for (int t = 0; t < 16; t++) {
                    const int Y = Y0 + (t >> 2), X = X0 + (t & 3);
                    if (Y >= 0 && Y < g.Hl && X >= 0 && X < g.Wl) mk |= 1u << t;
                }
                c.vmask[j] = mk;
            }
        }
    }
    template <int NW>
    __device__ void gather(const Ctx& c, int tid, int n0, int kc, int, float* wr, float* xr) const {
        const int K = g.Cl * 16;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int n = n0 + (tid >> 4) + 16 * i;
            wr[i] = n < g.Cs ? w[(size_t)n * K + kc + (tid & 15)] : 0.f;
        }
        const long long plane = (long long)(kc >> 4) * g.Hl * g.Wl;
#pragma unroll
        for (int j = 0; j < WM; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int t = (tid >> 7) + 2 * i;
                xr[j * 8 + i] = (c.vmask[j] >> t) & 1u ? L[c.xbase[j] + plane + (t >> 2) * g.Wl + (t & 3)] : 0.f;
            }
        }
    }
    template <int NW, int WS, int XS>
    __device__ void commit(int tid, float* Wb, float* Xb, const float* wr, const float* xr) const {
#pragma unroll
        for (int i = 0; i < NW; i++) Wb[(tid & 15) * WS + (tid >> 4) + 16 * i] = wr[i];
#pragma unroll
        for (int j = 0; j < WM; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) Xb[((tid >> 7) + 2 * i) * XS + (tid & 127) + 128 * j] = xr[j * 8 + i];
    }
    // the column (pixel) part of an output address, once per result column; -1: outside
    __device__ long long col(long long m, int) const {
        if (m >= cols()) return -1;
        const int HW = g.Hs * g.Ws;
        const long long b = m / HW;
        return b * g.Cs * HW + (m - b * HW);
    }
    __device__ void store(long long cb, int n, float v, int z) const {
        if (n >= g.Cs || cb < 0) return;
        float* dst = S + cb + (long long)n * g.Hs * g.Ws;
        if (z == 0 && bias) v += bias[n];
        if (nsplit > 1) atomicAdd(dst, v);
        else *dst = v;
    }
};

// weights repacked per parity: wp[z][cl][cs*4 + j*2 + i] = w[cs][cl][1-py+2j][1-px+2i],  z = py*2 + px
__global__ void __launch_bounds__(256) k_pack_up_weights(int Cs, int Cl, const float* __restrict__ w, float* __restrict__ wp) {
    const long long total = (long long)4 * Cl * Cs * 4;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int k = (int)(o % (Cs * 4));
        const long long r = o / (Cs * 4);
        const int cl = (int)(r % Cl), z = (int)(r / Cl);
        const int cs = k >> 2, j = (k >> 1) & 1, i = k & 1;
        const int ky = 1 - (z >> 1) + 2 * j, kx = 1 - (z & 1) + 2 * i;
        wp[o] = w[((size_t)cs * Cl + cl) * 16 + ky * 4 + kx];
    }
}

template <int WM>
struct OpUp {
    Geom g;
    const float* __restrict__ S;
    const float* __restrict__ wp;   // packed weights [4][Cl][Cs*4]
    const float* __restrict__ bias;
    float* __restrict__ L;
    struct Ctx {
        long long sbase[WM];
        unsigned vmask[WM];
    };
    __device__ int rows() const { return g.Cl; }
    __device__ long long cols() const { return (long long)g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int) const { return 0; }
    __device__ int k_end(int) const { return g.Cs * 4; }
    __device__ void init(Ctx& c, int tid, long long m0, int z) const {
        const int HW = g.Hs * g.Ws;
        const int py = z >> 1, px = z & 1;
#pragma unroll
        for (int j = 0; j < WM; j++) {
            const long long m = m0 + (tid & 127) + 128 * j;
            c.vmask[j] = 0;
            c.sbase[j] = 0;
            if (m < cols()) {
                const int b = (int)(m / HW), p = (int)(m - (long long)b * HW);
                const int q = p / g.Ws, r = p - q * g.Ws;
                c.sbase[j] = (long long)b * g.Cs * HW + (long long)q * g.Ws + r;
                unsigned mk = 0;
                for (int t = 0; t < 4; t++) {
                    const int yy = q + py - (t >> 1), xx = r + px - (t & 1);
                    if (yy >= 0 && yy < g.Hs && xx >= 0 && xx < g.Ws) mk |= 1u << t;
                }
                c.vmask[j] = mk;
            }
        }
    }
    template <int NW>
    __device__ void gather(const Ctx& c, int tid, int n0, int kc, int z, float* wr, float* xr) const {
        const int K = g.Cs * 4, HW = g.Hs * g.Ws;
        const int py = z >> 1, px = z & 1;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int n = n0 + (tid >> 4) + 16 * i;
            wr[i] = n < g.Cl ? wp[((size_t)z * g.Cl + n) * K + kc + (tid & 15)] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WM; j++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int k = kc + (tid >> 7) + 2 * i;
                const int cs = k >> 2, t = k & 3;
                const int off = (py - (t >> 1)) * g.Ws + (px - (t & 1));
                xr[j * 8 + i] = (c.vmask[j] >> t) & 1u ? S[c.sbase[j] + (long long)cs * HW + off] : 0.f;
            }
        }
    }
    template <int NW, int WS, int XS>
    __device__ void commit(int tid, float* Wb, float* Xb, const float* wr, const float* xr) const {
#pragma unroll
        for (int i = 0; i < NW; i++) Wb[(tid & 15) * WS + (tid >> 4) + 16 * i] = wr[i];
#pragma unroll
        for (int j = 0; j < WM; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) Xb[((tid >> 7) + 2 * i) * XS + (tid & 127) + 128 * j] = xr[j * 8 + i];
    }
    __device__ long long col(long long m, int z) const {
        if (m >= cols()) return -1;
        const int HW = g.Hs * g.Ws;
        const long long b = m / HW;
        const int p = (int)(m - b * HW);
        const int q = p / g.Ws, r = p - q * g.Ws;
        return (b * g.Cl * g.Hl + 2 * q + (z >> 1)) * g.Wl + 2 * r + (z & 1);
    }
    __device__ void store(long long cb, int n, float v, int) const {
        if (n >= g.Cl || cb < 0) return;
        L[cb + (long long)n * g.Hl * g.Wl] = v + (bias ? bias[n] : 0.f);
    }
};

template <int WM>
struct OpWgrad {
    Geom g;
    const float* __restrict__ S;
    const float* __restrict__ L;
    double* __restrict__ acc;
    int ksplit;   // chunks of 16 per z-slice
    struct Ctx {
        int dummy;
    };
    __device__ int rows() const { return g.Cs; }
    __device__ long long cols() const { return (long long)g.Cl * 16; }
    __device__ int ktotal() const { return g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(ktotal(), (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx&, int, long long, int) const {}
    template <int NW>
    __device__ void gather(const Ctx&, int tid, int n0, int kc, int, float* wr, float* xr, long long m0) const {
        const int HW = g.Hs * g.Ws;
        const int k = kc + (tid & 15);
        const bool kok = k < ktotal();
        const int b = k / HW, p = k - b * HW;
        const int y = p / g.Ws, x = p - y * g.Ws;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int n = n0 + (tid >> 4) + 16 * i;
            wr[i] = (kok && n < g.Cs) ? S[((size_t)b * g.Cs + n) * HW + p] : 0.f;
        }
        const long long lb = (long long)b * g.Cl * g.Hl * g.Wl + (long long)(2 * y - 1) * g.Wl + (2 * x - 1);
#pragma unroll
        for (int i = 0; i < 8 * WM; i++) {
            const int m = (int)m0 + (tid >> 4) + 16 * i;
            const int cl = m >> 4, ky = (m >> 2) & 3, kx = m & 3;
            const int Y = 2 * y - 1 + ky, X = 2 * x - 1 + kx;
            const bool ok = kok && cl < g.Cl && Y >= 0 && Y < g.Hl && X >= 0 && X < g.Wl;
            xr[i] = ok ? L[lb + (long long)cl * g.Hl * g.Wl + ky * g.Wl + kx] : 0.f;
        }
    }
    template <int NW, int WS, int XS>
    __device__ void commit(int tid, float* Wb, float* Xb, const float* wr, const float* xr) const {
#pragma unroll
        for (int i = 0; i < NW; i++) Wb[(tid & 15) * WS + (tid >> 4) + 16 * i] = wr[i];
#pragma unroll
        for (int i = 0; i < 8 * WM; i++) Xb[(tid & 15) * XS + (tid >> 4) + 16 * i] = xr[i];
    }
    __device__ long long col(long long m, int) const { return m < cols() ? m : -1; }
    __device__ void store(long long cb, int n, float v, int) const {
        if (n >= g.Cs || cb < 0 || v == 0.f) return;
        atomicAdd(&acc[(size_t)n * g.Cl * 16 + cb], (double)v);
    }
};

template <class Op>
struct is_wgrad {
    static constexpr bool value = false;
};
template <int WM>
struct is_wgrad<OpWgrad<WM>> {
    static constexpr bool value = true;
};


// ---------------------------------------------------------------------------------------------
// plain GEMM through the same tile engine (the Linear layers): D[n][m] = sum_k A[n][k] * Bm[k][m] with arbitrary
// strides (one of each operand's two strides is 1; the gather runs its lanes along that one).
//   store 0: out[m*o_sm + n*o_sn] = v + bias[n]
//   store 1: fp64 atomicAdd into scratch[n*cols + m]  (split K; k_gemm_finish applies store 0 afterwards)
//   store 2: accd[n*o_sn + m*o_sm] += v               (fp64 gradient accumulator, one writer per element)
// ---------------------------------------------------------------------------------------------
template <int WM>
struct OpGemm {
    int nrows, ncols, K;
    const float* __restrict__ a;
    long long a_sn, a_sk;
    const float* __restrict__ b;
    long long b_sk, b_sm;
    const float* __restrict__ bias;
    float* __restrict__ out;
    double* __restrict__ accd;
    long long o_sn, o_sm;
    int mode, ksplit;
    struct Ctx {
        int dummy;
    };
    __device__ int rows() const { return nrows; }
    __device__ long long cols() const { return ncols; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(K, (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx&, int, long long, int) const {}
    template <int NW>
    __device__ void gather(const Ctx&, int tid, int n0, int kc, int, float* wr, float* xr, long long m0) const {
        constexpr int TN = NW * 16, TM = 128 * WM;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            int nl, kl;
            if (a_sk == 1) {
                kl = tid & 15, nl = (tid >> 4) + 16 * i;
            } else {
                const int e = tid + 256 * i;
                nl = e % TN, kl = e / TN;
            }
            const int n = n0 + nl, k = kc + kl;
            wr[i] = (n < nrows && k < K) ? a[n * a_sn + k * a_sk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8 * WM; i++) {
            int ml, kl;
            if (b_sk == 1) {
                kl = tid & 15, ml = (tid >> 4) + 16 * i;
            } else {
                const int e = tid + 256 * i;
                ml = e % TM, kl = e / TM;
            }
            const long long m = m0 + ml;
            const int k = kc + kl;
            xr[i] = (m < ncols && k < K) ? b[k * b_sk + m * b_sm] : 0.f;
        }
    }
    template <int NW, int WS, int XS>
    __device__ void commit(int tid, float* Wb, float* Xb, const float* wr, const float* xr) const {
        constexpr int TN = NW * 16, TM = 128 * WM;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if (a_sk == 1) {
                Wb[(tid & 15) * WS + (tid >> 4) + 16 * i] = wr[i];
            } else {
                const int e = tid + 256 * i;
                Wb[(e / TN) * WS + e % TN] = wr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8 * WM; i++) {
            if (b_sk == 1) {
                Xb[(tid & 15) * XS + (tid >> 4) + 16 * i] = xr[i];
            } else {
                const int e = tid + 256 * i;
                Xb[(e / TM) * XS + e % TM] = xr[i];
            }
        }
    }
    __device__ long long col(long long m, int) const { return m < ncols ? m : -1; }
    __device__ void store(long long m, int n, float v, int) const {
        if (n >= nrows || m < 0) return;
        if (mode == 0) {
            out[m * o_sm + n * o_sn] = v + (bias ? bias[n] : 0.f);
        } else if (mode == 1) {
            if (v != 0.f) atomicAdd(&accd[(size_t)n * ncols + m], (double)v);
        } else {
            accd[n * o_sn + m * o_sm] += (double)v;
        }
    }
};
template <int WM>
struct is_wgrad<OpGemm<WM>> {
    static constexpr bool value = true;   // same gather signature (takes m0)
};


// grid (col tiles, row tiles, z): z = output parity (up) or K slice (wgrad)
template <int WN, int WM, class Op>
__global__ void __launch_bounds__(256) k_igemm(Op op) {
    constexpr int TN = 32 * WN, TM = 128 * WM, KC = IG_KC, WS = TN + 1, XS = TM + 1;
    constexpr int NW = TN / 16, NX = 8 * WM;
    constexpr int BUF = KC * (WS + XS);
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave - wn * WM;
    const long long m0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const int z = blockIdx.z;
    typename Op::Ctx ctx;
    op.init(ctx, tid, m0, z);
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float wr[NW], xr[NX];
    const int kb = op.k_begin(z), ke = op.k_end(z);
    auto gather = [&](int kc) {
        if constexpr (is_wgrad<Op>::value) op.template gather<NW>(ctx, tid, n0, kc, z, wr, xr, m0);
        else op.template gather<NW>(ctx, tid, n0, kc, z, wr, xr);
    };
    if (kb < ke) {
        gather(kb);
        op.template commit<NW, WS, XS>(tid, lds, lds + KC * WS, wr, xr);
    }
    __syncthreads();
    int buf = 0;
    for (int kc = kb; kc < ke; kc += KC) {
        const bool more = kc + KC < ke;
        if (more) gather(kc + KC);
        const float* Wb = lds + buf * BUF;
        const float* Xb = Wb + KC * WS;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a = Wb[kr * WS + wn * 32 + (lane & 31)];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float b = Xb[kr * XS + wm * 128 + c * 32 + (lane & 31)];
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            }
        }
        if (more) {
            float* Wn = lds + (buf ^ 1) * BUF;
            op.template commit<NW, WS, XS>(tid, Wn, Wn + KC * WS, wr, xr);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const long long cb = op.col(m0 + wm * 128 + c * 32 + (lane & 31), z);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            op.store(cb, n, acc[c][r], z);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: eligibility and launch
// ---------------------------------------------------------------------------------------------
inline bool mfma_geom(const Geom& g) {
    return g.kh == 4 && g.kw == 4 && g.s == 2 && g.p == 1 && g.Hl == 2 * g.Hs && g.Wl == 2 * g.Ws &&
           (long long)g.B * g.Cl * g.Hl * g.Wl < (1ll << 31) && (long long)g.B * g.Cs * g.Hs * g.Ws < (1ll << 31);
}
inline bool mfma_down_eligible(const Geom& g) { return mfma_geom(g) && g.Cs >= 8; }
inline bool mfma_up_eligible(const Geom& g) { return mfma_geom(g) && g.Cs % 4 == 0 && g.Cs >= 8; }
inline bool mfma_wgrad_eligible(const Geom& g) { return mfma_geom(g) && g.Cs >= 8; }

template <template <int> class OpT, class Fill>
inline void igemm_dispatch(int rows, long long cols, int zdim, hipStream_t s, Fill fill) {
    auto lds_bytes = [](int WN, int WM) { return (size_t)2 * IG_KC * ((32 * WN + 1) + (128 * WM + 1)) * sizeof(float); };
    if (rows <= 32) {
        OpT<4> op;
        fill(op);
        static bool big_lds = false;   // 69,888 B: above the 64 KiB a kernel gets without asking
        if (!big_lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm<1, 4, OpT<4>>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(1, 4));
            big_lds = true;
        }
        hipLaunchKernelGGL((k_igemm<1, 4, OpT<4>>), dim3((unsigned)((cols + 511) / 512), (rows + 31) / 32, zdim), dim3(256),
                           lds_bytes(1, 4), s, op);
    } else if (rows <= 64) {
        OpT<2> op;
        fill(op);
        hipLaunchKernelGGL((k_igemm<2, 2, OpT<2>>), dim3((unsigned)((cols + 255) / 256), (rows + 63) / 64, zdim), dim3(256),
                           lds_bytes(2, 2), s, op);
    } else {
        OpT<1> op;
        fill(op);
        hipLaunchKernelGGL((k_igemm<4, 1, OpT<1>>), dim3((unsigned)((cols + 127) / 128), (rows + 127) / 128, zdim), dim3(256),
                           lds_bytes(4, 1), s, op);
    }
}

inline void mfma_down_launch(const Geom& g, const float* L, const float* w, const float* bias, float* S, hipStream_t s) {
    // few output tiles and a long K (the deep layers): slice K over blockIdx.z so that every CU has work
    const int rows = g.Cs;
    const long long cols = (long long)g.B * g.Hs * g.Ws;
    const int TN = rows <= 32 ? 32 : rows <= 64 ? 64 : 128, TM = rows <= 32 ? 512 : rows <= 64 ? 256 : 128;
    const long long tiles = ((rows + TN - 1) / TN) * ((cols + TM - 1) / TM);
    const int chunks = g.Cl;   // K / 16
    int nsplit = 1;
#ifndef IG_SPLIT_TARGET
#define IG_SPLIT_TARGET 384
#endif
    while (tiles * nsplit < IG_SPLIT_TARGET && chunks / (nsplit * 2) >= 8 && nsplit < 8) nsplit *= 2;
    const int per = (chunks + nsplit - 1) / nsplit;
    if (nsplit > 1) (void)hipMemsetAsync(S, 0, (size_t)cols * rows * sizeof(float), s);
    igemm_dispatch<OpDown>(rows, cols, nsplit, s, [&](auto& op) {
        op.g = g, op.L = L, op.w = w, op.bias = bias, op.S = S, op.nsplit = nsplit, op.ksplit = per;
    });
}

// scratch: Cs*Cl*16 floats for the repacked weights
inline void mfma_up_launch(const Geom& g, const float* S, const float* w, const float* bias, float* L, float* scratch,
                           hipStream_t s) {
    const long long total = (long long)16 * g.Cl * g.Cs;
    hipLaunchKernelGGL(k_pack_up_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g.Cs, g.Cl, w, scratch);
    igemm_dispatch<OpUp>(g.Cl, (long long)g.B * g.Hs * g.Ws, 4, s, [&](auto& op) {
        op.g = g, op.S = S, op.wp = scratch, op.bias = bias, op.L = L;
    });
}

inline void mfma_wgrad_launch(const Geom& g, const float* S, const float* L, double* acc, hipStream_t s) {
    const int rows = g.Cs;
    const long long cols = (long long)g.Cl * 16;
    const int TN = rows <= 32 ? 32 : rows <= 64 ? 64 : 128, TM = rows <= 32 ? 512 : rows <= 64 ? 256 : 128;
    const long long tiles = ((rows + TN - 1) / TN) * ((cols + TM - 1) / TM);
    const int chunks = (g.B * g.Hs * g.Ws + IG_KC - 1) / IG_KC;
    long long want = (1024 + tiles - 1) / tiles;          // aim at ~1024 workgroups
    if (want < 1) want = 1;
    int per = (int)((chunks + want - 1) / want);
    if (per < 8) per = 8;                                 // at least 128 pixels per slice
    const int zdim = (chunks + per - 1) / per;
    igemm_dispatch<OpWgrad>(rows, cols, zdim, s, [&](auto& op) {
        op.g = g, op.S = S, op.L = L, op.acc = acc, op.ksplit = per;
    });
}

// out[m*o_sm + n*o_sn] = scratch[n*cols + m] + bias[n]; scratch is cleared for the next use
__global__ void __launch_bounds__(256) k_gemm_finish(int nrows, int ncols, double* __restrict__ scratch,
                                                     const float* __restrict__ bias, float* __restrict__ out, long long o_sn,
                                                     long long o_sm) {
    const int total = nrows * ncols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int n = e / ncols, m = e - n * ncols;
        out[m * o_sm + n * o_sn] = (float)(scratch[e] + (bias ? (double)bias[n] : 0.0));
        scratch[e] = 0.0;
    }
}

// accb[o] += sum_b g[b*nout + o]
__global__ void __launch_bounds__(256) k_col_sums(int B, int nout, const float* __restrict__ g, double* __restrict__ accb) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    double s = 0;
    for (int b = 0; b < B; b++) s += (double)g[(size_t)b * nout + o];
    accb[o] += s;
}

struct GemmDesc {
    int rows, cols, K;
    const float* a;
    long long a_sn, a_sk;
    const float* b;
    long long b_sk, b_sm;
    const float* bias;
    float* out;
    double* accd;
    long long o_sn, o_sm;
    int mode;
};

// scratch: zero-filled doubles, rows*cols of them, only touched when K is split
inline void gemm_launch(const GemmDesc& d, double* scratch, hipStream_t s) {
    const int TN = d.rows <= 32 ? 32 : d.rows <= 64 ? 64 : 128, TM = d.rows <= 32 ? 512 : d.rows <= 64 ? 256 : 128;
    const long long tiles = (long long)((d.rows + TN - 1) / TN) * ((d.cols + TM - 1) / TM);
    const int chunks = (d.K + IG_KC - 1) / IG_KC;
    int zdim = 1, per = chunks;
    if (d.mode == 0 && scratch && tiles < 256 && chunks >= 64) {      // few tiles, long K: split it
        long long want = (1024 + tiles - 1) / tiles;
        per = (int)((chunks + want - 1) / want);
        if (per < 16) per = 16;
        zdim = (chunks + per - 1) / per;
    }
    const bool split = zdim > 1;
    igemm_dispatch<OpGemm>(d.rows, d.cols, zdim, s, [&](auto& op) {
        op.nrows = d.rows, op.ncols = d.cols, op.K = d.K;
        op.a = d.a, op.a_sn = d.a_sn, op.a_sk = d.a_sk;
        op.b = d.b, op.b_sk = d.b_sk, op.b_sm = d.b_sm;
        op.bias = d.bias, op.out = d.out, op.accd = split ? scratch : d.accd;
        op.o_sn = d.o_sn, op.o_sm = d.o_sm;
        op.mode = split ? 1 : d.mode, op.ksplit = per;
    });
    if (split)
        hipLaunchKernelGGL(k_gemm_finish, dim3((d.rows * d.cols + 255) / 256), dim3(256), 0, s, d.rows, d.cols, scratch,
                           d.bias, d.out, d.o_sn, d.o_sm);
}

}  // namespace unet
