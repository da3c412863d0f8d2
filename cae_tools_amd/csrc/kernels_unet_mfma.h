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
// 32x32 accumulators (64 AGPRs).  K is walked in chunks of 16 through a double-buffered LDS image in which k is the
// CONTIGUOUS index: W[n][16+4], X[m][16+4].  The unit of every transfer is a quad = 4 consecutive k of one row:
//   * global -> registers: one 16-byte load where memory is contiguous along k (weights, activations along x), four
//     4-byte loads off one base address where it is not (the stride-2 im2col neighbours);
//   * registers -> LDS: one ds_write_b128 per quad;
//   * LDS -> MFMA operands: one ds_read_b128 per row gives a lane its operand for FOUR k-steps (lanes 0-31 take
//     k = 8G+j, lanes 32-63 k = 8G+4+j, j = 0..3), so a chunk is 10 LDS reads per wave for 32 MFMAs.
// Rows are unpadded (64 bytes); the four quads of a row are stored XOR-swizzled (ig_swz), which makes the 16-byte
// reads of every ds_read_b128 lane group and the 16-byte writes of 8 consecutive rows bank-conflict free without
// spending LDS on padding (two 32 x 512 tiles fit one CU).  The gathers of chunk c+1 are in flight while chunk c is multiplied.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_unet.h"

namespace unet {
namespace {   // internal linkage: this header is compiled into more than one translation unit

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IG_KC = 16;   // K chunk
constexpr int IG_KS = 16;   // LDS row pitch in floats: unpadded, quads XOR-swizzled (ig_swz)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// LDS swizzle: quad g of tile row r is stored in quad slot g ^ s(r), s = {bit 2, bit 1 ^ bit 3} of r.  Found by
// exhaustive search over XOR-of-bits functions: it is conflict-free both for the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, ...: 16 rows, one quad each, 64 banks) and for ds_write_b128 by 8 consecutive rows (32 banks).
__device__ __forceinline__ int ig_swz(int r) { return (((r >> 2) & 1) | ((((r >> 1) ^ (r >> 3)) & 1) << 1)) << 2; }

// ---------------------------------------------------------------------------------------------
// policies.  Common interface:
//   rows(), cols(), k_begin(z), k_end(z)
//   Ctx / init(ctx, tid, m0, z)                      per-thread constants of the tile
//   KCtx / kprep(tid, kc, z)                         per-thread constants of one chunk
//   wmap(q, TN, row, k4) / xmap(q, TM, row, k4)      which quad of the W / X tile is quad number q
//   wquad<TN>(ctx, kctx, q, n0, kc, z) / xquad<TM>(ctx, kctx, q, m0, kc, z)
//   col(m, z) / store(col, n, v, z)
// ---------------------------------------------------------------------------------------------
template <int WM>
struct OpDown {
    static constexpr int TM = 128 * WM, MD = TM > 256 ? TM / 256 : 1;
    Geom g;
    const float* __restrict__ L;
    const float* __restrict__ w;
    const float* __restrict__ bias;
    float* __restrict__ S;
    int nsplit, ksplit;   // K slices (blockIdx.z) of ksplit chunks; nsplit > 1: S is pre-zeroed and added to atomically
    struct Ctx {
        long long xbase[MD];
        unsigned vmask[MD];
    };
    struct KCtx {
        long long plane;
    };
    __device__ void remap(int&, int&, int&) const {}
    __device__ int rows() const { return g.Cs; }
    __device__ long long cols() const { return (long long)g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(g.Cl * 16, (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx& c, int tid, long long m0, int) const {
        const int HW = g.Hs * g.Ws;
#pragma unroll
        for (int j = 0; j < MD; j++) {
            const long long m = m0 + (TM >= 256 ? tid + 256 * j : (tid & (TM - 1)));
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
    __device__ KCtx kprep(int, int kc, int) const { return KCtx{(long long)(kc >> 4) * g.Hl * g.Wl}; }
    __device__ void wmap(int q, int, int& row, int& k4) const { k4 = (q & 3) * 4, row = q >> 2; }
    __device__ void xmap(int q, int tm, int& row, int& k4) const { row = q & (tm - 1), k4 = (q / tm) * 4; }
    template <int TN>
    __device__ float4 wquad(const Ctx&, const KCtx&, int q, int n0, int kc, int) const {
        const int n = n0 + (q >> 2);
        return n < g.Cs ? ld4(w + (size_t)n * g.Cl * 16 + kc + (q & 3) * 4) : make_float4(0, 0, 0, 0);
    }
    template <int TM_, int NT>
    __device__ float4 xquad(const Ctx& c, const KCtx& k, int tid, int i, long long, int, int) const {
        // quad number tid + 256*i: row = q mod TM, ky = q / TM; for TM = 512 the pixel slot (i & 1) and ky (i >> 1) are
        // compile-time, which keeps the context arrays in registers (a run-time slot index sends them to scratch)
        const int ky = TM > 256 ? i / MD : (tid + 256 * i) / TM;
        const int slot = i % MD;
        const unsigned mk = (c.vmask[slot] >> (4 * ky)) & 15u;
        const float* p = L + c.xbase[slot] + k.plane + ky * g.Wl;
        float4 v;
        v.x = mk & 1u ? p[0] : 0.f;
        v.y = mk & 2u ? p[1] : 0.f;
        v.z = mk & 4u ? p[2] : 0.f;
        v.w = mk & 8u ? p[3] : 0.f;
        return v;
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

// weights repacked per parity: wp[z][cl][cs*4 + j*2 + i] = w[cs][cl][1-py+2j][1-px+2i],  z = py*2 + px; every layer that runs
// OpUp this step in one launch (the weights only change in the optimiser step): layer l owns elements [begin[l], begin[l+1])
struct PackSet {
    int n;
    int Cs[8], Cl[8];
    int thin[8];        // 1: the image-end ConvTranspose2d's order instead (kernels_unet_thin.h: taps of a (cs, cl, ky) as kx 1, 2, 3, 0)
    const float* w[8];
    float* wp[8];
    long long begin[9];
};
__global__ void __launch_bounds__(256) k_pack_up_weights(PackSet ps) {
    const long long total = ps.begin[ps.n];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        int l = 0;
        while (l + 1 < ps.n && e >= ps.begin[l + 1]) l++;
        const int Cs = ps.Cs[l], Cl = ps.Cl[l];
        const long long o = e - ps.begin[l];
        if (ps.thin[l]) {
            ps.wp[l][o] = ps.w[l][(o & ~3ll) + ((o + 1) & 3)];
            continue;
        }
        const int k = (int)(o % (Cs * 4));
        const long long r = o / (Cs * 4);
        const int cl = (int)(r % Cl), z = (int)(r / Cl);
        const int cs = k >> 2, j = (k >> 1) & 1, i = k & 1;
        const int ky = 1 - (z >> 1) + 2 * j, kx = 1 - (z & 1) + 2 * i;
        ps.wp[l][o] = ps.w[l][((size_t)cs * Cl + cl) * 16 + ky * 4 + kx];
    }
}

template <int WM>
struct OpUp {
    static constexpr int TM = 128 * WM, MD = TM > 256 ? TM / 256 : 1;
    Geom g;
    const float* __restrict__ S;
    const float* __restrict__ wp;   // packed weights [4][Cl][Cs*4]
    const float* __restrict__ bias;
    float* __restrict__ L;
    struct Ctx {
        long long sbase[MD];
        unsigned vmask[MD];
        int off[4];
    };
    struct KCtx {
        int dummy;
    };
    __device__ void remap(int&, int&, int&) const {}
    __device__ int rows() const { return g.Cl; }
    __device__ long long cols() const { return (long long)g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int) const { return 0; }
    __device__ int k_end(int) const { return g.Cs * 4; }
    __device__ void init(Ctx& c, int tid, long long m0, int z) const {
        const int HW = g.Hs * g.Ws;
        const int py = z >> 1, px = z & 1;
#pragma unroll
        for (int t = 0; t < 4; t++) c.off[t] = (py - (t >> 1)) * g.Ws + (px - (t & 1));
#pragma unroll
        for (int j = 0; j < MD; j++) {
            const long long m = m0 + (TM >= 256 ? tid + 256 * j : (tid & (TM - 1)));
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
    __device__ KCtx kprep(int, int, int) const { return KCtx{0}; }
    __device__ void wmap(int q, int, int& row, int& k4) const { k4 = (q & 3) * 4, row = q >> 2; }
    __device__ void xmap(int q, int tm, int& row, int& k4) const { row = q & (tm - 1), k4 = (q / tm) * 4; }
    template <int TN>
    __device__ float4 wquad(const Ctx&, const KCtx&, int q, int n0, int kc, int z) const {
        const int n = n0 + (q >> 2);
        return n < g.Cl ? ld4(wp + ((size_t)z * g.Cl + n) * g.Cs * 4 + kc + (q & 3) * 4) : make_float4(0, 0, 0, 0);
    }
    template <int TM_, int NT>
    __device__ float4 xquad(const Ctx& c, const KCtx&, int tid, int i, long long, int kc, int) const {
        // one quad = the 2x2 neighbourhood of one input channel; slot / channel offset compile-time for TM = 512
        const int cs = (kc >> 2) + (TM > 256 ? i / MD : (tid + 256 * i) / TM);
        const int slot = i % MD;
        const unsigned mk = c.vmask[slot];
        const float* p = S + c.sbase[slot] + (long long)cs * g.Hs * g.Ws;
        float4 v;
        v.x = mk & 1u ? p[c.off[0]] : 0.f;
        v.y = mk & 2u ? p[c.off[1]] : 0.f;
        v.z = mk & 4u ? p[c.off[2]] : 0.f;
        v.w = mk & 8u ? p[c.off[3]] : 0.f;
        return v;
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
    float* __restrict__ part;   // nullptr: fp64 atomics into acc; else slice z stores its tile at part[(z * Cs + n) * Cl*16 + m]
    int ksplit;   // chunks of 16 per z-slice
    struct Ctx {
        int dummy;
    };
    struct KCtx {   // the 4 consecutive pixels (one quad along k) this thread gathers in this chunk
        bool ok;
        int b, p, y, x0;
    };
    // workgroups are dealt to the 8 XCDs by linear id mod 8: the tiles of ONE K slice (they read the same slice of both maps)
    // go to one XCD, so that the slice is fetched into one L2 instead of eight
    __device__ void remap(int& bx, int& by, int& bz) const {
        const int gx = gridDim.x, gy = gridDim.y, T = gx * gy;
        if (gridDim.z & 7) return;
        const int i = bx + gx * (by + gy * bz), xcd = i & 7, j = i >> 3;
        const int t = j % T;
        bz = (j / T) * 8 + xcd;
        bx = t % gx;
        by = t / gx;
    }
    __device__ int rows() const { return g.Cs; }
    __device__ long long cols() const { return (long long)g.Cl * 16; }
    __device__ int ktotal() const { return g.B * g.Hs * g.Ws; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(ktotal(), (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx&, int, long long, int) const {}
    __device__ KCtx kprep(int tid, int kc, int) const {
        KCtx k;
        const int HW = g.Hs * g.Ws;
        const int kk = kc + (tid & 3) * 4;
        k.ok = kk < ktotal();
        k.b = kk / HW;
        k.p = kk - k.b * HW;
        k.y = k.p / g.Ws;
        k.x0 = k.p - k.y * g.Ws;
        return k;
    }
    __device__ void wmap(int q, int, int& row, int& k4) const { k4 = (q & 3) * 4, row = q >> 2; }
    __device__ void xmap(int q, int, int& row, int& k4) const { k4 = (q & 3) * 4, row = q >> 2; }
    template <int TN>
    __device__ float4 wquad(const Ctx&, const KCtx& k, int q, int n0, int, int) const {
        const int n = n0 + (q >> 2);
        return (k.ok && n < g.Cs) ? ld4(S + ((size_t)k.b * g.Cs + n) * g.Hs * g.Ws + k.p) : make_float4(0, 0, 0, 0);
    }
    template <int TM_, int NT>
    __device__ float4 xquad(const Ctx&, const KCtx& k, int tid, int i, long long m0, int, int) const {
        const int m = (int)m0 + ((tid + NT * i) >> 2);
        const int cl = m >> 4, ky = (m >> 2) & 3, kx = m & 3;
        const int Y = 2 * k.y - 1 + ky, X = 2 * k.x0 - 1 + kx;   // X, X+2, X+4, X+6
        const bool rok = k.ok && cl < g.Cl && Y >= 0 && Y < g.Hl;
        const float* p = L + (((long long)k.b * g.Cl + (rok ? cl : 0)) * g.Hl + (rok ? Y : 0)) * g.Wl + X;
        float4 v;
        v.x = (rok && X >= 0) ? p[0] : 0.f;
        v.y = rok ? p[2] : 0.f;
        v.z = rok ? p[4] : 0.f;
        v.w = (rok && X + 6 < g.Wl) ? p[6] : 0.f;
        return v;
    }
    __device__ long long col(long long m, int) const { return m < cols() ? m : -1; }
    __device__ void store(long long cb, int n, float v, int z) const {
        if (n >= g.Cs || cb < 0) return;
        if (part) {
            part[((size_t)z * g.Cs + n) * (g.Cl * 16) + cb] = v;
            return;
        }
        if (v == 0.f) return;
#if defined(IG_ABL) && IG_ABL == 4
        if (v != 12345.678f) return;
#endif
#if defined(IG_ABL) && IG_ABL == 5
        atomicAdd(reinterpret_cast<float*>(acc) + (size_t)n * g.Cl * 16 + cb, v);
        return;
#endif
        atomicAdd(&acc[(size_t)n * g.Cl * 16 + cb], (double)v);
    }
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
    struct KCtx {
        int dummy;
    };
    __device__ void remap(int&, int&, int&) const {}
    __device__ int rows() const { return nrows; }
    __device__ long long cols() const { return ncols; }
    __device__ int k_begin(int z) const { return z * ksplit * IG_KC; }
    __device__ int k_end(int z) const { return min(K, (z + 1) * ksplit * IG_KC); }
    __device__ void init(Ctx&, int, long long, int) const {}
    __device__ KCtx kprep(int, int, int) const { return KCtx{0}; }
    __device__ void wmap(int q, int tn, int& row, int& k4) const {
        if (a_sk == 1) k4 = (q & 3) * 4, row = q >> 2;
        else row = q & (tn - 1), k4 = (q / tn) * 4;
    }
    __device__ void xmap(int q, int tm, int& row, int& k4) const {
        if (b_sk == 1) k4 = (q & 3) * 4, row = q >> 2;
        else row = q & (tm - 1), k4 = (q / tm) * 4;
    }
    template <int TN>
    __device__ float4 wquad(const Ctx&, const KCtx&, int q, int n0, int kc, int) const {
        int row, k4;
        wmap(q, TN, row, k4);
        const int n = n0 + row, k = kc + k4;
        const bool ok = n < nrows;
        const float* p = a + (ok ? n * a_sn : 0) + k * a_sk;
        float4 v;
        v.x = (ok && k < K) ? p[0] : 0.f;
        v.y = (ok && k + 1 < K) ? p[a_sk] : 0.f;
        v.z = (ok && k + 2 < K) ? p[2 * a_sk] : 0.f;
        v.w = (ok && k + 3 < K) ? p[3 * a_sk] : 0.f;
        return v;
    }
    template <int TM_, int NT>
    __device__ float4 xquad(const Ctx&, const KCtx&, int tid, int i, long long m0, int kc, int) const {
        int row, k4;
        xmap(tid + NT * i, TM_, row, k4);
        const long long m = m0 + row;
        const int k = kc + k4;
        const bool ok = m < ncols;
        const float* p = b + k * b_sk + (ok ? m * b_sm : 0);
        float4 v;
        v.x = (ok && k < K) ? p[0] : 0.f;
        v.y = (ok && k + 1 < K) ? p[b_sk] : 0.f;
        v.z = (ok && k + 2 < K) ? p[2 * b_sk] : 0.f;
        v.w = (ok && k + 3 < K) ? p[3 * b_sk] : 0.f;
        return v;
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

// grid (col tiles, row tiles, z): z = output parity (up) or K slice (down, wgrad, gemm).  WN*WM waves per workgroup:
// 4 for the convolutions; 1 (a 32 x 128 tile per single-wave workgroup, K sliced finely over blockIdx.z) for the weight
// gradients of layers with a handful of channels on one side, whose whole output is smaller than one 4-wave tile.
template <int WN, int WM, class Op>
__global__ void __launch_bounds__(256) k_igemm(Op op) {
    constexpr int TN = 32 * WN, TM = 128 * WM, KC = IG_KC, KS = IG_KS, NT = 64 * WN * WM;
    constexpr int NWQ = (TN * 4 + NT - 1) / NT, NXQ = TM * 4 / NT;   // quads per thread and chunk
    constexpr int BUF = (TN + TM) * KS;
    extern __shared__ float4 lds4[];
    float* lds = reinterpret_cast<float*>(lds4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave - wn * WM;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    op.remap(bx, by, bz);
    const long long m0 = (long long)bx * TM;
    const int n0 = by * TN;
    const int z = bz;
    typename Op::Ctx ctx;
    op.init(ctx, tid, m0, z);
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float4 wq[NWQ], xq[NXQ];
    const int kb = op.k_begin(z), ke = op.k_end(z);
    auto gather = [&](int kc) {
        const typename Op::KCtx kx = op.kprep(tid, kc, z);
#pragma unroll
        for (int i = 0; i < NWQ; i++) {
            const int q = tid + NT * i;
#if defined(IG_ABL) && (IG_ABL == 2 || IG_ABL >= 6)
            wq[i] = make_float4((float)q, 1.f, 2.f, 3.f);
#else
            wq[i] = (TN * 4 % NT == 0 || q < TN * 4) ? op.template wquad<TN>(ctx, kx, q, n0, kc, z) : make_float4(0, 0, 0, 0);
#endif
        }
#pragma unroll
        for (int i = 0; i < NXQ; i++) {
#if defined(IG_ABL) && (IG_ABL == 1 || IG_ABL == 2 || IG_ABL >= 6)
            xq[i] = make_float4((float)tid, (float)i, (float)kc, 1.f);
#else
            xq[i] = op.template xquad<TM, NT>(ctx, kx, tid, i, m0, kc, z);
#endif
        }
    };
    auto commit = [&](float* Wb) {
        float* Xb = Wb + TN * KS;
#pragma unroll
        for (int i = 0; i < NWQ; i++) {
            const int q = tid + NT * i;
            if (TN * 4 % NT == 0 || q < TN * 4) {
                int row, k4;
                op.wmap(q, TN, row, k4);
                *reinterpret_cast<float4*>(Wb + row * KS + (k4 ^ ig_swz(row))) = wq[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NXQ; i++) {
            int row, k4;
            op.xmap(tid + NT * i, TM, row, k4);
            *reinterpret_cast<float4*>(Xb + row * KS + (k4 ^ ig_swz(row))) = xq[i];
        }
    };
    if (kb < ke) {
        gather(kb);
        commit(lds);
    }
    __syncthreads();
    int buf = 0;
    const int swz = ig_swz(lane & 31);        // tile row bases are multiples of 32
    const int arow = (wn * 32 + (lane & 31)) * KS, brow = (wm * 128 + (lane & 31)) * KS;
    const int h4 = 4 * (lane >> 5);
    for (int kc = kb; kc < ke; kc += KC) {
        const bool more = kc + KC < ke;
        if (more) gather(kc + KC);
        const float* Wb = lds + buf * BUF;
        const float* Xb = Wb + TN * KS;
#pragma unroll
        for (int G = 0; G < KC / 8; G++) {
            const int ko = (8 * G + h4) ^ swz;
            const float4 a4 = ld4(Wb + arow + ko);
            float4 b4[4];
#pragma unroll
            for (int c = 0; c < 4; c++) b4[c] = ld4(Xb + brow + c * 32 * KS + ko);
#if defined(IG_ABL) && IG_ABL == 3
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][0] += a4.x * b4[c].x + a4.y * b4[c].y + a4.z * b4[c].z + a4.w * b4[c].w;
            continue;
#endif
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[c].x, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[c].y, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[c].z, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[c].w, acc[c], 0, 0, 0);
        }
#if defined(IG_ABL) && IG_ABL == 6
        continue;
#endif
#if defined(IG_ABL) && IG_ABL == 7
        __syncthreads();
        buf ^= 1;
        continue;
#endif
        if (more) commit(lds + (buf ^ 1) * BUF);
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
// the 16-byte loads of S along x need rows of S that are multiples of 4 pixels
inline bool mfma_wgrad_eligible(const Geom& g) { return mfma_geom(g) && g.Cs >= 8 && g.Ws % 4 == 0; }

template <int WN, int WM, class Op>
inline void igemm_launch(const Op& op, dim3 grid, hipStream_t s) {
    constexpr size_t bytes = (size_t)2 * (32 * WN + 128 * WM) * IG_KS * sizeof(float);
    static bool raised = false;   // above the 64 KiB a kernel gets without asking
    if (bytes > 65536 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_igemm<WN, WM, Op>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        raised = true;
    }
    hipLaunchKernelGGL((k_igemm<WN, WM, Op>), grid, dim3(64 * WN * WM), bytes, s, op);
}

// tile shape by row count: 32 x 512, 64 x 256 or 128 x 128
template <template <int> class OpT, class Fill>
inline void igemm_dispatch(int rows, long long cols, int zdim, hipStream_t s, Fill fill) {
    if (rows <= 32) {
        OpT<4> op;
        fill(op);
        igemm_launch<1, 4>(op, dim3((unsigned)((cols + 511) / 512), (rows + 31) / 32, zdim), s);
    } else if (rows <= 64) {
        OpT<2> op;
        fill(op);
        igemm_launch<2, 2>(op, dim3((unsigned)((cols + 255) / 256), (rows + 63) / 64, zdim), s);
    } else {
        OpT<1> op;
        fill(op);
        igemm_launch<4, 1>(op, dim3((unsigned)((cols + 127) / 128), (rows + 127) / 128, zdim), s);
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

// wp: the layer's weights repacked by k_pack_up_weights (Cs*Cl*16 floats)
inline void mfma_up_launch(const Geom& g, const float* S, const float* wp, const float* bias, float* L, hipStream_t s) {
    igemm_dispatch<OpUp>(g.Cl, (long long)g.B * g.Hs * g.Ws, 4, s, [&](auto& op) {
        op.g = g, op.S = S, op.wp = wp, op.bias = bias, op.L = L;
    });
}

// acc[e] += sum over the K slices of part[z * E + e]: fp64, slices in index order (no atomics: the result does not depend on
// timing).  grid-stride over E
__global__ void __launch_bounds__(256) k_wgrad_fold(const float* __restrict__ part, int slices, long long E, double* __restrict__ acc) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        double s = 0.0;
        int z = 0;
        for (; z + 4 <= slices; z += 4) {       // four loads in flight
            const float p0 = part[(size_t)z * E + e], p1 = part[(size_t)(z + 1) * E + e];
            const float p2 = part[(size_t)(z + 2) * E + e], p3 = part[(size_t)(z + 3) * E + e];
            s += (double)p0;
            s += (double)p1;
            s += (double)p2;
            s += (double)p3;
        }
        for (; z < slices; z++) s += (double)part[(size_t)z * E + e];
        acc[e] += s;
    }
}

// K slices (blockIdx.z) of the 4-wave tile path for this geometry, and the bytes of partial tiles they would write
inline int mfma_wgrad_slices(const Geom& g) {
    const int rows = g.Cs;
    const long long cols = (long long)g.Cl * 16;
    const int TN = rows <= 32 ? 32 : rows <= 64 ? 64 : 128, TM = rows <= 32 ? 512 : rows <= 64 ? 256 : 128;
    const long long tiles = ((rows + TN - 1) / TN) * ((cols + TM - 1) / TM);
    const int chunks = (g.B * g.Hs * g.Ws + IG_KC - 1) / IG_KC;
#ifndef IG_WGRAD_BLOCKS
#define IG_WGRAD_BLOCKS 512
#endif
    long long want = (IG_WGRAD_BLOCKS + tiles - 1) / tiles;   // workgroups to aim at
    if (want < 1) want = 1;
    int per = (int)((chunks + want - 1) / want);
    if (per < 8) per = 8;                                 // at least 128 pixels per slice
    return (chunks + per - 1) / per;
}
inline size_t mfma_wgrad_part_bytes(const Geom& g) {
#ifndef IG_WG1
#define IG_WG1 128
#endif
    if ((long long)g.Cl * 16 <= IG_WG1) return 0;         // the single-wave path keeps its atomics
    return (size_t)mfma_wgrad_slices(g) * g.Cs * g.Cl * 16 * sizeof(float);
}

// part: room for mfma_wgrad_part_bytes(g) (the slices' tiles, folded in order by k_wgrad_fold) or nullptr (fp64 atomics)
inline void mfma_wgrad_launch(const Geom& g, const float* S, const float* L, double* acc, float* part, hipStream_t s) {
    const int rows = g.Cs;
    const long long cols = (long long)g.Cl * 16;
    if (cols <= IG_WG1) {   // <= 8 channels on the big side: single-wave 32 x 128 tiles, the parallelism comes from K
        const int chunks = (g.B * g.Hs * g.Ws + IG_KC - 1) / IG_KC;
        const int rt = (rows + 31) / 32, ct = (int)((cols + 127) / 128);
        int per = (int)(((long long)chunks * rt * ct + 2047) / 2048);                // ~2048 single-wave workgroups (8 per CU)
        if (per < 8) per = 8;
        OpWgrad<1> op;
        op.g = g, op.S = S, op.L = L, op.acc = acc, op.part = nullptr, op.ksplit = per;
        igemm_launch<1, 1>(op, dim3(ct, rt, (chunks + per - 1) / per), s);
        return;
    }
    const int chunks = (g.B * g.Hs * g.Ws + IG_KC - 1) / IG_KC;
    const int zdim = mfma_wgrad_slices(g);
    const int per = (chunks + zdim - 1) / zdim;
    igemm_dispatch<OpWgrad>(rows, cols, zdim, s, [&](auto& op) {
        op.g = g, op.S = S, op.L = L, op.acc = acc, op.part = part, op.ksplit = per;
    });
    if (part) {
        const long long E = (long long)rows * cols;
        hipLaunchKernelGGL(k_wgrad_fold, dim3((unsigned)std::min<long long>((E + 255) / 256, 4096)), dim3(256), 0, s, part, zdim, E, acc);
    }
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

}  // namespace
}  // namespace unet
