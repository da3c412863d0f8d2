// kernels_rows.h — row-streaming backward of the thin stride-2 ConvTranspose2d layers in the middle of the decoder (8->4 and
// 4->2 channels at the benchmark geometry; decoder.py:44-48 under autograd), successor of k_s2_bwd2 / k_s2_bwd_split
// (kernels_s2.h), which gave every input pixel a thread that fetched its own kh x kw patch of the gradient map (2.25 x
// overlapping reads per element, two dependent round trips per tile, 0.7-1.5 TB/s).
//
// Same quad algebra as kernels_last.h: an output quad (m, n) = outputs (2m+py, 2n+px) meets the inputs a[m-j][n-i], j,i in
// {0,1}, through the taps (py+2j, px+2i) < (KH, KW); with gy = BatchNorm-backward(g, y) of this layer's output,
//     dW[ci][co][py+2j][px+2i] += a[ci][m-j][n-i] * gy[co][py][px]
//     P[ci][j][i]               = sum_{co,py,px} gy[co][py][px] * W[ci][co][py+2j][px+2i]   -> input pixel (m-j, n-i)
// A lane owns ONE quad column; a wave walks down a band of HB quad rows: every gradient / output element is read exactly once
// (8-byte loads, a row of lanes = one contiguous run), the previous input row and the j = 0 shares travel in registers, the
// i = 1 shares go to the left neighbour by a DPP wave shift.  Maps up to 32 quads wide put two images side by side in a wave.
// The four waves of a workgroup are either four consecutive bands of one image (CS = 1: the first row's j = 1 shares of a
// band go to the band above through LDS, one barrier per workgroup; only the last band re-reads one row of the next
// workgroup's) or the four input-channel groups of one band (CS = 4: 8 input channels x 4 x 9 weight-gradient accumulators do
// not fit one lane).  Loads run D rows ahead of their use.
#pragma once
#include "kernels_last.h"

namespace cae {

struct S2Rows {
    int B, H, W, OH, OW;
    int QH;                  // quad rows of the output
    int bands;               // bands of HB quad rows per image
    int groups;              // image groups (IMGS images each)
    const float* g;          // (B, COUT, OH, OW) masked upstream gradient
    const float* yout;       // (B, COUT, OH, OW) raw forward output of this layer
    BnDesc bn_out;           // BN_BWD of this layer's BatchNorm
    const float* ain;        // (B, CIN, H, W) raw output of the producer
    BnDesc bn_in;            // BN_SAVED of the producer's BatchNorm
    const float* w;          // (CIN, COUT, KH, KW)
    float* gin;              // (B, CIN, H, W) masked gradient for the producer
    double* stats_in;        // producer's [kStatShards][CIN][4] sums (slots 2, 3)
    double* wacc;            // sharded [kStatShards][wacc_stride]
    int wacc_stride;
    BnGradOut bg;            // this layer's BatchNorm parameter gradients (published by workgroup 0)
    long long* dbg;          // diagnostics (tools/last_phases.py rows): 8 wall-clock stamps per workgroup (first 384), or nullptr
};

// CT input channels per wave (CS = CIN / CT channel groups = waves sharing a band), IMGS images side by side in a wave
template <int CIN, int CT, int COUT, int KH, int KW, int HB, int IMGS, int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_s2_bwd_rows(S2Rows a) {
    kernarg_warm<sizeof(S2Rows)>();
    constexpr int CS = CIN / CT;             // waves that share a band (channel groups)
    constexpr int NB = 4 / CS;               // bands per workgroup
    constexpr int LW = 64 / IMGS;            // lanes (quad columns) per image
    constexpr int NACC = CT * COUT * KH * KW;
    constexpr int NRED = NACC + 2 * CT;
    constexpr int NR = HB + 1;               // rows a wave may compute: its band + the first row of the next band
    static_assert(CS * NB == 4 && CS * CT == CIN, "four waves");
    static_assert(KH == 3 && KW == 3, "3x3 stride-2 kernels");
    static_assert(CIN <= 8 && COUT <= 8, "one lane per (channel, shard) in the BatchNorm prologue");
    __shared__ float redf[4 * NRED];
    __shared__ __attribute__((aligned(16))) float wscr[4 * kWsumScratch];
    __shared__ float xch[NB > 1 ? 4 * CT * 64 : 1];
    // weights [ci][co][12]: nine taps padded to three 16-byte reads (72 of them do not fit the scalar registers next to the
    // BatchNorm constants; read back as wave-wide broadcasts where they are used)
    __shared__ __attribute__((aligned(16))) float wl[CIN * COUT * 12];

#define RW_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.x < 384) a.dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    RW_STAMP(0);
    const int cgrp = wv % CS, bsel = wv / CS;            // channel group, band inside the workgroup
    const int c0 = cgrp * CT;
    const int sub = lane / LW, n = lane - sub * LW;      // image inside the wave, quad column = pixel column
    const unsigned HW = a.H * a.W, OHW = a.OH * a.OW;

    const int wg_bands = (a.bands + NB - 1) / NB;        // band groups per image group
    const int grp = blockIdx.x / wg_bands, bg0 = blockIdx.x - grp * wg_bands;
    const int band = bg0 * NB + bsel;
    const bool band_ok = band < a.bands;                 // uniform per wave
    const int m0 = band * HB;
    const int b = grp * IMGS + sub;
    const bool img_ok = b < a.B;
    const int bc = img_ok ? b : a.B - 1;
    // the last band of a workgroup (or of the image) completes its last pixel row itself: one more quad row, not counted
    const bool has_below = NB > 1 && bsel + 1 < NB && band + 1 < a.bands;   // the band below is in this workgroup
    const int nrows = has_below ? HB : NR;

    // ---- small reads first (the memory counter retires in issue order: kernels_last.h), the band's rows behind them, the
    // arithmetic on the small reads behind that.  BatchNorm of this layer's output (backward form {mean, k1, k2, k3}) and of the
    // producer (saved statistics): lanes 8 c + shard read the shards of channel c (one 16-byte load), three DPP steps below.
    float4 ko[COUT], ki[CT];
    double sa = 0.0, sb = 0.0;
    if (lane < 8 * COUT) {
        const double2 t = *reinterpret_cast<const double2*>(a.bn_out.stats + ((size_t)(lane & 7) * COUT + (lane >> 3)) * 4 + 2);
        sa = t.x;
        sb = t.y;
    }
    float mean_o[COUT], istd_o[COUT], gam_o[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) {
        mean_o[co] = a.bn_out.saved[2 * co];
        istd_o[co] = a.bn_out.saved[2 * co + 1];
        gam_o[co] = a.bn_out.gamma[co];
    }
    float mean_i[CT], istd_i[CT], gam_i[CT], bet_i[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        mean_i[c] = a.bn_in.saved[2 * (c0 + c)];
        istd_i[c] = a.bn_in.saved[2 * (c0 + c) + 1];
        gam_i[c] = a.bn_in.gamma[c0 + c];
        bet_i[c] = a.bn_in.beta[c0 + c];
    }
    constexpr int NWL = (CIN * COUT * 12 + 255) / 256;      // weight-image elements a thread stages
    float wreg[NWL];
#pragma unroll
    for (int u = 0; u < NWL; u++) {
        const int i = min(tid + 256 * u, CIN * COUT * 12 - 1), cc = i / 12, t = i - cc * 12;
        wreg[u] = a.w[cc * 9 + min(t, 8)];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- row loads (clamped addresses; validity by select).  rg[r]: gradient and raw output of quad row m0 + r, two output
    // rows x two columns per output channel; rin[r]: the producer's raw output, input row m0 - 1 + r, this lane's column.
    float2 rg[NR][COUT][2], ry[NR][COUT][2];
    float rin[NR + 1][CT];
    const float* gb = a.g + (size_t)bc * COUT * OHW;
    const float* yb = a.yout + (size_t)bc * COUT * OHW;
    const float* ib = a.ain + ((size_t)bc * CIN + c0) * HW;
    // column pair (2n, 2n+1); where the row ends on column 2n the pair (2n-1, 2n) is read instead and shifted on use
    const bool pair_ok = 2 * n + 1 < a.OW;
    const int oxc = pair_ok ? 2 * n : min(max(2 * n - 1, 0), a.OW - 2);
    const int xc = min(n, a.W - 1);
    const bool last_band = band == a.bands - 1;
    auto load_in = [&](auto R) {
        constexpr int r = decltype(R)::value;
        const unsigned ro = (unsigned)min(max(m0 - 1 + r, 0), a.H - 1) * a.W;
#pragma unroll
        for (int c = 0; c < CT; c++) rin[r][c] = ib[c * HW + ro + xc];
    };
    auto load_g = [&](auto R) {
        constexpr int r = decltype(R)::value;
        if (r >= nrows) return;          // uniform: the band below reads that row
#pragma unroll
        for (int co = 0; co < COUT; co++)
#pragma unroll
            for (int py = 0; py < 2; py++) {
                const unsigned off = co * OHW + (unsigned)min(2 * (m0 + r) + py, a.OH - 1) * a.OW + oxc;
                rg[r][co][py] = *reinterpret_cast<const float2*>(gb + off);
                ry[r][co][py] = *reinterpret_cast<const float2*>(yb + off);
            }
    };
    load_in(std::integral_constant<int, 0>{});
    static_for<(D < NR ? D : NR)>([&](auto R) {
        constexpr int r = decltype(R)::value;
        load_in(std::integral_constant<int, r + 1>{});
        load_g(R);
    });
    __builtin_amdgcn_sched_barrier(0);
    RW_STAMP(1);
    {   // the constants, now that the rows are on their way
        sa += dpp_d<0xB1>(sa); sb += dpp_d<0xB1>(sb);
        sa += dpp_d<0x4E>(sa); sb += dpp_d<0x4E>(sb);
        sa += dpp_d<0x141>(sa); sb += dpp_d<0x141>(sb);
        const long long ba = __builtin_bit_cast(long long, sa), bb = __builtin_bit_cast(long long, sb);
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            const int lo1 = __builtin_amdgcn_readlane((int)ba, 8 * co), hi1 = __builtin_amdgcn_readlane((int)(ba >> 32), 8 * co);
            const int lo2 = __builtin_amdgcn_readlane((int)bb, 8 * co), hi2 = __builtin_amdgcn_readlane((int)(bb >> 32), 8 * co);
            const double dbeta = __builtin_bit_cast(double, ((long long)hi1 << 32) | (unsigned)lo1);
            const double dgamma = __builtin_bit_cast(double, ((long long)hi2 << 32) | (unsigned)lo2);
            const float mean = uniform_f(mean_o[co]), invstd = uniform_f(istd_o[co]);
            const float scale = uniform_f(gam_o[co]) * invstd;
            // bn_consts, BN_BWD: gy = k1 g - k2 - (y - mean) k3
            ko[co] = make_float4(mean, scale, (float)((double)scale * dbeta * a.bn_out.inv_count),
                                 (float)((double)scale * (double)invstd * dgamma * a.bn_out.inv_count));
            if (blockIdx.x == 0 && tid == 0 && a.bg.stats) {   // this layer's BatchNorm parameter gradients
                a.bg.beta_acc[co] = dbeta * a.bg.scale;
                a.bg.gamma_acc[co] = dgamma * a.bg.scale;
            }
        }
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const float invstd = uniform_f(istd_i[c]);
            ki[c] = make_float4(uniform_f(mean_i[c]), uniform_f(gam_i[c]) * invstd, uniform_f(bet_i[c]), invstd);
        }
#pragma unroll
        for (int u = 0; u < NWL; u++) {
            const int i = tid + 256 * u;
            if (i < CIN * COUT * 12) wl[i] = (i % 12) < 9 ? wreg[u] : 0.f;
        }
    }
    __syncthreads();         // wl is complete
    RW_STAMP(2);

    float dw[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) dw[i] = 0.f;
    float d1[CT], d2[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) d1[c] = d2[c] = 0.f;

    // BatchNorm + ReLU of input row r: {column n-1, n}; zero outside the map
    auto activate = [&](int r, int c, float (&act)[2]) {
        const int y = m0 - 1 + r;
        const float v = fmaxf(0.f, fmaf(rin[r][c] - ki[c].x, ki[c].y, ki[c].z));
        act[1] = (y >= 0 && y < a.H && n < a.W) ? v : 0.f;
        const float left = from_left(act[1], 0.f);
        act[0] = n == 0 ? 0.f : left;                    // the first column of an image has no left neighbour (IMGS = 2: lane 32)
    };
    float actP[CT][2], carry[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        activate(0, c, actP[c]);
        carry[c] = 0.f;
    }
    float xsave[CT];        // j = 1 shares of this band's first row: they complete the last pixel row of the band above
#pragma unroll
    for (int c = 0; c < CT; c++) xsave[c] = 0.f;

    static_for<NR>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int m = m0 + r;
        if constexpr (r + D < NR) {      // run the loads D rows ahead
            load_in(std::integral_constant<int, r + D + 1>{});
            load_g(std::integral_constant<int, r + D>{});
        }
        if (r < nrows && m < a.QH && band_ok) {
            const bool own_row = r < HB || last_band;    // the extra row is counted by the band that owns it (the image's last: here)
            float actN[CT][2];
#pragma unroll
            for (int c = 0; c < CT; c++) activate(r + 1, c, actN[c]);
            // BatchNorm-backward of this quad's outputs; zero outside the map
            float gy[COUT][2][2];
#pragma unroll
            for (int co = 0; co < COUT; co++)
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    const bool rowok = img_ok && 2 * m + py < a.OH;
                    const float g0 = pair_ok ? rg[r][co][py].x : rg[r][co][py].y, g1 = rg[r][co][py].y;
                    const float y0 = pair_ok ? ry[r][co][py].x : ry[r][co][py].y, y1 = ry[r][co][py].y;
                    const float v0 = ko[co].y * g0 - ko[co].z - (y0 - ko[co].x) * ko[co].w;
                    const float v1 = ko[co].y * g1 - ko[co].z - (y1 - ko[co].x) * ko[co].w;
                    gy[co][py][0] = (rowok && 2 * n < a.OW) ? v0 : 0.f;
                    gy[co][py][1] = (rowok && 2 * n + 1 < a.OW) ? v1 : 0.f;
                }
            // (an offset the compiler cannot see through: otherwise the weight reads of all rows are merged and the 72 weights
            // of a wave are kept in vector registers for the whole walk)
            int wofs = 0;
            asm volatile("" : "+v"(wofs));
            float P[CT][2][2];
#pragma unroll
            for (int c = 0; c < CT; c++) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 2; i++) P[c][j][i] = 0.f;
#pragma unroll
                for (int co = 0; co < COUT; co++) {
                    float wc[12];
                    {
                        const float4* wq = reinterpret_cast<const float4*>(&wl[((c0 + c) * COUT + co) * 12 + wofs]);
                        const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
                        wc[0] = w0.x; wc[1] = w0.y; wc[2] = w0.z; wc[3] = w0.w; wc[4] = w1.x; wc[5] = w1.y; wc[6] = w1.z; wc[7] = w1.w;
                        wc[8] = w2.x; wc[9] = w2.y; wc[10] = w2.z; wc[11] = w2.w;
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int i = 0; i < 2; i++) {
                            const float av = j ? actP[c][1 - i] : actN[c][1 - i];
#pragma unroll
                            for (int py = 0; py < 2; py++)
#pragma unroll
                                for (int px = 0; px < 2; px++)
                                    if (py + 2 * j < KH && px + 2 * i < KW) {
                                        const int t = (py + 2 * j) * KW + px + 2 * i;
                                        P[c][j][i] = fmaf(gy[co][py][px], wc[t], P[c][j][i]);
                                        if (own_row) dw[(c * COUT + co) * KH * KW + t] = fmaf(av, gy[co][py][px], dw[(c * COUT + co) * KH * KW + t]);
                                    }
                        }
                }
            }
            // pixel row y = m - 1, column n: carry (j = 0 shares of row m-1) + this row's j = 1 shares; i = 1 shares come
            // from the right-hand neighbour (the last lane of an image owns no pixel)
            const int y = m - 1;
#pragma unroll
            for (int c = 0; c < CT; c++) {
                const float rj1 = from_right(P[c][1][1], 0.f), rj0 = from_right(P[c][0][1], 0.f);
                const float tj1 = P[c][1][0] + (n == LW - 1 ? 0.f : rj1);
                const float tj0 = P[c][0][0] + (n == LW - 1 ? 0.f : rj0);
                if (r == 0) xsave[c] = tj1;
                if (r >= 1 && y < a.H) {
                    float gv = carry[c] + tj1;
                    const float e = rin[r][c] - ki[c].x;
                    gv = fmaf(e, ki[c].y, ki[c].z) > 0.f ? gv : 0.f;
                    const bool pok = img_ok && n < a.W;
                    const float s = pok ? gv : 0.f;
                    d1[c] += s;
                    d2[c] = fmaf(s, e * ki[c].w, d2[c]);
                    if (pok) a.gin[((size_t)b * CIN + c0 + c) * HW + (unsigned)y * a.W + n] = gv;
                }
                carry[c] = tj0;
                actP[c][0] = actN[c][0];
                actP[c][1] = actN[c][1];
            }
        }
        if constexpr (r == 0) RW_STAMP(3);
        if constexpr (r == 0 && NB > 1) {
            // bands of one workgroup: the first row's j = 1 shares go up one band through LDS
#pragma unroll
            for (int c = 0; c < CT; c++) xch[(wv * CT + c) * 64 + lane] = xsave[c];
            __syncthreads();
        }
    });
    if constexpr (NB > 1) {
        // a band with a neighbour below computed HB rows: its last pixel row (m0 + HB - 1) waits for that neighbour's first-row shares
        if (has_below && band_ok) {
            const int y = m0 + HB - 1;
            if (y < a.H) {
#pragma unroll
                for (int c = 0; c < CT; c++) {
                    float gv = carry[c] + xch[((wv + CS) * CT + c) * 64 + lane];
                    const float e = rin[HB][c] - ki[c].x;
                    gv = fmaf(e, ki[c].y, ki[c].z) > 0.f ? gv : 0.f;
                    const bool pok = img_ok && n < a.W;
                    const float s = pok ? gv : 0.f;
                    d1[c] += s;
                    d2[c] = fmaf(s, e * ki[c].w, d2[c]);
                    if (pok) a.gin[((size_t)b * CIN + c0 + c) * HW + (unsigned)y * a.W + n] = gv;
                }
            }
        }
    }

    RW_STAMP(4);
    // ---- reductions: wave (through LDS), workgroup (LDS; channel groups keep their own sums), then one fp64 atomic per value
    {
        float red[NRED];
#pragma unroll
        for (int i = 0; i < NACC; i++) red[i] = dw[i];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            red[NACC + 2 * c] = d1[c];
            red[NACC + 2 * c + 1] = d2[c];
        }
        wave_sums_lds<NRED>(red, wscr + wv * kWsumScratch, redf + wv * NRED, lane);
    }
    __syncthreads();
    const int shard = blockIdx.x & (kStatShards - 1);
    for (int i = tid; i < CS * NRED; i += 256) {
        const int gsel = i / NRED, j = i - gsel * NRED;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NB; q++) s += (double)redf[(q * CS + gsel) * NRED + j];
        if (j < NACC) {
            acc_add<ACC_GRAD>(&a.wacc[(size_t)shard * a.wacc_stride + (size_t)gsel * NACC + j], s);
        } else {
            const int jj = j - NACC;
            acc_add<ACC_GRAD>(&a.stats_in[((size_t)shard * CIN + gsel * CT + (jj >> 1)) * 4 + 2 + (jj & 1)], s);
        }
    }
    RW_STAMP(5);
#undef RW_STAMP
}

}  // namespace cae

namespace cae {

// =================================================================================================
// Forward of the same layers, same walk: a lane per quad column, a wave per band of HB quad rows.  The forward needs no
// neighbour's results, only input row m0 - 1 again (the inputs are the small side), so the four waves of a workgroup are four
// independent bands.  Raw output + the BatchNorm sums of this layer (train) or raw output only (eval: stats == nullptr).
// =================================================================================================
struct S2FwdRows {
    int B, H, W, OH, OW;
    int QH;
    int bands, groups;
    const float* in;         // (B, CIN, H, W) raw output of the producer
    BnDesc bn_in;            // BN_BATCH / BN_RUNNING of the producer
    const float* w;          // (CIN, COUT, 3, 3)
    const float* bias;       // (COUT)
    float* out;              // (B, COUT, OH, OW) raw output
    double* stats;           // [kStatShards][COUT][4] slots 0, 1, or nullptr
};

template <int CIN, int COUT, int HB, int IMGS>
__global__ void __launch_bounds__(256) k_s2_fwd_rows(S2FwdRows a) {
    kernarg_warm<sizeof(S2FwdRows)>();
    constexpr int KH = 3, KW = 3;
    constexpr int LW = 64 / IMGS;
    constexpr int NRED = 2 * COUT;
    static_assert(CIN <= 8, "one lane per (channel, shard) in the BatchNorm prologue");
    __shared__ float redf[4 * NRED];
    __shared__ __attribute__((aligned(16))) float wscr[4 * kWsumScratch];
    __shared__ __attribute__((aligned(16))) float wl[CIN * COUT * 12];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane / LW, n = lane - sub * LW;
    const unsigned HW = a.H * a.W, OHW = a.OH * a.OW;
    const int wg_bands = (a.bands + 3) / 4;
    const int grp = blockIdx.x / wg_bands, bg0 = blockIdx.x - grp * wg_bands;
    const int band = bg0 * 4 + wv;
    const bool band_ok = band < a.bands;
    const int m0 = band * HB;
    const int b = grp * IMGS + sub;
    const bool img_ok = b < a.B;
    const int bc = img_ok ? b : a.B - 1;

    // ---- small reads first, the band's input rows behind them, the arithmetic on the small reads behind that (kernels_last.h).
    // BatchNorm of the producer: batch statistics (train; workgroup 0 saves them and moves the running statistics) or running
    // statistics (eval).  Lanes 8 c + shard read the shards of channel c; three DPP steps; lane 8 c hands them out.
    float4 ki[CIN];
    const bool batch = a.bn_in.mode == BN_BATCH;
    double sa = 0.0, sb = 0.0;
    if (batch && lane < 8 * CIN) {
        const double2 t = *reinterpret_cast<const double2*>(a.bn_in.stats + ((size_t)(lane & 7) * CIN + (lane >> 3)) * 4);
        sa = t.x;
        sb = t.y;
    }
    float gam[CIN], bet[CIN], rmn[CIN], rvr[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c++) {
        gam[c] = a.bn_in.gamma[c];
        bet[c] = a.bn_in.beta[c];
        rmn[c] = a.bn_in.rmean[c];
        rvr[c] = a.bn_in.rvar[c];
    }
    constexpr int NWL = (CIN * COUT * 12 + 255) / 256;
    float wreg[NWL];
#pragma unroll
    for (int u = 0; u < NWL; u++) {
        const int i = min(tid + 256 * u, CIN * COUT * 12 - 1), cc = i / 12, t = i - cc * 12;
        wreg[u] = a.w[cc * 9 + min(t, 8)];
    }
    float bk[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) bk[co] = a.bias[co];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- every input row of the band up front: rows m0 - 1 .. m0 + HB - 1, this lane's column
    float rin[HB + 1][CIN];
    {
        const float* ib = a.in + (size_t)bc * CIN * HW;
        const int xc = min(n, a.W - 1);
#pragma unroll
        for (int r = 0; r <= HB; r++) {
            const unsigned ro = (unsigned)min(max(m0 - 1 + r, 0), a.H - 1) * a.W;
#pragma unroll
            for (int c = 0; c < CIN; c++) rin[r][c] = ib[c * HW + ro + xc];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        sa += dpp_d<0xB1>(sa); sb += dpp_d<0xB1>(sb);
        sa += dpp_d<0x4E>(sa); sb += dpp_d<0x4E>(sb);
        sa += dpp_d<0x141>(sa); sb += dpp_d<0x141>(sb);
        const long long ba = __builtin_bit_cast(long long, sa), bb = __builtin_bit_cast(long long, sb);
#pragma unroll
        for (int c = 0; c < CIN; c++) {
            float mean, invstd;
            if (batch) {
                const int lo1 = __builtin_amdgcn_readlane((int)ba, 8 * c), hi1 = __builtin_amdgcn_readlane((int)(ba >> 32), 8 * c);
                const int lo2 = __builtin_amdgcn_readlane((int)bb, 8 * c), hi2 = __builtin_amdgcn_readlane((int)(bb >> 32), 8 * c);
                const double s1 = __builtin_bit_cast(double, ((long long)hi1 << 32) | (unsigned)lo1);
                const double s2 = __builtin_bit_cast(double, ((long long)hi2 << 32) | (unsigned)lo2);
                const double mu = s1 * a.bn_in.inv_count;
                double var = s2 * a.bn_in.inv_count - mu * mu;
                var = var < 0.0 ? 0.0 : var;
                mean = (float)mu;
                invstd = 1.0f / sqrtf((float)(var + (double)a.bn_in.eps));
                if (blockIdx.x == 0 && tid == 0 && a.bn_in.update) {
                    a.bn_in.saved[2 * c] = mean;
                    a.bn_in.saved[2 * c + 1] = invstd;
                    a.bn_in.rmean[c] = (1.f - a.bn_in.momentum) * rmn[c] + a.bn_in.momentum * mean;
                    a.bn_in.rvar[c] = (1.f - a.bn_in.momentum) * rvr[c] + a.bn_in.momentum * (float)(var * a.bn_in.unbias);
                }
            } else {
                mean = uniform_f(rmn[c]);
                invstd = 1.0f / sqrtf(uniform_f(rvr[c]) + a.bn_in.eps);
            }
            ki[c] = make_float4(mean, uniform_f(gam[c]) * invstd, uniform_f(bet[c]), invstd);
        }
#pragma unroll
        for (int u = 0; u < NWL; u++) {
            const int i = tid + 256 * u;
            if (i < CIN * COUT * 12) wl[i] = (i % 12) < 9 ? wreg[u] : 0.f;
        }
#pragma unroll
        for (int co = 0; co < COUT; co++) bk[co] = uniform_f(bk[co]);
    }
    __syncthreads();         // wl is complete

    auto activate = [&](int r, int c, float (&act)[2]) {
        const int y = m0 - 1 + r;
        const float v = fmaxf(0.f, fmaf(rin[r][c] - ki[c].x, ki[c].y, ki[c].z));
        act[1] = (y >= 0 && y < a.H && n < a.W) ? v : 0.f;
        const float left = from_left(act[1], 0.f);
        act[0] = n == 0 ? 0.f : left;
    };
    float actP[CIN][2];
#pragma unroll
    for (int c = 0; c < CIN; c++) activate(0, c, actP[c]);
    float s1[COUT], s2[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) s1[co] = s2[co] = 0.f;
    const bool pair_ok = 2 * n + 1 < a.OW;
    float* ob = a.out + (size_t)bc * COUT * OHW;

    static_for<HB>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int m = m0 + r;
        if (m < a.QH && band_ok) {
            float actN[CIN][2];
#pragma unroll
            for (int c = 0; c < CIN; c++) activate(r + 1, c, actN[c]);
            int wofs = 0;
            asm volatile("" : "+v"(wofs));       // keeps the weight reads of different rows apart (kernels_rows.h, backward)
#pragma unroll
            for (int co = 0; co < COUT; co++) {
                float acc[2][2] = {{bk[co], bk[co]}, {bk[co], bk[co]}};
#pragma unroll
                for (int c = 0; c < CIN; c++) {
                    float wc[12];
                    {
                        const float4* wq = reinterpret_cast<const float4*>(&wl[(c * COUT + co) * 12 + wofs]);
                        const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
                        wc[0] = w0.x; wc[1] = w0.y; wc[2] = w0.z; wc[3] = w0.w; wc[4] = w1.x; wc[5] = w1.y; wc[6] = w1.z; wc[7] = w1.w;
                        wc[8] = w2.x; wc[9] = w2.y; wc[10] = w2.z; wc[11] = w2.w;
                    }
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int i = 0; i < 2; i++) {
                            const float av = j ? actP[c][1 - i] : actN[c][1 - i];
#pragma unroll
                            for (int py = 0; py < 2; py++)
#pragma unroll
                                for (int px = 0; px < 2; px++)
                                    if (py + 2 * j < KH && px + 2 * i < KW)
                                        acc[py][px] = fmaf(av, wc[(py + 2 * j) * KW + px + 2 * i], acc[py][px]);
                        }
                }
#pragma unroll
                for (int py = 0; py < 2; py++) {
                    const int oy = 2 * m + py;
                    const bool ok0 = img_ok && oy < a.OH && 2 * n < a.OW, ok1 = ok0 && pair_ok;
                    float* op = ob + co * OHW + (unsigned)min(oy, a.OH - 1) * a.OW + min(2 * n, a.OW - 1);
                    if (ok1) *reinterpret_cast<float2*>(op) = make_float2(acc[py][0], acc[py][1]);
                    else if (ok0) op[0] = acc[py][0];
                    const float v0 = ok0 ? acc[py][0] : 0.f, v1 = ok1 ? acc[py][1] : 0.f;
                    s1[co] += v0 + v1;
                    s2[co] = fmaf(v0, v0, fmaf(v1, v1, s2[co]));
                }
            }
#pragma unroll
            for (int c = 0; c < CIN; c++) {
                actP[c][0] = actN[c][0];
                actP[c][1] = actN[c][1];
            }
        }
    });
    if (!a.stats) return;
    {
        float red[NRED];
#pragma unroll
        for (int co = 0; co < COUT; co++) {
            red[2 * co] = s1[co];
            red[2 * co + 1] = s2[co];
        }
        wave_sums_lds<NRED>(red, wscr + wv * kWsumScratch, redf + wv * NRED, lane);
    }
    __syncthreads();
    if (tid < NRED) {
        const double s = (double)redf[tid] + (double)redf[NRED + tid] + (double)redf[2 * NRED + tid] + (double)redf[3 * NRED + tid];
        acc_add<ACC_STAT>(&a.stats[((size_t)(blockIdx.x & (kStatShards - 1)) * COUT + (tid >> 1)) * 4 + (tid & 1)], s);
    }
}

}  // namespace cae
