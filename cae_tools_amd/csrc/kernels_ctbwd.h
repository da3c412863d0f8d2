// kernels_ctbwd.h — LDS-staged implicit GEMM on the fp32 matrix cores for the BACKWARD pass of the channel-rich stride-2
// ConvTranspose2d layers at the head of the decoder (decoder.py:44-48; 64->32, 32->16, 16->8 channels on 3x3 .. 31x31 maps
// at the benchmark geometry): input gradient and weight gradient from ONE staging of the layer's gradient map.
// Successor of k_ig_bwd_pair (kernels_igemm.h), whose two halves each gathered every MFMA operand per lane from global
// memory and re-applied the BatchNorm transforms per use (bound by its vector-memory instruction count).  Counterpart of
// k_ct_fwd_lds (kernels_ctlds.h).
//
// A workgroup owns `imgs` consecutive images and one block of 16 input channels:
//   1. one burst of coalesced global reads -> LDS: the images' gradient maps with BatchNorm-backward applied ONCE per element
//      (gy' = k1 g - k2 - (y - mean) k3, all Cout channels), the raw producer output of the block's 16 channels (it is both
//      the weight gradient's activation operand and the ReLU mask of the epilogue), the block's weight rows [16][Cout*9];
//   2. tasks shared round-robin by the waves, each a whole K loop of v_mfma_f32_16x16x4_f32 with both operands read from LDS:
//        input gradient, one tile of 16 positions:  gin[p][ci] = sum_{co,tap} gy'[co][2y+ky][2x+kx] W[ci][co][tap]
//            (K = Cout x 12: the nine taps of a channel padded to three k-steps; the pad slots multiply a zero weight);
//        weight gradient, one tile of 16 (co,tap) columns:  dW[ci][(co,tap)] = sum_p act(a[ci][p]) gy'[co][2y+ky][2x+kx]
//            (K = the workgroup's positions; fp64 atomics into the accumulator: B / imgs per address);
//   3. epilogue of the input gradient: ReLU mask of the producer, its BatchNorm-backward sums (per channel: lanes, then LDS,
//      then one fp64 atomic per channel and workgroup), stores.
// Only 3x3 kernels at stride 2 without padding (OH >= 2H + 1): everything else stays on k_ig_bwd_pair.
#pragma once
#include "kernels_gemm.h"

namespace cae {

struct CtBwd {
    int B, Cin, H, W, Cout, OH, OW;
    int imgs;              // images per workgroup
    int bands, hb;         // bands > 1 (imgs == 1 only): gridDim.z workgroups split the image into bands of hb input rows
    int wstr;              // LDS row stride of the weight slice (odd)
    const float* g;        // (B,Cout,OH,OW) masked gradient
    const float* yout;     // raw forward output of this layer (BN_BWD) or nullptr
    BnDesc bn_out;         // BN_BWD or BN_NONE
    const float* ain;      // (B,Cin,H,W) raw output of the producer (or plain activations)
    BnDesc bn_in;          // BN_SAVED (activation = relu(bn(ain)), and the epilogue masks with it) or BN_NONE
    const float* w;        // (Cin, Cout*9)
    float* gin;            // (B,Cin,H,W)
    double* stats_prev;    // [shards][Cin][4] slots 2,3 (with bn_in) or nullptr
    double* wacc;          // (Cin, Cout*9) fp64 accumulator, or its sharded side table (shard stride wacc_stride doubles)
    long long wacc_stride; // 0: not sharded
    BnGradOut bg;
    long long* dbg;
};

constexpr int kCtbThreads = 512;
constexpr int kCtbWaves = kCtbThreads / 64;
constexpr int kCtbG4 = 7;      // 16-byte pieces of the gradient maps a thread stages (x2 arrays: g and the raw output)
constexpr int kCtbA4 = 2;      // ... of the producer outputs
constexpr int kCtbW4 = 3;      // ... of the weight rows
constexpr int kCtbKChunk = 64; // positions per weight-gradient task (16 k-steps)

inline size_t ct_bwd_lds_bytes(int Cin, int Cout, int imgs, int HW, int OHW, int wstr) {
    size_t floats = 32 * (size_t)kCtbWaves;               // producer sums, per wave
    floats += 4 * (size_t)(Cin + Cout);                   // BatchNorm constants
    floats += (size_t)imgs * Cout * OHW;                  // gradient maps (a multiple of 4 floats: Cout % 4 == 0)
    floats += (size_t)imgs * 16 * HW;                     // producer outputs of the channel block
    floats += 16 * (size_t)wstr;                          // weight rows
    floats += 3 * (size_t)imgs * HW;                      // position tables
    floats += (size_t)kCtbWaves * 16 * 17;                // per-wave transpose tile of the input-gradient epilogue
    return (floats + 8) * sizeof(float);
}

#ifdef CAE_CTBWD_KERNEL   // the kernel itself is compiled in its own translation unit (ctbwd.hip); engine.hip sees the structs only
// grid (B / imgs rounded up, Cin / 16, parts or bands), block kCtbThreads, dynamic LDS = ct_bwd_lds_bytes(...)
// BAND: the workgroup owns input rows [y0, y1) of ONE image: it stages only the gradient rows 2 y0 .. 2 y1 and its own
// positions' producer outputs (4-byte loads: the pieces are not 16-byte aligned) and runs all of the band's tasks; otherwise
// it stages whole images and gridDim.z workgroups split the task list.
template <bool BAND>
__device__ __forceinline__ void ct_bwd_body(const CtBwd& a) {
    extern __shared__ double lds_d[];
    float* lstat = reinterpret_cast<float*>(lds_d);                     // [waves][16][2]: one writer per slot, folded in wave order
    float4* cout4 = reinterpret_cast<float4*>(lstat + 32 * kCtbWaves);  // [Cout]
    float4* cin4 = cout4 + a.Cout;                                      // [Cin]
    const int HW = a.H * a.W, OHW = a.OH * a.OW, KK = 9, N = a.Cout * KK;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * a.imgs, cb = blockIdx.y;
    const int nimg = BAND ? 1 : min(a.imgs, a.B - b0);
    // the band (BAND) / the whole map
    const int y0 = BAND ? (int)blockIdx.z * a.hb : 0;
    const int nrow = BAND ? min(a.hb, a.H - y0) : a.H;
    const int P = nimg * nrow * a.W;                                    // positions of this workgroup
    const int grows = BAND ? 2 * nrow + 1 : a.OH;                       // gradient rows staged per channel
    const int gstr = grows * a.OW;                                      // channel stride inside gimg
    const int astr = nrow * a.W;                                        // channel stride inside araw
    float* gimg = reinterpret_cast<float*>(cin4 + a.Cin);              // [imgs][Cout][gstr]   (16-byte aligned)
    float* araw = gimg + ((nimg * a.Cout * gstr + 3) & ~3);            // [imgs][16][astr]     (16-byte aligned)
    float* wl = araw + ((nimg * 16 * astr + 3) & ~3);                  // [16][wstr]
    int* pos_g = reinterpret_cast<int*>(wl + 16 * a.wstr);             // [P] offset of (img, 2y, 2x) inside gimg
    int* pos_a = pos_g + P;                                            // [P] offset of (img, pos) inside araw (channel 0)
    int* pos_o = pos_a + P;                                            // [P] offset of (img, pos) inside gin (image b0, channel 0)
    float* tiles = reinterpret_cast<float*>(pos_o + P);                // [waves][16 positions][17]
#define CTB_STAMP(i) do { if (a.dbg && threadIdx.x == 0 && (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x < 384) a.dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
    CTB_STAMP(0);
    const bool designated = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    const bool bwd = a.bn_out.mode == BN_BWD, act = a.bn_in.mode != BN_NONE;
    const float inv_hw = 1.0f / (float)(nrow * a.W), inv_w = 1.0f / (float)a.W, inv_gstr = 1.0f / (float)gstr,
                inv_cout = 1.0f / (float)a.Cout, inv_astr = 1.0f / (float)astr;

    // ---- one burst: everything the workgroup reads, requested before anything waits ----
    // whole maps: 16-byte loads (Cout % 4 == 0 and 16 channels per block make every image's piece of each tensor a whole
    // number of 16-byte pieces); a band: 4-byte loads (its rows start anywhere)
    constexpr int NG = BAND ? 8 : kCtbG4, NA = BAND ? 2 : kCtbA4;
    using GT = typename std::conditional<BAND, float, f32x4>::type;
    const int n_g = BAND ? a.Cout * gstr : nimg * a.Cout * OHW / 4;     // pieces
    GT gv[NG], yv[NG];
    if constexpr (BAND) {
        const float* gb = a.g + ((size_t)b0 * a.Cout * a.OH + 2 * y0) * a.OW;
        const float* yb = bwd ? a.yout + ((size_t)b0 * a.Cout * a.OH + 2 * y0) * a.OW : gb;
#pragma unroll
        for (int u = 0; u < NG; u++) {
            const int i = min(tid + u * kCtbThreads, n_g - 1);
            const int co = div_small(i, inv_gstr), off = co * OHW + (i - co * gstr);
            gv[u] = gb[off];
            yv[u] = yb[off];
        }
    } else {
        const f32x4* gsrc = reinterpret_cast<const f32x4*>(a.g + (size_t)b0 * a.Cout * OHW);
        const f32x4* ysrc = bwd ? reinterpret_cast<const f32x4*>(a.yout + (size_t)b0 * a.Cout * OHW) : gsrc;
#pragma unroll
        for (int u = 0; u < NG; u++) {
            const int i = min(tid + u * kCtbThreads, n_g - 1);
            gv[u] = gsrc[i];
            yv[u] = ysrc[i];
        }
    }
    const int n_a = BAND ? 16 * astr : nimg * 16 * HW / 4;              // pieces
    GT av[NA];
    if constexpr (BAND) {
        const float* ab = a.ain + (((size_t)b0 * a.Cin + cb * 16) * a.H + y0) * a.W;
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int i = min(tid + u * kCtbThreads, n_a - 1);
            const int c = div_small(i, inv_astr);
            av[u] = ab[c * HW + (i - c * astr)];
        }
    } else {
        const int n_a4 = 16 * HW / 4;   // per image
        const float inv_na4 = 1.0f / (float)n_a4;
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int i = min(tid + u * kCtbThreads, n_a - 1);
            const int im = div_small(i, inv_na4), rem = i - im * n_a4;
            av[u] = reinterpret_cast<const f32x4*>(a.ain + ((size_t)(b0 + im) * a.Cin + cb * 16) * HW)[rem];
        }
    }
    const int n_w4 = 16 * N / 4;
    f32x4 wr[kCtbW4];
#pragma unroll
    for (int u = 0; u < kCtbW4; u++)
        wr[u] = reinterpret_cast<const f32x4*>(a.w + (size_t)cb * 16 * N)[min(tid + u * kCtbThreads, n_w4 - 1)];
    // (keeps the compiler from sinking the loads into the predicated LDS stores below: load, wait, store, one by one)
#pragma unroll
    for (int u = 0; u < NG; u++) asm volatile("" : "+v"(gv[u]), "+v"(yv[u]));
#pragma unroll
    for (int u = 0; u < NA; u++) asm volatile("" : "+v"(av[u]));
#pragma unroll
    for (int u = 0; u < kCtbW4; u++) asm volatile("" : "+v"(wr[u]));
    CTB_STAMP(1);

    bn_consts(a.bn_out, cout4, false);
    bn_consts(a.bn_in, cin4, false, 64);
    if (designated && a.bg.stats) {
        for (int c = tid; c < a.bg.C; c += kCtbThreads) {
            double sb = 0.0, sg = 0.0;
            for (int sh = 0; sh < kStatShards; sh++) {
                sb += a.bg.stats[((size_t)sh * a.bg.C + c) * 4 + 2];
                sg += a.bg.stats[((size_t)sh * a.bg.C + c) * 4 + 3];
            }
            a.bg.beta_acc[c] = sb * a.bg.scale;
            a.bg.gamma_acc[c] = sg * a.bg.scale;
        }
    }
    if (tid < 32 * kCtbWaves) lstat[tid] = 0.f;
    // the producer outputs, the weights and the position tables do not need the constants
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int i = tid + u * kCtbThreads;
        if (i < n_a) reinterpret_cast<GT*>(araw)[i] = av[u];
    }
    {
        const float inv_n = 1.0f / (float)N;
#pragma unroll
        for (int u = 0; u < kCtbW4; u++) {
            const int i = tid + u * kCtbThreads;
            if (i < n_w4) {   // N % 4 == 0: a piece lies inside one row; the odd row stride makes it four 4-byte stores
                const int row = div_small(4 * i, inv_n);
                float* d = wl + row * a.wstr + (4 * i - row * N);
                d[0] = wr[u][0]; d[1] = wr[u][1]; d[2] = wr[u][2]; d[3] = wr[u][3];
            }
        }
    }
    for (int p = tid; p < P; p += kCtbThreads) {
        const int im = div_small(p, inv_hw), pi = p - im * (nrow * a.W);   // pi: position inside the band / map
        const int y = div_small(pi, inv_w), x = pi - y * a.W;
        pos_g[p] = im * a.Cout * gstr + 2 * y * a.OW + 2 * x;
        pos_a[p] = im * 16 * astr + pi;
        pos_o[p] = im * a.Cin * HW + y0 * a.W + pi;
    }
    __syncthreads();
    CTB_STAMP(2);
    // the gradient maps: BatchNorm-backward once per element
#pragma unroll
    for (int u = 0; u < NG; u++) {
        const int i = tid + u * kCtbThreads;
        if (i < n_g) {
            if constexpr (BAND) {
                float v = gv[u];
                if (bwd) {
                    const float4 k = cout4[div_small(i, inv_gstr)];
                    v = k.y * v - k.z - (yv[u] - k.x) * k.w;
                }
                gimg[i] = v;
            } else {   // (a 16-byte piece may straddle two channel planes)
                f32x4 v = gv[u];
                if (bwd) {
                    const int pl = div_small(4 * i, inv_gstr);         // plane (image * Cout + channel) of the first element
                    const int next = (pl + 1) * gstr;                   // first element of the next plane
                    const int co0 = pl - div_small(pl, inv_cout) * a.Cout;
                    const int co1 = co0 + 1 == a.Cout ? 0 : co0 + 1;
                    const float4 k0 = cout4[co0], k1 = cout4[co1];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float4 k = 4 * i + j >= next ? k1 : k0;
                        v[j] = k.y * v[j] - k.z - (yv[u][j] - k.x) * k.w;
                    }
                }
                reinterpret_cast<f32x4*>(gimg)[i] = v;
            }
        }
    }
    __syncthreads();
    CTB_STAMP(3);

    // ---- tasks ----
    const int mtiles = (P + 15) >> 4;                         // input-gradient tasks: one tile of 16 positions each
    const int ntiles = (N + 15) >> 4;                         // weight-gradient tiles ...
    constexpr int kChunk = kCtbKChunk;                        // (32 in band mode: more, smaller tasks - and twice the atomics: 171.0 against 169.2 us per step)
    const int kchunks = (P + kChunk - 1) / kChunk;            // ... times chunks of positions = weight-gradient tasks
    const float4 kin = act ? cin4[cb * 16 + r] : make_float4(0.f, 0.f, 0.f, 0.f);   // lane r <-> channel cb*16 + r (weight-gradient tasks)
    // input-gradient epilogue: lane (pl, cg) = position pl of the tile, channels cb*16 + 4 cg .. + 3
    const int pl = lane & 15, cg = lane >> 4;
    float4 kin4[4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) kin4[cc] = act ? cin4[cb * 16 + 4 * cg + cc] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // producer sums of those four channels over this lane's positions
    float* tl = tiles + wv * (16 * 17);
    const float inv_kk = 1.0f / 9.0f, inv_3 = 1.0f / 3.0f;
    double* wacc_sh = a.wacc + (size_t)(blockIdx.x & (kStatShards - 1)) * a.wacc_stride;   // image groups spread over the shards
    float* gin0 = a.gin + ((size_t)b0 * a.Cin + cb * 16) * HW;
    // whole maps: gridDim.z workgroups stage the same images and share the tasks (the staging is a fixed cost per workgroup,
    // the tasks are what there is to spread over the chip); a band: all of its tasks
    const int tfirst = BAND ? wv : wv + kCtbWaves * (int)blockIdx.z, tstep = BAND ? kCtbWaves : kCtbWaves * (int)gridDim.z;
    for (int task = tfirst; task < mtiles + ntiles * kchunks; task += tstep) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc;
        if (task < mtiles) {
            // input gradient: rows = positions task*16 .. +15, columns = the 16 channels, K = Cout x (9 taps padded to 12)
            const int p = min(task * 16 + r, P - 1);
            const float* ap = gimg + pos_g[p];
            const float* bp = wl + r * a.wstr;
            // lane constants of the three k-steps of a channel: slot (ts, q) is tap 4 ts + q (taps 9..11: zero weight)
            int aoff[3], boff[3];
#pragma unroll
            for (int ts = 0; ts < 3; ts++) {
                const int tap = min(4 * ts + q, 8);
                const int ky = div_small(tap, inv_3);
                aoff[ts] = ky * a.OW + (tap - ky * 3);
                boff[ts] = tap;
            }
            const bool pad2 = q != 0;   // step 2 holds taps 8, 9, 10, 11
            // four channels per trip (host: Cout % 4 == 0): 24 LDS reads in flight ahead of 12 MFMAs on two accumulators
            // (a dependent 16x16x4 MFMA waits for the one before it)
            for (int co = 0; co < a.Cout; co += 4) {
                float xa[4][3], xb[4][3];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float* ac = ap + (co + c) * gstr;
                    const float* bc = bp + (co + c) * KK;
#pragma unroll
                    for (int ts = 0; ts < 3; ts++) {
                        xa[c][ts] = ac[aoff[ts]];
                        xb[c][ts] = bc[boff[ts]];
                    }
                    xb[c][2] = pad2 ? 0.f : xb[c][2];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int ts = 0; ts < 3; ts++) {
                        if ((c * 3 + ts) & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c][ts], xb[c][ts], acc1, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c][ts], xb[c][ts], acc, 0, 0, 0);
                    }
                }
            }
            // epilogue.  Register j of lane (r, q) is position task*16 + 4q + j, channel cb*16 + r: stored from that layout a
            // store instruction touches 64 lines (the tensors are channel-major).  Transposed through the wave's own LDS tile
            // a lane owns one position and four channels: 16 consecutive positions per instruction and channel.
#pragma unroll
            for (int j = 0; j < 4; j++) tl[(4 * q + j) * 17 + r] = acc[j] + acc1[j];
            const int pp = task * 16 + pl;
            if (pp < P) {
                const int oa = pos_a[pp], oo = pos_o[pp];
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    const int ch = 4 * cg + cc;
                    float v = tl[pl * 17 + ch];
                    if (act) {
                        const float d = araw[oa + ch * astr] - kin4[cc].x;
                        v = fmaf(d, kin4[cc].y, kin4[cc].z) > 0.f ? v : 0.f;
                        s1[cc] += v;
                        s2[cc] = fmaf(v, d * kin4[cc].w, s2[cc]);
                    }
                    gin0[oo + ch * HW] = v;
                }
            }
        } else {
            // weight gradient: rows = the 16 channels, columns = one tile of 16 (co, tap), K = one chunk of positions
            const int wt = task - mtiles;
            const int nt = wt % ntiles, kc = wt / ntiles;
            const int n = nt * 16 + r;
            const bool n_ok = n < N;
            const int nc = n_ok ? n : 0;
            const int co = div_small(nc, inv_kk), tap = nc - co * KK;
            const int ky = div_small(tap, inv_3);
            const float* gb = gimg + co * gstr + ky * a.OW + (tap - ky * 3);
            const float* ar = araw + r * astr;
            const int pbeg = kc * kChunk, pend = min(P, pbeg + kChunk);
            // eight k-steps per trip: the table reads, then the operand reads, then eight MFMAs on two accumulators (one step
            // at a time the loop is a chain of two dependent LDS reads and an MFMA per step)
            for (int p0 = pbeg; p0 < pend; p0 += 32) {
                int ia[8], ig[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int p = p0 + 4 * u + q;
                    ok[u] = p < pend;
                    const int pc = ok[u] ? p : pbeg;
                    ia[u] = pos_a[pc];
                    ig[u] = pos_g[pc];
                }
                float va[8], vb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    va[u] = ar[ia[u]];
                    vb[u] = gb[ig[u]];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    float x = va[u];
                    if (act) x = fmaxf(0.f, fmaf(x - kin.x, kin.y, kin.z));
                    va[u] = ok[u] ? x : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[u], vb[u], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[u + 1], vb[u + 1], acc1, 0, 0, 0);
                }
            }
            // register j of lane (r, q) is row (channel) 4q + j, column n
            if (n_ok) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc_add<ACC_GRAD>(&wacc_sh[(size_t)(cb * 16 + 4 * q + j) * N + n], (double)(acc[j] + acc1[j]));
            }
        }
    }
    CTB_STAMP(4);
    // ---- the producer's BatchNorm-backward sums ----
    if (a.stats_prev) {
        // the 16 lanes of a row hold the same four channels: DPP row sums, then one LDS atomic per channel and wave
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            float t1 = s1[cc], t2 = s2[cc];
            t1 += dpp_f<0xB1>(t1);  t2 += dpp_f<0xB1>(t2);
            t1 += dpp_f<0x4E>(t1);  t2 += dpp_f<0x4E>(t2);
            t1 += dpp_f<0x141>(t1); t2 += dpp_f<0x141>(t2);
            t1 += dpp_f<0x140>(t1); t2 += dpp_f<0x140>(t2);
            if (pl == 0) {   // the only lane of this wave with channel (cg, cc): a plain store, no atomics
                lstat[wv * 32 + 2 * (4 * cg + cc)] = t1;
                lstat[wv * 32 + 2 * (4 * cg + cc) + 1] = t2;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int c = cb * 16 + (tid >> 1);
            const int shard = (blockIdx.x + blockIdx.y + blockIdx.z) & (kStatShards - 1);
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kCtbWaves; w++) t += (double)lstat[w * 32 + tid];
            acc_add<ACC_GRAD>(&a.stats_prev[((size_t)shard * a.Cin + c) * 4 + 2 + (tid & 1)], t);
        }
    }
    CTB_STAMP(5);
#undef CTB_STAMP
}

__global__ void __launch_bounds__(kCtbThreads) k_ct_bwd_lds(CtBwd a) {
    kernarg_warm<sizeof(CtBwd)>();
    ct_bwd_body<false>(a);
}
__global__ void __launch_bounds__(kCtbThreads) k_ct_bwd_band(CtBwd a) {
    kernarg_warm<sizeof(CtBwd)>();
    ct_bwd_body<true>(a);
}
#endif   // CAE_CTBWD_KERNEL

}  // namespace cae
