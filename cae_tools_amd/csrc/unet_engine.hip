// unet_engine.hip — host side of the UNET path: tensor table, workspace carving, launch sequence and the
// extern "C" ABI declared in include/cae_unet.h.
//
// Which kernels a layer runs (conv_down / conv_up / conv_wgrad / lin_fwd / lin_bwd below decide, in this order):
//   4x4 / s2 / p1 layers with <= 4 channels on the large side (the image end)   kernels_unet_thin.h   (patch in LDS, weights in registers)
//   ... with maps 16 / 32 / 64 / 128 wide and channel counts multiples of 4 / 8   kernels_unet_patch.h  (patch in LDS, weight tiles in LDS)
//   any other 4x4 / s2 / p1 layer                                                 kernels_unet_mfma.h   (im2col tile engine)
//   anything else, and everything when `specialised` is off                      kernels_unet.h        (shape-generic)
//   Linear layers: both sides <= 1024 -> the ConvAE path's 16x16 GEMM (ugemm); batch <= 64 -> kernels_unet_lin.h; else the tile engine.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "cae_unet.h"
#include "kernels_unet.h"
#include "kernels_unet_mfma.h"
#include "kernels_unet_lin.h"
#include "kernels_unet_thin.h"
#include "kernels_unet_patch.h"

// The ConvAE path's 16x16-tile MFMA GEMM (kernels_gemm.h: four waves split K, one launch for a Linear layer's weight gradient
// beside its input gradient) for the SMALL Linear layers here (fc -> latent -> fc): the convolution tile engine's strided
// GEMM policy runs such a layer in a single workgroup's chunk loop, 30-40 us for a few hundred kFLOP.  Its own namespace, as
// in ctbwd.hip, so that the kernels' host stubs do not collide with engine.hip's.
namespace ugemm {
#include "kernels_generic.h"
#include "kernels_gemm.h"
}  // namespace ugemm

void cae_detail_set_error(const char* msg);   // engine.hip: message returned by cae_last_error()

using namespace unet;

namespace {

int ufail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cae_detail_set_error(buf);
    return code;
}

#define UHIP_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess)                                                                                   \
            return ufail(CAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr float kEps = 1e-5f;       // nn.BatchNorm default
constexpr float kMomentum = 0.1f;   // nn.BatchNorm default
constexpr int kLossSlots = 4096;

// dropout sites (oracle/unet_oracle.py)
constexpr uint32_t SITE_ENC_CONV = 0, SITE_ENC_FC0 = 100, SITE_ENC_FC1 = 101, SITE_DEC_FC0 = 102, SITE_DEC_FC1 = 103,
                   SITE_DEC_CONV = 200;

struct Bn {
    int C = 0;
    int64_t gamma = 0, beta = 0;     // parameter arena
    int64_t rmean = 0, rvar = 0;     // buffer arena
    int64_t saved = 0;               // workspace floats [C][2]
    int64_t sums = 0, bsums = 0;     // workspace doubles [C][2] each (forward / backward)
};

struct ConvLayer {
    Geom g;                          // B filled per call
    int64_t w = 0, b = 0;            // parameter arena
    int64_t wp = -1;                 // workspace: weights repacked per output parity (layers that run OpUp), or -1
    bool has_bn = false, has_skip = false;
    Bn bn;
    int R = 0;                       // attention hidden width
    int64_t w1 = 0, w2 = 0;          // attention weights
    // workspace (float offsets)
    int64_t z = 0, s = 0, a = 0;     // enc: raw conv out, relu(bn), dropout(relu(bn))
    int64_t u = 0, din_next = 0, pool = 0, psum = 0, hid = 0, att = 0, dpool = 0, da = 0;   // dec
    int64_t gz = 0, ga = 0;          // enc grads: g/dz (C,HW), grad wrt a
    int64_t gu = 0, gcat = 0, gdin = 0;   // dec grads: du, g/dcat, grad wrt this layer's input
};

struct Fc {
    int nin = 0, nout = 0;
    int64_t w = 0, b = 0;
    bool has_bn = false;
    Bn bn;
    int64_t h = 0, a = 0;            // raw output, activated (+dropout) output
    int64_t gh = 0, ga = 0;          // grad wrt raw output (in place of ga after activation backward), grad wrt a
};

struct DataSet {
    const float* x = nullptr;
    const float* t = nullptr;
    const float* m = nullptr;
    int mc = 0;
    int64_t n = 0;
};

}  // namespace

struct unet_engine {
    std::vector<ConvLayer> enc, dec;
    Fc fc[4];                        // enc_lin.0, enc_lin.4, dec_lin.0, dec_lin.4
    int fc_size = 0, latent = 0, max_batch = 0;
    int in_c = 0, in_h = 0, in_w = 0, out_c = 0, out_h = 0, out_w = 0;
    std::vector<cae_tensor_info_t> tensors;
    int64_t n_params = 0, n_buffers = 0;
    // workspace
    int64_t ws_bytes = 0;
    int64_t off_gacc = 0, off_dsum = 0, n_dsum = 0, off_losses = 0, off_ls = 0, off_f32 = 0, off_gscratch = 0, gscratch_bytes = 0;
    int64_t off_thinpart = 0, thinpart_bytes = 0;   // weight-gradient partial tiles: per workgroup (kernels_unet_thin.h), per K slice (tile engine)
    int64_t off_linpart = 0, linpart_bytes = 0;   // K-slice partial tiles of the big Linear layers (kernels_unet_lin.h)
    bool fc_f32[4] = {false, false, false, false};   // this backward stored fc[k]'s weight gradient as fp32 (F32Ranges)
    bool fc_f32_dirty[4] = {false, false, false, false};   // ... and nothing has cleared those accumulator slots since
    int64_t xb = 0, y = 0, coef = 0;
    char* ws = nullptr;
    float *params = nullptr, *m = nullptr, *v = nullptr, *buffers = nullptr;
    hipStream_t stream = nullptr;
    Hyper hyper{1e-3, 0.9, 0.999, 1e-8, 1e-5};
    double dropout = 0.1, lambda_p = 1.0;
    uint32_t seed = 0;
    int64_t step = 0;
    int specialised = 1;
    bool gacc_clean = false;   // the fp64 gradient accumulator is all zero (k_adamw clears what it consumes) but for fc_f32_dirty
    DataSet ds[2];

    float* f(int64_t off) const { return reinterpret_cast<float*>(ws + off_f32) + off; }
    double* dsum(int64_t off) const { return reinterpret_cast<double*>(ws + off_dsum) + off; }
    double* gacc(int64_t off) const { return reinterpret_cast<double*>(ws + off_gacc) + off; }
    float* P(int64_t off) const { return params + off; }
    float* Bf(int64_t off) const { return buffers + off; }
};

namespace {

int64_t add_tensor(unet_engine* e, const std::string& name, int arena, std::vector<int64_t> shape) {
    cae_tensor_info_t t;
    memset(&t, 0, sizeof t);
    snprintf(t.name, sizeof t.name, "%s", name.c_str());
    t.arena = arena;
    t.ndim = (int)shape.size();
    int64_t n = 1;
    for (size_t i = 0; i < shape.size(); i++) {
        t.shape[i] = shape[i];
        n *= shape[i];
    }
    int64_t& top = arena == 0 ? e->n_params : e->n_buffers;
    top = (top + 3) & ~int64_t(3);
    t.offset = top;
    t.numel = n;
    top += n;
    e->tensors.push_back(t);
    return t.offset;
}

void add_bn(unet_engine* e, const std::string& key, int C, Bn& bn) {
    bn.C = C;
    bn.gamma = add_tensor(e, key + ".weight", 0, {C});
    bn.beta = add_tensor(e, key + ".bias", 0, {C});
    bn.rmean = add_tensor(e, key + ".running_mean", 1, {C});
    bn.rvar = add_tensor(e, key + ".running_var", 1, {C});
}

int blocks_for(long long n, int cap = 8192) {
    long long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

Drop make_drop(const unet_engine* e, uint32_t site, bool train) {
    Drop d{0, 0, 1.f, 0};
    if (!train || e->dropout <= 0.0) return d;
    d.on = 1;
    const uint32_t k = pcg(e->seed + 0x9E3779B9u * site);
    d.key = pcg(k ^ (uint32_t)(e->step & 0xFFFFFFFF));
    double t = e->dropout * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    d.scale = (float)(1.0 / (1.0 - e->dropout));
    return d;
}

// ---- conv dispatch --------------------------------------------------------------------------------
void conv_down(unet_engine* e, const Geom& g, const float* L, const float* w, const float* bias, float* S) {
    static const int thin_off = getenv("CAE_UNET_THIN") ? atoi(getenv("CAE_UNET_THIN")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !thin_off && thin_geom(g)) {   // the image-end layers: kernels_unet_thin.h
        thin_down_launch(g, L, w, bias, S, e->stream);
        return;
    }
    static const int patch_off = getenv("CAE_UNET_PATCH") ? atoi(getenv("CAE_UNET_PATCH")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !patch_off && patch_geom(g)) {   // the wide layers: kernels_unet_patch.h
        pdown_launch(g, L, w, bias, S, e->stream);
        return;
    }
    if (e->specialised && mfma_down_eligible(g)) {
        mfma_down_launch(g, L, w, bias, S, e->stream);
        return;
    }
    const long long total = (long long)g.B * g.Cs * g.Hs * g.Ws;
    hipLaunchKernelGGL(k_down, dim3(blocks_for(total, 65536)), dim3(256), 0, e->stream, g, L, w, bias, S);
}

// wp: the layer's repacked weights (pack_up_weights below) or nullptr when the layer does not run the tile engine
void conv_up(unet_engine* e, const Geom& g, const float* S, const float* w, const float* wp, const float* bias, float* L) {
    static const int thin_off = getenv("CAE_UNET_THIN") ? atoi(getenv("CAE_UNET_THIN")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !thin_off && wp && thin_up_geom(g)) {   // ... 128 columns wide: the row walk of kernels_unet_thin.h
        thin_up_launch(g, S, wp, bias, L, e->stream);
        return;
    }
    if (e->specialised && mfma_geom(g) && g.Cl <= 4) {   // a handful of output channels: streaming kernel, not a GEMM
        const int blocks = blocks_for((long long)g.B * g.Hs * g.Ws, 65536);
        switch (g.Cl) {
            case 1: hipLaunchKernelGGL(k_up_thin<1>, dim3(blocks), dim3(256), 0, e->stream, g, S, w, bias, L); break;
            case 2: hipLaunchKernelGGL(k_up_thin<2>, dim3(blocks), dim3(256), 0, e->stream, g, S, w, bias, L); break;
            case 3: hipLaunchKernelGGL(k_up_thin<3>, dim3(blocks), dim3(256), 0, e->stream, g, S, w, bias, L); break;
            default: hipLaunchKernelGGL(k_up_thin<4>, dim3(blocks), dim3(256), 0, e->stream, g, S, w, bias, L); break;
        }
        return;
    }
    static const int patch_off = getenv("CAE_UNET_PATCH") ? atoi(getenv("CAE_UNET_PATCH")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !patch_off && wp && g.Cl > 4 && pup_geom(g)) {   // the wide layers: kernels_unet_patch.h
        pup_launch(g, S, wp, bias, L, e->stream);
        return;
    }
    if (e->specialised && wp && mfma_up_eligible(g)) {
        mfma_up_launch(g, S, wp, bias, L, e->stream);
        return;
    }
    const long long total = (long long)g.B * g.Cl * g.Hl * g.Wl;
    hipLaunchKernelGGL(k_up, dim3(blocks_for(total, 65536)), dim3(256), 0, e->stream, g, S, w, bias, L);
}

void conv_wgrad(unet_engine* e, const Geom& g, const float* S, const float* L, double* acc) {
    static const int thin_off = getenv("CAE_UNET_THIN") ? atoi(getenv("CAE_UNET_THIN")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !thin_off && thin_geom(g)) {   // the image-end layers: kernels_unet_thin.h
        float* part = thin_wgrad_part_bytes(g) <= (size_t)e->thinpart_bytes ? reinterpret_cast<float*>(e->ws + e->off_thinpart) : nullptr;
        thin_wgrad_launch(g, S, L, acc, part, e->stream);
        return;
    }
    static const int patch_off = getenv("CAE_UNET_PATCH") ? atoi(getenv("CAE_UNET_PATCH")) == 0 : 0;   // env: A/B measurements only
    if (e->specialised && !patch_off && pwgrad_geom(g) && pwgrad_part_bytes(g) <= (size_t)e->thinpart_bytes) {   // kernels_unet_patch.h
        pwgrad_launch(g, S, L, acc, reinterpret_cast<float*>(e->ws + e->off_thinpart), e->stream);
        return;
    }
    if (e->specialised && mfma_wgrad_eligible(g)) {
        static const int part_off = getenv("CAE_UNET_WGPART") ? atoi(getenv("CAE_UNET_WGPART")) == 0 : 0;   // env: A/B measurements only
        const size_t need = mfma_wgrad_part_bytes(g);
        float* part = (!part_off && need && need <= (size_t)e->thinpart_bytes) ? reinterpret_cast<float*>(e->ws + e->off_thinpart) : nullptr;
        mfma_wgrad_launch(g, S, L, acc, part, e->stream);
        return;
    }
    const long long per = (long long)g.B * g.Hs * g.Ws;
    int split = (int)((per + 256 * 16 - 1) / (256 * 16));
    if (split > 16) split = 16;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(k_wgrad, dim3(g.Cs * g.Cl * g.kh * g.kw, split), dim3(256), 0, e->stream, g, S, L, acc);
}

// kernels that end in a workgroup-wide sum + atomics: fewer, longer workgroups (the sum's two barriers and the atomics are a
// fixed 2-3 us; at 128 chunks a workgroup of the 128 x 128 maps had four loop trips of work in front of them)
dim3 red_grid(int B, int C, int HW) {
    static const int target = getenv("CAE_UNET_REDWGS") ? atoi(getenv("CAE_UNET_REDWGS")) : 1280;   // env: tuning runs only
    int chunks = (int)(((long long)B * HW + 256 * 4 - 1) / (256 * 4));
    const int want = std::max(4, target / std::max(1, C));     // about `target` workgroups over the C channels
    if (chunks > want) chunks = want;
    if (chunks < 1) chunks = 1;
    return dim3(chunks, C);
}

void chan_sums(unet_engine* e, const float* x, long long bs, int B, int C, int HW, double* sums, int sstride, int want_sq) {
    const int chunks = (int)red_grid(B, C, HW).x;
    hipLaunchKernelGGL(k_chan_sums, dim3(chunks, C), dim3(256), 0, e->stream, x, bs, B, HW, sums, sstride, want_sq);
}

dim3 ew_grid(int B, int C, int HW) {
    int chunks = (int)(((long long)B * HW + 256 * 4 - 1) / (256 * 4));
    if (chunks > 128) chunks = 128;
    if (chunks < 1) chunks = 1;
    return dim3(chunks, C);
}

// accumulate the batch sums of x (B,C,HW; batch stride bs); bn_act turns them into statistics
void bn_stats(unet_engine* e, const Bn& bn, const float* x, long long bs, int B, int HW) {
    chan_sums(e, x, bs, B, bn.C, HW, e->dsum(bn.sums), 2, 1);
}

const ZCat kNoCat{nullptr, nullptr, nullptr, 0};

// skip_sums: where the sums of the ReLU output go (the skip half of a decoder layer's BatchNorm statistics), or nullptr
void bn_act(unet_engine* e, const Bn& bn, const float* z, long long zbs, int B, int HW, bool train, Drop d, float* s_out,
            float* a_out, const ZCat& zc = kNoCat, double* skip_sums = nullptr) {
    // (with sums to leave, fewer and longer workgroups: red_grid)
    hipLaunchKernelGGL(k_bn_act, skip_sums ? red_grid(B, bn.C, HW) : ew_grid(B, bn.C, HW), dim3(256), 0, e->stream, z, zbs, B, bn.C, HW,
                       e->f(bn.saved), e->Bf(bn.rmean), e->Bf(bn.rvar), kEps, train ? 2 : 1, e->dsum(bn.sums), (double)B * HW, kMomentum,
                       e->P(bn.gamma), e->P(bn.beta), d, s_out, a_out, zc, skip_sums);
}

void bn_backward(unet_engine* e, const Bn& bn, const float* gA, long long gAbs, const float* gB, long long gBbs,
                 const float* z, long long zbs, int B, int HW, Drop d, float* g_io, const ZCat& zc = kNoCat) {
    // pass 1 only sums; pass 2 forms the masked gradient again from gA / gB / z and writes dz (g_io may alias neither input:
    // the callers pass distinct buffers)
    hipLaunchKernelGGL(k_bn_bwd_reduce, red_grid(B, bn.C, HW), dim3(256), 0, e->stream, gA, gAbs, gB, gBbs, z, zbs, B, bn.C,
                       HW, e->f(bn.saved), e->P(bn.gamma), e->P(bn.beta), d, (float*)nullptr, e->dsum(bn.bsums), zc);
    hipLaunchKernelGGL(k_bn_bwd_apply2, ew_grid(B, bn.C, HW), dim3(256), 0, e->stream, gA, gAbs, gB, gBbs, z, zbs, B, bn.C, HW,
                       e->f(bn.saved), e->P(bn.gamma), e->P(bn.beta), d, e->dsum(bn.bsums), (double)B * HW, e->gacc(bn.gamma),
                       e->gacc(bn.beta), g_io, zc);
}

// small Linear layers: the whole weight matrix is a few 16x16 tiles' worth (see the include of kernels_gemm.h above)
// (both sides <= 1024: the 16x16-tile kernel sums K in four fp32 chains of K / 4, fine for a few hundred terms; a K = 6144
// layer on it put the gradients behind a five-row BatchNorm1d 2-5 % from the oracle where the tile engine's 256-long chains
// with an fp64 combine stay within 1.5 % - tests/test_unet_hip_parity.py)
bool small_fc(const Fc& L) { return L.nin <= 1024 && L.nout <= 1024; }
size_t gemm16_lds() { return (4 * 256 + 256) * sizeof(float) + sizeof(float4); }

ugemm::cae::GemmArgs gemm16_args(int M, int N, int K) {
    ugemm::cae::GemmArgs g;
    memset(&g, 0, sizeof g);
    g.M = M, g.N = N, g.K = K;
    return g;
}

// the big layers on kernels_unet_lin.h: batch <= 64, 16-byte rows, room for the K slices' partial tiles
bool lin_big(const unet_engine* e, const Fc& L, int B) {
    if (!e->specialised || small_fc(L) || B > 64 || (L.nin & 3) || (L.nout & 3)) return false;
    static const int off = getenv("CAE_UNET_LIN") ? atoi(getenv("CAE_UNET_LIN")) == 0 : 0;   // env: A/B measurements only
    if (off) return false;
    const long long need = (long long)std::max((L.nin + 127) / 128 * (long long)L.nout, (L.nout + 127) / 128 * (long long)L.nin) * B * 4;
    return need <= e->linpart_bytes;
}

// Y[b][n] = bias[n] + sum_k X[b][k] W[..]: nt = W[n][k] (forward), else W[k][n] (input gradient)
void lin_gemm(unet_engine* e, bool nt, const float* X, long long x_ld, const float* W, long long w_ld, const float* bias, float* Y,
              int B, int N, int K) {
    LinArgs a;
    a.X = X, a.x_ld = x_ld, a.W = W, a.w_ld = w_ld, a.bias = bias, a.Y = Y, a.y_ld = N, a.B = B, a.N = N, a.K = K, a.kiters = 2;
    const int slices = (K + a.kiters * kLinKT - 1) / (a.kiters * kLinKT);
    a.part = slices > 1 ? reinterpret_cast<float*>(e->ws + e->off_linpart) : nullptr;
    const dim3 grid((N + kLinNT - 1) / kLinNT, slices);
    const int RB = B > 32 ? 2 : 1;
    if (nt) {
        const size_t lds = (size_t)(32 * RB + 128) * kLinPK * sizeof(float);
        if (RB == 1) hipLaunchKernelGGL(k_lin_nt<1>, grid, dim3(256), lds, e->stream, a);
        else hipLaunchKernelGGL(k_lin_nt<2>, grid, dim3(256), lds, e->stream, a);
    } else {
        const size_t lds = (size_t)(32 * RB * kLinPK + kLinKT * kLinPN) * sizeof(float);
        if (RB == 1) hipLaunchKernelGGL(k_lin_nn<1>, grid, dim3(256), lds, e->stream, a);
        else hipLaunchKernelGGL(k_lin_nn<2>, grid, dim3(256), lds, e->stream, a);
    }
    if (slices > 1)
        hipLaunchKernelGGL(k_lin_fold, dim3((unsigned)(((long long)B * N + 15) / 16)), dim3(256), 0, e->stream, a.part, slices, B, N, bias,
                           Y, (long long)N);
}

void lin_fwd(unet_engine* e, const Fc& L, int B, const float* in, float* out) {
    if (lin_big(e, L, B)) {   // out[b][o] = bias[o] + sum_i in[b][i] W[o][i]
        lin_gemm(e, true, in, L.nin, e->P(L.w), L.nin, e->P(L.b), out, B, L.nout, L.nin);
        return;
    }
    if (e->specialised && small_fc(L)) {   // out[b][o] = bias[o] + sum_i in[b][i] W[o][i]
        ugemm::cae::GemmArgs ga = gemm16_args(B, L.nout, L.nin);
        ga.A = in, ga.sa_m = L.nin, ga.sa_k = 1;
        ga.B = e->P(L.w), ga.sb_k = 1, ga.sb_n = L.nin;           // B[k][n] = W[n][k]
        ga.C = out, ga.sc_m = L.nout, ga.sc_n = 1;
        ga.epi = ugemm::cae::GE_STORE;
        ga.bias = e->P(L.b);
        const int tiles = ((B + 15) / 16) * ((L.nout + 15) / 16);
        hipLaunchKernelGGL(ugemm::cae::k_gemm16, dim3(tiles), dim3(256), gemm16_lds(), e->stream, ga);
        return;
    }
    if (e->specialised) {   // out[b][o] = bias[o] + sum_i in[b][i] W[o][i]
        GemmDesc d{L.nout, B, L.nin, e->P(L.w), L.nin, 1, in, 1, L.nin, e->P(L.b), out, nullptr, 1, L.nout, 0};
        gemm_launch(d, reinterpret_cast<double*>(e->ws + e->off_gscratch), e->stream);
        return;
    }
    hipLaunchKernelGGL(k_lin_fwd, dim3(L.nout, (B + 7) / 8), dim3(256), 0, e->stream, B, L.nin, L.nout, in, e->P(L.w),
                       e->P(L.b), out);
}

void lin_bwd(unet_engine* e, const Fc& L, int B, const float* in, const float* gout, float* gin) {
    const int fk = (int)(&L - e->fc);
    const bool big = lin_big(e, L, B);
    if (!big && e->fc_f32_dirty[fk]) {   // an earlier step left fp32 gradients in these fp64 slots
        (void)hipMemsetAsync(e->gacc(L.w), 0, (size_t)L.nin * L.nout * sizeof(double), e->stream);
        e->fc_f32_dirty[fk] = false;
    }
    e->fc_f32[fk] = big;
    if (big) {
        // dW[o][i] = sum_b gout[b][o] in[b][i] as fp32 in the first half of the weight's accumulator slots (F32Ranges),
        // db[o] = sum_b gout[b][o];  gin[b][i] = sum_o gout[b][o] W[o][i]
        e->fc_f32_dirty[fk] = true;
        const int B8 = (B + 7) & ~7;
        const size_t lds = (size_t)2 * B8 * kLinPN * sizeof(float);
        static bool raised = false;   // above the 64 KiB a kernel gets without asking (batch > 56)
        if (lds > 65536 && !raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lin_outer), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * kLinPN * 4);
            raised = true;
        }
        hipLaunchKernelGGL(k_lin_outer, dim3((L.nin + 127) / 128, (L.nout + 127) / 128), dim3(256), lds, e->stream, gout, (long long)L.nout, in,
                           (long long)L.nin, B, L.nout, L.nin, reinterpret_cast<float*>(e->gacc(L.w)), e->gacc(L.b));
        if (gin) lin_gemm(e, false, gout, L.nout, e->P(L.w), L.nin, nullptr, gin, B, L.nin, L.nout);
        return;
    }
    if (e->specialised && small_fc(L)) {
        // dW[o][i] = sum_b gout[b][o] in[b][i], db[o] = sum_b gout[b][o] (a ones column), K = the whole batch in one tile: plain
        // fp64 stores into the (zeroed) accumulator, one writer per element; beside it gin[b][i] = sum_o gout[b][o] W[o][i]
        ugemm::cae::GemmArgs gw = gemm16_args(L.nout, L.nin + 1, B);
        gw.A = gout, gw.sa_m = 1, gw.sa_k = L.nout;
        gw.B = in, gw.sb_k = L.nin, gw.sb_n = 1;
        gw.epi = ugemm::cae::GE_ACC64;
        gw.accW = e->gacc(L.w), gw.accB = e->gacc(L.b), gw.ones_col = 1;
        const int tiles_w = ((gw.M + 15) / 16) * ((gw.N + 15) / 16);
        if (!gin) {
            hipLaunchKernelGGL(ugemm::cae::k_gemm16, dim3(tiles_w), dim3(256), gemm16_lds(), e->stream, gw);
            return;
        }
        ugemm::cae::GemmArgs gd = gemm16_args(B, L.nin, L.nout);
        gd.A = gout, gd.sa_m = L.nout, gd.sa_k = 1;
        gd.B = e->P(L.w), gd.sb_k = L.nin, gd.sb_n = 1;           // B[k = o][n = i] = W[o][i]
        gd.C = gin, gd.sc_m = L.nin, gd.sc_n = 1;
        gd.epi = ugemm::cae::GE_STORE;
        const int tiles_d = ((gd.M + 15) / 16) * ((gd.N + 15) / 16);
        hipLaunchKernelGGL(ugemm::cae::k_gemm16_pair, dim3(tiles_w + tiles_d), dim3(256), gemm16_lds(), e->stream, gw, gd, tiles_w, 0);
        return;
    }
    if (e->specialised) {
        // dW[o][i] += sum_b gout[b][o] in[b][i];  db[o] += sum_b gout[b][o];  gin[b][i] = sum_o gout[b][o] W[o][i]
        GemmDesc w{L.nout, L.nin, B, gout, 1, L.nout, in, L.nin, 1, nullptr, nullptr, e->gacc(L.w), L.nin, 1, 2};
        gemm_launch(w, nullptr, e->stream);
        hipLaunchKernelGGL(k_col_sums, dim3((L.nout + 255) / 256), dim3(256), 0, e->stream, B, L.nout, gout, e->gacc(L.b));
        if (gin) {
            GemmDesc d{L.nin, B, L.nout, e->P(L.w), 1, L.nin, gout, 1, L.nout, nullptr, gin, nullptr, 1, L.nin, 0};
            gemm_launch(d, reinterpret_cast<double*>(e->ws + e->off_gscratch), e->stream);
        }
        return;
    }
    hipLaunchKernelGGL(k_lin_wgrad, dim3(blocks_for((long long)L.nout * L.nin, 65536)), dim3(256), 0, e->stream, B, L.nin,
                       L.nout, gout, in, e->gacc(L.w), e->gacc(L.b));
    if (gin)
        hipLaunchKernelGGL(k_lin_dgrad, dim3(blocks_for((long long)B * L.nin, 65536)), dim3(256), 0, e->stream, B, L.nin,
                           L.nout, gout, e->P(L.w), gin);
}

// the weights of every layer that runs OpUp (decoder forward; encoder input gradients when training), repacked in one launch
void pack_up_weights(unet_engine* e, bool train) {
    if (!e->specialised) return;
    PackSet ps;
    memset(&ps, 0, sizeof ps);
    auto add = [&](const ConvLayer& L) {
        if (L.wp < 0 || ps.n == 8) return;
        ps.thin[ps.n] = L.g.Cl <= 4;                           // (the image-end layer: k_thin_up's order)
        ps.Cs[ps.n] = L.g.Cs, ps.Cl[ps.n] = L.g.Cl, ps.w[ps.n] = e->P(L.w), ps.wp[ps.n] = e->f(L.wp);
        ps.begin[ps.n + 1] = ps.begin[ps.n] + (long long)16 * L.g.Cs * L.g.Cl;
        ps.n++;
    };
    for (auto& L : e->dec) add(L);
    if (train)
        for (size_t i = 1; i < e->enc.size(); i++) add(e->enc[i]);
    if (ps.n) hipLaunchKernelGGL(k_pack_up_weights, dim3(blocks_for(ps.begin[ps.n], 2048)), dim3(256), 0, e->stream, ps);
}

// ---- forward ----------------------------------------------------------------------------------------
// x: (B, in_c, in_h, in_w) contiguous.  Leaves the raw last-layer output in dec.back().u
int forward(unet_engine* e, const float* x, int B, bool train) {
    const int n = (int)e->enc.size();
    if (train) UHIP_TRY(hipMemsetAsync(e->dsum(0), 0, (size_t)e->n_dsum * sizeof(double), e->stream));
    pack_up_weights(e, train);
    const float* cur = x;
    for (int i = 0; i < n; i++) {
        ConvLayer& L = e->enc[i];
        Geom g = L.g;
        g.B = B;
        const int HW = g.Hs * g.Ws;
        conv_down(e, g, cur, e->P(L.w), e->P(L.b), e->f(L.z));
        if (train) bn_stats(e, L.bn, e->f(L.z), (long long)g.Cs * HW, B, HW);
        const Drop d = make_drop(e, SITE_ENC_CONV + i, train);
        // the skip is the ReLU output; the next layer sees it through the dropout (unet.py:105-107)
        // (a skip's sums are the second half of its decoder layer's BatchNorm statistics: dec[n - 2 - i], channels Cs ..)
        double* skip_sums = (train && i < n - 1) ? e->dsum(e->dec[n - 2 - i].bn.sums) + 2 * g.Cs : nullptr;
        bn_act(e, L.bn, e->f(L.z), (long long)g.Cs * HW, B, HW, train, d, e->f(L.s), d.on ? e->f(L.a) : nullptr, kNoCat, skip_sums);
        cur = d.on ? e->f(L.a) : e->f(L.s);
    }
    // encoder_lin / decoder_lin: Linear, BN1d, ReLU, Dropout, Linear, ReLU, Dropout (unet.py:92-100,121-129)
    const uint32_t sites[4] = {SITE_ENC_FC0, SITE_ENC_FC1, SITE_DEC_FC0, SITE_DEC_FC1};
    for (int k = 0; k < 4; k++) {
        Fc& L = e->fc[k];
        lin_fwd(e, L, B, cur, e->f(L.h));
        const Drop d = make_drop(e, sites[k], train);
        if (L.has_bn) {
            if (train) bn_stats(e, L.bn, e->f(L.h), L.nout, B, 1);
            bn_act(e, L.bn, e->f(L.h), L.nout, B, 1, train, d, nullptr, e->f(L.a));
        } else {
            hipLaunchKernelGGL(k_relu_drop, dim3(blocks_for((long long)B * L.nout)), dim3(256), 0, e->stream, e->f(L.h),
                               (long long)B * L.nout, d, e->f(L.a));
        }
        cur = e->f(L.a);
    }
    const int nd = (int)e->dec.size();
    for (int j = 0; j < nd; j++) {
        ConvLayer& L = e->dec[j];
        Geom g = L.g;
        g.B = B;
        const int HW = g.Hl * g.Wl, C = g.Cl;
        conv_up(e, g, cur, e->P(L.w), L.wp >= 0 ? e->f(L.wp) : nullptr, e->P(L.b), e->f(L.u));
        if (!L.has_bn) break;   // last layer: sigmoid is applied by the loss / score kernels
        const float* skip = e->f(e->enc[n - 2 - j].s);
        // the concatenated tensor (gated u | skip) is never written, and in training nothing reads its sources again for the
        // BatchNorm statistics: the pooling pass leaves sum u and sum u^2 per (b, c), the gate kernel scales them by att and
        // att^2, and the skip half's sums came with the encoder's BatchNorm + ReLU pass
        double* psum = reinterpret_cast<double*>(e->f(L.psum));
        hipLaunchKernelGGL(k_pool, dim3(B * C), dim3(256), 0, e->stream, e->f(L.u), HW, e->f(L.pool), train ? psum : nullptr);
        hipLaunchKernelGGL(k_att_fwd, dim3(B), dim3(256), (size_t)(2 * C + 2 * L.R) * sizeof(float), e->stream, e->f(L.pool),
                           C, L.R, e->P(L.w1), e->P(L.w2), e->f(L.att), e->f(L.hid), psum, train ? e->dsum(L.bn.sums) : nullptr);
        const Drop d = make_drop(e, SITE_DEC_CONV + j, train);
        bn_act(e, L.bn, nullptr, 0, B, HW, train, d, nullptr, e->f(L.din_next), ZCat{e->f(L.u), e->f(L.att), skip, C});
        cur = e->f(L.din_next);
    }
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

LossSrc loss_src(const unet_engine* e, int which, const int32_t* perm, int64_t start) {
    LossSrc s;
    s.target = e->ds[which].t;
    s.mask = e->ds[which].m;
    s.perm = perm;
    s.start = start;
    s.Cm = e->ds[which].m ? e->ds[which].mc : e->out_c;
    return s;
}

int loss_forward(unet_engine* e, int which, const int32_t* perm, int64_t start, int B, int slot, bool want_grad) {
    const ConvLayer& L = e->dec.back();
    const int C = e->out_c, HW = e->out_h * e->out_w;
    double* ls = reinterpret_cast<double*>(e->ws + e->off_ls);
    UHIP_TRY(hipMemsetAsync(ls, 0, (size_t)B * C * 8 * sizeof(double), e->stream));
    const LossSrc src = loss_src(e, which, perm, start);
    // (seven workgroup-wide fp64 sums end every workgroup: about 768 of them, each a long slice of its plane)
    static const int ls_wgs = getenv("CAE_UNET_LSWGS") ? atoi(getenv("CAE_UNET_LSWGS")) : 768;   // env: tuning runs only
    int chunks = std::max(1, std::min(32, ls_wgs / std::max(1, B * C)));
    chunks = std::min(chunks, (HW + 1023) / 1024);
    hipLaunchKernelGGL(k_loss_sums, dim3(chunks, B * C), dim3(256), 0, e->stream, e->f(L.u), 1, src, C, HW, ls);
    double* out2 = reinterpret_cast<double*>(e->ws + e->off_losses) + 2 * (size_t)slot;
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, e->stream, ls, B, C, src.Cm, src.mask ? 1 : 0, e->lambda_p,
                       out2, want_grad ? e->f(e->coef) : nullptr);
    if (want_grad) {
        // ... and, in the same pass, the last layer's bias gradient (its sum over the batch and the map); backward() clears the
        // accumulator before anything else adds to it, so the clear is issued here, ahead of this launch
        if (!e->gacc_clean) {
            UHIP_TRY(hipMemsetAsync(e->gacc(0), 0, (size_t)e->n_params * sizeof(double), e->stream));
            for (int k = 0; k < 4; k++) e->fc_f32_dirty[k] = false;
            e->gacc_clean = true;
        }
        int chunks = std::max(1, std::min(64, 1536 / std::max(1, B * C)));
        chunks = std::min(chunks, (HW + 1023) / 1024);
        hipLaunchKernelGGL(k_loss_grad, dim3(std::max(1, chunks), B * C), dim3(256), 0, e->stream, e->f(L.u), src, B, C, HW, e->f(e->coef),
                           e->f(L.gu), e->gacc(L.b));
    }
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

// ---- backward ---------------------------------------------------------------------------------------
int backward(unet_engine* e, const float* x, int B) {
    const int n = (int)e->enc.size(), nd = (int)e->dec.size();
    if (!e->gacc_clean) {
        UHIP_TRY(hipMemsetAsync(e->gacc(0), 0, (size_t)e->n_params * sizeof(double), e->stream));
        for (int k = 0; k < 4; k++) e->fc_f32_dirty[k] = false;
    }
    e->gacc_clean = false;
    for (int j = nd - 1; j >= 0; j--) {
        ConvLayer& L = e->dec[j];
        Geom g = L.g;
        g.B = B;
        const int HWl = g.Hl * g.Wl, C = g.Cl;
        const float* din = j == 0 ? e->f(e->fc[3].a) : e->f(e->dec[j - 1].din_next);
        // ConvTranspose2d: S = input, L = output
        conv_wgrad(e, g, din, e->f(L.gu), e->gacc(L.w));
        // bias gradient: summed by k_scale_bwd when it produced gu, and by the loss kernel for the last layer
        conv_down(e, g, e->f(L.gu), e->P(L.w), nullptr, e->f(L.gdin));
        if (j == 0) break;
        // gdin is the gradient wrt dropout(relu(bn(cat_{j-1})))
        ConvLayer& Pv = e->dec[j - 1];
        const int Cp = Pv.g.Cl, HWp = Pv.g.Hl * Pv.g.Wl;
        // BatchNorm + ReLU + dropout backward over the (never written) concatenated tensor: pass 1 sums; pass 2, one workgroup per
        // (b, c) plane, writes dz and - for the gated half - the gate's gradient da[b][c] = sum dz * u in the same pass.  The first
        // half of gcat goes on through the attention gate to u, the second half into the encoder skip (read in place later)
        {
            const Drop dd = make_drop(e, SITE_DEC_CONV + (j - 1), true);
            const ZCat zc{e->f(Pv.u), e->f(Pv.att), e->f(e->enc[n - 2 - (j - 1)].s), Cp};
            const Bn& bn = Pv.bn;
            hipLaunchKernelGGL(k_bn_bwd_reduce, red_grid(B, bn.C, HWp), dim3(256), 0, e->stream, e->f(L.gdin), (long long)2 * Cp * HWp,
                               (const float*)nullptr, 0LL, (const float*)nullptr, 0LL, B, bn.C, HWp, e->f(bn.saved), e->P(bn.gamma),
                               e->P(bn.beta), dd, (float*)nullptr, e->dsum(bn.bsums), zc);
            hipLaunchKernelGGL(k_bn_bwd_apply2_planes, dim3(1, bn.C, B), dim3(256), 0, e->stream, e->f(L.gdin), (long long)2 * Cp * HWp,
                               bn.C, HWp, e->f(bn.saved), e->P(bn.gamma), e->P(bn.beta), dd, e->dsum(bn.bsums), (double)B * HWp,
                               e->gacc(bn.gamma), e->gacc(bn.beta), e->f(Pv.gcat), zc, e->f(Pv.da));
        }
        hipLaunchKernelGGL(k_att_bwd, dim3(B), dim3(256), (size_t)(3 * Cp + 4 * Pv.R) * sizeof(float), e->stream,
                           e->f(Pv.pool), e->f(Pv.att), e->f(Pv.hid), e->f(Pv.da), Cp, Pv.R, e->P(Pv.w1), e->P(Pv.w2),
                           e->gacc(Pv.w1), e->gacc(Pv.w2), e->f(Pv.dpool));
        // ... and accumulates the previous layer's bias gradient while it writes du
        hipLaunchKernelGGL(k_scale_bwd, red_grid(B, Cp, HWp), dim3(256), 0, e->stream, e->f(Pv.gcat), e->f(Pv.att),
                           e->f(Pv.pool), e->f(Pv.dpool), B, Cp, HWp, e->f(Pv.gu), e->gacc(Pv.b));
    }
    // decoder_lin / encoder_lin backward
    const uint32_t sites[4] = {SITE_ENC_FC0, SITE_ENC_FC1, SITE_DEC_FC0, SITE_DEC_FC1};
    const float* gin = e->f(e->dec[0].gdin);   // grad wrt fc[3].a  (B, nout3)
    for (int k = 3; k >= 0; k--) {
        Fc& L = e->fc[k];
        const Drop d = make_drop(e, sites[k], true);
        if (L.has_bn) {
            bn_backward(e, L.bn, gin, L.nout, nullptr, 0, e->f(L.h), L.nout, B, 1, d, e->f(L.gh));
        } else {
            hipLaunchKernelGGL(k_relu_drop_bwd, dim3(blocks_for((long long)B * L.nout)), dim3(256), 0, e->stream, e->f(L.gh), gin,
                               e->f(L.h), (long long)B * L.nout, d);
        }
        const float* in = k == 0 ? (make_drop(e, SITE_ENC_CONV + n - 1, true).on ? e->f(e->enc[n - 1].a) : e->f(e->enc[n - 1].s))
                                 : e->f(e->fc[k - 1].a);
        lin_bwd(e, L, B, in, e->f(L.gh), e->f(L.ga));
        gin = e->f(L.ga);
    }
    // encoder backward; gin = grad wrt a_{n-1} (B, F)
    for (int i = n - 1; i >= 0; i--) {
        ConvLayer& L = e->enc[i];
        Geom g = L.g;
        g.B = B;
        const int HW = g.Hs * g.Ws, C = g.Cs;
        // gradient of the skip use of s_i (decoder layer n-2-i), read in place from that layer's gcat
        const float* gskip = nullptr;
        long long gskip_bs = 0;
        if (i < n - 1) {
            const ConvLayer& D = e->dec[n - 2 - i];
            gskip = e->f(D.gcat) + (size_t)C * HW;
            gskip_bs = (long long)2 * C * HW;
        }
        bn_backward(e, L.bn, gin, (long long)C * HW, gskip, gskip_bs, e->f(L.z), (long long)C * HW, B, HW,
                    make_drop(e, SITE_ENC_CONV + i, true), e->f(L.gz));
        const float* in = i == 0 ? x : (make_drop(e, SITE_ENC_CONV + i - 1, true).on ? e->f(e->enc[i - 1].a) : e->f(e->enc[i - 1].s));
        // Conv2d: S = output, L = input
        conv_wgrad(e, g, e->f(L.gz), in, e->gacc(L.w));
        // no bias gradient: this bias is added right before a BatchNorm, whose backward makes every channel of dz sum
        // to zero (the reference's autograd returns rounding noise of ~1e-8 here)
        if (i > 0) {
            conv_up(e, g, e->f(L.gz), e->P(L.w), L.wp >= 0 ? e->f(L.wp) : nullptr, nullptr, e->f(e->enc[i - 1].ga));
            gin = e->f(e->enc[i - 1].ga);
        }
    }
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int check_batch(unet_engine* e, int which, const int32_t* perm, int64_t start, int batch, int slot) {
    if (!e || !e->ws) return ufail(CAE_ERR_STATE, "unet: engine is not bound");
    if (which < 0 || which > 1 || !e->ds[which].x) return ufail(CAE_ERR_STATE, "unet: data set %d is not set", which);
    if (batch < 1 || batch > e->max_batch) return ufail(CAE_ERR_ARG, "unet: batch %d outside 1..%d", batch, e->max_batch);
    if (start < 0 || start + batch > e->ds[which].n)
        return ufail(CAE_ERR_ARG, "unet: samples %lld..%lld outside the data set (%lld)", (long long)start,
                     (long long)(start + batch), (long long)e->ds[which].n);
    if (slot < 0 || slot >= kLossSlots) return ufail(CAE_ERR_ARG, "unet: loss slot %d outside 0..%d", slot, kLossSlots - 1);
    (void)perm;
    return CAE_OK;
}

int gather_x(unet_engine* e, int which, const int32_t* perm, int64_t start, int B) {
    const long long E = (long long)e->in_c * e->in_h * e->in_w;
    hipLaunchKernelGGL(k_gather, dim3(blocks_for((long long)B * E, 65536)), dim3(256), 0, e->stream, e->ds[which].x, perm,
                       (long long)start, B, E, e->f(e->xb));
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

AdamwConsts adamw_consts(const unet_engine* e) {
    const Hyper& h = e->hyper;
    const double bc1 = 1.0 - pow(h.beta1, (double)e->step), bc2 = 1.0 - pow(h.beta2, (double)e->step);
    return AdamwConsts{(float)(h.lr / bc1), (float)sqrt(bc2), (float)(1.0 - h.lr * h.wd), (float)h.beta1, (float)h.beta2, (float)h.eps};
}
int adamw_blocks(const unet_engine* e) { return blocks_for((e->n_params + 3) / 4, 2048); }

F32Ranges f32_ranges(const unet_engine* e) {
    F32Ranges fr;
    memset(&fr, 0, sizeof fr);
    for (int k = 0; k < 4; k++)
        if (e->fc_f32[k]) fr.lo[fr.n] = e->fc[k].w, fr.hi[fr.n] = e->fc[k].w + (long long)e->fc[k].nin * e->fc[k].nout, fr.n++;
    return fr;
}

int train_or_fb(unet_engine* e, int which, const int32_t* perm, int64_t start, int batch, int slot, float* grads_out,
                double grad_scale = 1.0) {
    int rc = check_batch(e, which, perm, start, batch, slot);
    if (rc) return rc;
    if ((rc = gather_x(e, which, perm, start, batch))) return rc;
    if ((rc = forward(e, e->f(e->xb), batch, true))) return rc;
    if ((rc = loss_forward(e, which, perm, start, batch, slot, true))) return rc;
    if ((rc = backward(e, e->f(e->xb), batch))) return rc;
    if (grads_out) {
        hipLaunchKernelGGL(k_acc_to_f32, dim3(blocks_for(e->n_params, 65536)), dim3(256), 0, e->stream, (long long)e->n_params,
                           e->gacc(0), grads_out, grad_scale, f32_ranges(e));
    } else {
        e->step += 1;
        hipLaunchKernelGGL(k_adamw, dim3(adamw_blocks(e)), dim3(256), 0, e->stream, (long long)e->n_params, e->params, e->gacc(0), e->m,
                           e->v, adamw_consts(e), f32_ranges(e));
        e->gacc_clean = true;
    }
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // namespace

extern "C" {

int unet_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                       int latent_size, int max_batch, unet_engine** out) {
    if (!enc || !dec || !out || n_enc < 1 || n_dec < 1 || fc_size < 1 || latent_size < 1 || max_batch < 1)
        return ufail(CAE_ERR_ARG, "unet_engine_create: bad argument");
    if (n_enc != n_dec)
        return ufail(CAE_ERR_ARG, "unet_engine_create: the UNET decoder needs one layer per encoder layer (%d vs %d): every "
                                  "decoder layer but the last takes one skip connection (unet.py:141-145,157-161)", n_enc, n_dec);
    unet_engine* e = new unet_engine();
    e->fc_size = fc_size;
    e->latent = latent_size;
    e->max_batch = max_batch;
    auto bad = [&](const char* msg, int i) {
        const int rc = ufail(CAE_ERR_ARG, "unet_engine_create: layer %d: %s", i, msg);
        delete e;
        return rc;
    };
    // ---- geometry --------------------------------------------------------------------------------
    for (int i = 0; i < n_enc; i++) {
        const cae_layer_spec& l = enc[i];
        ConvLayer L;
        L.g = Geom{0, l.out_c, l.out_h, l.out_w, l.in_c, l.in_h, l.in_w, l.k_h, l.k_w, l.stride, l.output_padding};
        if (l.stride < 1 || l.k_h < 1 || l.k_w < 1 || l.output_padding < 0) return bad("bad kernel / stride / padding", i);
        if ((l.in_h + 2 * l.output_padding - l.k_h) / l.stride + 1 != l.out_h ||
            (l.in_w + 2 * l.output_padding - l.k_w) / l.stride + 1 != l.out_w)
            return bad("encoder output size does not follow from Conv2d(kernel, stride, padding)", i);
        if (i > 0 && (enc[i - 1].out_c != l.in_c || enc[i - 1].out_h != l.in_h || enc[i - 1].out_w != l.in_w))
            return bad("encoder input does not match the previous layer's output", i);
        L.has_bn = true;
        e->enc.push_back(L);
    }
    for (int j = 0; j < n_dec; j++) {
        const cae_layer_spec& l = dec[j];
        ConvLayer L;
        L.g = Geom{0, l.in_c, l.in_h, l.in_w, l.out_c, l.out_h, l.out_w, l.k_h, l.k_w, l.stride, l.output_padding};
        if (l.stride < 1 || l.k_h < 1 || l.k_w < 1 || l.output_padding < 0) return bad("bad kernel / stride / padding", j);
        if ((l.in_h - 1) * l.stride - 2 * l.output_padding + l.k_h != l.out_h ||
            (l.in_w - 1) * l.stride - 2 * l.output_padding + l.k_w != l.out_w)
            return bad("decoder output size does not follow from ConvTranspose2d(kernel, stride, padding)", j);
        L.has_bn = L.has_skip = j != n_dec - 1;
        if (L.has_skip) {
            const cae_layer_spec& s = enc[n_enc - 2 - j];
            if (s.out_c != l.out_c || s.out_h != l.out_h || s.out_w != l.out_w)
                return bad("decoder output does not match the encoder skip it is concatenated with", j);
            if (l.out_c < 8) return bad("ChannelAttention needs at least 8 channels (in_planes // 8 hidden units)", j);
            L.R = l.out_c / 8;
            if (dec[j + 1].in_c != 2 * l.out_c || dec[j + 1].in_h != l.out_h || dec[j + 1].in_w != l.out_w)
                return bad("the next decoder layer must take 2 x out_channels (skip concat) at this layer's size", j);
        }
        e->dec.push_back(L);
    }
    e->in_c = enc[0].in_c, e->in_h = enc[0].in_h, e->in_w = enc[0].in_w;
    e->out_c = dec[n_dec - 1].out_c, e->out_h = dec[n_dec - 1].out_h, e->out_w = dec[n_dec - 1].out_w;
    const int F = enc[n_enc - 1].out_c * enc[n_enc - 1].out_h * enc[n_enc - 1].out_w;
    const int G = dec[0].in_c * dec[0].in_h * dec[0].in_w;
    // ---- tensor table, in the reference's state_dict order --------------------------------------------
    for (int i = 0; i < n_enc; i++) {
        ConvLayer& L = e->enc[i];
        const std::string c = "enc/encoder_cnn." + std::to_string(4 * i), b = "enc/encoder_cnn." + std::to_string(4 * i + 1);
        L.w = add_tensor(e, c + ".weight", 0, {L.g.Cs, L.g.Cl, L.g.kh, L.g.kw});
        L.b = add_tensor(e, c + ".bias", 0, {L.g.Cs});
        add_bn(e, b, L.g.Cs, L.bn);
    }
    auto add_fc = [&](Fc& L, const std::string& key, int nin, int nout) {
        L.nin = nin;
        L.nout = nout;
        L.w = add_tensor(e, key + ".weight", 0, {nout, nin});
        L.b = add_tensor(e, key + ".bias", 0, {nout});
    };
    add_fc(e->fc[0], "enc/encoder_lin.0", F, fc_size);
    e->fc[0].has_bn = true;
    add_bn(e, "enc/encoder_lin.1", fc_size, e->fc[0].bn);
    add_fc(e->fc[1], "enc/encoder_lin.4", fc_size, latent_size);
    add_fc(e->fc[2], "dec/decoder_lin.0", latent_size, fc_size);
    e->fc[2].has_bn = true;
    add_bn(e, "dec/decoder_lin.1", fc_size, e->fc[2].bn);
    add_fc(e->fc[3], "dec/decoder_lin.4", fc_size, G);
    for (int j = 0; j < n_dec; j++) {
        ConvLayer& L = e->dec[j];
        if (!L.has_skip) continue;
        const std::string a = "dec/attention_layers." + std::to_string(j);
        L.w1 = add_tensor(e, a + ".fc1.weight", 0, {L.R, L.g.Cl, 1, 1});
        L.w2 = add_tensor(e, a + ".fc2.weight", 0, {L.g.Cl, L.R, 1, 1});
    }
    for (int j = 0; j < n_dec; j++) {
        ConvLayer& L = e->dec[j];
        const std::string c = "dec/decoder_conv." + std::to_string(4 * j), b = "dec/decoder_conv." + std::to_string(4 * j + 1);
        L.w = add_tensor(e, c + ".weight", 0, {L.g.Cs, L.g.Cl, L.g.kh, L.g.kw});
        L.b = add_tensor(e, c + ".bias", 0, {L.g.Cl});
        if (L.has_bn) add_bn(e, b, 2 * L.g.Cl, L.bn);
    }
    e->n_params = (e->n_params + 3) & ~int64_t(3);
    e->n_buffers = (e->n_buffers + 3) & ~int64_t(3);
    // ---- workspace -------------------------------------------------------------------------------------
    const int64_t B = max_batch;
    int64_t nd = 0;   // doubles: BN sums
    auto carve_bn = [&](Bn& bn) {
        bn.sums = nd;
        nd += 2 * bn.C;
        bn.bsums = nd;
        nd += 2 * bn.C;
    };
    for (auto& L : e->enc) carve_bn(L.bn);
    carve_bn(e->fc[0].bn);
    carve_bn(e->fc[2].bn);
    for (auto& L : e->dec)
        if (L.has_bn) carve_bn(L.bn);
    e->n_dsum = nd;
    int64_t nf = 0;   // floats
    auto F32 = [&](int64_t n) {
        const int64_t off = nf;
        nf += (n + 63) & ~int64_t(63);
        return off;
    };
    auto carve_saved = [&](Bn& bn) { bn.saved = F32(2 * bn.C); };
    e->xb = F32(B * e->in_c * e->in_h * e->in_w);
    for (auto& L : e->enc) {
        const int64_t n = B * L.g.Cs * L.g.Hs * L.g.Ws;
        L.z = F32(n), L.s = F32(n), L.a = F32(n), L.gz = F32(n), L.ga = F32(n);
        carve_saved(L.bn);
    }
    for (int k = 0; k < 4; k++) {
        Fc& L = e->fc[k];
        const int64_t n = B * L.nout;
        L.h = F32(n), L.a = F32(n), L.gh = F32(n);
        L.ga = F32(B * L.nin);
        if (L.has_bn) carve_saved(L.bn);
    }
    for (auto& L : e->dec) {
        const int64_t nl = B * L.g.Cl * L.g.Hl * L.g.Wl, ns = B * L.g.Cs * L.g.Hs * L.g.Ws;
        L.u = F32(nl), L.gu = F32(nl), L.gdin = F32(ns);
        if (L.has_skip) {
            L.din_next = F32(2 * nl), L.gcat = F32(2 * nl);
            L.pool = F32(3 * B * L.g.Cl), L.psum = F32(4 * B * L.g.Cl), L.hid = F32(2 * B * L.R), L.att = F32(B * L.g.Cl);
            L.dpool = F32(2 * B * L.g.Cl), L.da = F32(B * L.g.Cl);
            carve_saved(L.bn);
        }
    }
    e->coef = F32(4 * B * e->out_c);
    // weights repacked per output parity for the layers that run OpUp (kernels_unet_mfma.h): decoder forward, encoder dgrad
    auto carve_packed = [&](ConvLayer& L) {
        Geom g = L.g;
        g.B = (int)B;
        if ((mfma_up_eligible(g) && g.Cl > 4) || thin_up_geom(g)) L.wp = F32((int64_t)16 * g.Cs * g.Cl);
    };
    for (auto& L : e->dec) carve_packed(L);
    for (size_t i = 1; i < e->enc.size(); i++) carve_packed(e->enc[i]);
    int64_t off = 0;
    auto bytes = [&](int64_t n) {
        const int64_t o = off;
        off += (n + 255) & ~int64_t(255);
        return o;
    };
    e->off_gacc = bytes(e->n_params * 8);
    e->off_dsum = bytes(nd * 8);
    e->off_losses = bytes((int64_t)kLossSlots * 2 * 8);
    e->off_ls = bytes(B * e->out_c * 8 * 8);
    int64_t gs = 0;   // split-K scratch of the Linear GEMMs: rows x batch doubles, kept zero between uses
    for (int k = 0; k < 4; k++) gs = std::max<int64_t>(gs, (int64_t)std::max(e->fc[k].nin, e->fc[k].nout) * B);
    e->off_gscratch = bytes(gs * 8);
    e->gscratch_bytes = gs * 8;
    int64_t lp = 0;   // partial tiles of the big layers' K slices (128 wide), batch <= 64
    for (int k = 0; k < 4; k++) {
        const Fc& L = e->fc[k];
        if (small_fc(L)) continue;
        lp = std::max<int64_t>(lp, std::max<int64_t>((L.nin + 127) / 128 * (int64_t)L.nout, (L.nout + 127) / 128 * (int64_t)L.nin) * std::min<int64_t>(B, 64) * 4);
    }
    e->off_linpart = bytes(lp);
    e->linpart_bytes = lp;
    int64_t tp = 0;
    auto thin_need = [&](const ConvLayer& L) {
        Geom g = L.g;
        g.B = (int)B;
        if (thin_geom(g)) tp = std::max<int64_t>(tp, (int64_t)thin_wgrad_part_bytes(g));
        else if (mfma_wgrad_eligible(g)) tp = std::max<int64_t>(tp, (int64_t)mfma_wgrad_part_bytes(g));
        if (pwgrad_geom(g)) tp = std::max<int64_t>(tp, (int64_t)pwgrad_part_bytes(g));
    };
    for (auto& L : e->enc) thin_need(L);
    for (auto& L : e->dec) thin_need(L);
    e->off_thinpart = bytes(tp);
    e->thinpart_bytes = tp;
    e->off_f32 = bytes(nf * 4);
    e->ws_bytes = off;
    *out = e;
    return CAE_OK;
}

void unet_engine_destroy(unet_engine* e) { delete e; }

int64_t unet_param_count(const unet_engine* e) { return e ? e->n_params : 0; }
int64_t unet_buffer_count(const unet_engine* e) { return e ? e->n_buffers : 0; }
int unet_tensor_count(const unet_engine* e) { return e ? (int)e->tensors.size() : 0; }
int unet_tensor_info(const unet_engine* e, int index, cae_tensor_info_t* out) {
    if (!e || !out || index < 0 || index >= (int)e->tensors.size()) return ufail(CAE_ERR_ARG, "unet_tensor_info: bad argument");
    *out = e->tensors[index];
    return CAE_OK;
}
int64_t unet_workspace_bytes(const unet_engine* e) { return e ? e->ws_bytes : 0; }

int unet_bind(unet_engine* e, float* params, float* m, float* v, float* buffers, void* workspace, int64_t workspace_bytes) {
    if (!e || !params || !m || !v || !buffers || !workspace) return ufail(CAE_ERR_ARG, "unet_bind: null pointer");
    if (workspace_bytes < e->ws_bytes) return ufail(CAE_ERR_ARG, "unet_bind: workspace of %lld bytes, need %lld",
                                                     (long long)workspace_bytes, (long long)e->ws_bytes);
    if ((uintptr_t)workspace & 255) return ufail(CAE_ERR_ARG, "unet_bind: workspace must be 256-byte aligned");
    e->params = params, e->m = m, e->v = v, e->buffers = buffers, e->ws = (char*)workspace;
    return CAE_OK;
}

int unet_set_stream(unet_engine* e, void* hip_stream) {
    if (!e) return ufail(CAE_ERR_ARG, "unet_set_stream: null engine");
    e->stream = (hipStream_t)hip_stream;
    return CAE_OK;
}

int unet_set_kernel_mode(unet_engine* e, int specialised) {
    if (!e) return ufail(CAE_ERR_ARG, "unet_set_kernel_mode: null engine");
    e->specialised = specialised ? 1 : 0;
    return CAE_OK;
}

int unet_set_hyper(unet_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay, double dropout_rate,
                   double lambda_pearson, uint32_t dropout_seed) {
    if (!e) return ufail(CAE_ERR_ARG, "unet_set_hyper: null engine");
    if (!(dropout_rate >= 0.0 && dropout_rate < 1.0)) return ufail(CAE_ERR_ARG, "unet_set_hyper: dropout_rate must be in [0, 1)");
    e->hyper = Hyper{lr, beta1, beta2, eps, weight_decay};
    e->dropout = dropout_rate;
    e->lambda_p = lambda_pearson;
    e->seed = dropout_seed;
    return CAE_OK;
}

int unet_set_step(unet_engine* e, int64_t step) {
    if (!e || step < 0) return ufail(CAE_ERR_ARG, "unet_set_step: bad argument");
    e->step = step;
    return CAE_OK;
}

int unet_set_dataset(unet_engine* e, int which, const float* x, const float* target, const float* mask, int mask_channels,
                     int64_t n) {
    if (!e || which < 0 || which > 1 || !x || n < 1) return ufail(CAE_ERR_ARG, "unet_set_dataset: bad argument");
    if (mask && mask_channels != 1 && mask_channels != e->out_c)
        return ufail(CAE_ERR_ARG, "unet_set_dataset: mask has %d channels, expected 1 or %d", mask_channels, e->out_c);
    e->ds[which] = DataSet{x, target, mask, mask_channels, n};
    return CAE_OK;
}

int unet_train_step(unet_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    if (e && !e->ds[which & 1].t) return ufail(CAE_ERR_STATE, "unet_train_step: data set has no target");
    return train_or_fb(e, which, perm, start, batch, loss_slot, nullptr);
}

int unet_forward_backward(unet_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot,
                          float* grads, double grad_scale) {
    if (!grads) return ufail(CAE_ERR_ARG, "unet_forward_backward: null gradient buffer");
    if (e && !e->ds[which & 1].t) return ufail(CAE_ERR_STATE, "unet_forward_backward: data set has no target");
    return train_or_fb(e, which, perm, start, batch, loss_slot, grads, grad_scale);
}

int unet_apply_gradients(unet_engine* e, const float* grads) {
    if (!e || !e->ws || !grads) return ufail(CAE_ERR_ARG, "unet_apply_gradients: bad argument");
    hipLaunchKernelGGL(k_f32_to_acc, dim3(blocks_for(e->n_params, 65536)), dim3(256), 0, e->stream, (long long)e->n_params, grads,
                       e->gacc(0));
    e->step += 1;
    hipLaunchKernelGGL(k_adamw, dim3(adamw_blocks(e)), dim3(256), 0, e->stream, (long long)e->n_params, e->params, e->gacc(0), e->m,
                       e->v, adamw_consts(e), F32Ranges{{0, 0, 0, 0}, {0, 0, 0, 0}, 0});
    e->gacc_clean = true;
    for (int k = 0; k < 4; k++) e->fc_f32[k] = e->fc_f32_dirty[k] = false;   // every slot was rewritten as fp64 and cleared
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int unet_eval_step(unet_engine* e, int which, const int32_t* perm, int64_t start, int batch, int loss_slot) {
    int rc = check_batch(e, which, perm, start, batch, loss_slot);
    if (rc) return rc;
    if (!e->ds[which].t) return ufail(CAE_ERR_STATE, "unet_eval_step: data set has no target");
    if ((rc = gather_x(e, which, perm, start, batch))) return rc;
    if ((rc = forward(e, e->f(e->xb), batch, false))) return rc;
    return loss_forward(e, which, perm, start, batch, loss_slot, false);
}

int unet_score(unet_engine* e, const float* x, int batch, float* y) {
    if (!e || !e->ws) return ufail(CAE_ERR_STATE, "unet_score: engine is not bound");
    if (!x || !y || batch < 1 || batch > e->max_batch) return ufail(CAE_ERR_ARG, "unet_score: bad argument");
    int rc = forward(e, x, batch, false);
    if (rc) return rc;
    const long long n = (long long)batch * e->out_c * e->out_h * e->out_w;
    hipLaunchKernelGGL(k_sigmoid, dim3(blocks_for(n, 65536)), dim3(256), 0, e->stream, e->f(e->dec.back().u), n, y);
    UHIP_TRY(hipGetLastError());
    return CAE_OK;
}

int unet_loss_slots(const unet_engine* e) { return e ? kLossSlots : 0; }

int unet_read_losses(unet_engine* e, int first_slot, int count, double* out) {
    if (!e || !e->ws || !out || first_slot < 0 || count < 0 || first_slot + count > kLossSlots)
        return ufail(CAE_ERR_ARG, "unet_read_losses: bad argument");
    UHIP_TRY(hipMemcpyAsync(out, reinterpret_cast<double*>(e->ws + e->off_losses) + 2 * (size_t)first_slot,
                            (size_t)count * 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    UHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

int unet_sync(unet_engine* e) {
    if (!e) return ufail(CAE_ERR_ARG, "unet_sync: null engine");
    UHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

int unet_debug_read(unet_engine* e, const char* what, float* out, int64_t count) {
    if (!e || !e->ws || !what || !out || count < 1) return ufail(CAE_ERR_ARG, "unet_debug_read: bad argument");
    std::map<std::string, int64_t> table;
    for (size_t i = 0; i < e->enc.size(); i++) {
        const std::string k = std::to_string(i);
        table["enc_z" + k] = e->enc[i].z, table["enc_s" + k] = e->enc[i].s, table["enc_a" + k] = e->enc[i].a;
        table["enc_gz" + k] = e->enc[i].gz;
    }
    for (size_t j = 0; j < e->dec.size(); j++) {
        const std::string k = std::to_string(j);
        table["dec_u" + k] = e->dec[j].u, table["dec_gu" + k] = e->dec[j].gu, table["dec_gdin" + k] = e->dec[j].gdin;
        if (e->dec[j].has_skip)
            table["att" + k] = e->dec[j].att, table["dec_din" + std::to_string(j + 1)] = e->dec[j].din_next,
            table["dec_gcat" + k] = e->dec[j].gcat;
    }
    for (int k = 0; k < 4; k++) table["fc_h" + std::to_string(k)] = e->fc[k].h, table["fc_a" + std::to_string(k)] = e->fc[k].a;
    auto it = table.find(what);
    if (it == table.end()) return ufail(CAE_ERR_ARG, "unet_debug_read: unknown tensor '%s'", what);
    UHIP_TRY(hipMemcpyAsync(out, e->f(it->second), (size_t)count * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    UHIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

}  // extern "C"
