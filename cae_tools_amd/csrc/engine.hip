// engine.hip — host side of libcae_hip.so: launch plan, workspace carving, hipGraph capture
// and the extern "C" ABI declared in include/cae_hip.h.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "cae_hip.h"
#include "kernels_generic.h"
#include "kernels_s2.h"
#include "kernels_last.h"
#include "kernels_rows.h"
#include "kernels_gemm.h"
#include "kernels_igemm.h"
#include "kernels_ctlds.h"
#include "kernels_head.h"
#include "dp_comm.h"
#include "kernels_ctbwd.h"   // argument struct only: the kernel lives in ctbwd.hip
#include "trunk_api.h"

// The benchmark geometry's template kernels, instantiated HERE so that their code sits next to the non-template kernels of
// the same step (implicit instantiations are emitted at the end of the 3 MB text section, 2 MB away): see DESIGN.md §4.
namespace cae {
template __global__ void k_ct_fwd_lds<3, 3>(CtFwd);
template __global__ void k_s2_fwd_rows<8, 4, 1, 2>(S2FwdRows);
template __global__ void k_s2_fwd_rows<4, 2, 2, 1>(S2FwdRows);
template __global__ void k_s2_last_fused<2, 1, 4, 4, 4, true, true>(S2Last);
template __global__ void k_s2_bwd_rows<4, 4, 2, 3, 3, 2, 1, 3>(S2Rows);
template __global__ void k_s2_bwd_rows<8, 2, 4, 3, 3, 4, 2, 1>(S2Rows);
}  // namespace cae

using namespace cae;

namespace cae_internal {
int ctbwd_launch(const void* args, size_t bytes, unsigned gx, unsigned gy, unsigned gz, size_t lds, hipStream_t s);   // ctbwd.hip
}

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// shared with unet_engine.hip: sets the message cae_last_error() returns for the calling thread
void cae_detail_set_error(const char* msg) { g_err = msg; }

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(CAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr float kBnEps = 1e-5f;      // nn.BatchNorm2d default (encoder.py:45, decoder.py:47)
constexpr float kBnMomentum = 0.1f;  // nn.BatchNorm2d default
constexpr int kLossSlots = 1 << 16;

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct ConvLayer {
    bool transposed;
    int cin, hin, win, cout, hout, wout, kh, kw, stride, opad;
    bool has_bn;
    int bn_index;  // index into the BN tables, -1 when has_bn is false
    int64_t w_off, b_off, gamma_off, beta_off;  // parameter arena (floats)
    int64_t rm_off, rv_off;                     // buffer arena (floats)
    int64_t act_off, grad_off;                  // workspace (bytes): raw output y / masked gradient g
    int sh_w = -1, sh_b = -1;                   // offsets in the sharded gradient table (-1: not sharded)
    int64_t out_elems() const { return (int64_t)cout * hout * wout; }
    int64_t in_elems() const { return (int64_t)cin * hin * win; }
};

struct FcLayer {
    int nin, nout;
    int64_t w_off, b_off;
    int64_t act_off, grad_off;  // output activation / gradient wrt pre-activation of the output
    bool relu;
};

}  // namespace

struct cae_engine {
    std::vector<ConvLayer> enc, dec;
    FcLayer fc[4];  // encoder_lin.0, encoder_lin.2, decoder_lin.0, decoder_lin.2
    int fc_size = 0, latent = 0, max_batch = 0;
    int in_c = 0, in_h = 0, in_w = 0, out_c = 0, out_h = 0, out_w = 0;
    std::vector<cae_tensor_info_t> tensors;
    int64_t n_param = 0, n_buf = 0;
    int n_bn = 0;
    std::vector<int64_t> bn_stat_off;   // per BN: byte offset of its [C][4] double sums
    std::vector<int64_t> bn_saved_off;  // per BN: byte offset of its [C][2] float mean/invstd
    std::vector<int> bn_channels;
    int max_channels = 0;

    // workspace carve (byte offsets)
    int64_t off_state = 0, off_losses = 0, off_zero_begin = 0, off_gradacc = 0, off_zero_end = 0;
    int64_t off_glast = 0, off_scan = 0, off_sgacc = 0;
    ShardSegs segs{};                      // sharded gradient accumulators (thin stride-2 layers)
    int64_t ws_need = 0;

    // bound memory
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *bufs = nullptr;
    char* ws = nullptr;
    hipStream_t stream = nullptr;
    Hyper hp{1e-3, 0.9, 0.999, 1e-8, 1e-5};
    const float* ds_x[2] = {nullptr, nullptr};
    const float* ds_t[2] = {nullptr, nullptr};
    int64_t ds_n[2] = {0, 0};
    bool graph_mode = true;
    bool capture_only = false;   // cae_set_capture_only: step calls capture + cache their graph and launch nothing
    bool use_s2 = true;  // specialised stride-2 kernels (cae_set_kernel_mode)
    // trunk of the 'var' model (trunk_api.h): fc[1] is the pair of heads [mu | logvar] (2 * latent outputs), z = reparam(heads)
    // feeds fc[2]; the loss lives outside, so the last decoder layer can hand out its raw output and take its gradient
    bool variational = false;
    int64_t off_vz = 0, off_vgz = 0, off_zlast = 0;
    cae_internal::TrunkHooks hooks{nullptr, nullptr, nullptr};
    bool gather_fwd = false;   // cae_set_kernel_mode bit 2: channel-rich decoder layers' forward on the gather kernel k_ig_fwd_s2
    int ctbwd_mask = 0;  // bit l: decoder layer l's backward runs the LDS-staged kernel (kernels_ctbwd.h) where eligible
    int ctbwd_auto = 0;  // ... the mask chosen at creation (CAE_CTBWD, or the rule in cae_create): its layers have sharded accumulators
    int64_t off_xbatch = 0;     // the current batch's inputs, contiguous (written by k_head_fwd, read by k_adam's fused conv-0 weight gradient)
    bool x_published = false;   // this step's k_head_fwd wrote them
    AdamConv0 c0_pending{};     // filled by launch_backward when the conv-0 weight gradient is left to k_adam
    // profiling (cae_profile_begin/end): every launch bracketed by an event pair, plain launches
    bool profiling = false;
    struct ProfRec { const char* name; int layer; double bytes; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof;
    // key: (op, which, batch, global_batch, perm, nsteps, cursor_inc, BatchNorm mode = (dp_sync, bn_batch, world)): everything
    // a captured launch sequence bakes in that is not engine-wide state (engine-wide changes call drop_graphs())
    std::map<std::tuple<int, int, int, int, const void*, int, int, int, int, int>, hipGraphExec_t> graphs;

    // data-parallel state (cae_dp_init): one RCCL communicator, a second stream for the gradient buckets, fork/join events
    int dp_world = 0, dp_rank = 0;
    RcclApi::comm_t dp_comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool dp_graph_ok = true;            // RCCL calls captured inside the step graph (checked by cae_dp_init's self-test)
    bool dp_overlap = true;             // first gradient bucket on the second stream (cae_dp_set_overlap)
    std::vector<int> sync_order;        // BatchNorm tables in the order a SyncBN step all-reduces them
    size_t sync_pos = 0;
    int64_t bucket_split = 0;           // gradient buckets: [bucket_split, n_param) first (decoder), [0, bucket_split) last

    StepState* state() const { return reinterpret_cast<StepState*>(ws + off_state); }
    double* losses() const { return reinterpret_cast<double*>(ws + off_losses); }
    double* gradacc() const { return reinterpret_cast<double*>(ws + off_gradacc); }
    double* sgacc() const { return reinterpret_cast<double*>(ws + off_sgacc); }
    ShardSegs shard_segs() const {
        ShardSegs r = segs;
        r.base = sgacc();
        return r;
    }
    double* bn_stats(int j) const { return reinterpret_cast<double*>(ws + bn_stat_off[j]); }
    float* bn_saved(int j) const { return reinterpret_cast<float*>(ws + bn_saved_off[j]); }
    float* fptr(int64_t off) const { return reinterpret_cast<float*>(ws + off); }
    void drop_graphs() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
    }
};

namespace {

void add_tensor(cae_engine* e, const std::string& name, int arena, std::vector<int64_t> shape, int64_t* off_out) {
    cae_tensor_info_t t;
    memset(&t, 0, sizeof t);
    snprintf(t.name, sizeof t.name, "%s", name.c_str());
    t.arena = arena;
    t.ndim = (int)shape.size();
    t.numel = 1;
    for (size_t i = 0; i < shape.size(); i++) {
        t.shape[i] = shape[i];
        t.numel *= shape[i];
    }
    int64_t& top = arena == 0 ? e->n_param : e->n_buf;
    top = align_up(top, 4);  // 16-byte aligned tensors
    t.offset = top;
    top += t.numel;
    *off_out = t.offset;
    e->tensors.push_back(t);
}

int64_t carve(int64_t& top, int64_t bytes) {
    top = align_up(top, 256);
    int64_t o = top;
    top += bytes;
    return o;
}

// ---- descriptor helpers -----------------------------------------------------------------------

BnDesc bn_none() {
    BnDesc d;
    memset(&d, 0, sizeof d);
    d.mode = BN_NONE;
    return d;
}

// BN descriptor of conv layer L's BatchNorm in `mode`; count = elements per channel
BnDesc bn_of(const cae_engine* e, const ConvLayer& L, int mode, double count, int update) {
    if (!L.has_bn) return bn_none();
    BnDesc d;
    memset(&d, 0, sizeof d);
    d.mode = mode;
    d.C = L.cout;
    d.stats = e->bn_stats(L.bn_index);
    d.gamma = e->params + L.gamma_off;
    d.beta = e->params + L.beta_off;
    d.rmean = e->bufs + L.rm_off;
    d.rvar = e->bufs + L.rv_off;
    d.saved = e->bn_saved(L.bn_index);
    d.count = count;
    d.inv_count = count > 0.0 ? 1.0 / count : 0.0;
    d.unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
    d.momentum = kBnMomentum;
    d.eps = kBnEps;
    d.update = update;
    return d;
}

Src src_plain(const float* p, int C, int H, int W) {
    Src s;
    memset(&s, 0, sizeof s);
    s.p = p;
    s.C = C;
    s.H = H;
    s.W = W;
    return s;
}

Epi epi_plain(float* out) {
    Epi e;
    memset(&e, 0, sizeof e);
    e.kind = EPI_PLAIN;
    e.out = out;
    return e;
}

size_t lds_bytes(int c1, int c2) { return 4 * sizeof(double) + (size_t)(c1 + c2 + 1) * sizeof(float4); }

int grid1(int64_t n) { return (int)((n + 255) / 256); }
size_t gemm_lds(int channels) { return (4 * 256 + 256) * sizeof(float) + (size_t)(channels + 1) * sizeof(float4); }

// positions per block for k_wgrad: aim for ~2048 blocks in total, at least 256 positions each
int wgrad_ppb(int64_t positions, int64_t nweights) {
    int64_t target_blocks = 4096;
    int64_t nsplit = target_blocks / (nweights > 0 ? nweights : 1);
    if (nsplit < 1) nsplit = 1;
    int64_t ppb = (positions + nsplit - 1) / nsplit;
    if (ppb < 256) ppb = 256;
    ppb = align_up(ppb, 256);
    return (int)ppb;
}

enum Op { OP_TRAIN = 1, OP_FWDBWD = 2, OP_EVAL = 3, OP_ADAM = 4, OP_DP_TRAIN = 5 };

StepTail step_tail_of(cae_engine* e, int batch_inc, int slot_inc) {
    StepTail t;
    memset(&t, 0, sizeof t);
    t.zero_extra = reinterpret_cast<double*>(e->ws + e->off_zero_begin);
    t.zero_extra_n = (e->off_gradacc - e->off_zero_begin) / (long long)sizeof(double);
    t.acc_rw = e->gradacc();
    t.shard_rw = e->sgacc();
    t.st = e->state();
    t.batch_inc = batch_inc;
    t.slot_inc = slot_inc;
    return t;
}

// Brackets one launch with HIP events on the engine's stream while profiling is on.
// `bytes` = algorithmic bytes of the launch: every operand tensor read once, every result written once.
struct ProfScope {
    cae_engine* e;
    hipStream_t st;
    int idx = -1;
    ProfScope(cae_engine* e_, const char* name, int layer, double bytes, hipStream_t on = nullptr) : e(e_) {
        st = on ? on : e->stream;
        if (!e->profiling) return;
        cae_engine::ProfRec r{name, layer, bytes, nullptr, nullptr};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        e->prof.push_back(r);
        idx = (int)e->prof.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(e->prof[idx].e1, st);
    }
};

inline double f4(double n) { return 4.0 * n; }

// ---- the step, as a sequence of launches on e->stream -------------------------------------------

struct StepArgs {
    int which;
    const int32_t* perm;
    int batch;
    int global_batch;      // loss normalisation: batch summed over data-parallel ranks
    int bn_batch;          // samples behind the BatchNorm sums (local batch; global under SyncBN)
    bool train;            // train-mode forward (+ backward)
    bool use_cursor;       // samples come from the dataset through the cursor
    const float* x_direct; // score(): explicit input batch
    float* yhat;           // eval: sigmoid output destination (may be nullptr)
    bool want_loss;        // eval: accumulate MSE into the loss slot
    // SyncBN (cae_forward_backward_sync): after every launch that completes a BatchNorm sum table the
    // caller's function sums that table over the data-parallel ranks
    cae_allreduce_fn sync_fn = nullptr;
    void* sync_user = nullptr;
    int world = 1;
    int nsteps = 1;        // consecutive steps of the same batch size in one captured graph
    // data-parallel step inside the library (cae_dp_train_step): gradient buckets all-reduced by RCCL on the second stream;
    // dp_sync additionally all-reduces every BatchNorm sum table in-stream (SyncBN) instead of calling sync_fn
    bool dp = false, dp_sync = false;
    int cursor_inc = -1;   // samples the cursor moves per step (-1: batch; the global batch under data parallelism)
    // module-level forward (cae_encode / cae_decode): 0 = the whole network, 1 = encoder only (x_direct -> z_out),
    // 2 = decoder only (z_in -> yhat); eval mode, per-layer launches
    bool external_loss = false;  // trunk mode: the last decoder layer writes its RAW output to off_zlast; its gradient arrives in off_glast
    bool adam_follows = false;   // OP_TRAIN on one device: k_adam is the next launch (it may take the first encoder layer's weight gradient)
    int part = 0;
    const float* z_in = nullptr;
    float* z_out = nullptr;
    bool syncing() const { return sync_fn != nullptr || dp_sync; }
    int inc() const { return cursor_inc >= 0 ? cursor_inc : batch; }
};

#define NCCL_TRY(expr)                                                                                            \
    do {                                                                                                          \
        int _r = (expr);                                                                                          \
        if (_r != 0) return fail(CAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

int sync_bn_table(cae_engine* e, const StepArgs& a, int bn_index) {
    if (!a.syncing()) return CAE_OK;
    const int64_t n = (int64_t)kStatShards * e->bn_channels[bn_index] * 4;
    if (a.dp_sync) {
        // every rank must issue the same collectives in the same order: the order is a property of the model (sync_order),
        // and a step that deviates from it is an error here rather than a hang over there
        if (e->sync_pos >= e->sync_order.size() || e->sync_order[e->sync_pos] != bn_index)
            return fail(CAE_ERR_STATE, "SyncBN: table %d all-reduced out of order (position %zu)", bn_index, e->sync_pos);
        e->sync_pos++;
        NCCL_TRY(rccl().AllReduce(e->bn_stats(bn_index), e->bn_stats(bn_index), (size_t)n, RcclApi::kFloat64, RcclApi::kSum,
                                  e->dp_comm, e->stream));
        return CAE_OK;
    }
    if (a.sync_fn(a.sync_user, e->bn_stats(bn_index), n) != 0)
        return fail(CAE_ERR_STATE, "the all-reduce callback failed for BatchNorm table %d", bn_index);
    return CAE_OK;
}

// ---- specialised stride-2 kernels (kernels_s2.h): dispatch on (Cin, Cout, kh, kw) ----------------

#define S2_SHAPES(X) X(2, 1) X(4, 2) X(8, 4) X(6, 3)
#define S2_KERNELS(X, CI, CO) X(CI, CO, 3, 3) X(CI, CO, 4, 4) X(CI, CO, 3, 4) X(CI, CO, 4, 3)

bool s2_shape_ok(const ConvLayer& L) {
    if (!L.transposed || L.stride != 2) return false;
    if (L.kh < 3 || L.kh > 4 || L.kw < 3 || L.kw > 4) return false;
#define CHK(CI, CO) if (L.cin == CI && L.cout == CO) return true;
    S2_SHAPES(CHK)
#undef CHK
    return false;
}
int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

bool s2_eligible(const cae_engine* e, const ConvLayer& L) { return e->use_s2 && L.sh_w >= 0 && s2_shape_ok(L); }

template <int CIN, int COUT, int KH, int KW>
void s2_fwd_launch(S2Fwd a, hipStream_t s) {
    // grid caps measured on MI355X at batch 64 (workgroups walk the remaining tiles): more workgroups
    // only add fp64-atomic traffic at the end of the kernel
    static const int capf = env_int("CAE_CAP_F", 1024), capf2 = env_int("CAE_CAP_F2", 512);   // env: tuning only
    // each thread covers 2x2 quads; lanes run along the row
    const int px = ((a.OW + 1) / 2 + 1) / 2, py = ((a.OH + 1) / 2 + 1) / 2;   // thread columns / rows per image
    if ((long long)a.B * px * py < 100000) {
        // small maps: 4x4 outputs per thread would leave most SIMDs without a wave; one quad per thread
        const int qx = (a.OW + 1) / 2, qy = (a.OH + 1) / 2;
        if constexpr (CIN * COUT * KH * KW > 80 && (256 / COUT) % 64 == 0) {
            // intermediate layers with many weights: output channels split over the waves (k_s2_fwd_cs)
            static const int cs_on = env_int("CAE_S2_CS", 1);   // env: A/B measurements only
            if (cs_on && (a.epi == S2_RAW_STATS || a.epi == S2_RAW)) {
                constexpr int PIX = 256 / COUT;
                if (qx > 32) {
                    a.tiles_x = (qx + 63) / 64;
                    a.tiles_y = (qy + PIX / 64 - 1) / (PIX / 64);
                    a.total_tiles = a.B * a.tiles_x * a.tiles_y;
                    hipLaunchKernelGGL((k_s2_fwd_cs<CIN, COUT, KH, KW, 64>), dim3(a.total_tiles < capf ? a.total_tiles : capf), dim3(256), 0, s, a);
                } else {
                    a.tiles_x = (qx + 31) / 32;
                    a.tiles_y = (qy + PIX / 32 - 1) / (PIX / 32);
                    a.total_tiles = a.B * a.tiles_x * a.tiles_y;
                    hipLaunchKernelGGL((k_s2_fwd_cs<CIN, COUT, KH, KW, 32>), dim3(a.total_tiles < capf ? a.total_tiles : capf), dim3(256), 0, s, a);
                }
                return;
            }
        }
        if (qx > 32) {
            a.tiles_x = (qx + 63) / 64;
            a.tiles_y = (qy + 3) / 4;
            a.total_tiles = a.B * a.tiles_x * a.tiles_y;
            hipLaunchKernelGGL((k_s2_fwd<CIN, COUT, KH, KW, 64>), dim3(a.total_tiles < capf ? a.total_tiles : capf), dim3(256), 0, s, a);
        } else {
            a.tiles_x = (qx + 31) / 32;
            a.tiles_y = (qy + 7) / 8;
            a.total_tiles = a.B * a.tiles_x * a.tiles_y;
            hipLaunchKernelGGL((k_s2_fwd<CIN, COUT, KH, KW, 32>), dim3(a.total_tiles < capf ? a.total_tiles : capf), dim3(256), 0, s, a);
        }
        return;
    }
    if (px > 32) {
        a.tiles_x = (px + 63) / 64;
        a.tiles_y = (py + 3) / 4;
        a.total_tiles = a.B * a.tiles_x * a.tiles_y;
        hipLaunchKernelGGL((k_s2_fwd2<CIN, COUT, KH, KW, 64>), dim3(a.total_tiles < capf2 ? a.total_tiles : capf2), dim3(256), 0, s, a);
    } else if (px > 16) {
        a.tiles_x = (px + 31) / 32;
        a.tiles_y = (py + 7) / 8;
        a.total_tiles = a.B * a.tiles_x * a.tiles_y;
        hipLaunchKernelGGL((k_s2_fwd2<CIN, COUT, KH, KW, 32>), dim3(a.total_tiles < capf2 ? a.total_tiles : capf2), dim3(256), 0, s, a);
    } else {
        a.tiles_x = (px + 15) / 16;
        a.tiles_y = (py + 15) / 16;
        a.total_tiles = a.B * a.tiles_x * a.tiles_y;
        hipLaunchKernelGGL((k_s2_fwd2<CIN, COUT, KH, KW, 16>), dim3(a.total_tiles < capf2 ? a.total_tiles : capf2), dim3(256), 0, s, a);
    }
}

void s2_fwd_dispatch(const ConvLayer& L, const S2Fwd& a, hipStream_t s) {
#define ONE(CI, CO, KH_, KW_) \
    if (L.cin == CI && L.cout == CO && L.kh == KH_ && L.kw == KW_) return s2_fwd_launch<CI, CO, KH_, KW_>(a, s);
#define PAIR(CI, CO) S2_KERNELS(ONE, CI, CO)
    S2_SHAPES(PAIR)
#undef PAIR
#undef ONE
}

template <int CIN, int COUT, int KH, int KW>
void s2_bwd_launch(S2Bwd a, hipStream_t s) {
    // 512 workgroups: each ends with Cin*Cout*kh*kw + 2*Cin fp64 atomics, and those dominate beyond that
    // (measured per step at batch 64: 1536 -> 286 us, 512 -> 274 us)
    static const int cap2 = env_int("CAE_CAP_B2", 512), caps = env_int("CAE_CAP_BS", 512);
    if constexpr (CIN * COUT * KH * KW <= 72) {
        // direct variant: one input pixel per thread, no LDS staging
        if (a.W > 32) {
            a.tiles_x = (a.W + 63) / 64;
            a.tiles_y = (a.H + 3) / 4;
            a.total_tiles = a.B * a.tiles_x * a.tiles_y;
            hipLaunchKernelGGL((k_s2_bwd2<CIN, COUT, KH, KW, 64>), dim3(a.total_tiles < cap2 ? a.total_tiles : cap2), dim3(256), 0, s, a);
        } else {
            a.tiles_x = (a.W + 31) / 32;
            a.tiles_y = (a.H + 7) / 8;
            a.total_tiles = a.B * a.tiles_x * a.tiles_y;
            hipLaunchKernelGGL((k_s2_bwd2<CIN, COUT, KH, KW, 32>), dim3(a.total_tiles < cap2 ? a.total_tiles : cap2), dim3(256), 0, s, a);
        }
    } else if constexpr (CIN == 8 && CIN * COUT * KH * KW / 4 <= 72) {
        // channels split over the 4 waves: 2 per thread, 64 pixels (32 x 2) per workgroup pass
        a.tiles_x = (a.W + 31) / 32;
        a.tiles_y = (a.H + 1) / 2;
        a.total_tiles = a.B * a.tiles_x * a.tiles_y;
        hipLaunchKernelGGL((k_s2_bwd_split<CIN, 2, COUT, KH, KW, 32>), dim3(a.total_tiles < caps ? a.total_tiles : caps), dim3(256), 0, s, a);
    } else {
        constexpr int CT = (CIN % 2 == 0 && CIN != 6) ? 2 : 3;   // input channels per thread
        constexpr int CG = CIN / CT;                              // ci-groups per workgroup
        constexpr int PIX = 256 / CG;                             // pixels per tile
        constexpr int TPX = 32, TPY = PIX / 32;
        a.tiles_x = (a.W + TPX - 1) / TPX;
        a.tiles_y = (a.H + TPY - 1) / TPY;
        a.total_tiles = a.B * a.tiles_x * a.tiles_y;
        const int grid = a.total_tiles < 1024 ? a.total_tiles : 1024;
        hipLaunchKernelGGL((k_s2_bwd<CIN, CT, COUT, KH, KW, TPX, TPY>), dim3(grid), dim3(256), 0, s, a);
    }
}

void s2_bwd_dispatch(const ConvLayer& L, const S2Bwd& a, hipStream_t s) {
#define ONE(CI, CO, KH_, KW_) \
    if (L.cin == CI && L.cout == CO && L.kh == KH_ && L.kw == KW_) return s2_bwd_launch<CI, CO, KH_, KW_>(a, s);
#define PAIR(CI, CO) S2_KERNELS(ONE, CI, CO)
    S2_SHAPES(PAIR)
#undef PAIR
#undef ONE
}

// ---- row-streaming backward of the thin middle layers (kernels_rows.h) ---------------------------------------------------
// false: the layer does not fit (shape, kernel size, map wider than a wave); the caller runs k_s2_bwd2 / k_s2_bwd_split
template <int CIN, int CT, int COUT, int HB, int D>
void rows_go(S2Rows a, int lw, hipStream_t s) {
    constexpr int NB = 4 / (CIN / CT);
    const int hmax = a.H > a.QH - 1 ? a.H : a.QH - 1;
    a.bands = (hmax + HB - 1) / HB;
    const int imgs = 64 / lw;
    a.groups = (a.B + imgs - 1) / imgs;
    const dim3 grid((unsigned)(a.groups * ((a.bands + NB - 1) / NB)));
    if (imgs == 1) hipLaunchKernelGGL((k_s2_bwd_rows<CIN, CT, COUT, 3, 3, HB, 1, D>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_s2_bwd_rows<CIN, CT, COUT, 3, 3, HB, 2, D>), grid, dim3(256), 0, s, a);
}

bool rows_bwd_ok(const cae_engine* e, const ConvLayer& L) {
    static const int enabled = env_int("CAE_ROWS", 1);            // env: A/B measurements only
    if (!enabled || !s2_eligible(e, L) || L.kh != 3 || L.kw != 3 || !L.has_bn) return false;
    const int qw = (L.wout + 1) / 2;
    const int lw = qw <= 32 ? 32 : 64;
    if (qw > 64 || L.win > lw - 1) return false;      // a lane per quad column; the last lane of an image owns no pixel
    return (L.cin == 4 && L.cout == 2) || (L.cin == 8 && L.cout == 4);
}

void rows_bwd_launch(const ConvLayer& L, S2Rows a, hipStream_t s) {
    static const int hb42 = env_int("CAE_ROWS_HB42", 2), hb84 = env_int("CAE_ROWS_HB84", 4);   // env: tuning only
    a.QH = (L.hout + 1) / 2;
    const int lw = (L.wout + 1) / 2 <= 32 ? 32 : 64;
    // short bands with every row's loads issued up front (a wave pays the memory latency once) against tall bands that load one
    // row ahead (less re-reading at the band edges, but a round trip per row: a wave is alone on its SIMD)
    if (L.cin == 4) {
        if (hb42 == 4) rows_go<4, 4, 2, 4, 1>(a, lw, s);
        else rows_go<4, 4, 2, 2, 3>(a, lw, s);
    } else {
        if (hb84 == 4) rows_go<8, 2, 4, 4, 1>(a, lw, s);
        else rows_go<8, 2, 4, 2, 3>(a, lw, s);
    }
}

template <int CIN, int COUT, int HB>
void rows_fwd_go(S2FwdRows a, int lw, hipStream_t s) {
    const int hmax = a.QH;
    a.bands = (hmax + HB - 1) / HB;
    const int imgs = 64 / lw;
    a.groups = (a.B + imgs - 1) / imgs;
    const dim3 grid((unsigned)(a.groups * ((a.bands + 3) / 4)));
    if (imgs == 1) hipLaunchKernelGGL((k_s2_fwd_rows<CIN, COUT, HB, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_s2_fwd_rows<CIN, COUT, HB, 2>), grid, dim3(256), 0, s, a);
}

void rows_fwd_launch(const ConvLayer& L, S2FwdRows a, hipStream_t s) {
    static const int hb42 = env_int("CAE_ROWS_FHB42", 2), hb84 = env_int("CAE_ROWS_FHB84", 1);   // env: tuning only
    a.QH = (L.hout + 1) / 2;
    const int lw = (L.wout + 1) / 2 <= 32 ? 32 : 64;
    if (L.cin == 4) {
        if (hb42 == 4) rows_fwd_go<4, 2, 4>(a, lw, s);
        else rows_fwd_go<4, 2, 2>(a, lw, s);
    } else {
        if (hb84 == 2) rows_fwd_go<8, 4, 2>(a, lw, s);
        else rows_fwd_go<8, 4, 1>(a, lw, s);
    }
}

// ---- last decoder layer of a training step as one launch (kernels_last.h): forward + sigmoid + MSE + backward ---------
bool last_fused_ok(const cae_engine* e, const ConvLayer& L) {
    static const int enabled = env_int("CAE_LAST_FUSED", 1);   // env: A/B measurements only
    return enabled && s2_eligible(e, L) && L.cin * L.cout * L.kh * L.kw <= 72 && L.sh_b >= 0;
}

template <int CIN, int COUT, int KH, int KW, int HB>
void last_fused_go(const S2Last& a, hipStream_t s) {
    const dim3 grid((a.total + 3) / 4);
    const bool vec4 = a.strips == 1 && (a.OW & 3) == 0, bn = a.bn_in.mode != BN_NONE;
    if (vec4 && bn) hipLaunchKernelGGL((k_s2_last_fused<CIN, COUT, KH, KW, HB, true, true>), grid, dim3(256), 0, s, a);
    else if (bn) hipLaunchKernelGGL((k_s2_last_fused<CIN, COUT, KH, KW, HB, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_s2_last_fused<CIN, COUT, KH, KW, HB, false, false>), grid, dim3(256), 0, s, a);
}

template <int CIN, int COUT, int KH, int KW>
void last_fused_launch(S2Last a, hipStream_t s) {
    if constexpr (CIN * COUT * KH * KW <= 72) {
        static const int hb_env = env_int("CAE_LAST_HB", 0);   // env: tuning only
        a.QH = (a.OH + 1) / 2;
        a.QW = (a.OW + 1) / 2;
        const int wmax = a.W > a.QW - 1 ? a.W : a.QW - 1, hmax = a.H > a.QH - 1 ? a.H : a.QH - 1;
        a.strips = (wmax + kLastStripPx - 1) / kLastStripPx;
        // a wave walks a band of HB quad rows (+1 recomputed): taller bands recompute less, shorter ones give more waves
        // (measured at the benchmark geometry: 2048 waves of 4+1 rows, two per SIMD, beat 1024 of 8+1: 20.4 against 21.8 us;
        // and at batch 128 / 512, where 8+1 rows used to be chosen: 244.2 against 246.8 and 601.6 against 607.6 us per step -
        // the 8-row variant's register arrays end up in scratch)
        int hb = hb_env ? hb_env : 4;
        if (hb != 8) hb = 4;
        a.bands = (hmax + hb - 1) / hb;
        a.total = a.B * a.strips * a.bands;
        if (hb == 8) last_fused_go<CIN, COUT, KH, KW, 8>(a, s);
        else last_fused_go<CIN, COUT, KH, KW, 4>(a, s);
    }
}

void last_fused_dispatch(const ConvLayer& L, const S2Last& a, hipStream_t s) {
#define ONE(CI, CO, KH_, KW_) \
    if (L.cin == CI && L.cout == CO && L.kh == KH_ && L.kw == KW_) return last_fused_launch<CI, CO, KH_, KW_>(a, s);
#define PAIR(CI, CO) S2_KERNELS(ONE, CI, CO)
    S2_SHAPES(PAIR)
#undef PAIR
#undef ONE
}

template <class K>
void head_lds_attr(K kernel, size_t bytes) {
    static size_t granted = 64 * 1024;
    if (bytes > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        granted = bytes;
    }
}

// ---- LDS-staged implicit-GEMM forward of a channel-rich stride-2 ConvTranspose2d (kernels_ctlds.h) ---------------------
// false: the layer does not fit this kernel (odd channel counts, kernels other than 3/4 taps, an image + weight slice
// larger than LDS); the caller then runs the gather kernel k_ig_fwd_s2.
template <int KH, int KW>
void ct_fwd_go(const CtFwd& c, dim3 grid, int threads, size_t lds, hipStream_t s) {
    head_lds_attr(k_ct_fwd_lds<KH, KW>, lds);
    hipLaunchKernelGGL((k_ct_fwd_lds<KH, KW>), grid, dim3(threads), lds, s, c);
}

bool ct_fwd_launch(cae_engine* e, const StepArgs& a, const ConvLayer& L, int layer, const float* in, const BnDesc& bn_in,
                   float* out, double* stats) {
    static const int enabled = env_int("CAE_CTLDS", -1);       // env: A/B measurements only - layer mask (-1: by batch size, below)
    static const int mf_target = env_int("CAE_CT_MF", 24);     // env: tuning only - MFMAs per wave before K is split further
    // Against the gather kernel it replaces (k_ig_fwd_s2) the LDS-staged one has a third of the instructions and wins from
    // batch 128 up (239.9 / 342.1 / 590.8 against 243.6 / 349.5 / 598.3 us per step at 128 / 256 / 512); at the benchmark's 64
    // both take 8-11 us per launch, all of it latency, and the step is 0.8 us shorter with the gather kernels (167.7 against
    // 168.5, four alternating runs) - but their gradients at that size sit 6.8e-4 from the oracle's where the LDS-staged
    // forward's sit within the full-size test's 2e-4 (test_full_size_gpu.py): parity first, the LDS-staged kernels run.
    const int mask = e->gather_fwd ? 0 : (enabled >= 0 ? enabled : 0x7fffffff);
    if (!(layer < 31 && ((mask >> layer) & 1)) || L.cin % 4 || L.kh < 3 || L.kw < 3) return false;
    CtFwd c;
    memset(&c, 0, sizeof c);
    c.B = a.batch; c.Cin = L.cin; c.H = L.hin; c.W = L.win; c.Cout = L.cout; c.OH = L.hout; c.OW = L.wout;
    c.QH = (L.hout + 1) / 2; c.QW = (L.wout + 1) / 2;
    c.PW = c.QW + 1;
    c.plane = ((c.QH + 1) * c.PW) | 1;
    c.tiles = (c.QH * c.QW + 15) / 16;
    const int taps = ((L.kh + 1) / 2 + L.kh / 2) * ((L.kw + 1) / 2 + L.kw / 2);   // sum of n_p over the four parities = kh * kw
    const int mf = L.cin * taps / 4;
    int ks = 1;
    while (ks < 8 && mf / ks > mf_target && L.cin % (ks * 2 * 4) == 0) ks *= 2;
    int rt = 8 / ks;
    if (rt > c.tiles) rt = c.tiles;
    if (rt > 4) rt = 4;
    c.ks = ks; c.rt = rt;
    c.tg = (c.tiles + rt - 1) / rt;
    const int waves = rt * ks;
    const size_t lds = ct_fwd_lds_bytes(L.cin, c.plane, L.kh, L.kw, waves, ks);
    if (lds > 150 * 1024 || c.plane >= kDivSmallMaxD || (long long)L.cin * c.plane >= kDivSmallMaxN) return false;
    c.in = in; c.bn_in = bn_in; c.w = e->params + L.w_off; c.bias = e->params + L.b_off; c.out = out; c.stats = stats;
    {
        static const int dbg = env_int("CAE_HEAD_DBG", 0), dbg_layer = env_int("CAE_DBG_LAYER", 2);   // tools/ct_phases.py
        c.dbg = dbg == 3 && layer == dbg_layer && a.train ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
    }
    dim3 grid((unsigned)(a.batch * c.tg), (unsigned)((L.cout + 15) / 16));
    ProfScope _p(e, a.train ? "ct_convt_fwd" : "ct_convt_eval", layer, f4((double)a.batch * (L.in_elems() + L.out_elems())));
    if (L.kh == 3 && L.kw == 3) ct_fwd_go<3, 3>(c, grid, 64 * waves, lds, e->stream);
    else if (L.kh == 4 && L.kw == 4) ct_fwd_go<4, 4>(c, grid, 64 * waves, lds, e->stream);
    else if (L.kh == 3 && L.kw == 4) ct_fwd_go<3, 4>(c, grid, 64 * waves, lds, e->stream);
    else ct_fwd_go<4, 3>(c, grid, 64 * waves, lds, e->stream);
    return true;
}

// ---- fused head / tail (kernels_head.h) ----------------------------------------------------------

// Fills the descriptor shared by k_head_fwd and k_tail_bwd and lays out their LDS.  Returns false when the model or
// the batch does not fit (the caller then runs the per-layer launches).
bool head_plan(const cae_engine* e, const StepArgs& a, HeadArgs& h, size_t& lds_bytes) {
    static const int enabled = env_int("CAE_HEAD", 1);   // env: A/B measurements only
    if (!enabled || !e->use_s2 || a.syncing() || e->variational || (int)e->enc.size() > kHeadMaxEnc) return false;
    memset(&h, 0, sizeof h);
    h.B = a.batch;
    h.n_enc = (int)e->enc.size();
    h.train = a.train ? 1 : 0;
    h.momentum = kBnMomentum;
    h.eps = kBnEps;
    h.st = e->state();
    // (1 since the end of round 2: 168.2 against 169.7 us per step with 4 - more workgroups recompute the encoder, each holds
    // a quarter of the last Linear layer's weights and finishes its strip sooner; 8: 186.9)
    static const int tpw = env_int("CAE_HEAD_TPW", 1);   // env: tuning only - Linear-3 column tiles per workgroup
    h.tiles_per_wg = tpw < 1 ? 1 : tpw;
    double* acc = e->gradacc();
    int64_t top = 0;
    auto take = [&](int64_t floats) {
        top = align_up(top, 4);
        const int64_t o = top;
        top += floats;
        return (int)o;
    };
    h.o_perm = take(a.batch);
    int maxc = 1;
    for (int l = 0; l < h.n_enc; l++) {
        const ConvLayer& L = e->enc[l];
        if (L.cout > kHeadMaxC) return false;
        HeadConv& c = h.enc[l];
        c.cin = L.cin; c.hin = L.hin; c.win = L.win; c.cout = L.cout; c.hout = L.hout; c.wout = L.wout;
        c.kh = L.kh; c.kw = L.kw; c.s = L.stride;
        c.w = e->params + L.w_off; c.bias = e->params + L.b_off;
        c.gamma = e->params + L.gamma_off; c.beta = e->params + L.beta_off;
        c.rmean = e->bufs + L.rm_off; c.rvar = e->bufs + L.rv_off; c.saved = e->bn_saved(L.bn_index);
        c.y = e->fptr(L.act_off);
        {
            const double count = (double)a.batch * L.hout * L.wout;
            c.inv_count = 1.0 / count;
            c.unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
        }
        {   // two burst segments: [conv weight .. BatchNorm bias] of the parameter arena, [running mean .. var] of the buffers
            const int64_t pn = L.beta_off + L.cout - L.w_off, bnn = L.rv_off + L.cout - L.rm_off;
            if (pn > kHeadThreads || bnn > kHeadThreads || pn <= 0 || bnn <= 0 || h.n_seg + 2 > kHeadMaxSeg) return false;
            c.o_w = take(pn);
            c.o_b = c.o_w + (int)(L.b_off - L.w_off);
            c.o_gamma = c.o_w + (int)(L.gamma_off - L.w_off);
            c.o_beta = c.o_w + (int)(L.beta_off - L.w_off);
            c.o_rm = take(bnn);
            c.o_rv = c.o_rm + (int)(L.rv_off - L.rm_off);
            h.seg[h.n_seg++] = HeadSeg{e->params + L.w_off, (int)pn, c.o_w, 0};
            h.seg[h.n_seg++] = HeadSeg{e->bufs + L.rm_off, (int)bnn, c.o_rm, 0};
        }
        c.o_c = take(4 * (int64_t)L.cout);
        if (l + 1 == h.n_enc) c.o_y = take((int64_t)a.batch * L.out_elems());   // inner maps live in the union region
        if (L.cout > maxc) maxc = L.cout;
    }
    int maxd = 0;
    for (int i = 0; i < 4; i++) {
        const FcLayer& F = e->fc[i];
        HeadFc& f = h.fc[i];
        f.nin = F.nin; f.nout = F.nout; f.relu = F.relu ? 1 : 0;
        f.w = e->params + F.w_off; f.bias = e->params + F.b_off;
        f.act = e->fptr(F.act_off); f.grad = e->fptr(F.grad_off);
        f.w_acc = acc + F.w_off; f.b_acc = acc + F.b_off;
        if (i < 3 && F.nout > maxd) maxd = F.nout;
        if (i < 3 && F.nin > maxd) maxd = F.nin;
    }
    h.ld_h = (maxd + 31) / 32 * 32 + 2;   // row stride = 2 mod 32 banks: the 16 rows x 2 k of an MFMA operand read do not collide
    h.o_red = take(2 * 2 * (int64_t)maxc * kHeadWaves);   // doubles
    h.o_h[0] = take(16 * (int64_t)h.ld_h);
    h.o_h[1] = take(16 * (int64_t)h.ld_h + 32);   // + guard: an 8-deep k-batch may read up to 30 floats past row 15 (against zero B operands)
    h.o_part = take(kHeadWaves * 256);
    {   // union region: the gathered input and the encoder's inner maps while the encoder runs, the Linear weights after
        // (row stride = 4 mod 32 floats: 16-byte aligned rows, two-way bank conflicts at worst on the operand reads)
        int64_t wf = 0;
        int start4 = 0;
        for (int i = 0; i < 4; i++) {
            const FcLayer& F = e->fc[i];
            if (F.nin % 4) return false;   // whole k-steps and 16-byte rows
            const int rows = i == 3 ? 16 * h.tiles_per_wg : (F.nout + 15) / 16 * 16;
            HeadW& w = h.wmat[i];
            w.src = e->params + F.w_off;
            w.n4row = F.nin / 4;
            w.ldw = (F.nin + 31) / 32 * 32 + 4;
            w.start4 = start4;
            w.strip_floats = i == 3 ? 16 * h.tiles_per_wg * F.nin : 0;
            w.lds_off = (int)wf;   // relative, rebased below
            start4 += (i == 3 ? std::min(rows, F.nout) : F.nout) * w.n4row;
            wf += (int64_t)rows * w.ldw;
        }
        wf += 64;   // k-batches read past the last row
        h.w_total4 = start4;
        if (h.w_total4 > kHeadW4 * kHeadThreads) return false;
        for (int j = 0; j < kHeadW4; j++) {
            const int lo = j * kHeadThreads, hi = std::min((j + 1) * kHeadThreads, h.w_total4) - 1;   // float4s of piece j
            h.piece_m[j] = -1;
            for (int m = 0; m < 4 && hi >= lo; m++) {
                const int mend = m < 3 ? h.wmat[m + 1].start4 : h.w_total4;
                if (lo >= h.wmat[m].start4 && hi < mend) h.piece_m[j] = m;
            }
        }
        int64_t ef = (int64_t)a.batch * e->enc[0].in_elems();
        for (int l = 0; l + 1 < h.n_enc; l++) ef += align_up((int64_t)a.batch * e->enc[l].out_elems(), 4);
        const int base = take(std::max(wf, ef));
        h.o_x = base;
        int64_t o = base + (int64_t)a.batch * e->enc[0].in_elems();
        for (int l = 0; l + 1 < h.n_enc; l++) {
            o = align_up(o, 4);
            h.enc[l].o_y = (int)o;
            o += (int64_t)a.batch * e->enc[l].out_elems();
        }
        for (int i = 0; i < 4; i++) h.wmat[i].lds_off += base;
    }
    for (int i = 0; i < 4; i++) {
        const int cnt = i == 3 ? 16 * h.tiles_per_wg : e->fc[i].nout;
        if (cnt > kHeadThreads || h.n_seg + 1 > kHeadMaxSeg) return false;
        h.o_bias[i] = take(cnt);
        // Linear 3: a strip of 16 * tiles_per_wg biases per workgroup column; the last strip may read past the vector
        // (clamped reads inside the parameter arena, masked by the epilogue's n < N)
        h.seg[h.n_seg++] = HeadSeg{e->params + e->fc[i].b_off, i == 3 ? std::min(cnt, e->fc[3].nout) : cnt, h.o_bias[i], i == 3 ? 1 : 0};
    }
    for (int i = 0; i < 4; i++) {
        if (e->fc[i].nin % 4) return false;   // stage_prefetch walks whole k-steps
        const int tiles = i == 3 ? h.tiles_per_wg : (e->fc[i].nout + 15) / 16;
        h.fc_split[i] = stage_split(tiles, e->fc[i].nin);
    }
    lds_bytes = (size_t)align_up(top, 4) * sizeof(float);
    return lds_bytes <= 152 * 1024;
}


// k_tail_bwd (kernels_head.h): Linear 2..0 backward in one launch.  False: run the per-layer pair launches.
bool tail_plan(const cae_engine* e, const StepArgs& a, TailArgs& t, size_t& lds_bytes) {
    static const int enabled = env_int("CAE_TAIL", 1);   // env: A/B measurements only
    if (!enabled || !e->use_s2 || a.syncing() || e->variational) return false;
    memset(&t, 0, sizeof t);
    const ConvLayer& P = e->enc.back();
    double* acc = e->gradacc();
    t.B = a.batch;
    for (int i = 0; i < 3; i++) {
        const FcLayer& F = e->fc[i];
        if (F.nin % 4 || F.nout % 4) return false;          // 16-byte rows, whole k-steps
        if (16 * F.nout / 4 > 2 * kHeadThreads) return false;   // a 16-row panel in two loads per thread
        HeadFc& f = t.fc[i];
        f.nin = F.nin; f.nout = F.nout; f.relu = F.relu ? 1 : 0;
        f.w = e->params + F.w_off; f.bias = e->params + F.b_off;
        f.act = e->fptr(F.act_off); f.grad = e->fptr(F.grad_off);
        f.w_acc = acc + F.w_off; f.b_acc = acc + F.b_off;
        t.w4[i] = F.nin * F.nout / 4;
        if (t.w4[i] > 2 * kHeadThreads) return false;
        const int r16 = (F.nin + 15) / 16 * 16;
        t.ldw[i] = r16 % 32 == 0 ? r16 + 16 : r16;   // = 16 mod 32: the four k rows of an operand read land on distinct banks
    }
    if (P.cout > kHeadMaxC || e->fc[0].nin != P.cout * P.hout * P.wout) return false;
    t.y_last = e->fptr(P.act_off);
    t.g_last = e->fptr(P.grad_off);
    t.gamma = e->params + P.gamma_off;
    t.beta = e->params + P.beta_off;
    t.saved = e->bn_saved(P.bn_index);
    t.stats = e->bn_stats(P.bn_index);
    t.C = P.cout;
    t.hw = P.hout * P.wout;
    auto ld_of = [](int n) { return (n + 31) / 32 * 32 + 4; };
    t.ld2 = ld_of(e->fc[2].nout);
    t.ld1 = ld_of(e->fc[1].nout);
    t.ld0 = ld_of(e->fc[0].nout);
    int64_t top = 0;
    auto take = [&](int64_t floats) {
        top = align_up(top, 4);
        const int64_t o = top;
        top += floats;
        return (int)o;
    };
    t.o_c = take(4 * (int64_t)P.cout);
    t.o_red = take(2 * 2 * (int64_t)kHeadWaves);
    t.o_g2 = take(16 * (int64_t)t.ld2 + 32);   // + guard: a k-batch may read a few floats past row 15 (against zero B operands)
    t.o_g1 = take(16 * (int64_t)t.ld1 + 32);
    t.o_g0 = take(16 * (int64_t)t.ld0 + 32);
    // y and gx sit inside the cleared region too: the weight-gradient stage reads all 16 panel rows without predicates, and
    // rows past the batch must be finite (they meet zero gradient rows; LDS garbage could be NaN)
    t.o_y = take(16 * (int64_t)e->fc[0].nin + 32);
    t.o_gx = take(16 * (int64_t)e->fc[0].nin + 32);
    top = align_up(top, 4);
    t.zero4 = (int)((top - t.o_g2) / 4);
    int64_t wmax = 0;
    for (int i = 0; i < 3; i++) wmax = std::max<int64_t>(wmax, (int64_t)e->fc[i].nout * t.ldw[i]);
    t.o_w = take(wmax + 64);
    t.o_part = take(kHeadWaves * 256);
    lds_bytes = (size_t)align_up(top, 4) * sizeof(float);
    if (lds_bytes > 152 * 1024) return false;
    // chain (16 rows): g1 (N = fc2.nin, K = fc2.nout), g0 (N = fc1.nin, K = fc1.nout), gx (N = fc0.nin, K = fc0.nout)
    for (int i = 0; i < 3; i++) t.sp_d[i] = stage_split((e->fc[2 - i].nin + 15) / 16, e->fc[2 - i].nout);
    // weight-gradient shares: M = nout, N = nin + 1, K = 16 rows
    for (int i = 0; i < 3; i++) t.sp_w[i] = stage_split(((e->fc[2 - i].nout + 15) / 16) * ((e->fc[2 - i].nin + 16) / 16), 16);
    return true;
}

// ---- data-parallel gradient exchange (cae_dp_train_step) -------------------------------------------
// Two buckets in the order backward completes them: [bucket_split, n_param) = Linear 3 and the decoder convolutions, ready
// as soon as Linear 3's backward has run, narrowed to fp32 and all-reduced on the second stream while the main stream
// still runs Linear 2..0 and the encoder backward; then [0, bucket_split) on the main stream once the first has finished
// (one communicator, one collective at a time, ordered on the device by the join event).
// Under SyncBN every collective (tables and buckets) stays on the main stream: the tables are on the critical path anyway.
StepTail narrow_tail(cae_engine* e, bool with_step_tail, int batch_inc) {
    StepTail t = step_tail_of(e, batch_inc, 1);
    if (!with_step_tail) {
        t.zero_extra = nullptr;
        t.zero_extra_n = 0;
        t.st = nullptr;
    }
    return t;
}

int dp_allreduce_grads(cae_engine* e, int64_t lo, int64_t hi, hipStream_t on) {
    if (hi <= lo) return CAE_OK;
    NCCL_TRY(rccl().AllReduce(e->grads + lo, e->grads + lo, (size_t)(hi - lo), RcclApi::kFloat32, RcclApi::kSum, e->dp_comm, on));
    return CAE_OK;
}

// first bucket on the second stream (cae_dp_set_overlap; the env variable, read once, overrides it for measurements)
bool dp_overlap(const cae_engine* e, const StepArgs& a) {
    static const int forced = env_int("CAE_DP_OVERLAP", -1);
    return (forced >= 0 ? forced != 0 : e->dp_overlap) && !a.dp_sync;
}

// Without the overlap (and without SyncBN) there is nothing to gain from two buckets: ONE narrowing launch and ONE all-reduce
// of the whole gradient arena after backward - one collective latency per step instead of two.
bool dp_single_collective(const cae_engine* e, const StepArgs& a) { return !dp_overlap(e, a) && !a.dp_sync; }

int dp_first_bucket(cae_engine* e, const StepArgs& a) {
    if (!a.dp || dp_single_collective(e, a)) return CAE_OK;
    const int64_t lo = e->bucket_split, hi = e->n_param;
    const bool overlap = dp_overlap(e, a);
    hipStream_t on = overlap ? e->comm_stream : e->stream;
    if (overlap) {
        HIP_TRY(hipEventRecord(e->ev_fork, e->stream));
        HIP_TRY(hipStreamWaitEvent(e->comm_stream, e->ev_fork, 0));
    }
    {
        ProfScope _p(e, "dp_narrow_bucket0", 0, 12.0 * (hi - lo), on);
        hipLaunchKernelGGL(k_narrow_range, dim3(grid1(hi - lo)), dim3(256), 0, on, (long long)lo, (long long)hi, e->grads,
                           e->shard_segs(), narrow_tail(e, false, 0));
    }
    if (int rc = dp_allreduce_grads(e, lo, hi, on)) return rc;
    if (overlap) HIP_TRY(hipEventRecord(e->ev_join, e->comm_stream));
    return CAE_OK;
}

// after the last backward kernel: second bucket, join, Adam from the reduced fp32 gradients
int dp_finish_step(cae_engine* e, const StepArgs& a) {
    hipStream_t s = e->stream;
    const int64_t lo = 0, hi = dp_single_collective(e, a) ? e->n_param : e->bucket_split;
    {
        ProfScope _p(e, "dp_narrow_bucket1", 0, 12.0 * (hi - lo));
        hipLaunchKernelGGL(k_narrow_range, dim3(grid1(hi - lo > 0 ? hi - lo : 1)), dim3(256), 0, s, (long long)lo, (long long)hi,
                           e->grads, e->shard_segs(), narrow_tail(e, true, a.inc()));
    }
    // the first bucket's all-reduce has to be over before the second is enqueued: one communicator runs one collective at a
    // time, and the join orders the two on the device (one fork + one join per step; the second bucket is last on the
    // critical path either way, so it runs on the main stream)
    if (dp_overlap(e, a)) HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
    if (int rc = dp_allreduce_grads(e, lo, hi, s)) return rc;
    StepTail none;
    memset(&none, 0, sizeof none);
    ProfScope _p(e, "adam", 0, 28.0 * e->n_param);
    hipLaunchKernelGGL(k_adam, dim3(grid1(e->n_param)), dim3(256), 0, s, (long long)e->n_param, e->params,
                       (const float*)e->grads, e->m, e->v, e->hp, (const StepState*)e->state(), e->shard_segs(), none, 0,
                       std::log(e->hp.beta1), std::log(e->hp.beta2), AdamConv0{});
    return CAE_OK;
}

int launch_forward(cae_engine* e, const StepArgs& a) {
    hipStream_t s = e->stream;
    const int B = a.batch;
    const StepState* st = e->state();
    const int act_mode = a.train ? BN_BATCH : BN_RUNNING;

    HeadArgs head;
    size_t head_lds = 0;
    const bool fused_head = a.part == 0 && head_plan(e, a, head, head_lds);
    if (fused_head) {
        head.x = a.x_direct ? a.x_direct : e->ds_x[a.which];
        head.perm = a.x_direct ? nullptr : a.perm;
        head.use_cursor = a.x_direct ? 0 : 1;
        head.bump_adam = a.train ? 1 : 0;
        static const int dbg = env_int("CAE_HEAD_DBG", 0);   // tools/head_phases.py: stamps land in fc[3]'s gradient buffer
        head.dbg = dbg && !a.train ? reinterpret_cast<long long*>(e->fptr(e->fc[3].grad_off)) : nullptr;
        const int T = (e->fc[3].nout + 15) / 16;
        double bytes = 0;
        for (auto& L : e->enc) bytes += f4((double)B * (L.in_elems() + L.out_elems()));
        for (int i = 0; i < 4; i++) bytes += f4((double)B * (e->fc[i].nin + e->fc[i].nout) + (double)e->fc[i].nin * e->fc[i].nout);
        head_lds_attr(k_head_fwd, head_lds);
        e->x_published = false;
        if (a.train) {
            // EVERY training forward clears the first encoder layer's BatchNorm table before anything adds to it: a fused
            // optimiser launch (AdamConv0) reads that table and therefore leaves it dirty, and which kind of step ran last
            // is not something a captured graph can know (the per-layer path below does the same with a fill launch)
            head.clear0 = e->bn_stats(e->enc[0].bn_index);
            head.clear0_n = kStatShards * e->enc[0].cout * 4;
            if (a.adam_follows) {
                head.xbatch = e->fptr(e->off_xbatch);
                e->x_published = true;
            }
        }
        ProfScope _p(e, a.train ? "head_fwd" : "head_eval", 0, bytes);
        hipLaunchKernelGGL(k_head_fwd, dim3((B + 15) / 16, (T + head.tiles_per_wg - 1) / head.tiles_per_wg), dim3(kHeadThreads),
                           head_lds, s, head);
    }
    if (!fused_head && a.train && a.part == 0)   // (see the fused launch above: every training forward clears this table first)
        HIP_TRY(hipMemsetAsync(e->bn_stats(e->enc[0].bn_index), 0, (size_t)kStatShards * e->enc[0].cout * 4 * sizeof(double), s));
    // ---- encoder convs (encoder.py:40-46)
    for (size_t l = 0; !fused_head && a.part != 2 && l < e->enc.size(); l++) {
        const ConvLayer& L = e->enc[l];
        ConvGeom g{B, L.cout, L.hout, L.wout, L.cin, L.hin, L.win, L.kh, L.kw, L.stride};
        Src big;
        BnDesc bnb = bn_none();
        if (l == 0) {
            big = src_plain(a.x_direct ? a.x_direct : e->ds_x[a.which], L.cin, L.hin, L.win);
            big.perm = a.x_direct ? nullptr : a.perm;
            big.use_cursor = a.x_direct ? 0 : 1;
            big.bump_adam = a.train ? 1 : 0;
        } else {
            const ConvLayer& P = e->enc[l - 1];
            big = src_plain(e->fptr(P.act_off), L.cin, L.hin, L.win);
            bnb = bn_of(e, P, act_mode, (double)a.bn_batch * P.hout * P.wout, 1);
        }
        Epi ep = epi_plain(e->fptr(L.act_off));
        if (a.train) {
            ep.kind = EPI_STATS;
            ep.stats = e->bn_stats(L.bn_index);
            ep.stats_C = L.cout;
        }
        dim3 grid(grid1((int64_t)B * L.hout * L.wout), L.cout);
        ProfScope _p(e, a.train ? "enc_conv_fwd" : "enc_conv_eval", (int)l, f4((double)B * (L.in_elems() + L.out_elems())));
        hipLaunchKernelGGL(k_down, grid, dim3(256), lds_bytes(L.cin, L.cout), s, g, big, bnb, e->params + L.w_off,
                           e->params + L.b_off, ep, bn_none(), st);
        if (a.train)
            if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
    }
    // ---- encoder_lin / decoder_lin (encoder.py:54-58, decoder.py:31-35)
    if (!fused_head) {
        const ConvLayer& P = e->enc.back();
        const int hw = P.hout * P.wout;
        BnDesc bni = bn_of(e, P, act_mode, (double)a.bn_batch * hw, 1);
        const float* in = a.part == 2 ? a.z_in : e->fptr(P.act_off);
        const int fc_lo = a.part == 2 ? 2 : 0, fc_hi = a.part == 1 ? 2 : 4;
        for (int i = fc_lo; i < fc_hi; i++) {
            const FcLayer& F = e->fc[i];
            ProfScope _p(e, e->use_s2 ? "linear_fwd_mfma" : "linear_fwd", i, f4((double)B * (F.nin + F.nout) + (double)F.nin * F.nout));
            if (e->use_s2) {
                GemmArgs ga;
                memset(&ga, 0, sizeof ga);
                ga.M = B; ga.N = F.nout; ga.K = F.nin;
                ga.A = in; ga.sa_m = F.nin; ga.sa_k = 1;
                ga.B = e->params + F.w_off; ga.sb_k = 1; ga.sb_n = F.nin;   // B[k][n] = W[n][k]
                ga.C = e->fptr(F.act_off); ga.sc_m = F.nout; ga.sc_n = 1;
                ga.epi = GE_STORE;
                ga.bias = e->params + F.b_off;
                ga.relu = F.relu ? 1 : 0;
                ga.bn_a = i == 0 ? bni : bn_none();
                ga.hw_a = hw;
                ga.bn_c = bn_none();
                const int tiles = ((B + 15) / 16) * ((F.nout + 15) / 16);
                hipLaunchKernelGGL(k_gemm16, dim3(tiles), dim3(256), gemm_lds(i == 0 ? P.cout : 0), s, ga);
            } else {
                hipLaunchKernelGGL(k_lin_fwd, dim3(grid1((int64_t)B * F.nout)), dim3(256), lds_bytes(i == 0 ? P.cout : 0, 0),
                                   s, B, F.nin, F.nout, in, i == 0 ? bni : bn_none(), hw, e->params + F.w_off,
                                   e->params + F.b_off, F.relu ? 1 : 0, e->fptr(F.act_off));
            }
            in = e->fptr(F.act_off);
            if (i == 1 && e->variational) {   // heads -> z (trunk_api.h)
                if (!e->hooks.reparam) return fail(CAE_ERR_STATE, "trunk engine without a reparameterisation hook");
                e->hooks.reparam(e->hooks.user, s, in, B, e->latent, a.train ? 1 : 0, e->fptr(e->off_vz));
                in = e->fptr(e->off_vz);
            }
        }
        if (a.part == 1) {   // the latent vector leaves the engine: (batch, latent) fp32, contiguous like the Linear's output
            HIP_TRY(hipMemcpyAsync(a.z_out, e->fptr(e->fc[1].act_off), sizeof(float) * (size_t)B * e->fc[1].nout, hipMemcpyDeviceToDevice, s));
            return CAE_OK;
        }
    }
    // ---- decoder conv-transposes (decoder.py:40-48) + sigmoid (:77) + MSELoss (conv_ae_model.py:303)
    for (size_t l = 0; l < e->dec.size(); l++) {
        const ConvLayer& L = e->dec[l];
        const bool last = l + 1 == e->dec.size();
        ConvGeom g{B, L.cin, L.hin, L.win, L.cout, L.hout, L.wout, L.kh, L.kw, L.stride};
        Src small;
        BnDesc bns = bn_none();
        if (l == 0) {
            small = src_plain(e->fptr(e->fc[3].act_off), L.cin, L.hin, L.win);
        } else {
            const ConvLayer& P = e->dec[l - 1];
            small = src_plain(e->fptr(P.act_off), L.cin, L.hin, L.win);
            bns = bn_of(e, P, act_mode, (double)a.bn_batch * P.hout * P.wout, 1);
        }
        Epi ep;
        if (!last) {
            ep = epi_plain(e->fptr(L.act_off));
            if (a.train) {
                ep.kind = EPI_STATS;
                ep.stats = e->bn_stats(L.bn_index);
                ep.stats_C = L.cout;
            }
        } else {
            memset(&ep, 0, sizeof ep);
            // mean over the GLOBAL batch: each rank contributes sum(local terms) / global count, and the SUM all-reduce of the
            // gradients then yields the global-mean gradient (global_batch == batch on a single device)
            ep.inv_count = (float)(1.0 / ((double)a.global_batch * L.cout * L.hout * L.wout));
            ep.losses = e->losses();
            ep.perm = a.perm;
            ep.use_cursor = a.use_cursor ? 1 : 0;
            if (a.train) {
                ep.kind = EPI_SIGMSE;
                ep.out = e->fptr(e->off_glast);
                ep.target = e->ds_t[a.which];
                ep.bias_acc = e->gradacc() + L.b_off;
            } else {
                ep.kind = EPI_SIGOUT;
                ep.yhat = a.yhat;
                ep.target = a.want_loss ? e->ds_t[a.which] : nullptr;
            }
        }
        if (last && a.external_loss) {   // raw output for a loss computed outside the trunk: no sigmoid, no statistics
            ep = epi_plain(e->fptr(e->off_zlast));
        }
        if (last && a.train && !a.external_loss && last_fused_ok(e, L)) continue;   // forward, loss and backward of this layer: one launch, in launch_backward
        if (!last && l > 0 && e->dec[l - 1].has_bn && rows_bwd_ok(e, L)) {
            static const int fwd_rows = env_int("CAE_ROWS_FWD", 1);   // env: A/B measurements only
            if (fwd_rows) {
                S2FwdRows f;
                memset(&f, 0, sizeof f);
                f.B = B; f.H = L.hin; f.W = L.win; f.OH = L.hout; f.OW = L.wout;
                f.in = small.p; f.bn_in = bns;
                f.w = e->params + L.w_off; f.bias = e->params + L.b_off;
                f.out = ep.out;
                f.stats = a.train ? ep.stats : nullptr;
                ProfScope _p(e, a.train ? "s2_convt_fwd" : "s2_convt_eval", (int)l, f4((double)B * (L.in_elems() + L.out_elems())));
                rows_fwd_launch(L, f, s);
                if (a.train)
                    if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
                continue;
            }
        }
        if (s2_eligible(e, L)) {
            S2Fwd f;
            memset(&f, 0, sizeof f);
            f.B = B; f.H = L.hin; f.W = L.win; f.OH = L.hout; f.OW = L.wout;
            f.in = small.p;
            f.w = e->params + L.w_off;
            f.bias = e->params + L.b_off;
            f.bn_in = bns;
            f.st = st;
            if (!last) {
                f.out = ep.out;
                f.stats = ep.stats;
                f.epi = a.train ? S2_RAW_STATS : S2_RAW;
            } else if (a.external_loss) {
                f.out = ep.out;
                f.epi = S2_RAW;
            } else {
                f.out = a.train ? ep.out : ep.yhat;
                f.target = ep.target;
                f.perm = ep.perm;
                f.use_cursor = ep.use_cursor;
                f.losses = ep.losses;
                f.inv_count = ep.inv_count;
                f.bias_acc = e->sgacc() + L.sh_b;
                f.bias_stride = e->segs.n;
                f.epi = a.train ? S2_SIGMSE : S2_SIGOUT;
            }
            ProfScope _p(e, last ? (a.train ? "s2_convt_last_fwd_loss" : "s2_convt_last_eval") : (a.train ? "s2_convt_fwd" : "s2_convt_eval"), (int)l,
                         f4((double)B * (L.in_elems() + L.out_elems() * (last && (a.train || a.want_loss) ? 2.0 : 1.0))));
            s2_fwd_dispatch(L, f, s);
            if (a.train && !last)
                if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
            continue;
        }
        if (e->use_s2 && !last && L.stride == 2 && L.kh <= 4 && L.kw <= 4) {
            if (ct_fwd_launch(e, a, L, (int)l, small.p, bns, ep.out, a.train ? ep.stats : nullptr)) {
                if (a.train)
                    if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
                continue;
            }
            IgFwd f;
            memset(&f, 0, sizeof f);
            f.B = B; f.Cin = L.cin; f.H = L.hin; f.W = L.win; f.Cout = L.cout; f.OH = L.hout; f.OW = L.wout;
            f.KH = L.kh; f.KW = L.kw; f.QH = (L.hout + 1) / 2; f.QW = (L.wout + 1) / 2;
            f.in = small.p; f.bn_in = bns; f.w = e->params + L.w_off; f.bias = e->params + L.b_off;
            f.out = ep.out; f.stats = a.train ? ep.stats : nullptr;
            {
                static const int dbg = env_int("CAE_HEAD_DBG", 0);   // tools/head_phases.py ig: stamps of the LAST such layer
                f.dbg = dbg == 1 && a.train ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
            }
            const int mtiles = (B * f.QH * f.QW + 15) / 16;
            f.ksplit = L.cin >= 48 ? 4 : (L.cin >= 24 ? 2 : 1);
            // at most ~1024 workgroups over the 4 parities: every workgroup ends with up to 32 fp64 atomics
            {
                const int waves_m = 4 / f.ksplit;
                static const int target = env_int("CAE_IG_FWD_WGS", 1024);   // env: tuning only
                int tpw = (mtiles * 4 + waves_m * target - 1) / (waves_m * target);
                f.tiles_per_wave = tpw < 1 ? 1 : (tpw > 8 ? 8 : tpw);
            }
            const int per_block = (4 / f.ksplit) * f.tiles_per_wave;
            dim3 grid((mtiles + per_block - 1) / per_block, 4, (L.cout + 15) / 16);
            ProfScope _p(e, a.train ? "ig_convt_fwd" : "ig_convt_eval", (int)l, f4((double)B * (L.in_elems() + L.out_elems())));
            hipLaunchKernelGGL(k_ig_fwd_s2, grid, dim3(256), (64 + 1024) * sizeof(float) + (size_t)(L.cin + 1) * sizeof(float4), s, f);
            if (a.train)
                if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
            continue;
        }
        dim3 grid(grid1((int64_t)B * L.hout * L.wout), L.cout);
        ProfScope _p(e, last ? (a.train ? "dec_convt_last_fwd_loss" : "dec_convt_last_eval") : (a.train ? "dec_convt_fwd" : "dec_convt_eval"), (int)l,
                     f4((double)B * (L.in_elems() + L.out_elems() * (last && (a.train || a.want_loss) ? 2.0 : 1.0))));
        hipLaunchKernelGGL(k_up, grid, dim3(256), lds_bytes(L.cin, L.cout), s, g, small, bns, e->params + L.w_off,
                           e->params + L.b_off, ep, bn_none(), st);
        if (a.train && !last)
            if (int rc = sync_bn_table(e, a, L.bn_index)) return rc;
    }
    return CAE_OK;
}

int launch_backward(cae_engine* e, const StepArgs& a) {
    hipStream_t s = e->stream;
    const int B = a.batch;
    const StepState* st = e->state();
    double* acc = e->gradacc();

    // ---- decoder, last layer first
    for (int l = (int)e->dec.size() - 1; l >= 0; l--) {
        const ConvLayer& L = e->dec[l];
        const bool last = l + 1 == (int)e->dec.size();
        ConvGeom g{B, L.cin, L.hin, L.win, L.cout, L.hout, L.wout, L.kh, L.kw, L.stride};
        // gradient wrt this layer's raw output
        Src gy;
        BnDesc bng = bn_none();
        if (last) {
            gy = src_plain(e->fptr(e->off_glast), L.cout, L.hout, L.wout);
        } else {
            gy = src_plain(e->fptr(L.grad_off), L.cout, L.hout, L.wout);
            gy.q = e->fptr(L.act_off);
            bng = bn_of(e, L, BN_BWD, (double)a.bn_batch * L.hout * L.wout, 0);
        }
        // this layer's input activation
        Src ain;
        BnDesc bna = bn_none();
        if (l == 0) {
            ain = src_plain(e->fptr(e->fc[3].act_off), L.cin, L.hin, L.win);
        } else {
            const ConvLayer& P = e->dec[l - 1];
            ain = src_plain(e->fptr(P.act_off), L.cin, L.hin, L.win);
            bna = bn_of(e, P, BN_SAVED, 0, 0);
        }
        if (last && !a.external_loss && last_fused_ok(e, L)) {
            S2Last f;
            memset(&f, 0, sizeof f);
            f.B = B; f.H = L.hin; f.W = L.win; f.OH = L.hout; f.OW = L.wout;
            f.in = ain.p;
            f.w = e->params + L.w_off;
            f.bias = e->params + L.b_off;
            f.target = e->ds_t[a.which];
            f.perm = a.perm;
            f.use_cursor = a.use_cursor ? 1 : 0;
            f.st = st;
            f.losses = e->losses();
            // mean over the GLOBAL batch (see launch_forward)
            f.inv_count = (float)(1.0 / ((double)a.global_batch * L.cout * L.hout * L.wout));
            f.bias_acc = e->sgacc() + L.sh_b;
            f.wacc = e->sgacc() + L.sh_w;
            f.acc_stride = e->segs.n;
            if (l == 0) {
                f.bn_in = bn_none();
                f.gin = e->fptr(e->fc[3].grad_off);
            } else {
                const ConvLayer& P = e->dec[l - 1];
                // this launch is also the forward consumer of the producer's BatchNorm: batch statistics, saved for the layers behind
                f.bn_in = bn_of(e, P, BN_BATCH, (double)a.bn_batch * P.hout * P.wout, 1);
                f.gin = e->fptr(P.grad_off);
                f.stats_in = e->bn_stats(P.bn_index);
            }
            {
                static const int dbg = env_int("CAE_HEAD_DBG", 0);   // tools/last_phases.py
                f.dbg = dbg == 4 ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
            }
            ProfScope _p(e, "s2_convt_last_fused", l, f4((double)B * (L.in_elems() * 2.0 + L.out_elems())));
            last_fused_dispatch(L, f, s);
            if (l > 0)
                if (int rc = sync_bn_table(e, a, e->dec[l - 1].bn_index)) return rc;
            continue;
        }
        if (!last && l > 0 && e->dec[l - 1].has_bn && rows_bwd_ok(e, L)) {
            const ConvLayer& P = e->dec[l - 1];
            S2Rows f;
            memset(&f, 0, sizeof f);
            f.B = B; f.H = L.hin; f.W = L.win; f.OH = L.hout; f.OW = L.wout;
            f.g = gy.p; f.yout = gy.q; f.bn_out = bng;
            f.ain = ain.p; f.bn_in = bna;
            f.w = e->params + L.w_off;
            f.gin = e->fptr(P.grad_off);
            f.stats_in = e->bn_stats(P.bn_index);
            f.wacc = e->sgacc() + L.sh_w;
            f.wacc_stride = e->segs.n;
            f.bg.stats = e->bn_stats(L.bn_index);
            f.bg.gamma_acc = acc + L.gamma_off;
            f.bg.beta_acc = acc + L.beta_off;
            f.bg.C = L.cout;
            f.bg.scale = 1.0 / a.world;
            {
                static const int dbg = env_int("CAE_HEAD_DBG", 0), dbg_layer = env_int("CAE_DBG_LAYER", 4);   // tools/last_phases.py rows
                f.dbg = dbg == 5 && l == dbg_layer ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
            }
            ProfScope _p(e, "s2_convt_bwd", l, f4((double)B * (L.out_elems() * 2.0 + L.in_elems() * 2.0)));
            rows_bwd_launch(L, f, s);
            if (int rc = sync_bn_table(e, a, P.bn_index)) return rc;
            continue;
        }
        if (s2_eligible(e, L)) {
            S2Bwd f;
            memset(&f, 0, sizeof f);
            f.B = B; f.H = L.hin; f.W = L.win; f.OH = L.hout; f.OW = L.wout;
            f.g = gy.p;
            f.yout = gy.q;
            f.bn_out = bng;
            f.ain = ain.p;
            f.bn_in = bna;
            f.w = e->params + L.w_off;
            f.wacc = e->sgacc() + L.sh_w;
            f.wacc_stride = e->segs.n;
            if (l == 0) {
                f.gin = e->fptr(e->fc[3].grad_off);
            } else {
                const ConvLayer& P = e->dec[l - 1];
                f.gin = e->fptr(P.grad_off);
                f.stats_in = e->bn_stats(P.bn_index);
            }
            if (L.has_bn) {
                f.bg.stats = e->bn_stats(L.bn_index);
                f.bg.gamma_acc = acc + L.gamma_off;
                f.bg.beta_acc = acc + L.beta_off;
                f.bg.C = L.cout;
                f.bg.scale = 1.0 / a.world;
            }
            ProfScope _p(e, "s2_convt_bwd", l,
                         f4((double)B * (L.out_elems() * (last ? 1.0 : 2.0) + L.in_elems() * 2.0)));
            s2_bwd_dispatch(L, f, s);
            if (l > 0)
                if (int rc = sync_bn_table(e, a, e->dec[l - 1].bn_index)) return rc;
            continue;
        }
        if (e->use_s2) {
            // LDS-staged backward (kernels_ctbwd.h): 3x3 kernels at stride 2, whole 16-channel blocks, images that fit the
            // staging registers / LDS, and few enough images per accumulator address; otherwise the gather pair below
            if (l < 31 && ((e->ctbwd_mask >> l) & 1) && L.kh == 3 && L.kw == 3 && L.stride == 2 && L.cin % 16 == 0 && L.cout % 4 == 0 &&
                L.hout >= 2 * L.hin + 1 && L.wout >= 2 * L.win + 1) {
                const int HW = L.hin * L.win, OHW = L.hout * L.wout, N = L.cout * 9;
                int budget = std::min((4 * kCtbG4 * kCtbThreads) / (L.cout * OHW), (4 * kCtbA4 * kCtbThreads) / (16 * HW));
                int imgs = (int)(((int64_t)L.cin * N * B + 149999) / 150000);
                imgs = std::max(1, std::min(std::min(imgs, budget), B));
                const int groups = (B + imgs - 1) / imgs;
                const int wstr = N | 1;
                const size_t lds = ct_bwd_lds_bytes(L.cin, L.cout, imgs, HW, OHW, wstr);
                if (budget >= 1 && 16 * N <= 4 * kCtbW4 * kCtbThreads && lds <= 152 * 1024 &&
                    (int64_t)L.cin * N * groups <= 400000) {
                    CtBwd c;
                    memset(&c, 0, sizeof c);
                    c.B = B; c.Cin = L.cin; c.H = L.hin; c.W = L.win; c.Cout = L.cout; c.OH = L.hout; c.OW = L.wout;
                    c.imgs = imgs; c.wstr = wstr;
                    c.g = gy.p; c.yout = gy.q; c.bn_out = bng;
                    c.ain = ain.p; c.bn_in = bna;
                    c.w = e->params + L.w_off;
                    if (L.sh_w >= 0) {
                        c.wacc = e->sgacc() + L.sh_w;
                        c.wacc_stride = e->segs.n;
                    } else {
                        c.wacc = acc + L.w_off;
                    }
                    if (l == 0) {
                        c.gin = e->fptr(e->fc[3].grad_off);
                    } else {
                        const ConvLayer& P = e->dec[l - 1];
                        c.gin = e->fptr(P.grad_off);
                        c.stats_prev = e->bn_stats(P.bn_index);
                    }
                    if (L.has_bn) {
                        c.bg.stats = e->bn_stats(L.bn_index);
                        c.bg.gamma_acc = acc + L.gamma_off;
                        c.bg.beta_acc = acc + L.beta_off;
                        c.bg.C = L.cout;
                        c.bg.scale = 1.0 / a.world;
                    }
                    {
                        static const int dbg = env_int("CAE_HEAD_DBG", 0), dbg_layer = env_int("CAE_DBG_LAYER", 2);   // tools/last_phases.py ctb
                        c.dbg = dbg == 6 && l == dbg_layer ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
                    }
                    ProfScope _p(e, "ct_convt_bwd", l,
                                 f4((double)B * (L.out_elems() * (last ? 1.0 : 2.0) + L.in_elems() * (l == 0 ? 1.0 : 2.0))));
                    static const int wg_target = env_int("CAE_CTBWD_WGS", 256);   // env: tuning only
                    static const int band_env = env_int("CAE_CTBWD_BANDS", 1);   // env: A/B measurements only
                    int parts = std::max(1, std::min(8, wg_target / (groups * (L.cin / 16))));
                    size_t lds_launch = lds;
                    if (band_env && imgs == 1 && parts > 1) {
                        // one image per workgroup and workgroups to spare: bands of input rows instead of workgroups that
                        // stage the same image (where the band's pieces fit the band kernel's staging registers)
                        const int hb = (L.hin + parts - 1) / parts, bands = (L.hin + hb - 1) / hb;
                        const int gstr = (2 * hb + 1) * L.wout, astr = hb * L.win;
                        if (bands > 1 && L.cout * gstr <= 8 * kCtbThreads && 16 * astr <= 2 * kCtbThreads) {
                            c.bands = bands;
                            c.hb = hb;
                            parts = bands;
                            lds_launch = (32 * (size_t)kCtbWaves + 4 * (size_t)(L.cin + L.cout) + (size_t)L.cout * gstr + 4 + 16 * (size_t)astr + 4 +
                                          16 * (size_t)wstr + 3 * (size_t)astr + (size_t)kCtbWaves * 16 * 17 + 8) * sizeof(float);
                        }
                    }
                    if (cae_internal::ctbwd_launch(&c, sizeof c, (unsigned)groups, (unsigned)(L.cin / 16), (unsigned)parts, lds_launch, s))
                        return fail(CAE_ERR_ARG, "k_ct_bwd_lds: argument layout mismatch");
                    if (l > 0)
                        if (int rc = sync_bn_table(e, a, e->dec[l - 1].bn_index)) return rc;
                    continue;
                }
            }
            IgWgrad fw;
            memset(&fw, 0, sizeof fw);
            fw.B = B; fw.Cin = L.cin; fw.H = L.hin; fw.W = L.win; fw.Cout = L.cout; fw.OH = L.hout; fw.OW = L.wout;
            fw.KH = L.kh; fw.KW = L.kw; fw.S = L.stride;
            fw.ain = ain.p; fw.bn_in = bna; fw.g = gy.p; fw.yout = gy.q; fw.bn_out = bng;
            fw.wacc = acc + L.w_off;
            if (L.has_bn) {
                fw.bg.stats = e->bn_stats(L.bn_index);
                fw.bg.gamma_acc = acc + L.gamma_off;
                fw.bg.beta_acc = acc + L.beta_off;
                fw.bg.C = L.cout;
                fw.bg.scale = 1.0 / a.world;
            }
            const int wtiles = ((L.cin + 15) / 16) * ((L.cout * L.kh * L.kw + 15) / 16);
            const int steps = (B * L.hin * L.win + 3) / 4;
            static const int wgrad_target = env_int("CAE_IG_WGRAD_WGS", 2048);   // env: tuning only
            int chunks = wgrad_target / wtiles;
            if (chunks < 1) chunks = 1;
            int per = (steps + chunks - 1) / chunks;
            per = (per + 31) / 32 * 32;
            chunks = (steps + per - 1) / per;
            fw.ksteps_per_block = per;

            IgDgrad fd;
            memset(&fd, 0, sizeof fd);
            fd.B = B; fd.Cin = L.cin; fd.H = L.hin; fd.W = L.win; fd.Cout = L.cout; fd.OH = L.hout; fd.OW = L.wout;
            fd.KH = L.kh; fd.KW = L.kw; fd.S = L.stride;
            fd.g = gy.p; fd.yout = gy.q; fd.bn_out = bng; fd.w = e->params + L.w_off;
            if (l == 0) {
                fd.gin = e->fptr(e->fc[3].grad_off);
            } else {
                const ConvLayer& P = e->dec[l - 1];
                fd.gin = e->fptr(P.grad_off);
                fd.yprev = e->fptr(P.act_off);
                fd.bn_prev = bn_of(e, P, BN_SAVED, 0, 0);
                fd.stats_prev = e->bn_stats(P.bn_index);
            }
            const int mtiles = (B * L.hin * L.win + 15) / 16;
            const int ksteps = (L.cout * L.kh * L.kw + 3) / 4;
            fd.ksplit = ksteps > 24 ? 4 : (ksteps > 12 ? 2 : 1);   // <= 12 k-steps (one load batch) per wave where possible
            static const int tpw_env = env_int("CAE_IG_DGRAD_TPW", 0);   // env: tuning only
            fd.tiles_per_wave = tpw_env > 0 ? tpw_env : (mtiles >= 8192 ? 2 : 1);
            const int per_block = (4 / fd.ksplit) * fd.tiles_per_wave;
            const int d_gx = (mtiles + per_block - 1) / per_block, d_gy = (L.cin + 15) / 16;
            const size_t lds_d = (128 + 1024) * sizeof(float) + (size_t)(L.cin + L.cout + 1) * sizeof(float4) +
                                 (size_t)L.cout * L.kh * L.kw * 2 * sizeof(int);
            const size_t lds_w = 1024 * sizeof(float) + (size_t)(L.cin + L.cout + 1) * sizeof(float4);
            ProfScope _p(e, "ig_convt_bwd_pair", l,
                         f4((double)B * (L.out_elems() * (last ? 1.0 : 2.0) + L.in_elems() * (l == 0 ? 1.0 : 2.0))));
            {
                static const int dbg = env_int("CAE_HEAD_DBG", 0);   // tools/head_phases.py igb: stamps of decoder layer 2's pair
                fw.dbg = fd.dbg = dbg == 2 && l == 2 ? reinterpret_cast<long long*>(e->ws + e->off_scan) : nullptr;
            }
            // XCD-aware order (kernels_igemm.h): d_group input-gradient blocks cover the positions of one weight-gradient chunk
            int d_group = (per * 4) / (per_block * 16);
            if (d_group < 1) d_group = 1;
            const int w_n8 = (chunks + 7) / 8, d_n8 = ((d_gx + d_group - 1) / d_group + 7) / 8;
            hipLaunchKernelGGL(k_ig_bwd_pair, dim3(8 * wtiles * w_n8 + 8 * d_n8 * d_group * d_gy), dim3(256),
                               lds_d > lds_w ? lds_d : lds_w, s, fw, fd, wtiles, chunks, w_n8, d_gx, d_gy, d_group);
            if (l > 0)
                if (int rc = sync_bn_table(e, a, e->dec[l - 1].bn_index)) return rc;
            continue;
        }
        // weight gradient (+ BN parameter gradients of this layer)
        {
            const int64_t nw = (int64_t)L.cin * L.cout * L.kh * L.kw;
            const int64_t pos = (int64_t)B * L.hin * L.win;
            const int ppb = wgrad_ppb(pos, nw);
            BnGradOut bg;
            memset(&bg, 0, sizeof bg);
            if (L.has_bn) {
                bg.stats = e->bn_stats(L.bn_index);
                bg.gamma_acc = acc + L.gamma_off;
                bg.beta_acc = acc + L.beta_off;
                bg.C = L.cout;
                bg.scale = 1.0 / a.world;
            }
            dim3 grid((unsigned)nw, (unsigned)((pos + ppb - 1) / ppb));
            ProfScope _p(e, "dec_convt_wgrad", l, f4((double)B * (L.in_elems() + L.out_elems() * (last ? 1.0 : 2.0))));
            hipLaunchKernelGGL(k_wgrad, grid, dim3(256), lds_bytes(L.cin, L.cout), s, g, ain, bna, gy, bng,
                               acc + L.w_off, ppb, bg, st);
        }
        // input gradient
        {
            Epi ep;
            BnDesc bne = bn_none();
            if (l == 0) {
                ep = epi_plain(e->fptr(e->fc[3].grad_off));
            } else {
                const ConvLayer& P = e->dec[l - 1];
                ep = epi_plain(e->fptr(P.grad_off));
                ep.kind = EPI_MASKSTATS;
                ep.stats = e->bn_stats(P.bn_index);
                ep.stats_C = P.cout;
                ep.yprev = e->fptr(P.act_off);
                bne = bn_of(e, P, BN_SAVED, 0, 0);
            }
            dim3 grid(grid1((int64_t)B * L.hin * L.win), L.cin);
            ProfScope _p(e, "dec_convt_dgrad", l, f4((double)B * (L.out_elems() * (last ? 1.0 : 2.0) + L.in_elems() * (l == 0 ? 1.0 : 2.0))));
            hipLaunchKernelGGL(k_down, grid, dim3(256), lds_bytes(L.cout, L.cin), s, g, gy, bng, e->params + L.w_off,
                               (const float*)nullptr, ep, bne, st);
            if (l > 0)
                if (int rc = sync_bn_table(e, a, e->dec[l - 1].bn_index)) return rc;
        }
    }
    // ---- Linear layers, last first.  grad_off of fc[i] holds dL/d(pre-activation of fc[i] output).
    {
        const ConvLayer& P = e->enc.back();
        const int hw = P.hout * P.wout;
        TailArgs tail;
        size_t tail_lds = 0;
        const bool fused_tail = tail_plan(e, a, tail, tail_lds);
        for (int i = 3; i >= 0; i--) {
            const FcLayer& F = e->fc[i];
            if (i == 2)   // every decoder conv gradient and Linear 3's are complete: the first gradient bucket can leave
                if (int rc = dp_first_bucket(e, a)) return rc;
            if (fused_tail && i == 2) {
                double bytes = 0;
                for (int j = 0; j < 3; j++)
                    bytes += f4((double)B * (3.0 * e->fc[j].nin + 2.0 * e->fc[j].nout) + (double)e->fc[j].nin * e->fc[j].nout) +
                             8.0 * e->fc[j].nin * e->fc[j].nout;
                static const int dbg = env_int("CAE_HEAD_DBG", 0);
                tail.dbg = dbg ? reinterpret_cast<long long*>(e->fptr(e->fc[3].grad_off)) : nullptr;   // Linear 3's gradient is dead by now
                head_lds_attr(k_tail_bwd, tail_lds);
                ProfScope _p(e, "tail_bwd", 0, bytes);
                hipLaunchKernelGGL(k_tail_bwd, dim3((B + 15) / 16, 4), dim3(kHeadThreads), tail_lds, s, tail);
                break;
            }
            const float* gout = e->fptr(F.grad_off);
            const float* in = i == 0 ? e->fptr(P.act_off) : (i == 2 && e->variational ? e->fptr(e->off_vz) : e->fptr(e->fc[i - 1].act_off));
            BnDesc bni = i == 0 ? bn_of(e, P, BN_SAVED, 0, 0) : bn_none();
            if (e->use_s2) {
                // weight gradient: dW[o][i] = sum_b gout[b][o] * in[b][i], db[o] = sum_b gout[b][o] (ones column)
                GemmArgs gw;
                memset(&gw, 0, sizeof gw);
                gw.M = F.nout; gw.N = F.nin + 1; gw.K = B;
                gw.A = gout; gw.sa_m = 1; gw.sa_k = F.nout;       // A[m=o][k=b] = gout[b][o]
                gw.B = in; gw.sb_k = F.nin; gw.sb_n = 1;          // B[k=b][n=i] = in[b][i]
                gw.epi = GE_ACC64;
                gw.accW = acc + F.w_off; gw.accB = acc + F.b_off; gw.ones_col = 1;
                // input gradient: gin[b][i] = mask( sum_o gout[b][o] * W[o][i] )
                GemmArgs gd;
                memset(&gd, 0, sizeof gd);
                gd.M = B; gd.N = F.nin; gd.K = F.nout;
                gd.A = gout; gd.sa_m = F.nout; gd.sa_k = 1;
                gd.B = e->params + F.w_off; gd.sb_k = F.nin; gd.sb_n = 1;   // B[k=o][n=i] = W[o][i]
                gd.sc_m = F.nin; gd.sc_n = 1;
                size_t lds = gemm_lds(0);
                if (i == 2 && e->variational) {
                    gd.C = e->fptr(e->off_vgz);   // dL/dz; the hook below turns it into the heads' gradient
                    gd.epi = GE_STORE;
                } else if (i > 0) {
                    const FcLayer& G = e->fc[i - 1];
                    gd.C = e->fptr(G.grad_off);
                    gd.epi = G.relu ? GE_RELU_MASK : GE_STORE;
                    gd.H = e->fptr(G.act_off);
                } else {
                    gd.C = e->fptr(P.grad_off);
                    gd.epi = GE_BN_MASK;
                    gd.H = e->fptr(P.act_off);
                    gd.bn_c = bni; gd.hw_c = hw;
                    gd.stats_c = e->bn_stats(P.bn_index);
                    lds = gemm_lds(P.cout);
                }
                const int tiles_d = ((gd.M + 15) / 16) * ((gd.N + 15) / 16);
                if (i == 0) {
                    // the first encoder Linear's input carries BatchNorm+ReLU: compute dW^T = act(in)^T * gout so
                    // the transform sits on the A operand (channel = row / hw), store transposed; the ones ROW
                    // of A yields the bias gradient
                    gw.M = F.nin + 1; gw.N = F.nout; gw.K = B;
                    gw.A = in; gw.sa_m = 1; gw.sa_k = F.nin;            // A[m=i][k=b] = in[b][i]
                    gw.B = gout; gw.sb_k = F.nout; gw.sb_n = 1;         // B[k=b][n=o] = gout[b][o]
                    gw.epi = GE_ACC64_T;
                    gw.ones_col = 0; gw.ones_row = 1;
                    gw.bn_a = bni; gw.hw_a = hw; gw.bn_a_by_row = 1;
                }
                {
                    const int tiles_w = ((gw.M + 15) / 16) * ((gw.N + 15) / 16);
                    ProfScope _p(e, "linear_bwd_pair_mfma", i,
                                 f4((double)B * (3.0 * F.nin + 2.0 * F.nout) + (double)F.nin * F.nout) + 8.0 * F.nin * F.nout);
                    // the input gradient's contraction runs over nout: long for the last decoder Linear (576 at cfg2): burst variant
                    static const int burst_on = env_int("CAE_GEMM_BURST", 1);   // env: A/B measurements only
                    const bool burst = burst_on && gd.epi != GE_BN_MASK && gd.sa_k == 1 && gd.sb_n == 1 && gd.K % 4 == 0 && gd.K >= 128 &&
                                       (gd.K / 4 + 3) / 4 <= kBurstSteps && gd.sa_m % 4 == 0;
                    if (burst && gemm16_burst_lds(gd.K) > lds) lds = gemm16_burst_lds(gd.K);
                    hipLaunchKernelGGL(k_gemm16_pair, dim3(tiles_w + tiles_d), dim3(256), lds, s, gw, gd, tiles_w, burst ? 1 : 0);
                }
                if (i == 0)
                    if (int rc = sync_bn_table(e, a, P.bn_index)) return rc;
                if (i == 2 && e->variational) {
                    if (!e->hooks.reparam_bwd) return fail(CAE_ERR_STATE, "trunk engine without a reparameterisation hook");
                    e->hooks.reparam_bwd(e->hooks.user, s, e->fptr(e->off_vgz), e->fptr(e->fc[1].act_off), B, e->latent,
                                         e->fptr(e->fc[1].grad_off));
                }
                continue;
            }
            if (e->variational) return fail(CAE_ERR_STATE, "the trunk mode needs the specialised kernels (cae_set_kernel_mode)");
            {
                ProfScope _p(e, "linear_wgrad", i, f4((double)B * (F.nin + F.nout)) + 8.0 * F.nin * F.nout);
                hipLaunchKernelGGL(k_lin_wgrad, dim3(grid1((int64_t)F.nin * F.nout)), dim3(256),
                                   lds_bytes(i == 0 ? P.cout : 0, 0), s, B, F.nin, F.nout, gout, in, bni, hw,
                                   acc + F.w_off, acc + F.b_off);
            }
            if (i > 0) {
                const FcLayer& G = e->fc[i - 1];
                ProfScope _p(e, "linear_dgrad", i, f4((double)B * (2.0 * F.nin + F.nout) + (double)F.nin * F.nout));
                hipLaunchKernelGGL(k_lin_dgrad, dim3(grid1((int64_t)B * F.nin)), dim3(256), lds_bytes(0, 0), s, B,
                                   F.nin, F.nout, gout, e->params + F.w_off, G.relu ? 1 : 0, e->fptr(G.act_off),
                                   bn_none(), 1, (double*)nullptr, e->fptr(G.grad_off));
            } else {
                dim3 grid(grid1((int64_t)B * hw), P.cout);
                ProfScope _p(e, "linear_dgrad", i, f4((double)B * (2.0 * F.nin + F.nout) + (double)F.nin * F.nout));
                hipLaunchKernelGGL(k_lin_dgrad, grid, dim3(256), lds_bytes(P.cout, 0), s, B, F.nin, F.nout, gout,
                                   e->params + F.w_off, 2, e->fptr(P.act_off), bni, hw, e->bn_stats(P.bn_index),
                                   e->fptr(P.grad_off));
                if (int rc = sync_bn_table(e, a, P.bn_index)) return rc;
            }
        }
    }
    // ---- encoder convs
    for (int l = (int)e->enc.size() - 1; l >= 0; l--) {
        const ConvLayer& L = e->enc[l];
        ConvGeom g{B, L.cout, L.hout, L.wout, L.cin, L.hin, L.win, L.kh, L.kw, L.stride};
        Src gy = src_plain(e->fptr(L.grad_off), L.cout, L.hout, L.wout);
        gy.q = e->fptr(L.act_off);
        BnDesc bng = bn_of(e, L, BN_BWD, (double)a.bn_batch * L.hout * L.wout, 0);
        Src ain;
        BnDesc bna = bn_none();
        if (l == 0 && a.x_direct) {
            ain = src_plain(a.x_direct, L.cin, L.hin, L.win);
        } else if (l == 0) {
            ain = src_plain(e->ds_x[a.which], L.cin, L.hin, L.win);
            ain.perm = a.perm;
            ain.use_cursor = 1;
        } else {
            const ConvLayer& P = e->enc[l - 1];
            ain = src_plain(e->fptr(P.act_off), L.cin, L.hin, L.win);
            bna = bn_of(e, P, BN_SAVED, 0, 0);
        }
        if (l > 0 && e->use_s2 && !a.syncing()) {
            // weight gradient and input gradient share only their inputs: one launch (kernels_generic.h k_conv_bwd_pair)
            const ConvLayer& P = e->enc[l - 1];
            const int64_t nw = (int64_t)L.cin * L.cout * L.kh * L.kw;
            const int64_t pos = (int64_t)B * L.hout * L.wout;
            WgradArgs wa;
            memset(&wa, 0, sizeof wa);
            wa.g = g; wa.small = gy; wa.bns = bng; wa.big = ain; wa.bnb = bna;
            wa.acc = acc + L.w_off;
            wa.ppb = wgrad_ppb(pos, nw);
            wa.bg.stats = e->bn_stats(L.bn_index);
            wa.bg.gamma_acc = acc + L.gamma_off;
            wa.bg.beta_acc = acc + L.beta_off;
            wa.bg.C = L.cout;
            wa.bg.scale = 1.0 / a.world;
            UpArgs ua;
            memset(&ua, 0, sizeof ua);
            ua.g = g; ua.small = gy; ua.bns = bng; ua.w = e->params + L.w_off; ua.bias = nullptr;
            ua.e = epi_plain(e->fptr(P.grad_off));
            ua.e.kind = EPI_MASKSTATS;
            ua.e.stats = e->bn_stats(P.bn_index);
            ua.e.stats_C = P.cout;
            ua.e.yprev = e->fptr(P.act_off);
            ua.bne = bn_of(e, P, BN_SAVED, 0, 0);
            const int nwy = (int)((pos + wa.ppb - 1) / wa.ppb), ux = grid1((int64_t)B * L.hin * L.win);
            ProfScope _p(e, "enc_conv_bwd_pair", l, f4((double)B * (3.0 * L.in_elems() + 4.0 * L.out_elems())));
            hipLaunchKernelGGL(k_conv_bwd_pair, dim3((unsigned)(nw * nwy + (int64_t)ux * L.cin)), dim3(256), lds_bytes(L.cout, L.cin), s,
                               wa, ua, (int)nw, nwy, ux, st);
            continue;
        }
        memset(&e->c0_pending, 0, sizeof e->c0_pending);
        if (l == 0 && a.adam_follows && e->x_published && e->use_s2 && !a.syncing() && a.world == 1 && L.cout <= 64 &&
            (int64_t)L.cin * L.cout * L.kh * L.kw <= 4096 && (int64_t)B * L.hout * L.wout < kDivSmallMaxN &&
            L.hout * L.wout < kDivSmallMaxD && e->bn_stat_off[L.bn_index] == e->off_zero_begin) {
            // the last launch of backward folds into the optimiser launch (kernels_generic.h AdamConv0)
            AdamConv0& c0 = e->c0_pending;
            c0.on = 1;
            c0.nw = L.cin * L.cout * L.kh * L.kw;
            c0.C = L.cout;
            c0.w_off = L.w_off; c0.gamma_off = L.gamma_off; c0.beta_off = L.beta_off;
            c0.g = g; c0.small = gy; c0.bns = bng;
            c0.xb = e->fptr(e->off_xbatch);
            c0.bns.gamma = c0.xb + (int64_t)B * L.in_elems();   // k_head_fwd's copy: this launch rewrites the parameter itself
            c0.stats = e->bn_stats(L.bn_index);
            c0.scale = 1.0;
            continue;
        }
        {
            const int64_t nw = (int64_t)L.cin * L.cout * L.kh * L.kw;
            const int64_t pos = (int64_t)B * L.hout * L.wout;
            const int ppb = wgrad_ppb(pos, nw);
            BnGradOut bg;
            memset(&bg, 0, sizeof bg);
            bg.stats = e->bn_stats(L.bn_index);
            bg.gamma_acc = acc + L.gamma_off;
            bg.beta_acc = acc + L.beta_off;
            bg.C = L.cout;
            bg.scale = 1.0 / a.world;
            dim3 grid((unsigned)nw, (unsigned)((pos + ppb - 1) / ppb));
            ProfScope _p(e, "enc_conv_wgrad", l, f4((double)B * (L.in_elems() + 2.0 * L.out_elems())), s);
            hipLaunchKernelGGL(k_wgrad, grid, dim3(256), lds_bytes(L.cout, L.cin), s, g, gy, bng, ain, bna,
                               acc + L.w_off, ppb, bg, st);
        }
        if (l > 0) {
            const ConvLayer& P = e->enc[l - 1];
            Epi ep = epi_plain(e->fptr(P.grad_off));
            ep.kind = EPI_MASKSTATS;
            ep.stats = e->bn_stats(P.bn_index);
            ep.stats_C = P.cout;
            ep.yprev = e->fptr(P.act_off);
            BnDesc bne = bn_of(e, P, BN_SAVED, 0, 0);
            dim3 grid(grid1((int64_t)B * L.hin * L.win), L.cin);
            ProfScope _p(e, "enc_conv_dgrad", l, f4((double)B * (2.0 * L.out_elems() + 2.0 * L.in_elems())));
            hipLaunchKernelGGL(k_up, grid, dim3(256), lds_bytes(L.cout, L.cin), s, g, gy, bng, e->params + L.w_off,
                               (const float*)nullptr, ep, bne, st);
            if (int rc = sync_bn_table(e, a, P.bn_index)) return rc;
        }
    }
    return CAE_OK;
}

int launch_one(cae_engine* e, int op, const StepArgs& a);

int launch_op(cae_engine* e, int op, const StepArgs& a) {
    // the cursor lives on the device, so the same launch sequence repeated n times walks n batches:
    // n steps become one graph and the ~8.5 us the GPU idles between two graph replays is paid once
    for (int i = 0; i < a.nsteps; i++)
        if (int rc = launch_one(e, op, a)) return rc;
    return CAE_OK;
}

int launch_one(cae_engine* e, int op, const StepArgs& a) {
    hipStream_t s = e->stream;
    {   // while profiling: one EMPTY bracket per step = what an event pair itself adds to every bracketed launch
        ProfScope _cal(e, "event_pair", -1, 0.0);
    }
    if (op == OP_DP_TRAIN) {
        e->sync_pos = 0;
        if (a.batch > 0) {
            int rc = launch_forward(e, a);
            if (rc) return rc;
            rc = launch_backward(e, a);
            if (rc) return rc;
        } else {
            // a rank whose shard of a short last batch is empty: no kernels, but every collective of the step in order
            hipLaunchKernelGGL(k_bump_adam, dim3(1), dim3(1), 0, s, e->state());
            if (a.dp_sync)
                for (int bn : e->sync_order)
                    if (int rc = sync_bn_table(e, a, bn)) return rc;
            if (int rc = dp_first_bucket(e, a)) return rc;
        }
        if (a.dp_sync && e->sync_pos != e->sync_order.size())
            return fail(CAE_ERR_STATE, "SyncBN: %zu of %zu tables all-reduced", e->sync_pos, e->sync_order.size());
        if (int rc = dp_finish_step(e, a)) return rc;
    } else if (op == OP_TRAIN || op == OP_FWDBWD) {
        // the accumulators were zeroed by the previous step's last kernel (k_adam / k_acc_to_f32) or by
        // the caller's zero-filled workspace on the very first step
        static const int fuse_c0 = env_int("CAE_ADAM_CONV0", 1);   // env: A/B measurements only
        StepArgs af = a;
        af.adam_follows = op == OP_TRAIN && fuse_c0 != 0;
        memset(&e->c0_pending, 0, sizeof e->c0_pending);
        int rc = launch_forward(e, af);
        if (rc) return rc;
        rc = launch_backward(e, af);
        if (rc) return rc;
        if (op == OP_TRAIN) {
            ProfScope _p(e, "adam", 0, 32.0 * e->n_param);
            AdamConv0 c0 = e->c0_pending;
            StepTail tl = step_tail_of(e, a.inc(), 1);
            int nreg = grid1(e->n_param), grid = nreg;
            if (c0.on) {
                // that layer's BatchNorm table (the first of the swept range) is read by this launch: the next step's
                // k_head_fwd clears it
                const long long skip = (long long)kStatShards * c0.C * 4;
                tl.zero_extra += skip;
                tl.zero_extra_n -= skip;
                c0.n_regular = nreg;
                grid = nreg + c0.nw;
            }
            hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, s, (long long)e->n_param, e->params,
                               (const float*)nullptr, e->m, e->v, e->hp, (const StepState*)e->state(), e->shard_segs(),
                               tl, 0, std::log(e->hp.beta1), std::log(e->hp.beta2), c0);
        } else {
            hipLaunchKernelGGL(k_acc_to_f32, dim3(grid1(e->n_param)), dim3(256), 0, s, (long long)e->n_param, e->grads,
                               e->shard_segs(), step_tail_of(e, a.inc(), 1));
        }
    } else if (op == OP_EVAL) {
        int rc = launch_forward(e, a);
        if (rc) return rc;
        if (a.use_cursor) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, e->state(), a.inc(), 1, 0);
    } else if (op == OP_ADAM) {
        StepTail none;
        memset(&none, 0, sizeof none);
        // forward_backward already counted this optimiser step (its first kernel bumps adam_step)
        hipLaunchKernelGGL(k_adam, dim3(grid1(e->n_param)), dim3(256), 0, s, (long long)e->n_param, e->params,
                           (const float*)e->grads, e->m, e->v, e->hp, (const StepState*)e->state(), e->shard_segs(),
                           none, 0, std::log(e->hp.beta1), std::log(e->hp.beta2), AdamConv0{});
    }
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

// run an op either directly or through a cached hipGraph
int run_op(cae_engine* e, int op, const StepArgs& a, bool cacheable) {
    // the legacy NULL stream cannot be captured: plain launches there
    if (!e->graph_mode || !cacheable || e->stream == nullptr || e->profiling)
        return e->capture_only ? CAE_OK : launch_op(e, op, a);
    // a SyncBN step and a per-rank-BatchNorm step of the same sizes are DIFFERENT launch sequences (table all-reduces,
    // bn_batch in every BatchNorm descriptor, the 1/world scale of the BatchNorm parameter gradients)
    auto key = std::make_tuple(op, a.which, a.batch, a.global_batch, (const void*)a.perm, a.nsteps, a.cursor_inc,
                               a.dp_sync ? 1 : 0, a.bn_batch, a.world);
    auto it = e->graphs.find(key);
    if (it == e->graphs.end()) {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        int rc = launch_op(e, op, a);
        hipError_t ce = hipStreamEndCapture(e->stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (ce != hipSuccess) return fail(CAE_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        hipGraphExec_t exec = nullptr;
        hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail(CAE_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
        it = e->graphs.emplace(key, exec).first;
    }
    if (e->capture_only) return CAE_OK;
    HIP_TRY(hipGraphLaunch(it->second, e->stream));
    return CAE_OK;
}

int check_ready(const cae_engine* e, int which, int batch, bool need_target) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    if (!e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (batch < 1 || batch > e->max_batch) return fail(CAE_ERR_ARG, "batch %d outside [1, %d]", batch, e->max_batch);
    if (which < 0 || which > 1) return fail(CAE_ERR_ARG, "dataset index %d is not 0 or 1", which);
    if (!e->ds_x[which]) return fail(CAE_ERR_STATE, "dataset %d has not been set", which);
    if (need_target && !e->ds_t[which]) return fail(CAE_ERR_STATE, "dataset %d has no target array", which);
    return CAE_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================

extern "C" {

const char* cae_last_error(void) { return g_err.c_str(); }
int cae_abi_version(void) { return 1; }

static int engine_create_impl(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                              int latent_size, int max_batch, bool variational, cae_engine** out);

int cae_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                      int latent_size, int max_batch, cae_engine** out) {
    return engine_create_impl(enc, n_enc, dec, n_dec, fc_size, latent_size, max_batch, false, out);
}

static int engine_create_impl(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size,
                              int latent_size, int max_batch, bool variational, cae_engine** out) {
    if (!enc || !dec || !out || n_enc < 1 || n_dec < 1) return fail(CAE_ERR_ARG, "need >=1 encoder and decoder layer");
    if (fc_size < 1 || latent_size < 1 || max_batch < 1) return fail(CAE_ERR_ARG, "fc/latent/max_batch must be >= 1");
    cae_engine* e = new cae_engine();
    e->fc_size = fc_size;
    e->latent = latent_size;
    e->max_batch = max_batch;
    e->variational = variational;
    auto bad = [&](const char* msg, int i) {
        delete e;
        return fail(CAE_ERR_ARG, "layer %d: %s", i, msg);
    };
    // ---- geometry checks
    for (int i = 0; i < n_enc; i++) {
        const cae_layer_spec& s = enc[i];
        if (s.stride < 1 || s.k_h < 1 || s.k_w < 1 || s.in_c < 1 || s.out_c < 1) return bad("bad encoder spec", i);
        if (s.in_h < s.k_h || s.in_w < s.k_w) return bad("encoder kernel larger than input", i);
        if (s.out_h != (s.in_h - s.k_h) / s.stride + 1 || s.out_w != (s.in_w - s.k_w) / s.stride + 1)
            return bad("encoder output size != floor((in-k)/stride)+1", i);
        if (i > 0 && (enc[i - 1].out_c != s.in_c || enc[i - 1].out_h != s.in_h || enc[i - 1].out_w != s.in_w))
            return bad("encoder layer input does not match previous output", i);
    }
    for (int i = 0; i < n_dec; i++) {
        const cae_layer_spec& s = dec[i];
        if (s.stride < 1 || s.k_h < 1 || s.k_w < 1 || s.in_c < 1 || s.out_c < 1) return bad("bad decoder spec", i);
        if (s.output_padding < 0 || (s.output_padding >= s.stride && s.output_padding > 0))
            return bad("output_padding must be smaller than stride", i);
        if (s.out_h != (s.in_h - 1) * s.stride + s.k_h + s.output_padding ||
            s.out_w != (s.in_w - 1) * s.stride + s.k_w + s.output_padding)
            return bad("decoder output size != (in-1)*stride+k+output_padding", i);
        if (i > 0 && (dec[i - 1].out_c != s.in_c || dec[i - 1].out_h != s.in_h || dec[i - 1].out_w != s.in_w))
            return bad("decoder layer input does not match previous output", i);
    }
    e->in_c = enc[0].in_c;
    e->in_h = enc[0].in_h;
    e->in_w = enc[0].in_w;
    e->out_c = dec[n_dec - 1].out_c;
    e->out_h = dec[n_dec - 1].out_h;
    e->out_w = dec[n_dec - 1].out_w;

    // ---- parameter / buffer arenas in the reference's state_dict order
    int bn = 0;
    for (int i = 0; i < n_enc; i++) {
        const cae_layer_spec& s = enc[i];
        ConvLayer L{};
        L.transposed = false;
        L.cin = s.in_c; L.hin = s.in_h; L.win = s.in_w;
        L.cout = s.out_c; L.hout = s.out_h; L.wout = s.out_w;
        L.kh = s.k_h; L.kw = s.k_w; L.stride = s.stride; L.opad = 0;
        L.has_bn = true;
        L.bn_index = bn++;
        const std::string c = "enc/encoder_cnn." + std::to_string(3 * i);
        const std::string b = "enc/encoder_cnn." + std::to_string(3 * i + 1);
        add_tensor(e, c + ".weight", 0, {L.cout, L.cin, L.kh, L.kw}, &L.w_off);
        add_tensor(e, c + ".bias", 0, {L.cout}, &L.b_off);
        add_tensor(e, b + ".weight", 0, {L.cout}, &L.gamma_off);
        add_tensor(e, b + ".bias", 0, {L.cout}, &L.beta_off);
        add_tensor(e, b + ".running_mean", 1, {L.cout}, &L.rm_off);
        add_tensor(e, b + ".running_var", 1, {L.cout}, &L.rv_off);
        e->enc.push_back(L);
    }
    const ConvLayer& EL = e->enc.back();
    const int flat_enc = EL.cout * EL.hout * EL.wout;
    const int flat_dec = dec[0].in_c * dec[0].in_h * dec[0].in_w;
    const int heads = variational ? 2 * latent_size : latent_size;   // trunk mode: [mu | logvar]
    const int fdims[4][2] = {{flat_enc, fc_size}, {fc_size, heads}, {latent_size, fc_size}, {fc_size, flat_dec}};
    const char* fnames[4] = {"enc/encoder_lin.0", "enc/encoder_lin.2", "dec/decoder_lin.0", "dec/decoder_lin.2"};
    for (int i = 0; i < 4; i++) {
        FcLayer& F = e->fc[i];
        F.nin = fdims[i][0];
        F.nout = fdims[i][1];
        F.relu = (i == 0 || i == 2);
        if (i == 1 && variational) {
            // two named tensors per half, ONE (2 * latent, fc) matrix and ONE (2 * latent) bias vector in the arena: the table
            // lists mu.weight, mu.bias, logvar.weight, logvar.bias (the 'var' model's state_dict order) with those offsets
            int64_t& top = e->n_param;
            top = align_up(top, 4);
            F.w_off = top;
            top += (int64_t)heads * F.nin;
            top = align_up(top, 4);
            F.b_off = top;
            top += heads;
            const char* hn[2] = {"enc/encoder_mu", "enc/encoder_logvar"};
            for (int h = 0; h < 2; h++)
                for (int part = 0; part < 2; part++) {
                    cae_tensor_info_t t;
                    memset(&t, 0, sizeof t);
                    snprintf(t.name, sizeof t.name, "%s.%s", hn[h], part == 0 ? "weight" : "bias");
                    t.arena = 0;
                    t.ndim = part == 0 ? 2 : 1;
                    t.shape[0] = latent_size;
                    if (part == 0) t.shape[1] = F.nin;
                    t.numel = part == 0 ? (int64_t)latent_size * F.nin : latent_size;
                    t.offset = part == 0 ? F.w_off + (int64_t)h * latent_size * F.nin : F.b_off + (int64_t)h * latent_size;
                    e->tensors.push_back(t);
                }
            continue;
        }
        add_tensor(e, std::string(fnames[i]) + ".weight", 0, {F.nout, F.nin}, &F.w_off);
        add_tensor(e, std::string(fnames[i]) + ".bias", 0, {F.nout}, &F.b_off);
    }
    for (int i = 0; i < n_dec; i++) {
        const cae_layer_spec& s = dec[i];
        ConvLayer L{};
        L.transposed = true;
        L.cin = s.in_c; L.hin = s.in_h; L.win = s.in_w;
        L.cout = s.out_c; L.hout = s.out_h; L.wout = s.out_w;
        L.kh = s.k_h; L.kw = s.k_w; L.stride = s.stride; L.opad = s.output_padding;
        L.has_bn = i != n_dec - 1;
        L.bn_index = L.has_bn ? bn++ : -1;
        const std::string c = "dec/decoder_conv." + std::to_string(3 * i);
        const std::string b = "dec/decoder_conv." + std::to_string(3 * i + 1);
        add_tensor(e, c + ".weight", 0, {L.cin, L.cout, L.kh, L.kw}, &L.w_off);
        add_tensor(e, c + ".bias", 0, {L.cout}, &L.b_off);
        if (L.has_bn) {
            add_tensor(e, b + ".weight", 0, {L.cout}, &L.gamma_off);
            add_tensor(e, b + ".bias", 0, {L.cout}, &L.beta_off);
            add_tensor(e, b + ".running_mean", 1, {L.cout}, &L.rm_off);
            add_tensor(e, b + ".running_var", 1, {L.cout}, &L.rv_off);
        }
        e->dec.push_back(L);
    }
    e->n_param = align_up(e->n_param, 4);
    e->n_buf = align_up(e->n_buf, 4);
    e->n_bn = bn;
    // data parallelism: bucket boundary and the SyncBN collective order (forward: producers in layer order; backward: the
    // table of a layer's INPUT BatchNorm after that layer's input-gradient kernel) - see launch_forward / launch_backward
    e->bucket_split = e->fc[3].w_off;
    for (auto& L : e->enc) e->sync_order.push_back(L.bn_index);
    for (auto& L : e->dec)
        if (L.has_bn) e->sync_order.push_back(L.bn_index);
    for (int l = n_dec - 1; l >= 1; l--) e->sync_order.push_back(e->dec[l - 1].bn_index);
    e->sync_order.push_back(e->enc.back().bn_index);
    for (int l = n_enc - 1; l >= 1; l--) e->sync_order.push_back(e->enc[l - 1].bn_index);

    // ---- sharded accumulators for the layers the stride-2 kernels can take
    {
        int n = 0;
        for (size_t l = 0; l < e->dec.size(); l++) {
            ConvLayer& L = e->dec[l];
            // (... and the layers of the LDS-staged backward kernel, kernels_ctbwd.h: its workgroups all reach their weight-
            // gradient atomics at the same moment, one per image group and address; unsharded they queue up behind the
            // kernel's end - measured 8 us of a 16 us launch at the benchmark's third layer)
            // Which layers take it by default: one block of 16 input channels (no staging repeated per block) and few enough
            // weights that the optimiser's eight shard reads per weight stay cheap - at the benchmark geometry the 16 -> 8
            // layer: 169.6 against 171.5 us per step; the 32 -> 16 and 64 -> 32 layers lose (172.9 / 175.7 on their own).
            static const int ctb_env = env_int("CAE_CTBWD", -1);   // env: bit l = decoder layer l on the LDS-staged backward kernel
            const bool ctb_shape = l < 31 && L.transposed && L.stride == 2 && L.kh == 3 && L.kw == 3 && L.cin % 16 == 0 &&
                                   L.cout % 4 == 0;
            const bool ctb = ctb_shape && (ctb_env >= 0 ? ((ctb_env >> l) & 1) != 0 : (L.cin == 16 && L.cin * L.cout * 9 <= 2048));
            if (l == 0) e->ctbwd_auto = 0;
            if (ctb) e->ctbwd_auto |= 1 << l;
            e->ctbwd_mask = e->ctbwd_auto;
            if ((!s2_shape_ok(L) && !ctb) || e->segs.nseg + 2 > 12) continue;
            const int nw = L.cin * L.cout * L.kh * L.kw;
            L.sh_w = n;
            e->segs.seg[e->segs.nseg++] = ShardSeg{L.w_off, nw, n};
            n += (nw + 3) / 4 * 4;
            if (l + 1 == e->dec.size()) {
                L.sh_b = n;
                e->segs.seg[e->segs.nseg++] = ShardSeg{L.b_off, L.cout, n};
                n += (L.cout + 3) / 4 * 4;
            }
        }
        e->segs.n = n;
    }

    // ---- workspace carve
    int64_t top = 0;
    e->off_state = carve(top, sizeof(StepState));
    e->off_losses = carve(top, (int64_t)kLossSlots * kStatShards * sizeof(double));
    e->off_scan = carve(top, 1024 * 3 * sizeof(double));
    e->bn_stat_off.resize(bn);
    e->bn_saved_off.resize(bn);
    e->bn_channels.resize(bn);
    e->off_zero_begin = align_up(top, 256);
    auto reg_bn = [&](const ConvLayer& L) {
        if (!L.has_bn) return;
        e->bn_channels[L.bn_index] = L.cout;
        e->bn_stat_off[L.bn_index] = carve(top, (int64_t)kStatShards * L.cout * 4 * sizeof(double));
        if (L.cout > e->max_channels) e->max_channels = L.cout;
    };
    for (auto& L : e->enc) reg_bn(L);
    for (auto& L : e->dec) reg_bn(L);
    e->off_gradacc = carve(top, e->n_param * (int64_t)sizeof(double));
    e->off_sgacc = carve(top, (int64_t)kStatShards * (e->segs.n > 0 ? e->segs.n : 4) * (int64_t)sizeof(double));
    e->off_zero_end = align_up(top, 256);
    top = e->off_zero_end;
    for (auto& L : e->enc) e->bn_saved_off[L.bn_index] = carve(top, (int64_t)L.cout * 2 * sizeof(float));
    for (auto& L : e->dec)
        if (L.has_bn) e->bn_saved_off[L.bn_index] = carve(top, (int64_t)L.cout * 2 * sizeof(float));
    const int64_t mb = max_batch;
    for (auto& L : e->enc) {
        L.act_off = carve(top, mb * L.out_elems() * 4);
        L.grad_off = carve(top, mb * L.out_elems() * 4);
    }
    for (int i = 0; i < 4; i++) {
        e->fc[i].act_off = carve(top, mb * e->fc[i].nout * 4);
        e->fc[i].grad_off = carve(top, mb * e->fc[i].nout * 4);
    }
    for (auto& L : e->dec) {
        if (!L.has_bn) continue;
        L.act_off = carve(top, mb * L.out_elems() * 4);
        L.grad_off = carve(top, mb * L.out_elems() * 4);
    }
    e->off_glast = carve(top, mb * e->dec.back().out_elems() * 4);
    e->off_xbatch = carve(top, (mb * e->enc[0].in_elems() + e->enc[0].cout) * 4);   // + the layer's gamma before the update
    if (variational) {
        e->off_vz = carve(top, mb * latent_size * 4);
        e->off_vgz = carve(top, mb * latent_size * 4);
        e->off_zlast = carve(top, mb * e->dec.back().out_elems() * 4);
    }
    e->ws_need = align_up(top, 256);
    for (auto& L : e->enc)
        if (L.cin > e->max_channels) e->max_channels = L.cin;
    for (auto& L : e->dec)
        if (L.cin > e->max_channels) e->max_channels = L.cin;
    if (lds_bytes(e->max_channels, e->max_channels) > 60 * 1024) {
        delete e;
        return fail(CAE_ERR_ARG, "channel count %d exceeds the LDS constant table", e->max_channels);
    }
    *out = e;
    return CAE_OK;
}

void cae_engine_destroy(cae_engine* e) {
    if (!e) return;
    e->drop_graphs();
    (void)cae_dp_shutdown(e);
    delete e;
}

int64_t cae_param_count(const cae_engine* e) { return e ? e->n_param : 0; }
int64_t cae_buffer_count(const cae_engine* e) { return e ? e->n_buf : 0; }
int cae_tensor_count(const cae_engine* e) { return e ? (int)e->tensors.size() : 0; }
int cae_tensor_info(const cae_engine* e, int index, cae_tensor_info_t* out) {
    if (!e || !out || index < 0 || index >= (int)e->tensors.size()) return fail(CAE_ERR_ARG, "tensor index out of range");
    *out = e->tensors[index];
    return CAE_OK;
}
int64_t cae_workspace_bytes(const cae_engine* e) { return e ? e->ws_need : 0; }
int cae_loss_slots(const cae_engine*) { return kLossSlots; }

int cae_bind(cae_engine* e, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* buffers,
             void* workspace, int64_t workspace_bytes) {
    if (!e || !params || !grads || !exp_avg || !exp_avg_sq || !buffers || !workspace)
        return fail(CAE_ERR_ARG, "cae_bind: null pointer");
    if (workspace_bytes < e->ws_need)
        return fail(CAE_ERR_ARG, "workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)e->ws_need);
    if (((uintptr_t)workspace & 255) != 0) return fail(CAE_ERR_ARG, "workspace must be 256-byte aligned");
    e->drop_graphs();
    e->params = params;
    e->grads = grads;
    e->m = exp_avg;
    e->v = exp_avg_sq;
    e->bufs = buffers;
    e->ws = static_cast<char*>(workspace);
    return CAE_OK;
}

int cae_set_stream(cae_engine* e, void* hip_stream) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    if (e->stream != (hipStream_t)hip_stream) e->drop_graphs();
    e->stream = (hipStream_t)hip_stream;
    return CAE_OK;
}

int cae_set_graph_mode(cae_engine* e, int enabled) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    e->graph_mode = enabled != 0;
    if (!e->graph_mode) e->drop_graphs();
    return CAE_OK;
}

int cae_set_capture_only(cae_engine* e, int enabled) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    e->capture_only = enabled != 0;
    return CAE_OK;
}

int cae_set_kernel_mode(cae_engine* e, int specialised) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    static const int force_env = env_int("CAE_CTBWD_FORCE", -1);   // env: A/B measurements only - run-time mask (unsharded where not chosen at creation)
    const int ctb = (specialised & 2) ? 0x7fffffff : (force_env >= 0 ? force_env : e->ctbwd_auto);   // bit 1: every eligible layer
    const bool gather = (specialised & 4) != 0;
    if (e->use_s2 != ((specialised & 1) != 0) || ctb != e->ctbwd_mask || gather != e->gather_fwd) e->drop_graphs();
    e->use_s2 = (specialised & 1) != 0;
    e->ctbwd_mask = ctb;
    e->gather_fwd = gather;
    return CAE_OK;
}

int cae_set_hyper(cae_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    Hyper h{lr, beta1, beta2, eps, weight_decay};
    if (memcmp(&h, &e->hp, sizeof h) != 0) e->drop_graphs();  // hyper-parameters are baked into captured launches
    e->hp = h;
    return CAE_OK;
}

int cae_set_dataset(cae_engine* e, int which, const float* x, const float* t, int64_t n) {
    if (!e || which < 0 || which > 1 || !x || n < 1) return fail(CAE_ERR_ARG, "cae_set_dataset: bad argument");
    if (e->ds_x[which] != x || e->ds_t[which] != t) e->drop_graphs();
    e->ds_x[which] = x;
    e->ds_t[which] = t;
    e->ds_n[which] = n;
    return CAE_OK;
}

int cae_set_cursor(cae_engine* e, int64_t batch_start, int loss_slot) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (loss_slot < 0 || loss_slot >= kLossSlots) return fail(CAE_ERR_ARG, "loss slot out of range");
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, e->stream, e->state(), (long long)batch_start, loss_slot, 1, 0, 0);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_set_adam_step(cae_engine* e, int completed_steps) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    hipLaunchKernelGGL(k_set_state, dim3(1), dim3(1), 0, e->stream, e->state(), 0LL, 0, 0, completed_steps, 1);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_train_step(cae_engine* e, int which, const int32_t* perm, int batch) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    StepArgs a{which, perm, batch, batch, batch, true, true, nullptr, nullptr, true};
    return run_op(e, OP_TRAIN, a, true);
}

int cae_train_steps(cae_engine* e, int which, const int32_t* perm, int batch, int nsteps) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    if (nsteps < 1 || nsteps > 4096) return fail(CAE_ERR_ARG, "nsteps %d outside [1, 4096]", nsteps);
    StepArgs a{which, perm, batch, batch, batch, true, true, nullptr, nullptr, true};
    a.nsteps = nsteps;
    return run_op(e, OP_TRAIN, a, true);
}

int cae_eval_steps(cae_engine* e, int which, const int32_t* perm, int batch, int nsteps) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    if (nsteps < 1 || nsteps > 4096) return fail(CAE_ERR_ARG, "nsteps %d outside [1, 4096]", nsteps);
    StepArgs a{which, perm, batch, batch, batch, false, true, nullptr, nullptr, true};
    a.nsteps = nsteps;
    return run_op(e, OP_EVAL, a, true);
}

int cae_forward_backward(cae_engine* e, int which, const int32_t* perm, int batch, int global_batch) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    if (global_batch < batch) return fail(CAE_ERR_ARG, "global_batch < batch");
    StepArgs a{which, perm, batch, global_batch, batch, true, true, nullptr, nullptr, true};
    return run_op(e, OP_FWDBWD, a, true);
}

int cae_forward_backward_sync(cae_engine* e, int which, const int32_t* perm, int batch, int global_batch, int world,
                              cae_allreduce_fn fn, void* user) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    if (global_batch < batch || world < 1 || !fn) return fail(CAE_ERR_ARG, "cae_forward_backward_sync: bad argument");
    StepArgs a{which, perm, batch, global_batch, global_batch, true, true, nullptr, nullptr, true};
    a.sync_fn = fn;
    a.sync_user = user;
    a.world = world;
    return run_op(e, OP_FWDBWD, a, false);   // plain launches: the callback runs between them
}

// ---- data parallelism inside the library ---------------------------------------------------------

int cae_dp_unique_id(void* id128_host) {
    if (!id128_host) return fail(CAE_ERR_ARG, "cae_dp_unique_id: null pointer");
    if (const char* why = rccl().load()) return fail(CAE_ERR_STATE, "RCCL: %s", why);
    RcclApi::UniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id128_host, id.internal, sizeof id.internal);
    return CAE_OK;
}

int cae_dp_shutdown(cae_engine* e) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    if (e->dp_comm) {
        e->drop_graphs();
        if (e->stream) (void)hipStreamSynchronize(e->stream);
        if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
        (void)rccl().CommDestroy(e->dp_comm);
        e->dp_comm = nullptr;
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->comm_stream) (void)hipStreamDestroy(e->comm_stream);
    e->ev_fork = e->ev_join = nullptr;
    e->comm_stream = nullptr;
    e->dp_world = 0;
    return CAE_OK;
}

namespace {

// cae_dp_init's self-test: the fork / all-reduce / join pattern of a data-parallel step, captured into a hipGraph and
// replayed twice on ones: every element must read world^2 afterwards.  A capture or replay ERROR switches the engine to
// plain launches for data-parallel steps (dp_graph_ok = false); wrong VALUES fail the init.
int dp_self_test(cae_engine* e) {
    hipStream_t s = e->stream;
    const int64_t n = e->n_param, mid = e->bucket_split;
    auto body = [&]() -> int {
        HIP_TRY(hipEventRecord(e->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(e->comm_stream, e->ev_fork, 0));
        if (int rc = dp_allreduce_grads(e, mid, n, e->comm_stream)) return rc;
        HIP_TRY(hipEventRecord(e->ev_join, e->comm_stream));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));
        if (int rc = dp_allreduce_grads(e, 0, mid, s)) return rc;
        return CAE_OK;
    };
    auto fill = [&]() {
        hipLaunchKernelGGL(k_fill_f32, dim3(grid1(n)), dim3(256), 0, s, e->grads, (long long)n, 1.0f);
    };
    auto check = [&](double want, const char* what) -> int {
        float got[2] = {0.f, 0.f};
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(&got[0], e->grads, sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&got[1], e->grads + n - 1, sizeof(float), hipMemcpyDeviceToHost));
        if ((double)got[0] != want || (double)got[1] != want)
            return fail(CAE_ERR_STATE, "RCCL self-test (%s): all-reduce of ones gave %g / %g, expected %g", what, got[0], got[1], want);
        return CAE_OK;
    };
    // plain launches first: RCCL sets its channels up on the first collective of each size class, outside any capture
    fill();
    if (int rc = body()) return rc;
    if (int rc = check((double)e->dp_world, "plain")) return rc;
    static const int want_graph = env_int("CAE_DP_GRAPH", 1);   // env: 0 keeps RCCL calls out of captured graphs
    e->dp_graph_ok = want_graph != 0 && s != nullptr;
    if (!e->dp_graph_ok) return CAE_OK;
    fill();
    HIP_TRY(hipStreamSynchronize(s));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        const int rc = body();
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        ok = rc == CAE_OK && ce == hipSuccess && graph != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (ok) ok = hipGraphLaunch(exec, s) == hipSuccess && hipGraphLaunch(exec, s) == hipSuccess;
    if (ok) ok = hipStreamSynchronize(s) == hipSuccess;
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (!ok) {
        e->dp_graph_ok = false;   // data-parallel steps run as plain launches; the compute path is the same
        fprintf(stderr, "libcae_hip: RCCL collectives could not be captured into a hipGraph here; data-parallel steps use plain launches\n");
        return CAE_OK;
    }
    return check((double)e->dp_world * e->dp_world, "captured");
}

}  // namespace

int cae_dp_init(cae_engine* e, int world, int rank, const void* id128_host) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (world < 1 || rank < 0 || rank >= world || !id128_host) return fail(CAE_ERR_ARG, "cae_dp_init: bad argument");
    if (!e->stream) return fail(CAE_ERR_STATE, "cae_dp_init needs a non-default stream (cae_set_stream)");
    if (const char* why = rccl().load()) return fail(CAE_ERR_STATE, "RCCL: %s", why);
    (void)cae_dp_shutdown(e);
    RcclApi::UniqueId id;
    memcpy(id.internal, id128_host, sizeof id.internal);
    NCCL_TRY(rccl().CommInitRank(&e->dp_comm, world, id, rank));
    e->dp_world = world;
    e->dp_rank = rank;
    HIP_TRY(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // warm the size classes a step uses: the two gradient buckets (self-test) and the BatchNorm tables (fp64, on the
    // main stream), on the gradient arena as scratch - its contents mean nothing between steps
    if (int rc = dp_self_test(e)) return rc;
    // the whole arena in one collective (the structure without overlap): its size class too is met outside any capture first
    if (int rc = dp_allreduce_grads(e, 0, e->n_param, e->stream)) return rc;
    for (int c : e->bn_channels) {
        const size_t nd = (size_t)kStatShards * c * 4;
        if ((int64_t)nd * 2 > e->n_param) continue;
        NCCL_TRY(rccl().AllReduce(e->grads, e->grads, nd, RcclApi::kFloat64, RcclApi::kSum, e->dp_comm, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(e->comm_stream));
    return CAE_OK;
}

int cae_dp_info(const cae_engine* e, int* world, int* rank, int* graph_capture) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    if (world) *world = e->dp_world;
    if (rank) *rank = e->dp_rank;
    if (graph_capture) *graph_capture = e->dp_comm && e->dp_graph_ok && e->graph_mode ? 1 : 0;
    return CAE_OK;
}

int cae_dp_set_overlap(cae_engine* e, int enabled) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    if (e->dp_overlap != (enabled != 0)) e->drop_graphs();
    e->dp_overlap = enabled != 0;
    return CAE_OK;
}

int cae_dp_broadcast_state(cae_engine* e, int root, int what) {
    if (!e || !e->dp_comm) return fail(CAE_ERR_STATE, "cae_dp_init has not been called");
    if (root < 0 || root >= e->dp_world) return fail(CAE_ERR_ARG, "cae_dp_broadcast_state: bad root");
    hipStream_t s = e->stream;
    auto bc = [&](float* p, int64_t n) -> int {
        if (n > 0) NCCL_TRY(rccl().Broadcast(p, p, (size_t)n, RcclApi::kFloat32, root, e->dp_comm, s));
        return CAE_OK;
    };
    if (what & 1) if (int rc = bc(e->params, e->n_param)) return rc;
    if (what & 2) if (int rc = bc(e->bufs, e->n_buf)) return rc;
    if (what & 4) {
        if (int rc = bc(e->m, e->n_param)) return rc;
        if (int rc = bc(e->v, e->n_param)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return CAE_OK;
}

int cae_dp_train_steps(cae_engine* e, int which, const int32_t* perm, int batch, int global_batch, int sync_bn, int nsteps) {
    if (!e || !e->dp_comm) return fail(CAE_ERR_STATE, "cae_dp_init has not been called");
    if (batch == 0) {   // an empty shard (a last batch shorter than the number of ranks): collectives only
        if (!e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    } else if (int rc = check_ready(e, which, batch, true)) {
        return rc;
    }
    if (global_batch < batch || global_batch < 1) return fail(CAE_ERR_ARG, "global_batch %d < batch %d", global_batch, batch);
    if (nsteps < 1 || nsteps > 4096) return fail(CAE_ERR_ARG, "nsteps %d outside [1, 4096]", nsteps);
    StepArgs a{which, perm, batch, global_batch, sync_bn ? global_batch : batch, true, true, nullptr, nullptr, true};
    a.nsteps = nsteps;
    a.dp = true;
    a.dp_sync = sync_bn != 0;
    a.world = sync_bn ? e->dp_world : 1;   // BatchNorm parameter gradients come from GLOBAL sums under SyncBN: 1/world each
    a.cursor_inc = global_batch;           // consecutive steps of a rank are one GLOBAL batch apart in the frozen permutation
    return run_op(e, OP_DP_TRAIN, a, e->dp_graph_ok);
}

int cae_dp_train_step(cae_engine* e, int which, const int32_t* perm, int batch, int global_batch, int sync_bn) {
    return cae_dp_train_steps(e, which, perm, batch, global_batch, sync_bn, 1);
}

int cae_dp_eval_steps(cae_engine* e, int which, const int32_t* perm, int batch, int global_batch, int nsteps) {
    if (!e || !e->dp_comm) return fail(CAE_ERR_STATE, "cae_dp_init has not been called");
    if (global_batch < batch || global_batch < 1) return fail(CAE_ERR_ARG, "global_batch %d < batch %d", global_batch, batch);
    if (nsteps < 1 || nsteps > 4096) return fail(CAE_ERR_ARG, "nsteps %d outside [1, 4096]", nsteps);
    if (batch == 0) {   // empty shard: only the cursor and the loss slot move
        if (!e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
        if (e->capture_only) return CAE_OK;
        for (int i = 0; i < nsteps; i++) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, e->stream, e->state(), global_batch, 1, 0);
        HIP_TRY(hipGetLastError());
        return CAE_OK;
    }
    if (int rc = check_ready(e, which, batch, true)) return rc;
    StepArgs a{which, perm, batch, global_batch, batch, false, true, nullptr, nullptr, true};
    a.nsteps = nsteps;
    a.cursor_inc = global_batch;
    return run_op(e, OP_EVAL, a, true);
}

int cae_dp_read_losses(cae_engine* e, int first, int count, double* host_out) {
    if (!e || !e->dp_comm) return fail(CAE_ERR_STATE, "cae_dp_init has not been called");
    if (first < 0 || count < 0 || first + count > kLossSlots || !host_out) return fail(CAE_ERR_ARG, "bad loss range");
    if (count == 0) return CAE_OK;
    double* dev = e->losses() + (size_t)first * kStatShards;
    // every rank's slot holds sum(local terms) / global count: the sum over ranks is the global-batch mean
    NCCL_TRY(rccl().AllReduce(dev, dev, (size_t)count * kStatShards, RcclApi::kFloat64, RcclApi::kSum, e->dp_comm, e->stream));
    return cae_read_losses(e, first, count, host_out);
}

int cae_adam_step(cae_engine* e) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    StepArgs a{0, nullptr, 0, 0, 0, false, false, nullptr, nullptr, false};
    return run_op(e, OP_ADAM, a, true);
}

int cae_eval_step(cae_engine* e, int which, const int32_t* perm, int batch) {
    int rc = check_ready(e, which, batch, true);
    if (rc) return rc;
    StepArgs a{which, perm, batch, batch, batch, false, true, nullptr, nullptr, true};
    return run_op(e, OP_EVAL, a, true);
}

int cae_score(cae_engine* e, const float* x, int batch, float* y) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (!x || !y) return fail(CAE_ERR_ARG, "cae_score: null pointer");
    if (batch < 1 || batch > e->max_batch) return fail(CAE_ERR_ARG, "batch %d outside [1, %d]", batch, e->max_batch);
    StepArgs a{0, nullptr, batch, batch, batch, false, false, x, y, false};
    return run_op(e, OP_EVAL, a, false);
}

int cae_encode(cae_engine* e, const float* x, int batch, float* z) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (!x || !z) return fail(CAE_ERR_ARG, "cae_encode: null pointer");
    if (batch < 1 || batch > e->max_batch) return fail(CAE_ERR_ARG, "batch %d outside [1, %d]", batch, e->max_batch);
    StepArgs a{0, nullptr, batch, batch, batch, false, false, x, nullptr, false};
    a.part = 1;
    a.z_out = z;
    return run_op(e, OP_EVAL, a, false);
}

int cae_decode(cae_engine* e, const float* z, int batch, float* y) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (!z || !y) return fail(CAE_ERR_ARG, "cae_decode: null pointer");
    if (batch < 1 || batch > e->max_batch) return fail(CAE_ERR_ARG, "batch %d outside [1, %d]", batch, e->max_batch);
    StepArgs a{0, nullptr, batch, batch, batch, false, false, nullptr, y, false};
    a.part = 2;
    a.z_in = z;
    return run_op(e, OP_EVAL, a, false);
}

int cae_read_losses(cae_engine* e, int first, int count, double* host_out) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    if (first < 0 || count < 0 || first + count > kLossSlots || !host_out) return fail(CAE_ERR_ARG, "bad loss range");
    if (count == 0) return CAE_OK;
    std::vector<double> raw((size_t)count * kStatShards);
    double* dev = e->losses() + (size_t)first * kStatShards;
    HIP_TRY(hipMemcpyAsync(raw.data(), dev, raw.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemsetAsync(dev, 0, raw.size() * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < count; i++) {
        double t = 0.0;
        for (int sh = 0; sh < kStatShards; sh++) t += raw[(size_t)i * kStatShards + sh];
        host_out[i] = t;
    }
    return CAE_OK;
}

int cae_sync(cae_engine* e) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    return CAE_OK;
}

int cae_graph_count(const cae_engine* e) { return e ? (int)e->graphs.size() : 0; }

// ---- roctx ranges (SURVEY.md §5 tracing): visible in `rocprofv3 --marker-trace`, free when no profiler is attached ----------
namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    void load() {
        if (tried) return;
        tried = true;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
RoctxApi g_roctx;
}  // namespace

int cae_trace_range_push(const char* name) {
    g_roctx.load();
    if (!g_roctx.push || !name) return 0;
    g_roctx.push(name);
    return 1;
}

int cae_trace_range_pop(void) {
    if (!g_roctx.pop) return 0;
    g_roctx.pop();
    return 1;
}

int64_t cae_debug_read(cae_engine* e, const char* what, int index, void* host_out, int64_t cap) {
    if (!e || !e->ws || !what || !host_out) return fail(CAE_ERR_ARG, "cae_debug_read: bad argument");
    const void* src = nullptr;
    int64_t n = 0, esz = 4;
    const int64_t mb = e->max_batch;
    const std::string w(what);
    const int n_enc = (int)e->enc.size(), n_dec = (int)e->dec.size();
    auto conv_at = [&](int i) -> const ConvLayer* {
        if (i < 0) return nullptr;
        if (i < n_enc) return &e->enc[i];
        if (i < n_enc + n_dec - 1) return &e->dec[i - n_enc];
        return nullptr;
    };
    if (w == "act" || w == "grad") {
        const ConvLayer* L = conv_at(index);
        if (!L) return fail(CAE_ERR_ARG, "layer index out of range");
        src = e->ws + (w == "act" ? L->act_off : L->grad_off);
        n = mb * L->out_elems();
    } else if (w == "glast") {
        src = e->ws + e->off_glast;
        n = mb * e->dec.back().out_elems();
    } else if (w == "latent") {
        src = e->ws + e->fc[1].act_off;
        n = mb * e->fc[1].nout;
    } else if (w == "fc") {
        if (index < 0 || index > 3) return fail(CAE_ERR_ARG, "fc index out of range");
        src = e->ws + e->fc[index].act_off;
        n = mb * e->fc[index].nout;
    } else if (w == "fcgrad") {
        if (index < 0 || index > 3) return fail(CAE_ERR_ARG, "fc index out of range");
        src = e->ws + e->fc[index].grad_off;
        n = mb * e->fc[index].nout;
    } else if (w == "scan") {   // the loader's scratch: where tools/head_phases.py's decoder-kernel stamps land
        src = e->ws + e->off_scan;
        n = 1024 * 3;
        esz = 8;
    } else if (w == "grad_acc") {
        src = e->gradacc();
        n = e->n_param;
        esz = 8;
    } else if (w == "bn_stats") {
        if (index < 0 || index >= e->n_bn) return fail(CAE_ERR_ARG, "bn index out of range");
        src = e->bn_stats(index);
        n = (int64_t)kStatShards * e->bn_channels[index] * 4;
        esz = 8;
    } else {
        return fail(CAE_ERR_ARG, "unknown tensor '%s'", what);
    }
    if (n > cap) n = cap;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(host_out, src, (size_t)(n * esz), hipMemcpyDeviceToHost));
    return n;
}

// Measurement aid: per-node cost of a captured graph of n dependent no-op kernels on the engine's stream.
int cae_debug_launch_floor(cae_engine* e, int n, double* micros_per_kernel) {
    if (!e || !e->ws || !e->stream || n < 1 || !micros_per_kernel) return fail(CAE_ERR_ARG, "cae_debug_launch_floor: bad argument");
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, e->stream, e->state(), 0, 0, 0);
    HIP_TRY(hipStreamEndCapture(e->stream, &graph));
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) HIP_TRY(hipGraphLaunch(exec, e->stream));
    HIP_TRY(hipEventRecord(e0, e->stream));
    const int reps = 10;
    for (int i = 0; i < reps; i++) HIP_TRY(hipGraphLaunch(exec, e->stream));
    HIP_TRY(hipEventRecord(e1, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *micros_per_kernel = (double)ms * 1000.0 / (reps * (double)n);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    return CAE_OK;
}

int cae_profile_begin(cae_engine* e) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "cae_bind has not been called");
    for (auto& r : e->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    e->prof.clear();
    e->profiling = true;
    return CAE_OK;
}

int cae_profile_end(cae_engine* e, cae_profile_rec* out, int capacity) {
    if (!e) return fail(CAE_ERR_ARG, "null engine");
    e->profiling = false;
    HIP_TRY(hipStreamSynchronize(e->stream));
    int n = 0;
    for (auto& r : e->prof) {
        float ms = 0.f;
        hipError_t rc = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (rc == hipSuccess && out && n < capacity) {
            memset(&out[n], 0, sizeof out[n]);
            snprintf(out[n].name, sizeof out[n].name, "%s", r.name);
            out[n].layer = r.layer;
            out[n].micros = (double)ms * 1000.0;
            out[n].bytes = r.bytes;
            n++;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    e->prof.clear();
    return n;
}

// ---- loader -------------------------------------------------------------------------------------

int cae_scan_f32(const float* x, int64_t n, void* hip_stream, double* out3) {
    if (!x || n < 1 || !out3) return fail(CAE_ERR_ARG, "cae_scan_f32: bad argument");
    hipStream_t s = (hipStream_t)hip_stream;
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    double* part = nullptr;
    HIP_TRY(hipMalloc(&part, (size_t)blocks * 3 * sizeof(double)));
    hipLaunchKernelGGL(k_scan, dim3(blocks), dim3(256), 0, s, x, (long long)n, part);
    std::vector<double> host((size_t)blocks * 3);
    hipError_t ce = hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
    if (ce == hipSuccess) ce = hipStreamSynchronize(s);
    (void)hipFree(part);
    if (ce != hipSuccess) return fail(CAE_ERR_HIP, "cae_scan_f32: %s", hipGetErrorString(ce));
    double cnt = 0, lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < blocks; i++) {
        cnt += host[3 * i];
        lo = std::fmin(lo, host[3 * i + 1]);
        hi = std::fmax(hi, host[3 * i + 2]);
    }
    out3[0] = cnt;
    out3[1] = lo;
    out3[2] = hi;
    return CAE_OK;
}

int cae_normalise_pack_rows(const float* src, int64_t n, int c_src, int64_t hw, float* dst, int c_dst, int c_off,
                            float vmin, float range, int enable, const int32_t* dst_row_dev, void* hip_stream) {
    if (!src || !dst || n < 1 || c_src < 1 || hw < 1 || c_off < 0 || c_off + c_src > c_dst)
        return fail(CAE_ERR_ARG, "cae_normalise_pack: bad argument");
    if (n > 0x7fffffffLL) return fail(CAE_ERR_ARG, "cae_normalise_pack: more than 2^31 - 1 rows");
    const long long total = (long long)n * c_src * hw;
    int blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_normalise_pack, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, src, total, c_src,
                       (long long)hw, dst, c_dst, c_off, vmin, range, enable, (const int*)dst_row_dev);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_normalise_pack(const float* src, int64_t n, int c_src, int64_t hw, float* dst, int c_dst, int c_off,
                       float vmin, float range, int enable, void* hip_stream) {
    return cae_normalise_pack_rows(src, n, c_src, hw, dst, c_dst, c_off, vmin, range, enable, nullptr, hip_stream);
}

int cae_invert_permutation(const int32_t* perm_dev, int64_t n, int32_t* inverse_dev, void* hip_stream) {
    if (!perm_dev || !inverse_dev || n < 1 || n > 0x7fffffffLL) return fail(CAE_ERR_ARG, "cae_invert_permutation: bad argument");
    hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       (const int*)perm_dev, (long long)n, (int*)inverse_dev);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_denormalise_f64(const float* y, int64_t n, double vmin, double range, double* out, void* hip_stream) {
    if (!y || !out || n < 1) return fail(CAE_ERR_ARG, "cae_denormalise_f64: bad argument");
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_denorm_f64, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, y, (long long)n, vmin, range, out);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_metric_sums(const float* y, const float* actual, const float* mask, int64_t n_inst, int64_t inst_elems,
                    double vmin, double range, double* sums, void* hip_stream) {
    if (!y || !actual || !sums || n_inst < 1 || inst_elems < 1 || n_inst > 65535)
        return fail(CAE_ERR_ARG, "cae_metric_sums: bad argument");
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(hipMemsetAsync(sums, 0, (size_t)n_inst * 8 * sizeof(double), s));
    int chunks = (int)((inst_elems + 256 * 16 - 1) / (256 * 16));
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(k_metric_sums, dim3(chunks, (unsigned)n_inst), dim3(256), 0, s, y, actual, mask,
                       (long long)inst_elems, vmin, range, sums);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_bswap32(void* x, int64_t n, void* hip_stream) {
    if (!x || n < 0 || ((uintptr_t)x & 15)) return fail(CAE_ERR_ARG, "cae_bswap32: bad argument (16-byte aligned device pointer)");
    if (n == 0) return CAE_OK;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bswap32, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, (unsigned*)x, (long long)n);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // extern "C"

// =================================================================================================
// trunk_api.h: the engine as the convolutional trunk of the 'var' model (vae_engine.hip)
// =================================================================================================
namespace cae_internal {

int trunk_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec, int fc_size, int latent_size,
                 int max_batch, cae_engine** out) {
    return engine_create_impl(enc, n_enc, dec, n_dec, fc_size, latent_size, max_batch, true, out);
}

void trunk_set_hooks(cae_engine* e, const TrunkHooks& hooks) { e->hooks = hooks; }

float* trunk_raw_output(cae_engine* e) { return e->fptr(e->off_zlast); }
float* trunk_output_gradient(cae_engine* e) { return e->fptr(e->off_glast); }
double* trunk_output_bias_acc(cae_engine* e) { return e->gradacc() + e->dec.back().b_off; }

static StepArgs trunk_args(const float* x, int batch, bool train) {
    StepArgs a{0, nullptr, batch, batch, batch, train, false, x, nullptr, false};
    return a;
}

int trunk_forward(cae_engine* e, const float* x, int batch, bool train, bool external_loss, float* yhat) {
    if (!e || !e->ws || !e->variational) return fail(CAE_ERR_STATE, "trunk_forward: not a bound trunk engine");
    if (!x || batch < 1 || batch > e->max_batch) return fail(CAE_ERR_ARG, "trunk_forward: batch %d outside [1, %d]", batch, e->max_batch);
    StepArgs a = trunk_args(x, batch, train);
    a.external_loss = external_loss;
    a.yhat = yhat;
    if (train)   // (see launch_one: every training forward starts from a clean first BatchNorm table; the others are cleared by
                 // the step tail of the optimiser / gradient hand-over launch)
        memset(&e->c0_pending, 0, sizeof e->c0_pending);
    if (int rc = launch_forward(e, a)) return rc;
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int trunk_backward(cae_engine* e, const float* x, int batch) {
    if (!e || !e->ws || !e->variational) return fail(CAE_ERR_STATE, "trunk_backward: not a bound trunk engine");
    StepArgs a = trunk_args(x, batch, true);
    a.external_loss = true;
    if (int rc = launch_backward(e, a)) return rc;
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int trunk_adam(cae_engine* e) {
    if (!e || !e->ws) return fail(CAE_ERR_STATE, "trunk_adam: not a bound engine");
    hipLaunchKernelGGL(k_adam, dim3(grid1(e->n_param)), dim3(256), 0, e->stream, (long long)e->n_param, e->params,
                       (const float*)nullptr, e->m, e->v, e->hp, (const StepState*)e->state(), e->shard_segs(),
                       step_tail_of(e, 0, 0), 0, std::log(e->hp.beta1), std::log(e->hp.beta2), AdamConv0{});
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int trunk_gradients(cae_engine* e, float* out, double scale) {
    if (!e || !e->ws || !out) return fail(CAE_ERR_STATE, "trunk_gradients: bad argument");
    hipLaunchKernelGGL(k_acc_to_f32, dim3(grid1(e->n_param)), dim3(256), 0, e->stream, (long long)e->n_param, out, e->shard_segs(),
                       step_tail_of(e, 0, 0));
    if (scale != 1.0)
        hipLaunchKernelGGL(k_scale_f32, dim3(grid1(e->n_param)), dim3(256), 0, e->stream, out, (long long)e->n_param, (float)scale);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int trunk_adam_from(cae_engine* e, const float* grads) {
    if (!e || !e->ws || !grads) return fail(CAE_ERR_STATE, "trunk_adam_from: bad argument");
    StepTail none;
    memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_adam, dim3(grid1(e->n_param)), dim3(256), 0, e->stream, (long long)e->n_param, e->params, grads, e->m, e->v,
                       e->hp, (const StepState*)e->state(), e->shard_segs(), none, 0, std::log(e->hp.beta1), std::log(e->hp.beta2),
                       AdamConv0{});
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // namespace cae_internal
