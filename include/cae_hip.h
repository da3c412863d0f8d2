/* cae_hip.h — C ABI of libcae_hip.so, the MI355X (gfx950) implementation of the
 * cae_tools ConvAEModel hot path.
 *
 * The reference (surftemp/cae_tools) has no FFI layer: its hot path is Python on PyTorch.
 * This ABI is what a maintainer binds (ctypes stub: INTEGRATION.md) to replace these
 * reference sites, cited per entry point as file:line under src/cae_tools/:
 *
 *   models/model_sizer.py:16-67     LayerSpec                     -> cae_layer_spec
 *   models/encoder.py:36-64         Encoder.__init__/forward      -> cae_engine_create / cae_encode (module alone) / cae_score / cae_train_step
 *   models/decoder.py:24-78         Decoder.__init__/forward      -> (same engine) / cae_decode (module alone)
 *   models/conv_ae_model.py:185-203 ConvAEModel.__train_epoch     -> cae_set_cursor + cae_train_step per batch + cae_read_losses
 *   models/conv_ae_model.py:205-221 ConvAEModel.__test_epoch      -> cae_eval_step per batch + cae_read_losses
 *   models/conv_ae_model.py:223-239 ConvAEModel.score             -> cae_score
 *   models/conv_ae_model.py:303,310 MSELoss, Adam(lr, weight_decay)-> cae_set_hyper, fused in cae_train_step
 *   models/ds_dataset.py:43-67      NaN count / nanmin / nanmax   -> cae_scan_f32
 *   models/ds_dataset.py:99-113,137-147 normalise + channel concat -> cae_normalise_pack
 *   models/ds_dataset.py:131-135    denormalise_output            -> cae_denormalise_f64
 *   cli/train_cae.py:58-59          xr.open_mfdataset -> float32 batches (big-endian NetCDF-3 slabs) -> cae_bswap32
 *   models/model_metric.py:25-71    ModelMetric (base_model.py:69-100 evaluate) -> cae_metric_sums
 *
 * Conventions: plain pointers and sizes only.  Every pointer named *_dev is DEVICE memory
 * owned by the caller (the Python host allocates it as torch tensors and passes data_ptr());
 * the engine owns no device memory except its captured hipGraphs.  All work is enqueued on
 * the stream given to cae_set_stream (default: the NULL stream) and is asynchronous unless
 * stated.  Functions return 0 on success or a negative cae_status; cae_last_error() gives the
 * message for the calling thread.  One engine is used from one host thread at a time.
 * All arithmetic is fp32 (fp64 only inside reductions and in cae_denormalise_f64).
 */
#ifndef CAE_HIP_H
#define CAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cae_engine cae_engine;

enum cae_status {
    CAE_OK = 0,
    CAE_ERR_ARG = -1,      /* bad argument / unsupported geometry */
    CAE_ERR_STATE = -2,    /* call out of order (not bound, no dataset, ...) */
    CAE_ERR_HIP = -3       /* a HIP runtime call failed; message carries hipGetErrorString */
};

/* One convolutional layer: the fields of model_sizer.LayerSpec (model_sizer.py:16-23) with
 * kernel_size split into (k_h, k_w).  Encoder layers are Conv2d(k, stride, padding 0)
 * (encoder.py:43-44); decoder layers are ConvTranspose2d(k, stride, padding 0, output_padding)
 * (decoder.py:44-45). */
typedef struct {
    int32_t in_c, in_h, in_w;
    int32_t out_c, out_h, out_w;
    int32_t k_h, k_w;
    int32_t stride;
    int32_t output_padding;
} cae_layer_spec;

/* One named tensor of the model: a slice of the flat parameter arena (arena 0) or of the
 * BatchNorm running-statistics arena (arena 1).  `name` is the reference's state_dict key with
 * an "enc/" or "dec/" prefix, e.g. "dec/decoder_conv.3.weight"; shapes are PyTorch's
 * (Conv2d (Cout,Cin,kh,kw); ConvTranspose2d (Cin,Cout,kh,kw); Linear (out,in)). */
typedef struct {
    char name[96];
    int32_t arena;
    int32_t ndim;
    int64_t shape[4];
    int64_t offset;  /* in floats from the arena base */
    int64_t numel;
} cae_tensor_info_t;

const char* cae_last_error(void);
int cae_abi_version(void);

/* ---- engine lifetime and geometry ------------------------------------------------------ */

/* Build the launch plan for an encoder/decoder pair (encoder.py:36-58, decoder.py:24-50).
 * max_batch bounds every later `batch` argument and sizes the workspace. */
int cae_engine_create(const cae_layer_spec* enc, int n_enc, const cae_layer_spec* dec, int n_dec,
                      int fc_size, int latent_size, int max_batch, cae_engine** out);
void cae_engine_destroy(cae_engine* e);

int64_t cae_param_count(const cae_engine* e);   /* floats in the parameter arena */
int64_t cae_buffer_count(const cae_engine* e);  /* floats in the running-stat arena */
int cae_tensor_count(const cae_engine* e);
int cae_tensor_info(const cae_engine* e, int index, cae_tensor_info_t* out);
int64_t cae_workspace_bytes(const cae_engine* e);

/* Device memory the engine works on.  params/grads/exp_avg/exp_avg_sq: cae_param_count floats
 * each; buffers: cae_buffer_count floats; workspace: cae_workspace_bytes bytes, 256-B aligned,
 * zero-filled by the caller before the first use. */
int cae_bind(cae_engine* e, float* params_dev, float* grads_dev, float* exp_avg_dev,
             float* exp_avg_sq_dev, float* buffers_dev, void* workspace_dev, int64_t workspace_bytes);
int cae_set_stream(cae_engine* e, void* hip_stream);
/* 1: replay each step from a captured hipGraph (default); 0: plain launches */
int cae_set_graph_mode(cae_engine* e, int enabled);

/* 1: the step functions (cae_train_step(s), cae_eval_step(s), cae_dp_*_steps, cae_forward_backward, cae_adam_step) capture
 * and cache the hipGraph of the launch sequence they would run and launch NOTHING (no state changes on the device); 0
 * (default): normal operation.  Lets a host build every graph shape of an epoch loop before its first timed step - the
 * counterpart of the reference moving its modules to the device before the loop (conv_ae_model.py:312-313).  Without graph
 * mode the calls are no-ops while it is set. */
int cae_set_capture_only(cae_engine* e, int enabled);

/* 1 (default): use the specialised kernels where a layer is eligible; 0: shape-generic kernels
 * everywhere (kept as an on-device cross-check of the specialised ones).  Bit 1 (value 3): the backward pass of EVERY
 * eligible channel-rich 3x3 stride-2 decoder layer runs the LDS-staged kernel (kernels_ctbwd.h) instead of the gather
 * pair; with 1 only the layers it is faster on do (one block of 16 input channels: DESIGN.md section 4).  Bit 2 (value 5):
 * the forward pass of those layers runs the gather kernel (k_ig_fwd_s2: what layers with Cin % 4 != 0 or 2-tap kernels
 * always run) instead of the LDS-staged one - kept selectable so that the full-size parity tests cover it. */
int cae_set_kernel_mode(cae_engine* e, int specialised);

/* torch.optim.Adam(lr, betas, eps, weight_decay) with L2 decay added to the gradient
 * (conv_ae_model.py:310). */
int cae_set_hyper(cae_engine* e, double lr, double beta1, double beta2, double eps, double weight_decay);

/* ---- data ------------------------------------------------------------------------------ */

/* A device-resident, already normalised dataset: x (n, in_c, in_h, in_w), t (n, out_c, out_h,
 * out_w), both fp32 NCHW.  which: 0 = training set, 1 = test set.  t may be NULL for scoring. */
int cae_set_dataset(cae_engine* e, int which, const float* x_dev, const float* t_dev, int64_t n);

/* Device-side cursor shared by the step functions: the next batch takes samples
 * perm[batch_start .. batch_start+batch) and adds its loss to loss slot `loss_slot`; every
 * step advances batch_start by its batch and loss_slot by one.  (The reference keeps a frozen
 * list of batches, conv_ae_model.py:315-325; here the frozen object is the permutation.) */
int cae_set_cursor(cae_engine* e, int64_t batch_start, int loss_slot);
int cae_set_adam_step(cae_engine* e, int completed_steps);

/* ---- the hot path ---------------------------------------------------------------------- */

/* One iteration of __train_epoch (conv_ae_model.py:189-200): train-mode forward (batch-stat
 * BatchNorm, running stats updated), MSE loss, backward, Adam step.  perm_dev: int32 sample
 * indices into dataset `which` (NULL = identity). */
int cae_train_step(cae_engine* e, int which, const int32_t* perm_dev, int batch);

/* nsteps consecutive training / test steps of the same batch size (the cursor walks the permutation):
 * the inner loop of __train_epoch / __test_epoch over the full batches as ONE captured graph, so the
 * idle time between two graph replays is paid once per call instead of once per step. */
int cae_train_steps(cae_engine* e, int which, const int32_t* perm_dev, int batch, int nsteps);
int cae_eval_steps(cae_engine* e, int which, const int32_t* perm_dev, int batch, int nsteps);

/* Data-parallel split of the same step: forward+backward leaves the fp32 gradient of
 * sum(loss terms)/global_count in grads_dev (to be all-reduced by the caller), then
 * cae_adam_step applies Adam from grads_dev.  global_batch = batch summed over ranks. */
int cae_forward_backward(cae_engine* e, int which, const int32_t* perm_dev, int batch, int global_batch);
int cae_adam_step(cae_engine* e);

/* The same half-step with BatchNorm statistics over the GLOBAL batch (SyncBN): the reference
 * normalises over the whole batch on one device (encoder.py:45, decoder.py:47), so this is the mode
 * in which N ranks x batch/N reproduce its batch-N result.  After every launch that completes a
 * BatchNorm sum table ([8 shards][C][4] fp64: sum y, sum y^2 forward; sum g, sum g*xhat backward)
 * the library calls fn(user, table_dev, count): it must leave in the table the element-wise SUM over
 * all ranks, ordered on the engine's stream (e.g. an RCCL all-reduce enqueued on that stream), and
 * return 0.  Plain launches, 2 calls per BatchNorm layer per step. */
typedef int (*cae_allreduce_fn)(void* user, void* table_dev, int64_t count_doubles);
int cae_forward_backward_sync(cae_engine* e, int which, const int32_t* perm_dev, int batch, int global_batch,
                              int world, cae_allreduce_fn fn, void* user);

/* ---- data parallelism inside the library ---------------------------------------------------------
 * The reference trains on ONE device (device selection conv_ae_model.py:294-297, modules moved at :312-313); sharding a
 * step over the GPUs of a node is this build's addition.  One process per GPU, each with its own engine
 * holding the whole (replicated) model and data set; rank r takes rows [lo, hi) of every frozen global batch.  The library
 * owns an RCCL communicator (bound at run time from the librccl.so already resident in the process, else the library
 * path's) and a second HIP stream.  A data-parallel step = forward + backward of the local shard with the loss scaled by
 * 1/global count; the fp64 gradient accumulators are narrowed to fp32 and all-reduced (SUM) in two buckets in the order
 * backward completes them - Linear 3 + decoder convolutions on the second stream while Linear 2..0 and the encoder
 * backward still run, then the rest - and Adam runs from the reduced gradients: weights stay bit-identical on all ranks.
 * The whole step, collectives included, is replayed from one hipGraph when cae_dp_init's self-test could capture RCCL
 * calls (else plain launches: same kernels, same results).
 *
 * cae_dp_unique_id: rank 0 obtains the 128-byte rendezvous id; the host hands the same bytes to every rank (any
 *   channel: torch.distributed broadcast, MPI, a file).
 * cae_dp_init: collective over all ranks; the calling thread's current HIP device is this rank's GPU.  Runs a self-test
 *   (all-reduce of ones, plain and captured).  cae_bind and cae_set_stream (non-NULL stream) must have been called.
 * cae_dp_broadcast_state: what = 1 parameters | 2 BatchNorm running statistics | 4 Adam moments, from rank `root` (blocking).
 * cae_dp_train_step(s): batch = this rank's shard size (0 allowed: the rank only takes part in the collectives),
 *   global_batch = sum over ranks; the device cursor advances by global_batch per step, so the host points it at
 *   global_start + lo once and nsteps consecutive global batches run from one graph.  sync_bn = 1: BatchNorm statistics
 *   over the GLOBAL batch - every sum table ([8][C][4] fp64) is all-reduced in-stream right after the launch that completes
 *   it (2 per BatchNorm layer per step), which makes N ranks x batch/N reproduce the reference's single-device batch-N
 *   step (encoder.py:45, decoder.py:47 normalise over the whole batch); sync_bn = 0: per-rank statistics (what torch's
 *   DistributedDataParallel does without SyncBatchNorm), for throughput.
 * cae_dp_set_overlap: 1 (default) = the first bucket leaves on the second stream as described; 0 = ONE all-reduce of the
 *   whole gradient arena on the main stream after backward (no cross-stream edges, one collective latency per step).  Same arithmetic, same results; which is faster depends on the
 *   all-reduce latency against the cost of a fork/join between two hardware queues (measured: DESIGN.md §7), so the host
 *   times both on the live communicator and keeps the faster (HipEngine.dp_calibrate).  Every rank must use the same setting.
 * cae_dp_eval_steps: __test_epoch on this rank's shard of each global test batch (loss scaled by 1/global count).
 * cae_dp_read_losses: cae_read_losses of the SUM over ranks = the global-batch means (collective, blocking). */
int cae_dp_unique_id(void* id128_host);
int cae_dp_init(cae_engine* e, int world, int rank, const void* id128_host);
int cae_dp_shutdown(cae_engine* e);
int cae_dp_info(const cae_engine* e, int* world, int* rank, int* graph_capture);
int cae_dp_set_overlap(cae_engine* e, int enabled);
int cae_dp_broadcast_state(cae_engine* e, int root, int what);
int cae_dp_train_step(cae_engine* e, int which, const int32_t* perm_dev, int batch, int global_batch, int sync_bn);
int cae_dp_train_steps(cae_engine* e, int which, const int32_t* perm_dev, int batch, int global_batch, int sync_bn, int nsteps);
int cae_dp_eval_steps(cae_engine* e, int which, const int32_t* perm_dev, int batch, int global_batch, int nsteps);
int cae_dp_read_losses(cae_engine* e, int first, int count, double* host_out);

/* One iteration of __test_epoch (conv_ae_model.py:205-221): eval-mode forward + MSE into the
 * loss slot; nothing else is written. */
int cae_eval_step(cae_engine* e, int which, const int32_t* perm_dev, int batch);

/* score() for one batch (conv_ae_model.py:223-239): eval-mode forward of x_dev (batch, in_c,
 * in_h, in_w) into y_dev (batch, out_c, out_h, out_w). */
int cae_score(cae_engine* e, const float* x_dev, int batch, float* y_dev);

/* Module-level forward of the reference's two torch modules, eval mode (running BatchNorm statistics), one batch:
 *   Encoder.forward(x) -> z   (encoder.py:60-64):  x_dev (batch, in_c, in_h, in_w) -> z_dev (batch, latent)
 *   Decoder.forward(z) -> y   (decoder.py:73-78):  z_dev (batch, latent) -> y_dev (batch, out_c, out_h, out_w), sigmoid applied
 * cae_decode(cae_encode(x)) equals cae_score(x).  Asynchronous on the engine's stream. */
int cae_encode(cae_engine* e, const float* x_dev, int batch, float* z_dev);
int cae_decode(cae_engine* e, const float* z_dev, int batch, float* y_dev);

/* Blocking: wait for the stream, copy `count` per-step mean losses starting at slot `first`
 * to host memory, and zero those slots. */
int cae_read_losses(cae_engine* e, int first, int count, double* host_out);
int cae_loss_slots(const cae_engine* e);
int cae_sync(cae_engine* e);
/* Test hook: captured hipGraphs currently cached by the engine (one per distinct launch sequence). */
int cae_graph_count(const cae_engine* e);

/* Test hook (blocking): copy an internal tensor of the LAST train-mode step to host.
 * what: "act" raw conv output of layer `index` (encoder layers first, then decoder layers except
 * the last), "latent", "fc" (decoder FC output), "grad_acc" (fp64 gradient accumulator,
 * `index` ignored, count doubles).  Returns the number of elements copied or a negative status. */
int64_t cae_debug_read(cae_engine* e, const char* what, int index, void* host_out, int64_t capacity_elems);

/* ---- measurement ------------------------------------------------------------------------- */

/* Per-launch timing for bench.py's roofline figure.  Between begin and end every kernel launch
 * of the step functions is bracketed by a HIP event pair on the engine's stream (plain launches,
 * no graph replay).  cae_profile_end blocks, fills up to `capacity` records in launch order and
 * returns their number.  bytes = algorithmic bytes of the launch (each operand tensor read once,
 * each result written once). */
typedef struct {
    char name[48];
    int32_t layer;
    int32_t reserved;
    double micros;
    double bytes;
} cae_profile_rec;
int cae_profile_begin(cae_engine* e);
/* measurement aid: cost per node of a captured graph of n dependent no-op kernels (blocking) */
int cae_debug_launch_floor(cae_engine* e, int n, double* micros_per_kernel);
int cae_profile_end(cae_engine* e, cae_profile_rec* out, int capacity);

/* Tracing (the reference only prints elapsed times: conv_ae_model.py:301,336-341): named host ranges through roctx, bound at
 * run time (librocprofiler-sdk-roctx.so) - they show up under `rocprofv3 --marker-trace` around the kernels of a pass and
 * cost nothing without a profiler.  Return 1 when the range was recorded, 0 when no roctx library is present. */
int cae_trace_range_push(const char* name);
int cae_trace_range_pop(void);

/* ---- loader kernels (stateless) ---------------------------------------------------------- */

/* ds_dataset.py:43-46,53-58: out3 = {NaN count, nanmin, nanmax} of n floats (blocking). */
int cae_scan_f32(const float* x_dev, int64_t n, void* hip_stream, double* out3_host);

/* ds_dataset.py:99-113,137-147: dst[i, c_off + c, :] = (src[i, c, :] - vmin) / range in fp32
 * (0 when range == 0; plain copy when enable == 0).  src (n, c_src, hw), dst (n, c_dst, hw). */
int cae_normalise_pack(const float* src_dev, int64_t n, int c_src, int64_t hw, float* dst_dev,
                       int c_dst, int c_off, float vmin, float range, int enable, void* hip_stream);

/* conv_ae_model.py:291-292,315-325: DataLoader(shuffle=True) + default collate stack the shuffled samples into batches
 * ONCE and the list is reused every epoch.  Here the stacking happens in the normalisation pass itself: sample i is
 * written to row dst_row_dev[i] of dst (the inverse of the frozen sample order, so that a batch is a contiguous run of
 * rows; NULL = identity, i.e. cae_normalise_pack).  dst_row_dev: n int32 on the device, a permutation of 0..n-1.
 * cae_invert_permutation builds that table from the frozen order (inverse[perm[i]] = i). */
int cae_normalise_pack_rows(const float* src_dev, int64_t n, int c_src, int64_t hw, float* dst_dev, int c_dst, int c_off,
                            float vmin, float range, int enable, const int32_t* dst_row_dev, void* hip_stream);
int cae_invert_permutation(const int32_t* perm_dev, int64_t n, int32_t* inverse_dev, void* hip_stream);

/* ds_dataset.py:131-135 on base_model.py:123's float64 array: out = vmin + ((double)y * range). */
int cae_denormalise_f64(const float* y_dev, int64_t n, double vmin, double range, double* out_dev,
                        void* hip_stream);

/* cli/train_cae.py:58-59 / ds_dataset.py:137-147: a NetCDF-3 variable is a contiguous big-endian slab; the host
 * copies it to the device as raw bytes and this swaps n 32-bit words in place (x_dev 16-byte aligned). */
int cae_bswap32(void* x_dev, int64_t n, void* hip_stream);

/* model_metric.py:25-71 as used by base_model.py:69-100: per instance i of n_inst (each inst_elems
 * floats of y, actual and — unless NULL — mask), over the pixels with mask != 0:
 *   sums[i] = {n, S(a'), S(e'), S(a'^2), S(e'^2), S(a'e'), S|a-e|, S(a-e)^2},  e = vmin + (double)y*range,
 *   a' = a - vmin, e' = e - vmin  (fp64; sums_dev (n_inst, 8) is cleared by the call).  The host pools
 *   mse/rmse/mae over all instances and averages the per-instance Pearson r.  n_inst <= 65535 per call. */
int cae_metric_sums(const float* y_dev, const float* actual_dev, const float* mask_dev, int64_t n_inst,
                    int64_t inst_elems, double vmin, double range, double* sums_dev, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* CAE_HIP_H */
