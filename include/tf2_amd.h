/*
 * tf2_amd.h -- C ABI of the MI355X-native drop-in for TF2's Runtime_Engine/cnn path.
 *
 * The reference has no plugin/FFI API for this path: it sits behind three C++ classes
 * and a handful of free functions compiled against one network header
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it
 * replaces (paths relative to /root/reference/Runtime_Engine/cnn).  Plain pointers and
 * sizes only; no torch / HIP types in signatures (a stream is passed as void*).
 *
 * Conventions
 *   - every function returns 0 (TF2_OK) or a negative tf2_status; it never exits the
 *     process (the reference's checkError() -> exit(), common/src/AOCLUtils/opencl.cpp:226-250,
 *     is deliberately not reproduced); tf2_last_error() gives the message (thread-local).
 *   - "q" arrays hold the RUNTIME values of quantization.cpp:46, i.e. the NEGATED Q-file ints.
 *   - device pointers are raw HIP device addresses owned by the caller (PyTorch
 *     allocations); the library allocates no device memory in run calls.
 */
#ifndef TF2_AMD_H_
#define TF2_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int tf2_status;
enum {
  TF2_OK = 0,
  TF2_ERR_ARG = -1,        /* bad argument / inconsistent tables                      */
  TF2_ERR_STATE = -2,      /* call order (e.g. run before load_model)                 */
  TF2_ERR_SIZE = -3,       /* buffer too small / model stream length mismatch         */
  TF2_ERR_HIP = -4,        /* HIP runtime error (message carries hipGetErrorString)   */
  TF2_ERR_UNSUPPORTED = -5,/* layer shape not supported by any kernel                 */
  TF2_ERR_GROUP = -6       /* a group launch gave up a meeting (tf2_net_poll_error)   */
};

/* One row of the network program = one row of the k* tables of <net>.h
 * (host/inc/resnet50.h:119-1372), with the graph edges made explicit
 * (tf2_amd.config.build_plan).                                                        */
typedef struct tf2_layer_desc {
  int32_t src;          /* producer of the input: -1 image, >=0 layer, <=-2 concat -(id+2) (kInputLayer) */
  int32_t q_in_row;     /* Q-table row of the input channels (kInputLayer)               */
  int32_t C, H, W;      /* kInputChannels, kInputHeight, kInputWidth                     */
  int32_t N, k, stride; /* kOutputChannels, kFilterSize, kConvStride                     */
  int32_t pad_h, pad_w, dil; /* kPadHeight, kPadWidth, (dilation, 1 in the reference)    */
  int32_t OH, OW;       /* conv output (after conv stride)                               */
  int32_t bias_en, bn_en, relu, ipool; /* kBiasEnable, kBnEnable, kReluEnable, kIpoolEnable (1 = independent pooling row;
                           2 = this build's independent L2Norm row for SSD's conv4_3 branch: N float weights in
                           the model stream, its own Q row, no filter) */
  int32_t pool_en, pool_S, pool_st, pool_pad; /* kPoolEnable, kPoolWindow, kPoolStride2, kPoolPad */
  int32_t PH, PW;       /* kPoolOutputHeight/Width                                       */
  int32_t add_src, add_relu; /* kAdditionEnable (+ DDR page plan), kAdditionReluEnable   */
  int32_t endpool, endpool_mult; /* kEndPoolEnable; 669 for 7x7 (full_size_pool.cl:118)  */
  int32_t concat, n_start; /* kBranchTail/kConcatLayer, kNStart                          */
  int32_t model_C, model_k; /* filter dims in the model file (layer 0: INPUT_IMAGE_C, FIRST_FILTER_SIZE) */
} tf2_layer_desc;

typedef struct tf2_net_desc {
  int32_t n_layers;       /* NUM_LAYER                                                   */
  int32_t n_conv;         /* NUM_CONVOLUTIONS                                            */
  int32_t n_q_rows;       /* NUM_Q_LAYERS                                                */
  int32_t max_out_channel;/* MAX_OUT_CHANNEL                                             */
  int32_t image_c, image_h, image_w; /* INPUT_IMAGE_C/H/W                                */
  int32_t conv1_rewrite;  /* 1: layer 0 runs as 27-ch 3x3 on the 114x114 space-to-depth
                             image (model_loader.cpp:244-257, input_loader.cpp:98-116)   */
  int32_t n_concat;       /* number of concat tensors                                    */
} tf2_net_desc;

typedef struct tf2_net tf2_net;

/* ---- misc ------------------------------------------------------------------------ */
const char* tf2_last_error(void);
int tf2_abi_version(void);
/* 1 if the library was built with the gfx950 HIP kernels (always, for the shipped .so). */
int tf2_has_device_code(void);
/* 0 for the shipped library.  The same sources also build two TOOL libraries that must never serve a product: 1 = the timing-probe
 * build (-DTF2_PROBES: can leave work out of a step), 2 = the DMA-check build (-DTF2_CHECK_DMA: stamps and checks sentinels around every
 * LDS-DMA).  tf2_amd/_lib.py refuses to load a library whose kind is not 0 unless a tool asks for it (TF2_AMD_TOOL_LIB=1). */
int tf2_build_kind(void);

/* ---- load-time numerics (host CPU; replace model_loader.cpp / quantization.cpp) ----- */
/* Get_real(float, char): model_loader.cpp:98-126 */
uint8_t tf2_get_real(float w, int8_t expand);
/* Quantization(q, input, file): quantization.cpp:25-55.  Takes the Q-file TEXT. Fills
 * q[n_q_rows][max_out_channel] (caller zero-fills) from the net's layer table.        */
tf2_status tf2_quantization(const tf2_net* net, const char* q_text, size_t q_text_len,
                            int8_t* q, size_t q_capacity, int32_t* n_values_read);

/* ---- network handle (replaces NetWork::Init / InitNetwork / InitBuffer, network.cpp:22-150) */
tf2_status tf2_net_create(const tf2_net_desc* nd, const tf2_layer_desc* layers, tf2_net** out);
void tf2_net_destroy(tf2_net* net);
/* runtime q table [n_q_rows][max_out_channel] as produced by tf2_quantization */
tf2_status tf2_net_set_q(tf2_net* net, const int8_t* q, size_t n_bytes);
/* LoadModel(file, filter_raw, bias_bn, q): model_loader.cpp:129-258.  `model` is the
 * float32 stream of fpgamodel.bin / param.bin (already in memory).                     */
tf2_status tf2_net_load_model(tf2_net* net, const float* model, size_t n_floats);
/* The 4-bit packed model file of TransForm_Kit (Compression/compress_net/4bit_data_format.txt:1-44: per tensor
 * {int8 min_exp, int8 dtype, int16 N,C,H,W}, then 4-bit power-of-two codes in 16-bit words or float32).  The
 * reference documents the format and ships no reader; these are the canonical ones.  _decode: float32 LoadModel
 * stream into `floats` (capacity in floats; pass NULL/0 to get the count only).  _load_model_4bit walks the file in place:
 * every 4-bit code goes straight to the reference's byte code (Get_real of the code's value through a 16-entry table per
 * (tensor, expand)) -- no float32 copy of the weights is made; bit-identical to decode + LoadModel.                       */
tf2_status tf2_model4bit_decode(const void* bytes, size_t n_bytes, float* floats, size_t capacity, size_t* n_floats);
tf2_status tf2_net_load_model_4bit(tf2_net* net, const void* bytes, size_t n_bytes);
/* Introspection for per-function parity tests: byte codes [N][C][k][k] of a layer
 * (layer 0 after the conv1 rewrite: [N][27][3][3]) and its BiasBnParam (types.h:39-43). */
tf2_status tf2_net_get_codes(const tf2_net* net, int layer, uint8_t* codes, size_t capacity, size_t* n_bytes);
tf2_status tf2_net_get_bias_bn(const tf2_net* net, int layer, int32_t* bias, int32_t* alpha, int32_t* beta, size_t capacity);

/* ---- packed device image of the weights (what the one-time RCCL broadcast moves) ---- */
/* Options: conv kernel selection per layer class.  mode: 0 = auto (MFMA where the layer
 * qualifies, shift-accumulate VALU kernel otherwise), 1 = force the shift-accumulate
 * kernel for k>1 convs (MFMA for 1x1 only, the north-star split), 2 = shift kernel
 * everywhere.                                                                           */
tf2_status tf2_net_pack(tf2_net* net, int mode);
size_t tf2_net_packed_size(const tf2_net* net);
tf2_status tf2_net_packed_copy(const tf2_net* net, void* host_dst, size_t capacity);
/* Adopt a packed image received from another rank (host copy; same tables required).    */
tf2_status tf2_net_packed_adopt(tf2_net* net, const void* host_src, size_t n_bytes);
/* Tell the net where the packed image lives on the device (caller copied/broadcast it). */
tf2_status tf2_net_bind_device(tf2_net* net, const void* packed_dev, size_t n_bytes);

/* ---- running (replaces Runner::Run, runner.cpp:54-198, and the OpenCL device pipeline) */
/* keep_all != 0: every layer output gets its own buffer (per-layer parity tests).  Ask AFTER tf2_net_pack /
 * tf2_net_packed_adopt: the plan depends on which layer pairs the packed image runs as one launch (their inputs
 * stay live longer); a buffer sized before packing may be refused by tf2_net_run with TF2_ERR_SIZE.               */
size_t tf2_net_workspace_size(tf2_net* net, int batch, int keep_all);
/* Bytes of the dense int8 output of a run: [batch][H_last * W_last][N_last] (NHWC; H_last = W_last = 1 for the
 * classification networks, i.e. [batch][N_last] -- the buffer Runner::Run reads back, runner.cpp:176-186).      */
size_t tf2_net_logits_size(const tf2_net* net, int batch);
/* Options: ONE environment string, TF2_AMD_OPTS="name=value,name=value,flag" (INTEGRATION.md section 5 lists the product options:
 * alt_conc, bgroup, bband, c3, fc, fc4, share), parsed into an immutable snapshot at tf2_net_create and here -- nothing else in the
 * library reads the environment.  Unknown names, and test-only options (forced kernels, disabled proofs, thresholds: the test-suite's
 * and the A/B tools') without TF2_AMD_TEST=1, make both calls return TF2_ERR_ARG.  The snapshot is process-wide: packing (pack-time
 * options) and planning read the one taken last.  Drops the prepared launch plans.                              */
tf2_status tf2_net_reload_options(tf2_net* net);
/* images_dev: float32 [batch][image_c][image_h][image_w] on the device (the preprocessed
 * CHW floats LoadInputImage reads, input_loader.cpp:76-96).  Quantises with 2^Q0
 * (runner.cpp:158-164), runs every layer, writes the int8 output (tf2_net_logits_size bytes,
 * [batch][N_last] for a 1x1 final map) to logits_dev.  All kernel argument blocks of a step are
 * prepared once per (batch, workspace address) and reused by later calls.                       */
tf2_status tf2_net_run(tf2_net* net, const float* images_dev, int batch, void* workspace_dev,
                       size_t workspace_bytes, int8_t* logits_dev, void* hip_stream);
/* Same, from already-quantised int8 images [batch][image_c][image_h][image_w].          */
tf2_status tf2_net_run_q(tf2_net* net, const int8_t* images_q_dev, int batch, void* workspace_dev,
                         size_t workspace_bytes, int8_t* logits_dev, void* hip_stream);
/* The same step with per-call options (the reference has one frame loop on one command queue, runner.cpp:140-175; a
 * GPU server keeps several independent batches in flight on several streams):
 *  - images_are_q   1: int8 images (tf2_net_run_q), 0: float32.
 *  - concurrency    how the launch plan picks its tile shapes: 0 = this batch runs alone on the GPU, 1 = other batches are
 *                   in flight on other streams (their kernels fill the chip: wider tiles, unfused pairs), -1 = decide from
 *                   the streams of the last eight calls on this handle (what tf2_net_run does).  Results are bit-identical
 *                   whichever is chosen.
 *  - mark_event / mark_after_layer   a hipEvent_t recorded on hip_stream once every launch of layers 0..mark_after_layer
 *                   has been enqueued (ignored when mark_event is NULL).  A caller that pipelines batches lets the next
 *                   batch's stream wait for it (hipStreamWaitEvent), so that batch k+1 enters the chip-filling first
 *                   stage when batch k has left it instead of competing with it; bench.py does.
 * Threading: a handle may be used from several host threads, one stream AND one workspace per thread; the calls serialise
 * on an internal mutex only while they look up / build the launch plan, the enqueue of the step's launches runs side by side
 * (calls that share a workspace serialise for the whole enqueue; kernel execution is asynchronous as always).  Everything
 * else on a handle (create / set_q / load / pack / bind / reload_options / profile / destroy) must not run concurrently
 * with a run on the same handle; workspace_size / read_layer / describe_* / run_stats take the handle's mutex and may.  tf2_last_error
 * is per thread.                                                          */
typedef struct tf2_run_opts {
  uint32_t size;              /* sizeof(tf2_run_opts), for forward compatibility */
  int32_t images_are_q;
  int32_t concurrency;
  int32_t mark_after_layer;
  void* mark_event;
} tf2_run_opts;
tf2_status tf2_net_run_ex(tf2_net* net, const void* images_dev, int batch, void* workspace_dev, size_t workspace_bytes,
                          int8_t* logits_dev, void* hip_stream, const tf2_run_opts* opts);
/* Group launches (conv_bgroup.hip: the one-batch-at-a-time plan keeps the eight blocks of an image resident together, one block per
 * CU, and lets them meet inside the kernel) have PRECONDITIONS the library checks where it can and the caller owns where it cannot:
 *  - at least 64 CUs for the stream: checked per call of the one-batch-at-a-time path (hipExtStreamGetCUMask: the check counts MASK
 *    BITS, not granted CUs -- gfx950 ignores interleaved masks, so a stream masked to 32 interleaved bits is treated as small although
 *    it runs on the whole chip: safe, it merely loses the group launches); a stream masked to fewer runs the step without them.
 *  - one batch at a time: selected only for concurrency == 0, stated or inferred.  A caller that STATES concurrency = 0 on more than
 *    four streams of one device at once (or drives more than four handles that way) breaks the contract: each XCD has 32
 *    one-block slots, four concurrent group kernels always leave room for one complete group, a fifth need not -- a meeting
 *    that does not complete within 2^24 polls is given up and REPORTED (tf2_net_poll_error below: the step's logits are garbage, the
 *    context, the stream and every later step are intact; rounds 3-5 trapped here).  With
 *    concurrency = -1 (tf2_net_run) the library sees the several streams in its call history and never selects them there.
 *  - the workspace bytes behind the tensors (step counter, flags) are the library's; they are re-initialised by every step.
 * The batches-in-flight plan (concurrency = 1) has none of this: its fused launches (conv_bband.hip) exchange nothing between blocks.
 * tf2_net_run_stats: out4 = {steps run, steps whose plan had group launches, steps run with the in-flight plan, steps on a stream
 * of fewer than 64 CUs} since tf2_net_create -- what a test or a server asserts its deployment against.                          */
tf2_status tf2_net_run_stats(tf2_net* net, int64_t* out4);
/* Group launches that cannot complete REPORT, they never trap or hang the context (round 6): a block that has polled a meeting of its
 * image's eight members `bgroup_polls` times (1 << 24: seconds) writes a report into the workspace's error word and leaves the kernel, and
 * so do the other members of that image; every other image of the launch and every later launch run on (the affected step's logits are
 * garbage).  tf2_net_poll_error synchronises `stream`, reads and clears the error word of the workspace that was used for `batch` images and
 * returns TF2_ERR_GROUP (tf2_last_error says which meeting) or TF2_OK; the report is sticky across later steps until it is polled.  A
 * server polls where it synchronises anyway (when it reads a batch's logits).  Steps of the in-flight plan (concurrency = 1) have no group
 * launches and nothing to report.  The reference has no counterpart (its kernels are statically scheduled, sequencer.cl). */
tf2_status tf2_net_poll_error(tf2_net* net, int batch, void* workspace, size_t workspace_bytes, void* stream);
/* Introspection: the kernel launches one step of `batch` images consists of, in issue order, as the library's own launch
 * plan selects them (concurrency as in tf2_run_opts: 0 or 1).  Needs a packed image, no device.  rows[i].layer = the table
 * row the launch belongs to (-1: input preparation; a fused launch carries its first row).  Returns the number of launches
 * in *n (also when capacity is too small: TF2_ERR_SIZE).  tools/pmc_summary.py attaches rocprofv3 rows to layers with it. */
typedef struct tf2_launch_info {
  int32_t layer, grid, block, lds_bytes, vgprs;
  char kernel[96];
} tf2_launch_info;
tf2_status tf2_net_describe_launches(tf2_net* net, int batch, int concurrency, tf2_launch_info* rows, int capacity, int* n);
/* The liveness-planned workspace of `batch` images (no device needed): tensor t occupies [offset, offset + bytes) from table row
 * first_row (-1: the input) to last_row (n_layers: the network output); row_capacity >= n_layers entries of `rows` say which
 * tensors a row reads (in, res: -1 = none) and writes (conv: the convolution's own result; out: after pool / average).  Rows that
 * share a launch (fused pairs, group launches and their chains) keep everything they touch alive for the whole launch: two
 * tensors of such rows never overlap (tests/test_host_abi.py).  The reference has no counterpart (its feature maps live in
 * fixed on-chip buffers, feature_writer.cl:88-151).                                                                          */
typedef struct tf2_tensor_info { int64_t offset, bytes; int32_t first_row, last_row; } tf2_tensor_info;
typedef struct tf2_row_tensors { int32_t in_tensor, out_tensor, conv_tensor, res_tensor; } tf2_row_tensors;
tf2_status tf2_net_describe_workspace(tf2_net* net, int batch, int keep_all, tf2_tensor_info* tensors, int tensor_capacity, int* n_tensors,
                                      tf2_row_tensors* rows, int row_capacity);
/* After a keep_all run: copy layer `layer`'s output to the host as dense NCHW int8
 * [batch][N][PH][PW] (or [batch][N] after an end pool).  layer == -1: the quantised,
 * transformed network input [batch][C0][H0][W0].  Synchronises the stream.
 * This is the ONLY way to look at an intermediate map: inside the workspace a tensor may be
 * stored in an engine-private form (x alone instead of [x | xneg] for the image, the im2col
 * image of a 3x3 first layer on 3 channels, 2x - 128 on the "doubled" channels of internal
 * post-ReLU tensors); read_layer hands back the reference's int8 values.                   */
tf2_status tf2_net_read_layer(tf2_net* net, int layer, int batch, const void* workspace_dev,
                              int8_t* host_dst, size_t capacity, void* hip_stream);
/* Per-kernel timing hook for bench.py: records HIP events around every conv launch of
 * the next tf2_net_run calls on `hip_stream`; tf2_net_profile_read returns, per layer,
 * the accumulated milliseconds and launch count since profiling was enabled.  enable == 2
 * records ONE event pair around the whole layer loop instead (tf2_net_profile_loop_read): every
 * per-layer pair adds its own record handling, the loop pair does not, so callers rescale the
 * per-layer sum to the loop time.                                                         */
tf2_status tf2_net_profile(tf2_net* net, int enable);
tf2_status tf2_net_profile_loop_read(tf2_net* net, float* ms_total, int32_t* runs);
tf2_status tf2_net_profile_read(tf2_net* net, float* ms_per_layer, int32_t* launches_per_layer,
                                int32_t* kernel_kind_per_layer, int capacity);

/* ---- Evaluation (network_helper.cpp:143-207): top-k with the reference's tie rule ---- */
tf2_status tf2_topk(const int8_t* logits, const int8_t* q_last_row, int n, int k,
                    int32_t* labels, float* features);

#ifdef __cplusplus
}
#endif
#endif /* TF2_AMD_H_ */
