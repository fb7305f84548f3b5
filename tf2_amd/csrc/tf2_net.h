// tf2_net.h -- the network handle behind the C ABI (host side).
#pragma once
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "tf2_internal.h"

namespace tf2 {

struct LaunchRecord;

struct TensorPlan {
  int H = 0, W = 0, C = 0, Cp = 0;
  size_t bytes = 0;        // for the planned batch
  size_t offset = 0;
  int last_use = -1;       // last layer index reading it (n_layers = network output)
  int first_use = -1;      // the row from which the tensor holds memory (-1: the input); earlier than its producer when rows share a launch
};

struct LayerExec {
  int in_tensor = -1;      // tensor ids into WorkPlan::tensors
  int out_tensor = -1;     // final output of the layer (own tensor or concat tensor)
  int conv_tensor = -1;    // where the conv kernel writes (== out_tensor unless pool/endpool)
  int res_tensor = -1;
  int res_off = 0;
  int out_off = 0;         // channel offset inside out_tensor (concat slice)
};

struct WorkPlan {
  int batch = 0;
  bool keep_all = false;
  std::vector<TensorPlan> tensors;
  std::vector<LayerExec> exec;
  int input_tensor = -1;
  int final_tensor = -1;
  size_t scratch_off = 0, scratch_bytes = 0;   // partial sums of conv_fc launches (behind the control area; never zeroed)
  size_t ctrl_off = 0, ctrl_bytes = 0;   // group counters of conv_bgroup launches (two words per image and launch)
  size_t ks_ctr_off = 0, ks_ctr_bytes = 0;       // ticket words of the split-K-over-blocks launches (the tail of the control area: cleared by the step's first kernel)
  size_t ks_part_off = 0, ks_part_bytes = 0;     // their partial tiles (part of the scratch area, one launch at a time)
  size_t total_bytes = 0;
};

// One prepared kernel launch of a step (net.hip launch_plan): argument block + kernel selection.
struct Launch {
  enum Kind { PREP, CONV, POOL, AVG, L2N } kind = CONV;
  enum Sel { SEL_MFMA2, SEL_SK, SEL_PW, SEL_PWK, SEL_PWKPAIR, SEL_SHIFT, SEL_BNECK, SEL_STEM, SEL_PAIR, SEL_SKPAIR, SEL_BGROUP, SEL_BGROUPF, SEL_BFIRST, SEL_BBAND, SEL_C3, SEL_FC, SEL_FIRST, SEL_FIRE } sel = SEL_MFMA2;
  int layer = -1;
  int TM = 0, signed_in = 0, mul24 = 0, shape = 0;
  int TM2 = 0;               // SEL_PWKPAIR: the second layer's tile height
  int avg_fused = 0;         // the conv launch computes the layer's global average itself (no AVG step follows)
  ConvArgs conv{};
  ConvArgs conv2{};          // SEL_PAIR: the second (independent) layer of the launch, table row layer + 1
  ConvArgs conv_direct{};    // the last layer writing the dense logits itself (y patched per call)
  BneckArgs bneck{};
  FcArgs fc{};               // SEL_FC: a whole-window layer at batch <= 32 as a weight stream over the whole chip (conv_fc.hip)
  C3Args c3{};               // SEL_C3: a 3x3 / 1 / pad 1 layer from an LDS-resident halo tile (conv_c3.hip)
  BBandArgs bband{};         // SEL_BBAND: rows layer .. layer + 2 (an identity bottleneck) in one launch, no exchange between blocks
  BGroupArgs bgroup{};       // SEL_BGROUP: rows layer .. layer + 2 (an identity bottleneck) in one launch
  int bg_hw = 0, bg_c = 0, bg_m = 0;
  const BGroupArgs& bg_chain_last() const { return bg_chain.empty() ? bgroup : bg_chain.back(); }
  std::vector<BGroupArgs> bg_chain;   // SEL_BGROUP of the 14 x 14 maps: this and the following identity bottlenecks in ONE launch (bgroup first)
  StemArgs stem{};
  PoolArgs pool{};
  AvgArgs avg{};
  PrepArgs prep{};
  FireArgs fire{};           // SEL_FIRE: rows layer (squeeze), layer + 1 and layer + 2 (the merged expands) in one launch (conv_fire.hip)
  FirstArgs first{};         // kind PREP, SEL_FIRST: input preparation + table row 0 in one launch (conv_first_kernel); prep = its input side
  L2NormArgs l2n{};
};

struct LaunchPlan {
  int batch = 0;
  const WorkPlan* wp = nullptr;
  void* ws = nullptr;
  const uint8_t* packed_dev = nullptr;
  int concurrent = 0;               // built for several batches in flight (wide-tile alternatives from a smaller grid on)
  int groups = 0;                   // group launches (conv_bgroup.hip) allowed: one batch at a time on a stream of >= 64 CUs
  std::vector<Launch> steps;
  int logits_direct = -1;           // index of the conv step that writes the dense logits itself (its y is patched per call), else -1
  int n_groups = 0;                 // group launches (conv_bgroup.hip) in the plan
  // Net::run walks a plan OUTSIDE the handle's mutex (several host threads, one stream and workspace each, enqueue at the same
  // time): `enqueue` serialises walks of this plan (one workspace = one step at a time anyway), `walkers` (under the handle's
  // mutex) keeps the plan from being evicted while somebody walks it
  std::shared_ptr<std::mutex> enqueue = std::make_shared<std::mutex>();
  int walkers = 0;
};

struct RunOpts {           // run-time switches, read from the TF2_AMD_OPTS snapshot (opts.h) by Net::load_options (tf2_net_reload_options)
  int flags = 0;           // ConvGeom::flags
  int pw_mode = 1, sk_mode = 0;
  long bneck_min_blocks = 200;
  int fire_mode = 2;         // fire: a fire module (squeeze + merged expands) as ONE launch (conv_fire.hip): 0 never, 1 wherever it fits, 2 (default) on maps
                             // >= 28 wide (14 x 14 modules measured SLOWER fused: 64-128 blocks of long dependent chains, r05_experiments.txt item 10)
  int fire_pool = 3;         // fire_pool: fire modules with a pool behind their expands: 0 never, 1 on maps >= 56 wide (the pool stays a launch), 2 wherever `fire` allows (pool a launch), 3 (default) the pool inside the fire launch where its form fits one batch at a time / as 2 with batches in flight, 4 inside always
  int first_fuse = 1;        // first: a 3x3 / stride 1 first layer on the 3-channel image as ONE launch with its input preparation (conv_first_kernel): 1 (default) / 0
  int first_pool = 1;        // first_pool: ... and a first layer's 3x3 / 2 max pool in that launch too (conv_first_pool_kernel): 1 (default) / 0
  int c3_pool = 1;           // c3_pool: a layer's 2x2 / 2 max pool inside its conv_c3 launch (tiles of TH x 32 pixels): 1 (default) / 0 its own launch
  int c3_w9 = 1;             // c3_w9: conv_c3_w9_kernel 0 never, 1 (default) where a block walks at least eight tiles, 2 wherever the layer allows it (tests)
  int pw_slabs = 1; long pw_minpix = 8192;     // conv_pw eligibility: most K slabs, fewest pixels
  int stem_pk_small = 1;     // stem_pk_small: conv_stem_pool_kernel at small batches with fewer pooled rows per block (a grid of >= ~192 blocks): 1 (default) / 0
  int sk_kb = 1;             // sk_kb: split-K launches of at most sk_kb_blocks blocks (batch 1-4: the 7 x 7 and 14 x 14 maps) split K over blocks as well: 1 (default) / 0
  int sk_kb_blocks = 8;      // sk_kb_blocks: largest grid (64 x 64 output tiles) that takes it (the 7 x 7 maps at batch 1: -3 us per 3x3 row; 16-block grids -- the 14 x 14 maps -- measured 0.3-1.4 us SLOWER: the exchange costs ~3 us)
  int sk_kb_max = 8;         // sk_kb_max: most blocks per output tile
  int sk_kb_min = 8;         // sk_kb_min: fewest -- the slab list must be long enough for that many parts of eight wave items (two parts of a short list cost GoogLeNet's batch-1 step 25 us)
  int q128_flags = 1;        // q128: the input preparation tells conv_stem_pool_kernel per image whether a -128 is there (no scan of the input tile) where the step starts prep | stem + pool | conv_bfirst: 1 (default) / 0
  int pwk_mode = 1;          // pwk: short-K pointwise rows (2 .. 8 slabs) on conv_pwk.hip (the pixel tile's whole K extent resident in LDS) instead of the ring kernel: 0 never, 1 (default) with batches in flight, 2 one batch at a time as well
  long pwk_minpix = 4096;    // pwk_minpix: fewest output pixels
  int pwk_max_slabs = 4;     // pwk_slabs: most K slabs of a conv_pwk row (2, 4 or 8; 8 = the 512-channel rows too: 256 registers, no overlapped epilogue -- 1.3 % slower in flight than the ring kernel's pair)
  unsigned long long pwk_rows = 0, nopwk_rows = 0;   // test-only per-row switches (bit l = table row l): rows taken whatever pwk_units / pwk_minpix / pwk_slabs say, rows kept off
  int pwk_sk = 0;            // pwk_sk: 1 = rows that would take the in-block split-K kernel as well
  int c3_mode = 1;           // c3: 3x3 / 1 / pad 1 layers of big maps on conv_c3.hip (halo tile in LDS) instead of the ring kernel
  int c3_min_hw = 14;        // c3_min_hw: smallest map side that takes conv_c3
  long c3_min_blocks = 96;   // c3_min: smallest grid that takes it
  int fc_mode = 1;           // fc: whole-window layers at batch <= 32 on conv_fc.hip
  int fc_min_slabs = 64;     // fc_min: shortest K (64-byte slabs) that takes it
  long c3_min256 = 200;      // c3_min256: smallest grid of 256-channel blocks (one-window layers; else 128-channel blocks)
  long alt_min_blocks = 200;   // alt_min: smallest 128 x 128 grid that takes a wide-tile alternative, one batch at a time
  long alt_narrow_blocks = 64;     // alt_narrow
  long alt_min_blocks_conc = 90;   // alt_min_conc: the same when the caller keeps several batches in flight
  unsigned long long alt_rows = 0, noalt_rows = 0, sk_rows = 0, nosk_rows = 0;   // test-only per-row switches (bit l = table row l)
  int alt_conc_mode = 2;       // alt_conc: 0 never assume concurrency, 1 always, 2 auto (calls on >= 2 streams among the last 8)
  int dense_max_slabs = 17;   // dense_max: layers with more K slabs than this on grids of more than one round keep the header tables
  int dense_mode = 1;      // dense: gather words of dense layers computed from the step index (1) or read from the header tables (0)
  int bgroup_mode = 1;     // bgroup (default on): identity bottlenecks of the 14 x 14 maps in one launch, eight blocks per image (conv_bgroup.hip); one batch at a time only
  int bgroup_chain = 5;             // consecutive identity bottlenecks of the 28 x 28 / 14 x 14 / 7 x 7 maps per group launch (1: one launch each)
  int bgroup_min7 = 12, bgroup_min14 = 12, bgroup_min28 = 12, bgroup_min56f = 12;   // bgroup_min7 / _MIN14 / _MIN28 / _MIN56F: smallest batch that takes them (a group is 8 CUs per image whatever the batch)
  int bfirst_mode = 2;     // bfirst: the first bottleneck of the 56 x 56 stage (shortcut | reduce, 3x3, expand) as one launch of independent row bands (conv_bfirst.hip): 0 never, 1 (default) with batches in flight, 2 one batch at a time as well (instead of the group launch)
  int bfirst_min = 12;     // bfirst_min: smallest batch that takes it (14 blocks per image)
  int bband_mode = 1;      // bband: identity bottlenecks as band launches (conv_bband.hip): 0 never, 1 (default) with batches in flight, 2 also one batch at a time (instead of the group launches)
  int bband_rows_dd = 7;                      // most rows per band of a 28 x 28 bottleneck whose reduce AND 3x3 are two-window layers (rounds 4-5: 4)
  int bband_rows = 7, bband_rows_alone = 2;   // bband_rows / _ROWS_ALONE: output rows per block (several batches in flight / one batch at a time)
  int bband_alone_maps = 0;   // bband_alone_maps: maps that take band launches one batch at a time too, instead of the group launches (bit 1: 28 x 28, bit 2: 14 x 14; bband=2 = both)
  int bband_min = 8;       // bband_min: smallest batch that takes them
  int pair_mode = 1;       // pair: two independent neighbouring layers (shortcut | first 1x1) in one conv_mfma2 launch
  int avg_fuse = 2;        // avg_fuse (0 never / 1 always / 2 one batch at a time): the global average of an end-pool layer inside its conv launch (conv_mfma_sk AVG)
  int stem_pool = 1;       // stem_pool: fuse the first layer's 3x3 / stride 2 max pool into the conv_stem launch
  int stem_mode = 1;       // conv_stem.hip for the executed first layer: 1 auto (default), 0 never (stem)
  long sk_s3_blocks = 256, sk_s3_blocks_conc = 0;   // largest split-K grid on three ring stages (100 KiB of LDS: the block owns its CU), one batch at a time / in flight
  long bg_poll_limit = 1 << 24, bg_withhold = 0;   // group launches: polls of a meeting before it is reported as failed (bgroup_polls); test-only bgroup_withhold
  long sk8_blocks = 128;   // largest split-K grid that takes the 8-wave form (sk8)
  long long* dbg = nullptr; long long* dbg2 = nullptr; int dbg_layer = -1;
};

struct Net {
  tf2_net_desc nd{};
  std::vector<tf2_layer_desc> layers;
  std::vector<int8_t> q;                 // [n_q_rows][max_out_channel], runtime (negated)
  std::vector<LayerModel> models;
  bool model_loaded = false;

  // packed image (host copy) and its device binding
  std::vector<uint8_t> packed;
  bool packed_valid = false;
  int pack_mode = 0;
  const uint8_t* packed_dev = nullptr;
  size_t packed_dev_bytes = 0;

  // physical input layout of every layer (decided once from the graph)
  struct InLayout { int Cp_in = 0; int half = 0; int signed_in = 0; };
  // layer 0 executed as a pointwise layer over the im2col image (init(); PrepArgs::rewrite == 2)
  bool im2col0 = false; int im_stride = 1, im_pad_h = 0, im_pad_w = 0;
  std::vector<InLayout> in_layout;
  std::vector<int> out_Cp;               // channel padding of each layer's own output tensor
  std::vector<int> concat_C;             // channels of each concat tensor

  std::map<std::pair<int, int>, WorkPlan> plans;   // (batch, keep_all) -> plan
  std::list<LaunchPlan> launch_plans;              // prepared steps, keyed by (batch, plan, workspace, packed image); a list: growth and
                                                   // eviction never move a plan another thread is walking (oldest evicted at 64, under run_mutex)
  std::mutex run_mutex;                            // serialises run(): plan lookup / build, stream history, enqueue (tf2_amd.h threading note)
  RunOpts opts;

  // profiling
  bool profiling = false;
  std::vector<float> prof_ms;
  std::vector<int32_t> prof_launches;
  std::vector<std::pair<void*, void*>> prof_events;   // pending (start, stop) hipEvents
  bool profiling_loop = false; // tf2_net_profile(net, 2): ONE event pair around the whole layer loop
  float prof_loop_ms = 0.f;    // its accumulated time and count: the per-layer pairs of mode 1 each add their own record
  int prof_loop_n = 0;         // handling, so bench.py rescales their sum to this
  std::vector<int> prof_event_layer;

  tf2_status init(const tf2_net_desc* nd, const tf2_layer_desc* layers);
  tf2_status quantization(const char* text, size_t len, int8_t* q, size_t cap, int32_t* n_read) const;
  tf2_status load_model(const float* model, size_t n_floats);
  tf2_status load_model_4bit(const uint8_t* bytes, size_t n_bytes);    // straight from the 4-bit codes, no float32 copy of the weights
  tf2_status finish_model();
  tf2_status pack(int mode);
  bool stem_selected(int batch) const;     // layer 0 runs on conv_stem.hip (x-only image tensor)
  const PackLayer* pack_layer(int l) const;
  const PackLayer* pack_layer_alt(int l) const;
  bool pair_candidate(int l) const;        // rows l and l + 1 are independent plain conv rows (one launch may compute both)
  uint64_t tables_hash() const;
  const WorkPlan* plan(int batch, bool keep_all);
  const LaunchPlan* launch_plan(int batch, const WorkPlan* wp, void* ws, bool concurrent, bool allow_groups);
  size_t workspace_size(int batch, bool keep_all);      // plan(...)->total_bytes under the handle's mutex
  tf2_status describe_workspace(int batch, bool keep_all, std::vector<TensorPlan>* tensors, std::vector<LayerExec>* rows);
  bool bgroup_first_at(int l) const;       // rows l .. l + 3 = projection shortcut | reduce, 3x3, expand of the 56 x 56 stage (conv_bgroup56f_kernel)
  bool bband_at(int l, int rows) const;
    // rows l, l + 1, l + 2 are an identity bottleneck conv_bband.hip can take with `rows` output rows per block
  tf2_layer_desc exec_desc(int l) const;   // row l as executed (merged rows: the 3x3 layer of both rows' channels)
  bool out_nonneg(int l) const;            // the tensor layer l writes holds no negative value
  bool res_nonneg_single_clamp(int l) const;     // row l's residual epilogue may take the one-clamp form (requant_epilogue.h RNN)
  bool fire_at(int l) const;    // rows l (squeeze), l + 1, l + 2 (merged expands) are a fire module conv_fire.hip can take
  bool c3_at(int l) const;      // layer l can run on conv_c3.hip
  bool fc_at(int l, int batch) const;   // ... on conv_fc.hip
  bool bgroup_at(int l) const;             // rows l, l + 1, l + 2 are an identity bottleneck conv_bgroup.hip can take (tables + packed image)
  long long stat_steps = 0, stat_group_steps = 0, stat_inflight_steps = 0, stat_small_mask_steps = 0;   // tf2_net_run_stats
  void* recent_streams[8] = {};  // streams of the last calls to run(): several distinct ones = batches in flight
  int recent_pos = 0;
  void load_options();
  size_t logits_bytes(int batch) const;
  tf2_status poll_error(int batch, void* ws, size_t ws_bytes, void* stream);
  tf2_status run(const void* images, bool images_are_q, int batch, void* ws, size_t ws_bytes,
                 int8_t* logits, void* stream, int concurrency = -1, void* mark_event = nullptr, int mark_after_layer = -1);
  int issue(const Launch& st, const LaunchPlan* lp, const void* images, bool images_are_q, int8_t* logits, void* stream);
  tf2_status describe_launches(int batch, bool concurrent, std::vector<std::pair<int, struct LaunchRecord>>* out);
  tf2_status read_layer(int layer, int batch, const void* ws, int8_t* dst, size_t cap, void* stream);
  void drain_profile();
};

const std::string& last_error();

// model4bit.cpp: one tensor of the 4-bit packed model file, in place
struct M4Tensor {
  int min_exp = 0, dtype = 0;
  size_t N = 0, C = 0, H = 0, W = 0, cnt = 0, words_per_row = 0;
  const uint8_t* payload = nullptr;
  // 4-bit code of weight `i` of the flattened [N][C][H][W] order (dtype 0)
  int code(size_t i) const {
    size_t word, j;
    if (W == 1) { word = i / 4; j = i % 4; }
    else { const size_t r = i / W, x = i % W; word = r * words_per_row + x / 3; j = x % 3; }
    return (int)((((unsigned)payload[2 * word] | ((unsigned)payload[2 * word + 1] << 8)) >> (4 * j)) & 15u);
  }
  float f32(size_t i) const { float v; std::memcpy(&v, payload + 4 * i, 4); return v; }      // dtype 1
};
struct M4Cursor {
  const uint8_t* p; size_t n; size_t pos = 0; int index = 0;
  M4Cursor(const uint8_t* bytes, size_t len) : p(bytes), n(len) {}
  bool done() const { return pos >= n; }
  std::string next(M4Tensor* t);          // error text, empty = ok
};
float m4_code_value(int code, int min_exp);
// decoder of the whole file into a float stream; returns an error text (empty = ok)
std::string model4bit_decode(const uint8_t* bytes, size_t n, std::vector<float>* out, size_t* n_floats);

}  // namespace tf2
