// capi.cpp -- the extern "C" surface declared in include/tf2_amd.h.
#include <cstring>
#include <new>
#include "tf2_net.h"
#include "tf2_device.h"
#include "opts.h"

using namespace tf2;

struct tf2_net { Net impl; };

#define CHECK_NET(n)                                              \
  if (!(n)) { set_error("null tf2_net handle"); return TF2_ERR_ARG; }

extern "C" {

const char* tf2_last_error(void) { return last_error().c_str(); }
int tf2_abi_version(void) { return 1; }
int tf2_has_device_code(void) { return 1; }
// 0: the product; 1: the -DTF2_PROBES timing build (leaves work out: tools/probe_run.py); 2: the -DTF2_CHECK_DMA build (vm_track.h)
int tf2_build_kind(void) {
#if defined(TF2_PROBES)
  return 1;
#elif defined(TF2_CHECK_DMA)
  return 2;
#else
  return 0;
#endif
}
#ifdef TF2_CHECK_DMA
// tools/dma_stress.py only (not part of include/tf2_amd.h): {fragment reads that saw the sentinel of a DMA not landed, reads checked}
int tf2_check_dma_errors(unsigned long long out[2]) {
  unsigned long long a[2] = {0, 0}, b[2] = {0, 0};
  tf2::conv_bband_check_counts(a); tf2::conv_c3_check_counts(b);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1];
  return 0;
}
#endif

uint8_t tf2_get_real(float w, int8_t expand) { return get_real(w, expand); }

tf2_status tf2_net_create(const tf2_net_desc* nd, const tf2_layer_desc* layers, tf2_net** out) {
  if (!nd || !layers || !out) { set_error("tf2_net_create: null argument"); return TF2_ERR_ARG; }
  // the option snapshot (TF2_AMD_OPTS, opts.h) is taken here and by tf2_net_reload_options, nowhere else
  { const std::string e = opts_reload(); if (!e.empty()) { set_error(e); return TF2_ERR_ARG; } }
  tf2_net* n = new (std::nothrow) tf2_net();
  if (!n) { set_error("out of memory"); return TF2_ERR_SIZE; }
  tf2_status st = n->impl.init(nd, layers);
  if (st != TF2_OK) { delete n; return st; }
  n->impl.load_options();
  *out = n;
  return TF2_OK;
}

void tf2_net_destroy(tf2_net* net) {
  if (!net) return;
  net->impl.drain_profile();
  delete net;
}

tf2_status tf2_quantization(const tf2_net* net, const char* q_text, size_t q_text_len, int8_t* q,
                            size_t q_capacity, int32_t* n_values_read) {
  CHECK_NET(net);
  if (!q_text || !q) { set_error("tf2_quantization: null argument"); return TF2_ERR_ARG; }
  return net->impl.quantization(q_text, q_text_len, q, q_capacity, n_values_read);
}

tf2_status tf2_net_set_q(tf2_net* net, const int8_t* q, size_t n_bytes) {
  CHECK_NET(net);
  const size_t need = (size_t)net->impl.nd.n_q_rows * net->impl.nd.max_out_channel;
  if (!q || n_bytes != need) { set_error("tf2_net_set_q: expected " + std::to_string(need) + " bytes"); return TF2_ERR_SIZE; }
  net->impl.q.assign(q, q + n_bytes);
  net->impl.model_loaded = false;
  net->impl.packed_valid = false;
  net->impl.launch_plans.clear();
  return TF2_OK;
}

tf2_status tf2_net_load_model(tf2_net* net, const float* model, size_t n_floats) {
  CHECK_NET(net);
  if (!model) { set_error("tf2_net_load_model: null model"); return TF2_ERR_ARG; }
  return net->impl.load_model(model, n_floats);
}

tf2_status tf2_model4bit_decode(const void* bytes, size_t n_bytes, float* floats, size_t capacity, size_t* n_floats) {
  if (!bytes) { set_error("tf2_model4bit_decode: null input"); return TF2_ERR_ARG; }
  std::vector<float> out;
  size_t cnt = 0;
  const std::string err = tf2::model4bit_decode((const uint8_t*)bytes, n_bytes, floats ? &out : nullptr, &cnt);
  if (!err.empty()) { set_error(err); return TF2_ERR_ARG; }
  if (n_floats) *n_floats = cnt;
  if (floats) {
    if (capacity < cnt) { set_error("tf2_model4bit_decode: buffer too small"); return TF2_ERR_SIZE; }
    std::memcpy(floats, out.data(), cnt * sizeof(float));
  }
  return TF2_OK;
}

tf2_status tf2_net_load_model_4bit(tf2_net* net, const void* bytes, size_t n_bytes) {
  CHECK_NET(net);
  if (!bytes) { set_error("tf2_net_load_model_4bit: null input"); return TF2_ERR_ARG; }
  return net->impl.load_model_4bit((const uint8_t*)bytes, n_bytes);
}

tf2_status tf2_net_get_codes(const tf2_net* net, int layer, uint8_t* codes, size_t capacity, size_t* n_bytes) {
  CHECK_NET(net);
  const Net& N = net->impl;
  if (!N.model_loaded || layer < 0 || layer >= N.nd.n_layers) { set_error("tf2_net_get_codes: no model / bad layer"); return TF2_ERR_STATE; }
  const auto& c = N.models[layer].codes;
  if (n_bytes) *n_bytes = c.size();
  if (codes) {
    if (capacity < c.size()) { set_error("tf2_net_get_codes: buffer too small"); return TF2_ERR_SIZE; }
    std::memcpy(codes, c.data(), c.size());
  }
  return TF2_OK;
}

tf2_status tf2_net_get_bias_bn(const tf2_net* net, int layer, int32_t* bias, int32_t* alpha, int32_t* beta, size_t capacity) {
  CHECK_NET(net);
  const Net& N = net->impl;
  if (!N.model_loaded || layer < 0 || layer >= N.nd.n_layers) { set_error("tf2_net_get_bias_bn: no model / bad layer"); return TF2_ERR_STATE; }
  const LayerModel& m = N.models[layer];
  if (capacity < m.bias.size()) { set_error("tf2_net_get_bias_bn: buffer too small"); return TF2_ERR_SIZE; }
  if (bias) std::memcpy(bias, m.bias.data(), m.bias.size() * 4);
  if (alpha) std::memcpy(alpha, m.alpha.data(), m.alpha.size() * 4);
  if (beta) std::memcpy(beta, m.beta.data(), m.beta.size() * 4);
  return TF2_OK;
}

tf2_status tf2_net_pack(tf2_net* net, int mode) { CHECK_NET(net); net->impl.launch_plans.clear(); net->impl.plans.clear(); return net->impl.pack(mode); }

size_t tf2_net_packed_size(const tf2_net* net) { return net && net->impl.packed_valid ? net->impl.packed.size() : 0; }

tf2_status tf2_net_packed_copy(const tf2_net* net, void* host_dst, size_t capacity) {
  CHECK_NET(net);
  if (!net->impl.packed_valid) { set_error("tf2_net_packed_copy: nothing packed"); return TF2_ERR_STATE; }
  if (!host_dst || capacity < net->impl.packed.size()) { set_error("tf2_net_packed_copy: buffer too small"); return TF2_ERR_SIZE; }
  std::memcpy(host_dst, net->impl.packed.data(), net->impl.packed.size());
  return TF2_OK;
}

tf2_status tf2_net_packed_adopt(tf2_net* net, const void* host_src, size_t n_bytes) {
  CHECK_NET(net);
  Net& N = net->impl;
  if (!host_src || n_bytes < sizeof(PackHeader)) { set_error("tf2_net_packed_adopt: image too small"); return TF2_ERR_SIZE; }
  PackHeader h;
  std::memcpy(&h, host_src, sizeof h);
  if (h.magic != kPackMagic || h.version != kPackVersion) { set_error("tf2_net_packed_adopt: not a tf2_amd packed image of this version"); return TF2_ERR_ARG; }
  if (h.total_bytes != n_bytes || h.n_layers != (uint32_t)N.nd.n_layers) { set_error("tf2_net_packed_adopt: size / layer count mismatch"); return TF2_ERR_SIZE; }
  if (h.tables_hash != N.tables_hash()) { set_error("tf2_net_packed_adopt: image was packed for different network tables"); return TF2_ERR_ARG; }
  N.packed.assign((const uint8_t*)host_src, (const uint8_t*)host_src + n_bytes);
  N.packed_valid = true;
  N.packed_dev = nullptr; N.packed_dev_bytes = 0;
  N.launch_plans.clear(); N.plans.clear();
  return TF2_OK;
}

tf2_status tf2_net_bind_device(tf2_net* net, const void* packed_dev, size_t n_bytes) {
  CHECK_NET(net);
  Net& N = net->impl;
  if (!N.packed_valid) { set_error("tf2_net_bind_device: nothing packed"); return TF2_ERR_STATE; }
  if (!packed_dev || n_bytes != N.packed.size()) { set_error("tf2_net_bind_device: size mismatch"); return TF2_ERR_SIZE; }
  N.packed_dev = (const uint8_t*)packed_dev;
  N.packed_dev_bytes = n_bytes;
  N.launch_plans.clear();
  return TF2_OK;
}

size_t tf2_net_workspace_size(tf2_net* net, int batch, int keep_all) {
  if (!net || batch <= 0) return 0;
  return net->impl.workspace_size(batch, keep_all != 0);
}

size_t tf2_net_logits_size(const tf2_net* net, int batch) {
  if (!net || batch <= 0) return 0;
  return net->impl.logits_bytes(batch);
}

tf2_status tf2_net_reload_options(tf2_net* net) {
  CHECK_NET(net);
  { const std::string e = opts_reload(); if (!e.empty()) { set_error(e); return TF2_ERR_ARG; } }
  net->impl.load_options();
  return TF2_OK;
}

tf2_status tf2_net_run(tf2_net* net, const float* images_dev, int batch, void* ws, size_t ws_bytes,
                       int8_t* logits_dev, void* hip_stream) {
  CHECK_NET(net);
  if (!images_dev || !ws) { set_error("tf2_net_run: null device pointer"); return TF2_ERR_ARG; }
  return net->impl.run(images_dev, false, batch, ws, ws_bytes, logits_dev, hip_stream);
}

tf2_status tf2_net_run_q(tf2_net* net, const int8_t* images_q_dev, int batch, void* ws, size_t ws_bytes,
                         int8_t* logits_dev, void* hip_stream) {
  CHECK_NET(net);
  if (!images_q_dev || !ws) { set_error("tf2_net_run_q: null device pointer"); return TF2_ERR_ARG; }
  return net->impl.run(images_q_dev, true, batch, ws, ws_bytes, logits_dev, hip_stream);
}

tf2_status tf2_net_run_ex(tf2_net* net, const void* images_dev, int batch, void* ws, size_t ws_bytes, int8_t* logits_dev,
                          void* hip_stream, const tf2_run_opts* o) {
  CHECK_NET(net);
  if (!images_dev || !ws) { set_error("tf2_net_run_ex: null device pointer"); return TF2_ERR_ARG; }
  if (!o || o->size < sizeof(tf2_run_opts)) { set_error("tf2_net_run_ex: opts missing or older than this library's tf2_run_opts"); return TF2_ERR_ARG; }
  if (o->concurrency < -1 || o->concurrency > 1) { set_error("tf2_net_run_ex: concurrency must be -1, 0 or 1"); return TF2_ERR_ARG; }
  if (o->mark_event && (o->mark_after_layer < 0 || o->mark_after_layer >= net->impl.nd.n_layers)) {
    set_error("tf2_net_run_ex: mark_after_layer outside the layer table"); return TF2_ERR_ARG;
  }
  return net->impl.run(images_dev, o->images_are_q != 0, batch, ws, ws_bytes, logits_dev, hip_stream, o->concurrency, o->mark_event, o->mark_after_layer);
}

tf2_status tf2_net_run_stats(tf2_net* net, int64_t* out4) {
  CHECK_NET(net);
  if (!out4) { set_error("tf2_net_run_stats: null argument"); return TF2_ERR_ARG; }
  std::lock_guard<std::mutex> lock(net->impl.run_mutex);
  out4[0] = net->impl.stat_steps; out4[1] = net->impl.stat_group_steps; out4[2] = net->impl.stat_inflight_steps; out4[3] = net->impl.stat_small_mask_steps;
  return TF2_OK;
}

tf2_status tf2_net_poll_error(tf2_net* net, int batch, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_NET(net);
  std::lock_guard<std::mutex> lock(net->impl.run_mutex);
  return net->impl.poll_error(batch, workspace, workspace_bytes, stream);
}

tf2_status tf2_net_describe_launches(tf2_net* net, int batch, int concurrency, tf2_launch_info* rows, int capacity, int* n) {
  CHECK_NET(net);
  if (!n || (capacity > 0 && !rows)) { set_error("tf2_net_describe_launches: null argument"); return TF2_ERR_ARG; }
  std::vector<std::pair<int, LaunchRecord>> v;
  tf2_status st = net->impl.describe_launches(batch, concurrency != 0, &v);
  if (st != TF2_OK) return st;
  *n = (int)v.size();
  if ((int)v.size() > capacity) { set_error("tf2_net_describe_launches: " + std::to_string(v.size()) + " launches, capacity " + std::to_string(capacity)); return TF2_ERR_SIZE; }
  for (size_t i = 0; i < v.size(); i++) {
    tf2_launch_info& r = rows[i];
    r.layer = v[i].first; r.grid = (int32_t)v[i].second.grid; r.block = (int32_t)v[i].second.block;
    r.lds_bytes = (int32_t)v[i].second.lds; r.vgprs = v[i].second.vgprs;
    std::memcpy(r.kernel, v[i].second.kernel, sizeof r.kernel);
  }
  return TF2_OK;
}

tf2_status tf2_net_describe_workspace(tf2_net* net, int batch, int keep_all, tf2_tensor_info* tensors, int tensor_capacity, int* n_tensors,
                                      tf2_row_tensors* rows, int row_capacity) {
  CHECK_NET(net);
  if (!n_tensors || (tensor_capacity > 0 && !tensors) || (row_capacity > 0 && !rows)) { set_error("tf2_net_describe_workspace: null argument"); return TF2_ERR_ARG; }
  std::vector<tf2::TensorPlan> t;
  std::vector<tf2::LayerExec> r;
  tf2_status st = net->impl.describe_workspace(batch, keep_all != 0, &t, &r);
  if (st != TF2_OK) return st;
  *n_tensors = (int)t.size();
  if ((int)t.size() > tensor_capacity || (int)r.size() > row_capacity) { set_error("tf2_net_describe_workspace: " + std::to_string(t.size()) + " tensors, " + std::to_string(r.size()) + " rows: capacity too small"); return TF2_ERR_SIZE; }
  for (size_t i = 0; i < t.size(); i++) {
    tensors[i].offset = (int64_t)t[i].offset; tensors[i].bytes = (int64_t)t[i].bytes;
    tensors[i].first_row = t[i].first_use; tensors[i].last_row = t[i].last_use;
  }
  for (size_t i = 0; i < r.size(); i++) {
    rows[i].in_tensor = r[i].in_tensor; rows[i].out_tensor = r[i].out_tensor; rows[i].conv_tensor = r[i].conv_tensor; rows[i].res_tensor = r[i].res_tensor;
  }
  return TF2_OK;
}

tf2_status tf2_net_read_layer(tf2_net* net, int layer, int batch, const void* ws, int8_t* host_dst,
                              size_t capacity, void* hip_stream) {
  CHECK_NET(net);
  if (!ws || !host_dst) { set_error("tf2_net_read_layer: null pointer"); return TF2_ERR_ARG; }
  return net->impl.read_layer(layer, batch, ws, host_dst, capacity, hip_stream);
}

tf2_status tf2_net_profile(tf2_net* net, int enable) {
  CHECK_NET(net);
  Net& N = net->impl;
  N.drain_profile();
  if (enable) { std::fill(N.prof_ms.begin(), N.prof_ms.end(), 0.f); std::fill(N.prof_launches.begin(), N.prof_launches.end(), 0); N.prof_loop_ms = 0.f; N.prof_loop_n = 0; }
  N.profiling = enable == 1;
  N.profiling_loop = enable == 2;
  return TF2_OK;
}

tf2_status tf2_net_profile_read(tf2_net* net, float* ms, int32_t* launches, int32_t* kinds, int capacity) {
  CHECK_NET(net);
  Net& N = net->impl;
  if (capacity < N.nd.n_layers) { set_error("tf2_net_profile_read: capacity < n_layers"); return TF2_ERR_SIZE; }
  N.drain_profile();
  for (int l = 0; l < N.nd.n_layers; l++) {
    if (ms) ms[l] = N.prof_ms[l];
    if (launches) launches[l] = N.prof_launches[l];
    if (kinds) { const PackLayer* pl = N.pack_layer(l); kinds[l] = pl ? pl->kind : 0; }
  }
  return TF2_OK;
}

tf2_status tf2_net_profile_loop_read(tf2_net* net, float* ms_total, int32_t* runs) {
  CHECK_NET(net);
  Net& N = net->impl;
  N.drain_profile();
  if (ms_total) *ms_total = N.prof_loop_ms;
  if (runs) *runs = N.prof_loop_n;
  return TF2_OK;
}

// Evaluation(), network_helper.cpp:143-207: feature = out / (1 << Q); k bubble passes with
// '>' (ties: the larger index ends up on top).
tf2_status tf2_topk(const int8_t* logits, const int8_t* q_last_row, int n, int k, int32_t* labels, float* features) {
  if (!logits || !q_last_row || !labels || n <= 0 || k <= 0 || k > n) { set_error("tf2_topk: bad argument"); return TF2_ERR_ARG; }
  std::vector<float> f(n);
  std::vector<int32_t> lab(n);
  for (int i = 0; i < n; i++) {
    const int sh = -(int)q_last_row[i];
    if (sh < 0 || sh > 30) { set_error("tf2_topk: Q of the last layer must be in 0..30 (network_helper.cpp:181)"); return TF2_ERR_ARG; }
    f[i] = (float)logits[i] / (float)(1 << sh);
    lab[i] = i;
  }
  for (int pass = 0; pass < k; pass++)
    for (int j = 0; j + 1 < n - pass; j++)
      if (f[j] > f[j + 1]) { std::swap(f[j], f[j + 1]); std::swap(lab[j], lab[j + 1]); }
  for (int i = 0; i < k; i++) { labels[i] = lab[n - 1 - i]; if (features) features[i] = f[n - 1 - i]; }
  return TF2_OK;
}

}  // extern "C"
