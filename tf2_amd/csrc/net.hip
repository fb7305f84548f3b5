// net.hip -- network handle, workspace planning and the layer executor (host code,
// compiled with hipcc for the HIP runtime API).
//
// Replaces NetWork::Init/InitBuffer (host/src/network.cpp:22-150) and Runner::Run
// (host/src/runner.cpp:54-198): instead of one OpenCL queue per FPGA kernel and a
// cycle-scheduled pipeline, every layer is 1-3 kernel launches on the caller's HIP
// stream over NHWC int8 activation tensors that live in a caller-owned workspace.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include "tf2_net.h"

namespace tf2 {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
const std::string& last_error() { return g_err; }

#define HIP_OK(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
      return TF2_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

tf2_status Net::init(const tf2_net_desc* d, const tf2_layer_desc* ls) {
  nd = *d;
  if (nd.n_layers <= 0 || nd.n_q_rows < nd.n_layers + 1 || nd.max_out_channel <= 0) {
    set_error("tf2_net_create: bad net desc"); return TF2_ERR_ARG;
  }
  layers.assign(ls, ls + nd.n_layers);
  const int nl = nd.n_layers;
  // concat tensor widths
  concat_C.assign(std::max(0, nd.n_concat), 0);
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc& L = layers[l];
    if (L.concat >= 0) {
      if (L.concat >= nd.n_concat) { set_error("layer " + std::to_string(l) + ": concat id out of range"); return TF2_ERR_ARG; }
      if (L.n_start % 16) { set_error("layer " + std::to_string(l) + ": concat slice must start at a multiple of 16 channels"); return TF2_ERR_UNSUPPORTED; }
      concat_C[L.concat] = std::max(concat_C[L.concat], L.n_start + L.N);
    }
  }
  out_Cp.assign(nl, 0);
  in_layout.assign(nl, InLayout());
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc& L = layers[l];
    if (L.src >= l || L.add_src >= l) { set_error("layer " + std::to_string(l) + ": forward reference"); return TF2_ERR_ARG; }
    if (L.N <= 0 || L.N > nd.max_out_channel) { set_error("layer " + std::to_string(l) + ": bad N"); return TF2_ERR_ARG; }
    if (L.pool_en && L.add_src >= 0) { set_error("layer " + std::to_string(l) + ": pool + residual in one layer is not supported"); return TF2_ERR_UNSUPPORTED; }
    out_Cp[l] = round_up(L.N, 16);
    InLayout il;
    int srcC, srcH, srcW;
    if (L.src == -1) {
      il.half = round_up(L.C, 16);
      il.Cp_in = 2 * il.half;
      il.signed_in = 1;
      srcC = L.C; srcH = L.H; srcW = L.W;
    } else if (L.src >= 0) {
      const tf2_layer_desc& S = layers[L.src];
      il.Cp_in = S.concat >= 0 ? round_up(concat_C[S.concat], 16) : out_Cp[L.src];
      srcC = S.concat >= 0 ? concat_C[S.concat] : S.N;
      srcH = S.endpool ? 1 : S.PH; srcW = S.endpool ? 1 : S.PW;
    } else {
      const int cid = -(L.src + 2);
      if (cid >= nd.n_concat) { set_error("layer " + std::to_string(l) + ": concat source out of range"); return TF2_ERR_ARG; }
      il.Cp_in = round_up(concat_C[cid], 16);
      srcC = concat_C[cid];
      srcH = L.H; srcW = L.W;
      for (int j = 0; j < l; j++)
        if (layers[j].concat == cid) { srcH = layers[j].endpool ? 1 : layers[j].PH; srcW = layers[j].endpool ? 1 : layers[j].PW; }
    }
    if (!L.ipool && (srcC != L.C || srcH != L.H || srcW != L.W)) {
      set_error("layer " + std::to_string(l) + ": input " + std::to_string(L.C) + "x" + std::to_string(L.H) + "x" + std::to_string(L.W) +
                " does not match its producer's " + std::to_string(srcC) + "x" + std::to_string(srcH) + "x" + std::to_string(srcW));
      return TF2_ERR_ARG;
    }
    in_layout[l] = il;
    if (L.add_src >= 0) {
      const tf2_layer_desc& R = layers[L.add_src];
      if (R.N != L.N || R.PH != L.PH || R.PW != L.PW || R.endpool) { set_error("layer " + std::to_string(l) + ": residual shape mismatch"); return TF2_ERR_ARG; }
    }
  }
  prof_ms.assign(nl, 0.f); prof_launches.assign(nl, 0);
  return TF2_OK;
}

// ---- workspace planning: first-fit offsets with liveness-based reuse ---------------
const WorkPlan* Net::plan(int batch, bool keep_all) {
  auto key = std::make_pair(batch, keep_all ? 1 : 0);
  auto it = plans.find(key);
  if (it != plans.end()) return &it->second;
  WorkPlan wp;
  wp.batch = batch; wp.keep_all = keep_all;
  const int nl = nd.n_layers;
  auto add_tensor = [&](int H, int W, int C, int Cp) {
    TensorPlan t; t.H = H; t.W = W; t.C = C; t.Cp = Cp;
    t.bytes = ((size_t)batch * H * W * Cp + 255) / 256 * 256;
    wp.tensors.push_back(t);
    return (int)wp.tensors.size() - 1;
  };
  const tf2_layer_desc& L0 = layers[0];
  wp.input_tensor = add_tensor(L0.H, L0.W, L0.C, in_layout[0].Cp_in);
  std::vector<int> concat_tensor(std::max(0, nd.n_concat), -1);
  std::vector<int> layer_out(nl, -1);
  wp.exec.assign(nl, LayerExec());
  // creation index of every tensor (the layer that first writes it); temps die in-layer
  std::vector<int> born;
  born.push_back(-1);
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc& L = layers[l];
    LayerExec& E = wp.exec[l];
    E.in_tensor = L.src == -1 ? wp.input_tensor : (L.src >= 0 ? layer_out[L.src] : concat_tensor[-(L.src + 2)]);
    const int oh = L.endpool ? 1 : L.PH, ow = L.endpool ? 1 : L.PW;
    if (L.concat >= 0) {
      if (concat_tensor[L.concat] < 0) {
        concat_tensor[L.concat] = add_tensor(oh, ow, concat_C[L.concat], round_up(concat_C[L.concat], 16));
        born.push_back(l);
      }
      E.out_tensor = concat_tensor[L.concat];
      E.out_off = L.n_start;
    } else {
      E.out_tensor = add_tensor(oh, ow, L.N, out_Cp[l]);
      born.push_back(l);
    }
    layer_out[l] = E.out_tensor;
    E.conv_tensor = E.out_tensor;
    if (!L.ipool && (L.pool_en || L.endpool)) {
      // conv (+residual) result before pooling / global average
      const int th = L.pool_en ? L.OH : L.PH, tw = L.pool_en ? L.OW : L.PW;
      E.conv_tensor = add_tensor(th, tw, L.N, out_Cp[l]);
      born.push_back(l);
      wp.tensors[E.conv_tensor].last_use = l;
    }
    if (L.add_src >= 0) {
      E.res_tensor = layer_out[L.add_src];
      E.res_off = layers[L.add_src].concat >= 0 ? layers[L.add_src].n_start : 0;
    }
    // liveness
    wp.tensors[E.in_tensor].last_use = std::max(wp.tensors[E.in_tensor].last_use, l);
    if (E.res_tensor >= 0) wp.tensors[E.res_tensor].last_use = std::max(wp.tensors[E.res_tensor].last_use, l);
    wp.tensors[E.out_tensor].last_use = std::max(wp.tensors[E.out_tensor].last_use, l);
  }
  wp.final_tensor = layer_out[nl - 1];
  wp.tensors[wp.final_tensor].last_use = nl;
  // ---- offsets ----
  struct Seg { size_t off, len; };
  std::vector<Seg> free_list;           // sorted by offset
  size_t top = 0;
  auto alloc = [&](size_t len) -> size_t {
    for (size_t i = 0; i < free_list.size(); i++)
      if (free_list[i].len >= len) {
        size_t off = free_list[i].off;
        free_list[i].off += len; free_list[i].len -= len;
        if (free_list[i].len == 0) free_list.erase(free_list.begin() + i);
        return off;
      }
    // extend a free segment that touches the top
    if (!free_list.empty() && free_list.back().off + free_list.back().len == top) {
      size_t off = free_list.back().off;
      top = off + len;
      free_list.pop_back();
      return off;
    }
    size_t off = top; top += len; return off;
  };
  auto release = [&](size_t off, size_t len) {
    Seg s{off, len};
    auto pos = std::lower_bound(free_list.begin(), free_list.end(), s, [](const Seg& a, const Seg& b) { return a.off < b.off; });
    pos = free_list.insert(pos, s);
    size_t i = pos - free_list.begin();
    if (i + 1 < free_list.size() && free_list[i].off + free_list[i].len == free_list[i + 1].off) {
      free_list[i].len += free_list[i + 1].len; free_list.erase(free_list.begin() + i + 1);
    }
    if (i > 0 && free_list[i - 1].off + free_list[i - 1].len == free_list[i].off) {
      free_list[i - 1].len += free_list[i].len; free_list.erase(free_list.begin() + i);
    }
  };
  std::vector<char> placed(wp.tensors.size(), 0), freed(wp.tensors.size(), 0);
  wp.tensors[wp.input_tensor].offset = alloc(wp.tensors[wp.input_tensor].bytes);
  placed[wp.input_tensor] = 1;
  for (int l = 0; l < nl; l++) {
    for (size_t t = 0; t < wp.tensors.size(); t++)
      if (!placed[t] && born[t] == l) { wp.tensors[t].offset = alloc(wp.tensors[t].bytes); placed[t] = 1; }
    if (!keep_all)
      for (size_t t = 0; t < wp.tensors.size(); t++)
        if (placed[t] && !freed[t] && wp.tensors[t].last_use <= l && (int)t != wp.final_tensor) {
          release(wp.tensors[t].offset, wp.tensors[t].bytes); freed[t] = 1;
        }
  }
  wp.dump_off = (top + 255) / 256 * 256;      // 16 KiB scratch: where masked lanes of conv_mfma_p.hip store
  wp.total_bytes = wp.dump_off + 16384;
  auto res = plans.emplace(key, std::move(wp));
  return &res.first->second;
}

// ---- run ----------------------------------------------------------------------------
tf2_status Net::run(const void* images, bool images_are_q, int batch, void* ws, size_t ws_bytes,
                    int8_t* logits, void* stream) {
  if (!packed_valid) { set_error("tf2_net_run: no packed image (tf2_net_pack / tf2_net_packed_adopt)"); return TF2_ERR_STATE; }
  if (!packed_dev) { set_error("tf2_net_run: packed image not bound to the device (tf2_net_bind_device)"); return TF2_ERR_STATE; }
  if (q.empty()) { set_error("tf2_net_run: q table not set"); return TF2_ERR_STATE; }
  if (batch <= 0) { set_error("tf2_net_run: batch must be positive"); return TF2_ERR_ARG; }
  // keep_all plans are a superset in size; pick whichever plan fits the caller's buffer
  const WorkPlan* wp = nullptr;
  {
    const WorkPlan* a = plan(batch, false);
    auto itk = plans.find(std::make_pair(batch, 1));
    if (itk != plans.end() && ws_bytes >= itk->second.total_bytes) wp = &itk->second;
    else wp = a;
  }
  if (ws_bytes < wp->total_bytes) { set_error("tf2_net_run: workspace too small"); return TF2_ERR_SIZE; }
  hipStream_t s = (hipStream_t)stream;
  int8_t* base = (int8_t*)ws;
  const int nl = nd.n_layers;
  auto T = [&](int id) -> const TensorPlan& { return wp->tensors[id]; };
  const uint8_t* pk = packed_dev;
  int flags = 0;
  if (const char* e = getenv("TF2_AMD_NOSWAP")) flags |= (e[0] == '1');
  if (const char* e = getenv("TF2_AMD_EXP")) flags |= atoi(e) & 6;    // conv_mfma2 block shape A/B switch: 2 = 4-wave, 4 = 16-wave
  bool mfma_v1 = false;
  if (const char* e = getenv("TF2_AMD_MFMA_V1")) mfma_v1 = e[0] == '1';
  // weight-stationary kernel for short-K pointwise layers: measured slower than conv_mfma2 at batch 32-128
  // on MI355X so far, hence opt-in (TF2_AMD_WS=3 auto, =1 forced with short runs for the tests)
  int ws_mode = 2;
  if (const char* e = getenv("TF2_AMD_WS")) ws_mode = atoi(e);
  // persistent tile-streaming kernel (conv_mfma_p.hip): 0 (default) never, 1 whenever the layer is not a split-K
  // one, 2 when every resident block gets at least two pixel tiles.  Measured no faster than conv_mfma2 so far:
  // both are bound by VALU issue (4 cycles per wave64 instruction per SIMD), not by the latency it hides.
  int p_mode = 0;
  if (const char* e = getenv("TF2_AMD_P")) p_mode = atoi(e);
  int pw_mode = 1;          // register-resident pointwise kernel (conv_pw.hip): 1 auto (default), 0 never
  if (const char* e = getenv("TF2_AMD_PW")) pw_mode = atoi(e);
  int sk_mode = 0;          // 0 auto, 1 force the in-block split-K kernel for every 64-row layer, 2 never
  if (const char* e = getenv("TF2_AMD_SK")) sk_mode = atoi(e);
  const uint64_t zero_off = reinterpret_cast<const PackHeader*>(packed.data())->zero_off;

  // input: quantise + (space-to-depth) + [x | xneg]
  {
    const tf2_layer_desc& L0 = layers[0];
    PrepArgs pa{};
    pa.img = images; pa.y = base + T(wp->input_tensor).offset;
    pa.B = batch; pa.C = nd.image_c; pa.H = nd.image_h; pa.W = nd.image_w;
    pa.OH = L0.H; pa.OW = L0.W; pa.y_cp = in_layout[0].Cp_in; pa.half = in_layout[0].half;
    pa.rewrite = nd.conv1_rewrite; pa.q0 = q[0]; pa.src_is_q = images_are_q ? 1 : 0;
    if (!nd.conv1_rewrite && (L0.H != nd.image_h || L0.W != nd.image_w || L0.C != nd.image_c)) {
      set_error("layer 0 input does not match the image"); return TF2_ERR_ARG;
    }
    if (launch_prep_input(pa, stream)) { set_error(std::string("prep_input launch: ") + device_last_error()); return TF2_ERR_HIP; }
  }

  hipEvent_t loop0 = nullptr, loop1 = nullptr;
  if (profiling_loop) {
    HIP_OK(hipEventCreate(&loop0)); HIP_OK(hipEventCreate(&loop1));
    HIP_OK(hipEventRecord(loop0, s));
  }
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc& L = layers[l];
    const LayerExec& E = wp->exec[l];
    const PackLayer* pl = pack_layer(l);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (profiling) {
      HIP_OK(hipEventCreate(&ev0)); HIP_OK(hipEventCreate(&ev1));
      HIP_OK(hipEventRecord(ev0, s));
    }
    if (L.ipool) {
      PoolArgs pa{};
      const TensorPlan& ti = T(E.in_tensor); const TensorPlan& to = T(E.out_tensor);
      pa.x = base + ti.offset; pa.y = base + to.offset;
      pa.B = batch; pa.H = ti.H; pa.W = ti.W; pa.x_cp = ti.Cp; pa.x_off = 0;
      pa.PH = L.PH; pa.PW = L.PW; pa.y_cp = to.Cp; pa.y_off = E.out_off;
      pa.S = L.pool_S; pa.st = L.pool_st; pa.pad = L.pool_pad; pa.C16 = round_up(L.N, 16) / 16;
      if (launch_maxpool(pa, stream)) { set_error("maxpool launch failed"); return TF2_ERR_HIP; }
    } else {
      ConvArgs ca{};
      const TensorPlan& ti = T(E.in_tensor); const TensorPlan& tc = T(E.conv_tensor);
      ca.x = base + ti.offset; ca.y = base + tc.offset;
      ca.w = (const int8_t*)(pk + pl->off_w); ca.w2 = (const int8_t*)(pk + pl->off_w2);
      ca.entries = (const int32_t*)(pk + pl->off_entries); ca.dir = (const int32_t*)(pk + pl->off_dir);
      ca.kinfo = (const int32_t*)(pk + pl->off_kinfo);
      ca.bias = (const int32_t*)(pk + pl->off_bias); ca.alpha = (const int32_t*)(pk + pl->off_alpha);
      ca.beta = (const int32_t*)(pk + pl->off_beta); ca.lo = (const int32_t*)(pk + pl->off_lo);
      ca.dshift = (const int32_t*)(pk + pl->off_dshift);
      ca.zero = (const int8_t*)(pk + zero_off); ca.max_ent = pl->max_ent;
      ca.dump = base + wp->dump_off;
      ca.dual = pl->dual;
      set_fast_div((uint32_t)pl->n_mtiles, &ca.mt_m, &ca.mt_s);
      if (pl->kind == KIND_MFMA) {
        ca.hdr = (const int32_t*)(pk + pl->off_hdr); ca.hdr_bytes = (int32_t)pl->hdr_bytes;
        if (pl->n_mtiles <= kMaxMtiles) {
          const int32_t* hd = reinterpret_cast<const int32_t*>(packed.data() + pl->off_dir);
          for (int mt = 0; mt < pl->n_mtiles; mt++) ca.e_start[mt] = hd[(size_t)mt * (pl->n_phases + 1)];
          ca.e_start[pl->n_mtiles] = pl->n_entries;
        } else {
          mfma_v1 = true;      // very wide layers: the register-staged kernel has no m-tile limit
        }
      }
      if (const char* e = getenv("TF2_AMD_DBGPTR2")) {
        const char* el = getenv("TF2_AMD_DBGLAYER");
        if (el && atoi(el) == l) ca.dbg2 = (long long*)strtoull(e, nullptr, 0);
      }
      if (const char* e = getenv("TF2_AMD_DBGPTR")) ca.dbg = (long long*)strtoull(e, nullptr, 0) + (size_t)l * 16;
      ca.n_phases = pl->n_phases; ca.n_mtiles = pl->n_mtiles; ca.Np = pl->Np; ca.nslab = pl->nslab;
      ca.k = L.k; ca.dil = L.dil; ca.n_cchunk = pl->n_cchunk; ca.Cp_half = in_layout[l].half;
      ConvGeom& g = ca.g;
      g.H = L.H; g.W = L.W; g.Cp_in = ti.Cp;
      g.OH = L.OH; g.OW = L.OW; g.OHW = L.OH * L.OW;
      set_fast_div((uint32_t)g.OHW, &g.ohw_m, &g.ohw_s); set_fast_div((uint32_t)g.OW, &g.ow_m, &g.ow_s);
      g.stride = L.stride; g.pad_h = L.pad_h; g.pad_w = L.pad_w;
      g.n_pix = batch * L.OH * L.OW;
      const bool direct = E.conv_tensor == E.out_tensor;
      g.y_cp = tc.Cp; g.y_off = direct ? E.out_off : 0;
      g.y_nvalid = round_up(L.N, 16);
      g.relu = L.relu; g.add_relu = L.add_relu; g.has_res = L.add_src >= 0;
      g.fast = pl->fast;
      if (g.has_res) {
        const TensorPlan& tr = T(E.res_tensor);
        ca.res = base + tr.offset; g.res_cp = tr.Cp; g.res_off = E.res_off;
      }
      g.flags = flags;
      int rc;
      if (pl->kind == KIND_MFMA) {
        // small grid + long slab list: the four waves of a block split K (conv_mfma_sk.hip)
        const long blocks64 = (long)((g.n_pix + 63) / 64) * pl->n_mtiles;
        const bool sk = pl->TM == 64 && pl->n_mtiles <= kMaxMtiles && sk_mode != 2 &&
                        (sk_mode == 1 || (blocks64 <= 512 && (long)pl->n_entries * (pl->dual ? 2 : 1) >= 16L * pl->n_mtiles));
        const bool v1 = (mfma_v1 || (flags & 1)) && !pl->dual;      // the register-staged kernel reads single-window tiles
        if (v1) rc = launch_conv_mfma(ca, pl->TM, stream);
        else if (pw_mode && L.k == 1 && p_mode == 0 && ws_mode == 2 && sk_mode != 1) {    // no other kernel forced
          // every m-tile's entry list must be slabs 0..nslab-1 (dense weights): read from the host copy of the image
          const int32_t* hd = reinterpret_cast<const int32_t*>(packed.data() + pl->off_dir);
          bool dense = true;
          for (int mt = 0; mt < pl->n_mtiles && dense; mt++)
            dense = hd[(size_t)mt * (pl->n_phases + 1) + pl->n_phases] - hd[(size_t)mt * (pl->n_phases + 1)] == pl->nslab;
          rc = launch_conv_pw(ca, pl->TM, pl->nslab, L.k, dense ? 1 : 0, stream);
          if (rc == 1 && sk) rc = launch_conv_mfma_sk(ca, stream);
        }
        else if (sk) rc = launch_conv_mfma_sk(ca, stream);
        else if (pl->dual) rc = 1;                      // dual-window layers: conv_mfma2 / conv_mfma_sk only
        else if (p_mode == 1 || (p_mode == 2 && (long)((g.n_pix + (pl->TM == 128 ? 127 : 255)) / (pl->TM == 128 ? 128 : 256)) * pl->n_mtiles >= 1024 && pl->n_mtiles <= 64))
          rc = launch_conv_mfma_p(ca, pl->TM, stream);
        else rc = 1;
        if (rc == 1 && !sk && !v1) {
          rc = (ws_mode == 2 || pl->dual) ? 1 : launch_conv_mfma_ws(ca, pl->TM, stream);     // short-K pointwise layers
          if (rc == 1) rc = launch_conv_mfma2(ca, pl->TM, stream);
        }
      }
      else if (pl->kind == KIND_SHIFT) rc = launch_conv_shift(ca, pl->signed_in, pl->max_shift <= 22, stream);
      else { set_error("layer " + std::to_string(l) + " has no packed kernel"); return TF2_ERR_STATE; }
      if (rc) { set_error("conv launch failed at layer " + std::to_string(l) + ": " + device_last_error()); return TF2_ERR_HIP; }
      if (L.pool_en) {
        PoolArgs pa{};
        const TensorPlan& to = T(E.out_tensor);
        pa.x = base + tc.offset; pa.y = base + to.offset;
        pa.B = batch; pa.H = L.OH; pa.W = L.OW; pa.x_cp = tc.Cp; pa.x_off = 0;
        pa.PH = L.PH; pa.PW = L.PW; pa.y_cp = to.Cp; pa.y_off = E.out_off;
        pa.S = L.pool_S; pa.st = L.pool_st; pa.pad = L.pool_pad; pa.C16 = round_up(L.N, 16) / 16;
        if (launch_maxpool(pa, stream)) { set_error("maxpool launch failed"); return TF2_ERR_HIP; }
      } else if (L.endpool) {
        AvgArgs aa{};
        const TensorPlan& to = T(E.out_tensor);
        aa.x = base + tc.offset; aa.y = base + to.offset;
        aa.B = batch; aa.HW = L.PH * L.PW; aa.x_cp = tc.Cp; aa.x_off = 0;
        aa.y_cp = to.Cp; aa.y_off = E.out_off; aa.C = round_up(L.N, 16); aa.mult = L.endpool_mult;
        if (launch_global_avg(aa, stream)) { set_error("global_avg launch failed"); return TF2_ERR_HIP; }
      }
    }
    if (profiling) {
      HIP_OK(hipEventRecord(ev1, s));
      prof_events.emplace_back((void*)ev0, (void*)ev1);
      prof_event_layer.push_back(l);
    }
  }
  if (profiling_loop) {
    HIP_OK(hipEventRecord(loop1, s));
    prof_events.emplace_back((void*)loop0, (void*)loop1);
    prof_event_layer.push_back(-1);
  }
  // dense logits [batch][N_last]
  if (logits) {
    const TensorPlan& tf = T(wp->final_tensor);
    const tf2_layer_desc& LL = layers[nl - 1];
    const size_t rows = (size_t)batch * tf.H * tf.W;
    HIP_OK(hipMemcpy2DAsync(logits, (size_t)LL.N, base + tf.offset + wp->exec[nl - 1].out_off, (size_t)tf.Cp,
                            (size_t)LL.N, rows, hipMemcpyDeviceToDevice, s));
  }
  return TF2_OK;
}

void Net::drain_profile() {
  for (size_t i = 0; i < prof_events.size(); i++) {
    hipEvent_t e0 = (hipEvent_t)prof_events[i].first, e1 = (hipEvent_t)prof_events[i].second;
    float ms = 0.f;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
      if (prof_event_layer[i] < 0) {
        prof_loop_ms += ms; prof_loop_n++;
      } else {
        prof_ms[prof_event_layer[i]] += ms;
        prof_launches[prof_event_layer[i]] += 1;
      }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  prof_events.clear(); prof_event_layer.clear();
}

tf2_status Net::read_layer(int layer, int batch, const void* ws, int8_t* dst, size_t cap, void* stream) {
  auto itk = plans.find(std::make_pair(batch, 1));
  if (itk == plans.end()) { set_error("tf2_net_read_layer: plan the workspace with keep_all first"); return TF2_ERR_STATE; }
  const WorkPlan& wp = itk->second;
  if (layer < -1 || layer >= nd.n_layers) { set_error("tf2_net_read_layer: bad layer"); return TF2_ERR_ARG; }
  int tid, off, C;
  if (layer == -1) { tid = wp.input_tensor; off = 0; C = layers[0].C; }
  else { tid = wp.exec[layer].out_tensor; off = wp.exec[layer].out_off; C = layers[layer].N; }
  const TensorPlan& t = wp.tensors[tid];
  const size_t npix = (size_t)batch * t.H * t.W;
  if (cap < npix * C) { set_error("tf2_net_read_layer: destination too small"); return TF2_ERR_SIZE; }
  std::vector<int8_t> tmp(npix * t.Cp);
  HIP_OK(hipMemcpyAsync(tmp.data(), (const int8_t*)ws + t.offset, tmp.size(), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  const size_t HW = (size_t)t.H * t.W;
  for (int b = 0; b < batch; b++)
    for (int c = 0; c < C; c++)
      for (size_t p = 0; p < HW; p++)
        dst[((size_t)b * C + c) * HW + p] = tmp[((size_t)b * HW + p) * t.Cp + off + c];
  return TF2_OK;
}

}  // namespace tf2

