// net.hip -- network handle, workspace planning and the layer executor (host code,
// compiled with hipcc for the HIP runtime API).
//
// Replaces NetWork::Init/InitBuffer (host/src/network.cpp:22-150) and Runner::Run
// (host/src/runner.cpp:54-198): instead of one OpenCL queue per FPGA kernel and a
// cycle-scheduled pipeline, every layer is 1-3 kernel launches on the caller's HIP
// stream over NHWC int8 activation tensors that live in a caller-owned workspace.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "tf2_net.h"
#include "tf2_device.h"
#include "opts.h"

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
  // A 3x3 first layer on a 3-channel image (VGG16, SSD300, SqueezeNet 1.1) is executed as a POINTWISE layer over the im2col image:
  // the input kernel writes, per OUTPUT pixel, the 27 values x[c][oh * stride - pad + fh][ow * stride - pad + fw] in the order
  // c * 9 + fh * 3 + fw (zero outside the image: sequencer.cl:287) as [x | xneg], 64 bytes -- exactly the bytes the plain form writes
  // per INPUT pixel for 3 channels padded to 16 -- and the layer becomes C = 27, k = 1, stride 1 on an OH x OW map: one 64-byte K slab
  // per output instead of five (nine taps of 32 bytes, 6 of them real).  Same sums term for term; the filter codes [N][3][3][3] of
  // LoadModel ARE [N][27][1][1] in that channel order, so nothing else changes (the reference does the same kind of thing for its
  // 7x7 first layers: model_loader.cpp:244-257, input_loader.cpp:98-116).  TF2_AMD_IM2COL0=0 keeps the plain form.
  im2col0 = false;
  {
    const bool off = opt("im2col0", 1) == 0;       // (read per handle: tests build both forms)
    // (another row that reads the IMAGE would find the im2col bytes in the input tensor: such programs keep the plain form)
    bool only_consumer = true;
    for (int l = 1; l < nl; l++) if (layers[l].src == -1) only_consumer = false;
    tf2_layer_desc& L0 = layers[0];
    if (!off && only_consumer && !nd.conv1_rewrite && L0.src == -1 && !L0.ipool && L0.k == 3 && L0.model_k == 3 && L0.C == 3 && L0.model_C == 3 &&
        L0.dil <= 1 && nd.image_c == 3 && L0.H == nd.image_h && L0.W == nd.image_w && L0.stride >= 1 &&
        L0.OH == (L0.H + 2 * L0.pad_h - 3) / L0.stride + 1 && L0.OW == (L0.W + 2 * L0.pad_w - 3) / L0.stride + 1) {
      im2col0 = true; im_stride = L0.stride; im_pad_h = L0.pad_h; im_pad_w = L0.pad_w;
      L0.C = 27; L0.k = 1; L0.stride = 1; L0.pad_h = L0.pad_w = 0; L0.H = L0.OH; L0.W = L0.OW;
    }
  }
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
    // the q table rows Quantization / LoadModel index with this row (quantization.cpp:42-49, model_loader.cpp:159-162)
    if (L.q_in_row < 0 || L.q_in_row >= nd.n_q_rows) { set_error("layer " + std::to_string(l) + ": q_in_row outside the q table"); return TF2_ERR_ARG; }
    if (L.C <= 0 || (!L.ipool && (l == 0 && im2col0 ? 3 : L.C) > nd.max_out_channel)) { set_error("layer " + std::to_string(l) + ": input channels exceed MAX_OUT_CHANNEL"); return TF2_ERR_ARG; }
    if (L.n_start < 0 || L.n_start + L.N > nd.max_out_channel) { set_error("layer " + std::to_string(l) + ": n_start + N exceeds MAX_OUT_CHANNEL"); return TF2_ERR_ARG; }
    if (L.concat >= 0 && nd.n_conv + 1 + L.concat >= nd.n_q_rows) { set_error("layer " + std::to_string(l) + ": concat Q row outside the q table"); return TF2_ERR_ARG; }
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
    if (L.ipool == 2 && (srcC != L.C || L.N != L.C || L.pool_en || L.endpool || L.add_src >= 0 || L.concat >= 0 || L.src < 0)) {
      set_error("layer " + std::to_string(l) + ": an L2Norm row maps C channels of a layer output onto C channels, nothing else"); return TF2_ERR_ARG;
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
    const PackLayer* plm = packed_valid ? pack_layer(l) : nullptr;
    if (plm && plm->merged_into >= 0) {
      // computed by the merged launch of the row in front of it: its conv-stage tensor is that row's (channels behind that row's)
      E.conv_tensor = wp.exec[plm->merged_into].conv_tensor;
      wp.tensors[E.conv_tensor].last_use = std::max(wp.tensors[E.conv_tensor].last_use, l);
    } else if (!L.ipool && (L.pool_en || L.endpool)) {
      // conv (+residual) result before pooling / global average (merged rows: both rows' channels)
      const int th = L.pool_en ? L.OH : L.PH, tw = L.pool_en ? L.OW : L.PW;
      const int Nx = (plm && plm->merge_next > 0) ? L.N + layers[plm->merge_next].N : L.N;
      E.conv_tensor = add_tensor(th, tw, Nx, round_up(Nx, 16));
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
  // Fused pairs (conv_bneck.hip: layer l computes layer fuse_next as well, block by block): everything the launch reads
  // stays live until the LATER layer's index, and everything it writes exists from the EARLIER one -- otherwise the
  // first-fit planner hands the expand's output the memory of the 3x3's input, which other blocks are still reading.
  if (packed_valid)
    for (int l = 0; l < nl; l++) {
      const PackLayer* pl = pack_layer(l);
      int b = pl ? pl->fuse_next : 0;
      if (b <= 0 && opts.pair_mode && pair_candidate(l) && !(l >= 2 && pair_candidate(l - 1))) b = l + 1;   // (pairs do not chain)
      if (b <= 0) continue;
      TensorPlan& tin = wp.tensors[wp.exec[l].in_tensor];
      tin.last_use = std::max(tin.last_use, b);
      if (wp.exec[l].res_tensor >= 0) wp.tensors[wp.exec[l].res_tensor].last_use = std::max(wp.tensors[wp.exec[l].res_tensor].last_use, b);
      for (size_t t = 0; t < wp.tensors.size(); t++)
        if (born[t] == b) born[t] = l;
    }
  // Group launches (conv_bgroup.hip: rows l .. l + 2 in one launch, images at different layers at the same time): what the launch
  // reads stays live to its last row, what it writes exists from its first
  if (packed_valid && (opts.bgroup_mode || opts.bband_mode || opts.bfirst_mode))
    for (int l = 0; l + 2 < nl; l++) {
      if ((opts.bgroup_mode || opts.bfirst_mode) && bgroup_first_at(l)) {          // (conv_bgroup56f_kernel or conv_bfirst_kernel: rows l .. l + 3)
        TensorPlan& tin = wp.tensors[wp.exec[l].in_tensor];
        tin.last_use = std::max(tin.last_use, l + 3);
        TensorPlan& tm1 = wp.tensors[wp.exec[l + 1].out_tensor];
        tm1.last_use = std::max(tm1.last_use, l + 3);
        for (size_t t = 0; t < wp.tensors.size(); t++)
          if (born[t] > l && born[t] <= l + 3) born[t] = l;
        if (opts.bgroup_mode) {
          if (!wp.ctrl_bytes) wp.ctrl_bytes = 256;
          wp.ctrl_bytes += (size_t)((batch + 7) / 8 * 8) * 128;
        }
        if (opts.bfirst_mode && opts.q128_flags && !wp.ctrl_bytes) wp.ctrl_bytes = 256;      // (the control header: the -128 flags of launch_plan)
        l += 3;
        continue;
      }
      const bool grp = opts.bgroup_mode && bgroup_at(l);
      // (band launches: the batch gate of launch_plan applies here too; whether batches are in flight is not known to the workspace
      //  plan -- one workspace serves both launch plans -- so a batch that COULD take band launches pays their longer lifetimes in
      //  both: INTEGRATION.md "Workspace")
      if (!grp && !(opts.bband_mode && batch >= opts.bband_min && (bband_at(l, opts.bband_rows) || bband_at(l, opts.bband_rows_alone)))) continue;
      TensorPlan& tin = wp.tensors[wp.exec[l].in_tensor];
      tin.last_use = std::max(tin.last_use, l + 2);
      TensorPlan& tm1 = wp.tensors[wp.exec[l].out_tensor];
      tm1.last_use = std::max(tm1.last_use, l + 2);
      for (size_t t = 0; t < wp.tensors.size(); t++)
        if (born[t] == l + 1 || born[t] == l + 2) born[t] = l;
      if (grp) {
        if (!wp.ctrl_bytes) wp.ctrl_bytes = 256;                       // the step counter
        wp.ctrl_bytes += (size_t)((batch + 7) / 8 * 8) * 128;          // three rows of eight flag words per image (roll call, two meetings)
      }
      l += 2;
    }
  // Fire launches (conv_fire.hip: rows l .. l + 2 in one launch): what the launch reads stays live to its last row, what it writes exists from
  // its first
  if (packed_valid && opts.fire_mode)
    for (int l = 0; l + 2 < nl; l++) {
      if (!fire_at(l)) continue;
      TensorPlan& tin = wp.tensors[wp.exec[l].in_tensor];
      tin.last_use = std::max(tin.last_use, l + 2);
      for (size_t t = 0; t < wp.tensors.size(); t++)
        if (born[t] == l + 1 || born[t] == l + 2) born[t] = l;
      l += 2;
    }
  // ... and consecutive identity bottlenecks of the 14 x 14 maps may share a launch (bgroup_chain): nothing such a run touches
  // shares memory (the exchange inside a launch is ordered by flags and cache scopes, not by kernel boundaries)
  if (packed_valid && opts.bgroup_mode && opts.bgroup_chain > 1)
    for (int l = 0; l + 2 < nl;) {
      if (!bgroup_at(l)) { l++; continue; }
      int e = l + 2;
      while (e + 3 < nl && bgroup_at(e + 1) && layers[e + 1].H == layers[l].H) e += 3;
      if (e > l + 2)
        for (size_t t = 0; t < wp.tensors.size(); t++) {
          if (born[t] > l && born[t] <= e) born[t] = l;
          if (wp.tensors[t].last_use >= l && wp.tensors[t].last_use < e && born[t] <= e) wp.tensors[t].last_use = e;
        }
      l = e + 1;
    }
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
  for (size_t t = 0; t < wp.tensors.size(); t++) wp.tensors[t].first_use = born[t];
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
  wp.ctrl_off = (top + 255) / 256 * 256 + 256;
  wp.ctrl_bytes = (wp.ctrl_bytes + 255) / 256 * 256;
  // split-K over blocks (launch_plan: grids of at most sk_kb_blocks blocks): 1024 ticket words behind the group flags, cleared with them
  if (packed_valid && opts.sk_kb && batch <= 8) {
    if (!wp.ctrl_bytes) wp.ctrl_bytes = 256;
    wp.ks_ctr_off = wp.ctrl_off + wp.ctrl_bytes; wp.ks_ctr_bytes = 4096; wp.ctrl_bytes += wp.ks_ctr_bytes;
  }
  // partial sums of the conv_fc launches (one at a time: the largest)
  wp.scratch_off = wp.ctrl_off + wp.ctrl_bytes;
  wp.scratch_bytes = 0;
  if (packed_valid)
    for (int l = 0; l < nl; l++) {
      const PackLayer* pl = pack_layer(l);
      if (pl && (opts.fc_mode || pl->fc4) && fc_at(l, batch)) wp.scratch_bytes = std::max(wp.scratch_bytes, conv_fc_scratch_bytes(pl->Np, pl->nslab, pl->dual, batch));
    }
  wp.scratch_bytes = (wp.scratch_bytes + 255) / 256 * 256;
  if (wp.ks_ctr_bytes) {                                   // partial tiles: sk_kb_blocks x sk_kb_max x 16 KB behind the conv_fc partial sums
    wp.ks_part_off = wp.scratch_off + wp.scratch_bytes;
    wp.ks_part_bytes = (size_t)std::max(1, opts.sk_kb_blocks) * std::max(1, opts.sk_kb_max) * 16384;
    wp.scratch_bytes += wp.ks_part_bytes;
  }
  wp.total_bytes = wp.scratch_off + wp.scratch_bytes;
  auto res = plans.emplace(key, std::move(wp));
  return &res.first->second;
}

// Rows l and l + 1 may share a launch (conv_mfma2_pair_kernel): plain convolution rows (no pool / average / concat slice /
// L2Norm), neither reads what the other writes, neither is part of a fused bottleneck pair.  In ResNet-50: a stage's shortcut
// convolution and the first 1x1 of its first bottleneck (both read the previous stage's output).
bool Net::pair_candidate(int l) const {
  if (l < 1 || l + 1 >= nd.n_layers) return false;
  const tf2_layer_desc& A = layers[l]; const tf2_layer_desc& B = layers[l + 1];
  for (const tf2_layer_desc* L : {&A, &B})
    if (L->ipool || L->pool_en || L->endpool || L->concat >= 0 || L->src < 0) return false;
  if (B.src == l || B.add_src == l) return false;
  if (A.add_src >= 0 || B.add_src >= 0) return false;          // (a residual source may be the partner's input chain; keep it simple)
  const PackLayer* pa = pack_layer(l); const PackLayer* pb = pack_layer(l + 1);
  if (!pa || !pb || pa->kind != KIND_MFMA || pb->kind != KIND_MFMA) return false;
  if (pa->fuse_next > 0 || pa->fused_into >= 0 || pb->fuse_next > 0 || pb->fused_into >= 0) return false;
  return true;
}

// Rows l .. l + 3 = projection shortcut (1x1, 64 -> 256) and reduce (1x1, 64 -> 64) of the same 56 x 56 input, 3x3, expand +
// residual from the shortcut: conv_bgroup56f_kernel.  Shortcut, reduce and expand all two-window (packed dual) or all one-window.
bool Net::bgroup_first_at(int l) const {
  if (l < 1 || l + 3 >= nd.n_layers) return false;
  const tf2_layer_desc& S = layers[l]; const tf2_layer_desc& A = layers[l + 1]; const tf2_layer_desc& B = layers[l + 2]; const tf2_layer_desc& E = layers[l + 3];
  for (const tf2_layer_desc* L : {&S, &A, &B, &E})
    if (L->ipool || L->pool_en || L->endpool || L->concat >= 0 || L->stride != 1 || L->dil != 1 || L->H != 56 || L->W != 56) return false;
  if (S.src < 0 || S.src != A.src || layers[S.src].concat >= 0 || S.k != 1 || A.k != 1 || S.pad_h || A.pad_h || S.add_src >= 0 || A.add_src >= 0) return false;
  if (S.C != 64 || S.N != 256 || A.C != 64 || A.N != 64) return false;
  if (B.src != l + 1 || B.k != 3 || B.pad_h != 1 || B.pad_w != 1 || B.add_src >= 0 || B.C != 64 || B.N != 64) return false;
  if (E.src != l + 2 || E.k != 1 || E.pad_h || E.add_src != l || E.C != 64 || E.N != 256) return false;
  int n_dual = 0;
  for (int k = l; k <= l + 3; k++) {
    const PackLayer* pl = pack_layer(k);
    if (!pl || pl->kind != KIND_MFMA || pl->Cp_in != 64 || (long)pl->n_entries != (long)pl->n_mtiles * pl->nslab) return false;
    if (k == l ? (pl->TM != 64 && pl->TM != 128) : pl->TM != 64) return false;
    const bool one_window = pl->n_phases == 1 && !pl->dual, dual = pl->n_phases == 2 && pl->dual;
    if (!one_window && !dual) return false;
    if (k == l + 2) { if (!one_window) return false; } else n_dual += dual ? 1 : 0;
    if (k == l && pl->off_dbl) return false;            // (the shortcut's output is only ever a residual)
  }
  return n_dual == 0 || n_dual == 3;
}

// Row l as it is EXECUTED: the table row, or -- merged rows (PackLayer::merge_next, weight_pack.cpp: a 1x1 row and the 3x3 / pad 1 row
// behind it, same input, adjacent concat slices) -- the 3x3 layer of both rows' output channels
tf2_layer_desc Net::exec_desc(int l) const {
  tf2_layer_desc L = layers[l];
  const PackLayer* pl = pack_layer(l);
  if (pl && pl->merge_next > 0) { L.N += layers[pl->merge_next].N; L.k = 3; L.pad_h = L.pad_w = 1; }
  return L;
}

// The tensor layer l writes holds no negative value (its last operation is a ReLU)
bool Net::out_nonneg(int l) const {
  if (l < 0) return false;                                  // the image
  const tf2_layer_desc& L = layers[l];
  if (L.ipool == 2) return false;                           // L2Norm: sign(w) * sign(x)
  if (L.ipool) return out_nonneg(L.src);                    // a pool row keeps its input's range
  return L.add_src >= 0 ? L.add_relu != 0 : L.relu != 0;
}

// Row l adds a residual under the conditions of requant_epilogue.h's RNN form: no ReLU of its own, a post-ReLU residual tensor, the sum
// clamped to [0, 127] -- clamp(clamp(y, -128, 127) + r, 0, 127) == clamp(y + r, 0, 127), the first clamp is left out
bool Net::res_nonneg_single_clamp(int l) const {
  const tf2_layer_desc& L = layers[l];
  return L.add_src >= 0 && !L.relu && L.add_relu && layers[L.add_src].concat < 0 && out_nonneg(L.add_src);
}

// Rows l, l + 1, l + 2 = 1x1 reduce, 3x3 / 1 / pad 1, 1x1 expand + residual from the reduce's input, of a shape conv_bgroup.hip
// is instantiated for, every row single-window in 64- or 128-row dense tiles.
bool Net::bgroup_at(int l) const {
  if (l < 1 || l + 2 >= nd.n_layers) return false;
  const tf2_layer_desc& A = layers[l]; const tf2_layer_desc& B = layers[l + 1]; const tf2_layer_desc& E = layers[l + 2];
  for (const tf2_layer_desc* L : {&A, &B, &E})
    if (L->ipool || L->pool_en || L->concat >= 0 || L->stride != 1 || L->dil != 1) return false;
  // a global average may end the bottleneck where the split-K kernel could fuse it as well (7 x 7 shape only)
  if (A.endpool || B.endpool || (E.endpool && !(opts.avg_fuse && A.H == 7))) return false;
  if (A.src < 0 || A.k != 1 || A.pad_h || A.pad_w || A.add_src >= 0) return false;
  if (B.src != l || B.k != 3 || B.pad_h != 1 || B.pad_w != 1 || B.add_src >= 0 || B.C != A.N || B.N != A.N) return false;
  if (E.src != l + 1 || E.k != 1 || E.pad_h || E.pad_w || E.add_src != A.src || E.N != A.C) return false;
  if (layers[A.src].concat >= 0 || A.H != A.W || !conv_bgroup_shape_ok(A.H, A.C, A.N)) return false;
  if (!res_nonneg_single_clamp(l + 2)) return false;        // (the group kernels' expands use the single-clamp form)
  for (int k = l; k <= l + 2; k++) {
    const PackLayer* pl = pack_layer(k);
    if (!pl || pl->kind != KIND_MFMA) return false;
    if (pl->Cp_in % 64 != 0 || (long)pl->n_entries != (long)pl->n_mtiles * pl->nslab) return false;      // dense tiles
    const bool one_window = pl->n_phases == 1 && !pl->dual, dual = pl->n_phases == 2 && pl->dual;
    if (A.H == 28) {
      // the 28 x 28 kernel: 64- or 128-row tiles (its header slots hold a 128-row m-tile), the expand in 128-row tiles; the reduce
      // may be a two-window layer; rows packed for a conv_bneck pair qualify (the pair's own entries are one dense m-tile)
      if ((pl->TM != 64 && pl->TM != 128) || (k == l + 2 && pl->TM != 128)) return false;
      if (!(one_window || (k <= l + 1 && dual))) return false;
    } else {
      if (pl->fuse_next > 0 || pl->fused_into >= 0 || pl->TM != 64) return false;      // (2 KiB header slots: 64-row m-tiles)
      if (!(one_window || (k == l && A.H == 7 && dual))) return false;                 // the 7 x 7 kernel's reduce may be two-window
    }
  }
  return true;
}

// Rows l, l + 1, l + 2 = an identity bottleneck (as bgroup_at) of a shape conv_bband.hip is instantiated for, every row a dense
// single-window layer.
bool Net::bband_at(int l, int rows) const {
  if (l < 1 || l + 2 >= nd.n_layers) return false;
  const tf2_layer_desc& A = layers[l]; const tf2_layer_desc& B = layers[l + 1]; const tf2_layer_desc& E = layers[l + 2];
  for (const tf2_layer_desc* L : {&A, &B, &E})
    if (L->ipool || L->pool_en || L->endpool || L->concat >= 0 || L->stride != 1 || L->dil != 1) return false;
  if (A.src < 0 || A.k != 1 || A.pad_h || A.pad_w || A.add_src >= 0) return false;
  if (B.src != l || B.k != 3 || B.pad_h != 1 || B.pad_w != 1 || B.add_src >= 0 || B.C != A.N || B.N != A.N) return false;
  if (E.src != l + 1 || E.k != 1 || E.pad_h || E.pad_w || E.add_src != A.src || E.N != A.C) return false;
  if (layers[A.src].concat >= 0) return false;
  if (!res_nonneg_single_clamp(l + 2)) return false;        // (conv_bband's expand uses the single-clamp form)
  {
    const PackLayer* p0 = pack_layer(l); const PackLayer* p1 = pack_layer(l + 1);
    if (!p0 || !p1) return false;
    if (!conv_bband_shape_ok(A.H, A.W, A.C, A.N, std::min(conv_bband_pick_rows(A.W, A.N, p0->dual, p1->dual, rows, opts.bband_rows_dd), A.H))) return false;
  }
  if (out_Cp[A.src] != A.C) return false;                  // the input tensor holds exactly C bytes per pixel
  for (int k = l; k <= l + 2; k++) {
    const PackLayer* pl = pack_layer(k);
    if (!pl || pl->kind != KIND_MFMA || (pl->TM != 64 && pl->TM != 128)) return false;
    if (pl->Cp_in % 64 != 0 || (long)pl->n_entries != (long)pl->n_mtiles * pl->nslab) return false;      // dense tiles
    if (pl->w_share) return false;
    const bool one_window = pl->n_phases == 1 && !pl->dual, dual = pl->n_phases == 2 && pl->dual;
    if (!one_window && !(dual && k < l + 2)) return false;                                                // (the expand: single-window only)
  }
  return conv_bband_windows_ok(A.N, pack_layer(l)->dual, pack_layer(l + 1)->dual);
}

// Rows l (1x1 squeeze, ReLU), l + 1 and l + 2 (the merged expand1x1 | expand3x3 pair, PackLayer::merge_next) of a fire module whose
// squeeze output nothing else reads: conv_fire.hip takes them as one launch (a pool behind the expands follows as its own launch)
bool Net::fire_at(int l) const {
  if (l < 0 || l + 2 >= nd.n_layers) return false;
  const tf2_layer_desc& A = layers[l]; const tf2_layer_desc& B = layers[l + 1];
  if (A.ipool || A.pool_en || A.endpool || A.concat >= 0 || A.add_src >= 0 || !A.relu || A.k != 1 || A.stride != 1 || (A.pad_h | A.pad_w) || A.src == -1) return false;
  if (in_layout[l].Cp_in != A.C || A.C % 64 != 0 || in_layout[l].signed_in) return false;
  const PackLayer* p0 = pack_layer(l); const PackLayer* p1 = pack_layer(l + 1); const PackLayer* p2 = pack_layer(l + 2);
  if (!p0 || !p1 || !p2 || p0->kind != KIND_MFMA || p1->kind != KIND_MFMA || p1->merge_next != l + 2 || p2->merged_into != l + 1) return false;
  if (B.src != l || B.endpool || B.add_src >= 0 || !B.relu) return false;      // (a pool behind the expands: its own launch after the fire launch)
  for (int j = 0; j < nd.n_layers; j++)
    if (j != l + 1 && j != l + 2 && (layers[j].src == l || layers[j].add_src == l)) return false;      // the squeeze's tensor is not written
  if (p0->TM != 64 || p0->n_mtiles != 1 || (long)p0->n_entries != p0->nslab || !(p0->n_phases == 1 || p0->dual) || p0->w_share || p0->fuse_next > 0 || p0->fused_into >= 0) return false;
  if (p1->n_phases != 1 || p1->dual || p1->w_share || (p1->TM != 64 && p1->TM != 128) || p1->Cp_in != round_up(A.N, 16) || p1->off_dbl) return false;
  return conv_fire_geometry(A.H, A.W, A.C, round_up(A.N, 16), p1->Np, p0->TM, p1->TM, p0->dual, 0, nullptr, nullptr) && p1->Np == layers[l + 1].N + layers[l + 2].N;
}

// a 3x3 / stride 1 / pad 1 layer on an unsigned tensor that holds exactly C (a multiple of 64) bytes per pixel, dense one- or two-window
// tiles of its own: conv_c3.hip takes it (the halo tile of the input streamed through LDS once instead of nine gathers)
bool Net::c3_at(int l) const {
  const tf2_layer_desc& L = layers[l];
  if (L.ipool || L.k != 3 || L.stride != 1 || L.dil != 1 || L.pad_h != 1 || L.pad_w != 1 || L.src < 0 || L.add_src >= 0 || L.endpool) return false;
  if (layers[L.src].concat >= 0 || out_Cp[L.src] != L.C || L.OH != L.H || L.OW != L.W) return false;
  const PackLayer* pl = pack_layer(l);
  if (!pl || pl->kind != KIND_MFMA || (pl->TM != 64 && pl->TM != 128) || pl->w_share || pl->signed_in) return false;
  if (pl->Cp_in != L.C || pl->Cp_in % 64 != 0 || pl->nslab != 9 * (pl->Cp_in / 64) || (long)pl->n_entries != (long)pl->n_mtiles * pl->nslab) return false;
  if (pl->fuse_next > 0 || pl->fused_into >= 0) return false;
  const bool one_window = pl->n_phases == 1 && !pl->dual, dual = pl->n_phases == 2 && pl->dual;
  if (!one_window && !dual) return false;
  return conv_c3_shape_ok(L.H, L.W, L.C, pl->Np, opts.c3_min_hw);
}

// a layer whose input is ONE filter window per image (k x k / pad 0 on a k x k map), K long, at batch <= 32, not the network's last
// (that one stores the dense logits itself): conv_fc.hip streams its weights over the whole chip
bool Net::fc_at(int l, int batch) const {
  const tf2_layer_desc& L = layers[l];
  if (l == nd.n_layers - 1) return false;
  { const PackLayer* p4 = pack_layer(l); if (batch > 32 && !(p4 && p4->fc4)) return false; }      // (4-bit code layers: any batch, in chunks of 32 -- they have no int8 tiles)
  if (L.ipool || L.k != L.H || L.k != L.W || L.stride != 1 || L.dil != 1 || L.pad_h || L.pad_w || L.OH != 1 || L.OW != 1) return false;
  if (L.src < 0 || L.add_src >= 0 || L.endpool || L.pool_en || L.concat >= 0 || layers[L.src].concat >= 0 || out_Cp[L.src] != L.C) return false;
  const PackLayer* pl = pack_layer(l);
  if (!pl || pl->kind != KIND_MFMA || (pl->TM != 64 && pl->TM != 128) || pl->w_share || pl->signed_in || pl->Np % 128 != 0) return false;
  if (pl->Cp_in != L.C || pl->Cp_in % 64 != 0 || pl->nslab != L.k * L.k * (pl->Cp_in / 64) || (long)pl->n_entries != (long)pl->n_mtiles * pl->nslab) return false;
  if ((pl->nslab < opts.fc_min_slabs && !pl->fc4) || pl->fuse_next > 0 || pl->fused_into >= 0) return false;
  const bool one_window = pl->n_phases == 1 && !pl->dual, dual = pl->n_phases == 2 && pl->dual;
  return one_window || dual;
}

// conv_stem.hip takes layer 0 when the packed image holds its x-only weight tiles (weight_pack.cpp) and the fast
// space-to-depth prep applies; the input tensor then carries 32 bytes per pixel in the same allocation.
bool Net::stem_selected(int batch) const {
  const tf2_layer_desc& L0 = layers[0];
  const PackLayer* p0 = pack_layer(0);
  const long long pixels = (long long)batch * L0.H * L0.W;
  return opts.stem_mode != 0 && p0 && p0->kind == KIND_MFMA && p0->off_w2 != 0 && nd.conv1_rewrite && nd.image_c == 3 &&
         in_layout[0].Cp_in == 64 && in_layout[0].half == 32 && L0.OH == L0.H - 2 && L0.OW == L0.W - 2 &&
         pixels * 64 < (1ll << 31) && (long long)batch * 3 * nd.image_h * nd.image_w < (1ll << 31);
}

// ---- run-time switches (A/B experiments and forced kernels for the tests), read when a launch plan is built ----
void Net::load_options() {
  RunOpts o;
  // (the snapshot of TF2_AMD_OPTS was taken by the caller: tf2_net_create / tf2_net_reload_options, opts.h)
  o.flags |= (int)opt("exp", 0) & 0x7ff8;    // timing-probe bits of the -DTF2_PROBES build (tf2_device.h kProbe*, tools/probe_run.py); nothing in the product reads them
  o.pw_mode = (int)opt("pw", o.pw_mode);        // register-resident pointwise kernel: 1 auto (default), 0 never
  o.sk_mode = (int)opt("sk", o.sk_mode);        // 0 auto, 1 force the in-block split-K kernel for every 64-row layer, 2 never
  o.sk8_blocks = (long)opt("sk8", o.sk8_blocks);
  o.bg_poll_limit = (long)opt("bgroup_polls", o.bg_poll_limit); o.bg_withhold = (long)opt("bgroup_withhold", 0);
  o.sk_s3_blocks = (long)opt("sk_s3", o.sk_s3_blocks); o.sk_s3_blocks_conc = (long)opt("sk_s3_conc", o.sk_s3_blocks_conc);
  o.fc_mode = (int)opt("fc", o.fc_mode);
  o.fc_min_slabs = (int)opt("fc_min", o.fc_min_slabs);
  o.c3_mode = (int)opt("c3", o.c3_mode);
  o.c3_min_blocks = (long)opt("c3_min", o.c3_min_blocks);
  o.c3_min256 = (long)opt("c3_min256", o.c3_min256);
  o.c3_w9 = (int)opt("c3_w9", o.c3_w9);
  o.c3_pool = (int)opt("c3_pool", o.c3_pool);
  o.first_fuse = (int)opt("first", o.first_fuse);
  o.first_pool = (int)opt("first_pool", o.first_pool);
  o.fire_mode = (int)opt("fire", o.fire_mode);
  o.fire_pool = (int)opt("fire_pool", o.fire_pool);
  o.bneck_min_blocks = (long)opt("bneck_min", o.bneck_min_blocks);   // smallest grid that takes conv_bneck (default 200)
  o.stem_mode = (int)opt("stem", o.stem_mode);
  o.bgroup_min7 = (int)opt("bgroup_min7", o.bgroup_min7);    // smallest batch that takes the group launches of the 7 x 7 / 14 x 14 bottlenecks
  o.bgroup_min14 = (int)opt("bgroup_min14", o.bgroup_min14);
  o.bgroup_min28 = (int)opt("bgroup_min28", o.bgroup_min28);
  o.bgroup_min56f = (int)opt("bgroup_min56f", o.bgroup_min56f);
  o.bgroup_chain = (int)opt("bgroup_chain", o.bgroup_chain);
  o.bgroup_mode = (int)opt("bgroup", o.bgroup_mode);     // 1: identity bottlenecks of the small maps as group launches (conv_bgroup.hip), one batch at a time
  o.bfirst_mode = (int)opt("bfirst", o.bfirst_mode);     // the first 56 x 56 bottleneck as one launch of row bands (conv_bfirst.hip): 0 never, 1 with batches in flight, 2 always
  o.bfirst_min = (int)opt("bfirst_min", o.bfirst_min);
  o.bband_mode = (int)opt("bband", o.bband_mode);        // identity bottlenecks as band launches (conv_bband.hip): 0 never, 1 with batches in flight, 2 always
  o.bband_rows = (int)opt("bband_rows", o.bband_rows);
  o.bband_rows_alone = (int)opt("bband_rows_alone", o.bband_rows_alone);
  o.bband_rows_dd = (int)opt("bband_rows_dd", o.bband_rows_dd);
  o.c3_min_hw = (int)opt("c3_min_hw", o.c3_min_hw);
  o.bband_min = (int)opt("bband_min", o.bband_min);
  o.bband_alone_maps = (int)opt("bband_alone_maps", o.bband_alone_maps);
  if (o.bband_mode == 2) o.bband_alone_maps = 6;
  o.pair_mode = (int)opt("pair", o.pair_mode);          // 1 (default): independent neighbouring rows in one launch; 0: never
  o.stem_pool = (int)opt("stem_pool", o.stem_pool);     // 1 (default): conv1's 3x3/2 max pool inside the conv_stem launch; 0: its own launch
  o.avg_fuse = (int)opt("avg_fuse", o.avg_fuse);        // a layer's global average inside its split-K launch: 2 (default) one batch at a time, 1 always; 0: global_avg_kernel
  o.dense_max_slabs = (int)opt("dense_max", o.dense_max_slabs);
  o.dense_mode = (int)opt("dense", o.dense_mode);       // arithmetic gather words for dense layers: 1 (default), 0 = always the header tables
  if (opt("alt_min", -1) >= 0) o.alt_min_blocks = o.alt_min_blocks_conc = (long)opt("alt_min", 0);       // smallest 128 x 128 grid that takes a layer's wide-tile alternative
  o.alt_min_blocks_conc = (long)opt("alt_min_conc", o.alt_min_blocks_conc);
  o.alt_rows = (unsigned long long)opt("alt_rows", 0); o.noalt_rows = (unsigned long long)opt("noalt_rows", 0);
  o.sk_rows = (unsigned long long)opt("sk_rows", 0); o.nosk_rows = (unsigned long long)opt("nosk_rows", 0);
  o.alt_narrow_blocks = (long)opt("alt_narrow", o.alt_narrow_blocks);   // a 128-row layer takes its 64-row alternative below this many 128 x 128 blocks
  o.alt_conc_mode = (int)opt("alt_conc", o.alt_conc_mode);
  o.pw_slabs = (int)opt("pw_slabs", o.pw_slabs);
  o.pw_minpix = (long)opt("pw_minpix", o.pw_minpix);
  o.pwk_mode = (int)opt("pwk", o.pwk_mode);
  o.q128_flags = (int)opt("q128", o.q128_flags);
  o.stem_pk_small = (int)opt("stem_pk_small", o.stem_pk_small);
  o.sk_kb = (int)opt("sk_kb", o.sk_kb); o.sk_kb_blocks = (int)opt("sk_kb_blocks", o.sk_kb_blocks); o.sk_kb_max = (int)opt("sk_kb_max", o.sk_kb_max); o.sk_kb_min = (int)opt("sk_kb_min", o.sk_kb_min);
  o.pwk_minpix = (long)opt("pwk_minpix", o.pwk_minpix);
  o.pwk_sk = (int)opt("pwk_sk", o.pwk_sk);
  o.pwk_max_slabs = (int)opt("pwk_slabs", o.pwk_max_slabs);
  o.pwk_rows = (unsigned long long)opt("pwk_rows", 0); o.nopwk_rows = (unsigned long long)opt("nopwk_rows", 0);
  conv_pwk_set_slots((int)opt("pwk_slots", 0));
  conv_pwk_set_pipe((int)opt("pwk_pipe", 1));
  conv_pwk_set_min_units((int)opt("pwk_units", 512));
  o.dbg = (long long*)(uintptr_t)(unsigned long long)opt("dbgptr", 0);
  o.dbg2 = (long long*)(uintptr_t)(unsigned long long)opt("dbgptr2", 0);
  o.dbg_layer = (int)opt("dbglayer", -1);
  opts = o;
  launch_plans.clear();
  // tensor lifetimes depend on which rows may share a launch (TF2_AMD_PAIR): re-plan what was planned (a caller's keep_all
  // workspace keeps being recognised by run / read_layer)
  std::vector<std::pair<int, int>> keys;
  for (const auto& kv : plans) keys.push_back(kv.first);
  plans.clear();
  for (const auto& k : keys) (void)plan(k.first, k.second != 0);
}

// Group launches (conv_bgroup.hip) keep eight blocks per image resident together, one block per CU: they need a device of at
// least 64 CUs -- and a STREAM that may use at least 64 of them (stream_cu_count below; Net::run asks per call).  No device
// (describing a plan on the CPU): assume the full chip.
static bool device_fits_group_launches() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true;
  static std::mutex mu;
  static std::map<int, int> n_cu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = n_cu.find(dev);
  if (it == n_cu.end()) {
    hipDeviceProp_t prop;
    const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    it = n_cu.emplace(dev, n).first;
  }
  return it->second >= 64;
}

// CUs the stream's launches may use (hipExtStreamCreateWithCUMask: tf2_amd/streams.py); an unmasked stream reports the device's.
static int stream_cu_count(hipStream_t s) {
  uint32_t mask[16] = {0};
  if (hipExtStreamGetCUMask(s, 16, mask) != hipSuccess) { (void)hipGetLastError(); return 1 << 20; }
  int n = 0;
  for (uint32_t w : mask) n += __builtin_popcount(w);
  return n > 0 ? n : 1 << 20;
}

// ---- launch plan: every kernel argument block of one step, resolved once per (batch, workspace, packed image) ----
// Net::run used to rebuild ~60 argument structs, scan the packed directory and read a dozen environment variables
// per call; at batch 1 (57 launches of a few microseconds) that host work was the step.  Now a step is a loop over
// prepared launches; only the image and logits pointers change between calls.
const LaunchPlan* Net::launch_plan(int batch, const WorkPlan* wp, void* ws, bool concurrent, bool allow_groups) {
  if (concurrent) allow_groups = false;
  for (const LaunchPlan& lp : launch_plans)
    if (lp.batch == batch && lp.wp == wp && lp.ws == ws && lp.packed_dev == packed_dev && lp.concurrent == (concurrent ? 1 : 0) &&
        lp.groups == (allow_groups ? 1 : 0)) return &lp;
  if (launch_plans.size() >= 64)                       // evict the oldest plan nobody is walking
    for (auto it = launch_plans.begin(); it != launch_plans.end(); ++it)
      if (it->walkers == 0) { launch_plans.erase(it); break; }
  LaunchPlan lp;
  lp.batch = batch; lp.wp = wp; lp.ws = ws; lp.packed_dev = packed_dev; lp.concurrent = concurrent ? 1 : 0; lp.groups = allow_groups ? 1 : 0;
  int8_t* base = (int8_t*)ws;
  const int nl = nd.n_layers;
  auto T = [&](int id) -> const TensorPlan& { return wp->tensors[id]; };
  const uint8_t* pk = packed_dev;
  const uint64_t zero_off = reinterpret_cast<const PackHeader*>(packed.data())->zero_off;
  auto fail = [&](const std::string& m) -> const LaunchPlan* { set_error(m); return nullptr; };

  const bool stem = stem_selected(batch);
  // input: quantise + (space-to-depth) + [x | xneg]
  {
    const tf2_layer_desc& L0 = layers[0];
    Launch st; st.kind = Launch::PREP; st.layer = -1;
    PrepArgs& pa = st.prep;
    pa.img = nullptr; pa.y = base + T(wp->input_tensor).offset;
    pa.B = batch; pa.C = nd.image_c; pa.H = nd.image_h; pa.W = nd.image_w;
    pa.OH = L0.H; pa.OW = L0.W; pa.y_cp = in_layout[0].Cp_in; pa.half = in_layout[0].half;
    pa.rewrite = im2col0 ? 2 : nd.conv1_rewrite; pa.q0 = q[0]; pa.src_is_q = 0; pa.xonly = stem ? 1 : 0;
    pa.im_stride = im_stride; pa.im_pad_h = im_pad_h; pa.im_pad_w = im_pad_w;
    if (!nd.conv1_rewrite && !im2col0 && (L0.H != nd.image_h || L0.W != nd.image_w || L0.C != nd.image_c)) return fail("layer 0 input does not match the image");
    lp.steps.push_back(st);
  }
  auto pool_step = [&](int l, const TensorPlan& ti, const int8_t* x, int H, int W) {
    const tf2_layer_desc L = exec_desc(l);
    const LayerExec& E = wp->exec[l];
    Launch st; st.kind = Launch::POOL; st.layer = l;
    PoolArgs& pa = st.pool;
    const TensorPlan& to = T(E.out_tensor);
    pa.x = x; pa.y = base + to.offset;
    pa.B = batch; pa.H = H; pa.W = W; pa.x_cp = ti.Cp; pa.x_off = 0;
    pa.PH = L.PH; pa.PW = L.PW; pa.y_cp = to.Cp; pa.y_off = E.out_off;
    pa.S = L.pool_S; pa.st = L.pool_st; pa.pad = L.pool_pad; pa.C16 = round_up(L.N, 16) / 16;
    lp.steps.push_back(st);
  };
  // the global average inside the last expand's split-K launch (conv_mfma_sk AVG: 1024 blocks of 64 x 64 tiles, one image per pixel
  // tile) pays one batch at a time (a launch and the 7 x 7 map's round trip less); with batches in flight that launch costs 13 us of the
  // step against 2.6 us for the same row on 208 blocks of 128 x 128 tiles + a 512-block average (round 6, profiles/r06_experiments.txt
  // items 1-2: 94.8 -> 96.5 k img/s) -- avg_fuse = 2 (default): one batch at a time only
  const bool avg_fuse_now = opts.avg_fuse == 1 || (opts.avg_fuse == 2 && !concurrent);
  // argument block + kernel selection of one conv layer
  auto make_conv0 = [&](int l, Launch& st, bool allow_alt) -> bool {
    const tf2_layer_desc L = exec_desc(l);
    const LayerExec& E = wp->exec[l];
    const PackLayer* pl = pack_layer(l);
    // the wide-tile alternative (128-row tiles, weight_pack.cpp) where its grid still fills the chip: fewer operand bytes and
    // instructions per MAC; small batches keep the 64-row tiles (more blocks, split-K).  With several batches in flight the
    // other batches' kernels fill the chip, so the wide form pays from a much smaller grid on.
    // The reverse on the 28x28 maps: their 128-row layers have a 64-row alternative (more blocks, split-K) for grids of a few
    // blocks (batch 1-2).
    const PackLayer* pa = (allow_alt && !(L.endpool && avg_fuse_now && pl->TM == 64)) ? pack_layer_alt(l) : nullptr;   // (the fused global average runs on the 64-row tiles)
    if (pa) {
      const long blocks128 = ((long)batch * L.OH * L.OW + 127) / 128 * (pa->Np / 128);
      if (pa->TM == 128) { if (blocks128 >= (concurrent ? opts.alt_min_blocks_conc : opts.alt_min_blocks)) pl = pa; }
      else if (blocks128 < opts.alt_narrow_blocks) pl = pa;
      // (test-only, per-row A/B of the in-flight plan: rows forced onto / kept off their alternative tile height)
      if (l < 64 && ((opts.alt_rows >> l) & 1)) pl = pa;
      if (l < 64 && ((opts.noalt_rows >> l) & 1)) pl = pack_layer(l);
    }
    st.kind = Launch::CONV; st.layer = l;
    ConvArgs& ca = st.conv;
    const TensorPlan& ti = T(E.in_tensor); const TensorPlan& tc = T(E.conv_tensor);
    ca.x = base + ti.offset; ca.y = base + tc.offset;
    ca.w = (const int8_t*)(pk + pl->off_w); ca.w2 = (const int8_t*)(pk + pl->off_w2);
    ca.bias = (const int32_t*)(pk + pl->off_bias); ca.alpha = (const int32_t*)(pk + pl->off_alpha);
    ca.beta = (const int32_t*)(pk + pl->off_beta);
    ca.zero = (const int8_t*)(pk + (pl->off_pad ? pl->off_pad : zero_off)); ca.max_ent = pl->max_ent;
    ca.dual = pl->dual;
    set_fast_div((uint32_t)pl->n_mtiles, &ca.mt_m, &ca.mt_s);
    // weight-tile addressing (tf2_internal.h ConvArgs): own storage, or the main entry's tiles of the other height
    {
      const int wins = pl->dual ? 2 : 1;
      const int sTM = pl->w_share ? pl->w_main_TM : pl->TM;            // rows of a storage tile
      ca.w_ent_bytes = wins * sTM * 64; ca.w_win_stride = sTM * 64; ca.w_half_stride = 4096; ca.w_sub_step = 0; ca.e_mt_shl = 0; ca.e_mt_shr = 0;
      if (pl->w_share && sTM == 2 * pl->TM) { ca.w_sub_step = 4096; ca.e_mt_shr = 1; }                       // halves of 128-row tiles
      if (pl->w_share && 2 * sTM == pl->TM) {                                                               // pairs of 64-row tiles
        const int32_t* hd0 = reinterpret_cast<const int32_t*>(packed.data() + pl->off_dir);
        const int nent = hd0[pl->n_phases] - hd0[0];
        ca.w_half_stride = nent * ca.w_ent_bytes; ca.e_mt_shl = 1;
      }
    }
    bool dense = false;
    if (pl->kind == KIND_MFMA) {
      ca.hdr = (const int32_t*)(pk + pl->off_hdr); ca.hdr_bytes = (int32_t)pl->hdr_bytes;
      // every m-tile's entry list is slabs 0..nslab-1 (dense weights)?  From the host copy of the image.
      const int32_t* hd = reinterpret_cast<const int32_t*>(packed.data() + pl->off_dir);
      dense = true;
      for (int mt = 0; mt < pl->n_mtiles && dense; mt++)
        dense = hd[(size_t)mt * (pl->n_phases + 1) + pl->n_phases] - hd[(size_t)mt * (pl->n_phases + 1)] == pl->nslab;
      ca.ent0 = hd[pl->n_phases] - hd[0];
      // arithmetic gather (tf2_internal.h ConvArgs::dense): no header read in front of the first activation DMAs
      const int taps_l = L.k * L.k;
      ca.cslabs = pl->Cp_in / 64;
      // (long slab lists on large maps pay more for the per-step arithmetic than the shorter prologue saves: VGG16 -7 % with every
      //  layer dense; layers of more than dense_max_slabs slabs whose grid runs in several rounds keep the header tables)
      const long blocks_d = ((long)batch * L.OH * L.OW + 127) / 128 * std::max(1, pl->Np / 128);
      const bool dense_pays = pl->nslab <= opts.dense_max_slabs || blocks_d <= 512;
      if (opts.dense_mode && dense && dense_pays && (pl->n_phases == 1 || pl->dual) && pl->Cp_in % 64 == 0 && pl->nslab == taps_l * ca.cslabs && L.k <= 15) {
        ca.dense = 1;
        set_fast_div((uint32_t)ca.cslabs, &ca.cs_m, &ca.cs_s); set_fast_div((uint32_t)L.k, &ca.kk_m, &ca.kk_s);
      }
    }
    if (opts.dbg2 && opts.dbg_layer == l) ca.dbg2 = opts.dbg2;
    if (opts.dbg) ca.dbg = opts.dbg + (size_t)l * 16;
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
    g.dbl_out = pl->off_dbl != 0;
    if (g.has_res) {
      const TensorPlan& tr = T(E.res_tensor);
      ca.res = base + tr.offset; g.res_cp = tr.Cp; g.res_off = E.res_off;
    }
    g.flags = opts.flags;
    st.TM = pl->TM; st.signed_in = pl->signed_in; st.mul24 = pl->max_shift <= 22;
    if (pl->kind == KIND_MFMA) {
      // small grid + long slab list: the four (or eight) waves of a block split K (conv_mfma_sk.hip)
      const long blocks64 = (long)((g.n_pix + 63) / 64) * pl->n_mtiles;
      const bool sk = pl->TM == 64 && opts.sk_mode != 2 &&
                      (opts.sk_mode == 1 || (blocks64 <= 512 && (long)pl->n_entries * (pl->dual ? 2 : 1) >= 16L * pl->n_mtiles));
      st.sel = sk ? Launch::SEL_SK : Launch::SEL_MFMA2;
      if (pl->TM == 64 && l < 64 && ((opts.sk_rows >> l) & 1)) st.sel = Launch::SEL_SK;              // (test-only per-row switches)
      if (l < 64 && ((opts.nosk_rows >> l) & 1)) st.sel = Launch::SEL_MFMA2;
      st.shape = (int)(concurrent ? opts.sk_s3_blocks_conc : opts.sk_s3_blocks);      // SEL_SK: largest grid on three ring stages
      // the layer's global average inside the launch (conv_mfma_sk AVG): 64-row tiles, one image per pixel tile
      if (avg_fuse_now && L.endpool && !L.pool_en && pl->TM == 64 && g.OHW <= 64 && (g.pad_h | g.pad_w) == 0 && L.concat < 0 &&
          E.conv_tensor != E.out_tensor && !g.dbl_out && g.n_pix == batch * g.OHW) {
        const TensorPlan& to = T(E.out_tensor);
        ca.y = base + to.offset; g.y_cp = to.Cp; g.y_off = E.out_off; g.avg_mult = L.endpool_mult;
        st.sel = Launch::SEL_SK; st.avg_fused = 1;
      } else
      // register-resident pointwise kernel (conv_pw.hip) where the layer qualifies and no other kernel is forced
      if (opts.pw_mode && L.k == 1 && opts.sk_mode != 1 && !pl->w_share && conv_pw_eligible(ca, pl->TM, pl->nslab, L.k, dense ? 1 : 0, opts.pw_slabs, opts.pw_minpix)) st.sel = Launch::SEL_PW;
      // short-K pointwise rows on persistent four-wave blocks (conv_pwk.hip, round 6)
      if (opts.pwk_mode && (concurrent || opts.pwk_mode == 2) && !st.avg_fused && (st.sel == Launch::SEL_MFMA2 || (st.sel == Launch::SEL_SK && opts.pwk_sk)) &&
          L.k == 1 && !pl->w_share && L.concat < 0 && !(l < 64 && ((opts.nopwk_rows >> l) & 1))) {
        const bool forced = l < 64 && ((opts.pwk_rows >> l) & 1);
        if ((forced || pl->nslab <= opts.pwk_max_slabs) && conv_pwk_eligible(ca, pl->TM, L.k, dense ? 1 : 0, opts.pwk_minpix, forced)) st.sel = Launch::SEL_PWK;
      }
    } else if (pl->kind == KIND_SHIFT) {
      st.sel = Launch::SEL_SHIFT; st.shape = pl->fast;      // fast on a shift layer: packed 4-bit filters
    } else {
      set_error("layer " + std::to_string(l) + " has no packed kernel"); return false;
    }
    return true;
  };
  // (conv_pwk reads a layer's OWN weight tiles: a 1x1 row whose wide-tile alternative shares the main entry's tiles is tried again on
  //  the main entry)
  auto make_conv = [&](int l, Launch& st, bool allow_alt) -> bool {
    if (!make_conv0(l, st, allow_alt)) return false;
    if (allow_alt && opts.pwk_mode && (concurrent || opts.pwk_mode == 2) && exec_desc(l).k == 1 && !st.avg_fused &&
        (st.sel == Launch::SEL_MFMA2 || (st.sel == Launch::SEL_SK && opts.pwk_sk))) {
      Launch s2;
      if (make_conv0(l, s2, false) && s2.sel == Launch::SEL_PWK) st = s2;
    }
    return true;
  };
  std::vector<char> fused_done(nl, 0), pair_done(nl, 0);
  int bg_used = 0;                                           // group launches so far (each has its own counters)
  const bool groups_fit = allow_groups && device_fits_group_launches();
  bool stem_pool_fused = false;
  const bool profiling_pairs_off = false;
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc L = exec_desc(l);
    const LayerExec& E = wp->exec[l];
    if (pack_layer(l)->merged_into >= 0) continue;           // computed by the merged launch of the row in front of it
    if (L.ipool == 2) {                  // L2Norm row
      const PackLayer* pl2 = pack_layer(l);
      const TensorPlan& ti = T(E.in_tensor); const TensorPlan& to = T(E.out_tensor);
      Launch st; st.kind = Launch::L2N; st.layer = l;
      L2NormArgs& a = st.l2n;
      a.x = base + ti.offset; a.y = base + to.offset + E.out_off;
      a.a = (const double*)(pk + pl2->off_w); a.b = (const double*)(pk + pl2->off_w2); a.e = (const int32_t*)(pk + pl2->off_bias);
      a.n_pix = batch * ti.H * ti.W; a.C = round_up(L.N, 16); a.x_cp = ti.Cp; a.y_cp = to.Cp; a.qs = pl2->max_shift;
      lp.steps.push_back(st);
      continue;
    }
    if (L.ipool) {
      const TensorPlan& ti = T(E.in_tensor);
      pool_step(l, ti, base + ti.offset, ti.H, ti.W);
      continue;
    }
    const PackLayer* pl = pack_layer(l);
    if (pl->fused_into >= 0 && fused_done[l]) continue;      // computed by the launch of layer pl->fused_into (conv_bneck.hip)
    if (pair_done[l]) continue;                              // computed by the pair launch of layer l - 1 (or a group launch)
    // the first bottleneck of the 56 x 56 stage (shortcut | reduce, 3x3, expand) as ONE launch of independent row bands at two blocks per
    // CU (conv_bfirst.hip, round 6): the form for batches in flight (bfirst=2: one batch at a time as well, instead of the group launch)
    if (opts.bfirst_mode && (concurrent || opts.bfirst_mode == 2) && bgroup_first_at(l) && batch >= opts.bfirst_min) {
      Launch ss, s0, s1, s2;
      if (!make_conv(l, ss, false) || !make_conv(l + 1, s0, false) || !make_conv(l + 2, s1, false) || !make_conv(l + 3, s2, false)) return nullptr;
      if (ss.conv.dense && s0.conv.dense && s1.conv.dense && s2.conv.dense && s0.TM == 64 && s1.TM == 64 && s2.TM == 64 && !s1.conv.dual &&
          s0.conv.dual == s2.conv.dual && ss.conv.dual == s0.conv.dual) {
        Launch st; st.kind = Launch::CONV; st.sel = Launch::SEL_BFIRST; st.layer = l;
        BGroupArgs& f = st.bgroup;
        const ConvArgs& cs = ss.conv; const ConvArgs& c0 = s0.conv; const ConvArgs& c1 = s1.conv; const ConvArgs& c2 = s2.conv;
        f.x = c0.x; f.mid1 = c0.y; f.mid2 = c1.y; f.y = c2.y; f.res = nullptr;
        f.w1 = c0.w; f.w2 = c1.w; f.w3 = c2.w; f.hdr1 = c0.hdr; f.hdr2 = c1.hdr; f.hdr3 = c2.hdr;
        f.hdr1_bytes = c0.hdr_bytes; f.hdr2_bytes = c1.hdr_bytes; f.hdr3_bytes = c2.hdr_bytes;
        f.tm1 = s0.TM; f.tm2 = s1.TM; f.tm3 = s2.TM;
        f.zero = (const int8_t*)(pk + zero_off); f.zero2 = c1.zero;
        f.epoch = nullptr; f.ctr = nullptr; f.img0 = 0;
        f.B = batch;
        f.relu1 = c0.g.relu; f.relu2 = c1.g.relu; f.relu3 = c2.g.relu; f.add_relu = c2.g.add_relu; f.has_res = 1;
        f.fast1 = c0.g.fast; f.fast2 = c1.g.fast; f.fast3 = c2.g.fast;
        f.dbl1 = c0.g.dbl_out; f.dbl2 = c1.g.dbl_out; f.dbl3 = 0;
        f.dual1 = c0.dual; f.dual2 = 0; f.dual3 = c2.dual;
        f.avg_mult = 0; f.res_cp = 0; f.res_off = 0;
        f.y_cp = c2.g.y_cp; f.y_off = c2.g.y_off;
        f.ws = cs.w; f.hdrs = cs.hdr; f.hdrs_bytes = cs.hdr_bytes; f.tms = ss.TM; f.relu_s = cs.g.relu; f.fast_s = cs.g.fast;
        f.ys = cs.y; f.ys_cp = cs.g.y_cp; f.keep_s = wp->keep_all ? 1 : 0;
        f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
        pair_done[l + 1] = 1; pair_done[l + 2] = 1; pair_done[l + 3] = 1;
        lp.steps.push_back(st);
        continue;
      }
    }
    // ... the same rows as a group launch (conv_bgroup56f_kernel), one batch at a time
    if (opts.bgroup_mode && !concurrent && wp->ctrl_bytes && groups_fit && bgroup_first_at(l) && batch >= opts.bgroup_min56f &&
        256 + (size_t)(bg_used + 1) * ((batch + 7) / 8 * 8) * 128 <= wp->ctrl_bytes) {
      Launch ss, s0, s1, s2;
      if (!make_conv(l, ss, false) || !make_conv(l + 1, s0, false) || !make_conv(l + 2, s1, false) || !make_conv(l + 3, s2, false)) return nullptr;
      if (ss.conv.dense && s0.conv.dense && s1.conv.dense && s2.conv.dense) {
        Launch st; st.kind = Launch::CONV; st.sel = Launch::SEL_BGROUPF; st.layer = l;
        BGroupArgs& f = st.bgroup;
        const ConvArgs& cs = ss.conv; const ConvArgs& c0 = s0.conv; const ConvArgs& c1 = s1.conv; const ConvArgs& c2 = s2.conv;
        f.x = c0.x; f.mid1 = c0.y; f.mid2 = c1.y; f.y = c2.y; f.res = nullptr;
        f.w1 = c0.w; f.w2 = c1.w; f.w3 = c2.w; f.hdr1 = c0.hdr; f.hdr2 = c1.hdr; f.hdr3 = c2.hdr;
        f.hdr1_bytes = c0.hdr_bytes; f.hdr2_bytes = c1.hdr_bytes; f.hdr3_bytes = c2.hdr_bytes;
        f.tm1 = s0.TM; f.tm2 = s1.TM; f.tm3 = s2.TM;
        f.zero = (const int8_t*)(pk + zero_off); f.zero2 = c1.zero;
        f.epoch = reinterpret_cast<const unsigned*>(base + wp->ctrl_off);
        f.ctr = reinterpret_cast<unsigned*>(base + wp->ctrl_off + 256) + (size_t)bg_used * ((batch + 7) / 8 * 8) * 32;
        f.B = batch;
        f.relu1 = c0.g.relu; f.relu2 = c1.g.relu; f.relu3 = c2.g.relu; f.add_relu = c2.g.add_relu; f.has_res = 1;
        f.fast1 = c0.g.fast; f.fast2 = c1.g.fast; f.fast3 = c2.g.fast;
        f.dbl1 = c0.g.dbl_out; f.dbl2 = c1.g.dbl_out; f.dbl3 = 0;
        f.dual1 = c0.dual; f.dual2 = 0; f.dual3 = c2.dual;
        f.y_cp = c2.g.y_cp; f.y_off = c2.g.y_off;
        f.ws = cs.w; f.hdrs = cs.hdr; f.hdrs_bytes = cs.hdr_bytes; f.tms = ss.TM; f.relu_s = cs.g.relu; f.fast_s = cs.g.fast;
        f.ys = cs.y; f.ys_cp = cs.g.y_cp; f.keep_s = wp->keep_all ? 1 : 0;
        f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
        lp.steps[0].prep.epoch_ptr = reinterpret_cast<unsigned*>(base + wp->ctrl_off);
        lp.steps[0].prep.n_flag_words = (int32_t)((wp->ctrl_bytes - 256) / 4);
        lp.steps[0].prep.bg_poll_limit = (int32_t)opts.bg_poll_limit; lp.steps[0].prep.bg_withhold = (int32_t)opts.bg_withhold;
        bg_used++; lp.n_groups++;
        pair_done[l + 1] = 1; pair_done[l + 2] = 1; pair_done[l + 3] = 1;
        lp.steps.push_back(st);
        continue;
      }
    }
    // a fire module (squeeze + the merged expands) as ONE launch of independent row bands (conv_fire.hip)
    if (opts.fire_mode && fire_at(l) && (opts.fire_mode == 1 || L.W >= 28) &&
        (!layers[l + 1].pool_en || opts.fire_pool >= 2 || (opts.fire_pool == 1 && L.W >= 56))) {
      Launch s0, s1;
      if (!make_conv(l, s0, false) || !make_conv(l + 1, s1, false)) return nullptr;
      const PackLayer* p1 = pack_layer(l + 1);
      if (s0.conv.dense && s0.TM == 64 && s1.TM == p1->TM) {
        Launch st; st.kind = Launch::CONV; st.sel = Launch::SEL_FIRE; st.layer = l;
        FireArgs& f = st.fire;
        const ConvArgs& c0 = s0.conv; const ConvArgs& c1 = s1.conv;
        f.x = c0.x; f.mid = c0.y; f.y = c1.y; f.w1 = c0.w; f.w2 = c1.w; f.hdr1 = c0.hdr; f.hdr2 = c1.hdr; f.hdr2_bytes = c1.hdr_bytes;
        f.ent2 = reinterpret_cast<const int32_t*>(pk + p1->off_entries); f.dir2 = reinterpret_cast<const int32_t*>(pk + p1->off_dir); f.n_ent2 = (int32_t)p1->n_entries;
        f.zero = (const int8_t*)(pk + zero_off); f.zero2 = c1.zero;
        f.tm1 = s0.TM; f.tm2 = s1.TM; f.B = batch; f.H = L.H; f.W = L.W; f.Cin = L.C; f.Sp = round_up(L.N, 16); f.N2 = p1->Np;
        f.relu1 = c0.g.relu; f.relu2 = c1.g.relu; f.fast1 = c0.g.fast; f.fast2 = c1.g.fast; f.dbl1 = c0.g.dbl_out; f.dual1 = c0.dual;
        f.keep_mid = wp->keep_all ? 1 : 0; f.mid_cp = c0.g.y_cp; f.y_cp = c1.g.y_cp; f.y_off = c1.g.y_off; f.y_nvalid = c1.g.y_nvalid;
        f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
        // the pool behind the expands inside the launch where its form fits (3x3 / 2 / pad 0 in ceil mode behind a ReLU) -- one batch at a
        // time (fire_pool=3, the default; 4: always): its blocks hold 100-108 KB of LDS, one per CU, and with batches in flight that costs
        // more (317 k against 334-345 k img/s) than the pool launch it saves; alone it is 25.3 us against 19 + 9.2 (fire3), 12.6 against
        // 16.2 + 6.7 (fire5): profiles/r05_experiments.txt item 22
        const tf2_layer_desc L1 = exec_desc(l + 1);
        f.pool = 0;
        if (L1.pool_en && (opts.fire_pool == 4 || (opts.fire_pool == 3 && !concurrent)) && L1.pool_S == 3 && L1.pool_st == 2 && L1.pool_pad == 0 && c1.g.relu && !c1.g.dbl_out &&
            L1.PH == (L.H - 2) / 2 + 1 && L1.PW == L1.PH && c1.g.y_nvalid == f.N2 &&
            conv_fire_geometry(f.H, f.W, f.Cin, f.Sp, f.N2, f.tm1, f.tm2, f.dual1, 1, nullptr, nullptr)) {
          const TensorPlan& to1 = T(wp->exec[l + 1].out_tensor);
          f.pool = 1; f.PH = L1.PH; f.PW = L1.PW; f.yp = base + to1.offset; f.yp_cp = to1.Cp; f.yp_off = wp->exec[l + 1].out_off;
        }
        if (conv_fire_geometry(f.H, f.W, f.Cin, f.Sp, f.N2, f.tm1, f.tm2, f.dual1, f.pool, &f, nullptr)) {
          pair_done[l + 1] = 1;
          lp.steps.push_back(st);
          if (L1.pool_en && !f.pool) { const TensorPlan& tc1 = T(wp->exec[l + 1].conv_tensor); pool_step(l + 1, tc1, base + tc1.offset, L1.OH, L1.OW); }
          continue;
        }
      }
    }
    // an identity bottleneck as ONE launch of independent row bands (conv_bband.hip): no exchange between blocks, so it may share
    // the chip with anything -- the form for batches in flight (TF2_AMD_BBAND=2: one batch at a time as well, instead of the groups)
    {
      const int band_rows = concurrent ? opts.bband_rows : opts.bband_rows_alone;
      if (opts.bband_mode && (concurrent || (opts.bband_alone_maps & (L.H >= 28 ? 2 : 4))) && batch >= opts.bband_min && bband_at(l, band_rows)) {
        Launch s0, s1, s2;
        if (!make_conv(l, s0, false) || !make_conv(l + 1, s1, false) || !make_conv(l + 2, s2, false)) return nullptr;
        if (s0.conv.dense && s1.conv.dense && s2.conv.dense) {
          Launch st; st.kind = Launch::CONV; st.sel = Launch::SEL_BBAND; st.layer = l;
          BBandArgs& f = st.bband;
          const ConvArgs& c0 = s0.conv; const ConvArgs& c1 = s1.conv; const ConvArgs& c2 = s2.conv;
          f.x = c0.x; f.mid1 = c0.y; f.mid2 = c1.y; f.y = c2.y; f.res = c2.res;
          f.w1 = c0.w; f.w2 = c1.w; f.w3 = c2.w; f.hdr1 = c0.hdr; f.hdr2 = c1.hdr; f.hdr3 = c2.hdr;
          f.hdr1_bytes = c0.hdr_bytes; f.hdr2_bytes = c1.hdr_bytes; f.hdr3_bytes = c2.hdr_bytes;
          f.tm1 = s0.TM; f.tm2 = s1.TM; f.tm3 = s2.TM;
          f.zero = (const int8_t*)(pk + zero_off); f.zero2 = c1.zero;
          f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
          f.B = batch; f.H = L.H; f.W = L.W; f.R = std::min(conv_bband_pick_rows(L.W, L.N, c0.dual, c1.dual, band_rows, opts.bband_rows_dd), L.H);
          f.tiles_per_img = (L.H + f.R - 1) / f.R;
          f.relu1 = c0.g.relu; f.relu2 = c1.g.relu; f.relu3 = c2.g.relu; f.add_relu = c2.g.add_relu; f.has_res = c2.g.has_res;
          f.keep_mid = wp->keep_all ? 1 : 0;
#ifdef TF2_PROBES
          if (opts.flags & 16384) f.probe = 1;              // (probe, conv_bband.hip: one column tile fewer in phases 1-2)
#endif
          f.fast1 = c0.g.fast; f.fast2 = c1.g.fast; f.fast3 = c2.g.fast;
          f.dbl1 = c0.g.dbl_out; f.dbl2 = c1.g.dbl_out; f.dbl3 = c2.g.dbl_out;
          f.dual1 = c0.dual; f.dual2 = c1.dual;
          f.res_cp = c2.g.res_cp; f.res_off = c2.g.res_off; f.y_cp = c2.g.y_cp; f.y_off = c2.g.y_off;
          st.bg_c = L.C; st.bg_m = L.N;
          pair_done[l + 1] = 1; pair_done[l + 2] = 1;
          lp.steps.push_back(st);
          continue;
        }
      }
    }
    // an identity bottleneck of a small map as ONE launch, eight blocks per image (one batch at a time: two such kernels
    // sharing CUs could hold each other's slots while their groups wait)
    if (opts.bgroup_mode && !concurrent && wp->ctrl_bytes && groups_fit && bgroup_at(l) && batch >= (L.H == 7 ? opts.bgroup_min7 : L.H == 28 ? opts.bgroup_min28 : opts.bgroup_min14) && 256 + (size_t)(bg_used + 1) * ((batch + 7) / 8 * 8) * 128 <= wp->ctrl_bytes) {
      Launch s0, s1, s2;
      if (!make_conv(l, s0, false) || !make_conv(l + 1, s1, false) || !make_conv(l + 2, s2, false)) return nullptr;
      if (s0.conv.dense && s1.conv.dense && s2.conv.dense && (!layers[l + 2].endpool || s2.avg_fused)) {
        Launch st; st.kind = Launch::CONV; st.sel = Launch::SEL_BGROUP; st.layer = l;
        BGroupArgs& f = st.bgroup;
        const ConvArgs& c0 = s0.conv; const ConvArgs& c1 = s1.conv; const ConvArgs& c2 = s2.conv;
        f.x = c0.x; f.mid1 = c0.y; f.mid2 = c1.y; f.y = c2.y; f.res = c2.res;
        f.w1 = c0.w; f.w2 = c1.w; f.w3 = c2.w; f.hdr1 = c0.hdr; f.hdr2 = c1.hdr; f.hdr3 = c2.hdr;
        f.hdr1_bytes = c0.hdr_bytes; f.hdr2_bytes = c1.hdr_bytes; f.hdr3_bytes = c2.hdr_bytes;
        f.tm1 = s0.TM; f.tm2 = s1.TM; f.tm3 = s2.TM;
        f.zero = (const int8_t*)(pk + zero_off); f.zero2 = c1.zero;
        f.epoch = reinterpret_cast<const unsigned*>(base + wp->ctrl_off);
        f.ctr = reinterpret_cast<unsigned*>(base + wp->ctrl_off + 256) + (size_t)bg_used * ((batch + 7) / 8 * 8) * 32;
        f.B = batch;
        f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
        f.relu1 = c0.g.relu; f.relu2 = c1.g.relu; f.relu3 = c2.g.relu; f.add_relu = c2.g.add_relu; f.has_res = c2.g.has_res;
        f.fast1 = c0.g.fast; f.fast2 = c1.g.fast; f.fast3 = c2.g.fast;
        f.dbl1 = c0.g.dbl_out; f.dbl2 = c1.g.dbl_out; f.dbl3 = c2.g.dbl_out;
        f.dual1 = c0.dual; f.dual2 = c1.dual; f.dual3 = c2.dual; f.avg_mult = c2.g.avg_mult;
        f.res_cp = c2.g.res_cp; f.res_off = c2.g.res_off; f.y_cp = c2.g.y_cp; f.y_off = c2.g.y_off;
        st.bg_hw = L.H; st.bg_c = L.C; st.bg_m = L.N;
        // the step's first kernel (input preparation) advances the step counter
        lp.steps[0].prep.epoch_ptr = reinterpret_cast<unsigned*>(base + wp->ctrl_off);
        lp.steps[0].prep.n_flag_words = (int32_t)((wp->ctrl_bytes - 256) / 4);
        lp.steps[0].prep.bg_poll_limit = (int32_t)opts.bg_poll_limit; lp.steps[0].prep.bg_withhold = (int32_t)opts.bg_withhold;
        bg_used++; lp.n_groups++;
        pair_done[l + 1] = 1; pair_done[l + 2] = 1;
        // the 14 x 14 stage's identity bottlenecks follow one another: the groups of the previous launch carry on with this one
        // (its roll-call row doubles as the meeting "input complete")
        if (opts.bgroup_chain > 1 && !lp.steps.empty() && !f.dbg) {
          Launch& pv = lp.steps.back();
          const int pn = pv.bg_chain.empty() ? 1 : (int)pv.bg_chain.size();
          if (pv.kind == Launch::CONV && pv.sel == Launch::SEL_BGROUP && pv.bg_hw == L.H && pv.layer + 3 * pn == l && f.dual1 == pv.bgroup.dual1 && f.dual2 == pv.bgroup.dual2 && !pv.bg_chain_last().avg_mult && pn < std::min(kBgMaxChain, opts.bgroup_chain) &&
              !pv.bgroup.dbg && f.x == pv.bg_chain_last().y && f.has_res && f.res == f.x && f.res_off == 0 && pv.bg_chain_last().y_off == 0 &&
              f.res_cp == pv.bg_chain_last().y_cp) {
            if (pv.bg_chain.empty()) pv.bg_chain.push_back(pv.bgroup);
            pv.bg_chain.push_back(f);
            continue;
          }
        }
        lp.steps.push_back(st);
        continue;
      }
    }
    Launch st;
    // a fused launch needs enough row bands to fill the chip (one block per band): small batches run the two layers on their own
    const int bn_TN = pl->TM == 64 ? 256 : 128;
    const int bn_R = pl->fuse_next > 0 ? std::min(bn_TN / L.W, L.H) : 1;
    // (128-channel pairs: 42.6 against 46.8 us at batch 64 one batch at a time, but 0.62 against 0.51 us per further image --
    //  with batches in flight the two separate launches win)
    const bool fuse_now = pl->fuse_next > 0 && (long)batch * ((L.H + bn_R - 1) / bn_R) >= opts.bneck_min_blocks &&
                          !(concurrent && pl->TM == 128 && opts.bneck_min_blocks > 1);
    if (!make_conv(l, st, !fuse_now)) return nullptr;     // the fused launch needs the pair's own (one m-tile) entries
    if (!fuse_now && (opts.fc_mode || pack_layer(l)->fc4) && fc_at(l, batch) && wp->scratch_bytes) {
      Launch sc;
      const PackLayer* pm = pack_layer(l);
      if (make_conv(l, sc, false) && pm->TM == sc.TM && conv_fc_scratch_bytes(pm->Np, pm->nslab, pm->dual, batch) <= wp->scratch_bytes) {
        const ConvArgs& c = sc.conv;
        FcArgs& f = sc.fc;
        f.x = c.x; f.y = c.y; f.w = c.w; f.hdr = c.hdr; f.hdr_bytes = c.hdr_bytes; f.tm = sc.TM;
        f.part = reinterpret_cast<int32_t*>(base + wp->scratch_off);
        f.B = batch; f.nslab = pm->nslab; f.K = pm->nslab * 64; f.Np = pm->Np;
        f.ksplit = conv_fc_pick_ksplit(pm->Np, pm->nslab); f.slabs_per_split = (pm->nslab + f.ksplit - 1) / f.ksplit;
        f.dual = c.dual; f.relu = c.g.relu; f.fast = c.g.fast; f.dbl = c.g.dbl_out;
        f.fc4 = pm->fc4; f.n_cls = pm->n_cls; f.chunks = (batch + 31) / 32;
        f.lut = pk + pm->off_lut; f.cls = pk + pm->off_cls;
        f.y_cp = c.g.y_cp; f.y_off = c.g.y_off; f.y_nvalid = c.g.y_nvalid;
        sc.sel = Launch::SEL_FC; sc.avg_fused = 0;
        st = sc;
      }
    }
    if (pack_layer(l)->fc4 && st.sel != Launch::SEL_FC) return fail("layer " + std::to_string(l) + " is packed as 4-bit codes (fc4) but conv_fc cannot take it");
    bool c3_pool_fused = false;
    if (!fuse_now && st.sel != Launch::SEL_FC && opts.c3_mode && c3_at(l)) {
      int th = 0, tw = 0;
      conv_c3_pick_tile(L.H, L.W, &th, &tw);
      // the layer's 2x2 / stride 2 / pad 0 max pool inside the launch (conv_c3.hip POOL: tiles of TH x 32 pixels): ReLU layers whose
      // map the 32-column tiling fits; the conv map is then neither written nor read back, and the pool launch is gone
      const bool pool_in = opts.c3_pool && L.pool_en && L.relu && L.pool_S == 2 && L.pool_st == 2 && L.pool_pad == 0 && E.conv_tensor != E.out_tensor &&
                           L.PH == (L.OH + 1) / 2 && L.PW == (L.OW + 1) / 2 && conv_c3_pick_tile_pool(L.H, L.W, &th, &tw);
      const int tiles_x = (L.W + tw - 1) / tw, tiles = tiles_x * ((L.H + th - 1) / th);
      // 128 output channels per block (two waves per SIMD, the accumulators of four column tiles per wave) unless the layer has
      // 64-row tiles only, or is a one-window layer whose 128-channel grid would leave half the chip idle (VGG16's 14 x 14 maps at
      // batch 32: 29 / 33 us against 33 / 37; two-window rows spill at the 128 registers of the 64-channel form)
      const PackLayer* pm = pack_layer(l);
      int tmk = pm->Np % 128 == 0 ? 128 : 64;
      // one-window layers of 256+ channels whose 256-channel grid still covers the chip: two row tiles per wave (a B fragment feeds
      // two MFMAs: half the LDS reads per MFMA, half the blocks' prologues)
      if (tmk == 128 && !pm->dual && pm->Np % 256 == 0 && (long)batch * tiles * (pm->Np / 256) >= opts.c3_min256) tmk = 256;
      else if (tmk == 128 && !pm->dual && (long)batch * tiles * (pm->Np / 128) < 256) tmk = 64;
      if (opts.c3_mode == 2) tmk = 64; else if (opts.c3_mode == 3 && pm->Np % 128 == 0) tmk = 128;       // (experiments)
      Launch sc;
      if ((long)batch * tiles * (pl->Np / tmk) >= opts.c3_min_blocks && make_conv(l, sc, false) && pack_layer(l)->TM == sc.TM) {
        const ConvArgs& c = sc.conv;
        C3Args& f = sc.c3;
        f.x = c.x; f.y = c.y; f.w = c.w; f.hdr = c.hdr; f.hdr_bytes = c.hdr_bytes; f.zero2 = c.zero; f.tm = sc.TM; f.tmk = tmk;
        f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
        f.B = batch; f.H = L.H; f.W = L.W; f.C = L.C; f.M = pl->Np; f.x_cp = c.g.Cp_in;
        f.TH = th; f.TW = tw; f.tiles_x = tiles_x; f.tiles_per_img = tiles;
        set_fast_div((uint32_t)tw, &f.tw_m, &f.tw_s); set_fast_div((uint32_t)(tw + 2), &f.hc_m, &f.hc_s);
        set_fast_div((uint32_t)tiles_x, &f.tx_m, &f.tx_s); set_fast_div((uint32_t)tiles, &f.tpi_m, &f.tpi_s);
        f.relu = c.g.relu; f.fast = c.g.fast; f.dbl = c.g.dbl_out; f.dual = c.dual;
        f.y_cp = c.g.y_cp; f.y_off = c.g.y_off; f.y_nvalid = c.g.y_nvalid;
        if (pool_in) {
          const TensorPlan& to = T(E.out_tensor);
          f.pool = 1; f.PH = L.PH; f.PW = L.PW; f.y = base + to.offset; f.y_cp = to.Cp; f.y_off = E.out_off;
          c3_pool_fused = true;
        }
        f.w9 = conv_c3_takes_w9(f, opts.c3_w9) ? 1 : 0;
        sc.sel = Launch::SEL_C3;
        st = sc;
      }
    }
    if (fuse_now) {
      fused_done[pl->fuse_next] = 1;
      // this 3x3 + its only consumer (the 1x1 expand) in one launch; the expand's argument block supplies the second half
      Launch sb;
      if (!make_conv(pl->fuse_next, sb, false)) return nullptr;
      const PackLayer* pb = pack_layer(pl->fuse_next);
      BneckArgs& f = st.bneck;
      const ConvArgs& ca = st.conv; const ConvArgs& cb = sb.conv;
      f.x = ca.x; f.y_mid = ca.y; f.ymid_cp = ca.g.y_cp;
      f.w1 = ca.w; f.hdr1 = ca.hdr; f.hdr1_used = round_up((5 + pl->n_phases) * pl->TM * 4, 1024);
      f.dual1 = pl->dual; f.fast1 = pl->fast; f.relu1 = ca.g.relu;
      f.w2 = cb.w; f.hdr2 = cb.hdr; f.hdr2_bytes = cb.hdr_bytes; f.hdr2_used = round_up((5 + pb->n_phases) * pb->TM * 4, 1024);
      f.dual2 = pb->dual; f.fast2 = pb->fast; f.relu2 = cb.g.relu;
      f.y = cb.y; f.y_cp = cb.g.y_cp; f.y_off = cb.g.y_off; f.y_nvalid = cb.g.y_nvalid;
      f.res = cb.res; f.res_cp = cb.g.res_cp; f.res_off = cb.g.res_off; f.add_relu = cb.g.add_relu; f.has_res = cb.g.has_res;
      f.zero = ca.zero; f.keep_mid = wp->keep_all ? 1 : 0; f.dbl_mid = pl->off_dbl != 0; f.dbl_out = pb->off_dbl != 0;
      f.rnn = res_nonneg_single_clamp(pl->fuse_next) ? 1 : 0;
      f.B = batch; f.H = L.H; f.W = L.W; f.probe = opts.flags;
      f.dbg = (opts.dbg2 && opts.dbg_layer == l) ? opts.dbg2 : nullptr;
      set_fast_div((uint32_t)L.W, &f.w_m, &f.w_s); set_fast_div((uint32_t)(L.W + 2), &f.wp_m, &f.wp_s);
      const int TN = pl->TM == 64 ? 256 : 128;      // pixel capacity of a block (wave tile 32 x 64)
      f.R = std::min(TN / L.W, L.H);
      f.tiles_per_img = (L.H + f.R - 1) / f.R;
      st.sel = Launch::SEL_BNECK; st.shape = TN;
    }
    // this row and the next in one launch (independent rows, same ring-kernel instantiation)?
    if (opts.pair_mode && !fuse_now && !profiling_pairs_off && st.sel == Launch::SEL_MFMA2 && pair_candidate(l) && !(l >= 2 && pair_candidate(l - 1))) {
      Launch sb;
      if (!make_conv(l + 1, sb, true)) return nullptr;
      if (sb.sel == Launch::SEL_MFMA2 && conv_mfma2_pair_eligible(st.conv, st.TM, sb.conv, sb.TM)) {
        st.conv2 = sb.conv; st.sel = Launch::SEL_PAIR;
        fused_done[l + 1] = 1; pair_done[l + 1] = 1;
      }
    }
    // ... two conv_pwk rows of one instantiation (rows 11 | 12 with pwk=1)
    if (opts.pair_mode && !fuse_now && !profiling_pairs_off && st.sel == Launch::SEL_PWK && pair_candidate(l) && !(l >= 2 && pair_candidate(l - 1))) {
      Launch sb;
      if (!make_conv(l + 1, sb, true)) return nullptr;
      if (sb.sel == Launch::SEL_PWK && conv_pwk_pair_eligible(st.conv, sb.conv)) {
        st.conv2 = sb.conv; st.TM2 = sb.TM; st.sel = Launch::SEL_PWKPAIR;
        fused_done[l + 1] = 1; pair_done[l + 1] = 1;
      }
    }
    // ... the same for two split-K rows (small batches: a stage's shortcut convolution and the first 1x1 of its first bottleneck on the
    // 14 x 14 / 7 x 7 maps are both split-K launches of a few dozen blocks; round 6: batch-1 latency, two launches less)
    if (opts.pair_mode && !fuse_now && !profiling_pairs_off && st.sel == Launch::SEL_SK && !st.avg_fused && l + 1 < nl - 1 && pair_candidate(l) && !(l >= 2 && pair_candidate(l - 1))) {
      Launch sb;
      if (!make_conv(l + 1, sb, true)) return nullptr;
      if (sb.sel == Launch::SEL_SK && !sb.avg_fused && sb.shape == st.shape && conv_mfma_sk_pair_eligible(st.conv, sb.conv, opts.sk8_blocks, st.shape)) {
        st.conv2 = sb.conv; st.sel = Launch::SEL_SKPAIR;
        fused_done[l + 1] = 1; pair_done[l + 1] = 1;
      }
    }
    if (l == 0 && stem) {
      const ConvArgs& ca = st.conv;
      StemArgs& f = st.stem;
      f.x = ca.x; f.y = ca.y; f.w = (const int8_t*)(pk + pl->off_w2); f.hdr = ca.hdr; f.zero = ca.zero;
      f.unit = pl->off_unit ? (const int8_t*)(pk + pl->off_unit) : nullptr;
      f.hdr_used = round_up((5 + pl->n_phases) * 64 * 4, 1024);
      f.B = batch; f.H = L.H; f.W = L.W; f.OH = L.OH; f.OW = L.OW;
      set_fast_div((uint32_t)L.OW, &f.ow_m, &f.ow_s); set_fast_div((uint32_t)std::max(1, L.PW), &f.pw_m, &f.pw_s);
      f.relu = ca.g.relu; f.fast = ca.g.fast; f.y_cp = ca.g.y_cp; f.y_off = ca.g.y_off; f.y_nvalid = ca.g.y_nvalid; f.dbl_out = ca.g.dbl_out; f.probe = opts.flags; f.dbg2 = (opts.dbg2 && opts.dbg_layer == 0) ? opts.dbg2 : nullptr;
      // rows per block: the fewest rounds of (two blocks per CU) x rows; two blocks must share a CU's 160 KiB
      long best = -1;
      for (int R = 2; R <= 8; R++) {
        if (2 * conv_stem_lds_bytes(pl->off_unit ? 1 : pl->n_phases, R, L.W, (size_t)f.hdr_used) > 160 * 1024) break;
        const long blocks = (long)batch * ((L.OH + R - 1) / R);
        const long cost = ((blocks + 511) / 512) * R;
        if (best < 0 || cost < best || (cost == best && R == 7)) { best = cost; f.R = R; }
      }
      // the layer's 3x3 / stride 2 / pad 1 max pool in the same launch (conv_stem_pool_kernel): pooled rows per block = the most
      // that lets two blocks share a CU
      stem_pool_fused = false;
      if (opts.stem_pool && L.pool_en && L.pool_S == 3 && L.pool_st == 2 && L.pool_pad == 1 && pl->off_unit && 
          L.PH == (L.OH + 1) / 2 && L.PW == (L.OW + 1) / 2 && L.N == 64 && E.conv_tensor != E.out_tensor) {
        int pk = 0;
        for (int k = 1; k <= 8; k++)
          if (2 * conv_stem_pool_lds_bytes(k, L.W, L.OW, (size_t)f.hdr_used) <= 160 * 1024) pk = k;
        if (pk >= 2) {
          // small batches: fewer pooled rows per block while the grid has fewer than ~192 blocks (batch 1: 19 bands x 2 channel halves = 38 blocks of 7 conv
          // rows on 256 CUs; with one pooled row per block 112 blocks of 3 -- a third more conv rows in all, a shorter chain: round 6, stem_pk_small)
          if (opts.stem_pk_small)
            while (pk > 1 && (long)batch * 2 * ((L.PH + pk - 1) / pk) < 192) pk--;
          const TensorPlan& to = T(E.out_tensor);
          f.yp = base + to.offset; f.PH = L.PH; f.PW = L.PW; f.yp_cp = to.Cp; f.yp_off = E.out_off; f.pk = pk;
          f.bands_per_img = (L.PH + pk - 1) / pk; f.R = 2 * pk + 1;
          st.sel = Launch::SEL_STEM; st.shape = pl->n_phases;
          stem_pool_fused = true; best = -2;
        }
      }
      if (best == -2) {
      } else if (best >= 0) {
        f.bands_per_img = (L.OH + f.R - 1) / f.R;
        st.sel = Launch::SEL_STEM; st.shape = pl->n_phases;
      } else {
        return fail("conv_stem does not fit this first layer; run with TF2_AMD_STEM=0");
      }
    }
    // the last layer of a classifier (1x1 map, split-K kernel) stores the dense logits [batch][N] itself: no copy kernel
    if (l == nl - 1 && !wp->keep_all && st.sel == Launch::SEL_SK && !L.pool_en && !L.endpool && L.concat < 0 &&
        L.PH * L.PW == 1 && (L.N % 16 == 0 || L.N % 16 == 8)) {
      st.conv_direct = st.conv;
      ConvGeom& gd = st.conv_direct.g;
      gd.y_cp = L.N; gd.y_off = 0; gd.y_nvalid = L.N / 16 * 16; gd.y_tail = L.N % 16;
      lp.logits_direct = (int)lp.steps.size();
    }
    // a 3x3 / stride 1 first layer on the 3-channel image: input preparation and the pointwise layer over the im2col tile in ONE
    // launch (conv_first_kernel: the tile stays in LDS); the step's first launch then belongs to table row 0
    if (l == 0 && im2col0 && opts.first_fuse && (st.sel == Launch::SEL_PW || st.sel == Launch::SEL_MFMA2 || st.sel == Launch::SEL_SK) && !fuse_now &&
        st.TM == 64 && pl->TM == 64 && pl->n_mtiles == 1 && pl->nslab == 1 && pl->n_entries == 1 && !pl->w_share &&
        (pl->n_phases == 1 || pl->dual) && !L.endpool && L.concat < 0 && L.add_src < 0 && (L.pool_en != 0) == (E.conv_tensor != E.out_tensor)) {
      Launch& s0 = lp.steps[0];
      const int hdr_used = round_up((5 + pl->n_phases) * 64 * 4, 1024);
      int R = 0, WS = 0; size_t lds = 0;
      const ConvArgs& ca = st.conv;
      // ... and, with a 3x3 / stride 2 pool behind a ReLU (SqueezeNet 1.1's front: stride-2 conv1 + pool1), the pool as well
      // (conv_first_pool_kernel: the conv map stays in LDS)
      // (the kernels index with 32 bits: the launchers refuse batches beyond that, so the plan keeps the separate launches there)
      const bool idx32 = (long long)batch * L.OH * L.OW * 64 < (1ll << 31) && (long long)batch * 3 * nd.image_h * nd.image_w < (1ll << 31);
      const bool plain = idx32 && !L.pool_en && s0.kind == Launch::PREP && conv_first_fits(s0.prep, &R, &WS, &lds, hdr_used);
      const bool pooled = idx32 && L.pool_en && opts.first_pool && s0.kind == Launch::PREP && !ca.g.dbl_out && ca.g.y_nvalid == 64 &&
                          conv_first_pool_fits(s0.prep, L.pool_S, L.pool_st, L.pool_pad, L.PH, L.PW, ca.g.relu, hdr_used, &R, &WS, &lds);
      if (plain || pooled) {
        FirstArgs& f = s0.first;
        f.w = ca.w; f.hdr = ca.hdr; f.y = ca.y; f.im = s0.prep.y;
        f.hdr_used = hdr_used; f.dual = ca.dual; f.relu = ca.g.relu; f.fast = ca.g.fast; f.dbl = ca.g.dbl_out;
        f.y_cp = ca.g.y_cp; f.y_off = ca.g.y_off; f.y_nvalid = ca.g.y_nvalid; f.keep = wp->keep_all ? 1 : 0;
        f.pool = pooled ? 1 : 0;
        if (pooled) {
          const TensorPlan& to = T(E.out_tensor);
          f.yp = base + to.offset; f.PH = L.PH; f.PW = L.PW; f.ppad = L.pool_pad; f.yp_cp = to.Cp; f.yp_off = E.out_off;
        }
        s0.sel = Launch::SEL_FIRST; s0.layer = 0;
        continue;
      }
    }
    lp.steps.push_back(st);
    const TensorPlan& tc = T(E.conv_tensor);
    if (L.pool_en && !(l == 0 && stem_pool_fused) && !(c3_pool_fused && st.sel == Launch::SEL_C3)) {
      pool_step(l, tc, base + tc.offset, L.OH, L.OW);
    } else if (L.endpool && !st.avg_fused) {
      Launch sa; sa.kind = Launch::AVG; sa.layer = l;
      AvgArgs& aa = sa.avg;
      const TensorPlan& to = T(E.out_tensor);
      aa.x = base + tc.offset; aa.y = base + to.offset;
      aa.B = batch; aa.HW = L.PH * L.PW; aa.x_cp = tc.Cp; aa.x_off = 0;
      aa.y_cp = to.Cp; aa.y_off = E.out_off; aa.C = round_up(L.N, 16); aa.mult = L.endpool_mult;
      lp.steps.push_back(sa);
    }
  }
  // The -128 flags of the input preparation (round 6): where the step starts as [prep_rewrite3_rows_kernel][conv_stem_pool_kernel][conv_bfirst_kernel],
  // the preparation reports per image whether a quantised element is -128 (words 16 .. 16 + batch of the workspace's control header), the stem's
  // blocks read their image's word instead of scanning their input tile behind a second barrier (0.9 of a block's 9.8 us), and conv_bfirst --
  // the launch behind the stem -- clears the words for the next step.  A fresh (or re-used) workspace may hold anything there: a non-zero word
  // only sends the first step's blocks down the path that is exact for every image.
  if (opts.q128_flags && lp.steps.size() >= 3 && wp->ctrl_bytes >= 256 && batch <= 48 && lp.steps[0].kind == Launch::PREP && lp.steps[0].sel != Launch::SEL_FIRST &&
      prep_takes_rows_kernel(lp.steps[0].prep) && lp.steps[1].kind == Launch::CONV && lp.steps[1].sel == Launch::SEL_STEM && lp.steps[1].stem.yp &&
      lp.steps[1].layer == 0 && lp.steps[2].kind == Launch::CONV && lp.steps[2].sel == Launch::SEL_BFIRST && !lp.steps[2].bgroup.ctr) {
    unsigned* const q = reinterpret_cast<unsigned*>(base + wp->ctrl_off) + 16;
    lp.steps[0].prep.q128 = q; lp.steps[1].stem.q128 = q; lp.steps[2].bgroup.ctr = q;
  }
  // Split-K launches of a few blocks (batch 1-4: the 7 x 7 and 14 x 14 maps -- 8 or 16 blocks that each stream 150-300 KB of weights) split K over
  // BLOCKS as well (round 6, conv_mfma_sk.hip KSP): ks_parts blocks per output tile, partial tiles through the scratch area, the block that draws the
  // tile's last ticket finishes.  The ticket words sit behind the group flags and are cleared with them by the step's first kernel.
  if (opts.sk_kb && wp->ks_ctr_bytes && !lp.steps.empty() && lp.steps[0].kind == Launch::PREP) {
    size_t ctr_used = 0;
    for (size_t i = 1; i < lp.steps.size(); i++) {
      Launch& st = lp.steps[i];
      if (st.kind != Launch::CONV || st.sel != Launch::SEL_SK || st.avg_fused || (int)i == lp.logits_direct) continue;
      ConvArgs& c = st.conv;
      if (!conv_mfma_sk_ksplit_eligible(c)) continue;
      const long blocks = (long)((c.g.n_pix + 63) / 64) * c.n_mtiles;
      if (blocks > opts.sk_kb_blocks) continue;
      const int n_virt = c.nslab * (c.dual ? 2 : 1);
      int kb = 1;
      while (kb * 2 <= opts.sk_kb_max && n_virt / (kb * 2) >= 8 && c.nslab / (kb * 2) >= 1) kb *= 2;
      if (kb < 2 || kb < opts.sk_kb_min || (size_t)blocks * kb * 16384 > wp->ks_part_bytes || (ctr_used + blocks) * 4 > wp->ks_ctr_bytes) continue;
      c.ks_parts = kb;
      c.ks_part = reinterpret_cast<int32_t*>(base + wp->ks_part_off);
      c.ks_ctr = reinterpret_cast<unsigned*>(base + wp->ks_ctr_off) + ctr_used;
      ctr_used += (size_t)blocks;
    }
    if (ctr_used) {
      lp.steps[0].prep.epoch_ptr = reinterpret_cast<unsigned*>(base + wp->ctrl_off);
      lp.steps[0].prep.n_flag_words = (int32_t)((wp->ctrl_bytes - 256) / 4);
    }
  }
  launch_plans.push_back(std::move(lp));
  return &launch_plans.back();
}

static thread_local LaunchRecorder* g_recorder = nullptr;
LaunchRecorder*& launch_recorder() { return g_recorder; }

// one prepared launch of a step -> its kernel (or, with a recorder installed, its description)
int Net::issue(const Launch& st, const LaunchPlan* lp, const void* images, bool images_are_q, int8_t* logits, void* stream) {
  switch (st.kind) {
    case Launch::PREP: {
      PrepArgs pa = st.prep; pa.img = images; pa.src_is_q = images_are_q ? 1 : 0;
      if (st.sel == Launch::SEL_FIRST) { FirstArgs f = st.first; f.p = pa; return f.pool ? launch_conv_first_pool(f, stream) : launch_conv_first(f, stream); }
      return launch_prep_input(pa, stream);
    }
    case Launch::POOL: return launch_maxpool(st.pool, stream);
    case Launch::AVG: return launch_global_avg(st.avg, stream);
    case Launch::L2N: return launch_l2norm(st.l2n, stream);
    case Launch::CONV:
      switch (st.sel) {
        case Launch::SEL_PW: return launch_conv_pw(st.conv, st.TM, stream);
        case Launch::SEL_PWK: return launch_conv_pwk(st.conv, st.TM, stream);
        case Launch::SEL_PWKPAIR: return launch_conv_pwk_pair(st.conv, st.TM, st.conv2, st.TM2, stream);
        case Launch::SEL_SK:
          if (logits && lp->logits_direct >= 0 && &st == &lp->steps[lp->logits_direct]) {
            ConvArgs cd = st.conv_direct; cd.y = logits;
            return launch_conv_mfma_sk(cd, opts.sk8_blocks, st.shape, stream);
          }
          return launch_conv_mfma_sk(st.conv, opts.sk8_blocks, st.shape, stream);
        case Launch::SEL_MFMA2: return launch_conv_mfma2(st.conv, st.TM, stream);
        case Launch::SEL_BNECK: return launch_conv_bneck(st.bneck, st.TM, st.shape, stream);
        case Launch::SEL_PAIR: return launch_conv_mfma2_pair(st.conv, st.conv2, st.TM, stream);
        case Launch::SEL_SKPAIR: return launch_conv_mfma_sk_pair(st.conv, st.conv2, opts.sk8_blocks, st.shape, stream);
        case Launch::SEL_BBAND: return launch_conv_bband(st.bband, st.bg_c, st.bg_m, stream);
        case Launch::SEL_C3: return launch_conv_c3(st.c3, stream);
        case Launch::SEL_FIRE: return launch_conv_fire(st.fire, stream);
        case Launch::SEL_FC: return launch_conv_fc(st.fc, stream);
        case Launch::SEL_BGROUPF: return launch_conv_bgroup_first(st.bgroup, stream);
        case Launch::SEL_BFIRST: return launch_conv_bfirst(st.bgroup, stream);
        case Launch::SEL_BGROUP:
          if (!st.bg_chain.empty()) return launch_conv_bgroup(st.bg_chain.data(), (int)st.bg_chain.size(), st.bg_hw, st.bg_c, st.bg_m, stream);
          return launch_conv_bgroup(&st.bgroup, 1, st.bg_hw, st.bg_c, st.bg_m, stream);
        case Launch::SEL_STEM: return launch_conv_stem(st.stem, st.shape, stream);
        default: return launch_conv_shift(st.conv, st.signed_in, st.mul24, st.shape, stream);
      }
  }
  return -1;
}

// The launches one step of `batch` images consists of, as the library itself would issue them (kernel, grid, LDS, registers):
// tile shapes, fused pairs and split-K variants are decided in launch_plan / the launchers, nowhere else.  No device needed.
tf2_status Net::describe_launches(int batch, bool concurrent, std::vector<std::pair<int, LaunchRecord>>* out) {
  std::lock_guard<std::mutex> lock(run_mutex);
  if (!packed_valid) { set_error("tf2_net_describe_launches: no packed image"); return TF2_ERR_STATE; }
  if (batch <= 0) { set_error("tf2_net_describe_launches: batch must be positive"); return TF2_ERR_ARG; }
  const WorkPlan* wp = plan(batch, false);
  // the description's plan is built beside the cache (never evicts or replaces a run plan) and dropped again
  const uint8_t* saved = packed_dev;
  if (!packed_dev) packed_dev = packed.data();               // addresses are only formatted into argument blocks nobody launches
  static char fake_ws[16];
  std::list<LaunchPlan> keep;
  keep.swap(launch_plans);
  const LaunchPlan* lp = launch_plan(batch, wp, fake_ws, concurrent, true);
  std::list<LaunchPlan> mine;
  mine.swap(launch_plans);
  launch_plans.swap(keep);
  packed_dev = saved;
  if (!lp) return TF2_ERR_ARG;
  LaunchRecorder rec; rec.name[0] = 0;
  g_recorder = &rec;
  int rc = 0;
  for (const Launch& st : lp->steps) {
    const size_t before = rec.rows.size();
    rc = issue(st, lp, fake_ws, false, nullptr, nullptr);
    if (rc) break;
    for (size_t i = before; i < rec.rows.size(); i++) out->emplace_back(st.layer, rec.rows[i]);
  }
  g_recorder = nullptr;
  if (rc) { set_error("tf2_net_describe_launches: " + std::string(device_last_error())); return TF2_ERR_HIP; }
  return TF2_OK;
}

// The liveness-planned workspace of `batch` images as the library lays it out: every tensor's byte range and the rows between which
// it holds memory, and per row the tensors it reads / writes.  No device needed (tests/test_host_abi.py checks with it that rows
// sharing a launch never share memory).
tf2_status Net::describe_workspace(int batch, bool keep_all, std::vector<TensorPlan>* tensors, std::vector<LayerExec>* rows) {
  std::lock_guard<std::mutex> lock(run_mutex);
  if (!packed_valid) { set_error("tf2_net_describe_workspace: no packed image"); return TF2_ERR_STATE; }
  if (batch <= 0) { set_error("tf2_net_describe_workspace: batch must be positive"); return TF2_ERR_ARG; }
  const WorkPlan* wp = plan(batch, keep_all);
  *tensors = wp->tensors;
  *rows = wp->exec;
  return TF2_OK;
}

size_t Net::workspace_size(int batch, bool keep_all) {
  std::lock_guard<std::mutex> lock(run_mutex);
  const WorkPlan* wp = plan(batch, keep_all);
  return wp ? wp->total_bytes : 0;
}

size_t Net::logits_bytes(int batch) const {
  const tf2_layer_desc& LL = layers[nd.n_layers - 1];
  const size_t hw = LL.endpool ? 1 : (size_t)LL.PH * LL.PW;
  return (size_t)batch * hw * LL.N;
}

tf2_status Net::run(const void* images, bool images_are_q, int batch, void* ws, size_t ws_bytes,
                    int8_t* logits, void* stream, int concurrency, void* mark_event, int mark_after_layer) {
  std::unique_lock<std::mutex> lock(run_mutex);
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
  // batches in flight?  (calls on at least two different streams among the last eight)
  void* const tag = (void*)((uintptr_t)stream + 1);          // the null stream is a stream too; 0 = empty slot
  recent_streams[recent_pos] = tag; recent_pos = (recent_pos + 1) & 7;
  bool concurrent = opts.alt_conc_mode == 1;
  if (concurrency >= 0 && opts.alt_conc_mode == 2) concurrent = concurrency != 0;        // the caller's own statement (tf2_net_run_ex)
  else if (opts.alt_conc_mode == 2)
    for (int i = 0; i < 8 && !concurrent; i++) concurrent = recent_streams[i] != nullptr && recent_streams[i] != tag;
  hipStream_t s = (hipStream_t)stream;
  // group launches spin until the eight members of an image are resident together, one block per CU: never on a stream whose CU
  // mask leaves fewer than 64 CUs (asked per call: the handle does not know what the caller's next stream looks like)
  // (asked only where group launches could be selected: a runtime call per step under the handle's mutex otherwise;
  //  tf2_net_run_stats' small_mask_steps therefore counts such steps of the one-batch-at-a-time path only)
  const bool wide_stream = (concurrent || !opts.bgroup_mode) ? true : stream_cu_count(s) >= 64;
  const bool allow_groups = !concurrent && opts.bgroup_mode && wide_stream;
  const LaunchPlan* lp = launch_plan(batch, wp, ws, concurrent, allow_groups);
  if (!lp) return TF2_ERR_ARG;
  stat_steps++; stat_group_steps += lp->n_groups ? 1 : 0; stat_inflight_steps += concurrent ? 1 : 0; stat_small_mask_steps += wide_stream ? 0 : 1;
  const int nl = nd.n_layers;
  // The enqueue itself runs outside the handle's mutex, under the plan's own (tf2_amd.h threading note): host threads that
  // feed different streams issue their ~40 launches per step side by side.  (Profiling runs keep the handle's mutex: the
  // event lists are the handle's.)
  struct Walk {
    LaunchPlan* lp; std::unique_lock<std::mutex>& net_lock; std::unique_lock<std::mutex> plan_lock;
    ~Walk() {
      if (plan_lock.owns_lock()) plan_lock.unlock();
      if (!net_lock.owns_lock()) net_lock.lock();
      lp->walkers--;
    }
  } walk{const_cast<LaunchPlan*>(lp), lock, {}};
  walk.lp->walkers++;
  const std::shared_ptr<std::mutex> plan_mutex = lp->enqueue;
  if (!profiling && !profiling_loop) lock.unlock();
  walk.plan_lock = std::unique_lock<std::mutex>(*plan_mutex);

  hipEvent_t loop0 = nullptr, loop1 = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int ev_layer = -2;
  auto close_layer_event = [&]() -> tf2_status {
    if (ev_layer >= 0) {
      HIP_OK(hipEventRecord(ev1, s));
      prof_events.emplace_back((void*)ev0, (void*)ev1);
      prof_event_layer.push_back(ev_layer);
      ev_layer = -2;
    }
    return TF2_OK;
  };
  bool mark_pending = mark_event != nullptr;
#ifdef TF2_PROBES
  // tools/probe_run.py: leave out the launches of a layer range (results are then wrong; only durations are read)
  int skip_lo = 1 << 30, skip_hi = -1 << 30;
  if (const long long sk = opt("skip_layers", -1); sk >= 0) { skip_lo = (int)(sk >> 32); skip_hi = (int)(unsigned)sk; }
#endif
  for (const Launch& st : lp->steps) {
#ifdef TF2_PROBES
    if (st.layer >= skip_lo && st.layer <= skip_hi) continue;
#endif
    if (mark_pending && st.layer > mark_after_layer) {       // every launch of layers 0..mark_after_layer is enqueued
      HIP_OK(hipEventRecord((hipEvent_t)mark_event, s));
      mark_pending = false;
    }
    if (profiling && st.layer != ev_layer) {            // one event pair per layer (conv + its pool / average)
      if (tf2_status e = close_layer_event()) return e;
      if (st.layer >= 0) {
        HIP_OK(hipEventCreate(&ev0)); HIP_OK(hipEventCreate(&ev1));
        HIP_OK(hipEventRecord(ev0, s));
        ev_layer = st.layer;
      }
    }
    if (profiling_loop && st.layer == 0 && !loop0) {
      HIP_OK(hipEventCreate(&loop0)); HIP_OK(hipEventCreate(&loop1));
      HIP_OK(hipEventRecord(loop0, s));
    }
    const int rc = issue(st, lp, images, images_are_q, logits, stream);
    if (rc) { set_error("kernel launch failed at layer " + std::to_string(st.layer) + ": " + device_last_error()); return TF2_ERR_HIP; }
  }
  if (mark_pending) HIP_OK(hipEventRecord((hipEvent_t)mark_event, s));
  if (profiling) { if (tf2_status e = close_layer_event()) return e; }
  if (profiling_loop && loop0) {
    HIP_OK(hipEventRecord(loop1, s));
    prof_events.emplace_back((void*)loop0, (void*)loop1);
    prof_event_layer.push_back(-1);
  }
  // dense logits [batch][H_last * W_last][N_last]  (H = W = 1 for the classification networks)
  if (logits && lp->logits_direct < 0) {
    const TensorPlan& tf = wp->tensors[wp->final_tensor];
    const tf2_layer_desc& LL = layers[nl - 1];
    const size_t rows = (size_t)batch * tf.H * tf.W;
    HIP_OK(hipMemcpy2DAsync(logits, (size_t)LL.N, (const int8_t*)ws + tf.offset + wp->exec[nl - 1].out_off, (size_t)tf.Cp,
                            (size_t)LL.N, rows, hipMemcpyDeviceToDevice, s));
  }
  return TF2_OK;
}

// Did a group launch of a step on this workspace give up a meeting (conv_bgroup.hip bg_report)?  Reads the workspace's error word
// (synchronises the stream), clears it, TF2_ERR_GROUP with the report decoded if it was set.  The plan's workspace layout is a
// function of the batch alone, so the caller names the batch the workspace was used for.
tf2_status Net::poll_error(int batch, void* ws, size_t ws_bytes, void* stream) {
  if (!packed_valid || !packed_dev) { set_error("tf2_net_poll_error: no packed model bound"); return TF2_ERR_STATE; }
  if (batch <= 0 || !ws) { set_error("tf2_net_poll_error: bad argument"); return TF2_ERR_ARG; }
  const WorkPlan* wp = nullptr;
  {
    const WorkPlan* a = plan(batch, false);
    auto itk = plans.find(std::make_pair(batch, 1));
    if (itk != plans.end() && ws_bytes >= itk->second.total_bytes) wp = &itk->second;
    else wp = a;
  }
  if (ws_bytes < wp->total_bytes) { set_error("tf2_net_poll_error: workspace too small for this batch"); return TF2_ERR_SIZE; }
  if (!wp->ctrl_bytes) { HIP_OK(hipStreamSynchronize((hipStream_t)stream)); return TF2_OK; }      // no group launch can run on it
  unsigned word = 0;
  unsigned* dev = reinterpret_cast<unsigned*>((uint8_t*)ws + wp->ctrl_off) + 1;
  HIP_OK(hipMemcpyAsync(&word, dev, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  if (!bg_err_valid(word)) return TF2_OK;
  HIP_OK(hipMemsetAsync(dev, 0, 4, (hipStream_t)stream));
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  const unsigned code = word & 0xffffu, meet = code & 0xff, kb = code >> 8;
  set_error(std::string("a group launch gave up a meeting of its eight blocks per image (") +
            (meet == 0x01 ? "roll call" : meet == 0x10 ? "input of a chained bottleneck" : meet == 0x20 ? "first meeting" : "second meeting") +
            ", bottleneck " + std::to_string(kb) + " of its launch): the members were not resident together -- the step's logits are not valid; "
            "see the group-launch preconditions in tf2_amd.h (>= 64 CUs on the stream, at most four concurrent callers) or set TF2_AMD_OPTS=bgroup=0");
  return TF2_ERR_GROUP;
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
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  prof_events.clear(); prof_event_layer.clear();
}

tf2_status Net::read_layer(int layer, int batch, const void* ws, int8_t* dst, size_t cap, void* stream) {
  // the keep_all plan of this batch (planning is deterministic: rebuilt if tf2_net_reload_options dropped it since the run)
  if (batch <= 0) { set_error("tf2_net_read_layer: batch must be positive"); return TF2_ERR_ARG; }
  if (layer < -1 || layer >= nd.n_layers) { set_error("tf2_net_read_layer: bad layer"); return TF2_ERR_ARG; }
  const WorkPlan* wpp;
  { std::lock_guard<std::mutex> lock(run_mutex); wpp = plan(batch, true); }        // (std::map nodes are stable: the plan outlives the lock)
  const WorkPlan& wp = *wpp;
  int tid, off, C;
  if (layer == -1) { tid = wp.input_tensor; off = 0; C = layers[0].C; }
  else { tid = wp.exec[layer].out_tensor; off = wp.exec[layer].out_off; C = layers[layer].N; }
  const TensorPlan& t = wp.tensors[tid];
  const size_t npix = (size_t)batch * t.H * t.W;
  if (layer == -1 && im2col0) {
    // the input tensor holds the im2col image (Net::init): hand back the quantised image [batch][3][H][W] it was gathered from --
    // pixel (r, c) of channel ch is tap (fh, fw) of output pixel (oh, ow) with r + pad_h = oh * stride + fh (a pixel no window covers,
    // possible with stride > 1, is not in the tensor: 0)
    const int IH = nd.image_h, IW = nd.image_w;
    if (cap < (size_t)batch * 3 * IH * IW) { set_error("tf2_net_read_layer: destination too small"); return TF2_ERR_SIZE; }
    std::vector<int8_t> tmp(npix * t.Cp);
    HIP_OK(hipMemcpyAsync(tmp.data(), (const int8_t*)ws + t.offset, tmp.size(), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    for (int b = 0; b < batch; b++)
      for (int ch = 0; ch < 3; ch++)
        for (int r = 0; r < IH; r++)
          for (int c = 0; c < IW; c++) {
            const int oh = std::min((r + im_pad_h) / im_stride, t.H - 1), fh = r + im_pad_h - oh * im_stride;
            const int ow = std::min((c + im_pad_w) / im_stride, t.W - 1), fw = c + im_pad_w - ow * im_stride;
            int8_t v = 0;
            if (fh <= 2 && fw <= 2) v = tmp[(((size_t)b * t.H + oh) * t.W + ow) * t.Cp + ch * 9 + fh * 3 + fw];
            dst[(((size_t)b * 3 + ch) * IH + r) * IW + c] = v;
          }
    return TF2_OK;
  }
  if (cap < npix * C) { set_error("tf2_net_read_layer: destination too small"); return TF2_ERR_SIZE; }
  const size_t Cp = (layer == -1 && packed_valid && stem_selected(batch)) ? 32 : (size_t)t.Cp;     // x-only image tensor (conv_stem.hip)
  std::vector<int8_t> tmp(npix * Cp);
  HIP_OK(hipMemcpyAsync(tmp.data(), (const int8_t*)ws + t.offset, tmp.size(), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  const size_t HW = (size_t)t.H * t.W;
  // doubled channels are stored as 2y - 128 (weight_pack.cpp): hand back y
  const PackLayer* pl = layer >= 0 ? pack_layer(layer) : nullptr;
  const uint8_t* dblf = (pl && pl->off_dbl) ? packed.data() + pl->off_dbl : nullptr;
  for (int b = 0; b < batch; b++)
    for (int c = 0; c < C; c++)
      for (size_t p = 0; p < HW; p++) {
        int8_t v = tmp[((size_t)b * HW + p) * Cp + off + c];
        if (dblf && dblf[c]) v = (int8_t)(((int)v + 128) >> 1);
        dst[((size_t)b * C + c) * HW + p] = v;
      }
  return TF2_OK;
}

}  // namespace tf2

