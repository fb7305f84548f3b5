// weight_pack.cpp -- byte codes + BiasBnParam -> the packed device image.
//
// The reference re-lays filters for its PE array in FilterConvert
// (host/src/model_loader.cpp:263-322); this is the MI355X counterpart.  Two layouts:
//
//  MFMA  (conv_mfma2 / conv_mfma_sk / conv_pw / conv_bneck / conv_stem .hip)  A weight code is the integer +-2^s.  Per
//        output channel n the shifts that occur are covered, from the largest down, by windows of 7 exponents
//        [lo_p[n], lo_p[n]+6]; window p becomes an int8 matrix W_p[n][k] = +-2^(s-lo_p[n]) (zero elsewhere; |value| <= 64).
//        WHY THIS IS EXACT.  The reference accumulates acc = bias + sum_k MUL(x_k, code_k) in a 32-bit int that wraps
//        (pe.cl:43 "change from long int to int"), i.e. in the ring Z/2^32, where MUL is x * (+-2^s) with the shift done on
//        32 bits (pe.cl:36-37).  In that ring multiplication by 2^a is a homomorphism of addition, so for any partition of
//        the taps of row n into windows p with bases lo_p:
//            sum_k x_k * (+-2^{s_k})  ==  sum_p ( sum_{k in p} x_k * (+-2^{s_k - lo_p}) ) << lo_p       (mod 2^32).
//        The inner sums are what the int8 MFMA computes -- exactly, as integers: |x| <= 128, |W| <= 64, K <= 2^16 terms keep
//        every partial sum below 2^31 in magnitude only when the true sum is; when it is not, the MFMA's int32 accumulator
//        wraps mod 2^32 as well, and a wrapped inner sum shifted left by lo_p is still the right residue because
//        (a mod 2^32) << lo == (a << lo) mod 2^32.  Evaluating the outer sum Horner-style, acc = (acc << (lo_{p-1} - lo_p))
//        + inner_p, or keeping one accumulator per window and combining once ((hi << d) + lo, "dual" below) gives the same
//        residue for the same reason.  The -128 negate quirk (pe.cl:32-37: (int8)(-x)) is not a ring identity and is handled
//        separately: [x | xneg] input halves for image layers, the conv_stem correction term, the shift kernel elsewhere.  K is ordered (tap, physical channel) and cut
//        into 64-byte slabs; only slabs with a non-zero weight inside an (m-tile, phase)
//        are stored ("entries").  For image layers the input tensor is [x | xneg] and
//        negative weights become positive magnitudes on the xneg half (pe.cl:32-37 quirk).
//  SHIFT (conv_shift.hip) the integer +-2^s itself as int32, ordered
//        [n/8][c/16][tap][c-half][8 n][8 c] for wave-uniform scalar loads.
//
// The image is position independent (offsets relative to its start) so that rank 0 can
// broadcast it once over RCCL and every rank bind it at its own address.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "tf2_net.h"
#include "opts.h"

namespace tf2 {

namespace {
struct Blob {
  std::vector<uint8_t>& v;
  explicit Blob(std::vector<uint8_t>& vv) : v(vv) {}
  uint64_t alloc(size_t bytes) {
    size_t off = (v.size() + 255) / 256 * 256;
    v.resize(off + bytes, 0);
    return off;
  }
  template <typename T> T* at(uint64_t off) { return reinterpret_cast<T*>(v.data() + off); }
};

inline bool code_zero(uint8_t c) { return (c & 0x40) != 0; }
inline int code_shift(uint8_t c) { return c & 0x1f; }
inline bool code_neg(uint8_t c) { return (c & 0x80) != 0; }
}  // namespace

uint64_t Net::tables_hash() const {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  };
  mix(&nd, sizeof nd);
  mix(layers.data(), layers.size() * sizeof(tf2_layer_desc));
  return h;
}

const PackLayer* Net::pack_layer(int l) const {
  if (!packed_valid || l < 0 || l >= nd.n_layers) return nullptr;
  return reinterpret_cast<const PackLayer*>(packed.data() + sizeof(PackHeader)) + l;
}

// the layer's wide-tile alternative (128-row tiles), or null
const PackLayer* Net::pack_layer_alt(int l) const {
  if (!packed_valid || l < 0 || l >= nd.n_layers) return nullptr;
  const PackLayer* p = reinterpret_cast<const PackLayer*>(packed.data() + sizeof(PackHeader)) + nd.n_layers + l;
  return p->kind == KIND_MFMA ? p : nullptr;
}

tf2_status Net::pack(int mode) {
  if (!model_loaded) { set_error("tf2_net_pack: load a model first"); return TF2_ERR_STATE; }
  if (mode < 0 || mode > 2) { set_error("tf2_net_pack: mode must be 0, 1 or 2"); return TF2_ERR_ARG; }
  const int nl = nd.n_layers;
  packed.clear();
  Blob blob(packed);
  // directory: nl entries, then nl ALTERNATIVE entries (kind 0 where there is none): the same layer packed with 128-row
  // tiles for launches whose grid fills the chip with them (net.hip launch_plan picks per batch size)
  const size_t dir_bytes = sizeof(PackHeader) + (size_t)2 * nl * sizeof(PackLayer);

  // Can the tensor feeding layer l hold negative values (=> the -128 negate quirk matters)?
  std::vector<int> out_signed(nl, 0);
  std::vector<int> concat_signed(std::max(1, nd.n_concat), 0);
  auto src_signed = [&](int src) -> int {
    if (src == -1) return 1;
    if (src >= 0) return out_signed[src];
    return concat_signed[-(src + 2)];
  };
  for (int l = 0; l < nl; l++) {
    const tf2_layer_desc& L = layers[l];
    int s;
    if (L.ipool == 2) {                     // L2Norm: sign(w) * sign(x)
      s = src_signed(L.src);
      for (float wv : models[l].l2w) if (wv < 0) s = 1;
    }
    else if (L.ipool) s = src_signed(L.src);
    else if (L.add_src >= 0) s = L.add_relu ? 0 : 1;
    else s = L.relu ? 0 : 1;
    out_signed[l] = s;
    if (L.concat >= 0 && s) concat_signed[L.concat] = 1;
  }

  const uint64_t zero_off = (dir_bytes + 255) / 256 * 256;       // right behind the directory (allocated per attempt below)
  // ---- fused pairs (conv_bneck.hip): A = a 3x3 / stride 1 / pad 1 conv with C -> C channels (C = 64 / 128 / 256) whose
  // ONLY consumer is B = a 1x1 / stride-1 expand to 4C channels (branch2b -> branch2c of a ResNet bottleneck).  Both are
  // then packed with TM = C rows per m-tile (A: one m-tile, B: four).  Structural test here; after packing, the pair must
  // also pass the kernel's limits (dense weights, at most two exponent windows, LDS), else the image is repacked without it.
  std::vector<int> fuse_next(nl, 0), fused_into(nl, -1);
  std::vector<char> nofuse(nl, 0);
  const bool fusion_on = mode == 0 && !opt_set("nofuse");
  auto decide_fusion = [&]() {
    std::fill(fuse_next.begin(), fuse_next.end(), 0); std::fill(fused_into.begin(), fused_into.end(), -1);
    if (!fusion_on) return;
    std::vector<int> users(nl, 0), user_of(nl, -1);
    for (int j = 0; j < nl; j++) {
      if (layers[j].src >= 0) { users[layers[j].src]++; user_of[layers[j].src] = j; }
      if (layers[j].add_src >= 0) users[layers[j].add_src] += 2;          // a residual read: never fused away
    }
    for (int l = 0; l + 1 < nl; l++) {
      const tf2_layer_desc& A = layers[l];
      if (nofuse[l] || fused_into[l] >= 0) continue;
      if (A.ipool || A.pool_en || A.endpool || A.add_src >= 0 || A.concat >= 0 || A.src < 0) continue;
      if (A.k != 3 || A.stride != 1 || A.pad_h != 1 || A.pad_w != 1 || A.dil != 1 || A.C != A.N || !A.relu) continue;
      if (A.N != 64 && A.N != 128) continue;            // (256-channel pairs on 14x14 maps: too few blocks per launch, measured no faster)
      if (src_signed(A.src) || layers[A.src].concat >= 0) continue;       // unsigned input tensor with exactly C bytes per pixel
      if (users[l] != 1) continue;
      const int b = user_of[l];
      const tf2_layer_desc& B = layers[b];
      if (B.src != l || B.ipool || B.pool_en || B.endpool || B.concat >= 0 || B.k != 1 || B.stride != 1 || (B.pad_h | B.pad_w)) continue;
      if (B.add_src >= l) continue;                                       // the residual must exist when A's launch runs
      if (B.N != 4 * A.N || B.C != A.N) continue;
      const int TN = A.N == 64 ? 256 : 128;
      if (A.W > TN) continue;                                             // at least one full-width row per block
      fuse_next[l] = b; fused_into[b] = l;
    }
  };
  // ---- doubled channels.  The shift of a code is 15 + Q_in[c] - Q_out[n] - i (model_loader.cpp:159-162): a tensor whose
  // channels carry two Q values costs its consumers a ninth exponent level, i.e. a second 7-exponent window.  x is 7-bit after
  // ReLU, so the channels with the HIGHER Q can be stored as x' = 2x - 128 (int8) instead: w * x = (w/2) * x' + 64 * w -- the
  // consumer packs those channels' weights one exponent lower (all of them must be even) and adds 64 * sum(w) to its bias;
  // an out-of-range tap (x = 0) reads -128.  Only tensors nothing but MFMA convolutions read (no residual, pool, average,
  // concat, L2Norm, final output), written by a layer with ReLU, qualify: in ResNet-50 the 64/128/256-channel tensors inside
  // the bottlenecks.  The producer's header rows carry the -128 (requant_epilogue.h).
  std::vector<std::vector<uint8_t>> dbl(nl);
  if (mode == 0 && !opt_set("nodbl")) {
    const int M = nd.max_out_channel;
    for (int p = 0; p + 1 < nl; p++) {
      const tf2_layer_desc& P_ = layers[p];
      if (P_.ipool || !P_.relu || P_.pool_en || P_.endpool || P_.concat >= 0 || P_.add_src >= 0 || out_signed[p]) continue;
      bool ok = true; int n_cons = 0;
      for (int j = 0; j < nl && ok; j++) {
        if (layers[j].add_src == p) ok = false;
        if (layers[j].src != p) continue;
        n_cons++;
        if (layers[j].ipool || layers[j].C != P_.N) ok = false;
      }
      if (!ok || n_cons == 0) continue;
      const int8_t* qo = q.data() + (size_t)(p + 1) * M;
      int v0 = qo[0], v1 = qo[0]; bool two = true;
      for (int n = 0; n < P_.N && two; n++) {
        if (qo[n] == v0 || qo[n] == v1) continue;
        if (v0 == v1) { if (qo[n] < v0) v0 = qo[n]; else v1 = qo[n]; }
        else two = false;
      }
      if (!two || v0 == v1) continue;
      std::vector<uint8_t> f(P_.N, 0);
      for (int n = 0; n < P_.N; n++) f[n] = qo[n] == v1;
      // every consumer weight on a doubled channel must be even (shift >= 1)
      for (int j = 0; j < nl && ok; j++) {
        if (layers[j].src != p) continue;
        const tf2_layer_desc& J = layers[j];
        const int taps = J.k * J.k;
        for (int n = 0; n < J.N && ok; n++)
          for (int c = 0; c < J.C && ok; c++) {
            if (!f[c]) continue;
            for (int t = 0; t < taps; t++) {
              const uint8_t code = models[j].codes[((size_t)n * J.C + c) * taps + t];
              if (!code_zero(code) && code_shift(code) < 1) { ok = false; break; }
            }
          }
      }
      if (ok) dbl[p] = f;
    }
  }
  // ---- merged rows (PackLayer::merge_next): a 1x1 row A and the 3x3 / pad 1 row B right behind it, same input tensor, same geometry,
  // ReLU and pool, adjacent slices of one concat tensor, A's channels whole 64-row tiles.  B's launch disappears: A's packed layer is
  // the 3x3 layer [A's filters as centre taps | B's filters].
  std::vector<int> merge_next(nl, 0), merged_into(nl, -1);
  if (mode == 0 && opt("merge", 1) != 0)
    for (int l = 0; l + 1 < nl; l++) {
      const tf2_layer_desc& A = layers[l]; const tf2_layer_desc& B = layers[l + 1];
      if (merged_into[l] >= 0 || A.ipool || B.ipool || A.src != B.src || A.src == -1 || src_signed(A.src)) continue;
      if (A.k != 1 || A.stride != 1 || (A.pad_h | A.pad_w) != 0 || A.dil > 1 || B.k != 3 || B.stride != 1 || B.pad_h != 1 || B.pad_w != 1 || B.dil > 1) continue;
      if (A.C != B.C || A.H != B.H || A.W != B.W || A.OH != B.OH || A.OW != B.OW || A.relu != B.relu || A.q_in_row != B.q_in_row) continue;
      if (A.add_src >= 0 || B.add_src >= 0 || A.endpool || B.endpool) continue;
      if (A.pool_en != B.pool_en || (A.pool_en && (A.pool_S != B.pool_S || A.pool_st != B.pool_st || A.pool_pad != B.pool_pad || A.PH != B.PH || A.PW != B.PW))) continue;
      if (A.concat < 0 || A.concat != B.concat || B.n_start != A.n_start + A.N || A.N % 64 != 0) continue;
      merge_next[l] = l + 1; merged_into[l + 1] = l;
    }
  for (int attempt = 0; attempt < 8; attempt++) {
  decide_fusion();
  packed.clear();
  blob.alloc(dir_bytes);
  // the zero block: what an out-of-range tap reads (16 bytes at the segment's channel offset), for every layer without its own pad row
  size_t zero_bytes = 256;
  for (int l = 0; l < nl; l++) zero_bytes = std::max(zero_bytes, (size_t)round_up(in_layout[l].Cp_in + 16, 256));
  if (blob.alloc(zero_bytes) != zero_off) { set_error("tf2_net_pack: internal layout error"); return TF2_ERR_STATE; }
  for (int l = 0; l < nl; l++) {
    PackLayer pl{};
    pl.fused_into = -1; pl.merged_into = -1;
    if (merged_into[l] >= 0) {                // computed by the row in front of it (merged rows): no weights of its own
      pl.kind = KIND_MFMA; pl.merged_into = merged_into[l];
      *(blob.at<PackLayer>(sizeof(PackHeader)) + l) = pl;
      continue;
    }
    // the layer as it is packed and executed: the table row, or (merged rows) the 3x3 layer of both rows' output channels
    tf2_layer_desc Lx = layers[l];
    LayerModel merged_model;
    if (merge_next[l] > 0) {
      const tf2_layer_desc& B = layers[merge_next[l]];
      const LayerModel& ma = models[l]; const LayerModel& mb = models[merge_next[l]];
      const int Na = Lx.N, Nb = B.N, Cc = Lx.C;
      merged_model.codes.assign((size_t)(Na + Nb) * Cc * 9, 0x40);            // 0x40: the zero weight (model_loader.cpp:101-102)
      for (int n = 0; n < Na; n++)
        for (int c = 0; c < Cc; c++) merged_model.codes[((size_t)n * Cc + c) * 9 + 4] = ma.codes[(size_t)n * Cc + c];     // the centre tap
      std::copy(mb.codes.begin(), mb.codes.end(), merged_model.codes.begin() + (size_t)Na * Cc * 9);
      for (auto f : {&LayerModel::bias, &LayerModel::alpha, &LayerModel::beta}) {
        (merged_model.*f) = ma.*f;
        (merged_model.*f).insert((merged_model.*f).end(), (mb.*f).begin(), (mb.*f).end());
      }
      Lx.N = Na + Nb; Lx.k = 3; Lx.pad_h = Lx.pad_w = 1; Lx.model_k = 3;
      pl.merge_next = merge_next[l];
    }
    const tf2_layer_desc& L = Lx;
    const LayerModel& base_model = merge_next[l] > 0 ? merged_model : models[l];
    if (L.ipool == 2) {
      // L2Norm row: per channel a = 2^-Qx, b = w * 2^Qy (doubles), e = qs - Qx (left shifts of the exact integer sum of
      // squares), qs = max Qx -- the constants of tf2o_l2norm / l2norm_kernel
      const int M = nd.max_out_channel;
      const int8_t* qx = q.data() + (size_t)L.q_in_row * M;
      const int8_t* qy = q.data() + (size_t)(l + 1) * M;
      const int C = L.N, Cp = round_up(C, 16);
      int qs = -128;
      for (int c = 0; c < C; c++) qs = std::max(qs, -(int)qx[c]);
      std::vector<double> a(Cp, 0.0), b(Cp, 0.0);
      std::vector<int32_t> e(Cp, 0);
      for (int c = 0; c < C; c++) {
        a[c] = std::ldexp(1.0, (int)qx[c]);
        b[c] = (double)models[l].l2w[c] * std::ldexp(1.0, -(int)qy[c]);
        e[c] = qs + (int)qx[c];
        if (e[c] > 12) { set_error("layer " + std::to_string(l) + ": L2Norm input Q spread too wide"); return TF2_ERR_UNSUPPORTED; }
      }
      pl.kind = KIND_L2NORM; pl.Np = Cp; pl.max_shift = qs;
      pl.off_w = blob.alloc((size_t)Cp * 8); std::memcpy(blob.at<uint8_t>(pl.off_w), a.data(), (size_t)Cp * 8);
      pl.off_w2 = blob.alloc((size_t)Cp * 8); std::memcpy(blob.at<uint8_t>(pl.off_w2), b.data(), (size_t)Cp * 8);
      pl.off_bias = blob.alloc((size_t)Cp * 4); std::memcpy(blob.at<uint8_t>(pl.off_bias), e.data(), (size_t)Cp * 4);
      *(blob.at<PackLayer>(sizeof(PackHeader)) + l) = pl;
      continue;
    }
    if (L.ipool) {
      pl.kind = KIND_NONE;
      *(blob.at<PackLayer>(sizeof(PackHeader)) + l) = pl;
      continue;
    }
    int alt_TM = 0;
    // the main variant's entry structure, for the alternative to share its tile storage (PackLayer::w_share)
    std::vector<int32_t> main_entries, main_dir;
    int main_TM = 0, main_P = 0, main_dual = 0, main_nm = 0;
    uint64_t main_off_w = 0;
    for (int variant = 0; variant < 2; variant++) {        // 0: the layer's own entry, 1: its alternative tile height (if any)
    if (variant == 1) {
      const PackLayer& p0 = *(blob.at<PackLayer>(sizeof(PackHeader)) + l);
      // wide alternative (128-row tiles) of a 64-row layer on a small map; NARROW alternative (64-row tiles, split-K) of a
      // 128-row layer on a 28x28 map, for the tiny grids of batch 1-2 (never for the second half of a fused pair; the first
      // half keeps its one-m-tile entry for the fused launch and gets the narrow one for its own)
      const bool wide = p0.kind == KIND_MFMA && p0.TM == 64 && p0.Np % 128 == 0 && p0.Np >= 256 && L.OH * L.OW >= 16 /* not the 1x1-map FC rows: their grid never fills the chip */ && p0.fuse_next <= 0 && p0.fused_into < 0;
      const bool narrow = p0.kind == KIND_MFMA && p0.TM == 128 && p0.fused_into < 0 && L.OH * L.OW <= 3136;      // (round 6: 56 x 56 too -- ResNet-50 row 1 then shares its batch-1 launch with row 2)
      if (!(wide || narrow)) break;
      alt_TM = wide ? 128 : 64;
      pl = PackLayer{}; pl.fused_into = -1; pl.merged_into = -1; pl.merge_next = merge_next[l];
    }
    // doubled input channels: weights one exponent lower, 64 * sum(w) into the bias (Z/2^32 like the accumulator)
    const std::vector<uint8_t>* in_dbl = (L.src >= 0 && !dbl[L.src].empty()) ? &dbl[L.src] : nullptr;
    LayerModel mm;
    if (in_dbl) {
      mm = base_model;
      const int tp = L.k * L.k;
      for (int n = 0; n < L.N; n++)
        for (int c = 0; c < L.C; c++) {
          if (!(*in_dbl)[c]) continue;
          for (int t = 0; t < tp; t++) {
            uint8_t& code = mm.codes[((size_t)n * L.C + c) * tp + t];
            if (code_zero(code)) continue;
            const int sft = code_shift(code);
            const uint32_t half_w = 1u << (sft - 1);
            mm.bias[n] = (int32_t)((uint32_t)mm.bias[n] + (code_neg(code) ? 0u - half_w * 128u : half_w * 128u));
            code = (uint8_t)((code & 0xe0) | (sft - 1));
          }
        }
    }
    const LayerModel& m = in_dbl ? mm : base_model;
    const int N = L.N, C = L.C, k = L.k, taps = k * k;
    const InLayout& il = in_layout[l];
    const bool in_signed = src_signed(L.src) != 0;
    const bool is_image = L.src == -1;
    bool use_mfma;
    if (mode == 2) use_mfma = false;
    else if (mode == 1) use_mfma = (k == 1);
    else use_mfma = true;
    if (in_signed && !is_image) use_mfma = false;     // no xneg half on mid-network tensors
    if (taps > 49 && !use_mfma) { set_error("layer " + std::to_string(l) + ": filter larger than 7x7 needs the MFMA path"); return TF2_ERR_UNSUPPORTED; }

    int max_shift = 0;
    for (uint8_t c : m.codes) if (!code_zero(c)) max_shift = std::max(max_shift, code_shift(c));
    pl.max_shift = max_shift;
    pl.signed_in = in_signed ? 1 : 0;
    pl.Cp_in = il.Cp_in;

    if (use_mfma) {
      pl.kind = KIND_MFMA;
      const int Np = round_up(N, 64);
      // 128-row tiles for the big-map layers; 64-row tiles where one image has <= 14x14 output
      // pixels, so that the grid still covers the 256 CUs at small batch (conv_mfma2.hip)
      constexpr int tm128_minpix = 196;
      int TM = (Np % 128 == 0 && L.OH * L.OW > tm128_minpix) ? 128 : 64;
      if (fuse_next[l] > 0) TM = Np;                                   // fused pair: the 3x3 in one m-tile ...
      if (fused_into[l] >= 0) TM = layers[fused_into[l]].N;            // ... and the expand in four of the same height
      pl.fuse_next = fuse_next[l]; pl.fused_into = fused_into[l];
      if (variant == 1) { TM = alt_TM; pl.fuse_next = 0; pl.fused_into = -1; }      // an alternative is never launched fused
      const int n_mtiles = Np / TM;
      const int Ktot = taps * il.Cp_in;
      const int nslab = (Ktot + 63) / 64;
      const int Kp = nslab * 64;
      // ---- per-row exponent windows ----
      std::vector<std::vector<int>> row_lo(N);
      int P = 1;
      for (int n = 0; n < N; n++) {
        bool present[32] = {false};
        const uint8_t* rc = m.codes.data() + (size_t)n * C * taps;
        for (int i = 0; i < C * taps; i++) if (!code_zero(rc[i])) present[code_shift(rc[i])] = true;
        int top = 31;
        while (true) {
          while (top >= 0 && !present[top]) top--;
          if (top < 0) break;
          const int lo = std::max(top - 6, 0);
          row_lo[n].push_back(lo);
          top = lo - 1;
        }
        P = std::max(P, (int)row_lo[n].size());
      }
      // lo[p][n]; rows with fewer windows repeat their last one (Horner shift 0)
      std::vector<int32_t> lo((size_t)P * Np, 0);
      for (int n = 0; n < N; n++)
        for (int p = 0; p < P; p++) {
          const auto& r = row_lo[n];
          lo[(size_t)p * Np + n] = r.empty() ? 0 : r[std::min<size_t>(p, r.size() - 1)];
        }
      // ---- dense per-phase int8 matrices ----
      std::vector<int8_t> W((size_t)P * Np * Kp, 0);
      for (int n = 0; n < N; n++) {
        const auto& r = row_lo[n];
        for (int c = 0; c < C; c++)
          for (int t = 0; t < taps; t++) {
            const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
            if (code_zero(code)) continue;
            const int s = code_shift(code);
            int p = 0;
            while (p + 1 < (int)r.size() && s < r[p]) p++;
            const int rel = s - r[p];
            int kk = t * il.Cp_in + c;
            int val = 1 << rel;
            if (code_neg(code)) {
              if (in_signed) kk += il.half;       // +magnitude on the xneg half
              else val = -val;
            }
            W[((size_t)p * Np + n) * Kp + kk] = (int8_t)val;
          }
      }
      // ---- entries ----
      // Two-phase layers (the common case with per-channel Q: stages 2-4 of ResNet-50) are stored DUAL: one entry per
      // slab that is non-zero in either window, holding [hi window TM x 64][lo window TM x 64].  The kernels then
      // fetch each activation slab once, keep two accumulators and combine them once at the end,
      // (hi << dshift[1]) + lo -- exact in Z/2^32 like the Horner form -- which halves the K steps.
      const bool dual = P == 2 && !opt_set("nodual");
      pl.dual = dual ? 1 : 0;
      std::vector<int32_t> dir((size_t)n_mtiles * (P + 1), 0);
      std::vector<int32_t> entries;
      std::vector<int8_t> tiles;
      auto slab_nonzero = [&](int p, int mt, int sl) {
        for (int r = 0; r < TM; r++) {
          const int8_t* src = &W[((size_t)p * Np + mt * TM + r) * Kp + sl * 64];
          for (int b = 0; b < 64; b++) if (src[b]) return true;
        }
        return false;
      };
      auto push_tile = [&](int p, int mt, int sl) {
        const size_t base = tiles.size();
        tiles.resize(base + (size_t)TM * 64);
        for (int r = 0; r < TM; r++)
          std::memcpy(&tiles[base + (size_t)r * 64], &W[((size_t)p * Np + mt * TM + r) * Kp + sl * 64], 64);
      };
      for (int mt = 0; mt < n_mtiles; mt++) {
        if (dual) {
          dir[(size_t)mt * (P + 1)] = (int32_t)entries.size();
          for (int sl = 0; sl < nslab; sl++) {
            if (!slab_nonzero(0, mt, sl) && !slab_nonzero(1, mt, sl)) continue;
            entries.push_back(sl);
            push_tile(0, mt, sl); push_tile(1, mt, sl);
          }
          dir[(size_t)mt * (P + 1) + 1] = dir[(size_t)mt * (P + 1) + 2] = (int32_t)entries.size();
        } else {
          for (int p = 0; p < P; p++) {
            dir[(size_t)mt * (P + 1) + p] = (int32_t)entries.size();
            for (int sl = 0; sl < nslab; sl++) {
              if (!slab_nonzero(p, mt, sl)) continue;
              entries.push_back(sl);
              push_tile(p, mt, sl);
            }
          }
          dir[(size_t)mt * (P + 1) + P] = (int32_t)entries.size();
        }
        pl.max_ent = std::max<int32_t>(pl.max_ent, dir[(size_t)mt * (P + 1) + P] - dir[(size_t)mt * (P + 1)]);
      }
      pl.max_ent = round_up(std::max(pl.max_ent, 1) + 8 + P + 2, 4);   // + spare entries the kernels may read ahead (8-way split-K), + phase table, + {first, end} entry
      // ---- kinfo ----
      // one 32-bit word per 16-byte segment: coff (16 bits, 0xffff = padding) | dh << 16 | dw << 24
      std::vector<int32_t> kinfo((size_t)nslab * 4, 0);
      if (il.Cp_in >= 0xffff || (k - 1) * L.dil > 255) { set_error("layer " + std::to_string(l) + ": geometry exceeds the kinfo encoding"); return TF2_ERR_UNSUPPORTED; }
      for (int sl = 0; sl < nslab; sl++)
        for (int sg = 0; sg < 4; sg++) {
          const int kk0 = sl * 64 + sg * 16;
          const int t = kk0 / il.Cp_in, pc = kk0 % il.Cp_in;
          int32_t& ki = kinfo[(size_t)sl * 4 + sg];
          if (t >= taps) { ki = 0xffff; continue; }
          const int dh = (t / k) * L.dil, dw = (t % k) * L.dil;
          ki = (int32_t)((uint32_t)pc | ((uint32_t)dh << 16) | ((uint32_t)dw << 24));
        }
      pl.TM = TM; pl.n_mtiles = n_mtiles; pl.n_phases = P; pl.nslab = nslab; pl.Np = Np;
      pl.n_entries = (int32_t)entries.size();
      // ---- conv_stem.hip image (off_w2 != 0 on an MFMA layer): the executed first layer on ONE copy of x per pixel ----
      // [window][tap][K half][64 rows][16 bytes] of SIGNED window values (x half minus the magnitudes of the xneg half).
      // The x = -128 correction is derived from these in the kernel.
      if (is_image && in_signed && il.Cp_in == 64 && il.half == 32 && k == 3 && L.stride == 1 && L.dil == 1 && (L.pad_h | L.pad_w) == 0 &&
          Np == 64 && TM == 64 && P <= 2 && C <= 32 && !L.endpool) {
        // Is the low window nothing but the rewrite's unit taps?  (+x << 0 on the x half, identical positions in every row, window
        // base 0.)  Then it contributes the SAME per-pixel sum to every output channel: conv_stem adds that sum instead of sweeping
        // a second window (half the MFMAs and half the weight tile).  nounit keeps the two-window form.
        std::vector<int8_t> unit(9 * 32, 0);
        bool unit_ok = P == 2 && !opt_set("nounit");
        for (int t = 0; t < 9 && unit_ok; t++)
          for (int c = 0; c < 32 && unit_ok; c++) {
            const int8_t v0 = W[((size_t)1 * Np + 0) * Kp + (size_t)t * 64 + c];
            if (v0 != 0 && v0 != 1) unit_ok = false;
            unit[t * 32 + c] = v0;
            for (int r = 0; r < N && unit_ok; r++) {
              if (W[((size_t)1 * Np + r) * Kp + (size_t)t * 64 + c] != v0 || W[((size_t)1 * Np + r) * Kp + (size_t)t * 64 + 32 + c] != 0) unit_ok = false;
              if (lo[(size_t)1 * Np + r] != 0) unit_ok = false;
            }
          }
        const int PS = unit_ok ? 1 : P;                      // windows in the stem image
        std::vector<int8_t> st((size_t)PS * 9 * 64 * 32, 0);
        for (int p = 0; p < PS; p++)
          for (int t = 0; t < 9; t++)
            for (int r = 0; r < 64; r++) {
              const int8_t* src = &W[((size_t)p * Np + r) * Kp + (size_t)t * 64];
              int8_t* tile = &st[((size_t)p * 9 + t) * 64 * 32];
              for (int c = 0; c < 32; c++) tile[(c >> 4) * 1024 + r * 16 + (c & 15)] = (int8_t)(src[c] - src[32 + c]);
            }
        pl.off_w2 = blob.alloc(st.size());
        std::memcpy(blob.at<uint8_t>(pl.off_w2), st.data(), st.size());
        if (unit_ok) {
          pl.off_unit = blob.alloc(unit.size());
          std::memcpy(blob.at<uint8_t>(pl.off_unit), unit.data(), unit.size());
        }
      }
      // An alternative (the same layer with the other tile height) reads the MAIN entry's tiles when every m-tile of both walks
      // the same slab list phase by phase (dense weights always do): a 64-row tile is half of a 128-row storage tile, a 128-row
      // tile is the pair of storage tiles of two neighbouring m-tiles.  Saves the second copy of the weights.
      bool share = false;
      // (Only the WIDE alternatives -- 128-row tiles as pairs of the main 64-row tiles, 22 of the 24 duplicate MB of ResNet-50 -- share
      //  by default: the narrow ones serve the tiny grids of batch 1-2, where reading halves of 8-KiB-strided tiles measured +2 % on the
      //  batch-1 latency; share=2 shares those too, share=01 none.)
      const int share_mode = (int)opt("share", 1);
      if (variant == 1 && main_TM && share_mode && main_P == P && main_dual == (dual ? 1 : 0) &&
          (2 * main_TM == TM || (share_mode >= 2 && main_TM == 2 * TM))) {
        share = true;
        auto same_lists = [&](const std::vector<int32_t>& ent, const std::vector<int32_t>& dr, int nm) {
          for (int mt = 0; mt < nm && share; mt++)
            for (int p2 = 0; p2 < P && share; p2++) {
              const int a0 = dr[(size_t)mt * (P + 1) + p2], a1 = dr[(size_t)mt * (P + 1) + p2 + 1];
              const int r0 = main_dir[p2], r1 = main_dir[p2 + 1];
              if (a1 - a0 != r1 - r0) { share = false; break; }
              for (int i = 0; i < a1 - a0; i++) if (ent[a0 + i] != main_entries[r0 + i]) { share = false; break; }
            }
        };
        same_lists(main_entries, main_dir, main_nm);
        same_lists(entries, dir, n_mtiles);
      }
      // ---- conv_fc layers: the filters stay 4-BIT CODES in HBM (PackLayer::fc4; 4bit_data_format.txt) ----
      // A whole-window layer (k x k / pad 0 on a k x k map, 1 x 1 on 1 x 1: VGG16's fc6 / fc7) is a weight stream: every weight byte is
      // read once per batch of 32, the kernel is HBM-bound with idle VALUs -- the one place where the reference's storage format pays on
      // this chip.  INQ weights are sign + one of 7 exponents, and the shift of a code is 15 + Q_in[c] - Q_out[n] - i
      // (model_loader.cpp:159-162): s = A[n] + B[c] - e, e in 0..6.  Stored: one nibble {sign << 3 | e} (e = 7: zero) per weight, per
      // output channel a table "window value of e" for each of at most TWO input-channel classes (B[c] takes two values where the
      // input's channels carry two Q values) and each exponent window, and the class of every K position.  conv_fc.hip expands a
      // lane's 16 codes with two v_perm_b32 per four weights and window.  Every weight is decoded here once more and compared with
      // the dense window matrices: the form is used only when it reproduces them byte for byte.
      std::vector<uint8_t> nibt, lutv, clsm;
      int n_cls = 1;
      bool fc4 = variant == 0 && mode == 0 && opt("fc4", 1) != 0 && !L.ipool && L.k == L.H && L.k == L.W && L.stride == 1 && L.dil == 1 &&
                 (L.pad_h | L.pad_w) == 0 && L.OH == 1 && L.OW == 1 && L.src >= 0 && L.add_src < 0 && !L.endpool && !L.pool_en && L.concat < 0 &&
                 layers[L.src].concat < 0 && l != nl - 1 && !in_signed && (P == 1 || dual) && Np % 128 == 0 && il.Cp_in % 64 == 0 &&
                 il.Cp_in == C /* (Net::fc_at: the kernel walks whole 64-channel slabs of an unpadded tensor; C = 500 on Cp 512 stays int8 tiles for the split-K kernel) */ && Kp == Ktot && (long)entries.size() == (long)n_mtiles * nslab && nslab >= (int)opt("fc_min", 64) /* conv_fc's own threshold (RunOpts::fc_min_slabs): a layer packed this way can run nowhere else */ && fuse_next[l] <= 0 && fused_into[l] < 0;
      if (fc4) {
        const int Mq = nd.max_out_channel;
        std::vector<int> Bc(il.Cp_in, 0), An(Np, -1000);
        bool fits = false;
        for (int pass = 0; pass < 2 && !fits; pass++) {           // pass 0: B from the input's Q row; pass 1: B = 0
          std::fill(Bc.begin(), Bc.end(), 0); std::fill(An.begin(), An.end(), -1000);
          if (pass == 0) {
            if (L.q_in_row < 0 || C > Mq) continue;
            const int8_t* q_in = q.data() + (size_t)L.q_in_row * Mq;
            for (int c = 0; c < C; c++) Bc[c] = (int)q_in[c] - ((in_dbl && (*in_dbl)[c]) ? 1 : 0);      // (doubled channels: codes one exponent down)
          }
          fits = true;
          for (int n = 0; n < N; n++)
            for (int c = 0; c < C; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (!code_zero(code)) An[n] = std::max(An[n], code_shift(code) - Bc[c]);
              }
          for (int n = 0; n < N && fits; n++)
            for (int c = 0; c < C && fits; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (code_zero(code)) continue;
                const int e = An[n] + Bc[c] - code_shift(code);
                if (e < 0 || e > 6) { fits = false; break; }
              }
        }
        int b0 = 0, b1 = 0;
        if (fits) {
          b0 = b1 = Bc[0];
          for (int c = 0; c < C; c++) { b0 = std::min(b0, Bc[c]); b1 = std::max(b1, Bc[c]); }
          for (int c = 0; c < C && fits; c++) if (Bc[c] != b0 && Bc[c] != b1) fits = false;      // more than two classes: int8 tiles
          n_cls = b0 == b1 ? 1 : 2;
        }
        if (fits) {
          // look-up tables [Np][class][window][e]
          lutv.assign((size_t)Np * 32, 0);
          for (int n = 0; n < N; n++) {
            if (An[n] == -1000) continue;                          // an all-zero row
            const auto& r = row_lo[n];
            for (int k2 = 0; k2 < 2; k2++)
              for (int e = 0; e < 7; e++) {
                const int sft = An[n] + (k2 ? b1 : b0) - e;
                if (sft < 0 || sft > 31 || r.empty()) continue;
                int pw = 0;
                while (pw + 1 < (int)r.size() && sft < r[pw]) pw++;
                const int rel = sft - r[pw];
                if (pw < 2 && rel >= 0 && rel <= 6) lutv[(size_t)n * 32 + k2 * 16 + pw * 8 + e] = (uint8_t)(1 << rel);
              }
          }
          clsm.assign((size_t)nslab * 64, 0);
          for (int kk = 0; kk < Ktot; kk++) { const int pc = kk % il.Cp_in; if (n_cls == 2 && pc < C && Bc[pc] == b1) clsm[kk] = 0xff; }
          // nibble tiles [m-tile * nslab + slab][TM rows][32 bytes]; a row's 32 bytes = [K half h][K step ks] 8 bytes each, inside them
          // word w (4 bytes), byte j: low nibble = K position h * 16 + ks * 32 + 8 w + j of the slab, high nibble = that + 4 -- a lane's
          // 16 bytes (its K half) expand to the two 16-byte MFMA operands of the slab word by word (conv_fc.hip fc4_expand)
          nibt.assign((size_t)n_mtiles * nslab * TM * 32, 0x77);
          for (int n = 0; n < N; n++)
            for (int c = 0; c < C; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (code_zero(code)) continue;
                const int kk = t * il.Cp_in + c, sl = kk >> 6, ki = kk & 63;
                const int h = (ki >> 4) & 1, ks = ki >> 5, within = ki & 15, w = within >> 3, j = within & 3, hi_nib = (within >> 2) & 1;
                const unsigned v = (unsigned)(An[n] + Bc[c] - code_shift(code)) | (code_neg(code) ? 8u : 0u);
                uint8_t& bt = nibt[(((size_t)(n / TM) * nslab + sl) * TM + (n % TM)) * 32 + h * 16 + ks * 8 + w * 4 + j];
                bt = hi_nib ? (uint8_t)((bt & 0x0f) | (v << 4)) : (uint8_t)((bt & 0xf0) | v);
              }
          // the proof: every (row, K position) decoded as the kernel does == the dense window matrices
          for (int n = 0; n < N && fits; n++)
            for (int kk = 0; kk < Kp && fits; kk++) {
              const int sl = kk >> 6, ki = kk & 63;
              const int h = (ki >> 4) & 1, ks = ki >> 5, within = ki & 15, w = within >> 3, j = within & 3, hi_nib = (within >> 2) & 1;
              const uint8_t bt = nibt[(((size_t)(n / TM) * nslab + sl) * TM + (n % TM)) * 32 + h * 16 + ks * 8 + w * 4 + j];
              const unsigned nb = hi_nib ? bt >> 4 : bt & 15;
              const int cl = clsm[kk] ? 1 : 0;
              for (int pw = 0; pw < P; pw++) {
                int val = lutv[(size_t)n * 32 + cl * 16 + pw * 8 + (nb & 7)];
                if (nb & 8) val = -val;
                if (val != (int)W[((size_t)pw * Np + n) * Kp + kk]) { fits = false; break; }
              }
            }
        }
        fc4 = fits;
      }
      if (fc4) {
        pl.fc4 = 1; pl.n_cls = n_cls;
        pl.off_w = blob.alloc(nibt.size()); std::memcpy(blob.at<uint8_t>(pl.off_w), nibt.data(), nibt.size());
        pl.off_lut = blob.alloc(lutv.size()); std::memcpy(blob.at<uint8_t>(pl.off_lut), lutv.data(), lutv.size());
        pl.off_cls = blob.alloc(clsm.size()); std::memcpy(blob.at<uint8_t>(pl.off_cls), clsm.data(), clsm.size());
      } else if (share) {
        pl.off_w = main_off_w; pl.w_share = 1; pl.w_main_TM = main_TM;
        // the alternative's m-tile mt starts at the main m-tile's first entry: mt / 2 (halves of 128-row tiles) or 2 mt (pairs)
        const int nent = main_dir[P] - main_dir[0];
        for (int mt = 0; mt < n_mtiles; mt++) {
          const int mm_ = main_TM == 2 * TM ? mt >> 1 : 2 * mt;
          for (int p2 = 0; p2 <= P; p2++) dir[(size_t)mt * (P + 1) + p2] = mm_ * nent + (main_dir[p2] - main_dir[0]);
        }
        // (entries[] is indexed by these storage entry numbers from here on: the main list repeated per main m-tile)
        entries = main_entries;
        pl.n_entries = (int32_t)entries.size();
      } else {
        pl.off_w = blob.alloc(std::max<size_t>(tiles.size(), 64));
        std::memcpy(blob.at<uint8_t>(pl.off_w), tiles.data(), tiles.size());
      }
      if (variant == 0) { main_entries = entries; main_dir = dir; main_TM = TM; main_P = P; main_dual = dual ? 1 : 0; main_nm = n_mtiles; main_off_w = pl.off_w; }
      pl.off_entries = blob.alloc(std::max<size_t>(entries.size(), 1) * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_entries), entries.data(), entries.size() * 4);
      pl.off_dir = blob.alloc(dir.size() * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_dir), dir.data(), dir.size() * 4);
      pl.off_kinfo = blob.alloc(kinfo.size() * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_kinfo), kinfo.data(), kinfo.size() * 4);
      std::vector<int32_t> dshift((size_t)P * Np, 0), lo_last(Np, 0);
      for (int n = 0; n < Np; n++) {
        for (int p = 1; p < P; p++) dshift[(size_t)p * Np + n] = lo[(size_t)(p - 1) * Np + n] - lo[(size_t)p * Np + n];
        lo_last[n] = lo[(size_t)(P - 1) * Np + n];
      }
      pl.off_lo = blob.alloc((size_t)Np * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_lo), lo_last.data(), (size_t)Np * 4);
      pl.off_dshift = blob.alloc(dshift.size() * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_dshift), dshift.data(), dshift.size() * 4);
      // ---- per-m-tile LDS header images for conv_mfma2.hip ----
      // words: per row {bias, alpha, beta64.lo, beta64.hi} (4*TM; beta64 = (int64)beta << 20, the addend of the
      //        64-bit multiply-add; one 16-byte read per output row) | lo[TM] | dshift[P][TM] | steps[max_ent] (Horner phase steps to take before
      //        the entry) | goff[max_ent][4] (per 16-byte segment: byte offset from the pixel's tap origin,
      //        -1 = K padding) | ghw[max_ent][4] (dh | dw << 8 for the zero-padding test, channel offset << 16);
      //        steps[max_ent-2], steps[max_ent-1] = the m-tile's first and end entry (scalar-loaded by every block)
      // ---- range proof for the fast requantisation (requant_epilogue.h) ----
      // With |x| <= 128, |sum| <= amax[n] = 128 * sum |w|.  If for every row  amax + |bias| < 2^31  (no int32 wrap
      // of v = bias + (acc << lo)),  |alpha| << lo < 2^31,  and  (amax + |bias|) * |alpha| + (|beta| << 20) + 2^34 <
      // 2^51  (x = p >> 20 fits int32 and x + 2^14 does not saturate), then
      //   y = (acc * (alpha << lo) + B') >> 35,   B' = bias * alpha + (beta << 20) + 2^34
      // is the reference result exactly, and the kernels use it (3 VALU instructions per output instead of 6).
      bool fast = !opt_set("nofast");
      for (int n = 0; n < N && fast; n++) {
        unsigned __int128 amax = 0;
        const uint8_t* rc = m.codes.data() + (size_t)n * C * taps;
        for (int i = 0; i < C * taps; i++) if (!code_zero(rc[i])) amax += (unsigned __int128)128 << code_shift(rc[i]);
        const unsigned __int128 ab = (unsigned __int128)std::llabs((long long)m.bias[n]);
        const unsigned __int128 aa = (unsigned __int128)std::llabs((long long)m.alpha[n]);
        const unsigned __int128 abeta = (unsigned __int128)std::llabs((long long)m.beta[n]);
        const unsigned __int128 one = 1;
        if (amax + ab >= (one << 31)) fast = false;
        else if ((aa << lo_last[n]) >= (one << 31)) fast = false;
        else if ((amax + ab) * aa + (abeta << kAlphaInflat) + (one << 34) >= (one << 51)) fast = false;
      }
      // SEMI (requant_epilogue.h): the rows that may wrap v still get the short form when x cannot leave 32 bits
      bool semi = !fast && !opt_set("nofast") && !opt_set("nosemi");
      for (int n = 0; n < N && semi; n++) {
        const unsigned __int128 aa = (unsigned __int128)std::llabs((long long)m.alpha[n]);
        const unsigned __int128 abeta = (unsigned __int128)std::llabs((long long)m.beta[n]);
        const unsigned __int128 one = 1;
        if ((aa << 31) + (abeta << kAlphaInflat) + (one << 34) >= (one << 51)) semi = false;
      }
      pl.fast = fast ? 1 : (semi ? 2 : 0);
      if (!dbl[l].empty()) {
        std::vector<uint8_t> f(Np, 0);
        std::copy(dbl[l].begin(), dbl[l].end(), f.begin());
        pl.off_dbl = blob.alloc(Np);
        std::memcpy(blob.at<uint8_t>(pl.off_dbl), f.data(), Np);
      }
      if (in_dbl) {
        std::vector<int8_t> pad(il.Cp_in + 16, 0);
        for (int c = 0; c < C; c++) if ((*in_dbl)[c]) pad[c] = -128;
        pl.off_pad = blob.alloc(pad.size());
        std::memcpy(blob.at<uint8_t>(pl.off_pad), pad.data(), pad.size());
      }
      {
        const size_t words = (size_t)5 * TM + (size_t)P * TM + (size_t)9 * pl.max_ent;
        const size_t hb = (words * 4 + 1023) / 1024 * 1024;
        pl.hdr_bytes = hb;
        pl.off_hdr = blob.alloc(hb * n_mtiles);
        for (int mt = 0; mt < n_mtiles; mt++) {
          int32_t* h = blob.at<int32_t>(pl.off_hdr + (uint64_t)mt * hb);
          for (int r = 0; r < TM; r++) {
            const int n = mt * TM + r;
            const int64_t b64 = (int64_t)(n < N ? m.beta[n] : 0) << kAlphaInflat;
            int32_t* pr = h + 4 * r;                       // one row's parameters: one 16-byte LDS read
            if (fast) {
              const int64_t al = n < N ? (int64_t)m.alpha[n] : 0;
              const int64_t bp = (n < N ? (int64_t)m.bias[n] * al : 0) + b64 + ((int64_t)1 << 34);
              pr[0] = (n < N && !dbl[l].empty() && dbl[l][n]) ? -128 : 0;      // doubled channel: stored as 2y - 128 (requant_epilogue.h)
              pr[1] = (int32_t)(al << lo_last[n]);
              pr[2] = (int32_t)(uint32_t)((uint64_t)bp & 0xffffffffu);
              pr[3] = (int32_t)(uint32_t)((uint64_t)bp >> 32);
            } else {
              const int64_t add = b64 + (semi ? ((int64_t)1 << 34) : 0);       // SEMI rows: B'' = (beta << 20) + 2^34
              pr[0] = n < N ? m.bias[n] : 0; pr[1] = n < N ? m.alpha[n] : 0;
              pr[2] = (int32_t)(uint32_t)((uint64_t)add & 0xffffffffu);
              pr[3] = (int32_t)(uint32_t)((uint64_t)add >> 32);
            }
            // the final shift; generic rows of a doubled channel carry the -128 above it (FAST rows: word 0)
            h[4 * TM + r] = lo_last[n] | ((!fast && n < N && !dbl[l].empty() && dbl[l][n]) ? (int32_t)0xffff8000 : 0);
            for (int p = 0; p < P; p++) h[5 * TM + p * TM + r] = dshift[(size_t)p * Np + n];
          }
          int32_t* hs = h + 5 * TM + P * TM;
          int32_t* ko = hs + pl.max_ent;
          int32_t* kh = ko + 4 * pl.max_ent;
          const int e0 = dir[(size_t)mt * (P + 1)], e1 = dir[(size_t)mt * (P + 1) + P];
          // steps[p-1] = iteration index (relative to the m-tile's first entry) at which phase p starts
          for (int i = 0; i < pl.max_ent; i++) hs[i] = 0x7fffffff;
          if (!dual) for (int p = 1; p < P; p++) hs[p - 1] = dir[(size_t)mt * (P + 1) + p] - e0;
          for (int i = 0; i < pl.max_ent * 4; i++) { ko[i] = -1; kh[i] = 0; }
          hs[pl.max_ent - 2] = e0; hs[pl.max_ent - 1] = e1;
          for (int e = e0; e < e1; e++) {
            const int sl = entries[e];
            for (int sg = 0; sg < 4; sg++) {
              const int kk0 = sl * 64 + sg * 16;
              const int t = kk0 / il.Cp_in, pc = kk0 % il.Cp_in;
              int32_t& o = ko[(e - e0) * 4 + sg];
              int32_t& hw = kh[(e - e0) * 4 + sg];
              if (t >= taps) { o = -1; hw = 0; continue; }
              const int dh = (t / k) * L.dil, dw = (t % k) * L.dil;
              o = (dh * L.W + dw) * il.Cp_in + pc;
              hw = dh | (dw << 8) | (pc << 16);      // pc: where an out-of-range tap reads inside the layer's pad row
            }
          }
        }
      }
    } else {
      pl.kind = KIND_SHIFT;
      dbl[l].clear();                     // the shift kernel writes plain outputs
      if (in_dbl) { set_error("layer " + std::to_string(l) + ": internal error, doubled input on a shift-kernel layer"); return TF2_ERR_STATE; }
      const int Np = round_up(N, 8);
      const int n_cchunk = (C + 15) / 16;
      if (n_cchunk * 16 > (in_signed && is_image ? il.half : il.Cp_in)) {
        set_error("layer " + std::to_string(l) + ": channel chunks exceed the input tensor"); return TF2_ERR_ARG;
      }
      pl.Np = Np; pl.n_cchunk = n_cchunk; pl.nslab = 0; pl.n_phases = 1; pl.TM = 8; pl.n_mtiles = Np / 8;
      const size_t cnt = (size_t)(Np / 8) * n_cchunk * taps * 128;
      // ---- packed 4-bit filters (PackLayer::fast = 1 on a shift layer): INQ weights are sign + one of 7 exponents, and the
      // shift of a code is kInflat + Q_in[c] - Q_out[n] - i (model_loader.cpp:159-162), i.e. s = A[n] + B[c] - e with e in
      // 0..6.  Stored: one nibble per weight {sign << 3 | e, e = 7: zero weight} in the order of the int32 layout below, then
      // A[Np] and B[n_cchunk * 16] as int8; conv_shift.hip expands a block's nibbles to +-2^s in LDS.  A layer whose codes do
      // not fit that form (arbitrary shifts: tests/test_gpu_parity.py::test_wide_shift_range_phases) or whose expanded tile
      // does not fit LDS keeps the int32 form.
      {
        const int M = nd.max_out_channel;
        std::vector<int> Bc(n_cchunk * 16, 0), An(Np, -1000);
        bool ok4 = !opt_set("no4bit") && taps <= (in_signed ? 25 : 49);
        for (int pass = 0; pass < 2 && ok4; pass++) {          // pass 0: B from the input's Q row; pass 1: B = 0
          std::fill(Bc.begin(), Bc.end(), 0); std::fill(An.begin(), An.end(), -1000);
          if (pass == 0) {
            if (L.src == -1 || L.q_in_row < 0 || C > M) continue;
            const int8_t* q_in = q.data() + (size_t)L.q_in_row * M;
            for (int c = 0; c < C; c++) Bc[c] = (int)q_in[c];
          }
          bool fits = true;
          for (int n = 0; n < N; n++)
            for (int c = 0; c < C; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (!code_zero(code)) An[n] = std::max(An[n], code_shift(code) - Bc[c]);
              }
          for (int n = 0; n < N && fits; n++)
            for (int c = 0; c < C && fits; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (code_zero(code)) continue;
                const int e = An[n] + Bc[c] - code_shift(code);
                if (e < 0 || e > 6) { fits = false; break; }
              }
          for (int n = 0; n < Np; n++) if (An[n] == -1000) An[n] = 0;
          for (int n = 0; n < Np && fits; n++) if (An[n] < -128 || An[n] > 127) fits = false;
          if (fits) { ok4 = true; goto have4; }
        }
        ok4 = false;
      have4:
        if (ok4) {
          std::vector<uint8_t> nib(cnt / 2, 0x77);              // every weight zero
          for (int n = 0; n < N; n++)
            for (int c = 0; c < C; c++)
              for (int t = 0; t < taps; t++) {
                const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
                if (code_zero(code)) continue;
                const size_t idx = (((size_t)(n >> 3) * n_cchunk + (c >> 4)) * taps + t) * 128 +
                                   (size_t)((c >> 3) & 1) * 64 + (size_t)(n & 7) * 8 + (c & 7);
                const unsigned v = (unsigned)(An[n] + Bc[c] - code_shift(code)) | (code_neg(code) ? 8u : 0u);
                uint8_t& b = nib[idx >> 1];
                b = (idx & 1) ? (uint8_t)((b & 0x0f) | (v << 4)) : (uint8_t)((b & 0xf0) | v);
              }
          std::vector<int8_t> ab(Np + n_cchunk * 16, 0);
          for (int n = 0; n < Np; n++) ab[n] = (int8_t)An[n];
          for (int c = 0; c < n_cchunk * 16; c++) ab[Np + c] = (int8_t)Bc[c];
          pl.fast = 1;
          pl.off_w = blob.alloc(nib.size());
          std::memcpy(blob.at<uint8_t>(pl.off_w), nib.data(), nib.size());
          pl.off_w2 = blob.alloc(ab.size());
          std::memcpy(blob.at<uint8_t>(pl.off_w2), ab.data(), ab.size());
        }
      }
      if (!pl.fast) {
      std::vector<int32_t> w(cnt, 0), w2(in_signed ? cnt : 0, 0);
      for (int n = 0; n < N; n++)
        for (int c = 0; c < C; c++)
          for (int t = 0; t < taps; t++) {
            const uint8_t code = m.codes[((size_t)n * C + c) * taps + t];
            if (code_zero(code)) continue;
            const uint32_t mag = 1u << code_shift(code);
            const size_t idx = (((size_t)(n >> 3) * n_cchunk + (c >> 4)) * taps + t) * 128 +
                               (size_t)((c >> 3) & 1) * 64 + (size_t)(n & 7) * 8 + (c & 7);
            if (code_neg(code)) {
              if (in_signed) w2[idx] = (int32_t)mag;
              else w[idx] = (int32_t)(0u - mag);
            } else {
              w[idx] = (int32_t)mag;
            }
          }
      pl.off_w = blob.alloc(cnt * 4);
      std::memcpy(blob.at<uint8_t>(pl.off_w), w.data(), cnt * 4);
      if (in_signed) {
        pl.off_w2 = blob.alloc(cnt * 4);
        std::memcpy(blob.at<uint8_t>(pl.off_w2), w2.data(), cnt * 4);
      }
      }   // int32 form
    }
    // ---- per-channel epilogue parameters (padded rows: all zero => output 0) ----
    {
      const int Np = pl.Np;
      std::vector<int32_t> b(Np, 0), al(Np, 0), be(Np, 0);
      std::copy(m.bias.begin(), m.bias.end(), b.begin());
      std::copy(m.alpha.begin(), m.alpha.end(), al.begin());
      std::copy(m.beta.begin(), m.beta.end(), be.begin());
      pl.off_bias = blob.alloc((size_t)Np * 4); std::memcpy(blob.at<uint8_t>(pl.off_bias), b.data(), (size_t)Np * 4);
      pl.off_alpha = blob.alloc((size_t)Np * 4); std::memcpy(blob.at<uint8_t>(pl.off_alpha), al.data(), (size_t)Np * 4);
      pl.off_beta = blob.alloc((size_t)Np * 4); std::memcpy(blob.at<uint8_t>(pl.off_beta), be.data(), (size_t)Np * 4);
    }
    *(blob.at<PackLayer>(sizeof(PackHeader)) + l + (size_t)variant * nl) = pl;
    }   // variant
  }
  // ---- do the fused pairs pass the kernel's limits? ----
  bool redo = false;
  for (int l = 0; l < nl; l++) {
    if (fuse_next[l] <= 0) continue;
    const tf2_layer_desc& A = layers[l];
    PackLayer* pa = blob.at<PackLayer>(sizeof(PackHeader)) + l;
    PackLayer* pb = blob.at<PackLayer>(sizeof(PackHeader)) + fuse_next[l];
    bool ok = pa->kind == KIND_MFMA && pb->kind == KIND_MFMA && pa->n_mtiles == 1 && pb->n_mtiles == 4 && pb->TM == pa->TM;
    ok = ok && (pa->n_phases == 1 || pa->dual) && (pb->n_phases == 1 || pb->dual);
    ok = ok && pa->nslab == 9 * (pa->TM / 64) && pa->n_entries == pa->nslab && pb->nslab == pa->TM / 64 && pb->n_entries == 4 * pb->nslab;
    if (ok) {
      const int TN = pa->TM == 64 ? 256 : 128;
      const size_t h1 = (size_t)round_up((5 + pa->n_phases) * pa->TM * 4, 1024), h2 = (size_t)round_up((5 + pb->n_phases) * pb->TM * 4, 1024);
      ok = conv_bneck_lds_bytes(pa->TM, TN, TN / A.W, A.W, h1, h2) <= 160 * 1024;
    }
    if (!ok) { nofuse[l] = 1; redo = true; }
  }
  if (!redo) break;
  }   // attempt
  blob.alloc(0);
  PackHeader h{};
  h.magic = kPackMagic; h.version = kPackVersion; h.n_layers = (uint32_t)nl; h.dir_bytes = (uint32_t)dir_bytes;
  h.total_bytes = packed.size(); h.tables_hash = tables_hash(); h.zero_off = zero_off;
  *blob.at<PackHeader>(0) = h;
  packed_valid = true;
  pack_mode = mode;
  packed_dev = nullptr; packed_dev_bytes = 0;
  launch_plans.clear(); plans.clear();       // tensor lifetimes and launches depend on the image (fused / merged rows)
  return TF2_OK;
}

}  // namespace tf2
