// host_model.cpp -- load-time numerics of the drop-in (host CPU, C++).
//
// Replaces Runtime_Engine/cnn/host/src/model_loader.cpp (Get_real :98-126, LoadModel
// :129-258, filter_trans :25-96) and quantization.cpp (:25-55): float32 power-of-two
// weights -> one-byte codes (zero | sign | shift), bias/BN -> BiasBnParam fixed point,
// ASCII Q file -> runtime q table.  Results are bit-identical to the reference's own
// compiled functions (tests/test_golden_host.py); the FPGA re-layout (FilterConvert
// :263-322) is not reproduced -- weight_pack.cpp builds the GPU layouts instead.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "tf2_net.h"

namespace tf2 {

// Get_real, model_loader.cpp:98-126.  Comparisons against 0.99/1.01 * 2^-i and 1e-5 are
// done in double exactly as the reference's mixed float/double expressions are.
uint8_t get_real(float data, int8_t expand) {
  if (std::fabs((double)data) < 1.0e-05) return 0x40;
  const bool negative = data < 0;
  const float mag = negative ? -data : data;
  int level = 0;
  for (int i = 0; i < 15; i++) {
    const float unit = 1.0f / (float)(1 << i);
    const double lo = 0.99 * (double)unit, hi = 1.01 * (double)unit;
    if ((double)mag > lo && (double)mag < hi) { level = i; break; }
  }
  int8_t shift = (int8_t)(expand - (int8_t)level);   // char arithmetic in the reference
  if (shift < 0) shift = 0;
  uint8_t code = (uint8_t)shift;
  if (negative) code |= 0x80;
  return code;
}

// filter_trans in closed form (model_loader.cpp:25-96): the 7x7
// plane F becomes nine 3x3 planes; sub-filter s = 2*colphase + rowphase for s < 6,
// s = 6 + colphase for the three "row 6" planes.  Entries the reference never writes keep
// its initial values: 0x40 (= zero weight) inside the staging arrays, but 0x00 (= +x<<0)
// for rows 0-1 of planes 6..8, which LoadModel pre-clears with memset(0) (:246).
static void conv1_plane_rewrite(const uint8_t* F, uint8_t* out /*[9][3][3]*/) {
  auto col_pick = [&](int colphase, int row, int j) -> uint8_t {
    if (colphase == 0) return F[row * 7 + 2 * j];
    if (colphase == 1) return F[row * 7 + 2 * j + 1];
    return j == 2 ? F[row * 7 + 6] : (uint8_t)0x40;
  };
  for (int cp = 0; cp < 3; cp++) {
    for (int rp = 0; rp < 2; rp++) {
      uint8_t* o = out + (2 * cp + rp) * 9;
      for (int r = 0; r < 3; r++)
        for (int j = 0; j < 3; j++) o[r * 3 + j] = col_pick(cp, 2 * r + rp, j);
    }
    uint8_t* o6 = out + (6 + cp) * 9;
    for (int j = 0; j < 6; j++) o6[j] = 0x00;
    for (int j = 0; j < 3; j++) o6[6 + j] = col_pick(cp, 6, j);
  }
}

// Quantization(), quantization.cpp:25-55.
tf2_status Net::quantization(const char* text, size_t len, int8_t* q, size_t cap, int32_t* n_read) const {
  const size_t need = (size_t)nd.n_q_rows * nd.max_out_channel;
  if (cap < need) { set_error("tf2_quantization: q buffer too small"); return TF2_ERR_SIZE; }
  // fscanf("%d") semantics: whitespace-separated decimal ints; a failed read leaves 0.
  std::vector<int> vals;
  {
    size_t i = 0;
    while (i < len) {
      while (i < len && (text[i] == ' ' || text[i] == '\n' || text[i] == '\r' || text[i] == '\t')) i++;
      if (i >= len) break;
      size_t j = i;
      bool neg = false;
      if (text[j] == '-' || text[j] == '+') { neg = text[j] == '-'; j++; }
      if (j >= len || text[j] < '0' || text[j] > '9') break;   // not a number: fscanf stops matching
      long v = 0;
      while (j < len && text[j] >= '0' && text[j] <= '9') { v = v * 10 + (text[j] - '0'); j++; }
      vals.push_back((int)(neg ? -v : v));
      i = j;
    }
  }
  size_t pos = 0;
  const int M = nd.max_out_channel;
  for (int layer = 0; layer < nd.n_layers + 1; layer++) {
    const int conv_layer = layer == 0 ? 0 : layer - 1;
    const tf2_layer_desc& L = layers[conv_layer];
    const int channel = layer == 0 ? 3 : L.N;           // quantization.cpp:39 (literal 3)
    for (int c = 0; c < channel; c++) {
      int8_t* dst = q + (size_t)layer * M + c;
      if (L.ipool == 1) {                               // (2 = L2Norm row: reads its own Q values like a conv row)
        *dst = q[(size_t)L.q_in_row * M + c];           // :42-43
      } else {
        const int v = pos < vals.size() ? vals[pos] : 0;
        pos++;
        *dst = (int8_t)(-v);                            // :46
        if (L.concat >= 0)                              // :47-49
          q[(size_t)(nd.n_conv + 1 + L.concat) * M + L.n_start + c] = (int8_t)(-v);
      }
    }
  }
  if (n_read) *n_read = (int32_t)pos;
  return TF2_OK;
}

// bias_fix / alpha_fix / beta_fix of one layer (model_loader.cpp:176-188, 215-236); null pointers = disabled
static void fold_bias_bn(LayerModel& m, int N, const int8_t* q_out, const float* bias_f, const float* mean, const float* var,
                         float scale_factor, const float* gamma, const float* betaf) {
  m.bias.assign(N, 0); m.alpha.assign(N, 0); m.beta.assign(N, 0);
  for (int n = 0; n < N; n++) {
    const float coe = (float)(1 << (kInflat - q_out[n]));                 // :178,228
    if (bias_f) m.bias[n] = (int32_t)(bias_f[n] * coe);                   // :181
    float alpha_data = 1.0f, beta_data = 0.0f;
    if (mean) {
      const float eps = 0.00001f;                                         // :221
      const float a = mean[n] / scale_factor;                             // :223
      const float b = (float)std::sqrt((double)(var[n] / scale_factor + eps));  // :224
      alpha_data = gamma[n] / b;                                          // :225
      beta_data = -(gamma[n] / b * a) + betaf[n];                         // :226
    }
    m.alpha[n] = (int32_t)((double)alpha_data * std::pow(2.0, kAlphaInflat));   // :230
    const double bb = (double)(coe * beta_data);
    m.beta[n] = (int32_t)(beta_data > 0 ? bb + 0.5 : bb - 0.5);           // :231
  }
}

// LoadModel(), model_loader.cpp:129-258, from an in-memory float stream.
tf2_status Net::load_model(const float* model, size_t n_floats) {
  if (q.empty()) { set_error("tf2_net_load_model: call tf2_net_set_q first"); return TF2_ERR_STATE; }
  const int M = nd.max_out_channel;
  size_t pos = 0;
  auto need = [&](size_t n) -> bool { return pos + n <= n_floats; };
  models.assign(nd.n_layers, LayerModel());
  for (int l = 0; l < nd.n_layers; l++) {
    const tf2_layer_desc& L = layers[l];
    LayerModel& m = models[l];
    const int N = L.N, C = L.model_C, K = L.model_k;
    const int8_t* q_in = q.data() + (size_t)L.q_in_row * M;
    const int8_t* q_out = q.data() + (size_t)(l + 1) * M;
    if (!L.ipool) {
      const size_t cnt = (size_t)N * C * K * K;
      if (!need(cnt)) { set_error("tf2_net_load_model: model stream too short (filters of layer " + std::to_string(l) + ")"); return TF2_ERR_SIZE; }
      m.codes.resize(cnt);
      for (int n = 0; n < N; n++)
        for (int c = 0; c < C; c++) {
          const int8_t expand = (int8_t)(kInflat + q_in[c] - q_out[n]);      // :159-162
          const size_t base = ((size_t)n * C + c) * K * K;
          for (int t = 0; t < K * K; t++) m.codes[base + t] = get_real(model[pos + base + t], expand);
        }
      pos += cnt;
    } else if (L.ipool == 2) {                      // L2Norm row: N float scale weights (l2norm.py:13)
      if (!need(N)) { set_error("tf2_net_load_model: model stream too short (L2Norm weights of layer " + std::to_string(l) + ")"); return TF2_ERR_SIZE; }
      m.l2w.assign(model + pos, model + pos + N); pos += N;
    }
    std::vector<float> bias_f;
    if (L.bias_en) {
      if (!need(N)) { set_error("tf2_net_load_model: model stream too short (bias)"); return TF2_ERR_SIZE; }
      bias_f.assign(model + pos, model + pos + N); pos += N;
    }
    const float *mean = nullptr, *var = nullptr, *gamma = nullptr, *betaf = nullptr;
    float scale_factor = 0.0f;
    if (L.bn_en) {
      if (!need((size_t)4 * N + 1)) { set_error("tf2_net_load_model: model stream too short (bn)"); return TF2_ERR_SIZE; }
      mean = model + pos; pos += N;
      var = model + pos; pos += N;
      scale_factor = model[pos]; pos += 1;
      gamma = model + pos; pos += N;
      betaf = model + pos; pos += N;
    }
    fold_bias_bn(m, N, q_out, L.bias_en ? bias_f.data() : nullptr, mean, var, scale_factor, gamma, betaf);
  }
  if (pos != n_floats) {
    set_error("tf2_net_load_model: model stream has " + std::to_string(n_floats) + " floats, the tables need " + std::to_string(pos));
    return TF2_ERR_SIZE;
  }
  return finish_model();
}

// conv1 rewrite (model_loader.cpp:244-257) and state flags, shared by both loaders
tf2_status Net::finish_model() {
  if (nd.conv1_rewrite) {                                                   // :244-257
    const tf2_layer_desc& L0 = layers[0];
    if (L0.model_k != 7 || L0.model_C != 3 || L0.k != 3 || L0.C != 27) {
      set_error("conv1_rewrite needs a 3x7x7 first filter executed as 27x3x3");
      return TF2_ERR_ARG;
    }
    std::vector<uint8_t> re((size_t)L0.N * 27 * 9);
    for (int n = 0; n < L0.N; n++)
      for (int c = 0; c < 3; c++)
        conv1_plane_rewrite(models[0].codes.data() + ((size_t)n * 3 + c) * 49, re.data() + (size_t)n * 243 + (size_t)c * 81);
    models[0].codes.swap(re);
  }
  model_loaded = true;
  packed_valid = false;
  return TF2_OK;
}

// LoadModel from TransForm_Kit's 4-bit packed file (4bit_data_format.txt:1-44), tensor by tensor in LoadModel order.  A
// filter stored as 4-bit codes becomes the reference's byte codes directly: Get_real(value of the code, expand) through
// a 16-entry table per (tensor, expand) -- the same function on the same values as the float path, hence identical
// codes, without a float32 copy of the weights.  Bias / BN tensors are float32 in the file.
tf2_status Net::load_model_4bit(const uint8_t* bytes, size_t n_bytes) {
  if (q.empty()) { set_error("tf2_net_load_model_4bit: call tf2_net_set_q first"); return TF2_ERR_STATE; }
  const int M = nd.max_out_channel;
  M4Cursor cur(bytes, n_bytes);
  M4Tensor t;
  auto next = [&](size_t want, const char* what, int l) -> bool {
    if (cur.done()) { set_error(std::string("tf2_net_load_model_4bit: file ends before the ") + what + " of layer " + std::to_string(l)); return false; }
    const std::string err = cur.next(&t);
    if (!err.empty()) { set_error(err); return false; }
    if (t.cnt != want) {
      set_error(std::string("tf2_net_load_model_4bit: the ") + what + " of layer " + std::to_string(l) + " has " + std::to_string(t.cnt) +
                " values, the tables need " + std::to_string(want));
      return false;
    }
    return true;
  };
  auto floats = [&](std::vector<float>& v) {            // a small (per-channel) tensor as float32
    v.resize(t.cnt);
    for (size_t i = 0; i < t.cnt; i++) v[i] = t.dtype == 1 ? t.f32(i) : m4_code_value(t.code(i), t.min_exp);
  };
  models.assign(nd.n_layers, LayerModel());
  for (int l = 0; l < nd.n_layers; l++) {
    const tf2_layer_desc& L = layers[l];
    LayerModel& m = models[l];
    const int N = L.N, C = L.model_C, K = L.model_k;
    const int8_t* q_in = q.data() + (size_t)L.q_in_row * M;
    const int8_t* q_out = q.data() + (size_t)(l + 1) * M;
    if (!L.ipool) {
      const size_t cnt = (size_t)N * C * K * K;
      if (!next(cnt, "filters", l)) return TF2_ERR_SIZE;
      m.codes.resize(cnt);
      uint8_t lut[256][16];                                // [expand + 128][4-bit code] -> byte code, filled on demand
      bool have[256] = {false};
      for (int n = 0; n < N; n++)
        for (int c = 0; c < C; c++) {
          const int8_t expand = (int8_t)(kInflat + q_in[c] - q_out[n]);      // model_loader.cpp:159-162
          const size_t base = ((size_t)n * C + c) * K * K;
          if (t.dtype == 0) {
            uint8_t* lt = lut[(int)expand + 128];
            if (!have[(int)expand + 128]) {
              for (int k = 0; k < 16; k++) lt[k] = k == 15 ? 0xff : get_real(m4_code_value(k, t.min_exp), expand);
              have[(int)expand + 128] = true;
            }
            for (int i = 0; i < K * K; i++) {
              const int k = t.code(base + i);
              if (k == 15) { set_error("4-bit model: unused code 15 in the filters of layer " + std::to_string(l)); return TF2_ERR_ARG; }
              m.codes[base + i] = lt[k];
            }
          } else {
            for (int i = 0; i < K * K; i++) m.codes[base + i] = get_real(t.f32(base + i), expand);
          }
        }
    }
    if (L.ipool == 2) { if (!next(N, "L2Norm weights", l)) return TF2_ERR_SIZE; floats(m.l2w); }
    std::vector<float> bias_f, mean, var, sf, gamma, betaf;
    if (L.bias_en) { if (!next(N, "bias", l)) return TF2_ERR_SIZE; floats(bias_f); }
    if (L.bn_en) {
      if (!next(N, "BN mean", l)) return TF2_ERR_SIZE; floats(mean);
      if (!next(N, "BN variance", l)) return TF2_ERR_SIZE; floats(var);
      if (!next(1, "BN scale factor", l)) return TF2_ERR_SIZE; floats(sf);
      if (!next(N, "scale gamma", l)) return TF2_ERR_SIZE; floats(gamma);
      if (!next(N, "scale beta", l)) return TF2_ERR_SIZE; floats(betaf);
    }
    fold_bias_bn(m, N, q_out, L.bias_en ? bias_f.data() : nullptr, L.bn_en ? mean.data() : nullptr, var.data(),
                 L.bn_en ? sf[0] : 0.0f, gamma.data(), betaf.data());
  }
  if (!cur.done()) { set_error("tf2_net_load_model_4bit: the file holds more tensors than the tables need"); return TF2_ERR_SIZE; }
  return finish_model();
}

}  // namespace tf2
