// model4bit.cpp -- reader for TransForm_Kit's 4-bit packed model file.
//
// Format (TransForm_Kit/Compression/compress_net/4bit_data_format.txt:1-44; the reference documents it and ships no
// code, so this is the canonical implementation -- tf2_amd/model4bit.py is the writer and an independent reader):
//   a model = the parameter tensors of the network in LoadModel order (model_loader.cpp:154-213), each one
//     int8  min_exp        exponent of the smallest magnitude of the tensor            (4bit_data_format.txt:3-4)
//     int8  dtype          0 = 4-bit codes in 16-bit words (conv / FC filters), 1 = float32   (:5-6)
//     int16 N, C, H, W     little-endian, no padding                                     (:7)
//     payload
//   dtype 1: N*C*H*W float32.
//   dtype 0: one code per weight (:14-37): k = 0..6 -> -2^(min_exp+k), 7 -> 0.0, 8..14 -> +2^(min_exp+k-8), 15 unused.
//     Grouping into 16-bit words (:38-44): W == 1 (1x1 kernels, FC): 4 consecutive weights of the flattened
//     [N][C][H] order per word; otherwise every filter ROW of W weights takes floor(W/3) words of 3 codes (W == 2:
//     one word of 2) plus one word for the remaining W mod 3.  Code j of a word sits in bits [4j, 4j+3]; unused
//     high nibbles are zero.
// model4bit_decode gives the float stream tf2_net_load_model consumes (kept for tools / tests).  tf2_net_load_model_4bit
// does NOT go through it: M4Cursor walks the tensors in place and Net::load_model_4bit (host_model.cpp) turns every 4-bit
// code straight into the reference's byte code (Get_real applied to the code's value, looked up per tensor in a
// 16-entry table) -- no float32 copy of the 25.5 M weights is ever allocated.  Either way a 4-bit model loads
// bit-identically to the float32 file it was made from (tests/test_model4bit.py).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "tf2_net.h"

namespace tf2 {

static inline float code_value(int code, int min_exp) {
  if (code == 7) return 0.0f;
  if (code < 7) return -std::ldexp(1.0f, min_exp + code);
  return std::ldexp(1.0f, min_exp + code - 8);
}

// ---- in-place tensor cursor ------------------------------------------------------------------------------------
std::string M4Cursor::next(M4Tensor* t) {
  if (n - pos < 10) return "4-bit model: truncated tensor header (tensor " + std::to_string(index) + ")";
  t->min_exp = (int8_t)p[pos];
  t->dtype = (int8_t)p[pos + 1];
  int16_t d[4];
  std::memcpy(d, p + pos + 2, 8);
  pos += 10;
  if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0 || d[3] <= 0) return "4-bit model: non-positive dimension in tensor " + std::to_string(index);
  t->N = d[0]; t->C = d[1]; t->H = d[2]; t->W = d[3];
  t->cnt = t->N * t->C * t->H * t->W;
  t->payload = p + pos;
  size_t bytes;
  if (t->dtype == 1) bytes = t->cnt * 4;
  else if (t->dtype == 0) {
    if (t->min_exp < -40 || t->min_exp > 20) return "4-bit model: implausible minimum exponent in tensor " + std::to_string(index);
    const size_t rows = t->N * t->C * t->H;
    t->words_per_row = t->W == 1 ? 0 : (t->W / 3 + (t->W % 3 ? 1 : 0));
    bytes = 2 * (t->W == 1 ? (rows + 3) / 4 : rows * t->words_per_row);
  } else return "4-bit model: unknown data type " + std::to_string(t->dtype) + " in tensor " + std::to_string(index);
  if (n - pos < bytes) return "4-bit model: truncated payload (tensor " + std::to_string(index) + ")";
  pos += bytes;
  index++;
  return std::string();
}

float m4_code_value(int code, int min_exp) { return code_value(code, min_exp); }

// Appends the decoded floats to `out`; returns an error text (empty = ok).
std::string model4bit_decode(const uint8_t* p, size_t n, std::vector<float>* out, size_t* n_floats) {
  size_t pos = 0, total = 0;
  int tensor = 0;
  while (pos < n) {
    if (n - pos < 10) return "4-bit model: truncated tensor header (tensor " + std::to_string(tensor) + ")";
    const int min_exp = (int8_t)p[pos];
    const int dtype = (int8_t)p[pos + 1];
    int16_t d[4];
    std::memcpy(d, p + pos + 2, 8);
    pos += 10;
    if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0 || d[3] <= 0) return "4-bit model: non-positive dimension in tensor " + std::to_string(tensor);
    const size_t N = d[0], C = d[1], H = d[2], W = d[3];
    const size_t cnt = N * C * H * W;
    if (dtype == 1) {
      if (n - pos < cnt * 4) return "4-bit model: truncated float payload (tensor " + std::to_string(tensor) + ")";
      if (out) { const size_t o = out->size(); out->resize(o + cnt); std::memcpy(out->data() + o, p + pos, cnt * 4); }
      pos += cnt * 4;
    } else if (dtype == 0) {
      if (min_exp < -40 || min_exp > 20) return "4-bit model: implausible minimum exponent in tensor " + std::to_string(tensor);
      const size_t rows = N * C * H;
      const size_t words_per_row = W == 1 ? 0 : (W / 3 + (W % 3 ? 1 : 0));
      const size_t words = W == 1 ? (rows + 3) / 4 : rows * words_per_row;
      if (n - pos < words * 2) return "4-bit model: truncated code payload (tensor " + std::to_string(tensor) + ")";
      if (out) out->reserve(out->size() + cnt);
      auto word = [&](size_t i) { uint16_t w; std::memcpy(&w, p + pos + 2 * i, 2); return (unsigned)w; };
      bool bad = false;
      auto emit = [&](unsigned w, int j) {
        const int code = (w >> (4 * j)) & 15;
        if (code == 15) bad = true;
        if (out) out->push_back(code_value(code, min_exp));
      };
      if (W == 1) {
        for (size_t i = 0; i < rows; i++) emit(word(i / 4), (int)(i % 4));
      } else {
        for (size_t r = 0; r < rows; r++)
          for (size_t x = 0; x < W; x++) emit(word(r * words_per_row + x / 3), (int)(x % 3));
      }
      if (bad) return "4-bit model: unused code 15 in tensor " + std::to_string(tensor);
      pos += words * 2;
    } else {
      return "4-bit model: unknown data type " + std::to_string(dtype) + " in tensor " + std::to_string(tensor);
    }
    total += cnt;
    tensor++;
  }
  if (n_floats) *n_floats = total;
  return std::string();
}

}  // namespace tf2
