// Internal declarations shared by the host side (C++) and the HIP kernels.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include "../../include/tf2_amd.h"

namespace tf2 {

constexpr int kInflat = 15;        // host/inc/types.h:34
constexpr int kAlphaInflat = 20;   // host/inc/types.h:33
constexpr uint32_t kPackMagic = 0x32465441u;  // "ATF2"
constexpr uint32_t kPackVersion = 33;

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- conv kernel kinds -----------------------------------------------------------
enum ConvKind : int32_t { KIND_NONE = 0, KIND_MFMA = 1, KIND_SHIFT = 2, KIND_L2NORM = 3 };

// Directory entry of the packed weight image, one per layer.  All offsets are relative
// to the start of the image so that it can be broadcast and bound on any rank.
struct PackLayer {
  int32_t kind;        // ConvKind
  int32_t TM;          // rows (output channels) per block tile: 64 or 128 (MFMA)
  int32_t n_mtiles;    // ceil(Np / TM)
  int32_t n_phases;    // Horner phases (MFMA)
  int32_t nslab;       // 64-byte K slabs = ceil(taps * Cp_in / 64)
  int32_t Np;          // padded output channels (multiple of 64 for MFMA, 8 for shift)
  int32_t signed_in;   // input tensor is [x | xneg] (image layers)
  int32_t Cp_in;       // channels per pixel seen by the K ordering (incl. the xneg half)
  int32_t max_shift;   // largest shift amount in the layer (shift kernel: mul24 if <= 22)
  int32_t n_entries;   // number of (mtile, phase, slab) weight tiles stored
  int32_t n_cchunk;    // shift kernel: channel chunks of 16
  int32_t max_ent;     // MFMA: largest number of entries of one m-tile (multiple of 4)
  int32_t fast;        // MFMA: every output row passed the range proof of the 3-instruction requantisation
  int32_t dual;        // MFMA, two-phase layers: every entry holds BOTH exponent windows' tiles ([hi TM rows][lo TM rows],
                       // one activation slab), accumulated separately and combined once: (hi << dshift[1]) + lo
  int32_t fuse_next;   // > 0: this 3x3 layer and layer `fuse_next` (its only consumer, the 1x1 expand) run as ONE conv_bneck launch;
                       // both are packed with TM = this layer's channel count (64 / 128 / 256)
  int32_t fused_into;  // >= 0: the layer whose launch computes this one (-1 otherwise)
  int32_t w_share;     // alternative entries only: 1 = off_w is the MAIN entry's tile storage (tiles of w_main_TM rows); the kernels
  int32_t w_main_TM;   //   address half tiles / tile pairs of it (ConvArgs w_* fields) instead of a second copy of the weights
  uint64_t off_w;        // MFMA: n_entries * TM * 64 bytes; SHIFT: int32 weights (pos [, negmag])
  uint64_t off_w2;       // SHIFT signed mode: magnitudes of negative weights
  uint64_t off_entries;  // int32[n_entries] slab id
  uint64_t off_dir;      // int32[n_mtiles][n_phases + 1] cumulative entry starts
  uint64_t off_kinfo;    // int32[nslab * 4]  coff | dh << 16 | dw << 24  (coff == 0xffff: padding segment)
  uint64_t off_bias;     // int32[Np]
  uint64_t off_alpha;    // int32[Np]
  uint64_t off_beta;     // int32[Np]
  uint64_t off_lo;       // int32[Np]  final left shift of the accumulated sum
  uint64_t off_dshift;   // int32[n_phases][Np]  Horner shift applied when entering phase p>=1
  uint64_t off_hdr;      // MFMA: n_mtiles headers of hdr_bytes each (the LDS image conv_mfma2 copies per block)
  uint64_t hdr_bytes;    // multiple of 1024
  // "Doubled" channels (weight_pack.cpp): an internal post-ReLU tensor whose channels carry exactly two Q values stores the
  // channels with the higher one as 2x - 128, so that its consumers need ONE exponent window instead of two.
  uint64_t off_dbl;      // uint8[Np]: 1 = this layer stores output channel n as 2y - 128 (0: no such channel)
  uint64_t off_pad;      // uint8[Cp_in + 16]: what an out-of-range tap reads (the stored form of x = 0: -128 on doubled input
                         // channels), 0: the zero page
  uint64_t off_unit;     // conv_stem only, != 0: the layer's LOW exponent window is nothing but unit taps (+x << 0, the conv1
                         // rewrite's memset rows, model_loader.cpp:244-257), the same (tap, channel) set in every output row:
                         // int8[9][32] 0/1 mask; off_w2 then holds the HIGH window alone and the kernel adds the per-pixel sum
                         // of the masked inputs to every channel's accumulator instead of sweeping a second window
  // conv_fc layers with their filters as 4-bit codes in HBM (round 5; TransForm_Kit/Compression/compress_net/4bit_data_format.txt: sign +
  // 3-bit exponent code per weight).  fc4 != 0: off_w holds NIBBLE tiles [m-tile * nslab + slab][TM rows][32 bytes] instead of int8
  // window tiles -- no int8 copy of the layer's weights exists in the image; conv_fc.hip expands a lane's 16 codes to the int8 window
  // values in registers, in front of the MFMA, through the row's look-up tables:
  uint64_t off_lut;      // uint8[Np][2 classes][2 windows][8]: window value of exponent code e (e = 7: zero weight) for this output
                         // channel, per input-channel class (channels whose Q differs shift every code by that much) and window
  uint64_t off_cls;      // uint8[nslab * 64]: 0xff where K position k belongs to an input channel of class 1 (n_cls == 2), else 0
  int32_t fc4;
  int32_t n_cls;         // 1 or 2 input-channel classes
  // Merged rows (round 5, weight_pack.cpp): a 1x1 row and the 3x3 / pad 1 row behind it that read the SAME tensor and write ADJACENT
  // slices of the same concat tensor (SqueezeNet's expand1x1 | expand3x3; kNStart / kBranchTail, quantization.cpp:42-49) are ONE 3x3
  // layer of N1 + N3 output channels -- a 1x1 filter is a 3x3 filter whose only non-zero tap is the centre, and all-zero weight tiles
  // are not stored, so the 1x1 rows' m-tiles walk one K slab per input-channel slab, exactly as before.  One launch (and one pool
  // launch) instead of two.
  int32_t merge_next;    // > 0: this row's packed layer also computes row merge_next (its output channels behind this row's)
  int32_t merged_into;   // >= 0 (or -1): this row is computed by that row's launch and has no weights of its own in the image
};

struct PackHeader {
  uint32_t magic, version;
  uint32_t n_layers, dir_bytes;
  uint64_t total_bytes;
  uint64_t tables_hash;     // hash of the layer descs the image was packed for
  uint64_t zero_off;        // offset of an all-zero block of >= max Cp_in + 16 bytes (source of padded taps for LDS-DMA)
};

// ---- device-side parameter blocks -------------------------------------------------
struct ConvGeom {
  int32_t H, W, Cp_in;       // input tensor geometry (pixels, bytes per pixel)
  int32_t OH, OW, OHW;       // conv output geometry
  uint32_t ohw_m, ow_m;      // pixel decode without integer division: n / d == umulhi(n, m) >> s for n < 2^31
  int32_t ohw_s, ow_s;       //   (set_fast_div below; s < 0 means d == 1)
  int32_t stride, pad_h, pad_w;
  int32_t n_pix;             // batch * OH * OW
  int32_t y_cp, y_off;       // output bytes per pixel, channel offset of this layer's slice
  int32_t y_nvalid;          // valid output channels rounded up to the store granule
  int32_t y_tail;            // conv_mfma_sk only: 8 = the 16-channel group starting at y_nvalid holds 8 more valid channels (dense
                             // logits rows of N = 8 mod 16 channels, e.g. 1000: the padding must not run into the next row)
  int32_t res_cp, res_off;   // residual tensor bytes per pixel and channel offset
  int32_t relu, add_relu, has_res;
  int32_t fast;              // PackLayer::fast: header rows hold {0, alpha << lo, B'} (requant_epilogue.h)
  int32_t flags;             // timing-probe bits (TF2_AMD_EXP, read by -DTF2_PROBES builds only)
  int32_t dbl_out;           // the output tensor has doubled channels (PackLayer::off_dbl): header word 0 of a row = -128 or 0
  int32_t avg_mult;          // conv_mfma_sk AVG: != 0 -> the layer's global average (full_size_pool.cl) is computed in the launch:
                             // y / y_cp / y_off then describe the AVERAGED tensor [batch][y_cp], one pixel tile = one image
};

// n / d for 0 <= n < 2^31 as one 32x32->hi multiply and a shift: L = ceil(log2 d), m = floor(2^(31+L) / d) + 1,
// q = umulhi(n, m) >> (L - 1).  (m*d = 2^(31+L) + e, 0 < e <= d, so n*m / 2^(31+L) = n/d + err with err < 2^-L <= 1/d,
// which cannot carry frac(n/d) <= (d-1)/d over 1.)
inline void set_fast_div(uint32_t d, uint32_t* m, int32_t* s) {
  if (d <= 1) { *m = 0; *s = -1; return; }
  int L = 0;
  while ((1ull << L) < d) L++;
  *m = (uint32_t)(((1ull << (31 + L)) / d) + 1);
  *s = L - 1;
}

struct ConvArgs {
  const int8_t* x;
  int8_t* y;
  const int8_t* res;
  const int8_t* w;           // MFMA: int8 tiles; SHIFT: int32 weights
  const int8_t* w2;          // SHIFT signed: negative magnitudes
  const int32_t* bias;       // SHIFT kernel: per-channel BiasBnParam (the MFMA kernels read their LDS header)
  const int32_t* alpha;
  const int32_t* beta;
  const int8_t* zero;        // >= 16 zero bytes (LDS-DMA source for padded / out-of-range taps)
  long long* dbg;            // optional: 16 timestamps of block 0 (tools/layer_times.py), else null
  long long* dbg2;           // optional: per-block {start, end, hw id, xcc id} of one chosen layer
  const int32_t* hdr;        // per-m-tile LDS header images
  int32_t hdr_bytes;
  int32_t dual;              // PackLayer::dual
  uint32_t mt_m; int32_t mt_s; // set_fast_div(n_mtiles): block id -> (pixel tile, channel tile) without a division
  int32_t ent0;              // entries of m-tile 0 (launch heuristics; every block reads its own {first, end} entry from
                             // the last two words of its header's steps[] table -- no per-m-tile table in the kernarg:
                             // a 680-byte kernarg costs ~0.8 us more per launch than a 300-byte one, tools/ubench/launch_cost)
  int32_t max_ent;
  int32_t n_phases, n_mtiles, Np, nslab;
  int32_t k, dil, n_cchunk, Cp_half;   // shift kernel: filter size, dilation, chunks, x|xneg split
  // "Dense" layers (net.hip make_conv: every m-tile holds all nslab slabs in order, one window or dual, Cp_in a multiple of 64):
  // the ring kernels then derive a step's gather words from its index -- slab sl = (tap t, channel slab cs), t = sl / cslabs --
  // instead of reading them from the header image, so the first activation DMAs need nothing but the kernel arguments.
  // Where a block finds its weight tiles.  An entry of the storage holds (dual ? 2 : 1) windows of `storage TM` rows x 64 bytes;
  // a kernel with 64-row tiles can read the halves of 128-row storage tiles and a kernel with 128-row tiles can read pairs of
  // 64-row storage tiles (PackLayer::w_share: a layer's alternative tile height without a second copy of its weights):
  //   tile row group gi (16 rows, 1 KiB) of entry e = w + e * w_ent_bytes + (mtile & 1) * w_sub_step
  //                                                  + (gi & 3) * 1024 + [TM = 128: ((gi >> 2) & 1) * w_half_stride] + window(gi) * w_win_stride
  //   first entry of m-tile mt (DENSE) = ((mt << e_mt_shl) >> e_mt_shr) * nslab
  int32_t w_ent_bytes, w_win_stride, w_half_stride, w_sub_step;
  int32_t e_mt_shl, e_mt_shr;
  int32_t dense;             // 1: arithmetic gather (conv_mfma2 / conv_mfma_sk DENSE instantiations)
  // conv_mfma_sk, K split over BLOCKS as well (round 6; small grids with long K: batch 1-4): ks_parts blocks share an output tile, each walks a part of the
  // slab list and leaves its 64 x 64 int32 partial tile in ks_part[(tile * ks_parts + part)]; the block that draws the last ticket of ks_ctr[tile] adds them
  // up and requantises.  ks_ctr words are zero when the launch starts (cleared by the step's first kernel: PrepArgs::epoch_ptr / n_flag_words)
  int32_t ks_parts;          // 0 / 1: off
  int32_t* ks_part; unsigned* ks_ctr;
  int32_t cslabs;            // Cp_in / 64
  uint32_t cs_m; int32_t cs_s; // set_fast_div(cslabs)
  uint32_t kk_m; int32_t kk_s; // set_fast_div(k)
  ConvGeom g;
};

// conv_fc.hip: a layer whose input is one filter window per image (k x k / pad 0 on a k x k map, 1 x 1 on 1 x 1) at batch <= 32: the
// weight stream split over the whole chip (output channels x K slices), partial sums through a scratch area of the workspace
struct FcArgs {
  const int8_t* x; int8_t* y;
  const int8_t* w;             // dense weight tiles [mtile * nslab + slab][(hi | lo)][tm rows][64]; fc4: nibble tiles [mtile * nslab + slab][tm rows][32]
  const uint8_t* lut;          // fc4: PackLayer::off_lut
  const uint8_t* cls;          // fc4: PackLayer::off_cls
  int32_t fc4, n_cls;          // fc4: 4-bit codes expanded in registers (conv_fc.hip fc4_partial_kernel)
  int32_t chunks;              // image chunks of 32 (grid.z of the partial kernel; the int8 form: always 1)
  const int32_t* hdr;          // per storage m-tile header images (stride hdr_bytes)
  int32_t* part;               // [ksplit][windows][Np][32] int32
  int32_t hdr_bytes, tm;
  int32_t B, K, nslab, Np;     // images (<= 32), bytes per image (= nslab * 64), output channels rounded up to the tile
  int32_t ksplit, slabs_per_split;
  int32_t dual, relu, fast, dbl;
  int32_t y_cp, y_off, y_nvalid;
};
int conv_fc_pick_ksplit(int Np, int nslab);
size_t conv_fc_scratch_bytes(int Np, int nslab, int dual, int batch);
int launch_conv_fc(const FcArgs& a, void* stream);

// conv_c3.hip: a 3x3 / stride 1 / pad 1 layer of a big map, the input's halo tile streamed through LDS once (instead of nine gathers)
struct C3Args {
  const int8_t* x; int8_t* y;
  const int8_t* w;             // dense weight tiles [mtile * 9 * KS + tap * KS + slab][(hi | lo)][tm rows][64]
  const int32_t* hdr;          // per storage m-tile header images (stride hdr_bytes): rows | lo | dshift
  const int8_t* zero2;         // C bytes: the stored form of x = 0 per input channel (the pad row of a padded layer)
  long long* dbg;
  int32_t hdr_bytes, tm;       // rows of a storage tile (64 / 128)
  int32_t tmk;                 // output channels per block (64: four waves per SIMD, two blocks per CU; 128: two waves per SIMD)
  int32_t B, H, W, C, M;       // map, input channels (= bytes per input pixel), output channels rounded up to the tile
  int32_t x_cp;
  int32_t TH, TW, tiles_x, tiles_per_img;
  uint32_t tw_m, hc_m, tx_m, tpi_m; int32_t tw_s, hc_s, tx_s, tpi_s;     // set_fast_div(TW), (TW + 2), (tiles_x), (tiles_per_img)
  int32_t relu, fast, dbl, dual;
  int32_t y_cp, y_off, y_nvalid;
  int32_t pool, PH, PW;        // pool != 0: the layer's 2x2 / stride 2 / pad 0 max pool rides in the launch -- y / y_cp / y_off then describe the
                               // POOLED tensor [B][PH][PW][y_cp], tiles are TH (even) x 32 pixels (conv_c3_pick_tile_pool)
  int32_t w9;                  // the one-slab kernel (conv_c3_w9_kernel) takes the launch: decided in the launch PLAN (conv_c3_takes_w9)
};
bool conv_c3_takes_w9(const C3Args& a, int mode);      // mode = RunOpts::c3_w9: 0 never, 1 where a block walks at least eight tiles, 2 wherever allowed
bool conv_c3_pick_tile(int H, int W, int* TH, int* TW);
bool conv_c3_pick_tile_pool(int H, int W, int* TH, int* TW);
bool conv_c3_shape_ok(int H, int W, int C, int Np, int min_hw);
int launch_conv_c3(const C3Args& a, void* stream);
#ifdef TF2_CHECK_DMA
void conv_bband_check_counts(unsigned long long out[2]);      // -DTF2_CHECK_DMA builds only (vm_track.h)
void conv_c3_check_counts(unsigned long long out[2]);
#endif

// conv_bneck.hip: layer C (3x3 / stride 1 / pad 1, C -> C channels, C = 64 / 128 / 256) followed by its only consumer E
// (1x1, C -> 4C, + residual): one launch per R x W pixel band of an image.
struct BneckArgs {
  const int8_t* x;           // C's input, NHWC with exactly C bytes per pixel
  int8_t* y_mid;             // C's own output tensor (written only with keep_mid)
  int8_t* y;                 // E's output tensor
  const int8_t* res;
  const int8_t* w1;          // C: dense weight tiles [tap * slabs + slab][(hi | lo)][TM][64]
  const int8_t* w2;          // E: dense weight tiles [pass * slabs + slab][(hi | lo)][TM][64]
  const int32_t* hdr1;       // header image of C's single m-tile; the first hdr1_used bytes hold rows | lo | dshift
  const int32_t* hdr2;       // E's four header images (stride hdr2_bytes)
  const int8_t* zero;
  int32_t hdr1_used, hdr2_bytes, hdr2_used;
  int32_t B, H, W, R, tiles_per_img;     // R output rows per block, ceil(H / R) blocks per image
  int32_t dual1, fast1, relu1, dual2, fast2, relu2, add_relu, has_res, keep_mid;
  int32_t dbl_mid;           // the 3x3's output (the intermediate tile) has doubled channels
  int32_t dbl_out;           // the expand's output has doubled channels (only without a residual: weight_pack.cpp)
  int32_t rnn;               // the expand has no ReLU of its own, its residual is a post-ReLU tensor and the sum is clamped to [0, 127]:
                             // one clamp instead of two (requant_epilogue.h RNN; Net::res_nonneg_single_clamp)
  int32_t probe;             // timing probes (ConvGeom::flags of the pair; read by -DTF2_PROBES builds only)
  int32_t ymid_cp, y_cp, y_off, y_nvalid, res_cp, res_off;
  uint32_t w_m, wp_m; int32_t w_s, wp_s;     // set_fast_div(W), set_fast_div(W + 2): the pixel decodes without a run-time division
  long long* dbg;            // tools/bneck_timeline.py: 16 wall-clock stamps per block (waves 0 and 7), or null
};

// conv_bgroup.hip: an identity bottleneck of a small map (1x1 reduce C -> M, 3x3 / 1 / pad 1 M -> M, 1x1 expand M -> C + residual)
// in one launch, eight blocks per image; the intermediates go through the workspace tensors of the three rows
struct BGroupArgs {
  const int8_t* x;           // the bottleneck's input [B][HW*HW][C] (the residual is read through res / res_cp / res_off)
  int8_t* mid1;              // reduce output  [B][HW*HW][M]
  int8_t* mid2;              // 3x3 output     [B][HW*HW][M]
  int8_t* y;                 // expand output
  const int8_t* res;
  const int8_t* w1; const int8_t* w2; const int8_t* w3;        // dense weight tiles [m-tile * nslab + slab][tm rows][64]
  const int32_t* hdr1; const int32_t* hdr2; const int32_t* hdr3;
  const int8_t* zero;        // zero page
  const int8_t* zero2;       // the 3x3's pad row (the stored form of x = 0 of its input tensor)
  unsigned* ctr;             // [B][2][8] flag words of this launch's groups (conv_bgroup.hip bg_signal / bg_wait)
  const unsigned* epoch;     // the workspace's step counter (incremented by the step's first kernel): the value a set flag carries
  long long* dbg;            // optional: 16 wall-clock stamps per block (tools/bgroup_timeline.py), else null
  int32_t hdr1_bytes, hdr2_bytes, hdr3_bytes;
  int32_t tm1, tm2, tm3;     // rows per m-tile of the three layers (64 or 128)
  int32_t B;                 // images of the batch
  int32_t img0;              // first image of this launch (a launch takes at most 32: one block per CU)
  int32_t relu1, relu2, relu3, add_relu, has_res;
  int32_t fast1, fast2, fast3;     // PackLayer::fast
  int32_t dbl1, dbl2, dbl3;        // the layer's output tensor has doubled channels
  int32_t dual1;                   // the reduce is a two-window layer: entries [hi rows | lo rows] (7 x 7 and 28 x 28 kernels)
  int32_t dual2;                   // the 3x3 is a two-window layer (28 x 28 kernel only)
  int32_t dual3;                   // the expand is a two-window layer (56 x 56 kernel only)
  int32_t avg_mult;                // != 0: the bottleneck ends in the global average (full_size_pool.cl); y / y_cp / y_off then
                                   // describe the AVERAGED tensor [B][y_cp] (7 x 7 kernel only)
  int32_t res_cp, res_off, y_cp, y_off;
  // conv_bgroup56f_kernel only (the stage's first bottleneck): the projection shortcut computed inside the launch
  const int8_t* ws; const int32_t* hdrs;     // its dense weight tiles and header images
  int8_t* ys;                                // its own output tensor (written only with keep_s)
  int32_t hdrs_bytes, tms, relu_s, fast_s, keep_s, ys_cp;      // tms: rows per m-tile of the shortcut (64 or 128)
};

// conv_bband.hip: the same three rows (identity bottleneck) in one launch WITHOUT any exchange between blocks: a block owns R output
// rows x the full width of one image and all channels, recomputes the reduce for its halo rows, keeps both intermediates in LDS
struct BBandArgs {
  const int8_t* x;           // the bottleneck's input [B][H*W][C], exactly C bytes per pixel
  int8_t* mid1;              // reduce output  [B][H*W][M]  (written only with keep_mid)
  int8_t* mid2;              // 3x3 output     [B][H*W][M]  (written only with keep_mid)
  int8_t* y;                 // expand output
  const int8_t* res;
  const int8_t* w1; const int8_t* w2; const int8_t* w3;        // dense single-window weight tiles [m-tile * nslab + slab][tm rows][64]
  const int32_t* hdr1; const int32_t* hdr2; const int32_t* hdr3;
  const int8_t* zero;        // zero page
  const int8_t* zero2;       // the 3x3's pad row (the stored form of x = 0 of its input tensor)
  long long* dbg;            // optional: 16 wall-clock stamps per block (tools/bband_timeline.py), else null
  int32_t hdr1_bytes, hdr2_bytes, hdr3_bytes;
  int32_t tm1, tm2, tm3;     // rows per m-tile of the three layers (64 or 128)
  int32_t B, H, W, R, tiles_per_img;     // R output rows per block, ceil(H / R) blocks per image
  int32_t relu1, relu2, relu3, add_relu, has_res, keep_mid;
  int32_t fast1, fast2, fast3;     // PackLayer::fast
  int32_t dbl1, dbl2, dbl3;        // the layer's output tensor has doubled channels
  int32_t dual1, dual2;            // the reduce / the 3x3 is a two-window layer (entries [hi rows | lo rows])
  int32_t res_cp, res_off, y_cp, y_off;
  int32_t probe;                   // -DTF2_PROBES builds only (timing experiments of the launcher)
};

// consecutive identity bottlenecks of the 28 x 28, 14 x 14 or 7 x 7 maps in one launch (conv_bgroup28_kernel / conv_bgroup_kernel /
// conv_bgroup7_kernel: the groups run them back to back)
constexpr int kBgMaxChain = 5;
// the error word of a workspace (control word 1): kBgErrMagic | code once a group launch gave up a meeting (conv_bgroup.hip bg_report);
// code = meeting (0x01 roll call, 0x10 input of a chained bottleneck, 0x20 / 0x30 the two meetings) | bottleneck of the chain << 8
constexpr unsigned kBgErrMagic = 0xE77E0000u;
// (exactly 20 of the 2^32 words are reports: what a fresh or re-used buffer holds at that offset is not mistaken for one)
__host__ __device__ inline bool bg_err_valid(unsigned w) {
  const unsigned meet = w & 0xffu, kb = (w >> 8) & 0xffu;
  return (w & 0xffff0000u) == kBgErrMagic && (meet == 0x01u || meet == 0x10u || meet == 0x20u || meet == 0x30u) && kb < 5u;
}

struct BGroupChain {
  int32_t n;
  BGroupArgs b[kBgMaxChain];
};

// conv_stem.hip: layer 0 in its executed 3x3 / stride 1 / pad 0 form on the x-only image tensor (32 bytes per pixel)
struct StemArgs {
  const int8_t* x;           // [B][H][W][32]: 27 (or fewer) channels of x, zero padded
  int8_t* y;
  const int8_t* w;           // [window][tap][K half][64 rows][16] signed window values (weight_pack.cpp)
  const int32_t* hdr;        // the layer's header image; the first hdr_used bytes hold rows | lo | dshift
  const int8_t* zero;
  const int8_t* unit;        // PackLayer::off_unit: int8[9][32] unit-tap mask (w then holds the high window alone), or null
  int32_t hdr_used;
  int32_t B, H, W, OH, OW;   // input and output maps (OH = H - 2, OW = W - 2)
  int32_t R, bands_per_img;  // output rows per block, ceil(OH / R) blocks per image
  int32_t relu, fast;
  int32_t y_cp, y_off, y_nvalid;
  int32_t dbl_out;           // the output tensor has doubled channels (PackLayer::off_dbl; header rows carry the -128)
  int32_t probe;             // timing probes (-DTF2_PROBES builds only)
  long long* dbg2;           // optional: per-block {start, end, hw id, ...} stamps (tools/block_timeline.py), else null
  // fused 3x3 / stride 2 / pad 1 max pool (conv_stem_pool_kernel; pool.cl:152-260 + pool_tail.cl:91-216 in the reference's pipeline):
  int8_t* yp;                // pooled output tensor [B][PH][PW][yp_cp], or null (y then gets the conv map)
  int32_t PH, PW, yp_cp, yp_off;
  int32_t pk;                // pooled rows per block (2 * pk + 1 conv rows)
  uint32_t ow_m, pw_m; int32_t ow_s, pw_s;     // set_fast_div(OW), set_fast_div(PW)
  const unsigned* q128;      // conv_stem_pool_kernel: per image, != 0 when the image holds a -128 (written by the step's input preparation: PrepArgs::q128), or
                             // null: the block scans its own input tile
};

struct PoolArgs {
  const int8_t* x; int8_t* y;
  int32_t B, H, W, x_cp, x_off;
  int32_t PH, PW, y_cp, y_off;
  int32_t S, st, pad, C16;   // C16: number of 16-channel groups to process
};

struct AvgArgs {
  const int8_t* x; int8_t* y;
  int32_t B, HW, x_cp, x_off, y_cp, y_off, C, mult;
};

// L2Norm row (SSD conv4_3 branch, l2norm.py:19-24) in the engine's integer form -- see oracle/tf2_oracle.c tf2o_l2norm
struct L2NormArgs {
  const int8_t* x; int8_t* y;
  const double* a;            // 2^-Qx[c]
  const double* b;            // w[c] * 2^Qy[c]
  const int32_t* e;           // qs - Qx[c] >= 0
  int32_t n_pix, C, x_cp, y_cp, qs;
};

struct PrepArgs {
  const void* img; int8_t* y;
  int32_t B, C, H, W;         // source image dims
  int32_t OH, OW, y_cp, half; // destination dims, bytes per pixel, offset of the xneg half
  int32_t rewrite;            // 1: 7x7/s2 space-to-depth form (27 channels on 114x114); 2: im2col of a 3x3 first layer on 3 channels
                              //    (27 channels c * 9 + fh * 3 + fw per OUTPUT pixel: the layer then runs as a pointwise one, net.hip)
  int32_t im_stride, im_pad_h, im_pad_w;   // rewrite == 2: the 3x3 layer's own stride and padding
  int32_t q0;                 // runtime (negated) Q of image channel 0
  int32_t src_is_q;           // 1: source already int8
  int32_t xonly;              // 1 (rewrite form only): 32 bytes of x per pixel, no xneg half (conv_stem.hip)
  int32_t bg_poll_limit;      // control word 2 of this step: polls after which a group launch's meeting is reported as failed (conv_bgroup.hip)
  int32_t bg_withhold;        // control word 3 (test-only): 1 + index of a group-launch block that leaves its group at kernel entry, 0 = none
  unsigned* epoch_ptr;        // side job of the step's first kernel: the workspace's step counter += 1 (the value the flags of the
                              // step's conv_bgroup launches carry) and its n_flag_words flag words (256 bytes behind it) cleared, or null
  int32_t n_flag_words;
  unsigned* q128;             // prep_rewrite3_rows_kernel only: per image, |= 1 when a quantised element is -128 (conv_stem_pool_kernel then skips its own scan of
                              // the input tile; conv_bfirst_kernel clears the words behind it), or null
};

// conv_fire.hip: a fire module (squeeze 1x1, then the merged expand1x1 | expand3x3 layer) in one launch of independent row bands
struct FireArgs {
  const int8_t* x;           // the module's input [B][H*W][Cin], exactly Cin bytes per pixel
  int8_t* mid;               // the squeeze's output tensor [B][H*W][mid_cp] (written only with keep_mid)
  int8_t* y;                 // the concat tensor the expands write
  const int8_t* w1;          // squeeze: dense weight tiles [slab][(hi | lo)][tm1 rows][64] of its single m-tile
  const int8_t* w2;          // merged expand: weight tiles [entry][tm2 rows][64] (one window), entries per m-tile from dir2 / ent2
  const int32_t* hdr1; const int32_t* hdr2;      // header images (hdr2: one per m-tile, stride hdr2_bytes)
  const int32_t* ent2;       // merged expand: slab id of every entry (PackLayer::off_entries)
  const int32_t* dir2;       // merged expand: [m-tile][2] first / end entry (PackLayer::off_dir, one window)
  int32_t n_ent2;            // merged expand: entries of the layer (the unconditional prefetch of an EMPTY m-tile is clamped to the last one)
  const int8_t* zero;        // zero page
  const int8_t* zero2;       // the expand's pad row (the stored form of x = 0 of the squeeze's output tensor)
  int32_t hdr2_bytes, tm1, tm2;
  int32_t B, H, W, Cin, Sp, N2;                  // Sp: squeeze channels rounded up to 16 (= bytes per pixel of its tensor); N2: both expands' channels
  int32_t R, tiles_per_img, NT0, WM;             // set by conv_fire_geometry: rows per band, bands per image, halo-band column tiles, wave grid rows
  int32_t relu1, relu2, fast1, fast2, dbl1, dual1, keep_mid;
  int32_t mid_cp, y_cp, y_off, y_nvalid;
  uint32_t w_m, g_m; int32_t w_s, g_s;           // set_fast_div(W), set_fast_div(Sp / 16)
  long long* dbg;                                // tools/fire_timeline.py: 16 wall-clock stamps per block, or null
  // the 3x3 / stride 2 / pad 0 (ceil mode) max pool behind the expands inside the launch: a block then owns PR pooled rows = R = 2 PR + 1 expand rows
  int8_t* yp;                                    // the pooled concat tensor
  int32_t pool, PH, PW, PR, yp_cp, yp_off;
};
bool conv_fire_geometry(int H, int W, int Cin, int Sp, int N2, int tm1, int tm2, int dual1, int pool, FireArgs* f, size_t* lds_out);
int launch_conv_fire(const FireArgs& a, void* stream);      // 1: shape not instantiated / does not fit

// conv_first_kernel (misc_kernels.hip): a 3x3 / stride 1 first layer on the 3-channel image in one launch -- input preparation (the im2col
// tile stays in LDS) + the pointwise MFMA layer over it
struct FirstArgs {
  PrepArgs p;                  // the input side: image, quantisation, im2col geometry (rewrite == 2)
  const int8_t* w;             // the layer's one weight slab [window][64 rows][64 bytes]
  const int32_t* hdr;          // its header image (rows | lo | dshift)
  int8_t* y;                   // the layer's output tensor
  int8_t* im;                  // the im2col tensor (written only with keep: per-layer parity runs read the quantised image back from it)
  int32_t hdr_used, dual, relu, fast, dbl, y_cp, y_off, y_nvalid, keep;
  int32_t R, WS;               // set by the launcher: output (conv) rows per block, row stride of the LDS image tile
  // conv_first_pool_kernel: the layer's 3x3 / stride 2 max pool in the same launch
  int8_t* yp;                  // the pooled tensor
  int32_t pool, PH, PW, ppad, yp_cp, yp_off;
  int32_t PR;                  // set by the launcher: pooled rows per block
};
bool conv_first_fits(const PrepArgs& a, int* R_out, int* WS_out, size_t* lds_out, int hdr_used);
int launch_conv_first(const FirstArgs& f, void* stream);
bool conv_first_pool_fits(const PrepArgs& a, int pool_S, int pool_st, int pool_pad, int PH, int PW, int relu, int hdr_used,
                          int* PR_out, int* WS_out, size_t* lds_out);
int launch_conv_first_pool(const FirstArgs& f, void* stream);

// kernel launchers (tf2_kernels.hip)
int launch_conv_mfma2(const ConvArgs& a, int TM, void* stream);
bool conv_mfma2_pair_eligible(const ConvArgs& a0, int TM0, const ConvArgs& a1, int TM1);     // two independent layers, one launch
int launch_conv_mfma2_pair(const ConvArgs& a0, const ConvArgs& a1, int TM, void* stream);     // (TM: both rows' tile height, 128 or 64)
bool conv_mfma_sk_ksplit_eligible(const ConvArgs& a);     // the split-K kernel can split this layer's K over blocks as well (ConvArgs::ks_parts)
int launch_conv_mfma_sk(const ConvArgs& a, long sk8_blocks, long s3_blocks, void* stream);
bool conv_mfma_sk_pair_eligible(const ConvArgs& a0, const ConvArgs& a1, long sk8_blocks, long s3_blocks);      // two independent split-K rows in one launch (same instantiation)
int launch_conv_mfma_sk_pair(const ConvArgs& a0, const ConvArgs& a1, long sk8_blocks, long s3_blocks, void* stream);   // sk8_blocks: largest grid that takes the 8-wave form
bool conv_pw_eligible(const ConvArgs& a, int TM, int nslab, int k, int dense, int max_slab, long min_pix);   // register-resident pointwise kernel takes the layer? (max_slab / min_pix: RunOpts::pw_slabs / pw_minpix)
int launch_conv_pw(const ConvArgs& a, int TM, void* stream);
bool conv_pwk_eligible(const ConvArgs& a, int TM, int k, int dense, long min_pix, bool force = false);   // short-K pointwise kernel (conv_pwk.hip) takes the layer?
int launch_conv_pwk(const ConvArgs& a, int TM, void* stream);
bool conv_pwk_pair_eligible(const ConvArgs& a0, const ConvArgs& a1);   // two independent rows of one instantiation in one conv_pwk launch?
int launch_conv_pwk_pair(const ConvArgs& a0, int TM0, const ConvArgs& a1, int TM1, void* stream);
void conv_pwk_set_min_units(int u);   // pwk_units: fewest (tile, channel part) units of a conv_pwk row (default 512: two tiles per block and more)
void conv_pwk_set_pipe(int p);     // pwk_pipe (test-only): 0 = the plain epilogue for every row
void conv_pwk_set_slots(int t);    // pwk_slots (test-only): blocks a launch aims at, 0 = 256 (one per CU)
int launch_conv_shift(const ConvArgs& a, int signed_in, int mul24, int packed4, void* stream);   // packed4: a.w = 4-bit codes, a.w2 = A | B (weight_pack.cpp)
size_t conv_shift_lds_bytes(int taps, int signed_in, int packed4);
bool conv_bgroup_shape_ok(int HW, int C, int M);
int launch_conv_bgroup(const BGroupArgs* chain, int n_chain, int HW, int C, int M, void* stream);     // n_chain > 1: 14 x 14 only
int launch_conv_bgroup_first(const BGroupArgs& a, void* stream);            // rows shortcut | reduce, 3x3, expand of the 56 x 56 stage
// conv_bfirst.hip: the same four rows as ONE launch of independent row bands at two blocks per CU (no meetings: the form for batches in
// flight); takes a BGroupArgs (ctr / epoch / img0 unused; keep_s: the three inner tensors are written as well).  1: not instantiated
int launch_conv_bfirst(const BGroupArgs& a, void* stream);
size_t conv_bfirst_lds_bytes(int dual);
bool conv_bband_shape_ok(int H, int W, int C, int M, int R);
int conv_bband_pick_rows(int W, int M, int dual1, int dual2, int wanted, int rows_dd);     // rows per band used when `wanted` are asked for
bool conv_bband_windows_ok(int M, int dual1, int dual2);     // the instantiated (reduce, 3x3) window forms
int launch_conv_bband(const BBandArgs& a, int C, int M, void* stream);        // 1: shape not instantiated / does not fit
int launch_conv_bneck(const BneckArgs& a, int TM, int TN, void* stream);      // 1: shape not instantiated / does not fit
size_t conv_bneck_lds_bytes(int TM, int TN, int R, int W, size_t hdr1_used, size_t hdr2_used);
int launch_conv_stem(const StemArgs& a, int nwin, void* stream);              // 1: does not fit
size_t conv_stem_lds_bytes(int nwin, int R, int W, size_t hdr_used);
size_t conv_stem_pool_lds_bytes(int pk, int W, int OW, size_t hdr_used);
int launch_maxpool(const PoolArgs& a, void* stream);
int launch_global_avg(const AvgArgs& a, void* stream);
bool prep_takes_rows_kernel(const PrepArgs& a);      // launch_prep_input runs prep_rewrite3_rows_kernel (the one that reports -128s through PrepArgs::q128)
int launch_prep_input(const PrepArgs& a, void* stream);
int launch_l2norm(const L2NormArgs& a, void* stream);
const char* device_last_error();

// ---- host model -------------------------------------------------------------------
struct LayerModel {
  std::vector<uint8_t> codes;               // [N][C][k][k] (layer 0 rewritten when conv1_rewrite)
  std::vector<int32_t> bias, alpha, beta;   // BiasBnParam, types.h:39-43
  std::vector<float> l2w;                   // L2Norm rows: per-channel scale weights
};

struct Tensor {           // a device activation tensor [B][H][W][Cp]
  int H = 0, W = 0, C = 0, Cp = 0;
  size_t offset = 0;      // byte offset inside the workspace (per batch plan)
  size_t bytes_per_image = 0;
};

void set_error(const std::string& s);
uint8_t get_real(float data, int8_t expand);

}  // namespace tf2
