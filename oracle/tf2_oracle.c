/*
 * tf2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the TF2 Runtime_Engine/cnn integer inference
 * path.  It exists to CHECK the HIP path in
 * tf2_amd/ and to provide the `cpu_baseline` leg of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (tf2_amd/) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/Runtime_Engine/cnn).  Layout here is the reference's LOGICAL
 * layout: activations [C][H][W] int8 per image, filters [N][C][FH][FW] one byte
 * code each.  The FPGA vector re-layouts (FilterConvert / InputConvert,
 * model_loader.cpp:263-322, input_loader.cpp:123-157) carry no arithmetic and
 * are not restated.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - tf2o_get_real, tf2o_encode_filters, tf2o_fold_bias_bn, tf2o_filter_trans,
 *     tf2o_feature_trans, tf2o_q_table, tf2o_quantize_input, tf2o_topk are
 *     checked bit-for-bit against the reference's own compiled host functions
 *     (oracle/_ref/libtf2ref_*.so built by oracle/Makefile from the sources
 *     where they lie) -- tests/test_oracle_vs_ref.py -- and against golden
 *     vectors generated from them (tests/golden/).
 *   - tf2o_mul / tf2o_conv follow device/src/pe.cl:27-49,144-180 and are checked
 *     LIVE against the reference's own MUL / DotProduct (pe.cl compiled as C in
 *     place: oracle/ref_pe_probe.c -> oracle/_ref/libtf2ref_pe.so; every
 *     (feature, code) pair, wrapping 16-channel dot products) and
 *     against golden conv sums produced by the reference's Python emulator
 *     (TransForm_Kit/Quantization/debug, Conv2dInt8) on inputs without -128.
 *   - requant / relu / max-pool / stride-2 subsampling / residual add / global
 *     average / FC follow the OpenCL device code (pe.cl:185-203, relu.cl:50-56,
 *     pool.cl:152-260, pool_tail.cl:91-216, feature_writer.cl:88-137,
 *     full_size_pool.cl:95-125), which cannot be built here (needs Intel's aoc).
 *     They are pinned by EXECUTING the reference's own Python FPGA emulator
 *     (TransForm_Kit/Quantization/debug/...Batch-2.py: Bottleneck.forward
 *     :249-323 and ResNet.forward :395-443, AST-extracted by
 *     oracle/gen_golden.py gen_pyemu_block) and comparing every intermediate
 *     tensor (tests/golden/ref_pyemu_block.npz, tests/test_golden_pyemu.py):
 *     max-pool, clamp+ReLU, strided conv and the int16 residual add+clamp+ReLU
 *     exactly on every element; requant on every element that is not a
 *     rounding tie, over > 10^5 samples, with the exact ties enumerated (the
 *     emulator rounds half to even in float, the FPGA adds 1 and shifts,
 *     pe.cl:191-193 -- the oracle follows the FPGA, asserted); the global
 *     average's 669/2^15 rule (full_size_pool.cl:115-118) on every element
 *     outside the band where it provably differs from round(mean).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REALMAX 127            /* host/inc/types.h:31 */
#define REALMIN (-128)         /* host/inc/types.h:32 */
#define ALPHA_INFLAT 20        /* host/inc/types.h:33 */
#define INFLAT 15              /* host/inc/types.h:34 */

int tf2o_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
}
void tf2o_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* a11  Get_real  -- host/src/model_loader.cpp:98-126                        */
/* ------------------------------------------------------------------------ */
uint8_t tf2o_get_real(float data, int8_t expand) {
  int sign = 0;
  int8_t vals = 0;
  if (fabs(data) < 1.0e-05) return 0x40;               /* :101-102 (double compare) */
  if (data < 0) { sign = 1; data = -data; }            /* :103-106 */
  for (int i = 0; i < 15; i++) {                       /* :108-114 */
    float temps = 1.0f / (1 << i);
    if (data > 0.99 * temps && data < 1.01 * temps) { vals = (int8_t)i; break; }
  }
  int8_t oups = (int8_t)(expand - vals);               /* :116 (char arithmetic) */
  if (oups < 0) oups = 0;                              /* :117-119 */
  if (sign) oups = (int8_t)(oups | 0x80);              /* :121-123 */
  return (uint8_t)oups;
}

/* a12 (filters)  LoadModel filter loop -- model_loader.cpp:154-171.
 * q_in[c], q_out[n] are the runtime's NEGATED Q values (quantization.cpp:46). */
void tf2o_encode_filters(const float* w, int N, int C, int FH, int FW,
                         const int8_t* q_in, const int8_t* q_out, uint8_t* codes) {
  for (int n = 0; n < N; n++)
    for (int c = 0; c < C; c++) {
      int q_fixed = q_in[c];
      int q_fixed_gap = q_out[n];
      int8_t expand = (int8_t)(INFLAT + q_fixed - q_fixed_gap);     /* :162 */
      for (int k = 0; k < FH * FW; k++) {
        size_t a = ((size_t)n * C + c) * FH * FW + k;
        codes[a] = tf2o_get_real(w[a], expand);
      }
    }
}

/* a12 (bias / BN fold) -- model_loader.cpp:175-232.  All intermediate types as
 * in the reference: float variables, double only where the C++ promotes. */
void tf2o_fold_bias_bn(int N, int bias_en, int bn_en, const float* bias,
                       const float* mean, const float* variance, float scale_factor,
                       const float* gamma, const float* betaf, const int8_t* q_out,
                       int32_t* bias_fix, int32_t* alpha_fix, int32_t* beta_fix) {
  for (int n = 0; n < N; n++) {
    int q_fixed_gap = q_out[n];
    float bias_trans_coe = (float)(1 << (INFLAT - q_fixed_gap));    /* :178,228 */
    if (bias_en) bias_fix[n] = (int32_t)(bias[n] * bias_trans_coe); /* :181 */
    else bias_fix[n] = 0;                                           /* :185 */
    float a, b, alpha_data, beta_data;
    float eps = 0.00001;                                            /* :221 */
    float mn = bn_en ? mean[n] : 0.0f, vr = bn_en ? variance[n] : 0.0f;
    float gm = bn_en ? gamma[n] : 0.0f, bt = bn_en ? betaf[n] : 0.0f;
    float sf = bn_en ? scale_factor : 0.0f;
    a = mn / sf;                                                    /* :223 */
    b = (float)sqrt((double)(vr / sf + eps));                       /* :224 */
    alpha_data = bn_en ? gm / b : 1.0f;                             /* :225 */
    beta_data = bn_en ? -(gm / b * a) + bt : 0.0f;                  /* :226 */
    alpha_fix[n] = (int32_t)(alpha_data * pow(2, ALPHA_INFLAT));    /* :230 */
    beta_fix[n] = (int32_t)(beta_data > 0 ? (bias_trans_coe * beta_data + 0.5)
                                          : (bias_trans_coe * beta_data - 0.5)); /* :231 */
  }
}

/* a13  filter_trans -- model_loader.cpp:25-96: one 7x7 filter plane of byte
 * codes -> nine 3x3 planes (the 27-channel rewrite of conv1).  `out` must have
 * been pre-set by the caller exactly as LoadModel does (memset 0, :246).      */
void tf2o_filter_trans(const uint8_t* in49, uint8_t* out81) {
  uint8_t media[2][7][7], width[3][7][7], heights[6][7][7];
  memset(media, 0x40, sizeof media);                                /* :28-34 */
  memset(width, 0x40, sizeof width);                                /* :39-45 */
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) media[j % 2][i][j / 2] = in49[i * 7 + j];   /* :47-51 */
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 7; i++)
      for (int j = 0; j < 3; j++) width[t][i][j] = media[t][i][j];  /* :53-59 */
  for (int i = 0; i < 7; i++) width[2][i][2] = media[0][i][3];      /* :61-63 */
  memset(heights, 0, sizeof heights);                               /* :67-73  (0x00, not 0x40: quirk C-4) */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 7; k++) heights[i * 2 + k % 2][k / 2][j] = width[i][k][j]; /* :75-81 */
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) out81[i * 9 + j * 3 + k] = heights[i][j][k];      /* :84-88 */
    if (i % 2 == 0)
      for (int k = 0; k < 3; k++) out81[(i / 2 + 6) * 9 + 2 * 3 + k] = heights[i][3][k]; /* :90-94 */
  }
}

/* LoadModel tail -- model_loader.cpp:244-257: rewrite layer 0's [N][3][7][7]
 * codes into [N][27][3][3]. */
void tf2o_conv1_rewrite(const uint8_t* codes7 /*[N][3][49]*/, int N, uint8_t* codes3 /*[N][27][9]*/) {
  memset(codes3, 0, (size_t)N * 27 * 9);                            /* :246 */
  for (int n = 0; n < N; n++)
    for (int c = 0; c < 3; c++)
      tf2o_filter_trans(codes7 + ((size_t)n * 3 + c) * 49, codes3 + (size_t)n * 243 + (size_t)c * 81);
}

/* a13  feature_trans -- input_loader.cpp:27-73: one 224x224 float plane ->
 * nine 114x114 planes (space-to-depth of the pad-3 image).  The reference
 * writes 115-stride planes and LoadInputImage crops to 114 (:108-115); the
 * result here is the cropped [9][114][114]. */
void tf2o_feature_trans(const float* in, float* out) {
  enum { D = 224, P = 3, ND = D + 2 * P, HD = ND / 2 /*115*/, OD = 114 };
  float* pad = (float*)calloc((size_t)ND * ND, sizeof(float));
  float* media = (float*)calloc((size_t)3 * ND * ND, sizeof(float));
  float* ht = (float*)calloc((size_t)6 * HD * HD, sizeof(float));
  float* fin = (float*)calloc((size_t)9 * HD * HD + 1024, sizeof(float));
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) pad[(i + P) * ND + j + P] = in[i * D + j];          /* :30-34 */
  for (int i = 0; i < ND; i++)
    for (int j = 0; j < ND; j++) media[((j % 2) * ND + i) * ND + j / 2] = pad[i * ND + j]; /* :37-41 */
  for (int i = 0; i < ND; i++)
    for (int j = 0; j < ND - 1; j++) media[(2 * ND + i) * ND + j] = media[(0 * ND + i) * ND + j + 1]; /* :43-47 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < ND; j++)
      for (int k = 0; k < HD; k++) ht[((i * 2 + j % 2) * HD + j / 2) * HD + k] = media[(i * ND + j) * ND + k]; /* :50-56 */
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < HD; j++)
      for (int k = 0; k < HD; k++) fin[(i * HD + j) * HD + k] = ht[(i * HD + j) * HD + k];   /* :58-64 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < HD; j++)
      for (int k = 0; k < HD; k++) {
        float v = (j + 1 < HD) ? ht[((i * 2) * HD + j + 1) * HD + k] : 0.0f;                 /* :66-72 */
        fin[((i + 6) * HD + j) * HD + k] = v;
      }
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < OD; j++)
      for (int k = 0; k < OD; k++) out[(i * OD + j) * OD + k] = fin[(i * HD + j) * HD + k];  /* :108-115 */
  free(pad); free(media); free(ht); free(fin);
}

/* a15  Quantization -- host/src/quantization.cpp:25-55.  `vals` are the ints of
 * the ASCII Q file in file order; q is [n_q_rows][max_c] int8, zero-filled by
 * the caller; tables are the per-layer k* arrays.  Returns #values consumed. */
int tf2o_q_table(const int32_t* vals, int n_vals, int num_layer, int num_conv, int max_c,
                 const int32_t* kOutputChannels, const int32_t* kIpoolEnable,
                 const int32_t* kInputLayer, const int32_t* kBranchTail,
                 const int32_t* kConcatLayer, const int32_t* kNStart, int8_t* q) {
  int offset = 0, pos = 0;
  for (int layer = 0; layer < num_layer + 1; layer++) {
    int conv_layer = layer == 0 ? 0 : layer - 1;
    int channel = layer == 0 ? 3 : kOutputChannels[conv_layer];
    for (int c = 0; c < channel; c++) {
      if (kIpoolEnable[conv_layer] == 1) {           /* 2 = this build's L2Norm row: has its own Q values */
        q[offset + c] = q[kInputLayer[conv_layer] * max_c + c];                 /* :42-43 */
      } else {
        int q_value = pos < n_vals ? vals[pos] : 0; pos++;                      /* :45 */
        q[offset + c] = (int8_t)(-q_value);                                     /* :46 */
        if (kBranchTail[conv_layer])
          q[(num_conv + 1 + kConcatLayer[conv_layer]) * max_c + kNStart[conv_layer] + c] = (int8_t)(-q_value); /* :47-49 */
      }
    }
    offset += max_c;
  }
  return pos;
}

/* a16  input quantisation -- host/src/runner.cpp:158-164 (q0 = runtime q[0],
 * i.e. the NEGATED channel-0 Q of the image row, used for every channel). */
void tf2o_quantize_input(const float* x, size_t n, int q0, int8_t* out) {
  float trans = q0 > 0 ? (1.0f / (1 << q0)) : (float)(1 << (-q0));
  for (size_t i = 0; i < n; i++) {
    float tmp = x[i] * trans;
    int tmp_int = (int)(tmp > 0 ? tmp + 0.5 : tmp - 0.5);
    out[i] = (int8_t)(tmp_int > REALMAX ? REALMAX : tmp_int < REALMIN ? REALMIN : tmp_int);
  }
}

/* ------------------------------------------------------------------------ */
/* a1  MUL -- device/src/pe.cl:27-40                                         */
/* ------------------------------------------------------------------------ */
static inline int32_t mul_code(int8_t feature, uint8_t filter) {
  if (filter & 0x40) return 0;                                 /* :28-30 */
  if (filter & 0x80) feature = (int8_t)(-feature);             /* :32-34  (-(-128) wraps to -128) */
  return (int32_t)((uint32_t)(int32_t)feature << (filter & 0x1f)); /* :36-37 */
}
int32_t tf2o_mul(int8_t feature, uint8_t filter) { return mul_code(feature, filter); }

/* a2/a3/a10  conv core -- pe.cl:42-49,144-180 + sequencer.cl:264-312 geometry:
 * acc[n,oh,ow] = bias[n] + sum_{c,fh,fw} MUL(x[c, oh*s+fh*d-p, ow*s+fw*d-p], code[n,c,fh,fw]),
 * zero padding, int32 wrap-around.  `dil` (dilation) is 1 for every reference
 * network; it is here for the SSD "next" row.  One image.                   */
void tf2o_conv(const int8_t* x, int C, int H, int W, const uint8_t* codes, const int32_t* bias,
               int N, int FH, int FW, int stride, int pad_h, int pad_w, int dil,
               int OH, int OW, int32_t* acc) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; n++) {
    uint32_t* a = (uint32_t*)(acc + (size_t)n * OH * OW);
    for (int i = 0; i < OH * OW; i++) a[i] = (uint32_t)bias[n];        /* conv_start ? bias : result, :176-180 */
    for (int c = 0; c < C; c++)
      for (int fh = 0; fh < FH; fh++)
        for (int fw = 0; fw < FW; fw++) {
          uint8_t code = codes[(((size_t)n * C + c) * FH + fh) * FW + fw];
          if (code & 0x40) continue;
          int sh = code & 0x1f, neg = code & 0x80;
          for (int oh = 0; oh < OH; oh++) {
            int ih = oh * stride + fh * dil - pad_h;
            if (ih < 0 || ih >= H) continue;                           /* sequencer.cl:287 zero pad */
            const int8_t* xr = x + ((size_t)c * H + ih) * W;
            uint32_t* ar = a + (size_t)oh * OW;
            int ow0 = 0, ow1 = OW;
            int off = fw * dil - pad_w;
            while (ow0 < OW && ow0 * stride + off < 0) ow0++;
            while (ow1 > ow0 && (ow1 - 1) * stride + off >= W) ow1--;
            if (neg) {
              for (int ow = ow0; ow < ow1; ow++) {
                int8_t f = (int8_t)(-xr[ow * stride + off]);
                ar[ow] += (uint32_t)(int32_t)f << sh;
              }
            } else {
              for (int ow = ow0; ow < ow1; ow++)
                ar[ow] += (uint32_t)(int32_t)xr[ow * stride + off] << sh;
            }
          }
        }
  }
}

/* a4  requant -- pe.cl:185-203 */
static inline int8_t requant1(int32_t acc, int32_t alpha, int32_t beta) {
  int64_t bn_alpha_inflat = (int64_t)acc * (int64_t)alpha;              /* :191 */
  int32_t bn_alpha = (int32_t)(bn_alpha_inflat >> ALPHA_INFLAT);        /* :192 */
  int32_t t = (int32_t)((uint32_t)bn_alpha + (uint32_t)beta);           /* wraps like the 32-bit adder */
  int32_t bn_data = ((t >> (INFLAT - 1)) + 1) >> 1;                     /* :193 */
  return (int8_t)(bn_data > REALMAX ? REALMAX : bn_data < REALMIN ? REALMIN : bn_data); /* :194 */
}
void tf2o_requant(const int32_t* acc, int N, int HW, const int32_t* alpha, const int32_t* beta,
                  int relu, int8_t* y) {
#pragma omp parallel for
  for (int n = 0; n < N; n++)
    for (int i = 0; i < HW; i++) {
      int8_t v = requant1(acc[(size_t)n * HW + i], alpha[n], beta[n]);
      if (relu && !(v > 0)) v = 0;                                      /* a5 relu.cl:54 */
      y[(size_t)n * HW + i] = v;
    }
}

/* a6/a7  pool + pool_tail -- pool.cl:152-260, pool_tail.cl:91-216.
 * The stream kernel keeps a running max over the last 3 columns, then the last
 * 3 rows, with ZERO history/out-of-range values, reading window slots s < S
 * only (the other slots stay 0, so for S < 3 the result is also max'ed with 0);
 * pool_tail drops 2-pad leading rows/cols and keeps every second one for stride
 * 2.  Net effect: out[ph,pw] = max_{i,j<S} xz[ph*st-pad+i, pw*st-pad+j]
 * (and 0 when S<3), xz = x extended by zeros.                                */
void tf2o_maxpool(const int8_t* x, int C, int H, int W, int S, int st, int pad,
                  int PH, int PW, int8_t* y) {
#pragma omp parallel for
  for (int c = 0; c < C; c++)
    for (int ph = 0; ph < PH; ph++)
      for (int pw = 0; pw < PW; pw++) {
        int8_t m = S < 3 ? 0 : REALMIN;
        for (int i = 0; i < S; i++)
          for (int j = 0; j < S; j++) {
            int h = ph * st - pad + i, w = pw * st - pad + j;
            int8_t v = (h < 0 || h >= H || w < 0 || w >= W) ? 0 : x[((size_t)c * H + h) * W + w];
            if (v > m) m = v;
          }
        y[((size_t)c * PH + ph) * PW + pw] = m;
      }
}

/* a8  residual add -- feature_writer.cl:119-122 */
void tf2o_residual_add(int8_t* y, const int8_t* res, size_t n, int relu) {
  for (size_t i = 0; i < n; i++) {
    int16_t addition = (int16_t)((int16_t)y[i] + (int16_t)res[i]);
    int8_t r = (int8_t)(addition > REALMAX ? REALMAX : addition < REALMIN ? REALMIN : addition);
    y[i] = (!relu || r > 0) ? r : 0;
  }
}

/* a9  global average -- full_size_pool.cl:95-125.  669 = round(2^15/49) is the
 * reference's hard-wired constant for 7x7; `mult` lets other map sizes use
 * round(2^15/(H*W)).                                        */
void tf2o_global_avg(const int8_t* x, int C, int HW, int mult, int8_t* y) {
  for (int c = 0; c < C; c++) {
    int16_t s = 0;
    for (int i = 0; i < HW; i++) s = (int16_t)(s + x[(size_t)c * HW + i]);   /* Sreal accumulate :101-112 */
    int32_t m = (((int32_t)s * mult) >> 14);
    m = (m + 1) >> 1;                                                         /* :118 */
    y[c] = (int8_t)(m > REALMAX ? REALMAX : m < REALMIN ? REALMIN : m);
  }
}

/* a17  Evaluation top-k -- host/src/network_helper.cpp:160-201: k bubble
 * passes with '>' over feature = out / (1 << Q)  => ties keep the LARGER index
 * on top.  q are the runtime (negated) values of the last layer's row.        */
void tf2o_topk(const int8_t* logits, const int8_t* q, int n, int k, int32_t* labels, float* feats) {
  float* f = (float*)malloc(sizeof(float) * n);
  int* lab = (int*)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) {
    float trans = (float)(1 << (-q[i]));                                      /* :181 */
    f[i] = logits[i] / trans; lab[i] = i;                                     /* :185 */
  }
  for (int i = 0; i < k; i++)
    for (int j = 0; j < n - i - 1; j++)
      if (f[j] > f[j + 1]) {                                                  /* :194 */
        float tf = f[j]; f[j] = f[j + 1]; f[j + 1] = tf;
        int tl = lab[j]; lab[j] = lab[j + 1]; lab[j + 1] = tl;
      }
  for (int i = 0; i < k; i++) { labels[i] = lab[n - i - 1]; if (feats) feats[i] = f[n - i - 1]; }
  free(f); free(lab);
}

/* ------------------------------------------------------------------------ */
/* Whole fused layer for a batch (conv -> requant -> relu -> pool -> residual
 * -> global avg), images in parallel; used by the CPU baseline and by the
 * end-to-end parity tests.  x: [B][C][H][W], y: [B][N][PH][PW] (or [B][N] when
 * endpool).  `res` may be NULL.  scratch is allocated internally.            */
/* ------------------------------------------------------------------------ */
typedef struct {
  int C, H, W, N, FH, FW, stride, pad_h, pad_w, dil, OH, OW;
  int relu, pool_en, pool_S, pool_st, pool_pad, PH, PW;
  int add_en, add_relu, endpool, endpool_mult;
} tf2o_layer_t;

/* Built for a baseline ISA (the library travels to the GPU box); the hot loop nest is cloned for wider SIMD and picked at load
 * time, so that the CPU baseline bench.py reports is the host's honest vector speed. */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void conv_one(const tf2o_layer_t* L, const int8_t* x, const uint8_t* codes, const int32_t* bias,
                     int n, uint32_t* a) {
  const int OH = L->OH, OW = L->OW, H = L->H, W = L->W, C = L->C;
  for (int i = 0; i < OH * OW; i++) a[i] = (uint32_t)bias[n];
  for (int c = 0; c < C; c++)
    for (int fh = 0; fh < L->FH; fh++)
      for (int fw = 0; fw < L->FW; fw++) {
        uint8_t code = codes[(((size_t)n * C + c) * L->FH + fh) * L->FW + fw];
        if (code & 0x40) continue;
        int sh = code & 0x1f, neg = code & 0x80;
        int off = fw * L->dil - L->pad_w, st = L->stride;
        int ow0 = 0, ow1 = OW;
        while (ow0 < OW && ow0 * st + off < 0) ow0++;
        while (ow1 > ow0 && (ow1 - 1) * st + off >= W) ow1--;
        for (int oh = 0; oh < OH; oh++) {
          int ih = oh * st + fh * L->dil - L->pad_h;
          if (ih < 0 || ih >= H) continue;
          const int8_t* xr = x + ((size_t)c * H + ih) * W + off;
          uint32_t* ar = a + (size_t)oh * OW;
          if (neg) { for (int ow = ow0; ow < ow1; ow++) ar[ow] += (uint32_t)(int32_t)(int8_t)(-xr[ow * st]) << sh; }
          else     { for (int ow = ow0; ow < ow1; ow++) ar[ow] += (uint32_t)(int32_t)xr[ow * st] << sh; }
        }
      }
}

void tf2o_layer(const tf2o_layer_t* L, int B, const int8_t* x, const uint8_t* codes,
                const int32_t* bias, const int32_t* alpha, const int32_t* beta,
                const int8_t* res, int8_t* y) {
  const int OHW = L->OH * L->OW, PHW = L->PH * L->PW;
  const size_t xsz = (size_t)L->C * L->H * L->W;
  int8_t* pre = NULL;  /* per-image pre-endpool buffer when endpool */
  if (L->endpool) pre = (int8_t*)malloc((size_t)B * L->N * PHW);
#pragma omp parallel
  {
    uint32_t* a = (uint32_t*)malloc(sizeof(uint32_t) * OHW);
    int8_t* t = (int8_t*)malloc(OHW);
#pragma omp for collapse(2) schedule(dynamic, 4)
    for (int b = 0; b < B; b++)
      for (int n = 0; n < L->N; n++) {
        conv_one(L, x + (size_t)b * xsz, codes, bias, n, a);
        for (int i = 0; i < OHW; i++) {
          int8_t v = requant1((int32_t)a[i], alpha[n], beta[n]);
          if (L->relu && !(v > 0)) v = 0;
          t[i] = v;
        }
        int8_t* dst = (L->endpool ? pre : y) + ((size_t)b * L->N + n) * PHW;
        if (L->pool_en) {
          tf2o_layer_t dummy; (void)dummy;
          for (int ph = 0; ph < L->PH; ph++)
            for (int pw = 0; pw < L->PW; pw++) {
              int8_t m = L->pool_S < 3 ? 0 : REALMIN;
              for (int i = 0; i < L->pool_S; i++)
                for (int j = 0; j < L->pool_S; j++) {
                  int h = ph * L->pool_st - L->pool_pad + i, w = pw * L->pool_st - L->pool_pad + j;
                  int8_t v = (h < 0 || h >= L->OH || w < 0 || w >= L->OW) ? 0 : t[h * L->OW + w];
                  if (v > m) m = v;
                }
              dst[ph * L->PW + pw] = m;
            }
        } else {
          memcpy(dst, t, PHW);
        }
        if (L->add_en) tf2o_residual_add(dst, res + ((size_t)b * L->N + n) * PHW, PHW, L->add_relu);
        if (L->endpool) tf2o_global_avg(dst, 1, PHW, L->endpool_mult, y + (size_t)b * L->N + n);
      }
    free(a); free(t);
  }
  if (pre) free(pre);
}

/* ------------------------------------------------------------------------ */
/* SSD conv4_3 L2Norm row                      */
/* TransForm_Kit/Quantization/models/SSD/layers/modules/l2norm.py:19-24:     */
/*   norm = sqrt(sum_c x^2) + 1e-10 ; out = weight[c] * (x / norm)           */
/* The reference runtime ships no SSD tables, so the INTEGER form of this    */
/* float op is defined here (and mirrored bit for bit by l2norm_kernel):     */
/* dequantise with the input's per-channel Q, evaluate in IEEE double in a   */
/* fixed operation order, requantise with the output row's Q and the input   */
/* rounding rule of runner.cpp:158-163 (half away from zero, clamp).         */
/*   qs = max_c Qx[c];  S = sum_c (x[c] << (qs - Qx[c]))^2   (exact integer) */
/*   norm = sqrt((double)S) * 2^-qs + 1e-10                                  */
/*   v = ((double)x[c] * 2^-Qx[c]) / norm * ((double)w[c] * 2^Qy[c])         */
/* x: int8 [C][HW] of one image; qx / qy: RUNTIME q rows (negated file Q).   */
/* ------------------------------------------------------------------------ */
void tf2o_l2norm(const int8_t* x, int C, int HW, const int8_t* qx, const int8_t* qy, const float* w, int8_t* y) {
  int qs = -128;
  for (int c = 0; c < C; c++) if (-(int)qx[c] > qs) qs = -(int)qx[c];
#pragma omp parallel for
  for (int p = 0; p < HW; p++) {
    int64_t S = 0;
    for (int c = 0; c < C; c++) {
      const int64_t v = (int64_t)x[(size_t)c * HW + p] << (qs + (int)qx[c]);
      S += v * v;
    }
    const double norm = sqrt((double)S) * ldexp(1.0, -qs) + 1e-10;
    for (int c = 0; c < C; c++) {
      const double a = ldexp(1.0, (int)qx[c]);                 /* 2^-Qx */
      const double b = (double)w[c] * ldexp(1.0, -(int)qy[c]); /* w * 2^Qy */
      const double t = ((double)x[(size_t)c * HW + p] * a) / norm;
      const double v = t * b;
      double r = v > 0 ? floor(v + 0.5) : ceil(v - 0.5);
      if (r > 127.0) r = 127.0;
      if (r < -128.0) r = -128.0;
      y[(size_t)c * HW + p] = (int8_t)r;
    }
  }
}
