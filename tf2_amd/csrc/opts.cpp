// opts.cpp -- the option table and parser behind TF2_AMD_OPTS (opts.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <unistd.h>
#include <map>
#include <memory>
#include <mutex>
#include "opts.h"

extern char** environ;

namespace tf2 {

namespace {
struct OptSpec { const char* name; int test_only; const char* doc; };
const OptSpec kOptSpecs[] = {
  // ---- product options ----
  {"alt_conc", 0, "how Net::run decides that batches are in flight: 0 never, 1 always, 2 (default) the caller's tf2_net_run_ex statement, else calls on >= 2 streams among the last eight"},
  {"bgroup", 0, "group launches (conv_bgroup.hip: eight co-resident blocks per image that meet inside the kernel) one batch at a time: 1 (default) / 0"},
  {"bfirst", 0, "the first bottleneck of the 56 x 56 stage (shortcut | reduce, 3x3, expand + residual) as one launch of independent row bands (conv_bfirst.hip): 0 never, 1 (default) with batches in flight, 2 one batch at a time as well"},
  {"bband", 0, "band launches (conv_bband.hip) of identity bottlenecks: 0 never, 1 (default) with batches in flight, 2 one batch at a time as well"},
  {"c3", 0, "3x3 / 1 / pad 1 layers of big maps on conv_c3.hip: 1 (default), 0 the ring kernel; 2 / 3 force 64- / 128-channel blocks (tests)"},
  {"fc", 0, "whole-window layers at batch <= 32 on conv_fc.hip (weight stream): 1 (default) / 0"},
  {"fc4", 0, "pack time: conv_fc layers keep their filters as 4-bit codes in HBM (expanded in registers): 1 (default) / 0 int8 window tiles"},
  {"share", 0, "pack time: alternative tile heights share the main entry's weight tiles: 1 (default: the wide ones), 2 all, 0 none"},
  // ---- test-only: forced kernels, disabled proofs, thresholds (need TF2_AMD_TEST=1) ----
  {"merge", 1, "pack time: 0 = no merged rows (a 1x1 row and the 3x3 row behind it as one 3x3 layer)"},
  {"nodbl", 1, "pack time: no doubled channels"}, {"nofuse", 1, "pack time: no conv_bneck pairs"}, {"nodual", 1, "pack time: two-window layers in the Horner form"},
  {"nofast", 1, "pack time: generic requantisation everywhere"}, {"nosemi", 1, "pack time: no SEMI requantisation"}, {"nounit", 1, "pack time: conv1's low window as a window"},
  {"no4bit", 1, "pack time: shift-kernel layers keep int32 weights"}, {"im2col0", 1, "0: a 3x3 first layer on 3 channels keeps its plain form"},
  {"pw", 1, "conv_pw: 1 auto, 0 never"}, {"pw_slabs", 1, "conv_pw: most K slabs"}, {"pw_minpix", 1, "conv_pw: fewest pixels"},
  {"sk_kb", 1, "split-K launches of a few blocks split K over blocks as well (batch 1-4): 1 (default) / 0"}, {"sk_kb_blocks", 1, "... largest grid that takes it (default 8)"}, {"sk_kb_max", 1, "... most blocks per output tile (default 8)"}, {"sk_kb_min", 1, "... fewest (default 8: only slab lists long enough for eight parts)"}, {"stem_pk_small", 1, "conv_stem_pool_kernel at small batches: fewer pooled rows per block (>= ~192 blocks): 1 (default) / 0"}, {"q128", 1, "the input preparation reports -128s per image to conv_stem_pool_kernel (no scan of its input tile): 1 (default) / 0"}, {"pwk", 1, "conv_pwk (1x1 rows of 128 / 256 / 512 input channels: a block's weight fragments resident in registers, pixel tiles streamed through LDS): 0 never, 1 (default) with batches in flight, 2 one batch at a time as well"}, {"pwk_minpix", 1, "conv_pwk: fewest pixels"}, {"pwk_sk", 1, "conv_pwk: split-K rows as well"}, {"pwk_rows", 1, "bit mask of rows forced onto conv_pwk where the kernel can run them at all"}, {"nopwk_rows", 1, "... kept off it"}, {"pwk_slabs", 1, "conv_pwk: most K slabs of a row (2, 4 or 8)"}, {"pwk_units", 1, "conv_pwk: fewest (tile, channel part) units of a row (default 512)"}, {"pwk_pipe", 1, "conv_pwk: 0 = no requantisation between the next column group's MFMAs"}, {"pwk_slots", 1, "conv_pwk: blocks a launch aims at (0: 256, one per CU)"},
  {"sk", 1, "split-K kernel: 0 auto, 1 forced, 2 never"}, {"sk8", 1, "largest split-K grid in the 8-wave form"}, {"sk_s3", 1, "largest split-K grid on three ring stages, one batch at a time"}, {"sk_s3_conc", 1, "... with batches in flight"},
  {"fc_min", 1, "conv_fc: shortest K in slabs"}, {"c3_min", 1, "conv_c3: smallest grid"}, {"c3_min_hw", 1, "conv_c3: smallest map side (default 14)"}, {"c3_min256", 1, "conv_c3: smallest grid of 256-channel blocks"},
  {"fire", 1, "a fire module (squeeze + merged expands) as one launch: 0 never, 1 wherever it fits, 2 (default) on maps >= 28 wide"},
  {"fire_pool", 1, "fire modules with a pool behind their expands: 0 never, 1 / 2 fire launch + pool launch (maps >= 56 wide / wherever fire allows), 3 (default) the pool inside the fire launch one batch at a time, 4 always"},
  {"first", 1, "a 3x3 / stride 1 first layer on the image in one launch with its input preparation: 1 / 0"},
  {"first_pool", 1, "... a 3x3 first layer (stride 1 / 2) with a 3x3 / 2 max pool behind its ReLU, the pool included: 1 / 0"},
  {"c3_pool", 1, "a layer's 2x2 / 2 max pool inside its conv_c3 launch: 1 / 0"},
  {"c3_w9", 1, "conv_c3_w9_kernel: 0 never, 1 auto, 2 wherever allowed"},
  {"bneck_min", 1, "conv_bneck: smallest grid"}, {"stem", 1, "conv_stem: 1 auto, 0 never"}, {"stem_pool", 1, "conv1's pool in its launch"},
  {"avg_fuse", 1, "a layer's global average in its split-K launch: 0 never, 1 always, 2 (default) one batch at a time only"}, {"pair", 1, "pair launches"},
  {"bgroup_min7", 1, "smallest batch of the 7x7 group launches"}, {"bgroup_min14", 1, "... 14x14"}, {"bgroup_min28", 1, "... 28x28"}, {"bgroup_min56f", 1, "... the first 56x56 bottleneck"},
  {"bgroup_chain", 1, "identity bottlenecks per group launch"},
  {"bgroup_polls", 1, "group launches: polls of a meeting before it is reported as failed (default 1 << 24)"},
  {"bgroup_withhold", 1, "group launches: 1 + index of a block that leaves its group at kernel entry (the failure path of tf2_net_poll_error)"},
  {"bband_rows", 1, "band launches: rows per block with batches in flight"}, {"bband_rows_alone", 1, "... one batch at a time"},
  {"bband_rows_dd", 1, "band launches: most rows per block of a 28x28 bottleneck with two-window reduce and 3x3 (default 7; rounds 4-5: 4)"},
  {"bfirst_min", 1, "conv_bfirst: smallest batch"}, {"bband_min", 1, "band launches: smallest batch"}, {"bband_alone_maps", 1, "maps taking band launches one batch at a time (bit 1: 28x28, bit 2: 14x14)"},
  {"dense", 1, "arithmetic gather words"}, {"dense_max", 1, "longest slab list that takes them on multi-round grids"},
  {"alt_min", 1, "smallest grid taking a wide-tile alternative"}, {"alt_min_conc", 1, "... with batches in flight"}, {"alt_narrow", 1, "largest grid taking a narrow alternative"},
  {"alt_rows", 1, "bit mask of rows forced onto their alternative tile height"}, {"noalt_rows", 1, "... kept off it"},
  {"sk_rows", 1, "bit mask of 64-row-tile rows forced onto the in-block split-K kernel"}, {"nosk_rows", 1, "... kept off it"},
  {"exp", 1, "timing-probe bits (-DTF2_PROBES builds)"}, {"skip_layers", 1, "lo-hi: launches left out (-DTF2_PROBES builds)"},
  {"dbgptr", 1, "tools: device buffer for per-layer stamps"}, {"dbgptr2", 1, "tools: device buffer for per-block stamps"}, {"dbglayer", 1, "tools: the layer dbgptr2 records"},
};

using Snapshot = std::map<std::string, long long>;
std::mutex g_mu;
std::shared_ptr<const Snapshot> g_snap = std::make_shared<Snapshot>();

const OptSpec* find_spec(const std::string& n) {
  for (const OptSpec& s : kOptSpecs) if (n == s.name) return &s;
  return nullptr;
}
}  // namespace

long long opt(const char* name, long long dflt) {
  const std::shared_ptr<const Snapshot> s = std::atomic_load(&g_snap);
  auto it = s->find(name);
  return it == s->end() ? dflt : it->second;
}

std::string opts_reload() {
  std::lock_guard<std::mutex> lock(g_mu);
  auto snap = std::make_shared<Snapshot>();
  const char* env = getenv("TF2_AMD_OPTS");
  const char* test = getenv("TF2_AMD_TEST");
  const bool test_mode = test && atoi(test) != 0;
  std::string err;
  if (env) {
    std::string text(env);
    size_t pos = 0;
    while (pos <= text.size()) {
      size_t end = text.find_first_of(",; ", pos);
      if (end == std::string::npos) end = text.size();
      std::string item = text.substr(pos, end - pos);
      pos = end + 1;
      if (item.empty()) continue;
      std::string name = item, val = "1";
      const size_t eq = item.find('=');
      if (eq != std::string::npos) { name = item.substr(0, eq); val = item.substr(eq + 1); }
      if (eq != std::string::npos && val.empty()) { err = "TF2_AMD_OPTS: '" + item + "' has no value (a bare name means name=1)"; break; }
      const OptSpec* sp = find_spec(name);
      if (!sp) { err = "TF2_AMD_OPTS: unknown option '" + name + "'"; break; }
      if (sp->test_only && !test_mode) { err = "TF2_AMD_OPTS: '" + name + "' is a test-only option (set TF2_AMD_TEST=1 as well)"; break; }
      long long v;
      if (name == "skip_layers") {                        // lo-hi -> lo << 32 | hi
        int lo = 0, hi = -1;
        if (sscanf(val.c_str(), "%d-%d", &lo, &hi) != 2) { err = "TF2_AMD_OPTS: skip_layers wants lo-hi"; break; }
        v = ((long long)lo << 32) | (unsigned)hi;
      } else {
        char* endp = nullptr;
        v = (long long)strtoull(val.c_str(), &endp, 0);
        if (val[0] == '-') v = strtoll(val.c_str(), &endp, 0);
        if (!endp || *endp) { err = "TF2_AMD_OPTS: '" + item + "' is not name=integer"; break; }
      }
      (*snap)[name] = v;
    }
  }
  // Rounds 1-4 read 56 separate TF2_AMD_* variables (TF2_AMD_BGROUP=0, TF2_AMD_ALT_CONC, ...).  A deployment that still sets one of
  // them -- e.g. to keep the group launches off where their preconditions cannot be guaranteed -- must not silently get the default
  // back: any TF2_AMD_* name other than the four the library reads is an error that names its replacement.
  if (err.empty()) {
    static const char* const kKnown[] = {"TF2_AMD_OPTS", "TF2_AMD_TEST", "TF2_AMD_LIB", "TF2_AMD_TOOL_LIB"};
    for (char** e = environ; e && *e; e++) {
      if (strncmp(*e, "TF2_AMD_", 8) != 0) continue;
      const char* eqp = strchr(*e, '=');
      const std::string nm(*e, eqp ? (size_t)(eqp - *e) : strlen(*e));
      bool known = false;
      for (const char* k : kKnown) known = known || nm == k;
      if (known) continue;
      std::string low = nm.substr(8);
      for (char& c : low) c = (char)tolower((unsigned char)c);
      err = "environment variable " + nm + " is no longer read: use TF2_AMD_OPTS=\"" + low + "=" + (eqp ? eqp + 1 : "1") + "\"" +
            (find_spec(low) ? "" : " (INTEGRATION.md section 5 lists the current names)");
      break;
    }
  }
  if (!err.empty()) return err;                           // (the previous snapshot stays)
  std::atomic_store(&g_snap, std::shared_ptr<const Snapshot>(snap));
  return std::string();
}

std::string opts_describe() {
  std::string s;
  for (const OptSpec& o : kOptSpecs) s += std::string(o.name) + (o.test_only ? " (test-only): " : " (product): ") + o.doc + "\n";
  return s;
}

}  // namespace tf2
