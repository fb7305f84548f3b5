// opts.h -- TF2_AMD_OPTS: the ONE run-time option string of the library (round 5; rounds 1-4 had grown 56 separate TF2_AMD_*
// environment variables, read with getenv() in packers, planners and launchers).
//
//   TF2_AMD_OPTS="name=value,name=value,flag"        (a bare name means name=1)
//
// Parsed ONCE per tf2_net_create / tf2_net_reload_options into an immutable snapshot that every later read takes from (no getenv()
// anywhere else: launchers run on several feeder threads, and getenv racing with a setenv is undefined behaviour).  Unknown names are
// an error (TF2_ERR_ARG from create / reload), and so is a TEST-ONLY option unless TF2_AMD_TEST=1 is set as well: those force
// kernels, disable proofs or lower thresholds for the test-suite and the A/B tools and are no part of the product's interface.
// The table of names is in opts.cpp (kOptSpecs); INTEGRATION.md section 5 documents the product options.
#pragma once
#include <string>

namespace tf2 {

long long opt(const char* name, long long dflt);      // value of `name` in the current snapshot, dflt when it is not set
inline bool opt_set(const char* name) { return opt(name, 0) != 0; }
std::string opts_reload();                            // re-read TF2_AMD_OPTS / TF2_AMD_TEST; error text, empty = ok
std::string opts_describe();                          // "name (product|test-only): doc" per line

}  // namespace tf2
