// Host-side helper shared by every translation unit of the library.
#pragma once
#include <stdlib.h>

namespace vptq {

// Tuning / A-B knobs (VPTQ_K256_KERNEL, VPTQ_SLICED_SLICES, ...; tools/README.md lists them) are read only when VPTQ_TUNING=1 is set:
// the library's behaviour does not depend on stray environment variables.  (The package's four product knobs - VPTQ_ARITHMETIC,
// VPTQ_SLICED_LAYOUT, VPTQ_FUSED_GEMM_MAX_TOKENS, VPTQ_HIP_LIB - are read on the Python side.)
static inline const char* tune_env(const char* name) {
  static const bool on = [] { const char* e = getenv("VPTQ_TUNING"); return e && e[0] == '1'; }();
  return on ? getenv(name) : nullptr;
}

}  // namespace vptq
