/*
 * CPU ORACLE (plain C) for VPTQ's fused dequant+GEMV hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: loaded only by tests/, by
 * __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.  Nothing under
 * vptq_amd/ links or calls it.
 *
 * Restates (does not copy) the reference's pure-torch CPU path, same structure:
 *   vo_dequant : bit-stream unpack  (vptq/utils/pack.py:105-139)
 *                gather + residual add + outliers + perm + scale/bias
 *                                   (vptq/ops/quant_gemm.py:92-158)  -> dense W[O,I]
 *   vo_linear  : y = F.linear(x, W, bias)  (vptq/ops/quant_gemm.py:274)
 *   vo_forward : vo_dequant + vo_linear = VQuantLinear.forward on the fallback
 * Every 16-bit op rounds to nearest-even after each step, exactly like torch's
 * CPU half/bfloat16 kernels (widen to fp32, operate, narrow); the contraction
 * accumulates in double and rounds once.
 *
 * Pinned against the numpy oracle and, through it, against golden vectors of
 * the real reference (tests/test_oracle_golden.py, tests/test_oracle_c.py).
 * The layer descriptor is VptqLayerDesc (include/vptq_hip.h) with HOST pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vptq_hip.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define VO_API __attribute__((visibility("default")))

/* ---- 16-bit float <-> fp32, software, round-to-nearest-even ---- */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float h2f(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff;
  if (e == 0) {
    if (m == 0) return u2f(s);
    /* subnormal: value = m * 2^-24 */
    return (s ? -1.0f : 1.0f) * (float)m * 5.9604644775390625e-8f;
  }
  if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
  return u2f(s | ((e + 112) << 23) | (m << 13));
}

static inline uint16_t f2h(float f) {
  const uint32_t u = f2u(f);
  const uint32_t s = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0));
  if (a >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);          /* rounds to inf (>= 65520) */
  if (a < 0x33000001u) return (uint16_t)s;                       /* < 2^-25 (or == 2^-25: tie to even 0) */
  if (a < 0x38800000u) {                                         /* subnormal half */
    const int e = (int)(a >> 23);                                /* biased fp32 exponent, 102..112 */
    const uint32_t m = (a & 0x7fffffu) | 0x800000u;              /* 24-bit significand */
    const int shift = 126 - e;                                   /* 14..24: result = m >> shift */
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(s | r);
  }
  {
    uint32_t r = a - 0x38000000u;                                /* rebias exponent */
    const uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(s | r);
  }
}

static inline float b2f(uint16_t b) { return u2f((uint32_t)b << 16); }
static inline uint16_t f2b(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline float ld(uint16_t v, int dt) { return dt == VPTQ_DTYPE_F16 ? h2f(v) : b2f(v); }
static inline uint16_t st(float f, int dt) { return dt == VPTQ_DTYPE_F16 ? f2h(f) : f2b(f); }
static inline float rnd(float f, int dt) { return ld(st(f, dt), dt); }

VO_API uint16_t vo_f32_to_f16(float f) { return f2h(f); }
VO_API float vo_f16_to_f32(uint16_t h) { return h2f(h); }
VO_API uint16_t vo_f32_to_bf16(float f) { return f2b(f); }

/* element g of a packed row: bits [g*T, (g+1)*T) of the little-endian stream */
static inline uint32_t unpack_elem(const uint32_t* row, int g, int T) {
  const uint32_t bit = (uint32_t)g * (uint32_t)T, wi = bit >> 5, sh = bit & 31u;
  uint64_t w = row[wi];
  if (sh + (uint32_t)T > 32u) w |= (uint64_t)row[wi + 1] << 32;
  return (uint32_t)(w >> sh) & (T >= 32 ? 0xffffffffu : ((1u << T) - 1u));
}

/*
 * Dense W[O, I] (16-bit patterns).  res_mask_quirk != 0 reproduces the
 * reference CPU path's residual mask ((1 << index_bits) - 1, pack.py:137).
 */
VO_API int vo_dequant(const VptqLayerDesc* d, uint16_t* W, int res_mask_quirk) {
  const int I = d->in_features, O = d->out_features, v = d->vector_len;
  const int G = d->group_size, S = d->outlier_size, N = d->num_indices;
  const int T = d->index_bits + d->res_bits, dt = d->dtype;
  const uint32_t imask = (1u << d->index_bits) - 1u;
  const uint32_t rmask = d->res_bits ? ((1u << (res_mask_quirk ? d->index_bits : d->res_bits)) - 1u) : 0u;
  const uint16_t* cent = (const uint16_t*)d->centroids;
  const uint16_t* rcent = (const uint16_t*)d->res_centroids;
  const uint16_t* ocent = (const uint16_t*)d->outlier_centroids;
  const uint16_t* scale = (const uint16_t*)d->weight_scale;
  const uint16_t* wbias = (const uint16_t*)d->weight_bias;
  /* column c of the quantised matrix lands in output column perm[c] */
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    for (int c = 0; c < I; ++c) {
      const int j = d->perm ? (int)d->perm[c] : c;
      float w[16];
      if (c < S) {
        const int ov = d->outlier_vector_len;
        for (int t = 0; t < v; ++t) {
          const int o = n * v + t;
          w[t] = 0.f;
          if (o < O) {
            const int m = o / ov, tt = o - m * ov;
            w[t] = ld(ocent[(size_t)d->outlier_indices[(size_t)m * S + c] * ov + tt], dt);
          }
        }
      } else {
        const int cc = c - S, cb = cc / G, g = cc - cb * G;
        const uint32_t* row = (const uint32_t*)d->indices + ((size_t)cb * N + n) * d->row_words;
        const uint32_t e = unpack_elem(row, g, T);
        const uint16_t* cp = cent + ((size_t)cb * d->num_centroids + (e & imask)) * v;
        for (int t = 0; t < v; ++t) w[t] = ld(cp[t], dt);
        if (rmask) {
          const uint16_t* rp =
              rcent + ((size_t)cb * d->num_res_centroids + ((e >> d->index_bits) & rmask)) * v;
          for (int t = 0; t < v; ++t) w[t] = rnd(w[t] + ld(rp[t], dt), dt);
        }
      }
      if (scale) {
        const float s = ld(scale[j], dt), b = ld(wbias[j], dt);
        for (int t = 0; t < v; ++t) w[t] = rnd(rnd(w[t] * s, dt) + b, dt);
      }
      for (int t = 0; t < v; ++t) {
        const int o = n * v + t;
        if (o < O) W[(size_t)o * I + j] = st(w[t], dt);
      }
    }
  }
  return 0;
}

/* y[tokens, O] = x[tokens, I] @ W[O, I]^T + bias : double accumulate, one rounding */
VO_API int vo_linear(const uint16_t* W, const uint16_t* x, const uint16_t* bias, uint16_t* y,
                     int tokens, int I, int O, int dt) {
  float* xf = (float*)malloc((size_t)tokens * I * sizeof(float));
  if (!xf) return -1;
  for (size_t i = 0; i < (size_t)tokens * I; ++i) xf[i] = ld(x[i], dt);
#pragma omp parallel for schedule(static)
  for (int o = 0; o < O; ++o) {
    const uint16_t* wr = W + (size_t)o * I;
    for (int t = 0; t < tokens; ++t) {
      const float* xr = xf + (size_t)t * I;
      double acc = 0.0;
      for (int i = 0; i < I; ++i) acc += (double)xr[i] * (double)ld(wr[i], dt);
      if (bias) acc += (double)ld(bias[o], dt);
      y[(size_t)t * O + o] = st((float)acc, dt);
    }
  }
  free(xf);
  return 0;
}

/* VQuantLinear.forward on the reference's torch fallback: dense W, then linear.
 * W_scratch: O*I uint16 provided by the caller (so timing excludes malloc). */
VO_API int vo_forward(const VptqLayerDesc* d, const uint16_t* x, uint16_t* y, int tokens,
                      uint16_t* W_scratch, int res_mask_quirk) {
  int rc = vo_dequant(d, W_scratch, res_mask_quirk);
  if (rc) return rc;
  return vo_linear(W_scratch, x, (const uint16_t*)d->bias, y, tokens, d->in_features,
                   d->out_features, d->dtype);
}

VO_API int vo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

VO_API void vo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
