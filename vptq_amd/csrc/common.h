// Device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vptq_hip.h"
#include "tune_env.h"

// The 16-bit arithmetic below must round after EVERY op (that is what the
// reference CPU path does); never let the compiler fuse a*b+c into one fma.
#pragma clang fp contract(off)

namespace vptq {



typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef short s4_t __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// ---- 16-bit element types: a "pair" is two elements in one 32-bit register ----
struct F16 {
  static constexpr int kDtype = VPTQ_DTYPE_F16;
  static constexpr uint32_t kInfBits = 0x7c00u;      // magnitude bits of infinity; anything above: NaN
  static constexpr float kMaxFinite = 65504.f;
  // packed pair ops: one VALU instruction each (v_pk_add_f16 / v_pk_mul_f16)
  static __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {
    h2_t r = __builtin_bit_cast(h2_t, a) + __builtin_bit_cast(h2_t, b);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint32_t mul2(uint32_t a, uint32_t b) {
    h2_t r = __builtin_bit_cast(h2_t, a) * __builtin_bit_cast(h2_t, b);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ float lo(uint32_t p) {
    return (float)__builtin_bit_cast(h2_t, p).x;
  }
  static __device__ __forceinline__ float hi(uint32_t p) {
    return (float)__builtin_bit_cast(h2_t, p).y;
  }
  static __device__ __forceinline__ float to_float(uint16_t b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);  // RNE
  }
  // acc + lo(p)*xf : lowers to v_fma_mix_f32 (f16 source, f32 accumulate)
  static __device__ __forceinline__ float fma_lo(uint32_t p, float xf, float acc) {
    return __builtin_fmaf(lo(p), xf, acc);
  }
  static __device__ __forceinline__ float fma_hi(uint32_t p, float xf, float acc) {
    return __builtin_fmaf(hi(p), xf, acc);
  }
  // ---- "broadcast one half of a pair register" forms: the half select folds into
  // the instruction's op_sel bits, so no splat register / extra VALU op is needed.
  static __device__ __forceinline__ float half_of(uint32_t pair, int h) {
    const h2_t p = __builtin_bit_cast(h2_t, pair);
    return (float)(h ? p.y : p.x);
  }
  static __device__ __forceinline__ h2_t bcast(uint32_t pair, int h) {
    const h2_t p = __builtin_bit_cast(h2_t, pair);
    return h ? __builtin_shufflevector(p, p, 1, 1) : __builtin_shufflevector(p, p, 0, 0);
  }
  static __device__ __forceinline__ uint32_t mul2_bcast(uint32_t a, uint32_t pair, int h) {
    const h2_t r = __builtin_bit_cast(h2_t, a) * bcast(pair, h);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint32_t add2_bcast(uint32_t a, uint32_t pair, int h) {
    const h2_t r = __builtin_bit_cast(h2_t, a) + bcast(pair, h);
    return __builtin_bit_cast(uint32_t, r);
  }
  // acc + lo/hi(w) * half h of an f16 pair register: one v_fma_mix_f32
  static __device__ __forceinline__ float fma_lo_h(uint32_t w, uint32_t xpair, int h, float acc) {
    return __builtin_fmaf(lo(w), half_of(xpair, h), acc);
  }
  static __device__ __forceinline__ float fma_hi_h(uint32_t w, uint32_t xpair, int h, float acc) {
    return __builtin_fmaf(hi(w), half_of(xpair, h), acc);
  }
  // c + lo(a)*lo(b) + hi(a)*hi(b) in fp32: v_dot2_f32_f16
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
  }
  // the forms the GEMV kernels' reference roundings use (bf16 has a shorter route for them, see BF16): here the same instructions
  static __device__ __forceinline__ uint32_t add2_g(uint32_t a, uint32_t b) { return add2(a, b); }
  static __device__ __forceinline__ uint32_t mul2_bcast_g(uint32_t a, uint32_t pair, int h) { return mul2_bcast(a, pair, h); }
  static __device__ __forceinline__ uint32_t add2_bcast_g(uint32_t a, uint32_t pair, int h) { return add2_bcast(a, pair, h); }
  static __device__ __forceinline__ float fma_lo_h_g(uint32_t w, uint32_t xpair, int h, float acc) { return fma_lo_h(w, xpair, h, acc); }
  static __device__ __forceinline__ float fma_hi_h_g(uint32_t w, uint32_t xpair, int h, float acc) { return fma_hi_h(w, xpair, h, acc); }
  // ... on four packed pairs (= 8 elements of one vector) at a time: w = f16(w + r); w = f16(f16(w * s) + b), s / b = half sh / bh of
  // a pair register; acc[0..7] += w * x, x = half xh of a pair register
  static __device__ __forceinline__ void add4(uint32_t (&w)[4], const uint32_t (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = add2(w[p], r[p]);
  }
  static __device__ __forceinline__ void scale_bias4(uint32_t (&w)[4], uint32_t spair, int sh, uint32_t bpair, int bh) {
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = mul2_bcast(w[p], spair, sh);
#pragma unroll
    for (int p = 0; p < 4; ++p) w[p] = add2_bcast(w[p], bpair, bh);
  }
  static __device__ __forceinline__ void fma4(float* acc, const uint32_t (&w)[4], uint32_t xpair, int xh) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      acc[2 * p] = fma_lo_h(w[p], xpair, xh, acc[2 * p]);
      acc[2 * p + 1] = fma_hi_h(w[p], xpair, xh, acc[2 * p + 1]);
    }
  }
  // 16 independent 4x4x4 products, fp32 accumulate: v_mfma_f32_4x4x4_16b_f16.  Block b =
  // lanes 4b..4b+3; lane i supplies row i of the first matrix and column i of the second,
  // and receives column i of the result (register r = row r).
  static __device__ __forceinline__ f32x4 mfma4(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h4_t, a), __builtin_bit_cast(h4_t, b),
                                              c, 0, 0, 0);
  }
};

struct BF16 {
  static constexpr int kDtype = VPTQ_DTYPE_BF16;
  static constexpr uint32_t kInfBits = 0x7f80u;
  static constexpr float kMaxFinite = 3.3895314e38f;
  static __device__ __forceinline__ float lo(uint32_t p) { return __uint_as_float(p << 16); }
  static __device__ __forceinline__ float hi(uint32_t p) {
    return __uint_as_float(p & 0xffff0000u);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) {
    return __uint_as_float((uint32_t)b << 16);
  }
  // fp32 -> bf16 RNE in hardware: v_cvt_pk_bf16_f32 (gfx950)
  static __device__ __forceinline__ uint32_t pack(float l, float h) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {l, h};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return (uint16_t)pack(f, 0.f);
  }
  // torch's CPU bf16 ops: widen to fp32, operate, round back (RNE)
  static __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {
    return pack(lo(a) + lo(b), hi(a) + hi(b));
  }
  static __device__ __forceinline__ uint32_t mul2(uint32_t a, uint32_t b) {
    return pack(lo(a) * lo(b), hi(a) * hi(b));
  }
  static __device__ __forceinline__ float fma_lo(uint32_t p, float xf, float acc) {
    return __builtin_fmaf(lo(p), xf, acc);
  }
  static __device__ __forceinline__ float fma_hi(uint32_t p, float xf, float acc) {
    return __builtin_fmaf(hi(p), xf, acc);
  }
  static __device__ __forceinline__ float half_of(uint32_t pair, int h) {
    return h ? hi(pair) : lo(pair);
  }
  static __device__ __forceinline__ uint32_t mul2_bcast(uint32_t a, uint32_t pair, int h) {
    const float s = half_of(pair, h);
    return pack(lo(a) * s, hi(a) * s);
  }
  static __device__ __forceinline__ uint32_t add2_bcast(uint32_t a, uint32_t pair, int h) {
    const float s = half_of(pair, h);
    return pack(lo(a) + s, hi(a) + s);
  }
  static __device__ __forceinline__ float fma_lo_h(uint32_t w, uint32_t xpair, int h, float acc) {
    return __builtin_fmaf(lo(w), half_of(xpair, h), acc);
  }
  static __device__ __forceinline__ float fma_hi_h(uint32_t w, uint32_t xpair, int h, float acc) {
    return __builtin_fmaf(hi(w), half_of(xpair, h), acc);
  }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_fmaf(lo(a), lo(b), __builtin_fmaf(hi(a), hi(b), c));
  }
  // ---- the reference's widened roundings WITHOUT unpacking (round 6).  v_dot2_f32_bf16 computes a.lo b.lo + a.hi b.hi + c in fp32
  // from packed bf16 pairs: with b = (s, 0) that is lo(a) * s, with b = (1, 0) and c = widen(t) it is lo(a) + t, with b = (x, 0) and
  // c = acc the multiply-add of the product.  tools/dot2_bf16_probe.hip, 4 M operands per family: products and sums round to the
  // SAME bf16 as the widened arithmetic on checkpoint-like, reference-test, wide-exponent, tie and raw finite operands (a sum of two
  // bf16 values is exact in fp32 or far from a bf16 tie; a product of two is exact); different: the sign of an exact zero, denormal
  // results (flushed), and an infinite / NaN OTHER half of the pair (x 0 = NaN) - none of which a weight can carry into a sum that
  // matters; the accumulate form differs from fmaf in the last fp32 bit now and then.  dequant.hip - whose W must be the reference's
  // bit for bit, zeros' signs included - keeps the widened forms above.
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ float dotp(uint32_t a, uint32_t b, float c) {
    // (the builtin, not inline assembly: a dot instruction's result needs wait states before a VALU read, which the compiler inserts
    // only for instructions it sees)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), c, false);
  }
  // (1, 0) / (0, 1), hidden from the compiler: as a known constant it becomes the inline operand `1.0`, which the instruction reads
  // as the 32-bit pattern 0x3f800000 = (0, 1) - the probe's first version added the wrong half
  static __device__ __forceinline__ uint32_t one_lo() { uint32_t v = 0x00003f80u; asm("" : "+v"(v)); return v; }
  static __device__ __forceinline__ uint32_t one_hi() { uint32_t v = 0x3f800000u; asm("" : "+v"(v)); return v; }
  static __device__ __forceinline__ uint32_t add2_g(uint32_t a, uint32_t b) {
    return pack(dotp(a, one_lo(), lo(b)), dotp(a, one_hi(), hi(b)));
  }
  static __device__ __forceinline__ uint32_t mul2_bcast_g(uint32_t a, uint32_t pair, int h) {
    const uint32_t s0 = h ? pair >> 16 : pair & 0xffffu;          // (s, 0)
    const uint32_t s1 = h ? pair & 0xffff0000u : pair << 16;      // (0, s)
    return pack(dotp(a, s0, 0.f), dotp(a, s1, 0.f));
  }
  static __device__ __forceinline__ uint32_t add2_bcast_g(uint32_t a, uint32_t pair, int h) {
    const float t = half_of(pair, h);
    return pack(dotp(a, one_lo(), t), dotp(a, one_hi(), t));
  }
  static __device__ __forceinline__ float fma_lo_h_g(uint32_t w, uint32_t xpair, int h, float acc) {
    return dotp(w, h ? xpair >> 16 : xpair & 0xffffu, acc);
  }
  static __device__ __forceinline__ float fma_hi_h_g(uint32_t w, uint32_t xpair, int h, float acc) {
    return dotp(w, h ? xpair & 0xffff0000u : xpair << 16, acc);
  }
  // ---- the same on four packed pairs at a time, as ONE scheduled block each.  The builtin above lowers to v_dot2c_f32_bf16 (VOP2:
  // D += a . b), which costs a v_mov per use to seed D, and the dot instructions' hazards (3 wait states before another VALU
  // instruction reads a dot's result, 4 before one overwrites it) become s_nop between every dependent pair: gemv_sliced<EX> came out
  // at 48 instructions + 8 s_nop per element against 56 widened - and no faster.  Here: the three-source form v_dot2_f32_bf16,
  // stage-major over the four pairs so that every consumer is at least 3 instructions behind its producer; each block ends on 4
  // non-dot instructions (or s_nop 3) so that whatever the compiler places behind it is safe.  8 dots + 4 conversions per stage.
  static __device__ __forceinline__ void add4(uint32_t (&w)[4], const uint32_t (&r)[4]) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    const uint32_t ol = one_lo(), oh = one_hi();
    asm("v_lshlrev_b32 %[t0], 16, %[r0]\n\tv_and_b32 %[t1], 0xffff0000, %[r0]\n\t"
        "v_lshlrev_b32 %[t2], 16, %[r1]\n\tv_and_b32 %[t3], 0xffff0000, %[r1]\n\t"
        "v_lshlrev_b32 %[t4], 16, %[r2]\n\tv_and_b32 %[t5], 0xffff0000, %[r2]\n\t"
        "v_lshlrev_b32 %[t6], 16, %[r3]\n\tv_and_b32 %[t7], 0xffff0000, %[r3]\n\t"
        "v_dot2_f32_bf16 %[t0], %[w0], %[ol], %[t0]\n\tv_dot2_f32_bf16 %[t1], %[w0], %[oh], %[t1]\n\t"
        "v_dot2_f32_bf16 %[t2], %[w1], %[ol], %[t2]\n\tv_dot2_f32_bf16 %[t3], %[w1], %[oh], %[t3]\n\t"
        "v_dot2_f32_bf16 %[t4], %[w2], %[ol], %[t4]\n\tv_dot2_f32_bf16 %[t5], %[w2], %[oh], %[t5]\n\t"
        "v_dot2_f32_bf16 %[t6], %[w3], %[ol], %[t6]\n\tv_dot2_f32_bf16 %[t7], %[w3], %[oh], %[t7]\n\t"
        "v_cvt_pk_bf16_f32 %[w0], %[t0], %[t1]\n\tv_cvt_pk_bf16_f32 %[w1], %[t2], %[t3]\n\t"
        "v_cvt_pk_bf16_f32 %[w2], %[t4], %[t5]\n\tv_cvt_pk_bf16_f32 %[w3], %[t6], %[t7]"
        : [w0] "+v"(w[0]), [w1] "+v"(w[1]), [w2] "+v"(w[2]), [w3] "+v"(w[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
        : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [ol] "v"(ol), [oh] "v"(oh));
  }
  static __device__ __forceinline__ void scale_bias4(uint32_t (&w)[4], uint32_t spair, int sh, uint32_t bpair, int bh) {
    const uint32_t s0 = sh ? spair >> 16 : spair & 0xffffu;          // (s, 0)
    const uint32_t s1 = sh ? spair & 0xffff0000u : spair << 16;      // (0, s)
    const float b32 = half_of(bpair, bh);
    const uint32_t ol = one_lo(), oh = one_hi();
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_dot2_f32_bf16 %[t0], %[w0], %[s0], 0\n\tv_dot2_f32_bf16 %[t1], %[w0], %[s1], 0\n\t"
        "v_dot2_f32_bf16 %[t2], %[w1], %[s0], 0\n\tv_dot2_f32_bf16 %[t3], %[w1], %[s1], 0\n\t"
        "v_dot2_f32_bf16 %[t4], %[w2], %[s0], 0\n\tv_dot2_f32_bf16 %[t5], %[w2], %[s1], 0\n\t"
        "v_dot2_f32_bf16 %[t6], %[w3], %[s0], 0\n\tv_dot2_f32_bf16 %[t7], %[w3], %[s1], 0\n\t"
        "v_cvt_pk_bf16_f32 %[w0], %[t0], %[t1]\n\tv_cvt_pk_bf16_f32 %[w1], %[t2], %[t3]\n\t"
        "v_cvt_pk_bf16_f32 %[w2], %[t4], %[t5]\n\tv_cvt_pk_bf16_f32 %[w3], %[t6], %[t7]\n\t"
        "v_dot2_f32_bf16 %[t0], %[w0], %[ol], %[b]\n\tv_dot2_f32_bf16 %[t1], %[w0], %[oh], %[b]\n\t"
        "v_dot2_f32_bf16 %[t2], %[w1], %[ol], %[b]\n\tv_dot2_f32_bf16 %[t3], %[w1], %[oh], %[b]\n\t"
        "v_dot2_f32_bf16 %[t4], %[w2], %[ol], %[b]\n\tv_dot2_f32_bf16 %[t5], %[w2], %[oh], %[b]\n\t"
        "v_dot2_f32_bf16 %[t6], %[w3], %[ol], %[b]\n\tv_dot2_f32_bf16 %[t7], %[w3], %[oh], %[b]\n\t"
        "v_cvt_pk_bf16_f32 %[w0], %[t0], %[t1]\n\tv_cvt_pk_bf16_f32 %[w1], %[t2], %[t3]\n\t"
        "v_cvt_pk_bf16_f32 %[w2], %[t4], %[t5]\n\tv_cvt_pk_bf16_f32 %[w3], %[t6], %[t7]"
        : [w0] "+v"(w[0]), [w1] "+v"(w[1]), [w2] "+v"(w[2]), [w3] "+v"(w[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
        : [s0] "v"(s0), [s1] "v"(s1), [b] "v"(b32), [ol] "v"(ol), [oh] "v"(oh));
  }
  static __device__ __forceinline__ void fma4(float* acc, const uint32_t (&w)[4], uint32_t xpair, int xh) {
    const uint32_t x0 = xh ? xpair >> 16 : xpair & 0xffffu;          // (x, 0)
    const uint32_t x1 = xh ? xpair & 0xffff0000u : xpair << 16;      // (0, x)
    asm("v_dot2_f32_bf16 %[a0], %[w0], %[x0], %[a0]\n\tv_dot2_f32_bf16 %[a1], %[w0], %[x1], %[a1]\n\t"
        "v_dot2_f32_bf16 %[a2], %[w1], %[x0], %[a2]\n\tv_dot2_f32_bf16 %[a3], %[w1], %[x1], %[a3]\n\t"
        "v_dot2_f32_bf16 %[a4], %[w2], %[x0], %[a4]\n\tv_dot2_f32_bf16 %[a5], %[w2], %[x1], %[a5]\n\t"
        "v_dot2_f32_bf16 %[a6], %[w3], %[x0], %[a6]\n\tv_dot2_f32_bf16 %[a7], %[w3], %[x1], %[a7]\n\t"
        "s_nop 3"
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]),
          [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
        : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [x0] "v"(x0), [x1] "v"(x1));
  }
  static __device__ __forceinline__ f32x4 mfma4(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4_t, a),
                                                  __builtin_bit_cast(s4_t, b), c, 0, 0, 0);
  }
};

// A pointer that came out of inline assembly or integer arithmetic is "flat" to the compiler
// (flat_load: slower, and counted in lgkmcnt as well): say that it is global.
template <typename T>
static __device__ __forceinline__ T* as_global(T* p) {
  typedef T __attribute__((address_space(1))) global_t;
  return (T*)(global_t*)(uintptr_t)p;
}
// broadcast one 16-bit element into both halves of a register
static __device__ __forceinline__ uint32_t splat16(uint16_t v) {
  return (uint32_t)v * 0x00010001u;
}

// ---- LDS access by absolute byte address (no symbol + offset add per access) ----
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
static __device__ __forceinline__ u32x4 lds_load16(uint32_t byte_addr) {
  return *(const lds_u32x4_t*)(uintptr_t)byte_addr;  // ds_read_b128
}
static __device__ __forceinline__ void lds_store16(uint32_t byte_addr, u32x4 v) {
  *(lds_u32x4_t*)(uintptr_t)byte_addr = v;  // ds_write_b128
}

// ---- wave64 reductions -------------------------------------------------------

// x + (x rotated right by N lanes inside each row of 16 lanes): one v_add_f32 with
// a DPP row_ror modifier.  ror 8,4,2,1 in sequence = all-reduce over the row.
template <int N>
static __device__ __forceinline__ float row_ror_add(float x) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false);
  return x + __int_as_float(moved);
}
static __device__ __forceinline__ float row16_allsum(float x) {
  x = row_ror_add<8>(x);
  x = row_ror_add<4>(x);
  x = row_ror_add<2>(x);
  return row_ror_add<1>(x);
}
// Sum over the 64 lanes, returned in every lane: 4 DPP adds inside the rows of 16, then the
// gfx950 lane swaps join rows 0+1 / 2+3 and the two halves (swap(v, v) leaves [lo | lo] and
// [hi | hi]).  6 dependent VALU ops; a __shfl_xor butterfly is 6 dependent ds_bpermute_b32
// round trips (~0.3 us - it was a tenth of a 4096^2 launch, tools/trace_k256m.py).
static __device__ __forceinline__ float wave_sum(float v) {
  v = row16_allsum(v);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Full reduce-scatter of NV (8, 16, 32 or 64) per-lane partials over the 64 lanes of a
// wave: every step halves the values a lane carries, so the whole reduction costs ~2*NV
// VALU ops (a shuffle tree costs 12*NV and measured 5.7 us per launch).  Lane-pair
// exchanges: v_permlane32_swap / v_permlane16_swap (gfx950) for lane bits 5 and 4
// (swap(a, b) leaves a = [a.lo | b.lo], b = [a.hi | b.hi]: a + b = sum(a) in the low
// half, sum(b) in the high half), DPP row_mirror / row_half_mirror / quad_perm for
// bits 3..0.  On return v[0] of lane l holds the wave total of original entry
// l >> kShift (replicated over the low kShift lane bits).
template <int CTRL>
static __device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

template <int NV>
struct WaveReduce {
  static_assert(NV == 8 || NV == 16 || NV == 32 || NV == 64, "NV must be 8..64, power of two");
  static constexpr int kLog = NV == 8 ? 3 : NV == 16 ? 4 : NV == 32 ? 5 : 6;
  static constexpr int kShift = 6 - kLog;

  // one DPP halving step on lane bit BIT (3..0): lanes with the bit clear keep the first
  // half of v[0..n), the others the second half; each adds its partner's copy.
  template <int BIT, int n>
  static __device__ __forceinline__ void halve_dpp(float (&v)[NV], int lane) {
    constexpr int ctrl = BIT == 3 ? 0x140 /*row_mirror*/ : BIT == 2 ? 0x141 /*row_half_mirror*/
                         : BIT == 1 ? 0x4E /*quad_perm [2,3,0,1]*/ : 0xB1 /*quad_perm [1,0,3,2]*/;
    const bool up = (lane >> BIT) & 1;
#pragma unroll
    for (int i = 0; i < n / 2; ++i) {
      const float keep = up ? v[i + n / 2] : v[i];
      const float send = up ? v[i] : v[i + n / 2];
      v[i] = keep + dpp_mov<ctrl>(send);
    }
  }
  template <int BIT>
  static __device__ __forceinline__ void butterfly_dpp(float& x) {
    constexpr int ctrl = BIT == 3 ? 0x140 : BIT == 2 ? 0x141 : BIT == 1 ? 0x4E : 0xB1;
    x = x + dpp_mov<ctrl>(x);
  }

  static __device__ __forceinline__ void run(float (&v)[NV], int lane) {
    // bit 5
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]),
                                                __float_as_uint(v[i + NV / 2]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // bit 4
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]),
                                                __float_as_uint(v[i + NV / 4]), false, false);
      v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    constexpr int n3 = NV / 4;  // values left before the bit-3 step (>= 2)
    halve_dpp<3, n3>(v, lane);
    if (n3 / 2 >= 2) halve_dpp<2, (n3 / 2 >= 2 ? n3 / 2 : 2)>(v, lane); else butterfly_dpp<2>(v[0]);
    if (n3 / 4 >= 2) halve_dpp<1, (n3 / 4 >= 2 ? n3 / 4 : 2)>(v, lane); else butterfly_dpp<1>(v[0]);
    if (n3 / 8 >= 2) halve_dpp<0, (n3 / 8 >= 2 ? n3 / 8 : 2)>(v, lane); else butterfly_dpp<0>(v[0]);
  }
};

// ---- packed index bit stream (vptq/utils/pack.py:26-67) ------------------------
// element g of a row = bits [g*T, (g+1)*T) of the little-endian word stream.
// Reads word wi and, only when the element straddles, word wi+1 (which then
// exists inside the row): never touches memory past the row, unlike the
// reference's iterator (csrc/util/cuda_utils.cuh:131).
static __device__ __forceinline__ uint32_t unpack_elem(const uint32_t* __restrict__ row, int g,
                                                       int T) {
  const uint32_t bit = (uint32_t)g * (uint32_t)T;
  const uint32_t wi = bit >> 5, sh = bit & 31u;
  uint64_t w = row[wi];
  if (sh + (uint32_t)T > 32u) w |= (uint64_t)row[wi + 1] << 32;
  const uint32_t mask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
  return (uint32_t)(w >> sh) & mask;
}

}  // namespace vptq
