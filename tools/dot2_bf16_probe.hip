// Can v_dot2_f32_bf16 stand in for "widen, operate in fp32" in the reference's bf16 roundings?  (round 6)
// The per-lane kernels (gemv_sliced, gemv_k256's VALU form) cannot use the matrix-pipe form of gemv_k256m (every lane has its own column,
// hence its own scale).  Their widened arithmetic costs unpack + op + v_cvt_pk_bf16_f32 + unpack per stage.  v_dot2_f32_bf16 computes
// D = a.lo * b.lo + a.hi * b.hi + c in fp32 from PACKED bf16 operands: with b = (s, 0) it is lo(a) * s, with b = (1, 0) and c = b32 it
// is lo(a) + b32, with b = (x, 0) and c = acc it is the multiply-add of the product - no unpacking.  Its internal precision is not
// documented; this probe compares bit patterns against the widened arithmetic:
//   mul : bf16(lo(u) * s)                    vs  bf16(dot2(u, (s, 0), 0))            (and the hi half with (0, s))
//   add : bf16(lo(t) + b)                    vs  bf16(dot2(t, (1, 0), widen(b)))
//   fma : fmaf(lo(w), x, acc) (fp32 bits)    vs  dot2(w, (x, 0), acc)
//   hipcc --offload-arch=gfx950 -O2 tools/dot2_bf16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ static inline float up(uint32_t b) { return __uint_as_float(b << 16); }
__device__ static inline uint32_t rn(float f) { const f2_t v = {f, 0.f}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)) & 0xffffu; }
__device__ static inline float dot2(uint32_t a, uint32_t b, float c) {
  // (the builtin, not inline assembly: the result of a dot instruction needs wait states before a VALU read, which the compiler
  // only inserts for instructions it sees - the first version of this probe read garbage)
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), c, false);
}
// a: packed pairs (value | junk << 16); s, b: bf16; acc: fp32.  out[i] = {mul_ref, mul_dot, add_ref, add_dot, fma_ref, fma_dot, mulhi_ref, mulhi_dot}
__global__ void k(const uint32_t* a, const uint16_t* s, const uint16_t* b, const float* acc, uint32_t* out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t av = a[i], sv = s[i], bv = b[i];
    const float ac = acc[i];
    uint32_t* o = out + (size_t)i * 8;
    o[0] = rn(up(av & 0xffffu) * up(sv));
    o[1] = rn(dot2(av, sv, 0.f));                      // (s, 0)
    o[2] = rn(up(av & 0xffffu) + up(bv));
    uint32_t one = 0x3f80u;                             // (1, 0) - hidden from the compiler: it folds the constant into the inline
    asm volatile("" : "+v"(one));                       // operand `1.0`, which the instruction reads as (0, 1): the HIGH half got added
    o[3] = rn(dot2(av, one, up(bv)));                   // c = widen(b)
    o[4] = __float_as_uint(__builtin_fmaf(up(av & 0xffffu), up(sv), ac));
    o[5] = __float_as_uint(dot2(av, sv, ac));
    o[6] = rn(up(av >> 16) * up(sv));
    o[7] = rn(dot2(av, sv << 16, 0.f));                // (0, s): the high half
  }
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
static float gauss() { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; return s - 6.f; }
int main() {
  const int n = 1 << 22;
  uint32_t* ha = new uint32_t[n]; uint16_t *hs = new uint16_t[n], *hb = new uint16_t[n]; float* hacc = new float[n]; uint32_t* ho = new uint32_t[(size_t)n * 8];
  uint32_t *da, *dout; uint16_t *ds, *db; float* dacc;
  hipMalloc(&da, n * 4); hipMalloc(&ds, n * 2); hipMalloc(&db, n * 2); hipMalloc(&dacc, n * 4); hipMalloc(&dout, (size_t)n * 32);
  const char* names[6] = {"checkpoint-like (w ~ 1, s ~ 0.02, b ~ 0.002)", "reference test (all ~ 0.02 + 0.5 N)", "wide exponents (2^-30 .. 2^30)",
                          "ties (small integers and halves)", "denormal operands / results", "raw random bit patterns (finite)"};
  for (int mode = 0; mode < 6; ++mode) {
    srand(17 + mode);
    for (int i = 0; i < n; ++i) {
      float w, s, b, junk = gauss();
      if (mode == 0) { w = gauss(); s = 0.02f * expf(0.3f * gauss()); b = 0.002f * gauss(); }
      else if (mode == 1) { w = 0.02f + 0.5f * gauss(); s = 0.02f + 0.5f * gauss(); b = 0.02f + 0.5f * gauss(); }
      else if (mode == 2) { w = rnd() * exp2f(60.f * rnd() / 2); s = rnd() * exp2f(60.f * rnd() / 2); b = rnd() * exp2f(60.f * rnd() / 2); }
      else if (mode == 3) { w = (float)(rand() % 513 - 256) * 0.5f; s = (float)(rand() % 33 - 16) * 0.25f; b = (float)(rand() % 513 - 256) * 0.25f; }
      else if (mode == 4) { w = rnd() * exp2f(-100.f - 30.f * (float)rand() / RAND_MAX); s = rnd() * exp2f(-20.f * (float)rand() / RAND_MAX); b = rnd() * exp2f(-120.f - 10.f * (float)rand() / RAND_MAX); }
      else { w = s = b = 0.f; }
      uint16_t wb = f2bf(w), jb = f2bf(junk);
      hs[i] = f2bf(s); hb[i] = f2bf(b);
      if (mode == 5) {
        auto fin = [&]() { uint16_t v; do { v = (uint16_t)(rand() & 0xffff); } while ((v & 0x7f80) == 0x7f80); return v; };
        wb = fin(); jb = fin(); hs[i] = fin(); hb[i] = fin();
      }
      ha[i] = (uint32_t)wb | ((uint32_t)jb << 16);
      hacc[i] = mode == 4 ? 0.f : gauss() * (mode == 2 ? exp2f(40.f * rnd()) : 1.f);
    }
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs, n * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, n * 2, hipMemcpyHostToDevice); hipMemcpy(dacc, hacc, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, da, ds, db, dacc, dout, n);
    hipMemcpy(ho, dout, (size_t)n * 32, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0};
    int shown = 0;
    for (int i = 0; i < n; ++i)
      for (int t = 0; t < 4; ++t) {
        const uint32_t r = ho[(size_t)i * 8 + 2 * t], d = ho[(size_t)i * 8 + 2 * t + 1];
        const bool nan_both = t == 2 ? ((r & 0x7f800000u) == 0x7f800000u && (d & 0x7f800000u) == 0x7f800000u) : ((r & 0x7f80u) == 0x7f80u && (d & 0x7f80u) == 0x7f80u && (r & 0x7f) && (d & 0x7f));
        if (r != d && !nan_both) {
          ++bad[t];
          if (shown < 6) { ++shown; printf("   e.g. %s a=%08x s=%04x b=%04x acc=%08x ref=%08x dot2=%08x\n", t == 0 ? "mul" : t == 1 ? "add" : t == 2 ? "fma" : "mul-hi", ha[i], hs[i], hb[i], *(uint32_t*)&hacc[i], r, d); }
        }
      }
    printf("%-55s %d values: mul %ld, add %ld, fma %ld, mul-hi %ld differ\n", names[mode], n, bad[0], bad[1], bad[2], bad[3]);
  }
  return 0;
}
