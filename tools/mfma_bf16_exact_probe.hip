// Can the matrix pipe do the WIDENED bf16 arithmetic of the reference's roundings bit for bit?  (round 6)
// The reference CPU path computes w = bf16(bf16(bf16(c + r) * s) + b) with torch's bf16 ops: widen to fp32, operate, round (RNE).  On
// the VALU that is ~68 instructions per index (unpack / fp32 op / v_cvt_pk_bf16_f32 per stage) - 21 us per 8192^2 layer.  With
// v_mfma_f32_4x4x4_16b_bf16 and an identity first operand, lane j receives ITS OWN four second-operand values widened to fp32
// (D[i][j] = sum_k I[i][k] B[k][j] = B[i][j]); accumulating a second such product adds in fp32; a first operand s * I multiplies.
// Every intermediate is a sum of two bf16 values or a product of two: exactly representable in fp32 or rounded trivially, so the
// pipe's adder should give the IEEE result.  This probe checks that on random and adversarial operands.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_bf16_exact_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ static inline float up(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ static inline uint16_t rn(float f) { const f2_t v = {f, 0.f}; return (uint16_t)__builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)); }
// in: [n][4] c, r (per lane four values), s, b per BLOCK of 4 lanes; out: w_ref, w_mfma
__global__ void k(const uint16_t* c, const uint16_t* r, const uint16_t* s, const uint16_t* b, uint16_t* wref, uint16_t* wm, int n_waves) {
  const int l = threadIdx.x, i4 = l & 3;
  for (int it = blockIdx.x; it < n_waves; it += gridDim.x) {
    const size_t base = ((size_t)it * 64 + l) * 4;
    const size_t blk = (size_t)it * 16 + (l >> 2);
    const uint16_t sv = s[blk], bv = b[blk];
    s4 C, R, I = {0, 0, 0, 0}, SI = {0, 0, 0, 0}, BI = {0, 0, 0, 0}, ONES = {0x3f80, 0x3f80, 0x3f80, 0x3f80};
    I[i4] = 0x3f80; SI[i4] = (short)sv; BI[i4] = (short)bv;
    for (int q = 0; q < 4; ++q) { C[q] = (short)c[base + q]; R[q] = (short)r[base + q]; }
    // reference: widened VALU arithmetic
    uint16_t wr[4];
    for (int q = 0; q < 4; ++q) {
      const uint16_t w1 = rn(up(c[base + q]) + up(r[base + q]));
      const uint16_t w2 = rn(up(w1) * up(sv));
      wr[q] = rn(up(w2) + up(bv));
    }
    // matrix pipe
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(I, C, d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(I, R, d, 0, 0, 0);
    s4 W1; for (int q = 0; q < 4; ++q) W1[q] = (short)rn(d[q]);
    f4 z = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(SI, W1, z, 0, 0, 0);
    s4 W2; for (int q = 0; q < 4; ++q) W2[q] = (short)rn(d[q]);
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(I, W2, z, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(BI, ONES, d, 0, 0, 0);
    for (int q = 0; q < 4; ++q) { wref[base + q] = wr[q]; wm[base + q] = rn(d[q]); }
  }
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
int main() {
  const int W = 4096, n = W * 64 * 4, nb = W * 16;
  uint16_t *hc = new uint16_t[n], *hr = new uint16_t[n], *hs = new uint16_t[nb], *hb = new uint16_t[nb], *o1 = new uint16_t[n], *o2 = new uint16_t[n];
  uint16_t *dc, *dr, *ds, *db, *d1, *d2;
  hipMalloc(&dc, n * 2); hipMalloc(&dr, n * 2); hipMalloc(&ds, nb * 2); hipMalloc(&db, nb * 2); hipMalloc(&d1, n * 2); hipMalloc(&d2, n * 2);
  const char* names[5] = {"checkpoint-like (c ~ 1, r ~ 0.25, s ~ 0.02, b ~ 0.002)", "reference test (all ~ 0.02 + 0.5 N)", "wide exponents (2^-20 .. 2^20)",
                          "ties (integers and halves)", "raw random bit patterns (finite)"};
  for (int mode = 0; mode < 5; ++mode) {
    srand(1234 + mode);
    auto gen = [&](float scale, float mean) -> uint16_t {
      if (mode == 2) return f2bf(rnd() * ldexpf(1.f, rand() % 41 - 20));
      if (mode == 3) return f2bf((float)(rand() % 513 - 256) * 0.5f);
      if (mode == 4) { uint16_t v; do { v = (uint16_t)rand(); } while ((v & 0x7f80) == 0x7f80); return v; }
      return f2bf(mean + scale * rnd() * 1.7f);
    };
    for (int i = 0; i < n; ++i) { hc[i] = gen(mode == 1 ? 0.5f : 1.f, mode == 1 ? 0.02f : 0.f); hr[i] = gen(mode == 1 ? 0.5f : 0.25f, mode == 1 ? 0.02f : 0.f); }
    for (int i = 0; i < nb; ++i) { hs[i] = gen(mode == 1 ? 0.5f : 0.006f, 0.02f); hb[i] = gen(mode == 1 ? 0.5f : 0.002f, mode == 1 ? 0.02f : 0.f); }
    hipMemcpy(dc, hc, n * 2, hipMemcpyHostToDevice); hipMemcpy(dr, hr, n * 2, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs, nb * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb, nb * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, dc, dr, ds, db, d1, d2, W);
    hipMemcpy(o1, d1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(o2, d2, n * 2, hipMemcpyDeviceToHost);
    long bad = 0, bad_norm = 0; int shown = 0;
    for (int i = 0; i < n; ++i) if (o1[i] != o2[i]) {
      ++bad;
      const bool den = ((o1[i] & 0x7f80) == 0) || ((o2[i] & 0x7f80) == 0);   // (a denormal / zero result: flush-to-zero differences)
      if (!den) { ++bad_norm; if (shown++ < 4) printf("    c %04x r %04x s %04x b %04x: ref %04x mfma %04x\n", hc[i], hr[i], hs[(i / 4) / 4], hb[(i / 4) / 4], o1[i], o2[i]); }
    }
    printf("%-62s %d values: %ld differ, %ld of them with normal results\n", names[mode], n, bad, bad_norm);
  }
  return 0;
}
