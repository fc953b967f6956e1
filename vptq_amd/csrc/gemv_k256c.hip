// Fused dequant + GEMV for the canonical "2-bit" VPTQ format (v = 8, 256 + 256 centroids):
// ONE persistent launch that walks a CHAIN of layers.  Same contract as gemv_k256m.hip /
// gemv_k256.hip, same reference: n calls of `quant_gemv` (vptq/ops/quant_gemm.py:214-228), each
// of them csrc/kernels/quant_gemv.cuh:11-186 + the tmp.sum of csrc/quant_gemv.cu:203-235.
//
// Why.  One launch per layer (gemv_k256m.hip) pays, per 8192^2 layer, ~3 us of launch boundary,
// prologue (codebook image, activations) and epilogue around ~4 us of accumulation - and a
// workgroup that streams 64 KiB cannot keep HBM busy while it does the other things.  Here a
// workgroup's work is ONE FLAT STREAM OF SWEEPS over (layer, row group, sweep):
//  * a wave owns 128 consecutive columns of a sweep (16 waves = 2048 columns) for kCRows = 8
//    vector-rows: per sweep and lane two 16-byte index loads (8 columns of vector-rows j and
//    4 + j), and 4 bytes each of x, scale and bias (columns 2 lane, 2 lane + 1 of its block).
//    Everything is requested kCDepth sweeps ahead, across row-group and layer boundaries, into a
//    register queue with static slots (the loop is unrolled by the depth), so that every wait is
//    the counted, in-order `s_waitcnt vmcnt(n)`.
//  * the wave stages f16(scale * x) of its own 128 columns itself (wave-private 256-byte LDS slot:
//    a wave's LDS operations execute in order, no barrier), and sums bias * x on the way.
//  * arithmetic = the folded form of gemv_k256m.hip: lane = (column chunk of 8, vector-row j);
//    v_mfma_f32_4x4x4 with X = x' * I as 64 x 4 FMAs, main and residual entry gathered with
//    ds_read_b128 from the conflict-free lane-split image (row e = 8 replicas of main entry e +
//    8 of residual entry e).  Two row subgroups per sweep share the x operand (2 of the 6
//    v_perm_b32 per index pair) and halve the per-sweep bookkeeping per byte.
//    [A transposing gather (ds_read_b64_tr_b16: result lane (row, output) gets 4 columns of one
//    output = the B operand of a real contraction over columns, v_mfma_f32_16x16x32 with
//    identical A rows, no x-operand perms, no cross-lane reduction) was built first and measured:
//    46.8 against 34.6 SIMD cycles per index-wave in isolation (tools/ubench_loop_t.hip; both sit
//    on the LDS - 2 KiB of gathered entries per index-wave - and the matrix pipe, 32 cycles each;
//    the 4x4x4 form overlaps them better), no difference inside the kernel; DESIGN.md section 4.9.]
//  * the NEXT layer's codebook image is filled into the second image buffer by LDS-DMA
//    (global_load_lds_dwordx4: no registers, issued as inline assembly because a DMA the compiler
//    can see makes it wait for every load in flight) as soon as every wave has left the layer
//    that used the buffer; kCDepth sweeps later only younger loads are in flight (vmcnt retires in
//    order), so the fill has landed.  Hand-over between the waves (image free / image landed,
//    partial sums) goes through LDS counters with fences restricted to the LOCAL address space -
//    an ordinary workgroup-scope release also waits for every global load in flight, i.e. for the
//    whole queue.
//  * row groups are dealt to the workgroups in blocks of consecutive ones, at least kCMinSteps
//    sweeps per visit of a layer; a layer that needs fewer blocks than there are workgroups
//    leaves the rest to the next layers (independent chains: q / k / v, gate / up, a ring).
//  * the SIMDs serve their waves oldest first, so the waves of a workgroup drift apart until the
//    fast ones wait at every hand-over: issue priority by arrival order at the last row group.
//  * DEP = dependent chain (x of layer i + 1 is y of layer i): outputs are stored write-through at
//    device scope, one arrival flag per workgroup and layer, x is read with device-coherent loads;
//    index words, scales and the image of the next layer are still requested ahead.  Measured:
//    11.5 us per 8192^2 layer - the hand-over through memory costs ~6 us per layer, more than the
//    kernel boundary it replaces (7.3 us with one launch per layer): exists for completeness.
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "k256.h"

namespace vptq {

// profiling build (tools/chain_prof.py): every wave adds up the shader-clock cycles it spends
// waiting for its index words, consuming a sweep, requesting the next one and in the rare paths,
// and stores them at P.sync[(workgroup * waves + wave) * kCProfWords ...] (non-dependent launches); 2: + a timeline
#ifndef VPTQ_K256C_PROF
#define VPTQ_K256C_PROF 0
#endif
// waves per workgroup (= per CU): 16 (4 per SIMD, 128 registers each) or 8 (2 per SIMD, 256 registers: room for 4 row
// subgroups per sweep and a deeper gather lookahead; a step then covers twice the indices per wave)
#ifndef VPTQ_K256C_WAVES
#define VPTQ_K256C_WAVES 16
#endif
constexpr int kCWaves = VPTQ_K256C_WAVES;
constexpr int kCThreads = 64 * kCWaves;
static_assert(kCWaves == 16 || kCWaves == 8, "waves per workgroup");
constexpr int kCBlockCols = 128;                     // columns of one wave per sweep
// The waves of a workgroup as kCSplit row parts x kCColWaves column blocks: a sweep covers kCColWaves x 128 columns
// of kCSplit x 8 vector-rows.  2 parts: a row group of an 8192-column layer is 8 sweeps instead of 4, so the sums
// at the end of a row group (cross-lane reduction, deposit, hand-over: ~1500 clocks per wave) come half as often;
// the sum over the waves is over 8 instead of 16 partial sums.
#ifndef VPTQ_K256C_SPLIT
#define VPTQ_K256C_SPLIT 1
#endif
constexpr int kCSplit = VPTQ_K256C_SPLIT;
static_assert(kCSplit == 1 || kCSplit == 2 || kCSplit == 4, "row parts");
constexpr int kCColWaves = kCWaves / kCSplit;
static_assert(kCColWaves >= 4, "the sum over the waves is a tree of groups of 4");
constexpr int kCSweepCols = kCColWaves * kCBlockCols;   // 2048
#ifndef VPTQ_K256C_SUB
#define VPTQ_K256C_SUB 2
#endif
constexpr int kCSub = VPTQ_K256C_SUB;                // row subgroups (4 vector-rows each) per row group
constexpr int kCRowsW = 4 * kCSub;                   // vector-rows of one wave
constexpr int kCOutW = 8 * kCRowsW;                  // outputs of one wave
constexpr int kCRows = kCRowsW * kCSplit;            // vector-rows per row group
constexpr int kCOut = 8 * kCRows;                    // outputs per row group
static_assert(kCSub == 1 || kCSub == 2 || kCSub == 4, "row subgroups per sweep");
static_assert(kCSub != 1 || (kCWaves == 16 && kCSplit == 1), "1 subgroup: the 16-wave sum");
// gathers requested ahead of the arithmetic, in units (one index of one subgroup)
#ifndef VPTQ_K256C_AHEAD
#define VPTQ_K256C_AHEAD 2
#endif
// sweeps in flight per wave: bandwidth x latency is ~48 KiB per CU
#ifndef VPTQ_K256C_DEPTH
#define VPTQ_K256C_DEPTH 3
#endif
constexpr int kCDepth = VPTQ_K256C_DEPTH;
static_assert(kCDepth >= 2 && kCDepth <= 6, "queue depth");
constexpr int kCSlots = 4;                           // cross-wave partial-sum slots
constexpr uint32_t kCImgBytes = 65536;               // 256 rows x 16 units x 16 B
constexpr uint32_t kCXsOff = 2 * kCImgBytes;         // wave-private activation slots
// arithmetic of a launch: folded (sum (c + r) f16(s x) + sum b x), the reference's roundings per weight, or SELECTIVE:
// the folded form, with the reference's roundings wherever an activation column dominates (kCModeSel below)
constexpr int kCModeFolded = 0, kCModeExact = 1, kCModeSel = 2;
// folded form (and selective): 256 bytes per wave (f16(s x) of its 128 columns); reference roundings: x, scale, bias side by side
constexpr uint32_t c_xs_wave(int mode) { return mode == kCModeExact ? 768u : 256u; }
constexpr uint32_t c_red_off(int mode) { return kCXsOff + kCWaves * c_xs_wave(mode); }
constexpr uint32_t c_redb_off(int mode) { return c_red_off(mode) + kCSlots * kCWaves * kCOutW * 4; }
constexpr uint32_t c_cnt_off(int mode) { return c_redb_off(mode) + kCSlots * kCWaves * 4; }
// profiling build 2: consume start / end stamps of kCTlSteps steps per wave (a timeline of who computes when)
[[maybe_unused]] constexpr int kCTlFirst = 16, kCTlSteps = 32;
constexpr uint32_t c_tl_off(int mode) { return c_cnt_off(mode) + 64; }
constexpr uint32_t c_lds_bytes(int mode) {
#if VPTQ_K256C_PROF >= 2
  return c_tl_off(mode) + kCWaves * kCTlSteps * 8;
#else
  return c_cnt_off(mode) + 64;
#endif
}
[[maybe_unused]] constexpr int kCProfWords = 64;   // 8-byte words of profile output per wave
static_assert(c_lds_bytes(kCModeFolded) <= 163840 && (VPTQ_K256C_PROF >= 2 || c_lds_bytes(kCModeExact) <= 163840), "LDS");
constexpr int kCFlagStride = 256;   // DEP: arrival flags per layer (one per workgroup)

// timing-only ablations (results wrong): bit 0 no MFMAs, bit 1 no gathers, bit 2 no x / scale /
// bias loads, bit 3 no waits for the image hand-over between the waves, bit 4 no index loads
#ifndef VPTQ_K256C_ABLATE
#define VPTQ_K256C_ABLATE 0
#endif
// bring-up builds: give up a wait after that many polls, so that a protocol error shows as wrong
// results instead of a hung GPU
#ifndef VPTQ_K256C_SPIN_LIMIT
#define VPTQ_K256C_SPIN_LIMIT 0
#endif
#ifndef VPTQ_K256C_BALANCE
#define VPTQ_K256C_BALANCE 1
#endif
// 1: everything outside the consume phase runs at issue priority 3: those instructions compete with the
// other waves' MFMAs for the issue port and hold nothing another wave needs (measured: 4.91 -> 4.67 us)
#ifndef VPTQ_K256C_LDS_FENCE
#define VPTQ_K256C_LDS_FENCE 1
#endif
#ifndef VPTQ_K256C_PREADD
#define VPTQ_K256C_PREADD 0
#endif
#ifndef VPTQ_K256C_STAGGER
#define VPTQ_K256C_STAGGER 0
#endif
#ifndef VPTQ_K256C_PRIO
#define VPTQ_K256C_PRIO 1
#endif

// One layer of the launch: 88 bytes of kernel arguments (the K256Layer of the other kernels carries
// fields this one has no use for, and 32 of those + the search table would not fit 4 KiB).
struct CLayerArgs {
  const uint32_t* idx;
  const uint32_t* cent;
  const uint32_t* rcent;
  const uint16_t* x;
  uint16_t* y;
  const uint16_t* scale;
  const uint16_t* wbias;
  const uint16_t* bias;
  int N, G, O, row_words;
  int wgs;   // workgroup that owns block 0 of the layer
  int rpw;   // row groups per block
};
constexpr int kCHeadPad = 4;   // the layer search reads 4 headers at a time
struct K256CParams {
  int n_layers;
  int tokens;        // token count | kOutF32Bit
  uint32_t* sync;    // DEP: kCFlagStride arrival flags per layer (zeroed before the launch)
  CLayerArgs layer[kMaxGroup];
  // what the search for a workgroup's next layer needs, 8 bytes per layer:
  // {first workgroup | row groups per block << 16, row groups | sweeps per row group << 24}
  uint32_t head[kMaxGroup + kCHeadPad][2];
};
static_assert(sizeof(CLayerArgs) == 88 && offsetof(K256CParams, layer) == 16 &&
              offsetof(K256CParams, head) == 16 + kMaxGroup * 88 && sizeof(K256CParams) <= 4096, "kernarg layout");

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef const char __attribute__((address_space(4)))* kernarg_ptr_t;

// Layer L of the kernel arguments in one batch of scalar loads.  Everything a wave looks up on its
// way through the stream comes out of the kernel-argument segment through the scalar cache: the
// LDS is the busiest unit of this kernel (a read issued behind 16 waves' gathers comes back late),
// and indexing the by-value argument struct with a run-time index makes the compiler copy it to
// scratch memory.
static __device__ __forceinline__ CLayerArgs c_load_layer(int L) {
  typedef int i16_t __attribute__((ext_vector_type(16)));
  typedef int i4_t __attribute__((ext_vector_type(4)));
  typedef int i2_t __attribute__((ext_vector_type(2)));
  L = __builtin_amdgcn_readfirstlane(L);   // (wave-uniform by construction; the compiler cannot always prove it)
  const kernarg_ptr_t lp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr() + 16 + (size_t)L * sizeof(CLayerArgs);
  i16_t a; i4_t b; i2_t c;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "s_load_dwordx16 %0, %3, 0x0\n\t"
      "s_load_dwordx4 %1, %3, 0x40\n\t"
      "s_load_dwordx2 %2, %3, 0x50\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(a), "=&s"(b), "=&s"(c)
      : "s"(lp)
      : "memory");
#else
  a = i16_t{}; b = i4_t{}; c = i2_t{}; (void)lp;
#endif
  CLayerArgs Ly;
  __builtin_memcpy((char*)&Ly, &a, 64);
  __builtin_memcpy((char*)&Ly + 64, &b, 16);
  __builtin_memcpy((char*)&Ly + 80, &c, 8);
  Ly.idx = as_global(Ly.idx); Ly.cent = as_global(Ly.cent); Ly.rcent = as_global(Ly.rcent);
  Ly.x = as_global(Ly.x); Ly.y = as_global(Ly.y); Ly.scale = as_global(Ly.scale);
  Ly.wbias = as_global(Ly.wbias); Ly.bias = as_global(Ly.bias);
  return Ly;
}
// the two codebook pointers of layer L
struct CFillL { const uint32_t* cent; const uint32_t* rcent; };
static __device__ __forceinline__ CFillL c_load_fill(int L) {
  typedef int i4_t __attribute__((ext_vector_type(4)));
  L = __builtin_amdgcn_readfirstlane(L);
  const kernarg_ptr_t lp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr() + 16 + (size_t)L * sizeof(CLayerArgs);
  i4_t a;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_load_dwordx4 %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(lp) : "memory");
#else
  a = i4_t{}; (void)lp;
#endif
  CFillL F;
  __builtin_memcpy((char*)&F, &a, 16);
  F.cent = as_global(F.cent); F.rcent = as_global(F.rcent);
  return F;
}
// search headers of layers L .. L + 3 (the table is padded with kCHeadPad empty layers)
typedef int c_head4_t __attribute__((ext_vector_type(8)));
static __device__ __forceinline__ c_head4_t c_load_heads(int L) {
  L = __builtin_amdgcn_readfirstlane(L);
  const kernarg_ptr_t hp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(K256CParams, head) + (size_t)L * 8;
  c_head4_t h;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(hp) : "memory");
#else
  h = c_head4_t{}; (void)hp;
#endif
  return h;
}

// position in the workgroup's flat stream; everything is wave-uniform
struct CCursor {
  int L;    // layer; n_layers = past the end
  int rg;   // row group (kCRows vector-rows)
  int re;   // end of this workgroup's block of row groups in layer L
  int ns;   // sweeps per row group of layer L
  int ng;   // row groups of layer L
};
// the fields of a layer each side needs (the rest of a K256Layer dies right after the load)
struct CIssueL { const uint32_t* idx; const uint16_t* x; const uint16_t* scale; const uint16_t* wbias; int N, G, row_words; };
struct CConsL { uint16_t* y; const uint16_t* bias; int N, G, O; };
static __device__ __forceinline__ CIssueL c_issue_of(const CLayerArgs& L) {
  return CIssueL{L.idx, L.x, L.scale, L.wbias, L.N, L.G, L.row_words};
}
static __device__ __forceinline__ CConsL c_cons_of(const CLayerArgs& L) { return CConsL{L.y, L.bias, L.N, L.G, L.O}; }

// f(slot 0), f(slot 1), ... f(slot kCDepth - 1) with the slot as a compile-time constant
template <typename F>
static __device__ __forceinline__ void c_for_slots(F&& f) {
  f(std::integral_constant<int, 0>{});
  f(std::integral_constant<int, 1>{});
  if constexpr (kCDepth > 2) f(std::integral_constant<int, (kCDepth > 2 ? 2 : 0)>{});
  if constexpr (kCDepth > 3) f(std::integral_constant<int, (kCDepth > 3 ? 3 : 0)>{});
  if constexpr (kCDepth > 4) f(std::integral_constant<int, (kCDepth > 4 ? 4 : 0)>{});
  if constexpr (kCDepth > 5) f(std::integral_constant<int, (kCDepth > 5 ? 5 : 0)>{});
}

// EXACT: every weight rebuilt with the reference CPU path's three roundings r16(r16(r16(c + r) s) + b)
// (vptq/ops/quant_gemm.py:143-158; bit-identical to vptq_dequant), fp32 accumulation of x w by the same MFMAs
// - VPTQ_GEMV_EXACT inside the chain launch (fp16, independent layers).
//
// SEL (MODE = kCModeSel, VPTQ_GEMV_SELECTIVE; round 6): the folded form, EXCEPT where an activation column dominates the layer's -
// |f16(s x)| >= thr, thr = kappa x rms of f16(s x) over the layer's columns.  The folded form's distance to the reference is a sum
// of per-column rounding errors; with dense activations they average out over thousands of columns, a handful of dominant ones
// do not (profiles/r05/gate_count_*_folded_opt_in.txt).  Granularity = a wave's sweep block: 128 consecutive columns.  A block
// that holds a hot column contributes NOTHING in this kernel (its staged activations are zeroed: one ballot + two selects per
// sweep); k256c_hot_kernel, launched in front of this one, has rebuilt those blocks' weights with the reference's roundings and
// left their products in `corr` (float32 per output), which finalize() adds before the one rounding.  [Built first: the decision
// inside this kernel - a scalar branch per column, then per sweep in front of the EXACT and the folded loop body.  Either way the
// register allocator (128 registers, 4 waves per SIMD) spills - reloads are scratch loads, i.e. s_waitcnt vmcnt(0) in a loop
// whose waits must stay counted: 5.3 - 7.9 us per layer against 4.4 folded.]
// P.sync = per layer 4 words {thr, hot blocks, byte offset of the layer's corr from P.sync, 0}, written by k256c_hot_kernel.
template <typename DT, bool DEP, int MODE = kCModeFolded>
__global__ __launch_bounds__(kCThreads) void gemv_k256c_kernel(const K256CParams P) {
  constexpr bool EXACT = MODE == kCModeExact, SEL = MODE == kCModeSel;
  static_assert(MODE != kCModeExact || kCSub >= 2, "reference roundings in this loop: pairs of row subgroups");
  static_assert(MODE == kCModeFolded || !DEP, "reference / selective roundings: independent layers");
  static_assert(!SEL || VPTQ_K256C_PROF == 0, "selective: P.sync carries the thresholds");
  constexpr uint32_t kCXsWave = c_xs_wave(MODE), kCRedOff = c_red_off(MODE), kCRedBOff = c_redb_off(MODE),
                     kCCntOff = c_cnt_off(MODE);
#if VPTQ_K256C_PROF >= 2
  constexpr uint32_t kCTlOff = c_tl_off(MODE);
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
    if ((uint32_t)(uintptr_t)(lds_u8_t*)smem != 0u) __builtin_trap();  // absolute LDS addressing
  }
#if VPTQ_K256C_PROF
  unsigned long long pf_entry, pf_pro[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf_entry) :: "memory");
#define PF_PRO(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf_pro[i]) :: "memory")
#else
#define PF_PRO(i) ((void)0)
#endif
  constexpr int D = kCDepth;
  const int n_layers = P.n_layers;
  const bool out_f32 = (P.tokens & kOutF32Bit) != 0;
  // (a local copy: a lambda that refers to the by-value kernel argument P takes its address, and the
  // compiler then copies the whole 3.8 KB struct to scratch memory)
  uint32_t* const sync = as_global(P.sync);
  const int W = (int)gridDim.x, bid = (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- lane = (column chunk blk = lane >> 2, vector-row j = lane & 3).  The two gathers of an index
  // are split across the lanes: in gather A the lanes with bit 3 clear fetch the main entry (unit
  // lane & 7 of image row e), the others the residual entry (unit 8 + (lane & 7)); gather B is the
  // complement: every 16-lane group of a ds_read_b128 touches 16 different units (gemv_k256m.hip).
  // address = {0, base.byte2 = image buffer, index byte, base.byte0}; dword q of the index words holds
  // columns 2q (bytes 0 = main, 1 = residual index) and 2q + 1 (bytes 2, 3)
  const uint32_t jrow = (uint32_t)lane & 3u;
  const uint32_t hi8 = ((uint32_t)lane >> 3) & 1u;
  uint32_t baseA = ((hi8 << 3) | ((uint32_t)lane & 7u)) << 4;
  uint32_t baseB = (((hi8 ^ 1u) << 3) | ((uint32_t)lane & 7u)) << 4;
  const uint32_t selGA[2] = {0x0c020400u | (hi8 << 8), 0x0c020600u | (hi8 << 8)};
  const uint32_t selGB[2] = {0x0c020400u | ((hi8 ^ 1u) << 8), 0x0c020600u | ((hi8 ^ 1u) << 8)};
  // x operand of the 4x4x4 MFMA: x' * e_j as two packed pairs, cut out of a packed x' register
  const uint32_t selA[2] = {jrow == 0 ? 0x0c0c0504u : jrow == 1 ? 0x05040c0cu : 0x0c0c0c0cu,
                            jrow == 0 ? 0x0c0c0706u : jrow == 1 ? 0x07060c0cu : 0x0c0c0c0cu};
  const uint32_t selB[2] = {jrow == 2 ? 0x0c0c0504u : jrow == 3 ? 0x05040c0cu : 0x0c0c0c0cu,
                            jrow == 2 ? 0x0c0c0706u : jrow == 3 ? 0x07060c0cu : 0x0c0c0c0cu};

  // ---- LDS map: [0, 64 Ki) image buffer 0 | [64 Ki, 128 Ki) image buffer 1 | per wave 256 B of
  // staged activations | partial-sum slots | counters | layer table
  const uint32_t xs_base = kCXsOff + (uint32_t)wave * kCXsWave;
  const uint32_t st_addr = xs_base + (uint32_t)lane * 4u;           // this lane stages columns 2 lane, 2 lane + 1
  const uint32_t xq_addr = xs_base + ((uint32_t)lane >> 2) * 16u;   // ... and reads the 16 bytes of its chunk
  float* const red = (float*)(smem + kCRedOff);      // [slot][wave][kCOut]
  float* const red_b = (float*)(smem + kCRedBOff);   // [slot][wave]
  uint32_t* const slot_cnt = (uint32_t*)(smem + kCCntOff);  // [kCSlots] waves arrived
  uint32_t* const slot_done = slot_cnt + kCSlots;           // [kCSlots] row groups finished
  uint32_t* const free_cnt = slot_cnt + 2 * kCSlots;        // [2] waves that left a layer of image buffer b
  uint32_t* const ready_cnt = free_cnt + 2;                 // [2] waves whose part of a fill has landed
  uint32_t* const dep_seen = free_cnt + 4;                  // DEP: last layer whose producers wave 0 has seen arrive

  // ---- the flat stream ----
  // first layer >= c.L in which this workgroup owns row groups: a block of `rpw` consecutive ones,
  // block number (bid - first workgroup of the layer) mod W (CLayerArgs::wgs = the running total of
  // blocks mod W: the layers continue each other's round robin, so that layers that need fewer than
  // W blocks run side by side on different workgroups).  With 64 workgroups per layer a workgroup's
  // next layer is the fourth candidate: four 8-byte headers per scalar load.
  auto enter_layer = [&](CCursor& c) __attribute__((always_inline)) {
    while (c.L < n_layers) {
      const c_head4_t h = c_load_heads(c.L);
      bool found = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (found) continue;
        const uint32_t h0 = (uint32_t)h[2 * k], h1 = (uint32_t)h[2 * k + 1];
        const int wgs = (int)(h0 & 0xffffu), rpw = (int)(h0 >> 16);
        const int ng = (int)(h1 & 0xffffffu), ns = (int)(h1 >> 24);   // (padding: ng = 0)
        int r0 = bid - wgs;
        if (r0 < 0) r0 += W;
        r0 *= rpw;
        if (r0 < ng) {
          found = true;
          c.L += k; c.ng = ng; c.ns = ns; c.rg = r0; c.re = r0 + rpw < ng ? r0 + rpw : ng;
        }
      }
      if (found) break;
      c.L += 4;
    }
    if (c.L > n_layers) c.L = n_layers;
  };

#if VPTQ_K256C_PROF
  unsigned long long pf_wait = 0, pf_cons = 0, pf_issue = 0, pf_cold = 0, pf_steps = 0, pf_t0 = 0, pf_last = 0;
  unsigned long long pf_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now = [&](uint32_t dep) -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
  };
#if VPTQ_K256C_PROF >= 3   // light: only the consume start / end stamps of the timeline
#define PF_START() ((void)0)
#define PF_MARK(i) ((void)0)
#else
#define PF_START() (pf_last = now(0))
#define PF_MARK(i) do { const unsigned long long t_ = now(0); pf_x[i] += t_ - pf_last; pf_last = t_; } while (0)
#endif
#else
#define PF_START() ((void)0)
#define PF_MARK(i) ((void)0)
#endif

  // Issue side (D sweeps ahead of the consume side): cursor + incremental state, so that a step
  // costs a handful of instructions and everything rare sits behind ONE branch.
  CCursor ci{0, 0, 0, 1, 1};   // row group being requested
  bool ci_end = false;         // the stream has ended: the steps still issue loads (of one cached line: every
                               // step has the same set of loads)
  CIssueL Li;
  CConsL Lc;
  CFillL Lf;
  {
    enter_layer(ci);
    if (ci.L >= n_layers) return;     // (whole workgroup: nothing to do)
#if VPTQ_K256C_PROF
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf_pro[0]) :: "memory");
#endif
    const CLayerArgs L0 = c_load_layer(ci.L);
#if VPTQ_K256C_PROF
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pf_pro[1]) :: "memory");
#endif
    Li = c_issue_of(L0);
    Lc = c_cons_of(L0);
    Lf = CFillL{L0.cent, L0.rcent};
  }
  // SEL: the layer's threshold (magnitude bits of the 16-bit type) and where its corrections are (null: no hot block),
  // through the scalar cache
  const float* corr = nullptr;
  auto load_thr = [&](int L) __attribute__((always_inline)) -> uint32_t {
    uint32_t t = 0x7fffu;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SEL) {
      typedef int i4_t __attribute__((ext_vector_type(4)));
      const uint32_t* const tp = sync + 4 * __builtin_amdgcn_readfirstlane(L);
      i4_t h;
      asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(tp) : "memory");
      t = (uint32_t)h[0];
      corr = h[1] != 0 ? (const float*)((const char*)sync + (uint32_t)h[2]) : nullptr;
    }
#endif
    return t;
  };
  uint32_t thr = load_thr(ci.L);
  CCursor cc = ci;             // row group being consumed
  const uint32_t lane_chunk2 = ((uint32_t)lane >> 2) * 16u;   // byte offset of this lane's 8 index elements in a block
  uint32_t i_rowoff[kCSub];   // per lane: byte offset of its vector-row (subgroup q) in the layer's index tensor
  int i_col2 = 0;             // byte offset (2 x column) of the wave's block in the sweep to request
  int i_left = 0;             // sweeps of the row group still to request
  int i_max8 = 0, i_max2 = 0;   // 2 (G - 8), 2 (G - 2): columns past G re-read the last ones
  auto issue_row_group = [&]() __attribute__((always_inline)) {
    const int row0 = ci.rg * kCRows + (wave / kCColWaves) * kCRowsW;
#pragma unroll
    for (int q = 0; q < kCSub; ++q) {
      const int want = row0 + 4 * q + (int)jrow;
      const int r = want < Li.N ? want : Li.N - 1;   // rows past N re-read the last row (not stored)
      i_rowoff[q] = (uint32_t)r * ((uint32_t)Li.row_words * 4u);
    }
    i_col2 = (wave % kCColWaves) * (kCBlockCols * 2);
    i_left = ci.ns;
    i_max8 = (Li.G - 8) * 2;
    i_max2 = (Li.G - 2) * 2;
    if (ci_end) {
      // past the end of the stream the steps still issue their loads (the waits stay counted); every lane reads the
      // first bytes of the last layer's tensors - lines that are in the L2 - instead of re-reading the last row
      // group from HBM (D sweeps x 32 KiB per workgroup = 24 MiB per launch, 4.5 % of a 32-layer launch's traffic)
#pragma unroll
      for (int q = 0; q < kCSub; ++q) i_rowoff[q] = 0u;
      i_max8 = 0;
      i_max2 = 0;
    }
  };
  issue_row_group();
  auto issue_next_row_group = [&]() {   // cold
    if (!ci_end) {
      CCursor n = ci;
      n.rg += 1;
      if (n.rg >= n.re) {
        PF_START();
        ++n.L;
        enter_layer(n);
        if (n.L < n_layers) { ci = n; Li = c_issue_of(c_load_layer(n.L)); }
        else ci_end = true;
        PF_MARK(7);
      } else {
        ci = n;
      }
    }
    issue_row_group();
  };

  if (tid < 16) slot_cnt[tid] = 0u;
  __syncthreads();   // the only barrier: counters zeroed before anybody uses them

  // ---- image fill by LDS-DMA: wave w brings rows 16 w .. 16 w + 15 (4 instructions of 4 rows; 8 waves: 32 rows, 8;
  // lane l = unit l & 15 of row l >> 4: 16 bytes of entry (row) of table (unit >> 3)).  Invisible
  // loads can only make the compiler's counted waits stricter, never looser (vmcnt retires in order).
  auto fill_image = [&](const CFillL& F, uint32_t buf) __attribute__((always_inline)) {
    const uint32_t unit = (uint32_t)lane & 15u, r4 = (uint32_t)lane >> 4;
    // (table choice by arithmetic: as a per-lane select of two pointers the compiler built a two-entry
    // array in scratch memory and indexed it - a scratch load, waited for with vmcnt(0))
    const uint64_t tc = (uint64_t)(uintptr_t)F.cent, tr = (uint64_t)(uintptr_t)F.rcent;
    const uint64_t pick = (unit >> 3) ? ~0ull : 0ull;
    constexpr uint32_t kRowsW = 256u / (uint32_t)kCWaves;   // image rows per wave
    const uint64_t va = tc + ((tr - tc) & pick) + (uint64_t)(((uint32_t)wave * kRowsW + r4) * 16u);
    const uint32_t dst = buf * kCImgBytes + (uint32_t)wave * kRowsW * 256u;
#pragma unroll
    for (int i = 0; i < (int)(kRowsW / 4u); ++i) {
      const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)i * 1024u));
      const uint64_t v = va + (uint64_t)(i * 64);
      uint32_t keep_m0;   // (M0 belongs to the compiler: saved and restored inside the statement)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0) : "v"(v), "s"(d) : "memory");
    }
  };
  // Hand-over between the waves goes through LDS only, and a wave's LDS operations execute in
  // order: fences restricted to the local address space.  An ordinary workgroup-scope release also
  // waits for every global load in flight (s_waitcnt vmcnt(0)) - here the whole queue of index
  // words requested ahead.
  // (VPTQ_K256C_LDS_FENCE=0: the release is a compiler barrier only.  The fence makes the wave wait for its LDS
  // writes to complete - one LDS round trip behind 16 waves' gathers - before it issues the counter update; the LDS
  // executes one wave's operations in the order they were issued, so the update cannot overtake the writes.)
#if VPTQ_K256C_LDS_FENCE
  auto lds_release = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); };
#else
  auto lds_release = [&]() { asm volatile("" ::: "memory"); };
#endif
  auto lds_acquire = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); };
  auto lds_inc = [&](uint32_t* p) __attribute__((always_inline)) {
    lds_release();
    if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto lds_wait_ge = [&](uint32_t* p, uint32_t need) __attribute__((always_inline)) {
    if constexpr ((VPTQ_K256C_ABLATE & 8) != 0) { if (p >= free_cnt && p < free_cnt + 4) return; }
#if VPTQ_K256C_SPIN_LIMIT
    for (int it = 0; it < VPTQ_K256C_SPIN_LIMIT; ++it) {
      if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
#else
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
      __builtin_amdgcn_s_sleep(1);
#endif
    lds_acquire();
  };


  // ---- loads of one sweep
  u32x4 iw[D][kCSub];
  uint32_t xr[D], sr[D], br[D];
  int q_layer[D], q_col2[D];   // DEP: layer and block offset of the sweep in each queue slot
  constexpr int kLPS = ((VPTQ_K256C_ABLATE & 4) ? 0 : 3) + ((VPTQ_K256C_ABLATE & 16) ? 0 : kCSub);   // vector loads per sweep
  auto load_x = [&](auto slot_c, const uint16_t* xp, int max2, int col2) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    const int want = col2 + 4 * lane;
    const uint32_t* const xa32 = (const uint32_t*)as_global((const char*)xp + (uint32_t)(want < max2 ? want : max2));
    // DEP: another workgroup wrote it in this launch - a device-coherent load (sc1)
    xr[S] = DEP ? __hip_atomic_load(xa32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *xa32;
  };
  auto issue = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    const int want = i_col2 + (int)lane_chunk2;
    const uint32_t coff = (uint32_t)(want < i_max8 ? want : i_max8);
    if constexpr ((VPTQ_K256C_ABLATE & 4) != 0) {
      xr[S] = coff; sr[S] = 0x3c003c00u; br[S] = coff ^ 0x1234u;
      asm volatile("" : "+v"(xr[S]), "+v"(sr[S]), "+v"(br[S]));
    } else {
      const int want2 = i_col2 + 4 * lane;
      const uint32_t c2 = (uint32_t)(want2 < i_max2 ? want2 : i_max2);
      load_x(slot_c, Li.x, i_max2, i_col2);
      sr[S] = *(const uint32_t*)as_global((const char*)Li.scale + c2);
      br[S] = *(const uint32_t*)as_global((const char*)Li.wbias + c2);
    }
#pragma unroll
    for (int q = 0; q < kCSub; ++q) {
      if constexpr ((VPTQ_K256C_ABLATE & 16) != 0) {   // no index loads: what the consume side alone can do
        iw[S][q] = u32x4{i_rowoff[q] + coff, coff * 2654435761u, i_rowoff[q] ^ 0x5a5a5a5au, coff + 0x01234567u};
        asm volatile("" : "+v"(iw[S][q]));
      } else {
        iw[S][q] = __builtin_nontemporal_load((const u32x4*)as_global((const char*)Li.idx + (i_rowoff[q] + coff)));
      }
    }
    if (DEP) { q_layer[S] = ci_end ? -1 : ci.L; q_col2[S] = i_col2; }
    i_col2 += kCSweepCols * 2;
    if (--i_left == 0) issue_next_row_group();
  };

  // ---- state of the consume side
  f32x4 acc[kCSub][2];   // [subgroup][outputs 0-3 / 4-7 of this lane's vector-row]
  float accb = 0.f;
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < kCSub; ++q) { acc[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    accb = 0.f;
  };
  int c_col = (wave % kCColWaves) * kCBlockCols;   // first column of the wave's block in the sweep being consumed
  int c_left = cc.ns;               // sweeps of the row group still to consume
  bool done = false;
  uint32_t use = 0;            // how many layers this workgroup has entered before the current one
  uint32_t q_done = 0;         // row groups this workgroup has finished
  bool fill_pending = false;   // the next layer's image has not been requested yet
  int land_steps = 0;          // > 0: a fill was requested D - land_steps step ends ago
  // "everything but the youngest n sweeps' loads has landed".  vmcnt retires in order: at the end
  // of the j-th step after a fill the fill is older than j + 1 sweeps; at j = D - 1 those are
  // exactly the D sweeps in flight.
  auto wait_all_but = [&](int steps) __attribute__((always_inline)) {
    switch (steps) {
      case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 1) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 2) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 3) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 4) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 5) : "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * 6) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  static_assert(kLPS * kCDepth <= 63, "vmcnt is a 6-bit counter");
  CCursor cf = cc;             // next layer with work (its image goes to buffer (use + 1) & 1)
  auto plan_fill = [&]() __attribute__((always_inline)) {
    cf = cc;
    cf.L = cc.L + 1;
    enter_layer(cf);
    fill_pending = cf.L < n_layers;
    if (fill_pending) Lf = c_load_fill(cf.L);
  };

  // ---- counters, looked at without waiting: lane i of `pre` = counter word i (slot_cnt 0-3,
  // slot_done 4-7, free_cnt 8-9, ready_cnt 10-11), read at the START of every step - the value is
  // there when the sweep has been consumed.  A read issued where its value is needed sits behind the
  // gathers of 16 waves; counters only grow, so an old value can only err towards "not yet" (then
  // the wave polls, or, for the fill, looks again one step later).
  uint32_t pre = 0;
  auto peek_counters = [&]() __attribute__((always_inline)) {
    pre = __hip_atomic_load((uint32_t*)(smem + kCCntOff) + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto peeked = [&](uint32_t word) __attribute__((always_inline)) -> uint32_t {
    return (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)word);
  };

  // reduce over the 16 waves and store.  Per subgroup lane (blk, j) holds 8 partial outputs of
  // vector-row j for its column chunk: sum over the 16 chunks of the wave - lane bits 5 and 4 by
  // swap-and-add (halving the values carried), bits 3 and 2 by DPP row rotations.  Across the waves
  // through LDS slots + arrival counters instead of a barrier.  Nobody waits for an answer from the
  // LDS here: a wave deposits its partial sums, counts itself in and goes on; row group q is summed
  // and stored by wave q mod 16 when it comes back with row group q + 1 (by then the other 15 have
  // arrived, as a rule), or at once if it was the workgroup's last one of the layer.
  bool fin_pending = false;    // this wave still has to sum and store row group fin_rg (slot fin_slot)
  int fin_rg = 0;
  uint32_t fin_slot = 0, fin_q = 0;
  int bal = 1;                 // VPTQ_K256C_PRIO: issue priority of this wave's consume phase
  uint32_t arrived_v = 0;      // lane 0: this wave's arrival number at its last row group (in flight until looked at)
  auto finalize = [&]() __attribute__((always_inline)) {
    const uint32_t slot = fin_slot;
    lds_wait_ge(&slot_cnt[slot], (uint32_t)kCWaves);
    const int rg = fin_rg;
    {
      const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const float* const pr0 = red + slot * (kCWaves * kCOutW);
#pragma unroll
      for (int fin_pass = 0; fin_pass < (kCOut + 63) / 64; ++fin_pass) {
        const int part = (64 * fin_pass) / kCOutW;             // row part whose waves hold these outputs
        const float* const pr = pr0 + part * (kCColWaves * kCOutW);
        // sum b x: one term per wave of the part (every part forms the same sum; lanes past the wave count add 0)
        const float bpart = (ln & 15) < kCColWaves ? red_b[slot * kCWaves + part * kCColWaves + (ln & 15) % kCColWaves] : 0.f;
        const float bdot = row16_allsum(bpart);
        float sum;
        int ol;   // output (of this part of the row group) this lane stores
        bool mine;
        if constexpr (kCSub >= 2) {
          // 64 outputs per pass: lane = output, one partial per wave, summed as a tree
          ol = ln + (64 * fin_pass) % kCOutW; mine = true;
          float s[kCColWaves / 4];
#pragma unroll
          for (int k = 0; k < kCColWaves / 4; ++k)
            s[k] = (pr[(4 * k) * kCOutW + ol] + pr[(4 * k + 1) * kCOutW + ol]) + (pr[(4 * k + 2) * kCOutW + ol] + pr[(4 * k + 3) * kCOutW + ol]);
          if constexpr (kCColWaves == 16) sum = (s[0] + s[1]) + (s[2] + s[3]);
          else if constexpr (kCColWaves == 8) sum = s[0] + s[1];
          else sum = s[0];
        } else {
          // 32 outputs: lane (half, o) sums 8 of the 16 waves' partials, one lane swap joins the halves
          const int half = ln >> 5;
          ol = ln & 31; mine = ln < 32;
          const float* const ps = pr + (half * 8) * kCOutW + ol;
          const float s0 = (ps[0] + ps[kCOutW]) + (ps[2 * kCOutW] + ps[3 * kCOutW]);
          const float s1 = (ps[4 * kCOutW] + ps[5 * kCOutW]) + (ps[6 * kCOutW] + ps[7 * kCOutW]);
          const float hs = s0 + s1;
          auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(hs), __float_as_uint(hs), false, false);
          sum = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        const int row = rg * kCRows + part * kCRowsW + (ol >> 3);
        const int o = row * 8 + (ol & 7);
        const bool store = mine && row < Lc.N && o < Lc.O;
        float bv = 0.f;
        if (store && Lc.bias) bv = DT::to_float(as_global(Lc.bias)[o]);
        float total = sum + bdot;
        if constexpr (SEL) {
          if (corr != nullptr && store) total += corr[o];   // (the hot blocks' columns, reference roundings)
        }
        if (store) {
          if (out_f32) ((float*)as_global(Lc.y))[o] = total + bv;
          else if (DEP)   // write-through at device scope (sc1): another workgroup reads it in this launch
            __hip_atomic_store(as_global(Lc.y) + o, DT::from_float(total + bv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else as_global(Lc.y)[o] = DT::from_float(total + bv);
        }
      }
      if (DEP) {
        // the write-through stores of this wave have reached memory (every storing wave drains)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rg + 1 == cc.re) {
          // this workgroup's part of the layer is stored: raise its flag (device scope).  The wave
          // that stores the last row group waits for all 16 waves' partial sums of it, and a wave
          // deposits those only after it has stored (and drained) the row group it was in charge
          // of: all earlier row groups' stores were drained before this point.  No release fence: an
          // agent-scope release writes the L2 back (buffer_wbl2, ~1.3 us, serialised per XCD:
          // 41 us per layer for 256 workgroups); one flag word per workgroup instead of a counter:
          // 256 atomic increments of one word are serialised at the memory side (53 us per layer).
          if (lane == 0)
            __hip_atomic_store(&sync[(size_t)cc.L * kCFlagStride + (uint32_t)bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (lane == 0) {
        slot_cnt[slot] = 0u;
        lds_release();
        __hip_atomic_store(&slot_done[slot], fin_q + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    fin_pending = false;
  };
  auto finish = [&]() __attribute__((always_inline)) {
    PF_START();
    if (fin_pending) finalize();   // the previous row group (same layer: a layer's last one is not left pending)
    PF_MARK(1);
    const uint32_t slot = q_done % (uint32_t)kCSlots;
    if (q_done >= (uint32_t)kCSlots) {   // the slot's previous row group has been stored
      const uint32_t need = q_done - (uint32_t)kCSlots + 1u;
      if (peeked(kCSlots + slot) < need) lds_wait_ge(&slot_done[slot], need);
    }
    float* const rs = red + (slot * kCWaves + wave) * kCOutW;
#pragma unroll
    for (int q = 0; q < kCSub; ++q) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = acc[q][0][i]; v[4 + i] = acc[q][1][i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = row_ror_add<4>(row_ror_add<8>(v[i]));
      if ((lane & 12) == 0) {   // lane l now holds outputs 4 bit5 + 2 bit4 + {0, 1} of row l & 3
        const int o8 = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *(f32x2*)&rs[q * 32 + (int)jrow * 8 + o8] = f32x2{v[0], v[1]};
      }
    }
    const float sb = wave_sum(accb);
    if (lane == 0) red_b[slot * kCWaves + wave] = sb;
#if VPTQ_K256C_BALANCE
    // issue priority for the next row group by arrival order at the PREVIOUS one (its number has
    // come back by now): the early ones yield
    const uint32_t before = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived_v);
#if VPTQ_K256C_PRIO
    bal = before * (16u / (uint32_t)kCWaves) < 5u ? 0 : before * (16u / (uint32_t)kCWaves) < 11u ? 1 : 2;
#else
    if (before < 4u) __builtin_amdgcn_s_setprio(0);
    else if (before < 8u) __builtin_amdgcn_s_setprio(1);
    else if (before < 12u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#endif
#endif
    lds_release();
    if (lane == 0)
      arrived_v = __hip_atomic_fetch_add(&slot_cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((q_done % (uint32_t)kCWaves) == (uint32_t)wave) {
      fin_pending = true; fin_rg = cc.rg; fin_slot = slot; fin_q = q_done;
    }
    ++q_done;
    PF_MARK(0);
  };

  // ---- one sweep of this wave: 8 columns x kCSub vector-rows per lane.  A unit = one index: 2 perms
  // for the gather addresses, 2 gathers (two units ahead of the arithmetic), 4 MFMAs (main /
  // residual entry x outputs 0-3 / 4-7); the x operand (2 perms) is built once per column and
  // shared by the subgroups.
  auto consume = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    u32x4 sq = u32x4{0u, 0u, 0u, 0u}, bq = u32x4{0u, 0u, 0u, 0u};
    {
      const uint32_t keep = c_col + 2 * lane < Lc.G ? 0xffffffffu : 0u;
      const uint32_t xv = xr[S] & keep;
      if constexpr (EXACT) {
        // x, scale and bias of this lane's two columns side by side in the wave's slot (no folding: the weights are
        // rebuilt with the reference's roundings, so there is no sum b x either)
        *(lds_u32_t*)(uintptr_t)st_addr = xv;
        *(lds_u32_t*)(uintptr_t)(st_addr + 256u) = sr[S];
        *(lds_u32_t*)(uintptr_t)(st_addr + 512u) = br[S];
      } else if constexpr (SEL) {
        // the folded form; a sweep that holds a HOT column (|f16(s x)| >= thr; NaN / inf compare as large) contributes nothing
        // here - its 128 columns x the layer's rows come in through `corr`, rebuilt with the reference's roundings by
        // k256c_hot_kernel in front of this launch
        const uint32_t xp = DT::mul2(xv, sr[S]);
        const uint32_t m16 = (xp & 0x7fffu) > ((xp >> 16) & 0x7fffu) ? (xp & 0x7fffu) : ((xp >> 16) & 0x7fffu);
        const bool hot = __builtin_amdgcn_ballot_w64(m16 >= thr) != 0ull;   // (wave-uniform)
        const uint32_t xe = hot ? 0u : xv;
        accb = DT::dot2(xe, br[S], accb);
        asm volatile("" : "+v"(accb));
        *(lds_u32_t*)(uintptr_t)st_addr = hot ? 0u : xp;
      } else {
        // activations: f16(s x) of this lane's two columns into the wave's slot, sum b x
        accb = DT::dot2(xv, br[S], accb);
        // (anchored here: left alone, the compiler sinks this towards its use in finish(), keeps the
        // loaded register alive across the loop edge and copies it there - behind a wait for the
        // loads the step has just issued)
        asm volatile("" : "+v"(accb));
        *(lds_u32_t*)(uintptr_t)st_addr = DT::mul2(xv, sr[S]);
      }
    }
    const u32x4 xq = lds_load16(xq_addr);
    if constexpr (EXACT) { sq = lds_load16(xq_addr + 256u); bq = lds_load16(xq_addr + 512u); }
    // (bf16 reference roundings: 18 registers of rounding chains per index - one gather less in flight, or the loop spills)
    constexpr int kUnits = 8 * kCSub, kAhead = (EXACT && std::is_same<DT, BF16>::value) ? 1 : VPTQ_K256C_AHEAD, kNB = kAhead + 1;
    u32x4 cv[kNB], rv[kNB];
    auto gather = [&](int t) {   // unit t = column t / kCSub of subgroup t % kCSub
      const int u = t / kCSub, q = t % kCSub;
      const uint32_t w = iw[S][q][u >> 1];
      const uint32_t aC = __builtin_amdgcn_perm(w, baseA, selGA[u & 1]);
      const uint32_t aR = __builtin_amdgcn_perm(w, baseB, selGB[u & 1]);
      if constexpr ((VPTQ_K256C_ABLATE & 2) != 0) {
        asm volatile("" :: "v"(aC), "v"(aR));
        cv[t % kNB] = iw[S][q]; rv[t % kNB] = iw[S][q];
      } else {
        cv[t % kNB] = lds_load16(aC);
        rv[t % kNB] = lds_load16(aR);
      }
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t) gather(t);
    u32x2 xo = u32x2{0u, 0u};
    // bf16, reference roundings (round 6): the matrix-pipe form of gemv_k256m.hip (one-hot operands widen for free: 11 MFMAs + 12
    // conversions per index); per column: the scaled one-hot and the bias in every register
    [[maybe_unused]] u32x2 so = u32x2{0u, 0u};
    [[maybe_unused]] f32x4 bvec = f32x4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] const u32x2 bf_id = u32x2{__builtin_amdgcn_perm(0x3f803f80u, 0u, selA[0]), __builtin_amdgcn_perm(0x3f803f80u, 0u, selB[0])};
#pragma unroll
    for (int t = 0; t < kUnits; ++t) {
      // fenced: left alone, the scheduler sinks the gathers next to their use (LDS latency exposed)
      __builtin_amdgcn_sched_barrier(0);
      if (t + kAhead < kUnits) gather(t + kAhead);
      __builtin_amdgcn_sched_barrier(0);
      const int u = t / kCSub, q = t % kCSub;
      if (q == 0)
        xo = u32x2{__builtin_amdgcn_perm(xq[u >> 1], 0u, selA[u & 1]), __builtin_amdgcn_perm(xq[u >> 1], 0u, selB[u & 1])};
      const u32x4 c = cv[t % kNB], r = rv[t % kNB];
      if constexpr (EXACT && std::is_same<DT, BF16>::value) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
          so = u32x2{__builtin_amdgcn_perm(sq[u >> 1], 0u, selA[u & 1]), __builtin_amdgcn_perm(sq[u >> 1], 0u, selB[u & 1])};
          const u32x2 one_hot = u32x2{(u & 3) == 0 ? 0x00003f80u : (u & 3) == 1 ? 0x3f800000u : 0u,
                                      (u & 3) == 2 ? 0x00003f80u : (u & 3) == 3 ? 0x3f800000u : 0u};
          bvec = DT::mfma4(u32x2{bq[(u >> 2) * 2], bq[(u >> 2) * 2 + 1]}, one_hot, z);
        }
        f32x4 d0 = DT::mfma4(bf_id, u32x2{c[0], c[1]}, z);
        f32x4 d1 = DT::mfma4(bf_id, u32x2{c[2], c[3]}, z);
        d0 = DT::mfma4(bf_id, u32x2{r[0], r[1]}, d0);
        d1 = DT::mfma4(bf_id, u32x2{r[2], r[3]}, d1);
        u32x2 wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};
        u32x2 wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
        d0 = DT::mfma4(so, wa, z);
        d1 = DT::mfma4(so, wb, z);
        wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};
        wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
        d0 = DT::mfma4(bf_id, wa, bvec);
        d1 = DT::mfma4(bf_id, wb, bvec);
        wa = u32x2{DT::pack(d0[0], d0[1]), DT::pack(d0[2], d0[3])};
        wb = u32x2{DT::pack(d1[0], d1[1]), DT::pack(d1[2], d1[3])};
        acc[q][0] = DT::mfma4(xo, wa, acc[q][0]);
        acc[q][1] = DT::mfma4(xo, wb, acc[q][1]);
      } else if constexpr (EXACT) {
        // both row subgroups of column u together, stage-major over 8 independent chains (dependent packed
        // operations back to back cost wait states); the scale / bias half of the column is picked by op_sel
        if ((q & 1) == 0) {   // (row subgroups q, q + 1)
          const u32x4 c1 = cv[(t + 1) % kNB], r1 = rv[(t + 1) % kNB];
          uint32_t w[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) { w[k] = DT::add2(c[k], r[k]); w[4 + k] = DT::add2(c1[k], r1[k]); }
          asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] = DT::mul2_bcast(w[k], sq[u >> 1], u & 1);
          asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] = DT::add2_bcast(w[k], bq[u >> 1], u & 1);
          acc[q][0] = DT::mfma4(xo, u32x2{w[0], w[1]}, acc[q][0]);
          acc[q][1] = DT::mfma4(xo, u32x2{w[2], w[3]}, acc[q][1]);
          acc[q + 1][0] = DT::mfma4(xo, u32x2{w[4], w[5]}, acc[q + 1][0]);
          acc[q + 1][1] = DT::mfma4(xo, u32x2{w[6], w[7]}, acc[q + 1][1]);
        }
      } else if constexpr ((VPTQ_K256C_ABLATE & 1) != 0) {
        asm volatile("" :: "v"(c), "v"(r), "v"(xo));
      } else if constexpr (VPTQ_K256C_PREADD != 0 && std::is_same<DT, F16>::value) {
        // (experiment) f16(c + r) first - the reference's own first rounding (csrc/kernels/quant_gemv.cuh) - then 2
        // instead of 4 MFMAs per index: the MFMAs are the larger part of the energy, and the kernel runs at the
        // package power limit
        const u32x4 w = u32x4{DT::add2(c[0], r[0]), DT::add2(c[1], r[1]), DT::add2(c[2], r[2]), DT::add2(c[3], r[3])};
        acc[q][0] = DT::mfma4(xo, u32x2{w[0], w[1]}, acc[q][0]);
        acc[q][1] = DT::mfma4(xo, u32x2{w[2], w[3]}, acc[q][1]);
      } else {
        acc[q][0] = DT::mfma4(xo, u32x2{c[0], c[1]}, acc[q][0]);
        acc[q][1] = DT::mfma4(xo, u32x2{c[2], c[3]}, acc[q][1]);
        acc[q][0] = DT::mfma4(xo, u32x2{r[0], r[1]}, acc[q][0]);
        acc[q][1] = DT::mfma4(xo, u32x2{r[2], r[3]}, acc[q][1]);
      }
    }
  };

  // DEP: layer L > 0 reads what layer L - 1 of this launch wrote: wait until every workgroup that
  // owns a part of it has raised its flag, then fetch x for the sweeps of layer L that are already
  // in the queue (their index words, scales and bias values were requested ahead; x could not be)
  auto dep_enter = [&](int L) __attribute__((always_inline)) {
    if (L == 0) return;
    // one wave per workgroup polls (4096 waves polling starve the stores that feed them); the
    // others wait for its word in LDS
    if (wave == 0) {
      // the workgroups that own a block of layer L - 1: 0 .. blocks - 1 (every layer of a dependent
      // chain starts at workgroup 0); lane i looks at the flags of workgroups 4i .. 4i + 3
      const c_head4_t hp = c_load_heads(L - 1);
      const int ng = (int)((uint32_t)hp[1] & 0xffffffu), rpw = (int)((uint32_t)hp[0] >> 16);
      const int blocks = (ng + rpw - 1) / rpw;
      const uint32_t* const fl = sync + (size_t)(L - 1) * kCFlagStride + 4 * lane;
      for (int it = 0; VPTQ_K256C_SPIN_LIMIT == 0 || it < (VPTQ_K256C_SPIN_LIMIT ? VPTQ_K256C_SPIN_LIMIT : 1); ++it) {
        u32x4 f;
        // (coherent at device scope: sc0 sc1; relaxed polls; no acquire fence - ~1.7 us per workgroup:
        // the only data another workgroup produced is x, read with device-coherent loads)
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(f) : "v"(fl) : "memory");
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) ok = ok && (f[q] != 0u || 4 * lane + q >= blocks);
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        __builtin_amdgcn_s_sleep(8);
      }
      lds_release();
      if (lane == 0) __hip_atomic_store(dep_seen, (uint32_t)L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      lds_wait_ge(dep_seen, (uint32_t)L);
    }
    const CLayerArgs Lx = c_load_layer(L);
    const uint16_t* const xp = Lx.x;
    const int max2 = (Lx.G - 2) * 2;
    auto reload = [&](auto slot_c) __attribute__((always_inline)) {
      constexpr int S = decltype(slot_c)::value;
      if (q_layer[S] == L) load_x(slot_c, xp, max2, q_col2[S]);
    };
    reload(std::integral_constant<int, 0>{});
    reload(std::integral_constant<int, 1>{});
    if constexpr (D > 2) reload(std::integral_constant<int, (D > 2 ? 2 : 0)>{});
    if constexpr (D > 3) reload(std::integral_constant<int, (D > 3 ? 3 : 0)>{});
    if constexpr (D > 4) reload(std::integral_constant<int, (D > 4 ? 4 : 0)>{});
    if constexpr (D > 5) reload(std::integral_constant<int, (D > 5 ? 5 : 0)>{});
  };

  // ---- prologue: image of the first layer into buffer 0, first D sweeps requested
  PF_PRO(2);
  fill_image(Lf, 0u);
  c_for_slots([&](auto slot_c) {
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);   // (the slots in issue order: the counted waits of the loop rely on it)
  });
  PF_PRO(3);
  wait_all_but(D);   // the 4 fill instructions are older than the loads above
  PF_PRO(4);
  lds_inc(&ready_cnt[0]);
  plan_fill();
  PF_PRO(5);
  zero_acc();
  if (DEP) dep_enter(cc.L);
  lds_wait_ge(&ready_cnt[0], (uint32_t)kCWaves);
  PF_PRO(6);

  // ---- rare events, each behind one branch of the step
  // the next layer's image: requested as soon as its buffer is free (every wave has left the layer
  // before the current one), BEFORE the step's loads (a younger invisible load would make the next
  // step's counted wait cover those too); D step ends later only the D sweeps in flight are younger
  // than it, so it has landed once everything older has - which a wave that keeps pace has waited for
  auto fill_events_before_issue = [&]() __attribute__((always_inline)) {
    const uint32_t nb = (use + 1u) & 1u;
    const uint32_t need = (uint32_t)kCWaves * ((use + 1u) >> 1);
    if (peeked(2 * kCSlots + nb) >= need) {
      lds_acquire();
      fill_image(Lf, nb);
      fill_pending = false;
      land_steps = D;
    }
  };
  auto fill_events_after_issue = [&]() __attribute__((always_inline)) {
    if (--land_steps == 0) {
      wait_all_but(D);
      lds_inc(&ready_cnt[(use + 1u) & 1u]);
    }
  };
  // the row group is complete: sums, then the next row group, the next layer, or the end
  auto row_group_done = [&]() __attribute__((always_inline)) {
    finish();
    zero_acc();
    c_col = (wave % kCColWaves) * kCBlockCols;
    c_left = cc.ns;
    if (cc.rg + 1 < cc.re) { cc.rg += 1; return; }
    if (fin_pending) finalize();   // (the layer's arguments are about to change; DEP: its flag is due)
    PF_MARK(2);
    // leaving the layer: its image buffer is free once every wave has said so
    lds_inc(&free_cnt[use & 1u]);
    if (cf.L >= n_layers) {
      // (the steps that remain in this loop iteration consume re-read sweeps into accumulators
      // nobody looks at.  The loop has ONE exit, at its end: an exit between two steps becomes,
      // after control-flow structurisation, an edge into the loop header on which the queue slots
      // are in another order, and every counted wait of the first step degrades to vmcnt(0).)
      done = true;
      c_left = 0x7fffffff;
      return;
    }
    const uint32_t nb = (use + 1u) & 1u;
    if (fill_pending) {   // (rare: the buffer was not free at any step boundary)
      lds_wait_ge(&free_cnt[nb], (uint32_t)kCWaves * ((use + 1u) >> 1));
      fill_image(Lf, nb);
      fill_pending = false;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_inc(&ready_cnt[nb]);
    } else if (land_steps > 0) {
      // requested D - land_steps step ends ago: that many sweeps are younger than the fill
      wait_all_but(D - land_steps);
      lds_inc(&ready_cnt[nb]);
      land_steps = 0;
    }
    PF_MARK(3);
    ++use;
    baseA ^= 0x10000u;
    baseB ^= 0x10000u;
    cc = cf;
    Lc = c_cons_of(c_load_layer(cc.L));
    if constexpr (SEL) thr = load_thr(cc.L);
    c_left = cc.ns;
    PF_MARK(4);
    if (DEP) dep_enter(cc.L);
    {
      const uint32_t need = (uint32_t)kCWaves * ((use >> 1) + 1u);
      if (peeked(2 * kCSlots + 2 + nb) < need) lds_wait_ge(&ready_cnt[nb], need);
      else lds_acquire();
    }
    PF_MARK(5);
    plan_fill();
    PF_MARK(6);
  };

#if VPTQ_K256C_PROF
  pf_t0 = now(0);
#endif
#if VPTQ_K256C_STAGGER
  // (experiment) the four waves of a SIMD (wave & 3 = SIMD) start a fraction of a step apart, so that one wave's
  // bookkeeping falls into the others' consume phases
  for (int g = 0; g < (wave >> 2); ++g) __builtin_amdgcn_s_sleep(VPTQ_K256C_STAGGER);
#endif
  // ---- main loop: one step = wait for sweep k, consume it, request sweep k + D into its queue slot
  auto step = [&](auto slot_c) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256C_PROF >= 3
    const unsigned long long tb = now(0);   // (before the peek: the stamp waits for every LDS operation in flight)
#endif
    peek_counters();
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256C_PROF && VPTQ_K256C_PROF < 3
    constexpr int SS = decltype(slot_c)::value;
    const unsigned long long ta = now(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLPS * (D - 1)) : "memory");
    const unsigned long long tb = now(iw[SS][0][0]);
    pf_wait += tb - ta;
#endif
#if VPTQ_K256C_PRIO
    if (bal == 0) __builtin_amdgcn_s_setprio(0);
    else if (bal == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(2);
#endif
    // (after the workgroup's last row group the remaining steps of the loop iteration only keep the loads counted:
    // no gathers, no MFMAs - the kernel runs at the package power limit, energy spent on sums nobody stores is time)
    if (!done) consume(slot_c);
#if VPTQ_K256C_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256C_PROF
    const unsigned long long tc = now(__float_as_uint(acc[0][0][0] + acc[0][1][0]));
    pf_cons += tc - tb;
#if VPTQ_K256C_PROF >= 2
    if (lane == 0 && pf_steps >= (unsigned long long)kCTlFirst && pf_steps < (unsigned long long)(kCTlFirst + kCTlSteps)) {
      uint32_t* const tl = (uint32_t*)(smem + kCTlOff) + ((uint32_t)wave * kCTlSteps + (uint32_t)(pf_steps - kCTlFirst)) * 2u;
      tl[0] = (uint32_t)tb; tl[1] = (uint32_t)tc;
    }
#endif
#endif
    if (fill_pending) fill_events_before_issue();
    __builtin_amdgcn_sched_barrier(0);
    issue(slot_c);
    __builtin_amdgcn_sched_barrier(0);
#if VPTQ_K256C_PROF && VPTQ_K256C_PROF < 3
    const unsigned long long td = now(0);
    pf_issue += td - tc;
#endif
    if (land_steps > 0) fill_events_after_issue();
    c_col += kCSweepCols;
    if (--c_left == 0) row_group_done();
#if VPTQ_K256C_PROF && VPTQ_K256C_PROF < 3
    pf_cold += now(0) - td;
#endif
#if VPTQ_K256C_PROF
    ++pf_steps;
#endif
  };
  do {
    c_for_slots(step);
  } while (!done);
#if VPTQ_K256C_PROF
  if (!DEP && sync && lane == 0) {
    unsigned long long* o = (unsigned long long*)sync + ((size_t)bid * kCWaves + wave) * kCProfWords;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[8 + i] = pf_x[i];
    o[0] = pf_wait; o[1] = pf_cons; o[2] = pf_issue; o[3] = pf_cold; o[4] = pf_steps; o[5] = now(0) - pf_t0;
    o[6] = pf_t0 - pf_entry;   // kernel entry -> first step (prologue)
    o[7] = pf_entry;           // absolute (s_memtime) entry stamp: spread of the workgroups' start
#pragma unroll
    for (int i = 0; i < 7; ++i) o[16 + i] = pf_pro[i] - pf_entry;   // prologue marks, clocks since kernel entry
    {
      uint32_t hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      o[23] = hw;   // wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13
    }
#if VPTQ_K256C_PROF >= 2
    {
      const uint32_t* const tl = (const uint32_t*)(smem + kCTlOff) + (uint32_t)wave * kCTlSteps * 2u;
      for (int i = 0; i < kCTlSteps; ++i) o[24 + i] = (unsigned long long)tl[2 * i] | ((unsigned long long)tl[2 * i + 1] << 32);
    }
#endif
  }
#endif
}

// ---- SEL: thresholds and corrections of a launch's layers (gemv_k256c_kernel<.., kCModeSel>).  grid = (row chunks, layers),
// 256 threads.  Every workgroup of a layer finds the layer's threshold and hot blocks itself (two passes over x and scale: 48 KiB
// out of the L2): thr = magnitude bits (16-bit type) of kappa x rms of f16(s x), a block of 128 columns is hot when one of its
// columns is at or above it.  Chunk 0 writes the layer's header; without a hot block that is all (dense activations: kappa = 6
// puts one Gaussian column in 5 x 10^8 above the line).  Otherwise the workgroup rebuilds the hot blocks' weights of ITS kHotRows
// vector-rows as the reference rounds them - w = f16(f16(f16(c + r) s) + b), vptq/ops/quant_gemm.py:121,155-156 - and stores
// sum w x (fp32) of those columns as the rows' corrections.  kappa: VPTQ_SELECTIVE_KAPPA; the study behind the 6:
// tools/hybrid_error_study.py, profiles/r06/selective_*.txt.
constexpr int kHotRows = 64;          // vector-rows per workgroup: 4 rounds of 16 (16 lanes per row, 8 columns per lane and hot block);
                                      // every workgroup repeats the threshold / hot-block search (64 KiB out of the L2): 16 rows per
                                      // workgroup made that search 10 us per launch of 32 layers (rocprofv3, round 6), 64: a quarter
constexpr int kHotMaxBlocks = 256;    // 32768 columns
constexpr int kHotMaxHot = 16;        // hot blocks per layer the corrections cover (the most dominant ones)
struct CHotArgs {
  uint32_t corr_off[kMaxGroup];   // byte offset of layer l's corrections from P.sync
  float kappa;
};
template <typename DT>
__global__ __launch_bounds__(256) void k256c_hot_kernel(const K256CParams P, const CHotArgs H) {
  __shared__ float part[4];
  __shared__ uint32_t mask[kHotMaxBlocks / 32];
  __shared__ __attribute__((aligned(16))) uint32_t cb[2][256 * 4];   // both codebooks, 16 bytes per entry
  const int L = (int)blockIdx.y, tid = (int)threadIdx.x;
  const CLayerArgs Ly = c_load_layer(L);
  uint32_t* const hdr = as_global(P.sync) + 4 * L;
  if (tid < kHotMaxBlocks / 32) mask[tid] = 0u;
  // the columns in batches of 8192 per workgroup: a thread's 16 pairs of a batch are requested together (one latency per batch, not
  // per pair: 32 dependent trips to the L2 made this search 7 us per launch); the first batch stays in registers for the second pass
  constexpr int kPer = 16;
  uint32_t xp0[kPer];
  float ss = 0.f;
  for (int base = 0; base < Ly.G; base += 512 * kPer) {
    uint32_t xv[kPer], sv[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = base + i * 512 + 2 * tid;
      const int cc = c < Ly.G ? c : 0;
      xv[i] = *(const uint32_t*)(Ly.x + cc);
      sv[i] = *(const uint32_t*)(Ly.scale + cc);
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const uint32_t xp = base + i * 512 + 2 * tid < Ly.G ? DT::mul2(xv[i], sv[i]) : 0u;
      if (base == 0) xp0[i] = xp;
      const float a = DT::lo(xp), b = DT::hi(xp);
      ss = __builtin_fmaf(a, a, __builtin_fmaf(b, b, ss));
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  uint32_t thr = DT::kInfBits;                     // inf / NaN sums: only inf / NaN columns are hot
  {
    const float tot = (part[0] + part[1]) + (part[2] + part[3]);
    const float t = H.kappa * __builtin_sqrtf(tot / (float)Ly.G);
    if (t < DT::kMaxFinite) {
      thr = (uint32_t)DT::from_float(t) & 0x7fffu;
      if (DT::to_float((uint16_t)thr) < t) thr += 1u;   // (rounded up: at or above the threshold)
    }
    if (thr == 0u) thr = 1u;                       // (all-zero activations: nothing is hot)
  }
  for (int base = 0; base < Ly.G; base += 512 * kPer) {
    uint32_t xq[kPer];
    if (base == 0) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) xq[i] = xp0[i];
    } else {
      uint32_t xv[kPer], sv[kPer];
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int c = base + i * 512 + 2 * tid;
        const int cc = c < Ly.G ? c : 0;
        xv[i] = *(const uint32_t*)(Ly.x + cc);
        sv[i] = *(const uint32_t*)(Ly.scale + cc);
      }
#pragma unroll
      for (int i = 0; i < kPer; ++i) xq[i] = base + i * 512 + 2 * tid < Ly.G ? DT::mul2(xv[i], sv[i]) : 0u;
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = base + i * 512 + 2 * tid;
      if ((xq[i] & 0x7fffu) >= thr || ((xq[i] >> 16) & 0x7fffu) >= thr) atomicOr(&mask[c >> 12], 1u << ((c >> 7) & 31));
    }
  }
  __syncthreads();
  int nhot = 0;
#pragma unroll
  for (int i = 0; i < kHotMaxBlocks / 32; ++i) nhot += __builtin_popcount(mask[i]);
  // more than kHotMaxHot: the threshold goes up by a quarter until the most dominant blocks remain - a token with dozens of
  // "dominant" blocks is a dense one (the folded form's regime), and the products below are latency-bound work.  Every workgroup
  // of the layer walks the same sequence; the chain kernel reads the final threshold from the header.
  for (int it = 0; it < 24 && nhot > kHotMaxHot; ++it) {
    __syncthreads();
    if (tid < kHotMaxBlocks / 32) mask[tid] = 0u;
    __syncthreads();
    const float tf = DT::to_float((uint16_t)thr) * 1.25f;
    thr = tf < DT::kMaxFinite ? ((uint32_t)DT::from_float(tf) & 0x7fffu) + 1u : DT::kInfBits;
    for (int base = 0; base < Ly.G; base += 512 * kPer) {
      uint32_t xq[kPer];
      if (base == 0) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) xq[i] = xp0[i];
      } else {
        uint32_t xv[kPer], sv[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
          const int c = base + i * 512 + 2 * tid;
          const int cc = c < Ly.G ? c : 0;
          xv[i] = *(const uint32_t*)(Ly.x + cc);
          sv[i] = *(const uint32_t*)(Ly.scale + cc);
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) xq[i] = base + i * 512 + 2 * tid < Ly.G ? DT::mul2(xv[i], sv[i]) : 0u;
      }
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int c = base + i * 512 + 2 * tid;
        if ((xq[i] & 0x7fffu) >= thr || ((xq[i] >> 16) & 0x7fffu) >= thr) atomicOr(&mask[c >> 12], 1u << ((c >> 7) & 31));
      }
    }
    __syncthreads();
    nhot = 0;
#pragma unroll
    for (int i = 0; i < kHotMaxBlocks / 32; ++i) nhot += __builtin_popcount(mask[i]);
  }
  if (blockIdx.x == 0 && tid == 0) {
    hdr[0] = thr; hdr[1] = (uint32_t)nhot; hdr[2] = H.corr_off[L]; hdr[3] = 0u;
  }
  if (nhot == 0 || (int)blockIdx.x * kHotRows >= Ly.N) return;
  // both codebooks: thread t copies entry t of each (16 bytes)
  *(u32x4*)&cb[0][tid * 4] = *(const u32x4*)(Ly.cent + tid * 4);
  *(u32x4*)&cb[1][tid * 4] = *(const u32x4*)(Ly.rcent + tid * 4);
  __syncthreads();
  const int sub = tid & 15;                        // 8 columns of the block
#pragma unroll 1
  for (int round = 0; round < kHotRows / 16; ++round) {
  const int row = (int)blockIdx.x * kHotRows + round * 16 + (tid >> 4);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int rr = row < Ly.N ? row : Ly.N - 1;
  const char* const irow = (const char*)Ly.idx + (size_t)rr * ((size_t)Ly.row_words * 4);
  for (int wd = 0; wd < kHotMaxBlocks / 32; ++wd) {
    uint32_t m = mask[wd];
    while (m) {
      const int bit = __builtin_ctz(m);
      m &= m - 1u;
      const int c0 = (wd * 32 + bit) * kCBlockCols + sub * 8;
      if (c0 >= Ly.G) continue;                    // (G is a multiple of 8)
      const u32x4 iw = *(const u32x4*)(irow + (size_t)c0 * 2);
      const u32x4 xq = *(const u32x4*)(Ly.x + c0), sq = *(const u32x4*)(Ly.scale + c0), bq = *(const u32x4*)(Ly.wbias + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t e = (iw[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        const u32x4 c = *(const u32x4*)&cb[0][(e & 255u) * 4], r = *(const u32x4*)&cb[1][(e >> 8) * 4];
        const float xf = DT::half_of(xq[j >> 1], j & 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t w = DT::add2(c[k], r[k]);
          w = DT::mul2_bcast(w, sq[j >> 1], j & 1);
          w = DT::add2_bcast(w, bq[j >> 1], j & 1);
          acc[2 * k] = DT::fma_lo(w, xf, acc[2 * k]);
          acc[2 * k + 1] = DT::fma_hi(w, xf, acc[2 * k + 1]);
        }
      }
    }
  }
  // the 16 lanes of a row
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = row16_allsum(acc[k]);
  if (sub == 0 && row < Ly.N) {
    float* const co = (float*)((char*)as_global(P.sync) + H.corr_off[L]) + (size_t)row * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (row * 8 + k < Ly.O) co[k] = acc[k];
  }
  }   // round
}
// bytes of a SEL launch's workspace: 16 per layer, then 4 per output
size_t gemv_k256c_selective_bytes(const VptqLayerDesc* descs, int n) {
  size_t b = ((size_t)n * 16 + 255) / 256 * 256;
  for (int i = 0; i < n; ++i) b += ((size_t)descs[i].num_indices * 8 * 4 + 255) / 256 * 256;
  return b;
}
static float c_kappa() {
  static std::atomic<int> milli{-1};
  if (milli < 0) { const char* e = vptq::tune_env("VPTQ_SELECTIVE_KAPPA"); const double v = e ? atof(e) : 0.0; milli = v > 0.0 ? (int)(v * 1000.0) : 6000; }
  return (float)milli.load() * 1e-3f;
}

// ---- host side -------------------------------------------------------------------
bool gemv_k256c_eligible(const VptqLayerDesc& d, int tokens) {
  return tokens == 1 && d.perm == nullptr && gemv_k256_eligible(d, 1) &&
         (((uintptr_t)d.centroids | (uintptr_t)d.res_centroids) & 15) == 0 &&
         d.num_indices <= 0xffffff && d.group_size <= 0xff * kCSweepCols;   // (the search header's fields)
}

// Row groups are dealt to the workgroups in blocks of consecutive ones, `rpw` per workgroup and
// layer.  Independent layers: at least kCMinSteps sweeps per visit of a layer (the next layer's
// image is requested at the start of the visit and lands D sweeps later; a layer that gives every
// workgroup one short row group would make all of them wait for it), so a small layer occupies
// only some of the workgroups and the next layers run beside it.  Dependent layers follow each
// other anyway: every layer is spread over all workgroups.
#ifndef VPTQ_K256C_MIN_STEPS
#define VPTQ_K256C_MIN_STEPS 8
#endif
constexpr int kCMinSteps = VPTQ_K256C_MIN_STEPS;
static int c_groups(const VptqLayerDesc& d) { return (d.num_indices + kCRows - 1) / kCRows; }
static int c_rows_per_wg(const VptqLayerDesc& d, int cus, bool wide) {
  const int ng = c_groups(d), ns = (d.group_size + kCSweepCols - 1) / kCSweepCols;
  int rpw = (ng + cus - 1) / cus;
  if (wide) {
    const int want = (kCMinSteps + ns - 1) / ns;
    rpw = want > rpw ? want : rpw;
  }
  return rpw < 1 ? 1 : rpw;
}
static long long c_blocks(const VptqLayerDesc* descs, int n, int cus, bool wide) {
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const int rpw = c_rows_per_wg(descs[i], cus, wide);
    total += (c_groups(descs[i]) + rpw - 1) / rpw;
  }
  return total;
}

static int c_device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0
                   ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

static int c_workgroups(bool dependent) {
  static std::atomic<int> forced_wgs{-1};  // VPTQ_K256C_WGS: tuning override of the workgroup count
  if (forced_wgs < 0) { const char* e = vptq::tune_env("VPTQ_K256C_WGS"); forced_wgs = e ? atoi(e) : 0; }
  const int fw = forced_wgs.load();
  const int cus = fw > 0 ? fw : c_device_cus();
  return dependent && cus > kCFlagStride ? kCFlagStride : cus;
}

// wide blocks (>= kCMinSteps sweeps per visit) only while they still give every workgroup twice
// its share of blocks, and never for a dependent chain
static bool c_wide(const VptqLayerDesc* descs, int n, int cus, bool dependent) {
  return !dependent && c_blocks(descs, n, cus, true) >= 2ll * cus;
}
// does one launch of these layers fill the device (>= 3/4 of the workgroups busy)?
bool gemv_k256c_fills_device(const VptqLayerDesc* descs, int n, bool dependent) {
  const int cus = c_workgroups(dependent);
  return 4 * c_blocks(descs, n, cus, c_wide(descs, n, cus, dependent)) >= 3ll * cus;
}

template <typename DT, bool DEP, int MODE = kCModeFolded>
static hipError_t launch_c(const K256CParams& P, int grid, hipStream_t st) {
  auto kern = gemv_k256c_kernel<DT, DEP, MODE>;
  constexpr uint32_t lds = c_lds_bytes(MODE);
  static std::atomic<bool> attr_set[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  if (DEP) {
    // A dependent chain's workgroups wait for each other INSIDE the launch (arrival flags per layer): every one of them
    // must be resident at once, or the waiting ones starve the rest.  The grid is one workgroup per CU; refuse it unless
    // the runtime confirms that this kernel (1024 threads, this much LDS) fits a CU and the grid does not exceed the
    // CUs it reports.  (A device shared with another CU-pinning kernel, a CU mask or a partition mode the runtime does
    // not reflect here can still break the assumption: the library's advice for dependent layers is one launch per
    // layer)
    static std::atomic<int> resident[64];
    if (!resident[dev]) {
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, kCThreads, lds) != hipSuccess) per_cu = 0;
      resident[dev] = per_cu > 0 ? per_cu * c_device_cus() : -1;
    }
    if (resident[dev] < grid) return hipErrorCooperativeLaunchTooLarge;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kCThreads), lds, st, P);
  return hipGetLastError();
}

// the reference's roundings inside the chain launch: fp16, independent layers
bool gemv_k256c_exact_ok(const VptqLayerDesc& d, bool dependent) {
  (void)d;   // (bf16 since the end of round 6: its roundings on the matrix pipe)
  return !dependent && VPTQ_K256C_PROF < 2;
}
// ... and where an activation column dominates only (VPTQ_GEMV_SELECTIVE): the same, + workspace (gemv_k256c_selective_bytes)
bool gemv_k256c_selective_ok(const VptqLayerDesc& d, bool dependent) {
  return !dependent && VPTQ_K256C_PROF == 0 && d.group_size <= kHotMaxBlocks * kCBlockCols;   // (fp16 and bf16: the corrections are VALU arithmetic)
}

// ---- layers with an input permutation: x is gathered ONCE into the caller's workspace (xp[c] = x[perm[c]]) by a small
// launch in front of the chain launch, which then runs on (xp, scale_permuted, bias_permuted) without a permutation: the
// chain kernel's LDS is full (two 64 KiB codebook images), and a gather inside its load queue would put a dependent load -
// perm, then x - in front of every sweep.  (The reference applies `perm` inline, csrc/kernels/quant_gemv.cuh:53-54.)
struct PermXParams {
  int n;
  int start[kMaxGroup + 1];          // first workgroup of layer l
  const uint16_t* x[kMaxGroup];
  const uint16_t* perm[kMaxGroup];
  uint16_t* out[kMaxGroup];
  int I[kMaxGroup];
};
__global__ __launch_bounds__(256) void permute_x_kernel(const PermXParams P) {
  int l = 0;
  for (int i = 1; i < P.n; ++i)
    if ((int)blockIdx.x >= P.start[i]) l = i;
  l = __builtin_amdgcn_readfirstlane(l);
  // (layer fields through the scalar cache: constant-address-space loads with a uniform index)
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const char __attribute__((address_space(4)))* ka_t;
  const ka_t base = (ka_t)__builtin_amdgcn_kernarg_segment_ptr();
  const uint16_t* const x = *(const uint16_t* const __attribute__((address_space(4)))*)(base + offsetof(PermXParams, x) + (size_t)l * 8);
  const uint16_t* const perm = *(const uint16_t* const __attribute__((address_space(4)))*)(base + offsetof(PermXParams, perm) + (size_t)l * 8);
  uint16_t* const out = *(uint16_t* const __attribute__((address_space(4)))*)(base + offsetof(PermXParams, out) + (size_t)l * 8);
  const int I = *(const int __attribute__((address_space(4)))*)(base + offsetof(PermXParams, I) + (size_t)l * 4);
  const int b0 = *(const int __attribute__((address_space(4)))*)(base + offsetof(PermXParams, start) + (size_t)l * 4);
#else
  const uint16_t* const x = P.x[0]; const uint16_t* const perm = P.perm[0]; uint16_t* const out = P.out[0];
  const int I = P.I[0], b0 = P.start[0];
#endif
  const int c0 = (((int)blockIdx.x - b0) * 256 + (int)threadIdx.x) * 8;
  if (c0 >= I) return;   // (I is a multiple of 8: whole groups of 8 columns)
  const u32x4 pv = *(const u32x4*)(as_global(perm) + c0);
  u32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = (uint32_t)as_global(x)[pv[q] & 0xffffu] | ((uint32_t)as_global(x)[pv[q] >> 16] << 16);
  *(u32x4*)(as_global(out) + c0) = r;
}
size_t gemv_k256c_perm_bytes(const VptqLayerDesc& d) { return d.perm ? ((size_t)d.in_features * 2 + 255) / 256 * 256 : 0; }
hipError_t launch_permute_x(const VptqLayerDesc* descs, int n, const void* const* x, void* const* out, hipStream_t st) {
  PermXParams P = {};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].perm) continue;
    if (P.n == kMaxGroup) return hipErrorInvalidValue;
    P.start[P.n] = blocks;
    P.x[P.n] = (const uint16_t*)x[i];
    P.perm[P.n] = (const uint16_t*)descs[i].perm;
    P.out[P.n] = (uint16_t*)out[i];
    P.I[P.n] = descs[i].in_features;
    blocks += (descs[i].in_features / 8 + 255) / 256;
    ++P.n;
  }
  if (P.n == 0) return hipSuccess;
  for (int i = P.n; i <= kMaxGroup; ++i) P.start[i] = blocks;
  hipLaunchKernelGGL(permute_x_kernel, dim3(blocks), dim3(256), 0, st, P);
  return hipGetLastError();
}

// n <= kMaxGroup layers, all gemv_k256c_eligible and of one dtype; sync = kCFlagStride flags per
// layer (zeroed by the caller's memset node) when dependent
hipError_t launch_gemv_k256c(const VptqLayerDesc* descs, int n, const void* const* x, void* const* y,
                             int flags, bool dependent, uint32_t* sync, hipStream_t st) {
  if (n < 1 || n > kMaxGroup) return hipErrorInvalidValue;
  const int cus = c_workgroups(dependent);
  const bool wide = c_wide(descs, n, cus, dependent);
  const long long total = c_blocks(descs, n, cus, wide);
  const int grid = (int)(total < cus ? total : cus);
  K256CParams P;
  P.n_layers = n;
  P.tokens = 1 | ((flags & VPTQ_GEMV_OUT_F32) ? kOutF32Bit : 0);
  P.sync = sync;
  long long first = 0;   // workgroup that owns block 0 of the layer
  for (int i = 0; i < kMaxGroup + kCHeadPad; ++i) P.head[i][0] = P.head[i][1] = 0u;
  for (int i = 0; i < n; ++i) {
    const VptqLayerDesc& d = descs[i];
    CLayerArgs& Ly = P.layer[i];
    Ly.idx = (const uint32_t*)d.indices;
    Ly.cent = (const uint32_t*)d.centroids;
    Ly.rcent = (const uint32_t*)d.res_centroids;
    Ly.x = (const uint16_t*)x[i];
    Ly.y = (uint16_t*)y[i];
    Ly.scale = (const uint16_t*)d.weight_scale;
    Ly.wbias = (const uint16_t*)d.weight_bias;
    Ly.bias = (const uint16_t*)d.bias;
    Ly.N = d.num_indices;
    Ly.G = d.group_size;
    Ly.O = d.out_features;
    Ly.row_words = d.row_words;
    const int rpw = c_rows_per_wg(d, cus, wide);
    const int ng = c_groups(d), ns = (d.group_size + kCSweepCols - 1) / kCSweepCols;
    if (rpw > 0xffff || ng > 0xffffff || ns > 0xff || grid > 0xffff) return hipErrorInvalidValue;
    Ly.wgs = (int)(first % grid);
    Ly.rpw = rpw;
    P.head[i][0] = (uint32_t)Ly.wgs | ((uint32_t)rpw << 16);
    P.head[i][1] = (uint32_t)ng | ((uint32_t)ns << 24);
    // dependent chain: every layer starts at workgroup 0 (all of its row groups wait anyway)
    first = dependent ? 0 : first + (ng + rpw - 1) / rpw;
  }
  const bool f16 = descs[0].dtype == VPTQ_DTYPE_F16;
  if ((flags & VPTQ_GEMV_SELECTIVE) && !(flags & VPTQ_GEMV_EXACT)) {
    // sync = n thresholds, written by the launch in front of the chain launch (same stream)
    if (dependent || !sync) return hipErrorInvalidValue;   // (the caller routes those layer by layer / asks for EXACT)
#if VPTQ_K256C_PROF == 0
    CHotArgs H = {};
    size_t off = ((size_t)n * 16 + 255) / 256 * 256;
    int max_rows = 1;
    for (int i = 0; i < n; ++i) {
      if (descs[i].group_size > kHotMaxBlocks * kCBlockCols || off > 0xffffffffull) return hipErrorInvalidValue;
      H.corr_off[i] = (uint32_t)off;
      off += ((size_t)descs[i].num_indices * 8 * 4 + 255) / 256 * 256;
      max_rows = descs[i].num_indices > max_rows ? descs[i].num_indices : max_rows;
    }
    H.kappa = c_kappa();
    if (f16) hipLaunchKernelGGL(k256c_hot_kernel<F16>, dim3((max_rows + kHotRows - 1) / kHotRows, n), dim3(256), 0, st, P, H);
    else hipLaunchKernelGGL(k256c_hot_kernel<BF16>, dim3((max_rows + kHotRows - 1) / kHotRows, n), dim3(256), 0, st, P, H);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return f16 ? launch_c<F16, false, kCModeSel>(P, grid, st) : launch_c<BF16, false, kCModeSel>(P, grid, st);
#else
    return hipErrorInvalidValue;
#endif
  }
  if (flags & VPTQ_GEMV_EXACT) {
    if (dependent) return hipErrorInvalidValue;   // (the caller routes those layer by layer)
#if VPTQ_K256C_PROF < 2
    return f16 ? launch_c<F16, false, kCModeExact>(P, grid, st) : launch_c<BF16, false, kCModeExact>(P, grid, st);
#else
    return hipErrorInvalidValue;
#endif
  }
  if (dependent) return f16 ? launch_c<F16, true>(P, grid, st) : launch_c<BF16, true>(P, grid, st);
  return f16 ? launch_c<F16, false>(P, grid, st) : launch_c<BF16, false>(P, grid, st);
}

}  // namespace vptq
