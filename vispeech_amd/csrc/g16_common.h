// Shared device helpers of the split-f16 vocoder kernels (gen16.hip, gen16_rw.hip): vector typedefs, the diagnostic
// build switches, the MFMA / barrier / counted-wait wrappers and THE operand split (g16_split4) -- every generator
// kernel splits with this one function, which is what keeps the ResBlock implementations bit-identical.
#pragma once
#include "kernels.h"

#include <type_traits>
#include <utility>

namespace vsp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// G16_DIAG: timing-only ablation builds (tools/ablate.sh; results WRONG by construction): bit 0 no MFMA, bit 1 no
// weight DMA, bit 2 no barriers, bit 3 no epilogue memory traffic, bit 4 no window loads, bit 5 every weight slice
// copied from the SAME source bytes (slice 0: L2-hot), bit 6 no vmcnt waits.  0 in the product build.
#ifndef G16_DIAG
#define G16_DIAG 0
#endif

// G16_STAMPS (diagnostic build, tools/stamps_g16.py): waves 0 and NWV/2 of every 97th block of the VSP_STAMP_G16-th
// g16_conv launch record wall-clock stamps (s_memrealtime, 100 MHz) at their phase boundaries.
#ifdef G16_STAMPS
constexpr int G16_NSTAMP = 256, G16_NSAMPLE = 64;
static __device__ unsigned long long g_g16_stamps[G16_NSAMPLE][G16_NSTAMP];
static __device__ unsigned g_g16_stamp_count;
#define G16_STAMP()                                                                     \
  do {                                                                                  \
    if (stamp_slot >= 0 && stamp_n < G16_NSTAMP - 1 && lane == 0)                       \
      g_g16_stamps[stamp_slot][stamp_n] = __builtin_amdgcn_s_memrealtime();             \
    ++stamp_n;                                                                          \
  } while (0)
#define G16_STAMPT(tag)                                                                 \
  do {                                                                                  \
    if (stamp_slot >= 0 && stamp_n < G16_NSTAMP - 1 && lane == 0)                       \
      g_g16_stamps[stamp_slot][stamp_n] = (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull) | ((unsigned long long)(tag) << 56); \
    ++stamp_n;                                                                          \
  } while (0)
#else
#define G16_STAMP() ((void)0)
#define G16_STAMPT(tag) ((void)0)
#endif

constexpr int G16_HALO = 64;   // max (K-1)*dil
constexpr int G16_IMG_PADF = CL_IMG_PADF;   // operand images: zero rows in front of time 0 (>= the largest padding)
constexpr int G16_OOR = 0x7ffffff0;   // byte offset outside every buffer descriptor: loads give 0, stores are dropped

// one asm statement: the "memory" clobber keeps the compiler from moving LDS traffic across the barrier; LDS-DMA
// and global loads stay in flight (no vmcnt here)
#if G16_DIAG & 4
#define G16_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define G16_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// f32x4 MFMA wrapper (ablation bit 0 keeps the operands alive without the matrix instruction)
#if G16_DIAG & 1
#define G16_MFMA(a, b, c) ([&]() { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define G16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif

// s_waitcnt vmcnt(N): "at most N vector-memory operations outstanding".  They retire in issue order, so N = the
// number of operations issued AFTER the one that must have landed.  The count is an immediate: the callers pick
// among the few values that occur (pieces of one slice, plus 0 / 1 / 2 window-load groups of NL loads).
template <int N>
__device__ __forceinline__ void g16_vmcnt() {
  if constexpr ((G16_DIAG & 64) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int P, int NL>
__device__ __forceinline__ void g16_vm_wait(bool slice_behind, int groups) {
  if (groups == 0) { if (slice_behind) g16_vmcnt<P>(); else g16_vmcnt<0>(); }
  else if (groups == 1) { if (slice_behind) g16_vmcnt<P + NL>(); else g16_vmcnt<NL>(); }
  else { if (slice_behind) g16_vmcnt<P + 2 * NL>(); else g16_vmcnt<2 * NL>(); }
}

template <int N>
__device__ __forceinline__ void g16_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// ds_read_b128 the compiler's wait-count pass does not see: the main loop places its own counted lgkmcnt waits
// (hipcc waits lgkmcnt(0) at the first use of a fragment requested in the previous loop iteration, which stalls
// every step on the reads just issued).  LDS operations return in order, so "at most N outstanding" = everything
// but the N youngest has arrived.  Every wait is followed by a sched_barrier: hipcc would otherwise hoist a
// register-only MFMA across the asm wait.
template <int OFF>
__device__ __forceinline__ f16x8 g16_lds_read(unsigned addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <class F, int... I>
__device__ __forceinline__ void g16_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void g16_for(F&& f) {
  g16_for_impl(f, std::make_integer_sequence<int, N>{});
}

// leaky-relu + split of four fp32 values -> hi / lo f16x4.
// The split itself is kernels.h vsp_split_pair (round 6: FOUR vector instructions per two values -- one v_cvt_pk_f16_f32 for
// both hi parts, one v_fma_mix_f32 per lo part, one v_cvt_pkrtz for both lo parts; rounds 3-5: six, round 2: sixteen),
// + 2 per value for the leaky-relu: 16 per four values where round 5 had 20.
//   hi = x ROUNDED to f16 (inf beyond the f16 range: loud, vsp_status -- rounds 2-5 saturated at 65504 silently);
//   lo = x - hi, exact in fp32, packed by truncation: |lo| <= 2^-11 |x| -- a normal f16 number while |x| >= 2^-3, an f16
//        SUBNORMAL below (a multiple of 2^-24).
// So hi + lo keeps 22 bits of x where |x| >= 2^-3 and x to within 2^-24 ABSOLUTE below.  In a dot product that is
// |error| <= 2^-24 sum|w| per output: below the fp32 accumulation noise of a 96 .. 2816-term sum against outputs of
// O(1e-2 .. 1); where the output LEVEL makes it visible is measured in tests/test_amplitude_floor.py, and the generator
// carries its activations * 2^4 (model.h act_scale) to push that level down to about -92 dBFS.  tests/test_cl_ops.py holds
// the absolute bound with every activation at 1e-5 and a relative 1e-6 with weights of 1e-3.  Every generator kernel
// splits with this one function, so the ResBlock implementations stay bit-identical.
// G16_SPLIT_PLAIN (late round 4): the subtraction and the leaky-relu scaling as PLAIN f32 instructions (asm, so that hipcc does not
// SLP-pack them): beside an MFMA stream a v_pk_*_f32 costs far more issue time than the two plain instructions it
// replaces (MI355X_MICROARCH.md, constants table: "an anti-lever beside MFMAs"), and these kernels' vector work runs
// beside MFMAs.  Same arithmetic, same bits.  CAUTION: hipcc's hazard recognizer does not see into inline asm -- an asm
// instruction that reads an MFMA RESULT register directly gets no wait states and reads a stale value (tried in round 4
// on the fold of the then two accumulators as asm v_fma_f32: errors of 1e-4).  Every caller hands g16_split4 values that a compiler-generated vector
// instruction has produced (the fold, a select).
#ifndef G16_SPLIT_PLAIN
#define G16_SPLIT_PLAIN 1
#endif
// ONE accumulator per tile (round 5).  Rounds 2-4 carried the lo parts * 2^11 (normal f16 numbers at any magnitude) and gave
// the two cross products an accumulator of their own (cr), folded at the end as hh + cr / 2048.  The matrix core keeps f16
// DENORMAL inputs (tools/micro/mfma_f16_denorm.hip: exact products of 2^-24), so the lo parts can stay UNSCALED -- lo = x - hi,
// exact to 2^-25 absolute where it falls under 2^-14 -- and HH, CROSS, CROSS accumulate into one fp32 register set: half the
// accumulator registers, no fold, one vector instruction less per split value.  The WEIGHTS are packed * G16_WSCALE (a power
// of two: exact) so that their lo parts stay normal numbers for any trained magnitude (|w| of 1e-2 has lo parts of 5e-6);
// the bias rides in the accumulator * G16_WSCALE as well (weights.cpp, upload_cl_conv) and every result leaves the
// accumulator through * G16_UNSCALE -- exact, and x * G16_UNSCALE + residual contracts to one fma with the same bits.
// Measured: step 71.4 -> 68.9 ms same box with every parity test green (profiles/r05_one_accumulator.txt).
// (G16_WSCALE / G16_UNSCALE: kernels.h)
__device__ __forceinline__ void g16_split2(f32x2 x, f16x2& h, f16x2& l) {
  unsigned hi, lo;
  vsp_split_pair(x.x, x.y, hi, lo);          // (kernels.h: round 6's four-instruction form)
  h = __builtin_bit_cast(f16x2, hi);
  l = __builtin_bit_cast(f16x2, lo);
}
__device__ __forceinline__ void g16_split4(const f32x4 v, float slope, bool act, f16x4& eh, f16x4& el) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    f32x2 x = {v[2 * k], v[2 * k + 1]};
    if (act) {
#if G16_SPLIT_PLAIN
      f32x2 y;
      asm("v_mul_f32 %0, %1, %2" : "=v"(y.x) : "s"(slope), "v"(x.x));
      asm("v_mul_f32 %0, %1, %2" : "=v"(y.y) : "s"(slope), "v"(x.y));
#else
      const f32x2 y = x * slope;
#endif
      asm("v_max_f32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(y.x));   // leaky-relu = max(x, slope*x), 0 <= slope <= 1
      asm("v_max_f32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(y.y));
    }
    f16x2 xh, xl;
    g16_split2(x, xh, xl);
    eh[2 * k] = xh.x; eh[2 * k + 1] = xh.y;
    el[2 * k] = xl.x; el[2 * k + 1] = xl.y;
  }
}
// v / div the way the reference divides (IEEE division: x / 3 is not x * (1 / 3)), behind a REAL branch.  hipcc
// if-converts `if (div != 1.f) v /= div` into an unconditional division plus a select -- 12 vector instructions and a
// quarter-rate v_rcp per value, 64 values per lane of the 128-row tile: ~800 instructions in the epilogue of EVERY tile,
// although only the last convolution of a stage divides.  The asm statement in the block cannot be speculated, and the
// operand passes through it, so the division stays behind the (uniform, scalar) branch.
__device__ __forceinline__ void g16_div(f32x4& v, float div) {
  if (div != 1.f) {
    asm volatile("" : "+v"(v));
    v /= div;
  }
}
// Ragged batches (round 5): the time extent of utterance b in a launch whose uniform extent is T (kernels.h, ClConvArgs)
__device__ __forceinline__ int g16_len(const int* glen, int b, int grate, int T) {
  return glen ? __builtin_amdgcn_readfirstlane(glen[b]) * grate : T;
}
// A persistent block's run of tiles in the (utterance, tile) sequence of a ragged batch: utterance b has
// ceil(len_b / R) tiles.  Returns the run's first tile as (b0, tile0 within b0) and its length n.  Prologue code: plain
// scalar loads and divisions, once per block.
__device__ __forceinline__ void g16_ragged_run(const int* glen, int B, int grate, int R, int nblocks, int bid, int& b0,
                                               int& tile0, int& n) {
  int total = 0;
  for (int b = 0; b < B; ++b) total += (glen[b] * grate + R - 1) / R;
  const int per = total / nblocks, extra = total - per * nblocks;
  const int lo = bid * per + (bid < extra ? bid : extra);
  n = per + (bid < extra ? 1 : 0);
  int acc = 0, b = 0;
  for (; b < B - 1; ++b) {
    const int nt = (glen[b] * grate + R - 1) / R;
    if (lo < acc + nt) break;
    acc += nt;
  }
  // (loads through a plain global pointer are vector loads: the results are uniform but the compiler does not know it)
  b0 = __builtin_amdgcn_readfirstlane(b);
  tile0 = __builtin_amdgcn_readfirstlane(lo - acc);
  n = __builtin_amdgcn_readfirstlane(n);
}

__device__ __forceinline__ f32x4 g16_as_f32x4(const u32x4 v) {
  return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ u32x4 g16_as_u32x4(const f32x4 v) {
  return u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
}

}  // namespace vsp
