// Vocoder convolutions on the f16 matrix core with SPLIT operands -- fp32-accurate at ~4.6x the native f32 MFMA rate
// (measured on MI355X, profiles/r01_exp_split_f16.log: error of the 3-term split 6e-8 * sum|ab|, no worse than an f32
// fmaf chain):
//   x = xh + xl,  xh = x truncated to f16 precision,  xl = f16(x - xh)   (exact residual; an f16 subnormal where |x| < 2^-4:
//   w = wh + wl   (packed * 2^8: wl stays a normal number)                 the matrix core keeps those, g16_common.h)
//   x*w ~= xh*wh + xh*wl + xl*wh                  (dropped xl*wl term <= 2^-22 |x w|)
//   -> ONE f32 accumulator per output tile (round 5; rounds 2-4 carried the lo parts * 2^11 and gave the cross products an
//      accumulator of their own): three MFMAs per 32-deep step into the same registers, result = accumulator * 2^-8.
//
// Why a second generation of these kernels (round 2):
//   * MFMA shape.  Under the chip's power cap the 16x16x32 form sustains 1.15x the FLOP/s of 32x32x16 on
//     pseudo-random operands at identical LDS traffic (tools/micro/mfma_shape.hip, MI355X: 1976 vs 1717 TFLOP/s
//     free-running, 1844 vs 1626 with a barrier per 64-deep step).
//   * Orientation.  GEMM M = output channel (A = weights), N = time (B = activations), so a lane of a 16x16 D tile
//     holds FOUR CONSECUTIVE CHANNELS of one time row: the epilogue is one 16-byte load / store per tile instead
//     of sixteen 4-byte ones, and the conv1 -> conv2 hand-off of the fused pair is one ds_write_b64 per tile and
//     image instead of thirty-two ds_write_b16.
//   * Pipeline.  The weight ring has three slots filled by LDS-DMA two slices ahead and is synchronised with raw
//     s_barrier + counted vmcnt (nothing drains the DMA queue).  In the single-conv kernel the one barrier per step
//     sits BEFORE THE LAST SUB-STEP of the step, so the A fragments of the next slice are requested while the last
//     MFMAs of the current one issue and no step starts with an LDS round trip (the round-1 kernels began every
//     step with 16 ds_read_b128 + lgkmcnt(0) in all eight waves at once); its activation window is double-buffered
//     per 32-channel chunk (128-row tile): chunk c+1 is converted and written while chunk c multiplies.
//
// Layouts
//   activations  [B][T][C] fp32 channels-last in HBM (unchanged);
//   window (LDS) per 32-channel chunk, per image (hi, lo): 4 planes kq = (c % 32) / 8 of [row][8 halfs] (16 B per
//                row): a B fragment (lane = time l & 15, k = 8 (l >> 4) + j) is one conflict-free ds_read_b128 for
//                ANY row offset (16 consecutive 16-byte units per lane group);
//   weights      packed on the host in A-fragment order [chunk32][tap][m-tile][hi|lo][lane][8 halfs]
//                (lane = row l & 15, k = 8 (l >> 4) + j): a (chunk, tap) slice of a block's m-tiles is one
//                contiguous run, copied by global_load_lds_dwordx4 in 1 KiB pieces.
//   transposed conv: the `phases` = stride polyphase GEMMs are stacked along M (row = phase * Cout + co): one
//                block computes its rows of all phases from ONE window; D row (ph, co) of input time q is stored
//                at output row phases * q + ph - ups_p.
//
// Reference call sites: modules.py:210-223 (ResBlock1), models.py:255-257, 276-285 (ups, resblock averaging).
#include "kernels.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "g16_common.h"

namespace vsp {

size_t packed_g16_halfs(int rows, int Cin, int K) { return (size_t)K * Cin * rows * 2; }

// dense: W[row][ci][tap] fp32 (row = phase * Cout + co for a transposed conv) -> interleaved hi / lo fragment blocks
void pack_g16_weights(uint16_t* dst, int rows, int Cin, int K, const float* dense) {
  const int nmt = rows / 16;
  for (int r = 0; r < rows; ++r)
    for (int ci = 0; ci < Cin; ++ci)
      for (int tap = 0; tap < K; ++tap) {
        const float w = dense[((size_t)r * Cin + ci) * K + tap] * G16_WSCALE;   // (g16_common.h: exact, keeps lo normal)
        const _Float16 h = (_Float16)w;
        _Float16 l = (_Float16)(w - (float)h);
#ifdef VSP_EXPERIMENTS
        {   // power experiment: keep only the top VSP_WLO_BITS mantissa bits of the weights' lo parts (profiles/r06_power_and_clock.txt)
          static int keep = -1;
          if (keep < 0) { const char* e = getenv("VSP_WLO_BITS"); keep = e ? atoi(e) : 10; }
          if (keep < 10) { uint16_t u; std::memcpy(&u, &l, 2); u &= (uint16_t)~((1u << (10 - keep)) - 1u); std::memcpy(&l, &u, 2); }
        }
#endif
        const int chunk = ci / 32, kk = ci % 32, mt = r / 16, lane = (r % 16) + 16 * (kk / 8), j = kk % 8;
        const size_t blk = (((size_t)chunk * K + tap) * nmt + mt) * 2;
        std::memcpy(dst + (blk * 64 + lane) * 8 + j, &h, 2);
        std::memcpy(dst + ((blk + 1) * 64 + lane) * 8 + j, &l, 2);
      }
}


// (out == NULL -- an image-only result -- gets an empty descriptor: loads give 0, stores are dropped)
__device__ __forceinline__ float* g16_out_base(const ClConvArgs& a, int b) {
  return a.out ? a.out + (size_t)b * a.o_bs : nullptr;
}

// ------------------------------------------------------------------------------------------------------------
// Single convolution (and polyphase transposed convolution).  Block = WM x WN waves; a wave owns MW m-tiles
// (16 output rows each) x NW n-tiles (16 time columns each).  Contraction runs chunk-major: for each 32-channel
// chunk, for each tap: one "step" = MW*NW*3 MFMAs per wave, NW sub-steps of MW*3.
// The window is double-buffered per chunk: chunk c + 1 is converted and written during the MEM phase of chunk c's
// last step (no exposed chunk transition).
// XIN (round 4): the input is a producer-written OPERAND IMAGE (ClConvArgs::x_img: leaky-relu applied, hi / lo split, in
// the window's own plane layout) instead of the fp32 tensor: a chunk's window goes HBM -> LDS by LDS-DMA -- every wave
// copies one (image, plane) of it in NL pieces of 64 rows -- with no conversion arithmetic, no staging registers and,
// above all, without the vmcnt(0) hipcc puts in front of the staged registers' first use, which drained the weight ring
// once per chunk.  The same arithmetic on the same bits: results are bit-identical to the fp32-input form.
#ifndef G16_EPI_LDS
#define G16_EPI_LDS 1
#endif
#ifndef G16_IMG_EPI
#define G16_IMG_EPI 1
#endif
#ifndef G16_UPS_EPI_LDS
#define G16_UPS_EPI_LDS 1
#endif
template <int MW, int NW, int WM, int WN, int TERMS, bool XIN = false, bool IMG_EPI = false>
__global__ void __launch_bounds__(64 * WM * WN) g16_conv(ClConvArgs a) {
  constexpr int NWV = WM * WN, NTH = 64 * NWV;
  constexpr int BT = 16 * NW * WN;              // time columns per block
  constexpr int MTB = MW * WM;                  // m-tiles per block
  constexpr int WR = BT + G16_HALO;             // window rows allocated
  constexpr int PL = WR * 16;                   // bytes per plane
  constexpr int XIMG = 4 * PL;                  // bytes per image
  constexpr int XBUF = 2 * XIMG;                // hi + lo
  constexpr int SLOT = MTB * 2048;              // bytes per ring slot = one (chunk, tap) slice of the block's rows
  constexpr int NS = 4;
  constexpr int RPS = NTH / 8;                  // rows per staging sweep
  constexpr int NL = (WR + RPS - 1) / RPS;
  constexpr int NBLK = 2 * MTB;                 // 1 KiB pieces per slice
  constexpr int NBW = (NBLK + NWV - 1) / NWV;   // per wave
  static_assert(PL % 256 == 0, "plane size keeps the fragment reads conflict-free");
  static_assert(NW % 2 == 0, "B double buffer parity");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xw = lds;
  char* const Rg = lds + 2 * XBUF;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int nmt = (a.phases * a.Cout) >> 4, nch = a.Cin >> 5;
  // XCD-aware block numbering (workgroup ids go round-robin over the 8 XCDs): XCD k gets the k-th contiguous eighth of the
  // (utterance, time tile, row group) sequence, row group fastest
  const int gy = gridDim.y, gx = gridDim.x;
  const int nwg = gx * gy * gridDim.z, orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
  const int id = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int cb = id % gy, bx = (id / gy) % gx, b = id / (gy * gx);
  const int t0 = bx * BT;
  // ragged batch: this utterance's own extents (kernels.h); a tile behind its end has nothing to do
  int T_in = a.T_in, Nq = a.Nq, T_store = a.T_store;
  if (a.glen) {
    const int gl = __builtin_amdgcn_readfirstlane(a.glen[b]);
    T_in = gl * a.g_in; Nq = T_in + (a.Nq - a.T_in); T_store = gl * a.g_store;
    if (t0 >= Nq) return;
  }
  const int K = a.K, S = nch * K;
  const int xrows = BT + (K - 1) * a.dil;        // window rows actually needed
#ifdef G16_STAMPS
  int stamp_slot = -1, stamp_n = 0;
  if ((wave == 0 || wave == NWV / 2) && orig % 97 == 5 && (a.terms & 0x100)) {
    unsigned sl_ = 0;
    if (lane == 0) sl_ = atomicAdd(&g_g16_stamp_count, 1u);
    sl_ = __builtin_amdgcn_readfirstlane(sl_);
    stamp_slot = sl_ < (unsigned)G16_NSAMPLE ? (int)sl_ : -1;
    if (stamp_slot >= 0 && lane == 0) g_g16_stamps[stamp_slot][G16_NSTAMP - 1] = wave + 1;   // which half
  }
  G16_STAMP();                                   // 0: start
#endif

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, T_in * a.x_ts * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      g16_out_base(a, b), 0, a.out ? T_store * a.o_ts * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(
      a.res ? const_cast<float*>(a.res) + (size_t)b * a.r_bs : g16_out_base(a, b), 0,
      a.res ? T_store * a.r_ts * 4 : (a.out ? T_store * a.o_ts * 4 : 0), 0x00020000);

  // ---- window staging: 16 consecutive lanes write 128 contiguous bytes of one plane (conflict-free)
  const int g16 = tid >> 4, kq_s = g16 & 3, row_s = (g16 >> 2) * 8 + ((tid >> 1) & 7), half_s = tid & 1;
  const int st_voff = (row_s * a.x_ts + (2 * kq_s + half_s) * 4) * 4;
  const int st_loff = kq_s * PL + row_s * 16 + half_s * 8;
  u32x4 sv[NL];
  const float slope = a.in_act ? a.in_slope : 1.f;   // leaky-relu as max(x, slope * x): slope 1 = identity
  constexpr bool act = true;
  auto x_issue = [&](int chunk) {
    const int base = ((t0 - a.pad) * a.x_ts + chunk * 32) * 4;            // uniform, may be negative
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const bool in = (u + 1) * RPS <= BT || row_s + u * RPS < xrows;
      sv[u] = (G16_DIAG & 16) ? u32x4{1u, 2u, 3u, 4u}
                              : __builtin_amdgcn_raw_buffer_load_b128(rx, in ? st_voff + (base + u * RPS * a.x_ts * 4) : G16_OOR, 0, 0);
    }
  };
  auto x_write = [&](int buf) {
    char* dst0 = Xw + buf * XBUF + st_loff;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      f16x4 eh, el;
      g16_split4(g16_as_f32x4(sv[u]), slope, act, eh, el);
      // (rows >= xrows were loaded as zeros and are never read: no predicate, no branch)
      *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16) = eh;
      if constexpr (TERMS == 3) *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16 + XIMG) = el;
    }
  };
  // ---- XIN: the window chunk by LDS-DMA.  Image layout [chunk][hi | lo][plane][padded time][8 halfs] (zero rows in
  //      front of time 0 and behind time T - 1: the convolution's zero padding and the last tile's overshoot); wave w
  //      copies (image w / 4, plane w % 4): NL pieces of 64 rows, each 1 KiB contiguous on both sides.
  [[maybe_unused]] auto xi_issue = [&](int chunk) {
    static_assert(!XIN || (NWV == 8 && WR == NL * 64 && TERMS == 3), "one (image, plane) per wave, whole 64-row pieces");
    const int img = wave >> 2, plane = wave & 3;
    const size_t row0 = (((size_t)chunk * 2 + img) * 4 + plane) * a.xi_tpad + (size_t)(G16_IMG_PADF + t0 - a.pad);
    const uint4* gp0 = reinterpret_cast<const uint4*>(a.x_img + (size_t)b * a.xi_bs) + row0 + lane;
    char* lp0 = Xw + (chunk & 1) * XBUF + img * XIMG + plane * PL;
#pragma unroll
    for (int u = 0; u < NL; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp0 + u * 64),
                                       (__attribute__((address_space(3))) void*)(lp0 + u * 1024), 16, 0, 0);
  };
  // ---- weight slices by LDS-DMA: slice (chunk, tap) = NBLK pieces of 1 KiB, contiguous in the packed image.
  //      Slices are requested in step order (chunk-major, tap-minor).
  //      (consecutive slices of a block are nmt * 2 KiB apart: a running pointer -- the scalar work of a MEM phase is on
  //      the ping-pong's critical path)
  const char* dsrc = reinterpret_cast<const char*>(a.wh) + (size_t)cb * MTB * 2048;   // uniform
  const unsigned d_lane = (wave * 64 + lane) * 16;
  const int dstep = (G16_DIAG & 32) ? 0 : nmt * 2048;
  auto dma_next = [&](int slot) {
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
      const int blk = u * NWV + wave;
      if ((NBLK % NWV == 0 || blk < NBLK) && (G16_DIAG & 2) == 0) {
        char* lp = Rg + slot * SLOT + blk * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + u * NWV * 1024 + d_lane),
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
      }
    }
    dsrc += dstep;
  };
  // (a wave that copies no piece of a slice -- NBLK < NWV -- has nothing of its own to wait for there)
  const bool has_pieces = NBLK % NWV == 0 || wave < NBLK;

  // ---- fragments: asm reads with immediate offsets off two per-step base addresses
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned xb_lane = lds0 + (lane >> 4) * PL + (wn * NW * 16 + (lane & 15)) * 16;
  const unsigned wa_lane = lds0 + 2 * XBUF + lane * 16 + wm * MW * 2048;
  f16x8 Ah[MW], Al[MW], Bh[NW], Bl[NW];

  // accumulators start at the bias: a lane holds rows 4 (l >> 4) .. + 3 of its m-tiles
  f32x4 hh[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int row = ((cb * MTB + wm * MW + i) << 4) + 4 * (lane >> 4);
    const int co = row % a.Cout;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
    for (int j = 0; j < NW; ++j) hh[i][j] = bv;
  }

  // ---- prologue: window chunk 0, slices 0 .. 2
  int xl_a = 0, xl_b = 0;           // window loads issued in the previous / in this MEM phase (still counted by vmcnt)
  if constexpr (XIN) {
    xi_issue(0);                    // both window buffers are free: chunks 0 and 1 go out at once
    if (nch > 1) xi_issue(1);
    dma_next(0);
    if (S > 1) dma_next(1);
    if (S > 2) dma_next(2);
    G16_STAMP();
    g16_vmcnt<0>();
  } else {
    x_issue(0);
    dma_next(0);
    if (S > 1) dma_next(1);
    if (S > 2) dma_next(2);
    x_write(0);                       // (the compiler waits for the window loads here)
    G16_STAMP();                      // 1: first window converted and written
    if (nch > 1) { x_issue(1); xl_b = 1; }
    g16_vm_wait<NBW, NL>(false, xl_b);   // slices 0 .. 2 have landed
  }
  G16_STAMP();                      // 2: slices landed
  G16_BARRIER();
  G16_STAMP();                      // 3: prologue barrier
  // PING-PONG.  Waves w and w + NWV/2 share a SIMD.  Each wave alternates a MEM phase (all fragments of one step
  // into registers, its LDS-DMA pieces, window staging) with an MFMA phase (the step's MW*NW*3 MFMAs back to back),
  // one barrier between phases; the second half of the block runs one phase behind the first (it starts with a
  // barrier), so a SIMD's matrix core always has one wave multiplying while its partner does the memory work.
  // Priority (late round 4): STATIC -- the second-dispatched half of the block loses the SIMD's issue arbitration (priority,
  // then age) on every phase; one s_setprio 1 for it here and no per-phase flips (MI355X_MICROARCH.md, "Two waves per
  // SIMD", item 4) instead of a raised priority around every MFMA cluster: s0 -0.17 ms, s1 -0.3 ms, same box.
  if (wave >= NWV / 2) { G16_BARRIER(); __builtin_amdgcn_s_setprio(1); }

  int chunk = 0, tap = 0, slot = 0;   // of the current step
  for (int s = 0; s < S; ++s) {
    const bool last_tap = tap == K - 1;
    const int chunk_n = last_tap ? chunk + 1 : chunk, tap_n = last_tap ? 0 : tap + 1;   // of step s + 1
    const bool wr_step = last_tap && chunk + 1 < nch;      // window chunk + 1 is staged in this step's MEM phase
    // ================= MEM phase of step s =================
    {
      const unsigned b_cur = xb_lane + (chunk & 1) * XBUF + tap * a.dil * 16;
      const unsigned a_cur = wa_lane + slot * SLOT;
      // (order: the B reads, the requests, the A reads -- a wave issues one ds_read_b128 per ~19 clocks
      // (tools/micro/lds_read_bw.hip): the scalar work of the requests fits between them instead of following them)
      g16_for<NW>([&](auto J) {
        constexpr int j = decltype(J)::value;
        Bh[j] = g16_lds_read<j * 256>(b_cur);
        if constexpr (TERMS == 3) Bl[j] = g16_lds_read<j * 256 + XIMG>(b_cur);
      });
      xl_a = xl_b;
      xl_b = 0;
      if constexpr (XIN) {
        // the buffer of chunk - 1 is free from this chunk's first tap on (both wave halves have its last fragments in
        // registers): chunk + 1 is requested into it K steps before its first use.  It is OLDER than the slices the
        // counted waits of the chunk's last taps wait for (K >= 3), so it has landed by then.
        if (tap == 0 && chunk >= 1 && chunk + 1 < nch) { xi_issue(chunk + 1); xl_b = 1; }
      } else if (wr_step) {
        x_write(chunk_n & 1);                               // the other window buffer: last read a chunk ago
        if (chunk + 2 < nch) { x_issue(chunk + 2); xl_b = 1; }
      }
      if (s + 3 < S) dma_next(slot == 0 ? NS - 1 : slot - 1);   // slice s + 3 into the slot slice s - 1 left
      g16_for<MW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        Ah[i] = g16_lds_read<i * 2048>(a_cur);
        if constexpr (TERMS == 3) Al[i] = g16_lds_read<i * 2048 + 1024>(a_cur);
      });
      // my pieces of slice s + 1 have landed: issued after them are slices s + 2, s + 3 and the window loads of this
      // and the previous MEM phase
      if (s + 1 < S) {
        const int behind = (s + 2 < S ? 1 : 0) + (s + 3 < S ? 1 : 0);
        if (!has_pieces || behind == 0) g16_vm_wait<0, NL>(false, xl_a + xl_b);
        else if (behind == 1) g16_vm_wait<NBW, NL>(true, xl_a + xl_b);
        else g16_vm_wait<2 * NBW, NL>(true, xl_a + xl_b);
      }
      G16_STAMP();                                          // 4 + 4 s: MEM work issued, slice wait done
      G16_BARRIER();                                        // (lgkmcnt(0): the fragments are here)
      G16_STAMP();                                          // 5 + 4 s: barrier
    }
    // ================= MFMA phase of step s =================
    __builtin_amdgcn_sched_barrier(0);
    g16_for<NW>([&](auto J) {
      constexpr int j = decltype(J)::value;
      g16_for<MW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        hh[i][j] = G16_MFMA(Ah[i], Bh[j], hh[i][j]);
        if constexpr (TERMS == 3) {
          hh[i][j] = G16_MFMA(Al[i], Bh[j], hh[i][j]);
          hh[i][j] = G16_MFMA(Ah[i], Bl[j], hh[i][j]);
        }
      });
    });
    __builtin_amdgcn_sched_barrier(0);
    G16_STAMP();                                            // 6 + 4 s: MFMAs issued
    // (the second half skips the barrier behind its LAST MFMA phase: nothing in the LDS is read after it, and waiting there
    // would hold its epilogue until the first half's waves have ENDED -- the two epilogues would follow each other)
    if (s + 1 < S || wave < NWV / 2) G16_BARRIER();
    G16_STAMP();                                            // 7 + 4 s: barrier
    chunk = chunk_n;
    tap = tap_n;
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
  // (the first half of the block has executed one barrier fewer: it goes straight to its epilogue, which overlaps
  // the second half's last MFMA phase; a wave that has ended no longer takes part in the barrier)

  // ... and for the image-only results (a ResBlock pair's intermediate): the activated, split tile goes to the LDS in the
  //      image's own plane layout [chunk][hi | lo][plane][column][8 halfs], then a wave copies whole planes -- 64 consecutive
  //      time rows of one plane = 1 KiB contiguous per b128 store, sixteen per wave instead of thirty-two b64 stores of two
  //      256-byte segments each.  The same split of the same values: bit-identical images.
  // (its own instantiation, IMG_EPI -- picked by the launcher for exactly these launches: as a third path inside the one
  // kernel it cost 35-40 spilled registers)
  if constexpr (IMG_EPI) {
    static_assert(G16_EPI_LDS && TERMS == 3 && MTB == 8 && BT == 256 && NWV == 8, "the 128-row x 256-column tile");
    constexpr int EPL = BT * 16;                    // bytes of one plane of the tile in the LDS
    static_assert((size_t)32 * EPL <= (size_t)2 * XBUF + (size_t)NS * SLOT, "the image tile fits the dead window + ring");
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        f32x4 v = hh[i][j] * G16_UNSCALE;
        g16_div(v, a.div);
        f16x4 eh, el;
        g16_split4(v, a.oi_slope, true, eh, el);
        const int mt = wm * MW + i, col = (wn * NW + j) * 16 + (lane & 15), q4e = lane >> 4;
        char* dst = lds + (((mt >> 1) * 2) * 4 + 2 * (mt & 1) + (q4e >> 1)) * EPL + col * 16 + 8 * (q4e & 1);
        *reinterpret_cast<f16x4*>(dst) = eh;
        *reinterpret_cast<f16x4*>(dst + 4 * EPL) = el;
        __builtin_amdgcn_sched_barrier(0);            // (a tile at a time: sixteen splits in flight at once cost spills)
      }
    G16_BARRIER();
    const __amdgpu_buffer_rsrc_t ri2 = __builtin_amdgcn_make_buffer_rsrc(a.o_img + (size_t)b * a.oi_bs, 0, a.Cout * 4 * a.oi_tpad, 0x00020000);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      __builtin_amdgcn_sched_barrier(0);              // (a plane at a time: the copies of all sixteen pieces at once cost spills)
      const int p = wave * 4 + pp;                    // plane of the tile: (chunk of the block, hi | lo, plane)
      const int gplane = (cb * 4 + (p >> 3)) * 8 + (p & 7);
#pragma unroll
      for (int u = 0; u < BT / 64; ++u) {
        const int col = 64 * u + lane, t = t0 + col;
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + p * EPL + col * 16);
        const int off = t < Nq ? (gplane * a.oi_tpad + G16_IMG_PADF + t) * 16 : G16_OOR;
        __builtin_amdgcn_raw_buffer_store_b128(v, ri2, off, 0, 0);
      }
    }
  } else {
  // ---- epilogue through the LDS (round 5) for the plain fp32 destinations (no polyphase scatter, no operand image): in
  //      the D-tile layout a b128 access of a wave is 16 rows x 64 B -- sixteen half-line segments per instruction, and with
  //      one block per CU nothing hides the residual round trip and the store issue (2.8-3.8 us per tile by the stamps of
  //      round 4).  The tile (cross accumulator folded) goes to the dead window / ring as [column][block's channels] fp32
  //      (+ 16 B per column: conflict-free 16-byte writes), then a wave owns whole columns: RB / 4 lanes x 16 B = one
  //      column's channels, contiguous in the channels-last tensor -- residual, previous sum and result alike.  The same
  //      arithmetic in the same order as below: bit-identical.  Nobody reads the LDS any more: the first half has passed
  //      the barrier behind its last MFMA phase, which the second half reached only after its last fragment reads.
  constexpr int RB = 16 * MTB, ECS = RB * 4 + 16, LPC = RB / 4, CPI = 64 / LPC, NIT = BT / CPI / NWV;
  // (the 128-row tile only: the 64- / 32-row tiles of under-filled grids keep the direct form -- there the compiler spills)
  constexpr bool EPI_LDS_FITS = MTB == 8 && (size_t)BT * ECS <= (size_t)2 * XBUF + (size_t)NS * SLOT && BT % (CPI * NWV) == 0;
  if constexpr (G16_EPI_LDS && EPI_LDS_FITS) if (a.phases == 1 && a.out && !a.o_img) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        // (the accumulator as it is: the unscaling is folded into the residual's fma below)
        const int col = (wn * NW + j) * 16 + (lane & 15);
        *reinterpret_cast<f32x4*>(lds + col * ECS + ((wm * MW + i) * 16 + 4 * (lane >> 4)) * 4) = hh[i][j];
      }
    G16_BARRIER();
    const int lc = lane % LPC, cw = lane / LPC;
    // (in batches of at most eight accesses per lane: the operands of a batch are in flight together)
    constexpr int NBT = NIT > 8 ? 8 : NIT;
    static_assert(NIT % NBT == 0, "whole batches");
#pragma unroll
    for (int u0 = 0; u0 < NIT; u0 += NBT) {
      int eo[NBT];
      u32x4 rv[NBT];
      [[maybe_unused]] u32x4 pv[NBT];
#pragma unroll
      for (int u = 0; u < NBT; ++u) {
        const int t = t0 + (wave * NIT + u0 + u) * CPI + cw;
        const bool in_t = t < Nq && (G16_DIAG & 8) == 0;
        eo[u] = in_t ? (t * a.o_ts + cb * RB + 4 * lc) * 4 : G16_OOR;
        if (a.res) rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rr_, in_t ? (t * a.r_ts + cb * RB + 4 * lc) * 4 : G16_OOR, 0, 0);
      }
      if (a.acc_prev) {
#pragma unroll
        for (int u = 0; u < NBT; ++u) pv[u] = __builtin_amdgcn_raw_buffer_load_b128(ro, eo[u], 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NBT; ++u) {
        const int col = (wave * NIT + u0 + u) * CPI + cw;
        f32x4 v = *reinterpret_cast<const f32x4*>(lds + col * ECS + lc * 16);
        if (a.res) v = v * G16_UNSCALE + g16_as_f32x4(rv[u]);     // (exact product: the bits of unscale-then-add)
        else v *= G16_UNSCALE;
        if (a.acc_prev) v += g16_as_f32x4(pv[u]);
        g16_div(v, a.div);
        __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, eo[u], 0, 0);
      }
    }
    return;
  }
  // ---- epilogue: lane = 4 consecutive rows (channels) of one time column: 16-byte accesses.  All residual /
  //      accumulate operands of the wave's tiles are requested first (the fragment registers are dead now).
  G16_STAMP();                                              // epilogue start
  int oo[MW][NW], orr[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int row = ((cb * MTB + wm * MW + i) << 4) + 4 * (lane >> 4);
    const int ph = row / a.Cout, co = row - ph * a.Cout;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int t = t0 + (wn * NW + j) * 16 + (lane & 15);
      const int n = a.phases * t + ph - a.ups_p;            // output row (< 0 or >= T_store: dropped by the descriptor)
      const bool in_t = t < Nq && (G16_DIAG & 8) == 0;
      oo[i][j] = in_t ? n * a.o_ts * 4 + co * 4 : G16_OOR;
      orr[i][j] = in_t ? n * a.r_ts * 4 + co * 4 : G16_OOR;
    }
  }
  u32x4 rv[MW][NW];
  // (acc * G16_UNSCALE is exact: with a residual the unscaling and the addition are ONE fma with the bits of the two steps)
  if (a.res) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) rv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rr_, orr[i][j], 0, 0);
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = hh[i][j] * G16_UNSCALE + g16_as_f32x4(rv[i][j]);
  } else {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] *= G16_UNSCALE;
  }
  if (a.acc_prev) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) rv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ro, oo[i][j], 0, 0);
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] += g16_as_f32x4(rv[i][j]);
  }
  // o_img (round 4): the consumer's operand image of the result -- leaky-relu, hi / lo split, its window's plane layout
  // [chunk][hi | lo][plane][padded time][8 halfs] -- written here, ONCE, instead of being derived from the fp32 tensor
  // by every consumer block (XIN above).  A lane's four channels are half a 16-byte row unit: 8-byte stores, lanes
  // q4 = 0 / 1 (2 / 3) fill the two halves of 16 consecutive rows of one plane.  out == NULL: the image is all that is
  // kept (a ResBlock's intermediate).
  const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
      a.o_img ? a.o_img + (size_t)b * a.oi_bs : reinterpret_cast<uint16_t*>(a.out), 0,
      a.o_img ? a.Cout * 4 * a.oi_tpad : 0, 0x00020000);
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      f32x4 v = hh[i][j];
      g16_div(v, a.div);
      if (a.out) __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, oo[i][j], 0, 0);
      if constexpr (TERMS == 3) {
        if (a.o_img) {
          const int co = ((cb * MTB + wm * MW + i) << 4) + 4 * (lane >> 4);
          const int t = t0 + (wn * NW + j) * 16 + (lane & 15);
          const int unit = ((((co >> 5) * 2) * 4 + ((co & 31) >> 3)) * a.oi_tpad + G16_IMG_PADF + t) * 16 + 2 * (co & 7);
          f16x4 eh, el;
          g16_split4(v, a.oi_slope, true, eh, el);
          const int off = t < Nq ? unit : G16_OOR;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, eh), ri, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, el), ri, off, 4 * a.oi_tpad * 16, 0);
        }
      }
    }
#ifdef G16_STAMPS
  G16_STAMP();                                              // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G16_STAMP();                                              // stores retired
#endif
  }
}

#ifdef G16_STAMPS
extern "C" int vsp_debug_stamps_g16(unsigned long long* host, int max_samples, int reset) {
  unsigned n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_g16_stamp_count), sizeof n);
  if ((int)n > max_samples) n = max_samples;
  if (n > (unsigned)G16_NSAMPLE) n = G16_NSAMPLE;
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_g16_stamps), (size_t)n * G16_NSTAMP * sizeof(unsigned long long));
  if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_g16_stamp_count), &z, sizeof z); }
  return (int)n;
}
#endif

template <int MW, int NW, int WM, int WN, int TERMS, bool XIN = false, bool IMG_EPI = false>
static hipError_t launch_g16_tile(const ClConvArgs& a, int B, hipStream_t s) {
  constexpr int BT = 16 * NW * WN, MTB = MW * WM;
  constexpr size_t lds = (size_t)2 * 2 * 4 * (BT + G16_HALO) * 16 + (size_t)4 * MTB * 2048;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = g16_conv<MW, NW, WM, WN, TERMS, XIN, IMG_EPI>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  const int nmt = a.phases * a.Cout / 16;
  if (nmt % MTB || a.Cin % 32) return hipErrorInvalidValue;
  dim3 grid((a.Nq + BT - 1) / BT, nmt / MTB, B);
#ifdef G16_STAMPS
  {  // stamps only in the VSP_STAMP_G16-th launch of this tile type (0-based)
    static int launch_no = 0;
    static int target = -2;
    if (target == -2) { const char* e = getenv("VSP_STAMP_G16"); target = e ? atoi(e) : -1; }
    ClConvArgs as = a;
    if (launch_no++ == target) as.terms |= 0x100;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, as);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// g16_ups: the kernel-4 transposed convolutions (stride 4: one tap per phase; stride 2: two -- the 64- and 32-channel
// stages' up-convs, reference models.py:262-268, 282-284) as a STREAMING kernel.  Their GEMM is short (K = KT Cin = 128
// per output row) and wide in time: an HBM-bound op that the tiled kernel above ran at 12-15 % matrix issue and
// 1.6-3 TB/s, a block's life being window staging, a few ring steps and an epilogue.  Here
// time is the M axis of v_mfma_f32_16x16x32_f16: a lane's A fragment is 8 consecutive channels of ONE input row --
// 32 contiguous bytes of the channels-last activation, read straight from global memory and split in registers; the
// weights (the packed image above, read as B fragments: same bytes, rows <-> columns) stay in registers for the whole
// run of a wave; no LDS, no barrier.  A wave owns NMT of the rows' 16-row tiles (ROLES waves share a time range when
// the weights of all tiles do not fit one wave's registers) and walks TPW 16-row time tiles.
//   out[phases q + ph - p][co] = bias[co] + sum_tap sum_ci lrelu(x[q - (KT-1) + tap][ci]) w[ph][co][ci][tap]
static bool g16_ups_on() {
#ifdef VSP_EXPERIMENTS
  static int v = -1;
  if (v < 0) { const char* e = getenv("VSP_G16_UPS"); v = e ? atoi(e) != 0 : 1; }
  return v != 0;
#else
  return true;
#endif
}
template <int KT, int NCH, int NMT, int ROLES>
__global__ void __launch_bounds__(64 * (ROLES > 4 ? ROLES : 4)) g16_ups(ClConvArgs a, int tiles_per_wave) {
  constexpr int NWB = ROLES > 4 ? ROLES : 4;            // waves of a block
#if G16_UPS_EPI_LDS
  __shared__ __attribute__((aligned(16))) float ups_tile[NWB * 16 * (NMT * 16 + 4)];
#endif
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave % ROLES, stream = wave / ROLES;
  const int b = blockIdx.y;
  const int nmt_all = a.phases * a.Cout / 16, mt0 = role * NMT;
  int T_in = a.T_in, Nq = a.Nq, T_store = a.T_store;            // (ragged batch: this utterance's own extents, kernels.h)
  if (a.glen) {
    const int gl = __builtin_amdgcn_readfirstlane(a.glen[b]);
    T_in = gl * a.g_in; Nq = T_in + (a.Nq - a.T_in); T_store = gl * a.g_store;
  }
  const int ntiles = (Nq + 15) / 16;
  const int run = blockIdx.x * (NWB / ROLES) + stream;
  const int tile_lo = run * tiles_per_wave;
  const int tile_hi = tile_lo + tiles_per_wave < ntiles ? tile_lo + tiles_per_wave : ntiles;
  if (tile_lo >= ntiles) return;

  // weights: block (chunk, tap, m-tile) of the packed image, hi | lo, this lane's 16 bytes of each
  f16x8 Bh[KT][NCH][NMT], Bl[KT][NCH][NMT];
#pragma unroll
  for (int tap = 0; tap < KT; ++tap)
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const size_t blk = (((size_t)c * KT + tap) * nmt_all + mt0 + m) * 2;
        Bh[tap][c][m] = *reinterpret_cast<const f16x8*>(a.wh + (blk * 64 + lane) * 8);
        Bl[tap][c][m] = *reinterpret_cast<const f16x8*>(a.wh + ((blk + 1) * 64 + lane) * 8);
      }
  float bvu[NMT];
  int co_of[NMT], ph_of[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    const int r = (mt0 + m) * 16 + l15;
    ph_of[m] = r / a.Cout;
    co_of[m] = r - ph_of[m] * a.Cout;
    bvu[m] = a.bias ? a.bias[co_of[m]] * G16_UNSCALE : 0.f;   // (the bias arrives * G16_WSCALE, like the weights)
  }
  const float* xb = a.x + (size_t)b * a.x_bs + kg * 8;
  float* ob = a.out + (size_t)b * a.o_bs;
  const float slope = a.in_slope;
  const bool act = a.in_act != 0;

  // (Measured: requesting the rows of tile + 1 before computing tile costs 32 more registers -- one wave per SIMD instead
  // of two -- and runs 1.5x slower; splitting the m-tiles over more waves to make room repeats the conversion per wave
  // and is slower too.  At two waves per SIMD the kernel moves 4.1 GB in 1.0 ms (32-channel stage) / 3.1 GB in 0.95 ms.)
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int q0 = tile * 16;
    // ---- the tile's input rows (one per lane and tap), 8 channels per chunk: two 16-byte loads, all requested first
    f32x4 raw[KT][NCH][2];
#pragma unroll
    for (int tap = 0; tap < KT; ++tap) {
      const int xr = q0 + l15 - (KT - 1) + tap;
      const bool ok = xr >= 0 && xr < T_in;
      const float* p = xb + (size_t)(ok ? xr : 0) * a.x_ts;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        raw[tap][c][0] = ok ? *reinterpret_cast<const f32x4*>(p + c * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
        raw[tap][c][1] = ok ? *reinterpret_cast<const f32x4*>(p + c * 32 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 hh[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) hh[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < KT; ++tap)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f16x4 h0, l0, h1, l1;
        g16_split4(raw[tap][c][0], slope, act, h0, l0);
        g16_split4(raw[tap][c][1], slope, act, h1, l1);
        const f16x8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const f16x8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          hh[m] = G16_MFMA(ah, Bh[tap][c][m], hh[m]);
          hh[m] = G16_MFMA(ah, Bl[tap][c][m], hh[m]);
          hh[m] = G16_MFMA(al, Bh[tap][c][m], hh[m]);
        }
      }
    // ---- store: lane = output row r (phase, channel), registers = four consecutive times q
#if G16_UPS_EPI_LDS
    // Round 5: through a wave-private LDS tile [q][the wave's rows r] (no barrier: a wave's LDS operations execute in
    // order), read back as 16-byte pieces of four consecutive channels: a wave's store is then NMT * 64 contiguous bytes per
    // input time (at stride 2 the tile's 32 output rows x 32 channels are ONE contiguous 4 KiB) instead of sixteen 4-byte
    // stores of four 64-byte segments each.  Same values.
    {
      constexpr int RSF = NMT * 16 + 4;                       // floats per LDS row (+ 16 B: conflict-free 4-byte writes)
      constexpr int CPR = NMT * 4;                            // 16-byte pieces per row
      constexpr int RPI = 64 / CPR;                           // rows (input times) per b128 instruction
      static_assert(64 % CPR == 0 && 16 % RPI == 0, "whole rows per instruction");
      float* const et = ups_tile + wave * 16 * RSF;
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          et[(4 * kg + jj) * RSF + m * 16 + l15] = hh[m][jj] * G16_UNSCALE + bvu[m];
      const int pc = lane % CPR, ql = lane / CPR;
      const int r = mt0 * 16 + 4 * pc;                         // the piece's first row: four consecutive channels of one phase
      const int ph = r / a.Cout, co = r - ph * a.Cout;
#pragma unroll
      for (int u = 0; u < 16 / RPI; ++u) {
        const int qq = u * RPI + ql, q = q0 + qq;
        const int n = a.phases * q + ph - a.ups_p;
        const f32x4 v = *reinterpret_cast<const f32x4*>(et + qq * RSF + 4 * pc);
        if (q < Nq && n >= 0 && n < T_store) *reinterpret_cast<f32x4*>(ob + (size_t)n * a.o_ts + co) = v;
      }
    }
#else
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int q = q0 + 4 * kg + jj;
        const int n = a.phases * q + ph_of[m] - a.ups_p;
        if (q < Nq && n >= 0 && n < T_store)
          ob[(size_t)n * a.o_ts + co_of[m]] = hh[m][jj] * G16_UNSCALE + bvu[m];
      }
#endif
  }
}

template <int KT, int NCH, int NMT, int ROLES>
static hipError_t launch_g16_ups(const ClConvArgs& a, int B, hipStream_t s) {
  const int ntiles = (a.Nq + 15) / 16;
  const int tpw = 16;                                   // 256 input rows per wave: the weights (32 KiB) are read once per run
  constexpr int NWB = ROLES > 4 ? ROLES : 4;
  const int runs = (ntiles + tpw - 1) / tpw, per_block = NWB / ROLES;
  hipLaunchKernelGGL((g16_ups<KT, NCH, NMT, ROLES>), dim3((runs + per_block - 1) / per_block, B), dim3(64 * NWB), 0, s, a, tpw);
  return hipGetLastError();
}

hipError_t launch_g16_conv(const ClConvArgs& a, int B, hipStream_t s) {
  if ((a.K - 1) * a.dil > G16_HALO || a.K < 1 || a.Nq <= 0 || B <= 0 || a.Cout % 16 || a.Cin % 32 || a.phases < 1 ||
      (a.x_ts & 3) || (a.x_bs & 3) || (reinterpret_cast<uintptr_t>(a.x) & 15) || (a.o_ts & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.out) & 15) || (!a.out && !a.o_img) || (!a.x && !a.x_img))
    return hipErrorInvalidValue;
  if ((a.x_img || a.o_img) && (a.terms != 3 || a.phases != 1)) return hipErrorInvalidValue;
  if (a.o_img && (a.oi_tpad < cl_img_tpad(a.T_store) || (a.oi_bs & 7) || (reinterpret_cast<uintptr_t>(a.o_img) & 15)))
    return hipErrorInvalidValue;
  // an utterance's operand image is addressed with 32-bit byte offsets (descriptor num_records, `unit` in the epilogue);
  // the image-input windows are raw pointer reads of rows [t0 - pad, t0 + BT + halo): only the pad rows cover an overshoot
  if ((a.o_img && (size_t)a.Cout * 4 * (size_t)a.oi_tpad >= (size_t)1 << 31) ||
      (a.x_img && ((size_t)a.Cin * 4 * (size_t)a.xi_tpad >= (size_t)1 << 31 || a.Nq > a.T_in)))
    return hipErrorInvalidValue;
  const int rows = a.phases * a.Cout;
  // the short up-convs (kernel 4 at stride 4 or 2): streaming kernel (no residual / accumulate operands there)
  if (a.terms == 3 && a.phases > 1 && a.dil == 1 && a.pad == a.K - 1 && !a.res && !a.acc_prev && a.div == 1.f &&
      a.in_slope >= 0.f && a.in_slope <= 1.f && g16_ups_on()) {
    if (a.K == 2 && a.Cin == 64 && rows == 64) return launch_g16_ups<2, 2, 4, 1>(a, B, s);      // 64 -> 32 channels, stride 2
    if (a.K == 1 && a.Cin == 128 && rows == 256) return launch_g16_ups<1, 4, 4, 4>(a, B, s);    // 128 -> 64 channels, stride 4
  }
  // <MW, NW, WM, WN, TERMS>
  if (a.terms == 1) {
    if (rows % 128 == 0) return launch_g16_tile<4, 4, 2, 4, 1>(a, B, s);
    if (rows % 64 == 0) return launch_g16_tile<4, 2, 1, 8, 1>(a, B, s);
    if (rows % 32 == 0) return launch_g16_tile<2, 2, 1, 8, 1>(a, B, s);
    return hipErrorInvalidValue;
  }
  // Row tile: 128 rows x 256 columns (one block per CU) when that fills the chip; a launch with fewer blocks than CUs
  // (one or two utterances: the 256 / 128-channel stages) takes the 64- or 32-row tile and spreads over 2-4x the CUs.
#ifdef VSP_EXPERIMENTS
  static int force = -1, fill = 200;   // VSP_G16_ROWS=128|64|32 forces a row tile
  if (force < 0) {
    const char* e = getenv("VSP_G16_ROWS"); force = e ? atoi(e) : 0;
    if (const char* f = getenv("VSP_G16_FILL")) fill = atoi(f);
  }
#else
  constexpr int force = 0, fill = 200;
#endif
  const long col_tiles = (long)((a.Nq + 255) / 256) * B;
  int want = 128;
  if (force) want = force;
  else if (col_tiles * (rows / 128 > 0 ? rows / 128 : 1) < fill) want = col_tiles * (rows / 64 > 0 ? rows / 64 : 1) < fill ? 32 : 64;
  if (a.x_img) {                                                                        // input = operand image (LDS-DMA windows)
    if (a.K < 3 || a.phases != 1 || a.pad > CL_IMG_PADF || a.xi_tpad < cl_img_tpad(a.T_in) || (a.xi_bs & 7)) return hipErrorInvalidValue;
    if (rows % 128 == 0 && want >= 128) {
      if (!a.glen && g16_pipe_supported(a)) return launch_g16_pipe(a, B, s);             // persistent, pipelined across tiles (uniform batches)
      return launch_g16_tile<4, 4, 2, 4, 3, true>(a, B, s);
    }
    if (rows % 64 == 0 && want >= 64) return launch_g16_tile<4, 2, 1, 8, 3, true>(a, B, s);
    if (rows % 32 == 0) return launch_g16_tile<2, 2, 1, 8, 3, true>(a, B, s);
    return hipErrorInvalidValue;
  }
  // (an image-only result -- a ResBlock pair's intermediate -- of the 128-row tile: the instantiation whose epilogue goes
  // through the LDS, 1 KiB contiguous per store; bit-identical images)
#if G16_EPI_LDS && G16_IMG_EPI
  if (rows % 128 == 0 && want >= 128 && a.phases == 1 && !a.out && a.o_img && !a.res && !a.acc_prev)
    return launch_g16_tile<4, 4, 2, 4, 3, false, true>(a, B, s);
#endif
  if (rows % 128 == 0 && want >= 128) return launch_g16_tile<4, 4, 2, 4, 3>(a, B, s);   // 128 rows x 256 columns, one block per CU
  if (rows % 64 == 0 && want >= 64) return launch_g16_tile<4, 2, 1, 8, 3>(a, B, s);     //  64 rows x 256 columns
  if (rows % 32 == 0) return launch_g16_tile<2, 2, 1, 8, 3>(a, B, s);                   //  32 rows x 256 columns
  return hipErrorInvalidValue;
}

// pair kernel: a wave copies 0, 1 or 2 pieces of a slice (partial last slices of a chunk)
template <int NL>
__device__ __forceinline__ void g16_pair_wait(int pieces, int groups) {
  if (pieces == 0) g16_vm_wait<0, NL>(false, groups);
  else if (pieces == 1) g16_vm_wait<1, NL>(true, groups);
  else g16_vm_wait<2, NL>(true, groups);
}

// ------------------------------------------------------------------------------------------------------------
// Fused ResBlock1 conv PAIR of the 32- and 64-channel stages (reference modules.py:210-223):
//     y = x + conv2(lrelu(conv1(lrelu(x), dilation d) + b1), dilation 1) + b2   [+ previous sum] [/ div]
// One block = 8 waves x 32 time columns: conv1 on 256 columns from the staged x window, its tile (bias added,
// columns outside the utterance zeroed = conv2's padding) activated, split and written over the dead window as
// conv2's B image, conv2 on 256 - (K-1) columns, residual + store.  The intermediate never leaves the CU.
// The arithmetic per output (chunk-major, tap-minor, HH / CROSS / CROSS per step, bias in the accumulator) is that of
// g16_conv, so the result is bit-identical to the two-launch path.
//   NCH = C / 32 (1 or 2); G = taps per ring slot.
template <int NCH, int G, int TERMS, int NWV, int NS = 3>
__global__ void __launch_bounds__(64 * NWV, 4) g16_pair(ClPairArgs a) {
  constexpr int MW = 2 * NCH, NW = 2, C = 32 * NCH;
  constexpr int BT = 32 * NWV, WR = BT + G16_HALO, PL = WR * 16, XIMG = 4 * PL, XBUF = 2 * XIMG;
  constexpr int TAPB = MW * 2048;               // bytes of one tap in a ring slot
  constexpr int SLOT = G * TAPB;
  constexpr int RPS = 8 * NWV, NL = (WR + RPS - 1) / RPS;
  static_assert(NS == 2 || NS == 3, "ring depth: NS - 1 slices are requested ahead of the one being multiplied");
  constexpr int NPT = 2 * MW;                   // 1 KiB pieces per tap
  constexpr int NBWMAX = (G * NPT + NWV - 1) / NWV;
  constexpr bool EARLY_RES = NCH == 1 && TERMS == 3;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xw = lds;                         // x window chunk, later the t image chunk
  char* const Rg = lds + XBUF;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;

  // XCD-aware tile numbering: XCD k gets the k-th contiguous eighth of the (utterance, tile) sequence
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
  const int id = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int b = id / a.tiles, tile = id - b * a.tiles;

#ifdef G16_STAMPS
  // (diagnostic build, tools/stamps_pair.py: wave 0 of every 197th block of the VSP_STAMP_PAIR-th launch of this shape)
  int stamp_slot = -1, stamp_n = 0;
  if (wave == 0 && orig % 197 == 5 && (a.terms & 0x100)) {
    unsigned sl_ = 0;
    if (lane == 0) sl_ = atomicAdd(&g_g16_stamp_count, 1u);
    sl_ = __builtin_amdgcn_readfirstlane(sl_);
    stamp_slot = sl_ < (unsigned)G16_NSAMPLE ? (int)sl_ : -1;
  }
  G16_STAMPT(1);
#endif
  const int K = a.K, p2 = (K - 1) >> 1, p1 = a.dil * p2;
  const int R2 = BT - (K - 1);                  // output columns per block
  const int t0 = tile * R2;                     // first output column
  const int T = g16_len(a.glen, b, a.grate, a.T);   // (ragged batch: this utterance's own extent)
  if (t0 >= T) return;
  const int ns = (K + G - 1) / G;               // slices per chunk
  const int S1 = NCH * ns, S = 2 * S1;
  const int xrows = BT + (K - 1) * a.dil;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, T * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.o_bs, 0, T * C * 4,
                                                                      0x00020000);

  // ---- x window staging (as in g16_conv): row 0 of the window is time t0 - p2 - p1
  const int g16 = tid >> 4, kq_s = g16 & 3, row_s = (g16 >> 2) * 8 + ((tid >> 1) & 7), half_s = tid & 1;
  const int st_voff = (row_s * C + (2 * kq_s + half_s) * 4) * 4;
  const int st_loff = kq_s * PL + row_s * 16 + half_s * 8;
  u32x4 sv[NL];
  const float slope = a.slope;
  auto x_issue = [&](int chunk) {
    const int base = ((t0 - p2 - p1) * C + chunk * 32) * 4;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const bool in = (u + 1) * RPS <= BT || row_s + u * RPS < xrows;
      sv[u] = (G16_DIAG & 16) ? u32x4{1u, 2u, 3u, 4u}
                              : __builtin_amdgcn_raw_buffer_load_b128(rx, in ? st_voff + (base + u * RPS * C * 4) : G16_OOR, 0, 0);
    }
  };
  auto x_write = [&]() {
    char* dst0 = Xw + st_loff;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      f16x4 eh, el;
      g16_split4(g16_as_f32x4(sv[u]), slope, true, eh, el);
      *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16) = eh;
      if constexpr (TERMS == 3) *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16 + XIMG) = el;
    }
  };

  // ---- weight slices: global slice index s in [0, S): conv = s / S1, then chunk-major, G taps per slice.
  //      Cursor of the next slice to request: (dv, dc, dsl); returns the pieces THIS wave issued.
  int dv = 0, dc = 0, dsl = 0;
  auto dma_next = [&](int slot) -> int {
    const uint4* Wg = reinterpret_cast<const uint4*>(dv ? a.w2h : a.w1h);
    const int tap0 = dsl * G;
    const int pieces = ((K - tap0) < G ? (K - tap0) : G) * NPT;
    const size_t src = (G16_DIAG & 32) ? 0 : ((size_t)dc * K + tap0) * MW * 128;        // uint4 units
    int mine = 0;
#pragma unroll
    for (int u = 0; u < NBWMAX; ++u) {
      const int p = u * NWV + wave;
      if (p < pieces && (G16_DIAG & 2) == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wg + src + (size_t)p * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(Rg + slot * SLOT + p * 1024), 16, 0, 0);
        ++mine;
      }
    }
    if (++dsl == ns) { dsl = 0; if (++dc == NCH) { dc = 0; ++dv; } }
    return mine;
  };

  // ---- fragments
  const int xb_lane = q4 * PL + (wave * 32 + l15) * 16;
  const int wa_lane = lane * 16;
  f32x4 hh[MW][NW];
  // MFMAs of one slice: taps [tap0, tap0 + nt) of one chunk; rowstep = dilation of the conv
  auto slice = [&](int slot, int tap0, int nt, int rowstep) {
#ifdef G16_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    for (int g = 0; g < nt; ++g) {
      f16x8 Bh[NW], Bl[NW];
      const char* pb = Xw + xb_lane + (tap0 + g) * rowstep * 16;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        Bh[j] = *reinterpret_cast<const f16x8*>(pb + j * 256);
        if constexpr (TERMS == 3) Bl[j] = *reinterpret_cast<const f16x8*>(pb + j * 256 + XIMG);
      }
      const char* pa = Rg + slot * SLOT + g * TAPB + wa_lane;
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        f16x8 Ah, Al;
        Ah = *reinterpret_cast<const f16x8*>(pa + i * 2048);
        if constexpr (TERMS == 3) Al = *reinterpret_cast<const f16x8*>(pa + i * 2048 + 1024);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          hh[i][j] = G16_MFMA(Ah, Bh[j], hh[i][j]);
          if constexpr (TERMS == 3) {
            hh[i][j] = G16_MFMA(Al, Bh[j], hh[i][j]);
            hh[i][j] = G16_MFMA(Ah, Bl[j], hh[i][j]);
          }
        }
      }
    }
#ifdef G16_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto init_acc = [&](const float* bias) {
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * i + 4 * q4);
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = bv;
    }
  };

  // ================= prologue =================
  // EARLY_RES: the residual values are requested together with the window -- same lines at the same time, so the
  // second request hits L2 (by epilogue time they would have left it: a second HBM pass)
  [[maybe_unused]] u32x4 res_early[MW][NW];
  if constexpr (EARLY_RES) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int r = wave * 32 + 16 * j + l15;
        res_early[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, r < R2 ? (t0 + r) * C * 4 + (16 * i + 4 * q4) * 4 : G16_OOR, 0, 0);
      }
  }
  init_acc(a.b1);
  x_issue(0);
  int pc_a = dma_next(0);            // pieces of slice s + 1 / s + 2 issued by this wave (for the counted waits)
  int pc_b = NS == 3 ? dma_next(1) : 0;
  G16_STAMPT(2);                     // requests out
  x_write();
  G16_STAMPT(3);                     // x window converted and written
  int xl_a = 0, xl_b = 0;            // window loads issued behind slice s + 1 / s + 2
  int slot = 0;
  // at the top of step s: slices s and s + 1 are in flight or landed; pc_a / pc_b = my pieces of s + 1 / s + 2
  // (rotated below).  The invariant on entry: pc_a = pieces(slice 0)... kept as (cur, next) pair:
  int pc_cur = pc_a, pc_nxt = pc_b;
  (void)pc_cur;

  // ================= conv1 =================
  int chunk = 0, sl = 0;
  for (int s = 0; s < S1; ++s) {
    // slice s has landed: issued after it are slice s + 1 and the window loads of the last two steps
    // (two slots: slice s went out a step ago, AFTER the window request of the step before that: only the last step's
    // request is younger)
    G16_STAMPT(10);
    g16_pair_wait<NL>(pc_nxt, NS == 3 ? xl_a + xl_b : xl_b);
    G16_STAMPT(11);
    G16_BARRIER();                                   // slice s (and a freshly written window) visible to all
    G16_STAMPT(12);
    pc_cur = pc_nxt;
    if constexpr (NS == 3) pc_nxt = s + 2 < S ? dma_next(slot == 0 ? 2 : slot - 1) : 0;     // slot of slice s - 1
    else { if (s + 1 < S) (void)dma_next(slot ^ 1); pc_nxt = 0; }   // two slots: slice s + 1 goes out now, nothing else is in flight at the next wait
    G16_STAMPT(13);
    xl_a = xl_b;
    xl_b = 0;
    const bool last_sl = sl == ns - 1;
    if constexpr (NCH > 1) {
      // the next chunk's window is requested two slices before its staging (or at the chunk's start)
      if (chunk + 1 < NCH && sl == (ns > 2 ? ns - 3 : 0)) { x_issue(chunk + 1); xl_b = 1; }
    }
    const int tap0 = sl * G;
    slice(slot, tap0, (K - tap0) < G ? (K - tap0) : G, a.dil);
    G16_STAMPT(14);
    if constexpr (NCH > 1) {
      if (last_sl && chunk + 1 < NCH) {
        G16_BARRIER();                               // every wave is done reading this chunk's window
        G16_STAMPT(15);
        x_write();
        G16_STAMPT(16);
      }
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
    if (last_sl) { sl = 0; ++chunk; } else ++sl;
  }

  // conv1 tile -> activated fp32 values (bias is in hh); columns outside the utterance are conv2's zero padding
  f32x4 tv[MW][NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int tt = t0 - p2 + wave * 32 + 16 * j + l15;
    const float f = tt >= 0 && tt < T ? G16_UNSCALE : 0.f;   // (the unscaling and the zero padding of columns outside the utterance in ONE multiply: the factor is per column)
#pragma unroll
    for (int i = 0; i < MW; ++i) tv[i][j] = hh[i][j] * f;
  }

  G16_STAMPT(20);
  // ================= conv2 =================
  init_acc(a.b2);
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) {
    // t image chunk c2 = conv1 output channels [32 c2, 32 c2 + 32) = m-tiles 2 c2, 2 c2 + 1: a lane's four channels
    // 16 i + 4 q4 .. + 3 sit in plane 2 (i & 1) + (q4 >> 1) at byte 8 (q4 & 1) of the row's 16
    G16_STAMPT(21);
    G16_BARRIER();                                   // nobody still reads the region (window / previous t chunk)
    G16_STAMPT(22);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        f16x4 eh, el;
        g16_split4(tv[2 * c2 + ii][j], slope, true, eh, el);
        char* dst = Xw + (2 * ii + (q4 >> 1)) * PL + (wave * 32 + 16 * j + l15) * 16 + 8 * (q4 & 1);
        *reinterpret_cast<f16x4*>(dst) = eh;
        if constexpr (TERMS == 3) *reinterpret_cast<f16x4*>(dst + XIMG) = el;
      }
    G16_STAMPT(23);                                  // image chunk written
    for (int sl2 = 0; sl2 < ns; ++sl2) {
      const int s = S1 + c2 * ns + sl2;
      G16_STAMPT(10);
      g16_pair_wait<NL>(pc_nxt, NS == 3 ? xl_a + xl_b : xl_b);
      G16_STAMPT(11);
      G16_BARRIER();
      G16_STAMPT(12);
      pc_cur = pc_nxt;
      if constexpr (NS == 3) pc_nxt = s + 2 < S ? dma_next(slot == 0 ? 2 : slot - 1) : 0;
      else { if (s + 1 < S) (void)dma_next(slot ^ 1); pc_nxt = 0; }
      G16_STAMPT(13);
      xl_a = xl_b;
      xl_b = 0;
      const int tap0 = sl2 * G;
      slice(slot, tap0, (K - tap0) < G ? (K - tap0) : G, 1);
      G16_STAMPT(14);
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
  }

  // ---- epilogue: y = conv2 + x (+ previous resblock sum) (/ div); columns >= R2 belong to the next tile
  G16_STAMPT(30);
  // (round 5: the through-the-LDS form of g16_conv / g16_pp measured +-0 here -- two resident blocks per CU already hide
  // the epilogue's round trips -- and cost 4 more spilled registers: profiles/r05_coalesced_epilogues.txt)
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int r = wave * 32 + 16 * j + l15;
      const int off = r < R2 ? (t0 + r) * C * 4 + (16 * i + 4 * q4) * 4 : G16_OOR;
      f32x4 v = hh[i][j] * G16_UNSCALE;
      if constexpr (EARLY_RES) v += g16_as_f32x4(res_early[i][j]);
      else v += g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      if (a.acc_prev) v += g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0));
      g16_div(v, a.div);
      __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, off, 0, 0);
    }
#ifdef G16_STAMPS
  G16_STAMPT(31);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G16_STAMPT(32);
#endif
}

template <int NCH, int G, int TERMS, int NWV, int NS = 3>
static hipError_t launch_g16_pair_tile(ClPairArgs a, int B, hipStream_t s) {
  constexpr int BT = 32 * NWV;
  constexpr size_t lds = (size_t)2 * 4 * (BT + G16_HALO) * 16 + (size_t)NS * G * 2 * NCH * 2048;
  static_assert(lds <= (NWV == 8 ? 80 : 160) * 1024, "two 8-wave blocks or one 16-wave block per CU");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = g16_pair<NCH, G, TERMS, NWV, NS>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  const int R2 = BT - (a.K - 1);
  a.tiles = (a.T + R2 - 1) / R2;
  const long n = (long)a.tiles * B;
  if (n <= 0 || n > 0x7fffffffL) return hipErrorInvalidValue;
#ifdef G16_STAMPS
  {  // stamps only in the VSP_STAMP_PAIR-th launch of this kernel shape (0-based)
    static int launch_no = 0;
    static int target = -2;
    if (target == -2) { const char* e = getenv("VSP_STAMP_PAIR"); target = e ? atoi(e) : -1; }
    if (launch_no++ == target) a.terms |= 0x100;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64 * NWV), lds, s, a);
  return hipGetLastError();
}

bool g16_pair_supported(int C, int K, int dil) {
  return (C == 32 || C == 64) && K >= 1 && (K & 1) && (K - 1) * dil <= G16_HALO && K - 1 < 128;
}

hipError_t launch_g16_pair(const ClPairArgs& a, int B, hipStream_t s) {
  if (a.C == 128) return launch_g16_pp(a, B, s);       // the ping-pong tile pair (gen16_pp.hip)
  if (!g16_pair_supported(a.C, a.K, a.dil) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out)
    return hipErrorInvalidValue;
  // block shape: 8 waves x 32 columns, two blocks per CU (the 16-wave one-block-per-CU form -- one weight ring per CU,
  // half the LDS-DMA pieces -- measured 12 % slower: profiles/r02_same_box_ab.txt; experiment builds keep it)
#ifdef VSP_EXPERIMENTS
  static int nwv = -1;
  if (nwv < 0) { const char* e = getenv("VSP_PAIR_WAVES"); nwv = e ? atoi(e) : 8; }
#else
  constexpr int nwv = 8;
#endif
  // round 4: weights in registers, persistent blocks (gen16_rw.hip) where that kernel exists; ClPairArgs::ring (VSP_PAIR=ring) keeps the
  // LDS-ring kernel below (the second implementation under test: bit-identical)
  if (!a.ring && g16_rw_supported(a.C, a.K, a.dil, a.terms)) return launch_g16_rw(a, B, s);
  // round 5: the same for the kernel-3 pairs of the 64-channel stage (gen16_rw64.hip) -- bit-identical, measured 2-5 % SLOWER
  // than the ring kernel below (profiles/r05_g16_rw64_k3_pairs_64_channels.txt): opt-in (ClPairArgs::rw64, VSP_RW64=1)
  if (!a.ring && a.rw64 && g16_rw64_supported(a.C, a.K, a.dil, a.terms)) return launch_g16_rw64(a, B, s);
  if (a.terms == 1)
    return a.C == 32 ? launch_g16_pair_tile<1, 2, 1, 8>(a, B, s) : launch_g16_pair_tile<2, 1, 1, 8>(a, B, s);
  // round 3: TWO ring slots of twice the taps (32 channels: 4 taps = 16 KB, 64 channels: 2 taps = 16 KB; 72 KB of LDS
  // per block, still two blocks per CU): the slice hand-over -- counted wait, barrier, LDS-DMA issue: 0.5 us per slice
  // in the stamps of profiles/r03_pair_kernel_phase_stamps.txt, more than the slice's MFMAs -- happens half as often
  // (generator -1.2 ms same box; the request is only one slice ahead now, which gives part of it back)
  if (nwv == 8) return a.C == 32 ? launch_g16_pair_tile<1, 4, 3, 8, 2>(a, B, s) : launch_g16_pair_tile<2, 2, 3, 8, 2>(a, B, s);
#ifdef VSP_EXPERIMENTS
  return a.C == 32 ? launch_g16_pair_tile<1, 2, 3, 16>(a, B, s) : launch_g16_pair_tile<2, 1, 3, 16>(a, B, s);
#else
  return hipErrorInvalidValue;
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Fused ResBlock1 CHAIN (reference modules.py:210-223, the whole loop `for c1, c2 in zip(convs1, convs2)`): np conv
// pairs in ONE launch.  HBM sees x once and the result once (a pair launch per dilation moves 3 passes each); the
// price is the chain's halo H = sum of the paddings of its 2 np convolutions, recomputed per side of every tile.
//   * every convolution runs on the block's full BT columns with FIXED column <-> time mapping (column c = time
//     tb + c): its input image sits in LDS with GRD guard rows on either side, a tap reads row c + tap * dil - pad.
//     Columns within the accumulated padding of a tile edge compute garbage that never reaches a valid column
//     (a D column depends on its own B column only); columns [H, BT - H) are exact and are the ones stored;
//   * the running x_p lives in registers in D-tile layout (lane = column, four consecutive channels), so the
//     residual add is lane-local and x is read from HBM exactly once; the images (leaky-relu, hi / lo split, zero
//     outside the utterance = the reference's zero padding of EVERY convolution input) are written from registers;
//   * weights stream through the 3-slot LDS-DMA ring as one sequence of slices over the 2 np convolutions.
// Per output the arithmetic is that of g16_pair / g16_conv (chunk-major, tap-minor, HH / CROSS / CROSS, bias in the
// accumulator, acc * 2^-8 + x): bit-identical to the pair-per-launch and conv-per-launch paths.
template <int NCH, int NW, int G, int TERMS, int NWV>
__global__ void __launch_bounds__(64 * NWV, 2) g16_chain(ClChainArgs a) {
  constexpr int MW = 2 * NCH, C = 32 * NCH, CW = 16 * NW;   // CW = columns per wave
  constexpr int BT = CW * NWV, GRD = G16_HALO / 2, WR = BT + G16_HALO, PL = WR * 16, XIMG = 4 * PL, XBUF = 2 * XIMG;
  constexpr int TAPB = MW * 2048;               // bytes of one tap in a ring slot
  constexpr int SLOT = G * TAPB;
  constexpr int NS = 3;
  constexpr int NPT = 2 * MW;                   // 1 KiB pieces per tap
  constexpr int NBWMAX = (G * NPT + NWV - 1) / NWV;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xw = lds;                         // NCH chunk images of the current convolution's input
  char* const Rg = lds + NCH * XBUF;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;

  // XCD-aware tile numbering: XCD k gets the k-th contiguous eighth of the (utterance, tile) sequence
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
  const int id = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int b = id / a.tiles, tile = id - b * a.tiles;

#ifdef G16_STAMPS
  int stamp_slot = -1, stamp_n = 0;
  if (wave == 0 && orig % 97 == 5 && (a.terms & 0x100)) {
    unsigned sl_ = 0;
    if (lane == 0) sl_ = atomicAdd(&g_g16_stamp_count, 1u);
    sl_ = __builtin_amdgcn_readfirstlane(sl_);
    stamp_slot = sl_ < (unsigned)G16_NSAMPLE ? (int)sl_ : -1;
  }
  G16_STAMPT(1);
#endif
  const int K = a.K, p2 = (K - 1) >> 1, H = a.halo;
  const int R = BT - 2 * H;                     // columns stored per block
  const int tb = tile * R - H;                  // time of column 0
  const int T = g16_len(a.glen, b, a.grate, a.T);   // (ragged batch: this utterance's own extent)
  if (tile * R >= T) return;
  const int ns = (K + G - 1) / G;               // slices per chunk
  const int S = 2 * a.np * NCH * ns;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, T * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.o_bs, 0, T * C * 4,
                                                                      0x00020000);
  const float slope = a.slope;

  // ---- x_0 in D-tile layout; zero outside the utterance
  bool tval[NW];
  f32x4 xr[MW][NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int t = tb + wave * CW + 16 * j + l15;
    tval[j] = t >= 0 && t < T;
#pragma unroll
    for (int i = 0; i < MW; ++i)
      xr[i][j] = (G16_DIAG & 16) ? f32x4{1.f, 2.f, 3.f, 4.f}
                                 : g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(
                                       rx, tval[j] ? (t * C + 16 * i + 4 * q4) * 4 : G16_OOR, 0, 0));
  }

  // ---- weight slices: one sequence over the 2 np convolutions, chunk-major, G taps per slice
  int dv = 0, dc = 0, dsl = 0;
  auto dma_next = [&](int slot) -> int {
    const uint4* Wg = reinterpret_cast<const uint4*>(a.w[dv]);
    const int tap0 = dsl * G;
    const int pieces = ((K - tap0) < G ? (K - tap0) : G) * NPT;
    const size_t src = (G16_DIAG & 32) ? 0 : ((size_t)dc * K + tap0) * MW * 128;        // uint4 units
    int mine = 0;
#pragma unroll
    for (int u = 0; u < NBWMAX; ++u) {
      const int p = u * NWV + wave;
      if (p < pieces && (G16_DIAG & 2) == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wg + src + (size_t)p * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(Rg + slot * SLOT + p * 1024), 16, 0, 0);
        ++mine;
      }
    }
    if (++dsl == ns) { dsl = 0; if (++dc == NCH) { dc = 0; ++dv; } }
    return mine;
  };

  // ---- image of a D-layout tile set: leaky-relu, split, zero outside the utterance.  A lane's four channels
  //      16 i + 4 q4 .. + 3 sit in chunk i / 2, plane 2 (i & 1) + (q4 >> 1), at byte 8 (q4 & 1) of the row's 16.
  auto write_image = [&](const f32x4 (&v)[MW][NW]) {
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        f16x4 eh, el;
        g16_split4(tval[j] ? v[i][j] : f32x4{0.f, 0.f, 0.f, 0.f}, slope, true, eh, el);
        char* dst = Xw + (i >> 1) * XBUF + (2 * (i & 1) + (q4 >> 1)) * PL + (GRD + wave * CW + 16 * j + l15) * 16 + 8 * (q4 & 1);
        *reinterpret_cast<f16x4*>(dst) = eh;
        if constexpr (TERMS == 3) *reinterpret_cast<f16x4*>(dst + XIMG) = el;
      }
  };

  // ---- convolution main loop.  One STEP = one (chunk, tap): 2 MW A fragments (weights, from the ring) and 2 NW B
  //      fragments (image rows shifted by tap * dilation), MW * NW * 3 MFMAs.  Software pipeline inside every wave:
  //      the A fragments are double-buffered in registers (step s + 1 requested before the MFMAs of step s), the B
  //      fragments of an n-tile are re-requested in place right after its last MFMA of the step has issued; the
  //      waits are counted (LDS returns in order), so a wave's matrix work runs in the shadow of its own reads and
  //      the 4 waves of a SIMD do not have to find each other in different phases.
  //      Ring: slices n, n + 1, n + 2 are resident or in flight.  retire(): every wave has its last fragments of
  //      slice n in registers -> wait for slice n + 1, barrier, request slice n + 3 into the freed slot.  Inside a
  //      convolution this happens at the START of the slice's last step (whose fragments were requested a step ago).
  constexpr int RA = MW * (TERMS == 3 ? 2 : 1), RB = TERMS == 3 ? 2 : 1;   // LDS reads per A set / per n-tile
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned wa_lane = lds0 + NCH * XBUF + lane * 16;
  constexpr int gs = G;                         // taps per slice (the last one of a chunk may be shorter)
  f16x8 Ah[2][MW], Al[2][MW], Bh[NW], Bl[NW];
  f32x4 hh[MW][NW];

  int n_cur = 0, n_issued = 0, pc_last = 0;     // slice being read; slices requested; my pieces of the youngest one
  auto issue_slice = [&]() {
    if (n_issued < S) { pc_last = dma_next(n_issued % NS); ++n_issued; }
  };
  auto wait_landed = [&](int n) {               // slice n (< n_issued) has landed: younger are n + 1 .. n_issued - 1
    const int younger = n_issued - 1 - n;
    if (younger <= 0) g16_vmcnt<0>();
    else if (younger == 1) { if (pc_last == 0) g16_vmcnt<0>(); else if (pc_last == 1) g16_vmcnt<1>(); else g16_vmcnt<2>(); }
    else { if (pc_last == 0) g16_vmcnt<0>(); else g16_vmcnt<1>(); }   // (stricter than needed: the older one's count is not kept)
  };
  auto retire = [&]() {
    if (n_cur + 1 < S) {
      G16_STAMPT(11);
      wait_landed(n_cur + 1);
      G16_STAMPT(16);
      G16_BARRIER();
      G16_STAMPT(17);
      issue_slice();
      G16_STAMPT(12);
    }
    ++n_cur;
  };
  auto read_a = [&](auto P, unsigned a_addr) {
    constexpr int pp = decltype(P)::value;
    g16_for<MW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      Ah[pp][i] = g16_lds_read<i * 2048>(a_addr);
      if constexpr (TERMS == 3) Al[pp][i] = g16_lds_read<i * 2048 + 1024>(a_addr);
    });
  };
  auto read_b = [&](auto J, unsigned b_addr) {
    constexpr int j = decltype(J)::value;
    Bh[j] = g16_lds_read<j * 256>(b_addr);
    if constexpr (TERMS == 3) Bl[j] = g16_lds_read<j * 256 + XIMG>(b_addr);
  };
  auto mfma_col = [&](auto P, auto J) {
    constexpr int pp = decltype(P)::value, j = decltype(J)::value;
    __builtin_amdgcn_sched_barrier(0);
    g16_for<MW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      hh[i][j] = G16_MFMA(Ah[pp][i], Bh[j], hh[i][j]);
      if constexpr (TERMS == 3) {
        hh[i][j] = G16_MFMA(Al[pp][i], Bh[j], hh[i][j]);
        hh[i][j] = G16_MFMA(Ah[pp][i], Bl[j], hh[i][j]);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };

  // one convolution over the image in Xw: dilation `rowstep`, bias b; result in hh.  On entry the image and
  // slice n_cur (the convolution's first) are visible to every wave.
  auto conv = [&](const float* bias, int rowstep) {
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * i + 4 * q4);
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = bv;
    }
    const unsigned xb0 = lds0 + q4 * PL + (GRD + wave * CW + l15 - rowstep * p2) * 16;
    const int steps = NCH * K;
    int chunk = 0, tap = 0, gtap = 0;             // of the step being multiplied; gtap = tap within its slice
    unsigned a_addr = wa_lane + (n_cur % NS) * SLOT, b_addr = xb0;
    read_a(std::integral_constant<int, 0>{}, a_addr);
    g16_for<NW>([&](auto J) { read_b(J, b_addr); });
    G16_STAMPT(10);
    auto step = [&](auto P, int st) {
      constexpr int pp = decltype(P)::value;
      if (st + 1 < steps) {
        const bool slice_end = gtap == gs - 1 || tap == K - 1;
        if (slice_end) retire();                        // (the barrier drains this wave's reads: the counted waits below still hold)
        const bool chunk_end = tap == K - 1;
        chunk = chunk_end ? chunk + 1 : chunk;
        tap = chunk_end ? 0 : tap + 1;
        gtap = slice_end ? 0 : gtap + 1;
        a_addr = wa_lane + (n_cur % NS) * SLOT + gtap * TAPB;
        b_addr = xb0 + chunk * XBUF + tap * rowstep * 16;
      }
      // (the last step of a convolution re-requests its own fragments: one code path, uniform counted waits)
      read_a(std::integral_constant<int, 1 - pp>{}, a_addr);
      g16_for<NW>([&](auto J) {
        // in flight behind the fragments this n-tile needs: the other n-tiles' refills and the next A set
        g16_lgkmcnt<(RA + (NW - 1) * RB < 15 ? RA + (NW - 1) * RB : 15)>();   // (a 4-bit counter)
        mfma_col(P, J);
        read_b(J, b_addr);
      });
    };
    for (int st = 0; st < steps; st += 2) {
      step(std::integral_constant<int, 0>{}, st);
      if (st + 1 < steps) step(std::integral_constant<int, 1>{}, st + 1);
    }
    // nothing may still be writing the fragment registers when the compiler reuses them
    g16_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    G16_STAMPT(13);
#pragma unroll
    for (int i = 0; i < MW; ++i) asm volatile("" ::"v"(Ah[0][i]), "v"(Ah[1][i]), "v"(Al[0][i]), "v"(Al[1][i]));
#pragma unroll
    for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(Bh[j]), "v"(Bl[j]));
  };
  auto result = [&](int i, int j) -> f32x4 {
    return hh[i][j] * G16_UNSCALE;
  };

  issue_slice(); issue_slice(); issue_slice();
  write_image(xr);
  G16_STAMPT(2);
  wait_landed(0);
  G16_BARRIER();
  G16_STAMPT(3);
  const int ncv = 2 * a.np;
  for (int cv = 0; cv < ncv; ++cv) {
    const bool second = cv & 1;
    conv(a.b[cv], second ? 1 : a.dil[cv >> 1]);
    if (cv + 1 == ncv) break;
    f32x4 nx[MW][NW];                                  // the next convolution's input: conv1's output or the new x
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        nx[i][j] = result(i, j);
        if (second) { nx[i][j] += xr[i][j]; xr[i][j] = nx[i][j]; }
      }
    retire();                                          // also: nobody still reads the image
    write_image(nx);
    G16_STAMPT(14);
    G16_BARRIER();
    G16_STAMPT(15);
  }
  G16_STAMPT(20);

  // ---- epilogue: out = conv2 + x_{np-1} (+ previous resblock sum) (/ div) on the exact columns
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int col = wave * CW + 16 * j + l15;
      const int t = tb + col;
      const int off = (col >= H && col < H + R && t < T) ? (t * C + 16 * i + 4 * q4) * 4 : G16_OOR;
      f32x4 v = result(i, j);
      v += xr[i][j];
      if (a.acc_prev) v += g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0));
      g16_div(v, a.div);
      if ((G16_DIAG & 8) == 0) __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, off, 0, 0);
    }
#ifdef G16_STAMPS
  G16_STAMPT(21);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G16_STAMPT(22);
#endif
}

template <int NCH, int NW, int G, int TERMS, int NWV>
static hipError_t launch_g16_chain_tile(ClChainArgs a, int B, hipStream_t s) {
  constexpr int BT = 16 * NW * NWV;
  constexpr size_t lds = (size_t)NCH * 2 * 4 * (BT + G16_HALO) * 16 + (size_t)3 * G * 2 * NCH * 2048;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = g16_chain<NCH, NW, G, TERMS, NWV>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  const int R = BT - 2 * a.halo;
  if (R < 32) return hipErrorInvalidValue;
  a.tiles = (a.T + R - 1) / R;
  const long n = (long)a.tiles * B;
  if (n <= 0 || n > 0x7fffffffL) return hipErrorInvalidValue;
#ifdef G16_STAMPS
  {  // stamps only in the VSP_STAMP_CHAIN-th chain launch of this shape (0-based)
    static int launch_no = 0;
    static int target = -2;
    if (target == -2) { const char* e = getenv("VSP_STAMP_CHAIN"); target = e ? atoi(e) : -1; }
    if (launch_no++ == target) a.terms |= 0x100;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64 * NWV), lds, s, a);
  return hipGetLastError();
}

static int g16_chain_halo(int K, const int* dil, int np) {
  int h = 0;
  for (int p = 0; p < np; ++p) h += (dil[p] + 1) * ((K - 1) / 2);
  return h;
}

bool g16_chain_supported(int C, int K, const int* dil, int np) {
  if (!(C == 32 || C == 64 || C == 128) || np < 1 || np > 3 || K < 1 || !(K & 1)) return false;
  for (int p = 0; p < np; ++p)
    if (dil[p] < 1 || dil[p] * ((K - 1) / 2) > G16_HALO / 2) return false;
  // 128 channels: 128-column blocks (the four chunk images and the ring fill the LDS)
  return (C == 128 ? 128 : 256) - 2 * g16_chain_halo(K, dil, np) >= 32;
}

hipError_t launch_g16_chain(const ClChainArgs& a0, int B, hipStream_t s) {
  ClChainArgs a = a0;
  if (!g16_chain_supported(a.C, a.K, a.dil, a.np) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out)
    return hipErrorInvalidValue;
  // round 4: the kernel-3 ResBlock of the 32-channel stage as a role pipeline with the weights in registers (gen16_rc.hip)
  if (!a.ring && g16_rc_supported(a.C, a.K, a.dil, a.np, a.terms, a.acc_prev)) return launch_g16_rc(a, B, s);
  a.halo = g16_chain_halo(a.K, a.dil, a.np);
  // few, fat waves (64 columns x all channels each, up to 256 registers).  32 channels: one 8-wave block of 512
  // columns per CU when the halo would eat more than a quarter of a 256-column tile, else two 4-wave blocks of 256
  // columns (their convolution hand-offs overlap)
#ifdef VSP_EXPERIMENTS
  static int force = -1;                 // VSP_CHAIN_WAVES=4|8 forces one shape
  if (force < 0) { const char* e = getenv("VSP_CHAIN_WAVES"); force = e ? atoi(e) : 0; }
#else
  constexpr int force = 0;
#endif
  const bool wide = a.C == 32 && (force ? force == 8 : 2 * a.halo > 64);
  // <NCH, NW, G, TERMS, NWV>
  if (a.C == 128) return a.terms == 1 ? launch_g16_chain_tile<4, 1, 1, 1, 8>(a, B, s) : launch_g16_chain_tile<4, 1, 1, 3, 8>(a, B, s);
  if (a.terms == 1) {
    if (a.C == 64) return launch_g16_chain_tile<2, 2, 2, 1, 8>(a, B, s);
    return wide ? launch_g16_chain_tile<1, 4, 4, 1, 8>(a, B, s) : launch_g16_chain_tile<1, 4, 2, 1, 4>(a, B, s);
  }
  if (a.C == 64) return launch_g16_chain_tile<2, 2, 2, 3, 8>(a, B, s);
  return wide ? launch_g16_chain_tile<1, 4, 4, 3, 8>(a, B, s) : launch_g16_chain_tile<1, 4, 2, 3, 4>(a, B, s);
}

}  // namespace vsp
