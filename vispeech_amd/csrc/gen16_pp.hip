// g16_pp: the fused ResBlock1 conv PAIR of the 128-channel stage on the ping-pong tile of g16_conv (round 4).
//     y = x + conv2(lrelu(conv1(lrelu(x), dilation d) + b1), dilation 1) + b2   [+ previous sum] [/ div]
// (reference modules.py:215-222; the resblock sum / average of models.py:276-285 on a ResBlock's last pair).
//
// Why: the kernel-3 convolutions of this stage run in the HBM neighbourhood (a pair as two g16_conv launches moves five
// passes of T x 128 x 4 bytes -- x, the intermediate's operand image out and in, the residual, y -- in 1.65 ms: 3.1 TB/s);
// the intermediate's round trip and the second read of x are what a pair kernel removes.  The round-2 attempt
// (g16_chain with 16 columns x 128 channels per wave: 16 KB of A fragments per 24 MFMAs, no ping-pong) was LDS-bound
// and slower than the two launches; this one keeps g16_conv's arithmetic intensity.
//
// One block = 8 waves = 2 row groups (64 output channels each) x 4 column groups (16 NW columns each), one block per CU:
//   * the x window of ALL four 32-channel chunks is resident as operand images (leaky-relu applied, hi / lo split,
//     4 planes of [row][8 halfs] per chunk and image): chunk c + 1 is converted and written during chunk c's last step;
//   * conv1 on BT columns, steps chunk-major / tap-minor as everywhere, weight slices (16 KB: 128 rows x 32 channels x
//     hi | lo) through a 3-slot LDS-DMA ring two slices ahead, counted vmcnt, raw s_barrier;
//   * PING-PONG as in g16_conv: waves w and w + 4 share a SIMD and alternate MEM phases (all fragments of a step into
//     registers + the wave's share of DMA / staging) and MFMA phases (MW * NW * 3 MFMAs back to back);
//   * hand-over: conv1's tile (bias in the accumulator, columns outside the utterance zeroed = conv2's padding) is
//     activated, split and written OVER the dead window with one ds_write_b64 per tile and image; each wave half does
//     that in the phase in which the other one still multiplies / idles, the ring keeps streaming;
//   * conv2 on BT - (K - 1) columns from the t images, residual + store as g16_conv's epilogue.
// The arithmetic per output (chunk-major, tap-minor, HH / CROSS / CROSS, bias in the accumulator, acc * 2^-8 + x,
// + previous, / div) is that of g16_conv / g16_pair: the result is BIT-IDENTICAL to the two-launch path.
#include "kernels.h"

#include <cstdlib>
#include <cstring>

#include "g16_common.h"

namespace vsp {

// PP_STAMPS (diagnostic build, tools/stamps_pp.py): lane 0 of every wave of the middle block records tagged wall-clock
// stamps (s_memrealtime, 100 MHz) at its phase boundaries.
#ifdef PP_STAMPS
// (stamps live in REGISTERS -- lane n of three register pairs holds stamp n -- and are stored once at the end: a store per
// stamp would sit in the vmcnt queue and turn every counted wait of the stamped waves into a stricter one)
constexpr int PP_NSTAMP = 192;
#define pp_wl(v, l, old) (lane == (l) ? (v) : (old))
__device__ unsigned long long g_pp_stamps[8][PP_NSTAMP];
#define PP_STAMP(tag)                                                                                         \
  do {                                                                                                        \
    if (stamp_on) {                                                                                           \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                         \
      const unsigned lo_ = (unsigned)t_, hi_ = ((unsigned)(t_ >> 32) & 0xffffffu) | ((unsigned)(tag) << 24);  \
      const int g_ = stamp_n >> 6, l_ = stamp_n & 63;                                                         \
      if (g_ == 0) { st_lo[0] = pp_wl(lo_, l_, st_lo[0]); st_hi[0] = pp_wl(hi_, l_, st_hi[0]); } \
      else if (g_ == 1) { st_lo[1] = pp_wl(lo_, l_, st_lo[1]); st_hi[1] = pp_wl(hi_, l_, st_hi[1]); } \
      else if (g_ == 2) { st_lo[2] = pp_wl(lo_, l_, st_lo[2]); st_hi[2] = pp_wl(hi_, l_, st_hi[2]); } \
    }                                                                                                         \
    ++stamp_n;                                                                                                \
  } while (0)
extern "C" int vsp_debug_stamps_pp(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 8 * PP_NSTAMP);
}
#else
#define PP_STAMP(tag) ((void)0)
#endif

#ifndef PP_EPI_LDS
#define PP_EPI_LDS 1
#endif
template <int NW, int HALO>
__global__ void __launch_bounds__(512) g16_pp(ClPairArgs a) {
  constexpr int NCH = 4, C = 32 * NCH, MW = 4, WN = 4, NWV = 8, MTB = 8;
  constexpr int BT = 16 * NW * WN;              // conv1 columns per block
  constexpr int WR = BT + HALO;                 // window rows allocated
  constexpr int PL = WR * 16, XIMG = 4 * PL, XBUF = 2 * XIMG, WIN = NCH * XBUF;
  constexpr int SLOT = MTB * 2048, NS = 3;
  constexpr int NBW = 2 * MTB / NWV;            // 1 KiB pieces of a slice per wave
  constexpr int RPS = 64, NL = (WR + RPS - 1) / RPS;
  static_assert(PL % 256 == 0, "plane size keeps the fragment reads conflict-free");
  static_assert(WIN + NS * SLOT <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xw = lds;
  char* const Rg = lds + WIN;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;      // wm = the wave half: waves w and w + 4 share a SIMD

  // XCD-aware tile numbering: XCD k gets the k-th contiguous eighth of the (utterance, tile) sequence
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
  const int id = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int b = id / a.tiles, tile = id - b * a.tiles;
#ifdef PP_STAMPS
  int stamp_n = 0;
  const bool stamp_on = orig == nwg / 2 + 3;
  unsigned st_lo[3] = {0u, 0u, 0u}, st_hi[3] = {0u, 0u, 0u};
#endif
  PP_STAMP(1);

  // timing-only diagnostics (VSP_PP_DIAG, results WRONG): 1 no residual loads, 2 no stores, 4 no window loads, 8 no
  // hand-over work, 16 no MFMAs, 32 no window conversion / writes, 64 every weight slice from the same bytes, 128 no slice waits
  const int diag = a.terms >> 8;
  const int K = a.K, p2 = (K - 1) >> 1, p1 = a.dil * p2;
  const int R2 = BT - (K - 1);                  // output columns per block
  const int t0 = tile * R2;                     // first output column
  const int T = g16_len(a.glen, b, a.grate, a.T);   // (ragged batch: this utterance's own extent)
  if (t0 >= T) return;
  const int S1 = NCH * K, S = 2 * S1;
  const int xrows = BT + (K - 1) * a.dil;       // window rows conv1 reads

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, T * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.o_bs, 0, T * C * 4,
                                                                      0x00020000);

  // ---- x window staging: row 0 of the window is time t0 - p2 - p1; 16 consecutive lanes write 128 contiguous bytes of
  //      one plane.  Rows outside [0, T) are out of the descriptor's range: zeros = the convolution's padding.
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
      sv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (in && !(diag & 4)) ? st_voff + (base + u * RPS * C * 4) : G16_OOR, 0, 0);
    }
  };
  auto x_write = [&](int chunk) {
    char* dst0 = Xw + chunk * XBUF + st_loff;
    if (diag & 32) return;
    g16_for<NL>([&](auto U) {
      constexpr int u = decltype(U)::value;
      f16x4 eh, el;
      g16_split4(g16_as_f32x4(sv[u]), slope, true, eh, el);
      if ((u + 1) * RPS <= WR || row_s + u * RPS < WR) {       // (the last sweep overhangs the plane)
        *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16) = eh;
        *reinterpret_cast<f16x4*>(dst0 + u * RPS * 16 + XIMG) = el;
      }
    });
  };

  // ---- weight slices by LDS-DMA: global slice index gs in [0, S): conv = gs / S1, then chunk-major, one tap per slice:
  //      the slices of a convolution are consecutive 16 KiB runs of its packed image -- a running pointer (the scalar work
  //      of a MEM phase is on the ping-pong's critical path: no multiplies, no selects)
  const char* dsrc = reinterpret_cast<const char*>(a.w1h);      // (uniform; + d_lane: this lane's 16 bytes of the wave's piece)
  const unsigned d_lane = (wave * 64 + lane) * 16;
  int dleft = S1;                               // slices left in the convolution being requested
  auto dma_next = [&](int slot) {
    char* const dst = Rg + slot * SLOT + wave * 1024;
    const char* const src = (diag & 64) ? reinterpret_cast<const char*>(a.w1h) : dsrc;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + d_lane),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + NWV * 1024 + d_lane),
                                     (__attribute__((address_space(3))) void*)(dst + NWV * 1024), 16, 0, 0);
    dsrc += SLOT;
    if (--dleft == 0) { dleft = S1; dsrc = reinterpret_cast<const char*>(a.w2h); }
  };

  // ---- fragments: asm reads with immediate offsets off two per-step base addresses
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned xb_lane = lds0 + q4 * PL + (wn * NW * 16 + l15) * 16;
  const unsigned wa_lane = lds0 + WIN + lane * 16 + wm * MW * 2048;
  f16x8 Ah[MW], Al[MW], Bh[NW], Bl[NW];
  f32x4 hh[MW][NW];
  auto init_acc = [&](const float* bias) {
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 64 * wm + 16 * i + 4 * q4);
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = bv;
    }
  };

  // ================= prologue =================
  init_acc(a.b1);
  // conv2's bias now (a load at the hand-over would wait for the slices in flight as well)
  f32x4 bv2[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) bv2[i] = *reinterpret_cast<const f32x4*>(a.b2 + 64 * wm + 16 * i + 4 * q4);
  const unsigned ti_lane = lds0 + 2 * wm * XBUF + (q4 >> 1) * PL + (wn * NW * 16 + l15) * 16 + 8 * (q4 & 1);
  x_issue(0);
  dma_next(0);
  dma_next(1);
  PP_STAMP(2);                      // requests out
  x_write(0);                       // (the compiler waits for the window loads here)
  PP_STAMP(3);                      // first chunk converted and written
  int xl_a = 0, xl_b = 0;           // window loads issued behind the DMA of the previous / of this MEM phase
  x_issue(1);
  xl_b = 1;
  g16_vmcnt<NL>();                  // slices 0, 1 have landed; chunk 1's loads stay in flight
#pragma unroll
  for (int i = 0; i < MW; ++i) asm volatile("" ::"v"(bv2[i]));   // (retires the bias loads here, not inside the loops)
  PP_STAMP(4);
  G16_BARRIER();
  // (static priority for the second-dispatched half, no per-phase flips: gen16.hip, g16_conv)
  if (wm) { G16_BARRIER(); __builtin_amdgcn_s_setprio(1); }            // the second half runs one phase behind
  PP_STAMP(5);

  int slot = 0, gs = 0;
  // one convolution: S1 steps; rowstep = its dilation; STAGE: the remaining x chunks are staged on the way (conv1)
  auto run_conv = [&](auto CV, int rowstep) {
    constexpr int cv = decltype(CV)::value;
    int chunk = 0, tap = 0;
    for (int s = 0; s < S1; ++s, ++gs) {
      const bool last_tap = tap == K - 1;
      // ================= MEM phase =================
      {
        // (order: the B reads, the slice request and the staging decision, the A reads -- a wave issues one ds_read_b128
        // per ~19 clocks, the scalar work of the request fits between them instead of following them)
        const unsigned b_cur = xb_lane + chunk * XBUF + tap * rowstep * 16;
        const unsigned a_cur = wa_lane + slot * SLOT;
        g16_for<NW>([&](auto J) {
          constexpr int j = decltype(J)::value;
          Bh[j] = g16_lds_read<j * 256>(b_cur);
          Bl[j] = g16_lds_read<j * 256 + XIMG>(b_cur);
        });
        xl_a = xl_b;
        xl_b = 0;
        // slice gs + 2 goes into the slot slice gs - 1 left.  Staging steps (conv1, a chunk's last tap): the next chunk is
        // converted and written FIRST (hipcc drains vmcnt in front of LDS stores -- LDS-DMA may alias them --: at this
        // point only slice gs + 1 is in flight, which this phase waits for anyway), then the slice, then the window
        // loads of the chunk after (one straight-line block: a merge point in between costs another vmcnt(0))
        if (cv == 0 && last_tap && chunk + 1 < NCH) {
          x_write(chunk + 1);                                  // its region was never read: no hazard
          dma_next(slot == 0 ? NS - 1 : slot - 1);             // (gs + 2 < S always holds in conv1)
          if (chunk + 2 < NCH) { x_issue(chunk + 2); xl_b = 1; }
        } else if (gs + 2 < S) {
          dma_next(slot == 0 ? NS - 1 : slot - 1);
        }
        g16_for<MW>([&](auto I) {
          constexpr int i = decltype(I)::value;
          Ah[i] = g16_lds_read<i * 2048>(a_cur);
          Al[i] = g16_lds_read<i * 2048 + 1024>(a_cur);
        });
        // my pieces of slice gs + 1 have landed: issued after them are the previous phase's window loads, slice gs + 2
        // and this phase's window loads
        if (gs + 1 < S && !(diag & 128)) {
          if (gs + 2 < S) g16_vm_wait<NBW, NL>(true, xl_a + xl_b);
          else g16_vm_wait<0, NL>(false, xl_a + xl_b);
        }
        // (the second half's last MEM phase of conv2 keeps its barrier: the first half's last MFMA phase pairs with it)
        PP_STAMP(11);                                          // MEM work issued, slice wait done
        G16_BARRIER();
        PP_STAMP(12);
      }
      // ================= MFMA phase =================
      __builtin_amdgcn_sched_barrier(0);
      if (!(diag & 16))
      g16_for<NW>([&](auto J) {
        constexpr int j = decltype(J)::value;
        g16_for<MW>([&](auto I) {
          constexpr int i = decltype(I)::value;
          hh[i][j] = G16_MFMA(Ah[i], Bh[j], hh[i][j]);
          hh[i][j] = G16_MFMA(Al[i], Bh[j], hh[i][j]);
          hh[i][j] = G16_MFMA(Ah[i], Bl[j], hh[i][j]);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP(13);                                            // MFMAs issued
      // (conv2: the second half skips the barrier behind its LAST MFMA phase -- its epilogue would otherwise wait for
      // the first half's waves to end)
      if (cv == 0 || s + 1 < S1 || wm == 0) G16_BARRIER();
      PP_STAMP(14);
      slot = slot == NS - 1 ? 0 : slot + 1;
      if (last_tap) { tap = 0; ++chunk; } else ++tap;
    }
  };

  // ================= conv1 =================
  run_conv(std::integral_constant<int, 0>{}, a.dil);

  // ================= hand-over =================
  // Every window read of conv1 is done (the second half's last MEM phase ended with the barrier the first half has just
  // passed; the second half arrives one phase later): the conv1 tile -> activated, split -> t images over the window.
  // Channel 64 wm + 16 i + 4 q4 + e sits in chunk 2 wm + (i >> 1), plane 2 (i & 1) + (q4 >> 1), byte 8 (q4 & 1) + 2 e of
  // the row's 16.  Image row j = time t0 - p2 + j; rows outside the utterance are conv2's zero padding.
  PP_STAMP(20);
  if (!(diag & 8))
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int row = wn * NW * 16 + 16 * j + l15;
    const int tt = t0 - p2 + row;
    const float f = tt >= 0 && tt < T ? G16_UNSCALE : 0.f;   // (the unscaling and the zero padding of columns outside the utterance in ONE multiply: the factor is per column)
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const f32x4 v = hh[i][j] * f;
      f16x4 eh, el;
      g16_split4(v, slope, true, eh, el);
      // (asm stores: in front of a visible LDS store hipcc drains vmcnt -- the two conv2 slices in flight)
      const unsigned dst = ti_lane + (i >> 1) * XBUF + (2 * (i & 1)) * PL + j * 256;
      asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(eh) : "memory");
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(dst), "v"(el), "n"(XIMG) : "memory");
    }
  }
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) hh[i][j] = bv2[i];
  PP_STAMP(21);                     // t tiles written
  G16_BARRIER();
  G16_BARRIER();                    // both halves' tiles are written; the halves are one phase apart again
  PP_STAMP(22);

  // ================= conv2 =================
  run_conv(std::integral_constant<int, 1>{}, 1);

  // ---- epilogue: y = conv2 + x (+ previous resblock sum) (/ div); columns >= R2 belong to the next tile
  PP_STAMP(30);
#if PP_EPI_LDS
  // Round 5: through the LDS, so that every global access of the epilogue is 1 KiB CONTIGUOUS.  In the D-tile layout a lane
  // holds four channels of one time row: a b128 load / store of a wave is 16 rows x 64 B -- sixteen half-line segments per
  // instruction, and the stamps of round 4 showed the 12 stores of a wave taking 2.8-3.8 us to issue (one block per CU:
  // nothing hides it).  The conv2 tile (cross accumulator folded) goes to the dead window as [column][128 channels] fp32
  // (528-byte column stride: conflict-free 16-byte writes), then a wave owns whole column PAIRS: 32 lanes x 16 B = one
  // column's 512 B, two adjacent columns = 1 KiB of the channels-last tensor -- residual, previous sum and result alike.
  // Same arithmetic in the same order (acc * 2^-8 + x [+ previous] [/ div]): bit-identical.
  // Nobody reads the window any more: the first half has passed the barrier behind its last MFMA phase, which the second
  // half reached only after its last fragment reads.
  constexpr int ECS = C * 4 + 16;                 // column stride in the LDS tile
  static_assert(BT * ECS <= WIN, "the fp32 tile fits the dead window");
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      // (the accumulator as it is: the unscaling is folded into the residual's fma below)
      const int col = wn * NW * 16 + 16 * j + l15;
      *reinterpret_cast<f32x4*>(Xw + col * ECS + (64 * wm + 16 * i + 4 * q4) * 4) = hh[i][j];
    }
  G16_BARRIER();
  {
    constexpr int NPAIR = BT / 2 / NWV;           // column pairs per wave
    const int half = lane >> 5, l31 = lane & 31;
    int eo[NPAIR];
    u32x4 rv[NPAIR];
    [[maybe_unused]] u32x4 pv[NPAIR];
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
      const int col = 2 * (wave * NPAIR + u) + half;
      eo[u] = (col < R2 && !(diag & 2)) ? ((t0 + col) * C + 4 * l31) * 4 : G16_OOR;   // (rows at and beyond T: out of the descriptor's range)
      rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (diag & 1) ? G16_OOR : eo[u], 0, 0);
    }
    if (a.acc_prev) {
#pragma unroll
      for (int u = 0; u < NPAIR; ++u) pv[u] = __builtin_amdgcn_raw_buffer_load_b128(ro, eo[u], 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NPAIR; ++u) {
      const int col = 2 * (wave * NPAIR + u) + half;
      f32x4 v = *reinterpret_cast<const f32x4*>(Xw + col * ECS + l31 * 16);
      v = v * G16_UNSCALE + g16_as_f32x4(rv[u]);          // (exact product: the bits of unscale-then-add)
      if (a.acc_prev) v += g16_as_f32x4(pv[u]);
      g16_div(v, a.div);
      __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, eo[u], 0, 0);
    }
  }
#else
  int oo[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int r = wn * NW * 16 + 16 * j + l15;
      oo[i][j] = r < R2 ? ((t0 + r) * C + 64 * wm + 16 * i + 4 * q4) * 4 : G16_OOR;
    }
  u32x4 rv[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) rv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (diag & 1) ? G16_OOR : oo[i][j], 0, 0);
#ifdef PP_STAMPS
  PP_STAMP(31);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_STAMP(32);                     // residual operands here
#endif
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      hh[i][j] *= G16_UNSCALE;
      hh[i][j] += g16_as_f32x4(rv[i][j]);
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
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      f32x4 v = hh[i][j];
      g16_div(v, a.div);
      __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, (diag & 2) ? G16_OOR : oo[i][j], 0, 0);
    }
#endif
#ifdef PP_STAMPS
  PP_STAMP(33);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_STAMP(34);                     // stores retired
  if (stamp_on)
    for (int g = 0; g < 3; ++g)
      g_pp_stamps[wave][g * 64 + lane] = ((unsigned long long)st_hi[g] << 32) | st_lo[g];
#endif
}

template <int NW, int HALO>
static hipError_t launch_g16_pp_tile(ClPairArgs a, int B, hipStream_t s) {
  constexpr int BT = 64 * NW;
  constexpr size_t lds = (size_t)4 * 2 * 4 * (BT + HALO) * 16 + (size_t)3 * 8 * 2048;
  static std::atomic<uint64_t> attr_done{0};
  auto kern = g16_pp<NW, HALO>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  const int R2 = BT - (a.K - 1);
  a.tiles = (a.T + R2 - 1) / R2;
  const long n = (long)a.tiles * B;
  if (n <= 0 || n > 0x7fffffffL) return hipErrorInvalidValue;
  // (timing-only ablations, results WRONG by construction: experiment builds only)
#ifdef VSP_EXPERIMENTS
  static const int diag = []() { const char* e = getenv("VSP_PP_DIAG"); return e ? atoi(e) : 0; }();
  a.terms |= diag << 8;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(512), lds, s, a);
  return hipGetLastError();
}

// 128 channels, fp32-accurate products, kernel 3 (window halo <= 16 rows) and kernel 7 (<= 32 rows).  The generator
// asks per context (VSP_PP=0: the two-launch path, second implementation, bit-identical).
bool g16_pp_supported(int C, int K, int dil, int terms) {
  return C == 128 && terms == 3 && dil >= 1 && ((K == 3 && 2 * dil <= 16) || (K == 7 && 6 * dil <= 32));
}

hipError_t launch_g16_pp(const ClPairArgs& a, int B, hipStream_t s) {
  if (!g16_pp_supported(a.C, a.K, a.dil, a.terms) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out ||
      (long)a.T * a.C * 4 >= (1L << 31))
    return hipErrorInvalidValue;
  return a.K == 3 ? launch_g16_pp_tile<3, 16>(a, B, s) : launch_g16_pp_tile<3, 32>(a, B, s);
}

}  // namespace vsp
