// g16_convp: the 128 / 256-channel ResBlock convolutions as PERSISTENT blocks whose pipeline runs ACROSS tiles (round 4).
// Used for the tiles of few steps (g16_pipe_supported below); VSP_G16_PIPE=0 / 1 = never / always, all bit-identical
// (tests/test_cl_ops.py runs the forced form in a child process).  Measurements: profiles/r04_g16_conv_persistent_blocks.txt.
//
// The idea.  g16_conv (gen16.hip) reaches 86 % of the power-limited MFMA rate inside a tile's step loop, but a tile's time is
// steps x 0.94-1.0 us PLUS 12-17 us (128 channels) / 19 us (256 channels) -- fitted over K = 3 / 7 / 11 --, and with one
// block per CU (147 KB of LDS) nothing overlaps that constant.  Here one block per CU walks a contiguous run of tiles and
// NOTHING restarts at a tile boundary:
//   * the weight ring keeps streaming: slice g + 3 is requested in step g whatever tile it belongs to (the next tile's
//     first slices are in flight during this tile's last steps);
//   * the window stream keeps streaming: the input is a producer-written operand image (ClConvArgs::x_img; g16_conv's XIN
//     form), so a chunk's window is NL LDS-DMA pieces per wave with no registers and no conversion -- chunk c + 1 is
//     requested at chunk c's first tap, and "chunk c + 1" may be the next tile's chunk 0;
//   * the ping-pong of the two wave halves never stops: a tile's epilogue (cross accumulators folded, residual, stores,
//     operand image of the result) sits at the top of the MEM phase of the NEXT tile's first step, so the first half's
//     epilogue runs beside the second half's last MFMA phase and vice versa; accumulators restart at the bias.
// What it took (same box, per launch, second convolutions of the pairs at the C3 size).  First version: 12-20 % SLOWER
// than one block per tile (1.78 -> 2.03 ms at 128 channels K = 11): the MEM phase of a step is the critical path of the
// ping-pong (0.48 us against 0.46 us of MFMAs), and run-time wait counts, three carried stream cursors and 28 spilled
// scalar registers read back by v_readlane sat in it.  Immediates for the steady-state waits, the epilogue's parameters
// re-read from the kernel-argument segment, request cursors derived from the compute cursor: equal within 2 %.  Then the
// SECOND wave half's epilogue moved to the end of its last MFMA phase -- the phase in which the first half runs its own
// at the top of the next tile's first MEM phase -- so the two epilogues overlap instead of following each other: K = 3
// -12 % (128 channels) / -5 % (256), K = 7 -2 % / +1 %, K = 11 -1 % / +6 %.  Starting blocks staggered (de-synchronised
// epilogues, across or within XCDs): worse.  What still sits at a tile boundary is one epilogue's length (16 tiles per wave:
// fold, addresses, residual round trip, 16-48 stores), and `vmcnt` retires a wave's loads, stores and LDS-DMA in ONE order:
// every wait for a slice requested after the stores is also a wait for the stores.
// Same arithmetic per output as g16_conv (chunk-major, tap-minor, HH / CROSS / CROSS, bias in the accumulator): results
// are bit-identical.
//
// Counted waits.  vmcnt retires loads, stores and LDS-DMA in issue order.  In step g a wave waits for ITS pieces of slice
// g + 1 (requested in step g - 2); younger than those are slices g + 2, g + 3 and the window groups of this and the
// previous phase -- and, in the first two steps of a tile, the previous tile's epilogue operations issued at the top of
// step 0.  With a residual operand the epilogue itself waits vmcnt(0) for its loads (the youngest operations at that
// point), which implies every older slice: those two steps then need no wait at all; without one the epilogue's stores
// are counted in.  The counts are run-time values here (a 64-way switch over s_waitcnt immediates).
//
// Reference call sites: modules.py:210-223 (ResBlock1).
#include "g16_common.h"

#include <cstdlib>

namespace vsp {

__device__ __forceinline__ void g16_vmcnt_rt(int n) {
  n = n < 0 ? 0 : (n > 63 ? 63 : n);
#define G16_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    G16_VMC(0) G16_VMC(1) G16_VMC(2) G16_VMC(3) G16_VMC(4) G16_VMC(5) G16_VMC(6) G16_VMC(7) G16_VMC(8) G16_VMC(9)
    G16_VMC(10) G16_VMC(11) G16_VMC(12) G16_VMC(13) G16_VMC(14) G16_VMC(15) G16_VMC(16) G16_VMC(17) G16_VMC(18) G16_VMC(19)
    G16_VMC(20) G16_VMC(21) G16_VMC(22) G16_VMC(23) G16_VMC(24) G16_VMC(25) G16_VMC(26) G16_VMC(27) G16_VMC(28) G16_VMC(29)
    G16_VMC(30) G16_VMC(31) G16_VMC(32) G16_VMC(33) G16_VMC(34) G16_VMC(35) G16_VMC(36) G16_VMC(37) G16_VMC(38) G16_VMC(39)
    G16_VMC(40) G16_VMC(41) G16_VMC(42) G16_VMC(43) G16_VMC(44) G16_VMC(45) G16_VMC(46) G16_VMC(47) G16_VMC(48) G16_VMC(49)
    G16_VMC(50) G16_VMC(51) G16_VMC(52) G16_VMC(53) G16_VMC(54) G16_VMC(55) G16_VMC(56) G16_VMC(57) G16_VMC(58) G16_VMC(59)
    G16_VMC(60) G16_VMC(61) G16_VMC(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef G16_VMC
}

struct G16Tile { int cb, bx, b; };   // row group, time tile, utterance

template <int MW, int NW, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) g16_convp(ClConvArgs a) {
  constexpr int NWV = WM * WN, NTH = 64 * NWV;
  constexpr int BT = 16 * NW * WN;              // time columns per tile
  constexpr int MTB = MW * WM;                  // m-tiles per tile
  constexpr int WR = BT + G16_HALO;             // window rows allocated
  constexpr int PL = WR * 16, XIMG = 4 * PL, XBUF = 2 * XIMG;
  constexpr int SLOT = MTB * 2048;              // bytes per ring slot = one (chunk, tap) slice of the tile's rows
  constexpr int NS = 4;
  constexpr int NL = WR / 64;                   // window pieces (64 rows of one plane) per wave and chunk
  constexpr int NBLK = 2 * MTB;                 // 1 KiB pieces per slice
  constexpr int NBW = (NBLK + NWV - 1) / NWV;   // per wave
  static_assert(NWV == 8 && WR % 64 == 0 && PL % 256 == 0 && NW % 2 == 0 && NTH == 512, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xw = lds;
  char* const Rg = lds + 2 * XBUF;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int nmt = a.Cout >> 4, nch = a.Cin >> 5;
  const int K = a.K, S = nch * K;
  const int gy = a.pt_gy, gx = a.pt_gx;

  // ---- this block's run of tiles in the (utterance, time tile, row group) sequence, row group fastest
  const int nblk = gridDim.x, per = a.pt_total / nblk, extra = a.pt_total - per * nblk;
  const int id_lo = blockIdx.x * per + ((int)blockIdx.x < extra ? (int)blockIdx.x : extra);
  const int ntl = per + ((int)blockIdx.x < extra ? 1 : 0);
  if (ntl <= 0) return;
  const int total_steps = ntl * S, total_chunks = ntl * nch;
  auto tile_at = [&](int id) { return G16Tile{id % gy, (id / gy) % gx, id / (gy * gx)}; };
  auto tile_next = [&](G16Tile t) {
    if (++t.cb == gy) { t.cb = 0; if (++t.bx == gx) { t.bx = 0; ++t.b; } }
    return t;
  };
  auto tile_prev = [&](G16Tile t) {
    if (--t.cb < 0) { t.cb = gy - 1; if (--t.bx < 0) { t.bx = gx - 1; --t.b; } }
    return t;
  };
  G16Tile tc = tile_at(id_lo);                 // tile being multiplied
  // (the request cursors are DERIVED from the compute cursor -- the window stream runs one chunk ahead, the slice stream
  // three steps -- instead of being carried: every scalar carried around the step loop beyond ~90 is spilled to vector
  // lanes and read back in the MEM phase, the critical path of the ping-pong)

  // ---- window chunks by LDS-DMA (as g16_conv<.., XIN>): wave w copies (image w / 4, plane w % 4)
  auto xi_issue = [&](G16Tile tw, int cw) {
    const int img = wave >> 2, plane = wave & 3;
    const size_t row0 = (((size_t)cw * 2 + img) * 4 + plane) * a.xi_tpad + (size_t)(G16_IMG_PADF + tw.bx * BT - a.pad);
    const uint4* gp0 = reinterpret_cast<const uint4*>(a.x_img + (size_t)tw.b * a.xi_bs) + row0 + lane;
    char* lp0 = Xw + (cw & 1) * XBUF + img * XIMG + plane * PL;     // (chunk parity = stream parity: nch is even)
#pragma unroll
    for (int u = 0; u < NL; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp0 + u * 64),
                                       (__attribute__((address_space(3))) void*)(lp0 + u * 1024), 16, 0, 0);
  };
  // ---- weight slices by LDS-DMA: one endless sequence over the block's tiles; (dc, dt) = the next slice to request,
  //      cb_r = the row group of its tile
  const uint4* Wg = reinterpret_cast<const uint4*>(a.wh);
  int dc = 0, dt = 0;
  auto dma_next = [&](int slot, int cb_r) {
    const size_t src = (((size_t)dc * K + dt) * nmt + (size_t)cb_r * MTB) * 128;   // uint4 units (2 KiB per m-tile)
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
      const int blk = u * NWV + wave;
      if (NBLK % NWV == 0 || blk < NBLK) {
        const uint4* gp = Wg + src + (size_t)blk * 64 + lane;
        char* lp = Rg + slot * SLOT + blk * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
      }
    }
    if (++dt == K) { dt = 0; if (++dc == nch) dc = 0; }
  };
  const bool has_pieces = NBLK % NWV == 0 || wave < NBLK;
  const int my_pieces = has_pieces ? NBW : 0;   // (NBLK < NWV: at most one piece per wave, NBW == 1)

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned xb_lane = lds0 + (lane >> 4) * PL + (wn * NW * 16 + (lane & 15)) * 16;
  const unsigned wa_lane = lds0 + 2 * XBUF + lane * 16 + wm * MW * 2048;
  f16x8 Ah[MW], Al[MW], Bh[NW], Bl[NW];
  f32x4 hh[MW][NW];
  auto acc_init = [&](int cb) {
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int row = ((cb * MTB + wm * MW + i) << 4) + 4 * (lane >> 4);
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + row);
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = bv;
    }
  };

  // ---- a tile's epilogue: lane = 4 consecutive channels of one time column (16-byte accesses)
  const bool res_type = a.res != nullptr || a.acc_prev != 0;      // the epilogue waits for loads (vmcnt(0))
  const int epi_stores = (a.out ? MW * NW : 0) + (a.o_img ? 2 * MW * NW : 0);
  auto epilogue = [&](G16Tile t) {
    // (the epilogue's parameters are re-read from the kernel-argument segment through a pointer the compiler cannot see
    // through: kept in scalar registers across the step loop they were spilled to vector lanes and read back in it)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ClConvArgs* KArgs;
    KArgs ea = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ea));
    const ClConvArgs a = *ea;
#endif
    const int t0 = t.bx * BT;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        a.out ? a.out + (size_t)t.b * a.o_bs : nullptr, 0, a.out ? a.T_store * a.o_ts * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(
        a.res ? const_cast<float*>(a.res) + (size_t)t.b * a.r_bs : nullptr, 0, a.res ? a.T_store * a.r_ts * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(
        a.o_img ? a.o_img + (size_t)t.b * a.oi_bs : nullptr, 0, a.o_img ? a.Cout * 4 * a.oi_tpad : 0, 0x00020000);
    int lane_e = lane;                                  // (addresses derived here, not hoisted out of the step loop)
    asm volatile("" : "+v"(lane_e));
    int oo[MW][NW], orr[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int co = ((t.cb * MTB + wm * MW + i) << 4) + 4 * (lane_e >> 4);
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int tt = t0 + (wn * NW + j) * 16 + (lane_e & 15);
        const bool in_t = tt < a.Nq;
        oo[i][j] = in_t ? tt * a.o_ts * 4 + co * 4 : G16_OOR;
        orr[i][j] = in_t ? tt * a.r_ts * 4 + co * 4 : G16_OOR;
      }
    }
    // (requested unconditionally: an empty descriptor returns zeros at once -- registers written under a condition inside
    // the loop would stay live around it)
    u32x4 rv[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) rv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rr_, orr[i][j], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (res_type) g16_vmcnt<0>();     // (what the skipped slice waits of the tile's first two steps rely on)
    // (acc * G16_UNSCALE is exact: unscaling and residual are ONE fma; without a residual the loads above returned zeros)
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
      for (int j = 0; j < NW; ++j) hh[i][j] = hh[i][j] * G16_UNSCALE + g16_as_f32x4(rv[i][j]);
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
        if (a.out) __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v), ro, oo[i][j], 0, 0);
        if (a.o_img) {
          const int co = ((t.cb * MTB + wm * MW + i) << 4) + 4 * (lane_e >> 4);
          const int tt = t0 + (wn * NW + j) * 16 + (lane_e & 15);
          const int unit = ((((co >> 5) * 2) * 4 + ((co & 31) >> 3)) * a.oi_tpad + G16_IMG_PADF + tt) * 16 + 2 * (co & 7);
          f16x4 eh, el;
          g16_split4(v, a.oi_slope, true, eh, el);
          const int off = tt < a.Nq ? unit : G16_OOR;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, eh), ri, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, el), ri, off, 4 * a.oi_tpad * 16, 0);
        }
      }
  };

  // ---- prologue: the stream's first two window chunks (both buffers are free), slices 0 .. 2
  acc_init(tc.cb);
  xi_issue(tc, 0);
  if (total_chunks > 1) xi_issue(tc, 1);      // (nch >= 2: the stream's second chunk is this tile's)
  dma_next(0, tc.cb);                         // (S >= 6: the first three slices are this tile's)
  dma_next(1, tc.cb);
  dma_next(2, tc.cb);
  g16_vmcnt<0>();
  G16_BARRIER();
  if (wave >= NWV / 2) G16_BARRIER();            // the second wave half runs one phase behind the first

  int xl_a = 0, xl_b = 0;                        // window groups requested in the previous / in this MEM phase
  int chunk = 0, tap = 0, slot = 0, cgi = 0;     // of the current step; cgi = chunk index in the block's stream
  int since_epi = 2;                             // MEM phases since this wave's last epilogue (>= 2: none pending)
  const bool first_half = wave < NWV / 2;
  for (int gs = 0; gs < total_steps; ++gs) {
    // ================= MEM phase of step gs =================
    // The previous tile is complete: its epilogue, then a fresh accumulator.  FIRST wave half: here, at the top of the next
    // tile's first MEM phase (beside the second half's last MFMA phase).  SECOND half: at the END of its last MFMA phase
    // (below) -- the same phase --, so the two halves' epilogues overlap instead of following each other.
    if (first_half && chunk == 0 && tap == 0 && gs > 0) {
      epilogue(tile_prev(tc));
      acc_init(tc.cb);
      since_epi = 0;
    }
    {
      const unsigned b_cur = xb_lane + (chunk & 1) * XBUF + tap * a.dil * 16;
      const unsigned a_cur = wa_lane + slot * SLOT;
      g16_for<NW>([&](auto J) {
        constexpr int j = decltype(J)::value;
        Bh[j] = g16_lds_read<j * 256>(b_cur);
        Bl[j] = g16_lds_read<j * 256 + XIMG>(b_cur);
      });
      g16_for<MW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        Ah[i] = g16_lds_read<i * 2048>(a_cur);
        Al[i] = g16_lds_read<i * 2048 + 1024>(a_cur);
      });
      xl_a = xl_b;
      xl_b = 0;
      // the buffer of the stream's previous chunk is free from this chunk's first tap on: the next chunk -- maybe the next
      // tile's first -- is requested into it K steps before its first use
      if (tap == 0 && cgi >= 1 && cgi + 1 < total_chunks) {
        if (chunk + 1 < nch) xi_issue(tc, chunk + 1); else xi_issue(tile_next(tc), 0);
        xl_b = 1;
      }
      if (gs + 3 < total_steps)                             // slice gs + 3 into the slot slice gs - 1 left
        dma_next(slot == 0 ? NS - 1 : slot - 1, chunk * K + tap + 3 >= S ? (tc.cb + 1 == gy ? 0 : tc.cb + 1) : tc.cb);
      if (gs + 1 < total_steps) {
        // my pieces of slice gs + 1 (requested in step gs - 2) have landed: younger are slices gs + 2, gs + 3, the window
        // groups of this and the previous phase and, in a tile's first two steps, the previous tile's epilogue stores
        const int behind = (gs + 2 < total_steps ? 1 : 0) + (gs + 3 < total_steps ? 1 : 0);
        if (since_epi >= 2) {                               // the steady state: immediates, as in g16_conv
          if (!has_pieces || behind == 0) g16_vm_wait<0, NL>(false, xl_a + xl_b);
          else if (behind == 1) g16_vm_wait<NBW, NL>(true, xl_a + xl_b);
          else g16_vm_wait<2 * NBW, NL>(true, xl_a + xl_b);
        } else if (!res_type) {
          // (+ MW * NW: the epilogue's residual-operand loads are issued unconditionally -- on the empty descriptor here --
          // and sit in the same queue, younger than the slice waited for)
          g16_vmcnt_rt(behind * my_pieces + (xl_a + xl_b) * NL + epi_stores + MW * NW);
        }
        // (else: the epilogue waited vmcnt(0) for its operands: every slice requested before it has landed)
      }
      G16_BARRIER();                                        // (lgkmcnt(0): the fragments are here)
    }
    // ================= MFMA phase of step gs =================
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    g16_for<NW>([&](auto J) {
      constexpr int j = decltype(J)::value;
      g16_for<MW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        hh[i][j] = G16_MFMA(Ah[i], Bh[j], hh[i][j]);
        hh[i][j] = G16_MFMA(Al[i], Bh[j], hh[i][j]);
        hh[i][j] = G16_MFMA(Ah[i], Bl[j], hh[i][j]);
      });
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    ++since_epi;
    slot = slot == NS - 1 ? 0 : slot + 1;
    if (++tap == K) {
      tap = 0;
      ++cgi;
      if (++chunk == nch) {
        chunk = 0; tc = tile_next(tc);
        if (!first_half && gs + 1 < total_steps) {          // second half: the finished tile's epilogue in this very phase
          epilogue(tile_prev(tc));
          acc_init(tc.cb);
          since_epi = 0;
        }
      }
    }
    G16_BARRIER();
  }
  // (the first wave half has executed one barrier fewer: it goes straight to the last tile's epilogue, which overlaps the
  // second half's last MFMA phase; a wave that has ended no longer takes part in the barrier)
  epilogue(tile_prev(tc));
}

template <int MW, int NW, int WM, int WN>
static hipError_t launch_g16_pipe_tile(ClConvArgs a, int B, hipStream_t s) {
  constexpr int BT = 16 * NW * WN, MTB = MW * WM;
  constexpr size_t lds = (size_t)2 * 2 * 4 * (BT + G16_HALO) * 16 + (size_t)4 * MTB * 2048;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = g16_convp<MW, NW, WM, WN>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  const int nmt = a.Cout / 16;
  if (nmt % MTB || a.Cin % 64) return hipErrorInvalidValue;            // (an even number of 32-channel chunks)
  a.pt_gx = (a.Nq + BT - 1) / BT; a.pt_gy = nmt / MTB;
  const long total = (long)a.pt_gx * a.pt_gy * B;
  if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
  a.pt_total = (int)total;
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  if (cus <= 0) cus = 256;
  const long resident = (long)cus * ((160 * 1024) / (long)lds);
  const long nblocks = total < resident ? total : resident;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}

// Where it pays (same box, per launch, second convolutions of the pairs at the C3 size; profiles/r04_g16_conv_persistent_blocks.txt):
// tiles of few steps, where the boundary is a large share -- 128 channels K = 3 -12 %, K = 7 -2 %, 256 channels K = 3 -5 %;
// at 44-88 steps per tile (K = 11, 256 channels K = 7) the leaner step of the one-block-per-tile kernel wins by 1-6 %.
// Default: tiles of at most 28 steps.  VSP_G16_PIPE=0: never, =1: always (both bit-identical to the default).
bool g16_pipe_supported(const ClConvArgs& a) {
  static const int mode = []() { const char* e = getenv("VSP_G16_PIPE"); return e ? atoi(e) : -1; }();
  if (mode == 0) return false;
  if (!(a.x_img && a.terms == 3 && a.phases == 1 && a.K >= 3 && a.pad <= CL_IMG_PADF && a.Cin % 64 == 0 &&
        a.Cout % 128 == 0 && (a.K - 1) * a.dil <= G16_HALO))
    return false;
  return mode == 1 || (a.Cin / 32) * a.K <= 28;
}

// (the 128-row x 256-column tile: what launch_g16_conv picks when the launch fills the chip)
hipError_t launch_g16_pipe(const ClConvArgs& a, int B, hipStream_t s) {
  if (!g16_pipe_supported(a) || a.Nq <= 0 || B <= 0 || (!a.out && !a.o_img)) return hipErrorInvalidValue;
  return launch_g16_pipe_tile<4, 4, 2, 4>(a, B, s);
}

}  // namespace vsp
