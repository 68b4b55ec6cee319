// g16_rw: fused ResBlock1 conv PAIR of the 32-channel stage (reference modules.py:210-223) with the WEIGHTS IN REGISTERS
// (round 4).  Same arithmetic per output as g16_pair / g16_conv (tap-minor, HH / CROSS / CROSS per step, bias in the
// accumulator, acc * 2^-8 + x) -- bit-identical results -- but none of the ring kernel's hand-shakes:
//
//   * a 32-channel convolution's whole weight image is K x 2 m-tiles x (hi | lo) A fragments = 16 K registers per lane
//     (112 at K = 7, 176 at K = 11).  Waves 0-3 of a block hold conv1's, waves 4-7 conv2's, for the block's whole life:
//     no weight slices, no LDS-DMA ring, no per-slice barrier / counted wait (34 % of a ring block's life by the stamps of
//     profiles/r03_pair_kernel_phase_stamps_two_slot_ring.txt), and the A operands never touch the LDS: a tap's six
//     MFMAs per 16 columns read two B fragments (341 B of LDS per MFMA instead of 683).
//   * PERSISTENT blocks, one per CU, software-pipelined over their run of 192-column tiles: in iteration i the conv1
//     waves multiply tile i from the x image XW[i & 1] and write their activated, split result as the t image TI[i & 1];
//     the conv2 waves multiply tile i - 1 from TI[(i - 1) & 1], add the residual and store.  ONE barrier per tile.
//   * the fp32 window of a tile arrives by LDS-DMA in a staging area, TWO tiles ahead: every wave owns four 1 KiB pieces
//     of it, splits them into XW[(i + 1) & 1] (leaky-relu, hi / lo) and re-requests them for tile i + 2 right away -- the
//     conv2 waves at the top of an iteration, the conv1 waves at the bottom, so that the two waves of a SIMD (w and
//     w + 4: one of each role) are out of step: one's vector work falls into the other's MFMAs.
//   * the conv2 waves fetch their residual (and the previous ResBlock sum) one 16-column group ahead into registers.
//
// What the in-kernel stamps say bounds it (profiles/r04_rw_*): vector issue slots.  Per tile and SIMD the kernel issues
// 252 MFMAs (K = 7) and ~570 vector instructions (two operand splits per element are 9 of them per value, and nothing in
// the ISA makes them cheaper); beside a saturated MFMA stream a wave's vector instructions retire at about one per MFMA.
//
// LDS: XW 2 x 32 KB, TI 2 x 26 KB, staging 32 KB = 148 KB: one block of 8 waves per CU, 2 waves per SIMD, 256 registers.
#include "g16_common.h"

#include <cstdlib>

namespace vsp {

namespace {
constexpr int RW_BT = 192;                 // conv1 columns per tile
constexpr int RW_CW = RW_BT / 4;           // columns per role wave
constexpr int RW_G = RW_CW / 16;           // 16-column groups per role wave and tile
constexpr int RW_WRX = RW_BT + G16_HALO;   // x image rows allocated (256)
constexpr int RW_PL = RW_WRX * 16, RW_XIMG = 4 * RW_PL, RW_XBUF = 2 * RW_XIMG;
constexpr int RW_WRT = RW_BT + 16;         // t image rows allocated (conv2 reads K - 1 <= 12 rows past the tile)
constexpr int RW_PLT = RW_WRT * 16, RW_TIMG = 4 * RW_PLT, RW_TBUF = 2 * RW_TIMG;
constexpr int RW_STG = RW_WRX * 128;       // fp32 staging: rows of 32 floats
constexpr int RW_LDS_TILES = 2 * RW_XBUF + 2 * RW_TBUF + RW_STG;
constexpr int RW_MAXHALO = 56;                          // (K - 1) dil: the window never reaches the staging area's last 8 rows,
constexpr int RW_BIAS = RW_LDS_TILES - 256;             // where the two biases sit (2 x 128 B)
constexpr int RW_LDS = RW_LDS_TILES;
static_assert(RW_LDS <= 160 * 1024, "LDS budget");
static_assert(RW_PL % 256 == 0 && RW_PLT % 256 == 0, "plane sizes keep the fragment reads conflict-free");
}  // namespace

// RW_STAMPS (diagnostic build, tools/stamps_rw.py): lane 0 of every wave of the middle block records tagged wall-clock
// stamps (s_memrealtime, 100 MHz) over iterations 20 .. 23.
#ifdef RW_STAMPS
constexpr int RW_NSTAMP = 64;
__device__ unsigned long long g_rw_stamps[8][RW_NSTAMP];
#define RW_STAMP(tag)                                                                                         \
  do {                                                                                                        \
    if (stamp_on && stamp_n < RW_NSTAMP && (tid & 63) == 0)                                                   \
      g_rw_stamps[wave][stamp_n] = (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull) | ((unsigned long long)(tag) << 56); \
    if (stamp_on) ++stamp_n;                                                                                  \
  } while (0)
extern "C" int vsp_debug_stamps_rw(unsigned long long* host) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rw_stamps), sizeof(unsigned long long) * 8 * RW_NSTAMP);
}
#else
#define RW_STAMP(tag) ((void)0)
#endif

template <int K, bool ACC>
__global__ void __launch_bounds__(512) g16_rw(ClPairArgs a, int total_tiles) {
  constexpr int p2 = (K - 1) >> 1, R2 = RW_BT - (K - 1);
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const XW = lds;
  char* const TI = lds + 2 * RW_XBUF;
  char* const STG = TI + 2 * RW_TBUF;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  const int tid = threadIdx.x, lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is1 = wave < 4;               // conv1 waves 0-3, conv2 waves 4-7 (w and w + 4 share a SIMD)
  const int wr = wave & 3;

  // this block's run of tiles in the (utterance, tile) sequence; ragged batch (ClPairArgs::glen): utterance b has
  // ceil(glen[b] * grate / R2) tiles instead of a.tiles
  const int nb = gridDim.x, bid = blockIdx.x;
  int lo_b, lo_tile, n;
  if (a.glen) {
    g16_ragged_run(a.glen, a.B, a.grate, R2, nb, bid, lo_b, lo_tile, n);
  } else {
    const int per = total_tiles / nb, extra = total_tiles - per * nb;
    const int lo = bid * per + (bid < extra ? bid : extra);
    n = per + (bid < extra ? 1 : 0);
    lo_b = lo / a.tiles;
    lo_tile = lo - lo_b * a.tiles;
  }
  const int p1 = a.dil * p2;
  const int xrows = RW_BT + (K - 1) * a.dil;

  // ---- the role's weights: A fragments of the packed image [tap][m-tile][hi | lo][lane][8 halfs], and its bias
  const uint16_t* const wsrc = is1 ? a.w1h : a.w2h;
  f16x8 Wh[K][2], Wl[K][2];
#pragma unroll
  for (int tap = 0; tap < K; ++tap)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const size_t blk = ((size_t)tap * 2 + mt) * 2;
      Wh[tap][mt] = *reinterpret_cast<const f16x8*>(wsrc + (blk * 64 + lane) * 8);
      Wl[tap][mt] = *reinterpret_cast<const f16x8*>(wsrc + ((blk + 1) * 64 + lane) * 8);
    }
  // The weights are first USED inside the persistent loop: left at that, hipcc's wait-count pass puts the wait for these
  // loads -- a vmcnt(0), the loop's own LDS-DMA and stores being in the same queue -- in front of the first MFMA of EVERY
  // iteration, which exposes the whole latency of the window request.  A use here retires the loads before the loop.
#pragma unroll
  for (int tap = 0; tap < K; ++tap)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      asm volatile("" ::"v"(Wh[tap][mt]));
      asm volatile("" ::"v"(Wl[tap][mt]));
    }
  if (wr == 0 && lane < 32) reinterpret_cast<float*>(lds + RW_BIAS)[(is1 ? 0 : 32) + lane] = (is1 ? a.b1 : a.b2)[lane];
  const float slope = a.slope;
#ifdef RW_STAMPS
  int stamp_n = 0;
  bool stamp_on = false;
#endif
  // timing-only diagnostics (VSP_RW_DIAG, results WRONG): 1 no residual loads, 2 no stores, 4 no window DMA / split,
  // 8 no t-image writes
  const int diag = a.terms >> 8;

  // ---- one 16-column group: K taps x (2 B fragments from LDS, 6 MFMAs).  B double-buffered, a tap's reads requested a
  //      whole tap ahead (same box, K = 11: 1.87 ms per launch against 2.03 for the in-place form below).  Only the
  //      K = 11 kernel that also carries the previous ResBlock sum (176 weight registers + 16 operand registers) keeps ONE
  //      B set refilled in place -- the high image right after its four MFMAs have issued, the low image after its two --
  //      behind counted waits (LDS returns in order).
  //      A group never requests its own first fragments: prime() does, at the top of the iteration for the tile's first
  //      group and from inside the previous group's LAST tap otherwise, so that the bias (kept in LDS, not in 8 registers)
  //      and the first B fragments arrive under the previous group's vector work.
  //      baddr = LDS byte address of this lane's tap-0 fragment (plane q4, row column + l15), step = bytes per tap.
  constexpr bool INPLACE = K >= 11 && ACC;
  const unsigned bias_a = lds0 + RW_BIAS + (is1 ? 0 : 128) + q4 * 16;
  f16x8 nBh, nBl;                                            // primed: first fragments and bias of the next group
  f32x4 nh0, nh1;
  auto prime_hi = [&](unsigned baddr) {
    if constexpr (!INPLACE) {                                // (<11, ACC> has no registers for the bias before the group starts)
      nh0 = __builtin_bit_cast(f32x4, g16_lds_read<0>(bias_a));
      nh1 = __builtin_bit_cast(f32x4, g16_lds_read<64>(bias_a));
    }
    nBh = g16_lds_read<0>(baddr);
  };
  auto conv_group = [&](auto IMG, auto LAST, unsigned baddr, unsigned step, unsigned bnext, f32x4& hh0, f32x4& hh1) {
    constexpr int img = decltype(IMG)::value;
    constexpr bool last = decltype(LAST)::value;
    if constexpr (INPLACE) {
      hh0 = __builtin_bit_cast(f32x4, g16_lds_read<0>(bias_a));
      hh1 = __builtin_bit_cast(f32x4, g16_lds_read<64>(bias_a));
    } else {
      hh0 = nh0; hh1 = nh1;
    }
    __builtin_amdgcn_s_setprio(1);                           // (the MFMA stream wins the SIMD's issue port; the partner's vector
                                                             //  work fills the gaps the matrix pipe leaves)
    if constexpr (!INPLACE) {
      f16x8 Bh[2], Bl[2];
      Bh[0] = nBh; Bl[0] = nBl;
      g16_for<K>([&](auto T) {
        constexpr int tap = decltype(T)::value, cur = tap & 1;
        if constexpr (tap + 1 < K) {
          const unsigned an = baddr + (tap + 1) * step;
          Bh[cur ^ 1] = g16_lds_read<0>(an);
          Bl[cur ^ 1] = g16_lds_read<img>(an);
          g16_lgkmcnt<2>();
        } else if constexpr (!last) {
          prime_hi(bnext);
          nBl = g16_lds_read<img>(bnext);
          g16_lgkmcnt<4>();
        } else {
          g16_lgkmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        hh0 = G16_MFMA(Wh[tap][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wh[tap][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wl[tap][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wl[tap][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wh[tap][0], Bl[cur], hh0);
        hh1 = G16_MFMA(Wh[tap][1], Bl[cur], hh1);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      f16x8 Bh = nBh, Bl = nBl;
      g16_for<K>([&](auto T) {
        constexpr int tap = decltype(T)::value;
        constexpr bool more = tap + 1 < K;
        // outstanding, in issue order: Bh(tap), Bl(tap); at tap 0 also the bias, requested last
        g16_lgkmcnt<tap == 0 ? 0 : 1>();
        __builtin_amdgcn_sched_barrier(0);
        hh0 = G16_MFMA(Wh[tap][0], Bh, hh0);
        hh1 = G16_MFMA(Wh[tap][1], Bh, hh1);
        hh0 = G16_MFMA(Wl[tap][0], Bh, hh0);
        hh1 = G16_MFMA(Wl[tap][1], Bh, hh1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) Bh = g16_lds_read<0>(baddr + (tap + 1) * step);
        else if constexpr (!last) { prime_hi(bnext); Bh = nBh; }
        g16_lgkmcnt<(more || !last) ? 1 : 0>();              // the low image of this tap: older than the read just requested
        __builtin_amdgcn_sched_barrier(0);
        hh0 = G16_MFMA(Wh[tap][0], Bl, hh0);
        hh1 = G16_MFMA(Wh[tap][1], Bl, hh1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) Bl = g16_lds_read<img>(baddr + (tap + 1) * step);
        else if constexpr (!last) Bl = g16_lds_read<img>(bnext);
      });
      nBh = Bh; nBl = Bl;                                    // (the next group's, when this one was not the last)
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- x window of a tile: fp32 rows by LDS-DMA into the staging area (wave w: the 1 KiB pieces w, w + 8, w + 16,
  //      w + 24 of 8 rows each), split into an x image by the wave that requested them.  Window row r is time
  //      t0 - p2 - p1 + r; rows outside the utterance are fetched from a clamped address and zeroed at the split (the
  //      reference's zero padding).
  // (b, t0) of a tile: one integer division per BLOCK -- the loop carries the coordinates of tiles i - 1 .. i + 2 and
  // steps them with scalar compares (a per-iteration id / tiles costs ~20 vector instructions a call, and vector issue slots
  // are what this kernel is short of)
  // a tile: first output column, and utterance | extent << 8 in ONE scalar (B <= 256 per launch, T < 2^24): a cursor is two
  // scalar registers, as it was before the extent became per-utterance
  struct TileAt {
    int t0; unsigned bT;
    __device__ int b() const { return (int)(bT & 255u); }
    __device__ int T() const { return (int)(bT >> 8); }
  };
  // (the ragged-batch parameters are re-read from the kernel-argument segment in the rare branch that needs them: kept
  // in scalar registers across the persistent loop they push other scalars into vector lanes -- the trick of g16_convp)
  auto T_of = [&](int b) -> int {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ClPairArgs* KArgs;
    KArgs ea = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ea));                             // (opaque: not hoisted out of the branch, not kept live)
    const int* gl = ea->glen;
    if (!gl) return ea->T;
    const int nB = ea->B;
    return __builtin_amdgcn_readfirstlane(gl[b < nB ? b : nB - 1]) * ea->grate;   // (a cursor may step past the last utterance: never used then)
#else
    return 0;
#endif
  };
  auto tile_step = [&](TileAt t) -> TileAt {
    t.t0 += R2;
    if (t.t0 >= t.T()) { t.t0 = 0; const int nb_ = t.b() + 1; t.bT = (unsigned)(nb_ & 255) | ((unsigned)T_of(nb_) << 8); }
    return t;
  };
  auto dma_window = [&](TileAt at, int lane) {
    const char* xb = reinterpret_cast<const char*>(a.x + (size_t)at.b() * a.x_bs);
    const int tb = at.t0 - p2 - p1;
    const int Tu = at.T();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = wave + 8 * u;
      if (8 * p < xrows) {
        int t = tb + 8 * p + (lane >> 3);
        t = t < 0 ? 0 : (t >= Tu ? Tu - 1 : t);
        const char* gp = xb + ((unsigned)t * 128u + (unsigned)(lane & 7) * 16u);   // (uniform base + 32-bit lane offset)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(STG + p * 1024), 16, 0, 0);
      }
    }
    asm volatile("" ::: "memory");                           // (vector-memory operations below stay below: the counted wait)
  };
  // the caller has waited for this wave's pieces
  auto split_window = [&](TileAt at, int buf, int lane) {
    const int tb = at.t0 - p2 - p1;
    char* const xw = XW + buf * RW_XBUF;
    const int kq = lane & 3;
    f32x4 v[2][2];
    int r[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave + 8 * (2 * q + (lane >> 5));
      r[q] = 8 * p + ((lane >> 2) & 7);
      const char* sp = STG + r[q] * 128 + kq * 32;
      v[q][0] = *reinterpret_cast<const f32x4*>(sp);
      v[q][1] = *reinterpret_cast<const f32x4*>(sp + 16);
    }
    RW_STAMP(13);
    const int Tu = at.T();
    if (tb < 0 || tb + RW_WRX > Tu) {                        // (uniform: a window that leaves the utterance)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = tb + r[q];
        if (t < 0 || t >= Tu) { v[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; v[q][1] = v[q][0]; }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f16x4 h0, l0, h1, l1;
      g16_split4(v[q][0], slope, true, h0, l0);
      g16_split4(v[q][1], slope, true, h1, l1);
      const f16x8 eh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const f16x8 el = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      *reinterpret_cast<f16x8*>(xw + kq * RW_PL + r[q] * 16) = eh;
      *reinterpret_cast<f16x8*>(xw + kq * RW_PL + r[q] * 16 + RW_XIMG) = el;
    }
  };

  // ================= prologue: the first tile's window split, the second one's requested =================
  TileAt tc, tn, tn2, tp;                                    // tiles i, i + 1, i + 2, i - 1
  tc.t0 = lo_tile * R2;
  tc.bT = (unsigned)(lo_b & 255) | ((unsigned)T_of(lo_b) << 8);
  tn = tile_step(tc);
  tn2 = tile_step(tn);
  tp = tc;
  if (n > 0) {
    dma_window(tc, lane);
    g16_vmcnt<0>();
    split_window(tc, 0, lane);
    if (n > 1) dma_window(tn, lane);
  }
  G16_BARRIER();

  for (int i = 0; i <= n; ++i) {
#ifdef RW_STAMPS
    stamp_on = blockIdx.x == gridDim.x / 2 && i >= 20 && i < 24;
#endif
    RW_STAMP(1);
    // (per-lane addresses are re-derived from the lane number every iteration: hoisted out of the loop they cost the
    // K = 11 kernel registers it does not have)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int q4 = ln >> 4, l15 = ln & 15;
    // The window of tile i + 1 is split in this iteration from pieces every wave requested an iteration ago; right after
    // its split a wave requests its pieces of tile i + 2 (they are its private part of the staging area).  The conv2 waves
    // do this at the TOP of the iteration, the conv1 waves at the bottom: the two waves of a SIMD are out of step, one's
    // vector work falls into the other's MFMAs.
    const bool stage = i + 1 < n && !(diag & 4);
    auto stage_next = [&]() {
      RW_STAMP(9);
      split_window(tn, (i + 1) & 1, ln);
      RW_STAMP(12);
      if (i + 2 < n) dma_window(tn2, ln);
      RW_STAMP(10);
    };
    if (is1) {
      // ---- conv1 of tile i: columns [48 wr, 48 wr + 48), column c is time t0 - p2 + c
      const unsigned xb0 = lds0 + (i & 1) * RW_XBUF + q4 * RW_PL + (wr * RW_CW + l15) * 16;
      if (i < n) { prime_hi(xb0); nBl = g16_lds_read<RW_XIMG>(xb0); }
      RW_STAMP(2);
      if (i < n) {
        TileAt at = tc;
        asm volatile("" : "+s"(at.t0), "+s"(at.bT));      // (fresh values inside the role's branch: gen16_rc.hip)
        const int t0 = at.t0, Tc = at.T();
        char* const ti = TI + (i & 1) * RW_TBUF;
        const bool inside = t0 - p2 >= 0 && t0 - p2 + RW_BT <= Tc;    // (uniform: every conv1 column of the tile is in the utterance)
        g16_for<RW_G>([&](auto GG) {
          constexpr int g = decltype(GG)::value;
          f32x4 hh0, hh1;
          conv_group(std::integral_constant<int, RW_XIMG>{}, std::integral_constant<bool, g + 1 == RW_G>{}, xb0 + g * 256,
                     (unsigned)a.dil * 16, xb0 + (g + 1) * 256, hh0, hh1);
          RW_STAMP(3 + g);
          // activated, split tile -> t image; columns outside the utterance are conv2's zero padding.  A lane's four
          // channels 16 i + 4 q4 .. + 3 sit in plane 2 i + (q4 >> 1) at byte 8 (q4 & 1) of the row's 16
          const int col = wr * RW_CW + 16 * g + l15;
          const int tt = t0 - p2 + col;
          const float f = inside || (tt >= 0 && tt < Tc) ? G16_UNSCALE : 0.f;   // (the unscaling and the zero padding of columns outside the utterance in ONE multiply: the factor is per column)
          const f32x4 t0v = hh0 * f, t1v = hh1 * f;
          if (!(diag & 8)) {
            f16x4 eh, el;
            char* dst = ti + (q4 >> 1) * RW_PLT + col * 16 + 8 * (q4 & 1);
            g16_split4(t0v, slope, true, eh, el);
            *reinterpret_cast<f16x4*>(dst) = eh;
            *reinterpret_cast<f16x4*>(dst + RW_TIMG) = el;
            g16_split4(t1v, slope, true, eh, el);
            *reinterpret_cast<f16x4*>(dst + 2 * RW_PLT) = eh;
            *reinterpret_cast<f16x4*>(dst + 2 * RW_PLT + RW_TIMG) = el;
          }
          RW_STAMP(6 + g);
        });
      }
      if (stage) {
        g16_vmcnt<0>();                                      // (the pieces are this wave's only vector-memory traffic)
        stage_next();
      }
    } else {
      // ---- conv2 of tile i - 1: output column c is time t0 + c and reads t image rows c .. c + K - 1
      const unsigned tb0 = lds0 + 2 * RW_XBUF + ((i - 1) & 1) * RW_TBUF + q4 * RW_PLT + (wr * RW_CW + l15) * 16;
      if (stage) {
        // this wave's pieces went out before the previous iteration's 2 (4) loads + 2 stores per group -- out-of-range
        // offsets instead of predicates, so every group issues them --: they have landed once at most that many
        // operations are outstanding (vector-memory operations retire in issue order)
        if (i <= 1) g16_vmcnt<0>();
        else g16_vmcnt<(ACC ? 6 : 4) * RW_G>();
        stage_next();
      }
      if (i >= 1) { prime_hi(tb0); nBl = g16_lds_read<RW_TIMG>(tb0); }
      RW_STAMP(2);
      if (i >= 1) {
        TileAt at = tp;
        asm volatile("" : "+s"(at.t0), "+s"(at.bT));
        const int b = at.b(), t0 = at.t0, Tp = at.T();
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, Tp * 128, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.o_bs, 0, Tp * 128, 0x00020000);
        auto off_of = [&](int g) -> int {
          const int r = wr * RW_CW + 16 * g + l15;
          return (r < R2 && t0 + r < Tp) ? ((t0 + r) * 32 + 4 * q4) * 4 : G16_OOR;
        };
        // residual (and previous ResBlock sum) operands: one group ahead where the registers allow (K = 7), with the
        // group otherwise
        u32x4 res[2][2];
        [[maybe_unused]] u32x4 prv[2][2];
        auto fetch = [&](auto GG) {
          constexpr int g = decltype(GG)::value, s = g & 1;
          const int off = (diag & 1) ? G16_OOR : off_of(g);
          res[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
          res[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 64, 0);
          if constexpr (ACC) {
            prv[s][0] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0);
            prv[s][1] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 64, 0);
          }
        };
        constexpr bool AHEAD = K < 11;     // (K = 11: 176 weight registers leave room for one group's operands only)
        if constexpr (AHEAD) fetch(std::integral_constant<int, 0>{});
        g16_for<RW_G>([&](auto GG) {
          constexpr int g = decltype(GG)::value, s = g & 1;
          if constexpr (!AHEAD) fetch(GG);
          else if constexpr (g + 1 < RW_G) fetch(std::integral_constant<int, g + 1>{});
          f32x4 hh0, hh1;
          conv_group(std::integral_constant<int, RW_TIMG>{}, std::integral_constant<bool, g + 1 == RW_G>{}, tb0 + g * 256, 16u,
                     tb0 + (g + 1) * 256, hh0, hh1);
          RW_STAMP(3 + g);
          const int off = (diag & 2) ? G16_OOR : off_of(g);
          f32x4 v0 = hh0 * G16_UNSCALE, v1 = hh1 * G16_UNSCALE;
          v0 += g16_as_f32x4(res[s][0]);
          v1 += g16_as_f32x4(res[s][1]);
          if constexpr (ACC) { v0 += g16_as_f32x4(prv[s][0]); v1 += g16_as_f32x4(prv[s][1]); }
          g16_div(v0, a.div); g16_div(v1, a.div);
          __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v0), ro, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v1), ro, off, 64, 0);
          RW_STAMP(6 + g);
        });
      }
    }
    RW_STAMP(11);
    G16_BARRIER();
    tp = tc; tc = tn; tn = tn2;
    tn2 = tile_step(tn2);
  }
}

bool g16_rw_supported(int C, int K, int dil, int terms) {
  return C == 32 && (K == 3 || K == 7 || K == 11) && dil >= 1 && (K - 1) * dil <= RW_MAXHALO && terms == 3;
}

template <int K, bool ACC>
static hipError_t launch_g16_rw_k(ClPairArgs a, int B, hipStream_t s) {
  auto kern = g16_rw<K, ACC>;
  static std::atomic<uint64_t> attr_done{0};
  hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), RW_LDS, attr_done);
  if (e != hipSuccess) return e;
  constexpr int R2 = RW_BT - (K - 1);
  a.tiles = (a.T + R2 - 1) / R2;
  if (B > 256) {                                // (a tile cursor keeps the utterance in 8 bits: 256 utterances per launch)
    for (int b0 = 0; b0 < B; b0 += 256) {
      ClPairArgs c = a;
      c.x = a.x + (size_t)b0 * a.x_bs; c.out = a.out + (size_t)b0 * a.o_bs;
      if (a.glen) c.glen = a.glen + b0;
      if (hipError_t e2 = launch_g16_rw_k<K, ACC>(c, B - b0 < 256 ? B - b0 : 256, s); e2 != hipSuccess) return e2;
    }
    return hipSuccess;
  }
  a.B = B;
  const long total = (long)a.tiles * B;        // (ragged batch: the upper bound; the blocks count the real tiles themselves)
  if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
  int dev = 0, cus = 0;
  e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  if (cus <= 0) cus = 256;
  const int nb = total < cus ? (int)total : cus;          // one persistent block per CU
  hipLaunchKernelGGL(kern, dim3(nb), dim3(512), RW_LDS, s, a, (int)total);
  return hipGetLastError();
}

hipError_t launch_g16_rw(const ClPairArgs& a0, int B, hipStream_t s) {
  ClPairArgs a = a0;
  // (timing-only ablations, results WRONG by construction: experiment builds only -- the product build keeps terms == 3)
#ifdef VSP_EXPERIMENTS
  static const int diag = []() { const char* e = getenv("VSP_RW_DIAG"); return e ? atoi(e) : 0; }();
#else
  constexpr int diag = 0;
#endif
  a.terms = 3 | (diag << 8);
  if (!g16_rw_supported(a0.C, a0.K, a0.dil, a0.terms) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out ||
      (size_t)a.T * 128 >= (size_t)1 << 31)
    return hipErrorInvalidValue;
  if (a.K == 3) return a.acc_prev ? launch_g16_rw_k<3, true>(a, B, s) : launch_g16_rw_k<3, false>(a, B, s);
  if (a.acc_prev) return a.K == 7 ? launch_g16_rw_k<7, true>(a, B, s) : launch_g16_rw_k<11, true>(a, B, s);
  return a.K == 7 ? launch_g16_rw_k<7, false>(a, B, s) : launch_g16_rw_k<11, false>(a, B, s);
}

}  // namespace vsp
