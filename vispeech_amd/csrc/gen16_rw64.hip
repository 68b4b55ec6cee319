// g16_rw64: the kernel-3 ResBlock1 conv PAIRS of the 64-channel stage (reference modules.py:210-223) with the WEIGHTS IN
// REGISTERS (round 5) -- g16_rw's role pipeline (gen16_rw.hip) at twice the channels.
//
// Why only kernel 3.  A 64-channel convolution is K x 4 m-tiles x 2 chunks x (hi | lo) A fragments = 64 K registers per
// lane for all output rows: no wave can hold that.  HALF the rows (two m-tiles = 32 output channels) of a kernel-3
// convolution are 96 registers.  So a role has FOUR waves = 2 row halves x 2 column halves: waves 0-3 hold conv1's
// weights, waves 4-7 conv2's, for the life of a PERSISTENT block per CU -- no weight ring, no per-slice wait / barrier /
// LDS-DMA issue (g16_pair's k3 block spends 13 % of its life waiting for slices, 17 % in its epilogue, 24 % multiplying:
// profiles/NOTEBOOK_r01_r04.md), and a tap's B fragments (2 LDS reads) feed 6 MFMAs, as in g16_rw.
//
//   * tiles of 96 conv1 columns (94 output columns): wave (rh, ch) of a role owns output channels [32 rh, 32 rh + 32) of
//     columns [48 ch, 48 ch + 48) = three 16-column groups; a group is 2 chunks x 3 taps = 6 steps of 2 B reads + 6 MFMAs;
//   * the pipeline is g16_rw's: in iteration i the conv1 waves multiply tile i from the x image XW[i & 1] and write their
//     activated, split result as the t image TI[i & 1]; the conv2 waves multiply tile i - 1 from TI[(i - 1) & 1], add the
//     residual and store; ONE barrier per tile; the fp32 window of a tile arrives by LDS-DMA in a staging area two tiles
//     ahead (wave-private 1 KiB pieces of 4 rows x 64 channels), split by the wave that requested it -- conv2 waves at the
//     top of an iteration, conv1 waves at the bottom, so that the two waves of a SIMD are out of step;
//   * the arithmetic per output (chunk-major, tap-minor, HH / CROSS / CROSS per step, bias in the accumulator,
//     acc * 2^-8 + x) is that of g16_pair / g16_conv: results are BIT-IDENTICAL (tests/test_cl_ops.py,
//     tests/test_hip_parity.py: VSP_PAIR=ring keeps the LDS-ring kernel as the second implementation).
//
// LDS: x images 2 x 28 KB, t images 2 x 28 KB, staging 28 KB, biases = 141 KB: one block of 8 waves per CU.
#include "g16_common.h"

#include <cstdlib>

namespace vsp {

namespace {
constexpr int R6_K = 3;
constexpr int R6_BT = 96;                  // conv1 columns per tile
constexpr int R6_CW = R6_BT / 2;           // columns per role wave
constexpr int R6_G = R6_CW / 16;           // 16-column groups per role wave and tile
constexpr int R6_WR = 112;                 // image / staging rows allocated: x window BT + 2 dil <= 112, t image BT + 2
constexpr int R6_PL = R6_WR * 16;          // one plane: [row][8 halfs]
constexpr int R6_IMG = 4 * R6_PL;          // one image (hi or lo) of one 32-channel chunk
constexpr int R6_CH = 2 * R6_IMG;          // hi + lo of one chunk
constexpr int R6_BUF = 2 * R6_CH;          // both chunks
constexpr int R6_STG = R6_WR * 256;        // fp32 staging: rows of 64 floats
constexpr int R6_BIAS = 4 * R6_BUF + R6_STG;
constexpr int R6_LDS = R6_BIAS + 512;      // + the two biases (2 x 256 B)
constexpr int R6_MAXDIL = (R6_WR - R6_BT) / 2;
static_assert(R6_LDS <= 160 * 1024, "LDS budget");
static_assert(R6_PL % 256 == 0, "plane size keeps the fragment reads conflict-free");
}  // namespace

template <bool ACC>
__global__ void __launch_bounds__(512) g16_rw64(ClPairArgs a, int total_tiles) {
  constexpr int K = R6_K, p2 = 1, R2 = R6_BT - (K - 1), NSTEP = 2 * K;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const XW = lds;
  char* const TI = lds + 2 * R6_BUF;
  char* const STG = TI + 2 * R6_BUF;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  const int tid = threadIdx.x, lane = tid & 63, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is1 = wave < 4;               // conv1 waves 0-3, conv2 waves 4-7 (w and w + 4 share a SIMD)
  const int wr = wave & 3, rh = wr >> 1, chh = wr & 1;      // row half, column half

  // this block's run of tiles in the (utterance, tile) sequence; ragged batch: utterance b has ceil(glen[b] grate / R2)
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
  const int xrows = R6_BT + (K - 1) * a.dil;

  // ---- the role's weights: A fragments [chunk][tap][m-tile][hi | lo][lane][8 halfs] of its two m-tiles, and its bias
  const uint16_t* const wsrc = is1 ? a.w1h : a.w2h;
  f16x8 Wh[NSTEP][2], Wl[NSTEP][2];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t blk = ((size_t)s * 4 + 2 * rh + i) * 2;       // s = chunk * K + tap; four m-tiles per (chunk, tap)
      Wh[s][i] = *reinterpret_cast<const f16x8*>(wsrc + (blk * 64 + lane) * 8);
      Wl[s][i] = *reinterpret_cast<const f16x8*>(wsrc + ((blk + 1) * 64 + lane) * 8);
    }
  // (a use here retires the loads before the persistent loop: gen16_rw.hip)
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      asm volatile("" ::"v"(Wh[s][i]));
      asm volatile("" ::"v"(Wl[s][i]));
    }
  if (wr == 0) reinterpret_cast<float*>(lds + R6_BIAS)[(is1 ? 0 : 64) + lane] = (is1 ? a.b1 : a.b2)[lane];
  const float slope = a.slope;

  // ---- one 16-column group: 6 steps (chunk-major, tap-minor) x (2 B fragments from LDS, 6 MFMAs), B double-buffered, a
  //      step's reads requested a whole step ahead; the group's first fragments and its bias are PRIMED from inside the
  //      previous group's last step (gen16_rw.hip).  baddr = LDS byte address of this lane's (chunk 0, tap 0) fragment.
  const unsigned bias_a = lds0 + R6_BIAS + (is1 ? 0 : 256) + (32 * rh + 4 * q4) * 4;
  // R6_AHEAD = 2: a step's fragments are requested TWO steps ahead (three rotating B sets): 6 MFMAs are 96 clocks, an LDS
  // round trip under load is longer.
#ifndef R6_AHEAD
#define R6_AHEAD 2
#endif
  f16x8 nBh, nBl;
  [[maybe_unused]] f16x8 nBh1, nBl1;
  f32x4 nh0, nh1;
  unsigned prime_step = 0;                                   // (tap stride of the convolution being primed: set by the role)
  auto prime = [&](unsigned baddr) {
    nh0 = __builtin_bit_cast(f32x4, g16_lds_read<0>(bias_a));
    nh1 = __builtin_bit_cast(f32x4, g16_lds_read<64>(bias_a));
    nBh = g16_lds_read<0>(baddr);
    nBl = g16_lds_read<R6_IMG>(baddr);
    if constexpr (R6_AHEAD == 2) {
      nBh1 = g16_lds_read<0>(baddr + prime_step);            // step 1 = (chunk 0, tap 1)
      nBl1 = g16_lds_read<R6_IMG>(baddr + prime_step);
    }
  };
  auto conv_group = [&](auto LAST, unsigned baddr, unsigned step, unsigned bnext, f32x4& hh0, f32x4& hh1) {
    constexpr bool last = decltype(LAST)::value;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (R6_AHEAD == 2) {
      f16x8 Bh[3], Bl[3];
      Bh[0] = nBh; Bl[0] = nBl; Bh[1] = nBh1; Bl[1] = nBl1;
      g16_for<NSTEP>([&](auto S) {
        constexpr int s = decltype(S)::value, cur = s % 3, nx2 = (s + 2) % 3;
        // outstanding after this step's requests, in issue order: step s + 1's two reads, then what is requested here
        if constexpr (s + 2 < NSTEP) {
          constexpr int cn = (s + 2) / K, tn = (s + 2) % K;
          const unsigned an = baddr + cn * R6_CH + tn * step;
          Bh[nx2] = g16_lds_read<0>(an);
          Bl[nx2] = g16_lds_read<R6_IMG>(an);
          g16_lgkmcnt<4>();
        } else if constexpr (s + 2 == NSTEP) {
          if constexpr (!last) { prime(bnext); g16_lgkmcnt<8>(); }      // (step s + 1's reads + the six primed ones)
          else g16_lgkmcnt<2>();
        } else {
          if constexpr (!last) g16_lgkmcnt<6>();                        // (the primed reads stay in flight)
          else g16_lgkmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s == 0) { hh0 = nh0; hh1 = nh1; }            // (the primed bias: landed, behind the wait above)
        hh0 = G16_MFMA(Wh[s][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wh[s][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wl[s][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wl[s][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wh[s][0], Bl[cur], hh0);
        hh1 = G16_MFMA(Wh[s][1], Bl[cur], hh1);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      f16x8 Bh[2], Bl[2];
      Bh[0] = nBh; Bl[0] = nBl;
      g16_for<NSTEP>([&](auto S) {
        constexpr int s = decltype(S)::value, cur = s & 1;
        if constexpr (s + 1 < NSTEP) {
          constexpr int cn = (s + 1) / K, tn = (s + 1) % K;
          const unsigned an = baddr + cn * R6_CH + tn * step;
          Bh[cur ^ 1] = g16_lds_read<0>(an);
          Bl[cur ^ 1] = g16_lds_read<R6_IMG>(an);
          g16_lgkmcnt<2>();
        } else if constexpr (!last) {
          prime(bnext);
          g16_lgkmcnt<4>();
        } else {
          g16_lgkmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s == 0) { hh0 = nh0; hh1 = nh1; }            // (the primed bias: landed, behind the wait above)
        hh0 = G16_MFMA(Wh[s][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wh[s][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wl[s][0], Bh[cur], hh0);
        hh1 = G16_MFMA(Wl[s][1], Bh[cur], hh1);
        hh0 = G16_MFMA(Wh[s][0], Bl[cur], hh0);
        hh1 = G16_MFMA(Wh[s][1], Bl[cur], hh1);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- tiles: first output column, and utterance | extent << 8 in ONE scalar (B <= 256 per launch, T < 2^24)
  struct TileAt {
    int t0; unsigned bT;
    __device__ int b() const { return (int)(bT & 255u); }
    __device__ int T() const { return (int)(bT >> 8); }
  };
  auto T_of = [&](int b) -> int {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ClPairArgs* KArgs;
    KArgs ea = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ea));                             // (opaque: not hoisted out of the branch, not kept live)
    const int* gl = ea->glen;
    if (!gl) return ea->T;
    const int nB = ea->B;
    return __builtin_amdgcn_readfirstlane(gl[b < nB ? b : nB - 1]) * ea->grate;
#else
    return 0;
#endif
  };
  auto tile_step = [&](TileAt t) -> TileAt {
    t.t0 += R2;
    if (t.t0 >= t.T()) { t.t0 = 0; const int nb_ = t.b() + 1; t.bT = (unsigned)(nb_ & 255) | ((unsigned)T_of(nb_) << 8); }
    return t;
  };
  // ---- x window of a tile: fp32 rows by LDS-DMA into the staging area (wave w: the 1 KiB pieces w, w + 8, w + 16, w + 24 of
  //      4 rows x 64 channels each), split into an x image by the wave that requested them.  Window row r is time
  //      t0 - p2 - p1 + r; rows outside the utterance are fetched from a clamped address and zeroed at the split.
  auto dma_window = [&](TileAt at, int ln) {
    const char* xb = reinterpret_cast<const char*>(a.x + (size_t)at.b() * a.x_bs);
    const int tb = at.t0 - p2 - p1;
    const int Tu = at.T();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = wave + 8 * u;
      if (4 * p < xrows) {
        int t = tb + 4 * p + (ln >> 4);
        t = t < 0 ? 0 : (t >= Tu ? Tu - 1 : t);
        const char* gp = xb + ((unsigned)t * 256u + (unsigned)(ln & 15) * 16u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(STG + p * 1024), 16, 0, 0);
      }
    }
    asm volatile("" ::: "memory");                           // (vector-memory operations below stay below: the counted wait)
  };
  // (the caller has waited for this wave's pieces)
  auto split_window = [&](TileAt at, int buf, int ln) {
    const int tb = at.t0 - p2 - p1;
    const int Tu = at.T();
    char* const xw = XW + buf * R6_BUF;
    const int u5 = ln & 31, kg = u5 & 7;                     // a lane: 8 channels (chunk kg >> 2, plane kg & 3) of one row
    f32x4 v[2][2];
    int r[2];
    bool on[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave + 8 * (2 * q + (ln >> 5));
      on[q] = 4 * p < xrows;
      r[q] = 4 * p + (u5 >> 3);
      const char* sp = STG + (on[q] ? r[q] : 0) * 256 + kg * 32;
      v[q][0] = *reinterpret_cast<const f32x4*>(sp);
      v[q][1] = *reinterpret_cast<const f32x4*>(sp + 16);
    }
    if (tb < 0 || tb + R6_WR > Tu) {                         // (uniform: a window that leaves the utterance)
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
      if (on[q]) {
        char* dst = xw + (kg >> 2) * R6_CH + (kg & 3) * R6_PL + r[q] * 16;
        *reinterpret_cast<f16x8*>(dst) = eh;
        *reinterpret_cast<f16x8*>(dst + R6_IMG) = el;
      }
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
    // (per-lane addresses are re-derived from the lane number every iteration: registers)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int q4 = ln >> 4, l15 = ln & 15;
    const bool stage = i + 1 < n;
    auto stage_next = [&]() {
      split_window(tn, (i + 1) & 1, ln);
      if (i + 2 < n) dma_window(tn2, ln);
    };
    if (is1) {
      // ---- conv1 of tile i: columns [48 chh, 48 chh + 48), column c is time t0 - p2 + c
      const unsigned xb0 = lds0 + (i & 1) * R6_BUF + q4 * R6_PL + (chh * R6_CW + l15) * 16;
      prime_step = (unsigned)a.dil * 16;
      if (i < n) prime(xb0);
      if (i < n) {
        TileAt at = tc;
        asm volatile("" : "+s"(at.t0), "+s"(at.bT));         // (fresh values inside the role's branch: gen16_rc.hip)
        const int t0 = at.t0, Tc = at.T();
        char* const ti = TI + (i & 1) * R6_BUF + rh * R6_CH;
        const bool inside = t0 - p2 >= 0 && t0 - p2 + R6_BT <= Tc;
        g16_for<R6_G>([&](auto GG) {
          constexpr int g = decltype(GG)::value;
          f32x4 hh0, hh1;
          conv_group(std::integral_constant<bool, g + 1 == R6_G>{}, xb0 + g * 256, (unsigned)a.dil * 16, xb0 + (g + 1) * 256,
                     hh0, hh1);
          // activated, split tile -> t image (chunk rh); columns outside the utterance are conv2's zero padding.  A lane's
          // four channels 16 i + 4 q4 .. + 3 of the chunk sit in plane 2 i + (q4 >> 1) at byte 8 (q4 & 1) of the row's 16
          const int col = chh * R6_CW + 16 * g + l15;
          const int tt = t0 - p2 + col;
          const float f = inside || (tt >= 0 && tt < Tc) ? G16_UNSCALE : 0.f;   // (unscaling and zero padding in one multiply)
          const f32x4 t0v = hh0 * f, t1v = hh1 * f;
          f16x4 eh, el;
          char* dst = ti + (q4 >> 1) * R6_PL + col * 16 + 8 * (q4 & 1);
          g16_split4(t0v, slope, true, eh, el);
          *reinterpret_cast<f16x4*>(dst) = eh;
          *reinterpret_cast<f16x4*>(dst + R6_IMG) = el;
          g16_split4(t1v, slope, true, eh, el);
          *reinterpret_cast<f16x4*>(dst + 2 * R6_PL) = eh;
          *reinterpret_cast<f16x4*>(dst + 2 * R6_PL + R6_IMG) = el;
        });
      }
      if (stage) {
        g16_vmcnt<0>();                                      // (the pieces are this wave's only vector-memory traffic)
        stage_next();
      }
    } else {
      // ---- conv2 of tile i - 1: output column c is time t0 + c and reads t image rows c .. c + 2
      const unsigned tb0 = lds0 + 2 * R6_BUF + ((i - 1) & 1) * R6_BUF + q4 * R6_PL + (chh * R6_CW + l15) * 16;
      if (stage) {
        // this wave's pieces went out before the previous iteration's 2 (4) loads + 2 stores per group: they have landed
        // once at most that many operations are outstanding (vector-memory operations retire in issue order)
        if (i <= 1) g16_vmcnt<0>();
        else g16_vmcnt<(ACC ? 6 : 4) * R6_G>();
        stage_next();
      }
      prime_step = 16u;
      if (i >= 1) prime(tb0);
      if (i >= 1) {
        TileAt at = tp;
        asm volatile("" : "+s"(at.t0), "+s"(at.bT));
        const int b = at.b(), t0 = at.t0, Tp = at.T();
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, Tp * 256, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.o_bs, 0, Tp * 256, 0x00020000);
        auto off_of = [&](int g) -> int {
          const int r = chh * R6_CW + 16 * g + l15;
          return (r < R2 && t0 + r < Tp) ? ((t0 + r) * 64 + 32 * rh + 4 * q4) * 4 : G16_OOR;
        };
        // residual (and previous ResBlock sum) operands one group ahead
        u32x4 res[2][2];
        [[maybe_unused]] u32x4 prv[2][2];
        auto fetch = [&](auto GG) {
          constexpr int g = decltype(GG)::value, s = g & 1;
#ifdef R6_DIAG_NORES
          const int off = G16_OOR;                                     // (timing-only ablation: no residual read; results wrong)
#else
          const int off = off_of(g);
#endif
          res[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
          res[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 64, 0);
          if constexpr (ACC) {
            prv[s][0] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0);
            prv[s][1] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 64, 0);
          }
        };
        fetch(std::integral_constant<int, 0>{});
        g16_for<R6_G>([&](auto GG) {
          constexpr int g = decltype(GG)::value, s = g & 1;
          if constexpr (g + 1 < R6_G) fetch(std::integral_constant<int, g + 1>{});
          f32x4 hh0, hh1;
          conv_group(std::integral_constant<bool, g + 1 == R6_G>{}, tb0 + g * 256, 16u, tb0 + (g + 1) * 256, hh0, hh1);
          const int off = off_of(g);
          f32x4 v0 = hh0 * G16_UNSCALE, v1 = hh1 * G16_UNSCALE;
          v0 += g16_as_f32x4(res[s][0]);
          v1 += g16_as_f32x4(res[s][1]);
          if constexpr (ACC) { v0 += g16_as_f32x4(prv[s][0]); v1 += g16_as_f32x4(prv[s][1]); }
          g16_div(v0, a.div); g16_div(v1, a.div);
          __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v0), ro, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v1), ro, off, 64, 0);
        });
      }
    }
    G16_BARRIER();
    tp = tc; tc = tn; tn = tn2;
    tn2 = tile_step(tn2);
  }
}

bool g16_rw64_supported(int C, int K, int dil, int terms) {
  return C == 64 && K == R6_K && dil >= 1 && dil <= R6_MAXDIL && terms == 3;
}

template <bool ACC>
static hipError_t launch_g16_rw64_k(ClPairArgs a, int B, hipStream_t s) {
  auto kern = g16_rw64<ACC>;
  static std::atomic<uint64_t> attr_done{0};
  hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), R6_LDS, attr_done);
  if (e != hipSuccess) return e;
  constexpr int R2 = R6_BT - (R6_K - 1);
  a.tiles = (a.T + R2 - 1) / R2;
  if (B > 256) {                                // (a tile cursor keeps the utterance in 8 bits: 256 utterances per launch)
    for (int b0 = 0; b0 < B; b0 += 256) {
      ClPairArgs c = a;
      c.x = a.x + (size_t)b0 * a.x_bs; c.out = a.out + (size_t)b0 * a.o_bs;
      if (a.glen) c.glen = a.glen + b0;
      if (hipError_t e2 = launch_g16_rw64_k<ACC>(c, B - b0 < 256 ? B - b0 : 256, s); e2 != hipSuccess) return e2;
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
  hipLaunchKernelGGL(kern, dim3(nb), dim3(512), R6_LDS, s, a, (int)total);
  return hipGetLastError();
}

hipError_t launch_g16_rw64(const ClPairArgs& a0, int B, hipStream_t s) {
  ClPairArgs a = a0;
  a.terms = 3;
  if (!g16_rw64_supported(a0.C, a0.K, a0.dil, a0.terms) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out ||
      (size_t)a.T * 256 >= (size_t)1 << 31)
    return hipErrorInvalidValue;
  return a.acc_prev ? launch_g16_rw64_k<true>(a, B, s) : launch_g16_rw64_k<false>(a, B, s);
}

}  // namespace vsp
