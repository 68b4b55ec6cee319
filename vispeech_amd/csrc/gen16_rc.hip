// g16_rc: the whole kernel-3 ResBlock1 of the 32-channel stage (reference modules.py:210-223: three dilation pairs,
// x_{p+1} = x_p + conv2_p(lrelu(conv1_p(lrelu(x_p), d_p) + b1_p), 1) + b2_p) in ONE launch, as a ROLE PIPELINE with the
// WEIGHTS IN REGISTERS (round 4) -- g16_rw's machinery (gen16_rw.hip) applied to g16_chain's problem (gen16.hip):
//
//   * six 32-channel kernel-3 convolutions are 6 x 48 registers of A fragments: waves 0-3 of a block hold conv1 of the
//     three pairs, waves 4-7 conv2 of the three pairs, for the block's whole life: no weight ring, no slice hand-over.
//   * PERSISTENT blocks, one per CU, walk a run of 192-column tiles.  A tile goes through three PASSES (one per pair); a
//     pass is conv1 by the conv1 waves, then -- an iteration later -- conv2 + residual by the conv2 waves, whose result is
//     the next pass's input.  Two tiles are in flight, three iterations apart, so that every iteration has exactly one item
//     per role:   iteration 3m: (tile m, pass 0)   3m + 1: (tile m - 1, pass 2)   3m + 2: (tile m, pass 1)
//     and the conv2 waves work on the item of the iteration before.  ONE barrier per iteration.
//   * images with a FIXED column <-> time mapping (column c of a tile is time tb + c in every convolution, GRD guard rows
//     on either side; a tap reads row c + (tap - 1) * dilation): the conv2 waves own the same columns in every pass, so the
//     running x_p stays in their REGISTERS in D-tile layout (fp32, lane-local residual add); columns within the
//     accumulated padding H of a tile edge compute garbage that never reaches a stored column (a D column depends on
//     its own B column only); the H = 12 columns per side are recomputed by the neighbouring tile.
//   * the pass-0 image of a tile is split from its fp32 window, which arrives by LDS-DMA into a staging area three
//     iterations ahead (every wave owns up to four 1 KiB pieces, splits them and re-requests them for the tile after the
//     next); the pass-1 / pass-2 images are written by the conv2 waves from registers; the t images by the conv1 waves.
//   HBM sees x once (+ the conv2 waves' fp32 residual read of it, an L2 hit) and the result once: a third of the pair
//   path's traffic.  Per output the arithmetic is that of g16_chain / g16_pair / g16_conv: BIT-IDENTICAL results.
//
// LDS: x images 2 x 32 KB, t images 2 x 26 KB, staging 32 KB (+ the six biases in its tail) = 148 KB.
#include "g16_common.h"

#include <cstdlib>
#include <cstring>

namespace vsp {

namespace {
constexpr int RC_K = 3, RC_NP = 3;
constexpr bool RC_PRIO = true;    // (the MFMA clusters at raised priority: 2.66 against 2.69 ms without)
constexpr int RC_BT = 192;                 // columns per tile
constexpr int RC_CW = RC_BT / 4;           // columns per role wave
constexpr int RC_G = RC_CW / 16;           // 16-column groups per role wave and tile
constexpr int RC_GRD = 8;                  // guard rows on either side of an image (>= the largest dilation)
constexpr int RC_WR = RC_BT + 2 * RC_GRD;  // image rows (208)
constexpr int RC_PL = RC_WR * 16, RC_IMG = 4 * RC_PL, RC_BUF = 2 * RC_IMG;     // plane, image (hi or lo), hi + lo
constexpr int RC_STG = 256 * 128;          // fp32 staging: rows of 32 floats (208 used) + the biases in the last rows
constexpr int RC_BIAS = 4 * RC_BUF + RC_STG - 1024;
constexpr int RC_LDS = 4 * RC_BUF + RC_STG;
static_assert(RC_LDS <= 160 * 1024, "LDS budget");
static_assert(RC_PL % 256 == 0, "plane size keeps the fragment reads conflict-free");
static_assert(RC_WR * 128 <= RC_STG - 1024, "window and biases share the staging area");
}  // namespace

template <bool ACC>
__global__ void __launch_bounds__(512) g16_rc(ClChainArgs a, int total_tiles) {
  constexpr int K = RC_K;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const XW = lds;                    // x images of the two tiles in flight (buffer = tile & 1)
  char* const TI = lds + 2 * RC_BUF;       // t images (buffer = iteration & 1)
  char* const STG = lds + 4 * RC_BUF;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is1 = wave < 4;               // conv1 waves 0-3, conv2 waves 4-7 (w and w + 4 share a SIMD)
  const int wr = wave & 3;

  // this block's run of tiles in the (utterance, tile) sequence; ragged batch (ClChainArgs::glen): utterance b has
  // ceil(glen[b] * grate / R) tiles instead of a.tiles
  const int nb = gridDim.x, bid = blockIdx.x;
  const int H = a.halo, R = RC_BT - 2 * H;
  int lo_b, lo_tile, n;
  if (a.glen) {
    g16_ragged_run(a.glen, a.B, a.grate, R, nb, bid, lo_b, lo_tile, n);
  } else {
    const int per = total_tiles / nb, extra = total_tiles - per * nb;
    const int lo = bid * per + (bid < extra ? bid : extra);
    n = per + (bid < extra ? 1 : 0);
    lo_b = lo / a.tiles;
    lo_tile = lo - lo_b * a.tiles;
  }

  // ---- the role's weights: A fragments of the three pairs' packed images [tap][m-tile][hi | lo][lane][8 halfs]
  f16x8 Wh[RC_NP][K][2], Wl[RC_NP][K][2];
#pragma unroll
  for (int p = 0; p < RC_NP; ++p) {
    const uint16_t* const wsrc = a.w[2 * p + (is1 ? 0 : 1)];
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const size_t blk = ((size_t)tap * 2 + mt) * 2;
        Wh[p][tap][mt] = *reinterpret_cast<const f16x8*>(wsrc + (blk * 64 + lane) * 8);
        Wl[p][tap][mt] = *reinterpret_cast<const f16x8*>(wsrc + ((blk + 1) * 64 + lane) * 8);
      }
  }
  // (a use here retires the loads before the persistent loop: left to hipcc, the wait for them -- a vmcnt(0) -- would sit
  // in front of the first MFMA of every iteration)
#pragma unroll
  for (int p = 0; p < RC_NP; ++p)
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        asm volatile("" ::"v"(Wh[p][tap][mt]));
        asm volatile("" ::"v"(Wl[p][tap][mt]));
      }
  // biases: [pass][conv1 | conv2][32] floats
  if (wr == 0 && lane < 32) {
#pragma unroll
    for (int p = 0; p < RC_NP; ++p)
      reinterpret_cast<float*>(lds + RC_BIAS)[(2 * p + (is1 ? 0 : 1)) * 32 + lane] = a.b[2 * p + (is1 ? 0 : 1)][lane];
  }
  const float slope = a.slope;

  // ---- one 16-column group of one convolution: K taps x (2 B fragments from LDS, 6 MFMAs), B double-buffered, a tap's
  //      reads requested a whole tap ahead (g16_rw's form).  prime() requests a group's first fragments and its bias: at
  //      the top of an item for its first group, from inside the previous group's last tap otherwise.
  f16x8 nBh, nBl;
  auto prime = [&](unsigned baddr, unsigned) {
    nBh = g16_lds_read<0>(baddr);
    nBl = g16_lds_read<RC_IMG>(baddr);
  };
  auto conv_group = [&](auto P, auto LAST, unsigned baddr, unsigned step, unsigned bnext, unsigned bias_a, f32x4& hh0,
                        f32x4& hh1) {
    constexpr int p = decltype(P)::value;
    constexpr bool last = decltype(LAST)::value;
    // (the bias straight into the accumulators: requested behind the group's primed fragments, it is the youngest read)
    hh0 = __builtin_bit_cast(f32x4, g16_lds_read<0>(bias_a));
    hh1 = __builtin_bit_cast(f32x4, g16_lds_read<64>(bias_a));
    if constexpr (RC_PRIO) __builtin_amdgcn_s_setprio(1);
    f16x8 Bh[2], Bl[2];
    Bh[0] = nBh; Bl[0] = nBl;
    g16_for<K>([&](auto T) {
      constexpr int tap = decltype(T)::value, cur = tap & 1;
      if constexpr (tap + 1 < K) {
        const unsigned an = baddr + (tap + 1) * step;
        Bh[cur ^ 1] = g16_lds_read<0>(an);
        Bl[cur ^ 1] = g16_lds_read<RC_IMG>(an);
        g16_lgkmcnt<2>();
      } else if constexpr (!last) {
        prime(bnext, bias_a);
        g16_lgkmcnt<2>();
      } else {
        g16_lgkmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      hh0 = G16_MFMA(Wh[p][tap][0], Bh[cur], hh0);
      hh1 = G16_MFMA(Wh[p][tap][1], Bh[cur], hh1);
      hh0 = G16_MFMA(Wl[p][tap][0], Bh[cur], hh0);
      hh1 = G16_MFMA(Wl[p][tap][1], Bh[cur], hh1);
      hh0 = G16_MFMA(Wh[p][tap][0], Bl[cur], hh0);
      hh1 = G16_MFMA(Wh[p][tap][1], Bl[cur], hh1);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (RC_PRIO) __builtin_amdgcn_s_setprio(0);
  };

  // ---- tiles: the run's tiles m - 1 and m as carried cursors (utterance, time of column 0, the utterance's extent); tiles
  //      m + 1 / m + 2 are stepped from tile m where the staging needs them (scalar adds and compares)
  // (utterance | extent << 8 in ONE scalar -- B <= 256 per launch, T < 2^24 --: a cursor is two scalar registers)
  struct TileAt {
    int tb; unsigned bT;
    __device__ int b() const { return (int)(bT & 255u); }
    __device__ int T() const { return (int)(bT >> 8); }
  };
  // (the ragged-batch parameters are re-read from the kernel-argument segment in the rare branch that needs them: kept
  // in scalar registers across the persistent loop they push other scalars into vector lanes -- the trick of g16_convp)
  auto T_of = [&](int b) -> int {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ClChainArgs* KArgs;
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
    t.tb += R;
    if (t.tb + H >= t.T()) { t.tb = -H; const int nb_ = t.b() + 1; t.bT = (unsigned)(nb_ & 255) | ((unsigned)T_of(nb_) << 8); }
    return t;
  };
  // ---- the fp32 window of a tile: rows [tb - GRD, tb + BT + GRD) by LDS-DMA into the staging area (wave w: the 1 KiB
  //      pieces w, w + 8, w + 16, w + 24 of 8 rows each), split into its pass-0 x image by the wave that requested them.
  //      Rows outside the utterance are fetched from a clamped address and zeroed at the split (the reference's padding).
  auto dma_window = [&](TileAt at, int ln) {
    const char* xb = reinterpret_cast<const char*>(a.x + (size_t)at.b() * a.x_bs);
    const int t00 = at.tb - RC_GRD;
    const int Tu = at.T();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = wave + 8 * u;
      if (8 * p < RC_WR) {
        int t = t00 + 8 * p + (ln >> 3);
        t = t < 0 ? 0 : (t >= Tu ? Tu - 1 : t);
        const char* gp = xb + ((unsigned)t * 128u + (unsigned)(ln & 7) * 16u);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(STG + p * 1024), 16, 0, 0);
      }
    }
    asm volatile("" ::: "memory");
  };
  auto split_window = [&](TileAt at, int buf, int ln) {      // (the caller has waited for this wave's pieces)
    const int t00 = at.tb - RC_GRD;
    const int Tu = at.T();
    char* const xw = XW + buf * RC_BUF;
    const int kq = ln & 3;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave + 8 * (2 * q + (ln >> 5));
      const int r = 8 * p + ((ln >> 2) & 7);
      if (r < RC_WR) {
        const char* sp = STG + r * 128 + kq * 32;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 16);
        const int t = t00 + r;
        if (t < 0 || t >= Tu) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
        f16x4 h0, l0, h1, l1;
        g16_split4(v0, slope, true, h0, l0);
        g16_split4(v1, slope, true, h1, l1);
        const f16x8 eh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const f16x8 el = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        *reinterpret_cast<f16x8*>(xw + kq * RC_PL + r * 16) = eh;
        *reinterpret_cast<f16x8*>(xw + kq * RC_PL + r * 16 + RC_IMG) = el;
      }
    }
  };
  // an activated, split D-layout tile pair (channels 4 q4 .. + 3 and 16 + 4 q4 .. + 3 of column `col`) -> image; columns
  // outside the utterance are the next convolution's zero padding.  A lane's four channels 16 i + 4 q4 .. + 3 sit in plane
  // 2 i + (q4 >> 1) at byte 8 (q4 & 1) of the row's 16
  // (scale: G16_UNSCALE for accumulators, 1 for finished values -- the unscaling and the zero padding of columns outside
  // the utterance are ONE multiply, the factor is per column)
  auto write_image = [&](auto MASK, char* img, int col, int q4, int t, int T, f32x4 a0, f32x4 a1, float scale) {
    float f = scale;
    if constexpr (decltype(MASK)::value) f = t >= 0 && t < T ? scale : 0.f;
    const f32x4 v0 = a0 * f, v1 = a1 * f;
    f16x4 eh, el;
    char* dst = img + (q4 >> 1) * RC_PL + (RC_GRD + col) * 16 + 8 * (q4 & 1);
    g16_split4(v0, slope, true, eh, el);
    *reinterpret_cast<f16x4*>(dst) = eh;
    *reinterpret_cast<f16x4*>(dst + RC_IMG) = el;
    g16_split4(v1, slope, true, eh, el);
    *reinterpret_cast<f16x4*>(dst + 2 * RC_PL) = eh;
    *reinterpret_cast<f16x4*>(dst + 2 * RC_PL + RC_IMG) = el;
  };

  // ================= prologue: tile 0's window split, tile 1's requested =================
  TileAt tq, tc;                                             // tiles m - 1, m
  tc.tb = lo_tile * R - H;
  tc.bT = (unsigned)(lo_b & 255) | ((unsigned)T_of(lo_b) << 8);
  tq = tc;
  if (n > 0) {
    dma_window(tc, lane);
    g16_vmcnt<0>();
    split_window(tc, 0, lane);
    if (n > 1) dma_window(tile_step(tc), lane);
  }
  G16_BARRIER();

  // the conv2 waves' running x_p of the two tiles in flight, D-tile layout: [group][m-tile]
  f32x4 xa[RC_G][2], xb[RC_G][2];          // xa: the tile of this body's passes 0 / 1 (tile m); xb: tile m - 1

  // one ITEM of a role: (tile m, pass P) at iteration j -- conv1 waves run item (j), conv2 waves item (j - 1)
  // (the tile passes through an asm statement at the top of an item: as values of the enclosing body, live across the
  // uniform branch between the two instantiations, the cursor's scalars cost the register allocator 20 spilled VECTOR
  // registers -- weights reloaded from scratch inside the MFMA clusters)
  auto conv1_item_m = [&](auto P, auto MASK, TileAt at, int m, int j, int ln) {
    constexpr int p = decltype(P)::value;
    const int q4 = ln >> 4, l15 = ln & 15;
    asm volatile("" : "+s"(at.tb), "+s"(at.bT));
    const int Tu = at.T();
    const int d = a.dil[p];
    const unsigned xb0 = lds0 + (m & 1) * RC_BUF + q4 * RC_PL + (RC_GRD + wr * RC_CW + l15 - d) * 16;
    const unsigned bias_a = lds0 + RC_BIAS + (2 * p) * 128 + q4 * 16;
    prime(xb0, bias_a);
    char* const ti = TI + (j & 1) * RC_BUF;
    g16_for<RC_G>([&](auto GG) {
      constexpr int g = decltype(GG)::value;
      f32x4 hh0, hh1;
      conv_group(P, std::integral_constant<bool, g + 1 == RC_G>{}, xb0 + g * 256, (unsigned)d * 16, xb0 + (g + 1) * 256, bias_a,
                 hh0, hh1);
      const int col = wr * RC_CW + 16 * g + l15;
      const int t = at.tb + col;
      write_image(MASK, ti, col, q4, t, Tu, hh0, hh1, G16_UNSCALE);
    });
  };
  auto conv2_item_m = [&](auto P, auto MASK, TileAt at, int m, int j, int ln, f32x4 (&xr)[RC_G][2]) {
    constexpr int p = decltype(P)::value;
    const int q4 = ln >> 4, l15 = ln & 15;
    asm volatile("" : "+s"(at.tb), "+s"(at.bT));
    const int Tu = at.T();
    const unsigned tb0 = lds0 + 2 * RC_BUF + (j & 1) * RC_BUF + q4 * RC_PL + (RC_GRD + wr * RC_CW + l15 - 1) * 16;
    const unsigned bias_a = lds0 + RC_BIAS + (2 * p + 1) * 128 + q4 * 16;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)at.b() * a.x_bs, 0, Tu * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)at.b() * a.o_bs, 0, Tu * 128, 0x00020000);
    if constexpr (p == 0) {
      // x_0 at this wave's columns (fp32, zero outside the utterance): the residual operand of pair 0
#pragma unroll
      for (int g = 0; g < RC_G; ++g) {
        const int t = at.tb + wr * RC_CW + 16 * g + l15;
        const int off = (t >= 0 && t < Tu) ? (t * 32 + 4 * q4) * 4 : G16_OOR;
        xr[g][0] = g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        xr[g][1] = g16_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rx, off, 64, 0));
      }
    }
    // the previous ResBlock sum (last pass of an accumulating launch): one group ahead (16 registers, not 24)
    [[maybe_unused]] u32x4 prv[2][2];
    [[maybe_unused]] auto fetch_prv = [&](auto GG) {
      constexpr int g = decltype(GG)::value, sl = g & 1;
      const int col = wr * RC_CW + 16 * g + l15, t = at.tb + col;
      const int off = (col >= H && col < H + R && t < Tu) ? (t * 32 + 4 * q4) * 4 : G16_OOR;
      prv[sl][0] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0);
      prv[sl][1] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 64, 0);
    };
    if constexpr (p == RC_NP - 1 && ACC) fetch_prv(std::integral_constant<int, 0>{});
    prime(tb0, bias_a);
    char* const xw = XW + (m & 1) * RC_BUF;
    g16_for<RC_G>([&](auto GG) {
      constexpr int g = decltype(GG)::value;
      if constexpr (p == RC_NP - 1 && ACC && g + 1 < RC_G) fetch_prv(std::integral_constant<int, g + 1>{});
      f32x4 hh0, hh1;
      conv_group(P, std::integral_constant<bool, g + 1 == RC_G>{}, tb0 + g * 256, 16u, tb0 + (g + 1) * 256, bias_a, hh0, hh1);
      const int col = wr * RC_CW + 16 * g + l15;
      const int t = at.tb + col;
      f32x4 v0 = hh0 * G16_UNSCALE, v1 = hh1 * G16_UNSCALE;
      v0 += xr[g][0];
      v1 += xr[g][1];
      if constexpr (p + 1 < RC_NP) {
        xr[g][0] = v0; xr[g][1] = v1;                           // x_{p+1}
        write_image(MASK, xw, col, q4, t, Tu, v0, v1, 1.f);    // ... and the next pass's input image
      } else {
        const int off = (col >= H && col < H + R && t < Tu) ? (t * 32 + 4 * q4) * 4 : G16_OOR;
        if constexpr (ACC) { v0 += g16_as_f32x4(prv[g & 1][0]); v1 += g16_as_f32x4(prv[g & 1][1]); }
        g16_div(v0, a.div); g16_div(v1, a.div);
        __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v0), ro, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(g16_as_u32x4(v1), ro, off, 64, 0);
      }
    });
  };

  // tiles that lie inside their utterance (all but the first and the last of an utterance) skip the zero masks of the
  // image writes: a UNIFORM branch between two instantiations of an item (a run-time flag inside one instantiation cost
  // four spilled registers and 17 %)
  auto tile_inside = [&](TileAt at) { return at.tb >= 0 && at.tb + RC_BT <= at.T(); };
  auto conv1_item = [&](auto P, TileAt at, int m, int j, int ln) {
    if (tile_inside(at)) conv1_item_m(P, std::false_type{}, at, m, j, ln);
    else conv1_item_m(P, std::true_type{}, at, m, j, ln);
  };
  auto conv2_item = [&](auto P, TileAt at, int m, int j, int ln, f32x4 (&xr)[RC_G][2]) {
    if (tile_inside(at)) conv2_item_m(P, std::false_type{}, at, m, j, ln, xr);
    else conv2_item_m(P, std::true_type{}, at, m, j, ln, xr);
  };

  // ================= the pipeline: body m = iterations 3m, 3m + 1, 3m + 2 =================
  //   iteration 3m:     conv1 (m, 0)       conv2 (m - 1, 1) [item of iteration 3m - 1]
  //   iteration 3m + 1: conv1 (m - 1, 2)   conv2 (m, 0)
  //   iteration 3m + 2: conv1 (m, 1)       conv2 (m - 1, 2); the window of tile m + 1 is split into XW[(m + 1) & 1] (its
  //                     last reader -- conv1 (m - 1, 2) -- ran an iteration ago) and tile m + 2's is requested
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  for (int m = 0; m <= n; ++m) {
    int ln = lane;
    asm volatile("" : "+v"(ln));                               // (per-lane addresses re-derived per body: registers)
    // ---- iteration 3m
    if (is1) { if (m < n) conv1_item(P0{}, tc, m, 3 * m, ln); }
    else { if (m >= 1) conv2_item(P1{}, tq, m - 1, 3 * m - 1, ln, xb); }
    G16_BARRIER();
    // ---- iteration 3m + 1
    if (is1) { if (m >= 1) conv1_item(P2{}, tq, m - 1, 3 * m + 1, ln); }
    else { if (m < n) conv2_item(P0{}, tc, m, 3 * m, ln, xa); }
    G16_BARRIER();
    // ---- iteration 3m + 2
    const bool stage = m + 1 < n;
    if (is1) {
      if (m < n) conv1_item(P1{}, tc, m, 3 * m + 2, ln);
      if (stage) {
        g16_vmcnt<0>();                                        // (the pieces are this wave's only vector-memory traffic)
        const TileAt tn = tile_step(tc);
        split_window(tn, (m + 1) & 1, ln);
        if (m + 2 < n) dma_window(tile_step(tn), ln);
      }
    } else {
      if (stage) {
        // this wave's pieces went out three iterations ago, ahead of that body's loads and stores: everything older than
        // the final pass's operands below has to have landed anyway
        g16_vmcnt<0>();
        const TileAt tn = tile_step(tc);
        split_window(tn, (m + 1) & 1, ln);
        if (m + 2 < n) dma_window(tile_step(tn), ln);
      }
      if (m >= 1) conv2_item(P2{}, tq, m - 1, 3 * m + 1, ln, xb);
    }
    G16_BARRIER();
    tq = tc;
    tc = tile_step(tc);
    // the tile of passes 0 / 1 becomes the "previous" tile
#pragma unroll
    for (int g = 0; g < RC_G; ++g) { xb[g][0] = xa[g][0]; xb[g][1] = xa[g][1]; }
  }
}

// 32 channels, kernel 3, exactly three pairs with dilations <= RC_GRD, fp32-accurate products, no previous ResBlock sum
// to add (the k3 ResBlock is the first of a stage; the accumulating form has no registers for that operand: g16_chain)
bool g16_rc_supported(int C, int K, const int* dil, int np, int terms, int acc_prev) {
  if (C != 32 || K != RC_K || np != RC_NP || terms != 3 || acc_prev) return false;
  for (int p = 0; p < np; ++p)
    if (dil[p] < 1 || dil[p] > RC_GRD) return false;
  return true;
}

hipError_t launch_g16_rc(const ClChainArgs& a0, int B, hipStream_t s) {
  ClChainArgs a = a0;
  if (!g16_rc_supported(a.C, a.K, a.dil, a.np, a.terms, a.acc_prev) || a.T <= 0 || B <= 0 || (a.x_bs & 3) || (a.o_bs & 3) ||
      (reinterpret_cast<uintptr_t>(a.x) & 15) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.x == a.out ||
      (size_t)a.T * 128 >= (size_t)1 << 31)
    return hipErrorInvalidValue;
  a.halo = 0;
  for (int p = 0; p < a.np; ++p) a.halo += (a.dil[p] + 1) * ((a.K - 1) / 2);
  const int R = RC_BT - 2 * a.halo;
  if (R < 32) return hipErrorInvalidValue;
  a.tiles = (a.T + R - 1) / R;
  if (B > 256) {                                  // (a tile cursor keeps the utterance in 8 bits: 256 utterances per launch)
    for (int b0 = 0; b0 < B; b0 += 256) {
      ClChainArgs c = a0;
      c.x = a0.x + (size_t)b0 * a0.x_bs; c.out = a0.out + (size_t)b0 * a0.o_bs;
      if (a0.glen) c.glen = a0.glen + b0;
      if (hipError_t e2 = launch_g16_rc(c, B - b0 < 256 ? B - b0 : 256, s); e2 != hipSuccess) return e2;
    }
    return hipSuccess;
  }
  a.B = B;
  const long total = (long)a.tiles * B;          // (ragged batch: the upper bound; the blocks count the real tiles themselves)
  if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return e;
  if (cus <= 0) cus = 256;
  const int nb = total < cus ? (int)total : cus;            // one persistent block per CU
  static std::atomic<uint64_t> attr_done{0};
  e = set_max_dynamic_lds(reinterpret_cast<const void*>(g16_rc<false>), RC_LDS, attr_done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(g16_rc<false>, dim3(nb), dim3(512), RC_LDS, s, a, (int)total);
  return hipGetLastError();
}

}  // namespace vsp
