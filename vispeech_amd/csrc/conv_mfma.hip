// conv1d (and polyphase ConvTranspose1d) as an implicit GEMM on the gfx950 f32 matrix core.
//
//   GEMM view:  M = output rows (channels, or (channel,phase) pairs for the transposed conv),
//               N = time, K = (tap, input channel).
//   v_mfma_f32_32x32x2_f32: A lane l holds A[row l&31][k l>>5], B lane l holds B[k l>>5][col l&31],
//   D register r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31.  The product is an exact
//   f32 fmaf chain in k order, so results track the reference's fp32 convolution to rounding.
//
//   * the input tile x[ci chunk][t0-pad .. t0+BN+(K-1)dil) is staged ONCE per chunk into LDS with the
//     prologue (mask, leaky-relu) applied at staging time, so every tap re-reads activated values
//     from LDS (conflict-free: a half-wave reads 32 consecutive floats of one row);
//   * the weights are pre-packed on the host in A-fragment order (4 KiB per 32-row tile, chunk and tap).  f32 form:
//     a wave fetches four k-steps of one tile with a single coalesced 1 KiB global_load_dwordx4, one tap ahead,
//     straight into registers.  Split-f16 form (the default): the block's slice of a step is copied once by
//     LDS-DMA into a three-slot ring two steps ahead and read as ds_read_b128 fragments (RING, below);
//   * the epilogue (bias, conditioning, relu / WN gate, masks, residual, accumulate, divide,
//     polyphase scatter) runs on the accumulators, so every conv layer is one HBM read + one write.
//
// Reference call sites this kernel serves: modules.py:148-176 (WN), :210-223 (ResBlock1),
// :324-343 (coupling pre/post), attentions.py:138-145, 277-285 (1x1 projections, FFN),
// models.py:119-133, 271-290, 526-529; frame_prior_network.py:50-55.
#include "kernels.h"

#include <cstdlib>
#include <cstring>

namespace vsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

size_t packed_conv_floats(int M, int Cin, int K) {
  const size_t mt = (M + 31) / 32, nc = (Cin + CONV_CK - 1) / CONV_CK;
  return mt * nc * (size_t)K * 1024;
}

void pack_conv_weights(float* dst, int M, int Cin, int K, const float* dense) {
  const int nc = (Cin + CONV_CK - 1) / CONV_CK;
  std::memset(dst, 0, packed_conv_floats(M, Cin, K) * sizeof(float));
  for (int row = 0; row < M; ++row) {
    const int mtile = row >> 5, rin = row & 31;
    for (int ci = 0; ci < Cin; ++ci) {
      const int chunk = ci / CONV_CK, cc = ci % CONV_CK;
      const int kk = cc >> 1, hh = cc & 1, lane = rin + 32 * hh, kg = kk >> 2, i = kk & 3;
      for (int tap = 0; tap < K; ++tap) {
        const size_t idx =
            (((((size_t)mtile * nc + chunk) * K + tap) * (CONV_CK / 8) + kg) * 64 + lane) * 4 + i;
        dst[idx] = dense[((size_t)row * Cin + ci) * K + tap];
      }
    }
  }
}

// Split-f16 variant of the same packing (F16S kernels below): the 4 KiB block of one (m-tile, chunk,
// tap) holds [k-step 0..1][hi, lo][lane][8 halfs]; lane l = row (l & 31), k = 16*ks + 8*(l >> 5) + j.
// w = wh + wl * 2^-11 exactly as in gen16.hip.
void pack_conv_weights_f16s(float* dst_f, int M, int Cin, int K, const float* dense) {
  const int nc = (Cin + CONV_CK - 1) / CONV_CK;
  std::memset(dst_f, 0, packed_conv_floats(M, Cin, K) * sizeof(float));
  uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
  for (int row = 0; row < M; ++row) {
    const int mtile = row >> 5, rin = row & 31;
    for (int ci = 0; ci < Cin; ++ci) {
      const int chunk = ci / CONV_CK, cc = ci % CONV_CK;
      const int ks = cc >> 4, hh = (cc >> 3) & 1, j = cc & 7, lane = rin + 32 * hh;
      for (int tap = 0; tap < K; ++tap) {
        const float w = dense[((size_t)row * Cin + ci) * K + tap];
        const _Float16 h = (_Float16)w;
        const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
        const size_t blk = (((size_t)mtile * nc + chunk) * K + tap) * 2048;   // halfs per 4 KiB block
        std::memcpy(dst + blk + ((size_t)(ks * 2 + 0) * 64 + lane) * 8 + j, &h, 2);
        std::memcpy(dst + blk + ((size_t)(ks * 2 + 1) * 64 + lane) * 8 + j, &l, 2);
      }
    }
  }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// F16S: the same implicit GEMM on v_mfma_f32_32x32x16_f16 with fp32-accurate split operands (three
// MFMAs per product into an HH and a CROSS accumulator, gen16.hip) -- 16/3 x the f32 matrix rate.
// The staged window is then kept as f16 PAIRS of adjacent input channels, P[ci/2][t] (hi image, lo
// image; time contiguous, same bytes as the f32 window): a B fragment (8 consecutive ci of one time
// column) is four conflict-free ds_read_b32, and staging stays a 16-byte ds_write per four columns.
//
// RING (F16S only): the weights no longer come from global memory into every wave's registers one tap ahead (a tap of
// these small GEMMs is 12-24 MFMAs per wave: ~0.2 us of matrix work in front of a ~0.8 us L2 round trip, and the WN
// waves of a row group each fetched the same bytes).  The block's (m-tiles, chunk, tap) slice -- MT*WM contiguous
// 4 KiB fragment images -- is copied ONCE by LDS-DMA into a ring of NS slots, D = NS - 1 steps ahead of its use,
// synchronised like gen16.hip: counted vmcnt (a wave waits for ITS pieces of the step), one raw s_barrier per step
// (nothing drains the DMA queue), A fragments by ds_read_b128.
// CONV_DIAG: timing-only ablation builds (results wrong): bit 0 no MFMA, 1 no window loads, 2 no epilogue memory traffic,
// 3 no weight copies.  0 in the product build.
#ifndef CONV_DIAG
#define CONV_DIAG 0
#endif
#if CONV_DIAG & 1
#define CONV_MFMA16(a, b, c) ([&]() { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define CONV_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
#define CONV_RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int N>
__device__ __forceinline__ void conv_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
constexpr int conv_ring_slots(int MT, int WM) { return 3; }   // (four slots for the 8 KiB steps measured slower: 57 KiB per block = two blocks per CU instead of three)

template <int MT, int NT, int WM, int WN, bool F16S, bool RING = false>
__global__ void __launch_bounds__(64 * WM * WN, F16S ? 2 : 1) conv1d_f32_mfma(ConvArgs a) {
  static_assert(!RING || F16S, "the weight ring serves the split-f16 path");
  constexpr int BN = 32 * NT * WN;
  constexpr int LWP = BN + CONV_HALO;
  constexpr int NW = WM * WN;
  constexpr int KG = CONV_CK / 8;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [CONV_CK][LWP] (+ the weight ring)
  constexpr int SB = MT * WM * 4096;                  // bytes of one step's weights (RING)
  constexpr int NS = conv_ring_slots(MT, WM), DEPTH = NS - 1;
  constexpr int PW = MT * WM * 4 / NW;                // 1 KiB pieces per wave and step
  static_assert(!RING || (MT * WM * 4) % NW == 0, "pieces divide over the waves");
  [[maybe_unused]] char* const ring = reinterpret_cast<char*>(xs) + (size_t)CONV_CK * LWP * sizeof(float);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * BN;
  const int mtile0 = (blockIdx.y * WM + wm) * MT;
  const int n_mtiles = (a.M + 31) >> 5;
  const int len = a.lengths ? (int)a.lengths[b] : 0x7fffffff;
  const int LW = BN + (a.K - 1) * a.dil;
  const int total_it = a.nchunks * a.K;

  f32x16 acc[MT][NT];
  [[maybe_unused]] f32x16 crs[MT][NT];          // F16S: CROSS accumulator (acc is HH)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mt][nt][r] = 0.f;
        if constexpr (F16S) crs[mt][nt][r] = 0.f;
      }
  unsigned* const ph = reinterpret_cast<unsigned*>(xs);          // F16S: hi image [16][LWP] words
  unsigned* const pl = ph + (CONV_CK / 2) * LWP;                 //       lo image
  // hi = the fp32 value truncated to f16 precision (one v_and_b32; packed without rounding by v_cvt_pkrtz, which also
  // saturates instead of producing inf), lo = the exact fp32 residual * 2^11 (gen16.hip: g16_split2)
  auto split_pair = [&](float x0, float x1, unsigned& whi, unsigned& wlo) {
    const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffffe000u);
    whi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    wlo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((x0 - h0) * 2048.f, (x1 - h1) * 2048.f));
  };

  const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
  float4 a_cur[MT][KG], a_nxt[MT][KG];
  auto load_a = [&](int it, float4(&dst)[MT][KG]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mtile = mtile0 + mt;
      if (mtile < n_mtiles) {
        // it = chunk*K + tap, and the packed order is [mtile][chunk][tap][kg][lane]
        const float4* p = wp4 + (((size_t)mtile * total_it + it) * KG) * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) dst[mt][kg] = p[kg * 64];
      } else {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) dst[mt][kg] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if constexpr (!RING) load_a(0, a_cur);
  // RING: step `it` = (chunk, tap); piece p of a step = (m-tile p / 4 of the block, 1 KiB sub-image p % 4).  M-tiles past
  // the last one re-read it (their accumulators are never stored), so every wave issues PW pieces per step.
  [[maybe_unused]] auto ring_dma = [&](int step) {
    char* dst = ring + (step % NS) * SB;
    if constexpr ((CONV_DIAG & 8) != 0) return;
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      const int p = u * NW + wave;
      int mtile = blockIdx.y * (WM * MT) + (p >> 2);
      mtile = mtile < n_mtiles ? mtile : n_mtiles - 1;
      const float4* src = wp4 + (((size_t)mtile * total_it + step) * KG + (p & 3)) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  };
  // PREF (RING tiles of <= 128 columns, one staging sweep): the next chunk's window is requested right behind the
  // weight copy of the chunk's first step and converted / written after the chunk's last tap -- its HBM / L2 round trip
  // runs in the shadow of the chunk's matrix work.  The loads are younger than the weight copies of the next DEPTH
  // steps: the counted waits of those steps leave them in flight.
  constexpr bool PREF = RING && ((BN + CONV_HALO + 3) >> 2) <= 64;
  constexpr int RWPF = PREF ? (CONV_CK / 2) / NW : 1;
  constexpr int NPF = 2 * RWPF;                        // window loads per wave and chunk
  [[maybe_unused]] int pf_step = -1000;                // step at which the outstanding window request was issued
  [[maybe_unused]] auto ring_wait = [&](int step) {   // my pieces of `step` have landed: younger are the steps after it
    const int younger = total_it - 1 - step < DEPTH - 1 ? total_it - 1 - step : DEPTH - 1;
    const bool pf = PREF && step > pf_step && step - pf_step <= DEPTH;
    if (pf) {
      if (younger <= 0) conv_vmcnt<NPF>();
      else if (younger == 1) conv_vmcnt<PW + NPF>();
      else conv_vmcnt<2 * PW + NPF>();
    } else {
      if (younger <= 0) conv_vmcnt<0>();
      else if (younger == 1) conv_vmcnt<PW>();
      else conv_vmcnt<2 * PW>();
    }
  };
  static_assert(DEPTH <= 3, "ring_wait covers up to two younger steps");
  if constexpr (RING) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < total_it) ring_dma(d);
  }

  const float* xb = a.x + (size_t)b * a.x_bs;
  // vectorised staging needs 16-byte aligned rows (true for every internal [B][C][Ts] buffer)
  const bool vec = ((a.x_cs & 3) == 0) && ((a.x_bs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  const int t_start = vec ? (((t0 - a.pad) >> 2) << 2) : (t0 - a.pad);
  const int off = (t0 - a.pad) - t_start;
  const int LW4 = (LW + off + 3) >> 2;
  const float* xsb = xs + h * LWP + wn * (NT * 32) + l31 + off;
  [[maybe_unused]] float4 pva[RWPF], pvb[RWPF];
  [[maybe_unused]] auto st_load = [&](int chunk) {
    const int t4 = t_start + 4 * lane;
    const bool tin = lane < LW4 && t4 >= 0 && t4 < a.T_in;
#pragma unroll
    for (int j = 0; j < RWPF; ++j) {
      const int ci = chunk * CONV_CK + 2 * (wave + j * NW);
      // (always two loads per pair, so that the counted waits know how many are in flight: absent ones re-read row 0)
      const float* pa = xb + (size_t)(tin && ci < a.Cin ? ci : 0) * a.x_cs + (tin ? t4 : 0);
      const float* pb = xb + (size_t)(tin && ci + 1 < a.Cin ? ci + 1 : 0) * a.x_cs + (tin ? t4 : 0);
      pva[j] = *reinterpret_cast<const float4*>(pa);
      pvb[j] = *reinterpret_cast<const float4*>(pb);
    }
  };
  // conversion of a staged pair of channels x four times: leaky-relu as max(x, slope x) (0 <= slope <= 1; slope 1 =
  // none), packed f32 arithmetic, the truncating split of split_pair
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  [[maybe_unused]] const float slope_eff = a.in_act ? a.in_slope : 1.f;
  [[maybe_unused]] auto st_write = [&](int chunk) {
    if (lane >= LW4) return;
    const int t4 = t_start + 4 * lane;
    const bool tin = t4 >= 0 && t4 < a.T_in;
    const int lim = a.in_mask ? (len < a.T_in ? len : a.T_in) : a.T_in;
    const bool tfull = tin && t4 + 3 < lim;
#pragma unroll
    for (int j = 0; j < RWPF; ++j) {
      const int ci = chunk * CONV_CK + 2 * (wave + j * NW);
      const bool ina = tin && ci < a.Cin, inb = tin && ci + 1 < a.Cin;
      const float ea[4] = {pva[j].x, pva[j].y, pva[j].z, pva[j].w}, eb[4] = {pvb[j].x, pvb[j].y, pvb[j].z, pvb[j].w};
      unsigned wh4[4], wl4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x2v x = {ea[u], eb[u]};
        if (!(tfull && inb)) {                          // window edge / ragged channel count: zero what is not there
          x.x = (ina && t4 + u < lim) ? x.x : 0.f;
          x.y = (inb && t4 + u < lim) ? x.y : 0.f;
        }
        const f32x2v y = x * slope_eff;
        asm("v_max_f32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(y.y));
        const f32x2v hf = {__uint_as_float(__float_as_uint(x.x) & 0xffffe000u), __uint_as_float(__float_as_uint(x.y) & 0xffffe000u)};
        const f32x2v lf = (x - hf) * 2048.f;          // (packed arithmetic: v_pk_add_f32, v_pk_mul_f32)
        wh4[u] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(hf.x, hf.y));
        wl4[u] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lf.x, lf.y));
      }
      const int o = (wave + j * NW) * LWP + 4 * lane;
      *reinterpret_cast<u32x4*>(ph + o) = u32x4{wh4[0], wh4[1], wh4[2], wh4[3]};
      *reinterpret_cast<u32x4*>(pl + o) = u32x4{wl4[0], wl4[1], wl4[2], wl4[3]};
    }
  };
  const bool pref = PREF && vec && (CONV_DIAG & 2) == 0 && slope_eff >= 0.f && slope_eff <= 1.f;
  if constexpr (PREF) { if (pref) st_load(0); }
  int it = 0;
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    // ---- stage x[chunk] -> LDS with the prologue applied
    if (pref) {
      if constexpr (PREF) st_write(chunk);      // (the compiler waits for the window here: a chunk after its request)
    } else if (vec && F16S) {
      // pairs of adjacent input channels: two 16-byte loads -> one 16-byte LDS store per image
      constexpr int RWP = (CONV_CK / 2) / NW;
      for (int q0 = 0; q0 < LW4; q0 += 64) {
        const int q4 = q0 + lane;
        const int t4 = t_start + 4 * q4;
        const bool tin = q4 < LW4 && t4 >= 0 && t4 < a.T_in;
        float4 va[RWP], vb[RWP];
#pragma unroll
        for (int j = 0; j < RWP; ++j) {
          const int ci = chunk * CONV_CK + 2 * (wave + j * NW);
          va[j] = vb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (tin && ci < a.Cin && !(CONV_DIAG & 2)) va[j] = *reinterpret_cast<const float4*>(xb + (size_t)ci * a.x_cs + t4);
          if (tin && ci + 1 < a.Cin && !(CONV_DIAG & 2)) vb[j] = *reinterpret_cast<const float4*>(xb + (size_t)(ci + 1) * a.x_cs + t4);
        }
        if (q4 < LW4) {
          const int lim = a.in_mask ? (len < a.T_in ? len : a.T_in) : a.T_in;
#pragma unroll
          for (int j = 0; j < RWP; ++j) {
            float ea[4] = {va[j].x, va[j].y, va[j].z, va[j].w}, eb[4] = {vb[j].x, vb[j].y, vb[j].z, vb[j].w};
            unsigned wh4[4], wl4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float x0 = (t4 + u < lim) ? ea[u] : 0.f, x1 = (t4 + u < lim) ? eb[u] : 0.f;
              if (a.in_act) { x0 = x0 > 0.f ? x0 : x0 * a.in_slope; x1 = x1 > 0.f ? x1 : x1 * a.in_slope; }
              split_pair(x0, x1, wh4[u], wl4[u]);
            }
            const int o = (wave + j * NW) * LWP + 4 * q4;
            *reinterpret_cast<u32x4*>(ph + o) = u32x4{wh4[0], wh4[1], wh4[2], wh4[3]};
            *reinterpret_cast<u32x4*>(pl + o) = u32x4{wl4[0], wl4[1], wl4[2], wl4[3]};
          }
        }
      }
    } else if (F16S) {
      for (int cp = wave; cp < CONV_CK / 2; cp += NW) {
        const int ci = chunk * CONV_CK + 2 * cp;
        const float* xr0 = xb + (size_t)ci * a.x_cs;
        const float* xr1 = xr0 + a.x_cs;
        for (int col = lane; col < LW; col += 64) {
          const int t = t0 - a.pad + col;
          float x0 = 0.f, x1 = 0.f;
          if (t >= 0 && t < a.T_in && !(a.in_mask && t >= len)) {
            if (ci < a.Cin) x0 = xr0[t];
            if (ci + 1 < a.Cin) x1 = xr1[t];
            if (a.in_act) { x0 = x0 > 0.f ? x0 : x0 * a.in_slope; x1 = x1 > 0.f ? x1 : x1 * a.in_slope; }
          }
          unsigned wh1, wl1;
          split_pair(x0, x1, wh1, wl1);
          ph[cp * LWP + col] = wh1;
          pl[cp * LWP + col] = wl1;
        }
      }
    } else if (vec) {
      // 16-byte loads, one row per (wave, j) and all RW rows of a column block in flight at once;
      // the staged window starts at t_start = floor4(t0 - pad) so every load is 16-byte aligned.
      constexpr int RW = CONV_CK / NW;
      for (int q0 = 0; q0 < LW4; q0 += 64) {
        const int q4 = q0 + lane;
        const int t4 = t_start + 4 * q4;
        const bool tin = q4 < LW4 && t4 >= 0 && t4 < a.T_in;
        float4 v[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
          const int ci = chunk * CONV_CK + wave + j * NW;
          v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (tin && ci < a.Cin) v[j] = *reinterpret_cast<const float4*>(xb + (size_t)ci * a.x_cs + t4);
        }
        if (q4 < LW4) {
          const int lim = a.in_mask ? (len < a.T_in ? len : a.T_in) : a.T_in;
#pragma unroll
          for (int j = 0; j < RW; ++j) {
            float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float x = (t4 + u < lim) ? e[u] : 0.f;
              if (a.in_act) x = x > 0.f ? x : x * a.in_slope;
              e[u] = x;
            }
            *reinterpret_cast<float4*>(xs + (wave + j * NW) * LWP + 4 * q4) = make_float4(e[0], e[1], e[2], e[3]);
          }
        }
      }
    } else {
      for (int c = wave; c < CONV_CK; c += NW) {
        const int ci = chunk * CONV_CK + c;
        const bool cvalid = ci < a.Cin;
        const float* xr = xb + (size_t)ci * a.x_cs;
        float* dst = xs + c * LWP;
        for (int col = lane; col < LW; col += 64) {
          const int t = t0 - a.pad + col;
          float v = 0.f;
          if (cvalid && t >= 0 && t < a.T_in) {
            v = xr[t];
            if (a.in_mask && t >= len) v = 0.f;
            if (a.in_act) v = v > 0.f ? v : v * a.in_slope;
          }
          dst[col] = v;
        }
      }
    }
    if constexpr (!RING) __syncthreads();
    for (int tap = 0; tap < a.K; ++tap, ++it) {
      if constexpr (RING) {
        ring_wait(it);
        CONV_RAW_BARRIER();          // the step's weights (tap 0: and the window) are visible; step it - 1 is read out
        if (it + DEPTH < total_it) ring_dma(it + DEPTH);
        if constexpr (PREF) {
          if (pref && tap == 0 && chunk + 1 < a.nchunks) { st_load(chunk + 1); pf_step = it; }
        }
      } else {
        if (it + 1 < total_it) load_a(it + 1, a_nxt);
      }
      if constexpr (F16S) {
        // column of this lane in the pair images, rows 8*ks + 4*h + i
        const int col = wn * (NT * 32) + l31 + off + tap * a.dil;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 bh[NT], bl[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const unsigned* p = ph + (8 * ks + 4 * h) * LWP + col + nt * 32;
            const unsigned* q = pl + (8 * ks + 4 * h) * LWP + col + nt * 32;
            bh[nt] = __builtin_bit_cast(f16x8, u32x4{p[0], p[LWP], p[2 * LWP], p[3 * LWP]});
            bl[nt] = __builtin_bit_cast(f16x8, u32x4{q[0], q[LWP], q[2 * LWP], q[3 * LWP]});
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f16x8 ah, al;
            if constexpr (RING) {
              const char* pa = ring + (it % NS) * SB + (wm * MT + mt) * 4096 + ks * 2048 + lane * 16;
              ah = *reinterpret_cast<const f16x8*>(pa);
              al = *reinterpret_cast<const f16x8*>(pa + 1024);
            } else {
              ah = __builtin_bit_cast(f16x8, a_cur[mt][2 * ks]);
              al = __builtin_bit_cast(f16x8, a_cur[mt][2 * ks + 1]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = CONV_MFMA16(ah, bh[nt], acc[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) crs[mt][nt] = CONV_MFMA16(ah, bl[nt], crs[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) crs[mt][nt] = CONV_MFMA16(al, bh[nt], crs[mt][nt]);
          }
        }
      }
      const float* xt = xsb + tap * a.dil;
#pragma unroll
      for (int kg = 0; kg < (F16S ? 0 : KG); ++kg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = kg * 4 + i;
          float bv[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[nt] = xt[(2 * kk) * LWP + nt * 32];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float av = i == 0 ? a_cur[mt][kg].x : i == 1 ? a_cur[mt][kg].y : i == 2 ? a_cur[mt][kg].z
                                                                                         : a_cur[mt][kg].w;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nt], acc[mt][nt], 0, 0, 0);
          }
        }
      }
      if constexpr (!RING) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int kg = 0; kg < KG; ++kg) a_cur[mt][kg] = a_nxt[mt][kg];
      }
    }
    if constexpr (RING) CONV_RAW_BARRIER();   // the window is read out
    else __syncthreads();
  }

  if constexpr (F16S) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] += crs[mt][nt][r] * (1.f / 2048.f);
  }
  if constexpr ((CONV_DIAG & 4) != 0) {
    float keep = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) keep += acc[mt][nt][0];
    if (keep == 1.2345e-30f) a.out[0] = keep;
    return;
  }
  // ---- epilogue
  const float* resb = a.res ? a.res + (size_t)b * a.r_bs : nullptr;
  float* outb = a.out + (size_t)b * a.o_bs;
  const float* condb = a.cond ? a.cond + (size_t)b * a.cond_bs : nullptr;
  if (a.ups_s == 0) {
    // Plain (non-polyphase) store, the form every frame-rate convolution takes.  A lane owns 16 rows of a tile as four
    // groups of four CONSECUTIVE rows (8 g + 4 h + 0..3): bias / conditioning come as one 16-byte load per group,
    // addresses are 32-bit offsets into buffer descriptors, and every predicate (row < M, column < Nq, operand
    // present) selects the out-of-range offset instead of a branch (loads give 0, stores are dropped).
    constexpr int OOR = 0x7ffffff0;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, OOR, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resb ? resb : outb), 0, OOR, 0x00020000);
    const bool gate_ep = a.act == 2;
    auto row_vec = [&](const float* p, int row0, float (&v)[4]) {     // p[row0 .. row0 + 3], rows >= M read as 0
      v[0] = v[1] = v[2] = v[3] = 0.f;
      if (!p) return;
      if (row0 + 3 < a.M && (reinterpret_cast<uintptr_t>(p + row0) & 15) == 0) {
        const float4 t = *reinterpret_cast<const float4*>(p + row0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (row0 + i < a.M) v[i] = p[row0 + i];
      }
    };
    if (gate_ep) {
      // WN gate (reference commons.py:100-107): tiles (2i, 2i+1) hold the tanh / sigmoid halves.
      if constexpr (MT % 2 == 0) {
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
          const int mtile = mtile0 + 2 * mp;
          if (mtile + 1 >= n_mtiles) continue;
          float ba[4][4], bb[4][4], ca[4][4], cb[4][4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            row_vec(a.bias, mtile * 32 + 8 * g + 4 * h, ba[g]);
            row_vec(a.bias, mtile * 32 + 32 + 8 * g + 4 * h, bb[g]);
            row_vec(condb, mtile * 32 + 8 * g + 4 * h, ca[g]);
            row_vec(condb, mtile * 32 + 32 + 8 * g + 4 * h, cb[g]);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int q = t0 + (wn * NT + nt) * 32 + l31;
            const bool qin = q < a.Nq, valid = q < len;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int g = r >> 2, i = r & 3;
              float va = acc[2 * mp][nt][r], vb = acc[2 * mp + 1][nt][r];
              if (a.bias) { va += ba[g][i]; vb += bb[g][i]; }
              if (condb) { va += ca[g][i]; vb += cb[g][i]; }
              float v = tanhf(va) * (1.f / (1.f + expf(-vb)));
              if (a.mask_post && !valid) v = 0.f;
              const int orow = (mtile >> 1) * 32 + 8 * g + 4 * h + i;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, qin ? (orow * (int)a.o_cs + q) * 4 : OOR, 0, 0);
            }
          }
        }
      }
      return;
    }
    const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(
        a.split_row ? a.out2 + (size_t)b * a.o2_bs : outb, 0, OOR, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int mtile = mtile0 + mt;
      if (mtile >= n_mtiles) continue;
      float bv[4][4], cv[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        row_vec(a.bias, mtile * 32 + 8 * g + 4 * h, bv[g]);
        row_vec(condb, mtile * 32 + 8 * g + 4 * h, cv[g]);
      }
      if (a.split_row && mtile * 32 >= a.split_row) {
        // second destination: out2[row - split_row] = conv + bias (+ out2)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int q = t0 + (wn * NT + nt) * 32 + l31;
          const bool qin = q < a.Nq, keep = !a.mask_post2 || q < len;
          int oo[16];
          unsigned pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            oo[r] = (qin && row < a.M) ? ((row - a.split_row) * (int)a.o2_cs + q) * 4 : OOR;
            pv[r] = __builtin_amdgcn_raw_buffer_load_b32(ro2, a.acc_prev2 ? oo[r] : OOR, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mt][nt][r];
            if (a.bias) v += bv[r >> 2][r & 3];
            if (a.acc_prev2) v += __uint_as_float(pv[r]);
            if (!keep) v = 0.f;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro2, oo[r], 0, 0);
          }
        }
        continue;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = t0 + (wn * NT + nt) * 32 + l31;
        const bool qin = q < a.Nq, valid = q < len;
        // all loads of the tile first (res / out_prev may alias out), then the arithmetic and the stores
        int oo[16];
        unsigned rv[16], pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          const bool st = qin && row < a.M;
          oo[r] = st ? (row * (int)a.o_cs + q) * 4 : OOR;
          rv[r] = __builtin_amdgcn_raw_buffer_load_b32(rr, (st && resb) ? (row * (int)a.r_cs + q) * 4 : OOR, 0, 0);
          pv[r] = __builtin_amdgcn_raw_buffer_load_b32(ro, a.acc_prev ? oo[r] : OOR, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int g = r >> 2, i = r & 3;
          float v = acc[mt][nt][r];
          if (a.bias) v += bv[g][i];
          if (condb) v += cv[g][i];
          if (a.act == 1) v = fmaxf(v, 0.f);
          if (a.mask_pre && !valid) v = 0.f;
          if (a.alpha != 1.f) v *= a.alpha;
          if (resb) v += __uint_as_float(rv[r]);
          if (a.acc_prev) v += __uint_as_float(pv[r]);
          if (a.div != 1.f) v /= a.div;
          if (a.mask_post && !valid) v = 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, oo[r], 0, 0);
        }
      }
    }
    return;
  }
  if (a.act == 2) {
    // WN gate (reference commons.py:100-107): tiles (2i, 2i+1) hold the tanh / sigmoid halves.
    if constexpr (MT % 2 == 0) {
#pragma unroll
      for (int mp = 0; mp < MT / 2; ++mp) {
        const int mtile = mtile0 + 2 * mp;
        if (mtile + 1 >= n_mtiles) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int q = t0 + (wn * NT + nt) * 32 + l31;
          if (q >= a.Nq) continue;
          const bool valid = q < len;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ra = mtile * 32 + rin, rb = ra + 32;
            float va = acc[2 * mp][nt][r], vb = acc[2 * mp + 1][nt][r];
            if (a.bias) { va += a.bias[ra]; vb += a.bias[rb]; }
            if (condb) { va += condb[ra]; vb += condb[rb]; }
            float v = tanhf(va) * (1.f / (1.f + expf(-vb)));
            if (a.mask_post && !valid) v = 0.f;
            const int orow = (mtile >> 1) * 32 + rin;
            outb[(size_t)orow * a.o_cs + q] = v;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mtile = mtile0 + mt;
    if (mtile >= n_mtiles) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int q = t0 + (wn * NT + nt) * 32 + l31;
      if (q >= a.Nq) continue;
      const bool valid = q < len;
      // all loads of the tile first (res / out_prev may alias out: a load behind a store would
      // have to wait for that store), then the arithmetic and the stores
      size_t oidx[16];
      bool st[16];
      float rv[16], pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        st[r] = row < a.M;
        if (a.ups_s > 0) {
          const int co = row / a.ups_s, rr = row - co * a.ups_s;
          const int n = a.ups_s * q + rr - a.ups_p;
          st[r] = st[r] && n >= 0 && n < a.T_store;
          oidx[r] = (size_t)co * a.o_cs + n;
        } else {
          oidx[r] = (size_t)row * a.o_cs + q;
        }
        rv[r] = (resb && st[r]) ? resb[(size_t)row * a.r_cs + q] : 0.f;
        pv[r] = (a.acc_prev && st[r]) ? outb[oidx[r]] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!st[r]) continue;
        float v = acc[mt][nt][r];
        if (a.bias) v += a.bias[row];
        if (condb) v += condb[row];
        if (a.act == 1) v = fmaxf(v, 0.f);
        if (a.mask_pre && !valid) v = 0.f;
        if (a.alpha != 1.f) v *= a.alpha;
        if (resb) v += rv[r];
        if (a.acc_prev) v += pv[r];
        if (a.div != 1.f) v /= a.div;
        if (a.mask_post && !valid) v = 0.f;
        outb[oidx[r]] = v;
      }
    }
  }
}

template <int MT, int NT, int WM, int WN, bool F16S = false, bool RING = false>
static hipError_t launch_tile(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int BN = 32 * NT * WN, BM = 32 * MT * WM;
  constexpr size_t lds = (size_t)CONV_CK * (BN + CONV_HALO) * sizeof(float) +
                         (RING ? (size_t)conv_ring_slots(MT, WM) * MT * WM * 4096 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget (<= 80 KiB: two blocks per CU)");
  static bool attr_set = false;
  auto kern = conv1d_f32_mfma<MT, NT, WM, WN, F16S, RING>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid((a.Nq + BN - 1) / BN, (a.M + BM - 1) / BM, B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_conv(const ConvArgs& a, int B, hipStream_t s) {
  if ((a.K - 1) * a.dil + 3 > CONV_HALO || a.K < 1 || a.Nq <= 0 || B <= 0 || (a.split_row & 31) ||
      (a.split_row && (a.ups_s > 0 || a.act == 2 || !a.out2)))
    return hipErrorInvalidValue;
  const bool gate = a.act == 2;
  // (ring = 0: weights from global memory into registers, the round-1 form of these kernels; an experiment knob)
#ifdef VSP_EXPERIMENTS
  static int ring = -1;
  if (ring < 0) { const char* e = getenv("VSP_FRAME_RING"); ring = e ? atoi(e) != 0 : 1; }
#else
  constexpr bool ring = true;
#endif
  if (ring && a.f16s) {
    if (a.ups_s > 0) return hipErrorInvalidValue;
    if (a.M <= 32 && !gate) {
      if (a.Nq >= 1024) return launch_tile<1, 4, 1, 4, true, true>(a, B, s);
      return launch_tile<1, 1, 1, 2, true, true>(a, B, s);
    }
    if (a.Nq <= 96) return launch_tile<2, 1, 1, 2, true, true>(a, B, s);
    // M % 128 == 0: 128-row x 128-column tiles (two blocks per CU) for long time axes; utterance-sized ones take the
    // 64-row tile -- twice the blocks at three per CU fill the chip's rounds better than the halved window reuse
    // costs (C3 -0.15 ms, one utterance -11 %; the 5168-frame utterance +4 % if it did).
#ifdef VSP_EXPERIMENTS
    static int force_rows = -1;
    if (force_rows < 0) { const char* e = getenv("VSP_FRAME_TILE"); force_rows = e ? atoi(e) : 0; }
#else
    constexpr int force_rows = 0;
#endif
    const bool wide = force_rows ? force_rows == 128 : a.Nq >= 1024;
    if (a.M <= 64 || (a.M % 128 != 0 && a.M < 256) || !wide) {
      if (a.Nq >= 1024) return launch_tile<2, 2, 1, 4, true, true>(a, B, s);
      return launch_tile<2, 1, 1, 4, true, true>(a, B, s);
    }
    return launch_tile<2, 2, 2, 2, true, true>(a, B, s);
  }
  if (a.f16s) {
    // split-f16 path: two accumulators per tile, so at most MT*NT = 4 tiles per wave
    if (a.ups_s > 0) return hipErrorInvalidValue;
    if (a.M <= 32 && !gate) {
      if (a.Nq >= 1024) return launch_tile<1, 4, 1, 4, true>(a, B, s);
      return launch_tile<1, 1, 1, 2, true>(a, B, s);
    }
    if (a.Nq <= 96) return launch_tile<2, 1, 1, 2, true>(a, B, s);
    if (a.M <= 64 || (a.M % 128 != 0 && a.M < 256)) {
      if (a.Nq >= 1024) return launch_tile<2, 2, 1, 4, true>(a, B, s);
      return launch_tile<2, 1, 1, 4, true>(a, B, s);
    }
    return launch_tile<2, 2, 2, 2, true>(a, B, s);
  }
  if (a.M <= 32 && !gate) {
    if (a.Nq >= 1024) return launch_tile<1, 4, 1, 4>(a, B, s);
    return launch_tile<1, 1, 1, 2>(a, B, s);
  }
  if (a.Nq <= 96) return launch_tile<2, 1, 1, 2>(a, B, s);
  if (a.M <= 64 || (a.M % 128 != 0 && a.M < 256)) {
    if (a.Nq >= 1024) return launch_tile<2, 4, 1, 4>(a, B, s);
    return launch_tile<2, 1, 1, 4>(a, B, s);
  }
  return launch_tile<2, 2, 2, 2>(a, B, s);
}

}  // namespace vsp
