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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

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
// w * 2^8 = wh + wl exactly as in gen16.hip (round 5: unscaled lo parts, weights * G16_WSCALE, ONE accumulator).
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
        const float w = dense[((size_t)row * Cin + ci) * K + tap] * G16_WSCALE;   // (kernels.h: exact; keeps lo normal)
        const _Float16 h = (_Float16)w;
        const _Float16 l = (_Float16)(w - (float)h);
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
// MFMAs per product into ONE accumulator: unscaled lo parts, weights * 2^8, g16_common.h) -- 16/3 x the f32 matrix rate.
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

// Plain (non-polyphase) store of a block's accumulators, shared by the frame-rate kernels below, in two halves: every
// operand the epilogue reads (bias and conditioning rows, the residual, the accumulate-into destination) is loaded into
// ConvEpi by conv_epi_load -- the latency kernel does that FIRST, so that the loads' round trip runs under its main loop
// -- and conv_epi_store does the arithmetic and the stores.  A lane owns 16 rows of a tile as four groups of four
// CONSECUTIVE rows (8 g + 4 h + 0..3): bias / conditioning come as one 16-byte load per group, addresses are 32-bit
// offsets into buffer descriptors, and every predicate (row < M, column < Nq, operand present) selects the
// out-of-range offset instead of a branch (loads give 0, stores are dropped).
template <int MT, int NT>
struct ConvEpi {
  float bv[MT][16], cv[MT][16];          // bias / conditioning of the lane's rows
  unsigned rv[MT][NT][16], pv[MT][NT][16];   // residual, previous value of the destination
};
constexpr int CONV_OOR = 0x7ffffff0;

// HOIST (round 5, the latency kernels): one uniform branch per operand kind and m-tile with the element loops inside -- with
// the branches per ELEMENT conv_frame_f16s spent 3000 of its 10000 instructions in this function, 900 of them reloads of
// spilled scalar registers (one utterance 3.27 -> 3.17 ms).  The throughput kernel, which comes here once per 32 x 32
// tile, keeps the per-element form (the hoisted one costs it 0.15 ms on C3).
template <int MT, int NT, bool HOIST = true>
__device__ __forceinline__ void conv_epi_load(const ConvArgs& a, ConvEpi<MT, NT>& e, int mtile0, int n_mtiles, int t0, int wn,
                                              int l31, int h, int b, int what = 3) {   // what: 1 = rows, 2 = tiles
  const float* resb = a.res ? a.res + (size_t)b * a.r_bs : nullptr;
  float* outb = a.out + (size_t)b * a.o_bs;
  const float* condb = a.cond ? a.cond + (size_t)b * a.cond_bs : nullptr;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, CONV_OOR, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(resb ? resb : outb), 0, CONV_OOR, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(
      a.split_row ? a.out2 + (size_t)b * a.o2_bs : outb, 0, CONV_OOR, 0x00020000);
  auto row_vec = [&](const float* p, int row0, float* v) {     // p[row0 .. row0 + 3], rows >= M read as 0
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
  const bool gate_ep = a.act == 2;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mtile = mtile0 + mt;
    if (what & 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        row_vec(a.bias, mtile * 32 + 8 * g + 4 * h, &e.bv[mt][4 * g]);
        row_vec(condb, mtile * 32 + 8 * g + 4 * h, &e.cv[mt][4 * g]);
      }
    }
    if (!(what & 2)) continue;
    const bool second = a.split_row && mtile * 32 >= a.split_row;
    if constexpr (!HOIST) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = t0 + (wn * NT + nt) * 32 + l31;
        const bool qin = q < a.Nq && !gate_ep && mtile < n_mtiles;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          const bool st = qin && row < a.M;
          const int oo = st ? (row * (int)a.o_cs + q) * 4 : CONV_OOR;
          const int oo2 = st ? ((row - a.split_row) * (int)a.o2_cs + q) * 4 : CONV_OOR;
          // (uniform branches: a launch without a residual / an accumulating destination issues none of these)
          e.rv[mt][nt][r] = 0u;
          e.pv[mt][nt][r] = 0u;
          if (resb && !second) e.rv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(rr, st ? (row * (int)a.r_cs + q) * 4 : CONV_OOR, 0, 0);
          if (a.acc_prev && !second) e.pv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(ro, oo, 0, 0);
          if (a.acc_prev2 && second) e.pv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(ro2, oo2, 0, 0);
        }
      }
      continue;
    }
    const bool want_res = resb && !second, want_prev = a.acc_prev && !second, want_prev2 = a.acc_prev2 && second;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { e.rv[mt][nt][r] = 0u; e.pv[mt][nt][r] = 0u; }
    if (!(want_res || want_prev || want_prev2)) continue;
    const bool tile_in = !gate_ep && mtile < n_mtiles;
    if (want_res) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = t0 + (wn * NT + nt) * 32 + l31;
        const bool qin = q < a.Nq && tile_in;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          e.rv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(rr, (qin && row < a.M) ? (row * (int)a.r_cs + q) * 4 : CONV_OOR, 0, 0);
        }
      }
    }
    if (want_prev) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = t0 + (wn * NT + nt) * 32 + l31;
        const bool qin = q < a.Nq && tile_in;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          e.pv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(ro, (qin && row < a.M) ? (row * (int)a.o_cs + q) * 4 : CONV_OOR, 0, 0);
        }
      }
    }
    if (want_prev2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = t0 + (wn * NT + nt) * 32 + l31;
        const bool qin = q < a.Nq && tile_in;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mtile * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
          e.pv[mt][nt][r] = __builtin_amdgcn_raw_buffer_load_b32(ro2, (qin && row < a.M) ? ((row - a.split_row) * (int)a.o2_cs + q) * 4 : CONV_OOR, 0, 0);
        }
      }
    }
  }
}

// v / div as the reference divides (IEEE), behind a REAL scalar branch: hipcc if-converts `if (div != 1.f) v /= div` into
// an unconditional division + select (12 vector instructions and a quarter-rate v_rcp per value: 200 per 32 x 32 tile in
// every epilogue, although only the f32 generator's last ResBlock convolution divides).  The operand passes through an asm
// statement inside the block, which cannot be speculated.
__device__ __forceinline__ void conv_div(float& v, float div) {
  if (div != 1.f) {
    asm volatile("" : "+v"(v));
    v /= div;
  }
}

// (round 5: every optional step is ONE uniform branch around a 16-element loop, and full m-tiles address their rows through
// the buffer instruction's SCALAR offset -- written per element, with a multiply per address, the epilogue was ~200
// instructions per value in every frame-rate kernel.  Same operations in the same order per element: identical results.)
template <int MT, int NT>
__device__ __forceinline__ void conv_epi_store(const ConvArgs& a, f32x16 (&acc)[MT][NT], const ConvEpi<MT, NT>& e, int mtile0,
                                               int n_mtiles, int t0, int wn, int l31, int h, int b, int len) {
  float* outb = a.out + (size_t)b * a.o_bs;
  const bool has_cond = a.cond != nullptr, has_res = a.res != nullptr;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, CONV_OOR, 0x00020000);
  const int ocs4 = (int)a.o_cs * 4;
  // the 16 rows of a lane in tile rows [r0, r0 + 32): r0 + 4 h + 8 g + i -- full tiles: one vector offset + scalar row offsets
  auto store_tile = [&](const __amdgpu_buffer_rsrc_t& rs, const float (&v)[16], int row_first, int rows_end, int cs4, int q, bool qin) {
    if (row_first + 32 <= rows_end) {
      const int vo = qin ? (row_first + 4 * h) * cs4 + q * 4 : CONV_OOR;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs, vo, (8 * (r >> 2) + (r & 3)) * cs4, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_first + 8 * (r >> 2) + 4 * h + (r & 3);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs, (qin && row < rows_end) ? row * cs4 + q * 4 : CONV_OOR, 0, 0);
      }
    }
  };
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
          const bool qin = q < a.Nq, valid = q < len;
          float va[16], vb[16], v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) { va[r] = acc[2 * mp][nt][r]; vb[r] = acc[2 * mp + 1][nt][r]; }
          if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { va[r] += e.bv[2 * mp][r]; vb[r] += e.bv[2 * mp + 1][r]; }
          }
          if (has_cond) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { va[r] += e.cv[2 * mp][r]; vb[r] += e.cv[2 * mp + 1][r]; }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = tanhf(va[r]) * (1.f / (1.f + expf(-vb[r])));
          if (a.mask_post && !valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
          }
          // (output rows (mtile >> 1) * 32 + ..: every gated row exists -- the launcher requires Cout % 64 == 0)
          store_tile(ro, v, (mtile >> 1) * 32, (mtile >> 1) * 32 + 32, ocs4, q, qin);
        }
      }
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(
      a.split_row ? a.out2 + (size_t)b * a.o2_bs : outb, 0, CONV_OOR, 0x00020000);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mtile = mtile0 + mt;
    if (mtile >= n_mtiles) continue;
    const bool second = a.split_row && mtile * 32 >= a.split_row;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int q = t0 + (wn * NT + nt) * 32 + l31;
      const bool qin = q < a.Nq, valid = q < len;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[mt][nt][r];
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += e.bv[mt][r];
      }
      if (second) {
        // second destination: out2[row - split_row] = conv + bias (+ out2)
        if (a.acc_prev2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(e.pv[mt][nt][r]);
        }
        if (a.mask_post2 && !valid) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = 0.f;
        }
        store_tile(ro2, v, mtile * 32 - a.split_row, a.M - a.split_row, (int)a.o2_cs * 4, q, qin);
        continue;
      }
      if (has_cond) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += e.cv[mt][r];
      }
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.mask_pre && !valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
      }
      if (a.alpha != 1.f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= a.alpha;
      }
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(e.rv[mt][nt][r]);
      }
      if (a.acc_prev) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(e.pv[mt][nt][r]);
      }
      if (a.div != 1.f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(v[r])); v[r] /= a.div; }
      }
      if (a.mask_post && !valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
      }
      store_tile(ro, v, mtile * 32, a.M, ocs4, q, qin);
    }
  }
}

// LEAN epilogue of ONE 32 x 32 tile of the throughput kernel (round 5).  conv_epi_load / conv_epi_store evaluate every
// optional step of the epilogue per ELEMENT and compute every address with a multiply: ~200 instructions per value and
// 1500 reloads of spilled scalar registers in conv1d_f32_mfma<2,1,1,4> (9700 vector instructions around 24 MFMAs).  Here
// the lane's address is computed once (the row offsets are uniform: they ride in the buffer instruction's scalar offset),
// and every optional step is ONE uniform branch around a 16-element loop.  Same operations in the same order per
// element as conv_epi_store's plain path: identical results.  Tiles that need the gate, the second destination or a
// partial last m-tile keep the general path.
__device__ __forceinline__ bool conv_tile_lean_ok(const ConvArgs& a, int mtile) {
  return a.act != 2 && (a.split_row & 31) == 0 && mtile * 32 + 32 <= a.M && (a.M & 3) == 0 &&
         (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
         (!a.cond || ((reinterpret_cast<uintptr_t>(a.cond) & 15) == 0 && (a.cond_bs & 3) == 0));
}
__device__ __forceinline__ void conv_tile_lean(const ConvArgs& a, f32x16& acc, int mtile, int tq, int l31, int h, int b, int len) {
  const int q = tq + l31;
  const bool qin = q < a.Nq, valid = q < len;
  const int row0 = mtile * 32 + 4 * h;                       // the lane's rows: row0 + 8 g + i
  float* outb = a.out + (size_t)b * a.o_bs;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, CONV_OOR, 0x00020000);
  const int ocs4 = (int)a.o_cs * 4;
  const int vo = qin ? (row0 * (int)a.o_cs + q) * 4 : CONV_OOR;   // + (8 g + i) o_cs 4 as the SCALAR offset of the instruction
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = acc[r];
  if (a.bias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(a.bias + row0 + 8 * g);
      v[4 * g] += t.x; v[4 * g + 1] += t.y; v[4 * g + 2] += t.z; v[4 * g + 3] += t.w;
    }
  }
  if (a.split_row && mtile * 32 >= a.split_row) {
    // second destination (conv_epi_store): out2[row - split_row] = conv + bias (+ out2), masked by mask_post2
    const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(a.out2 + (size_t)b * a.o2_bs, 0, CONV_OOR, 0x00020000);
    const int o2cs4 = (int)a.o2_cs * 4;
    const int vo2 = qin ? ((row0 - a.split_row) * (int)a.o2_cs + q) * 4 : CONV_OOR;
    if (a.acc_prev2) {
      unsigned pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_raw_buffer_load_b32(ro2, vo2, (8 * (r >> 2) + (r & 3)) * o2cs4, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(pv[r]);
    }
    if (a.mask_post2 && !valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro2, vo2, (8 * (r >> 2) + (r & 3)) * o2cs4, 0);
    return;
  }
  if (a.cond) {
    const float* condb = a.cond + (size_t)b * a.cond_bs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(condb + row0 + 8 * g);
      v[4 * g] += t.x; v[4 * g + 1] += t.y; v[4 * g + 2] += t.z; v[4 * g + 3] += t.w;
    }
  }
  if (a.act == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
  }
  if (a.mask_pre && !valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
  }
  if (a.alpha != 1.f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] *= a.alpha;
  }
  if (a.res) {
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res + (size_t)b * a.r_bs), 0, CONV_OOR, 0x00020000);
    const int rcs4 = (int)a.r_cs * 4;
    const int vr = qin ? (row0 * (int)a.r_cs + q) * 4 : CONV_OOR;
    unsigned rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = __builtin_amdgcn_raw_buffer_load_b32(rr, vr, (8 * (r >> 2) + (r & 3)) * rcs4, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(rv[r]);
  }
  if (a.acc_prev) {
    unsigned pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_raw_buffer_load_b32(ro, vo, (8 * (r >> 2) + (r & 3)) * ocs4, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += __uint_as_float(pv[r]);
  }
  if (a.div != 1.f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(v[r])); v[r] /= a.div; }
  }
  if (a.mask_post && !valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, vo, (8 * (r >> 2) + (r & 3)) * ocs4, 0);
}

// ... and of one tanh / sigmoid tile PAIR of the WN gate (reference commons.py:100-107; conv_epi_store's gate path)
__device__ __forceinline__ bool conv_gate_lean_ok(const ConvArgs& a, int mtile, int n_mtiles) {
  return a.act == 2 && mtile + 1 < n_mtiles && (a.M & 3) == 0 && mtile * 32 + 64 <= a.M &&
         (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
         (!a.cond || ((reinterpret_cast<uintptr_t>(a.cond) & 15) == 0 && (a.cond_bs & 3) == 0));
}
__device__ __forceinline__ void conv_gate_lean(const ConvArgs& a, f32x16& acc_t, f32x16& acc_s, int mtile, int tq, int l31, int h, int b,
                                               int len) {
  const int q = tq + l31;
  const bool qin = q < a.Nq, valid = q < len;
  const int row0 = mtile * 32 + 4 * h;                       // tanh rows row0 + 8 g + i, sigmoid rows 32 further
  float* outb = a.out + (size_t)b * a.o_bs;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, CONV_OOR, 0x00020000);
  const int ocs4 = (int)a.o_cs * 4;
  const int vo = qin ? (((mtile >> 1) * 32 + 4 * h) * (int)a.o_cs + q) * 4 : CONV_OOR;
  float va[16], vb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { va[r] = acc_t[r]; vb[r] = acc_s[r]; }
  if (a.bias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(a.bias + row0 + 8 * g), u = *reinterpret_cast<const float4*>(a.bias + row0 + 32 + 8 * g);
      va[4 * g] += t.x; va[4 * g + 1] += t.y; va[4 * g + 2] += t.z; va[4 * g + 3] += t.w;
      vb[4 * g] += u.x; vb[4 * g + 1] += u.y; vb[4 * g + 2] += u.z; vb[4 * g + 3] += u.w;
    }
  }
  if (a.cond) {
    const float* condb = a.cond + (size_t)b * a.cond_bs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(condb + row0 + 8 * g), u = *reinterpret_cast<const float4*>(condb + row0 + 32 + 8 * g);
      va[4 * g] += t.x; va[4 * g + 1] += t.y; va[4 * g + 2] += t.z; va[4 * g + 3] += t.w;
      vb[4 * g] += u.x; vb[4 * g + 1] += u.y; vb[4 * g + 2] += u.z; vb[4 * g + 3] += u.w;
    }
  }
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = tanhf(va[r]) * (1.f / (1.f + expf(-vb[r])));
  if (a.mask_post && !valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, vo, (8 * (r >> 2) + (r & 3)) * ocs4, 0);
}

// (the throughput kernel, at two blocks per CU: one tile -- for the gate, one pair of tiles -- at a time)
template <int S, int MT, int NT>
__device__ __forceinline__ void conv_store_slices(const ConvArgs& a, f32x16 (&acc)[MT][NT], int mtile0, int n_mtiles, int t0,
                                                  int wn, int l31, int h, int b, int len) {
#pragma unroll
  for (int ms = 0; ms < MT / S; ++ms)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int tq = t0 + (wn * NT + nt) * 32;
      if constexpr (S == 1) {
        const int mtile = mtile0 + ms;
        if (mtile < n_mtiles && conv_tile_lean_ok(a, mtile)) {      // (uniform)
          conv_tile_lean(a, acc[ms][nt], mtile, tq, l31, h, b, len);
          continue;
        }
      } else if constexpr (S == 2) {
        const int mtile = mtile0 + 2 * ms;
        if (conv_gate_lean_ok(a, mtile, n_mtiles)) {                 // (uniform)
          conv_gate_lean(a, acc[2 * ms][nt], acc[2 * ms + 1][nt], mtile, tq, l31, h, b, len);
          continue;
        }
      }
      f32x16 t[S][1];
#pragma unroll
      for (int u = 0; u < S; ++u) t[u][0] = acc[S * ms + u][nt];
      ConvEpi<S, 1> e;
      conv_epi_load<S, 1, false>(a, e, mtile0 + S * ms, n_mtiles, tq, 0, l31, h, b);
      conv_epi_store<S, 1>(a, t, e, mtile0 + S * ms, n_mtiles, tq, 0, l31, h, b, len);
    }
}
template <int MT, int NT>
__device__ __forceinline__ void conv_store_plain(const ConvArgs& a, f32x16 (&acc)[MT][NT], int mtile0, int n_mtiles, int t0,
                                                 int wn, int l31, int h, int b, int len) {
  if (a.act == 2) {
    if constexpr (MT % 2 == 0) conv_store_slices<2, MT, NT>(a, acc, mtile0, n_mtiles, t0, wn, l31, h, b, len);
  } else {
    conv_store_slices<1, MT, NT>(a, acc, mtile0, n_mtiles, t0, wn, l31, h, b, len);
  }
}

// PRUNE (round 5): the instantiation for what the frame-rate path launches almost always -- split-f16, weight ring, the
// prefetched window path, no transposed-convolution epilogue -- WITHOUT the code of the four other window stagings and of
// the polyphase epilogue: the launcher checks what those branches test at run time.
template <int MT, int NT, int WM, int WN, bool F16S, bool RING = false, bool PRUNE = false>
__global__ void __launch_bounds__(64 * WM * WN, F16S ? 2 : 1) conv1d_f32_mfma(ConvArgs a) {
  static_assert(!RING || F16S, "the weight ring serves the split-f16 path");
  static_assert(!PRUNE || (F16S && RING), "the pruned form is the split-f16 ring kernel");
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

  f32x16 acc[MT][NT];                           // F16S: HH, CROSS, CROSS into the one register set (g16_common.h, round 5)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mt][nt][r] = 0.f;
      }
  unsigned* const ph = reinterpret_cast<unsigned*>(xs);          // F16S: hi image [16][LWP] words
  unsigned* const pl = ph + (CONV_CK / 2) * LWP;                 //       lo image
  // hi = the fp32 values rounded to f16 (inf beyond the f16 range), lo = the exact residual (kernels.h: vsp_split_pair)
  auto split_pair = [&](float x0, float x1, unsigned& whi, unsigned& wlo) { vsp_split_pair(x0, x1, whi, wlo); };

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
        vsp_split_pair(x.x, x.y, wh4[u], wl4[u]);       // (kernels.h; unscaled lo parts: ONE accumulator)
      }
      const int o = (wave + j * NW) * LWP + 4 * lane;
      *reinterpret_cast<u32x4*>(ph + o) = u32x4{wh4[0], wh4[1], wh4[2], wh4[3]};
      *reinterpret_cast<u32x4*>(pl + o) = u32x4{wl4[0], wl4[1], wl4[2], wl4[3]};
    }
  };
  static_assert(!PRUNE || PREF, "the pruned form stages its window through the prefetch path");
  const bool pref = PRUNE || (PREF && vec && (CONV_DIAG & 2) == 0 && slope_eff >= 0.f && slope_eff <= 1.f);
  if constexpr (PREF) { if (pref) st_load(0); }
  int it = 0;
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    // ---- stage x[chunk] -> LDS with the prologue applied
    if (pref) {
      if constexpr (PREF) st_write(chunk);      // (the compiler waits for the window here: a chunk after its request)
    } else if constexpr (PRUNE) {
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
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = CONV_MFMA16(ah, bl[nt], acc[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = CONV_MFMA16(al, bh[nt], acc[mt][nt]);
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
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= G16_UNSCALE;       // (the weights are packed * G16_WSCALE)
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
  if (PRUNE || a.ups_s == 0) {
    conv_store_plain<MT, NT>(a, acc, mtile0, n_mtiles, t0, wn, l31, h, b, len);
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
        conv_div(v, a.div);
        if (a.mask_post && !valid) v = 0.f;
        outb[oidx[r]] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_frame_f16s: the LATENCY form of the split-f16 ring kernel above, for grids that do not fill the chip (one utterance,
// or one long one: a few dozen blocks whose K loop is a chain of dependent steps).  There a launch costs
// (steps) x (what a step serialises), and in-kernel stamps of the kernel above put a step at 0.5-0.8 us against 0.16 us
// of matrix issue: a counted wait and a barrier per tap with the fragment reads exposed behind it, ~1 us of conversion
// per chunk on the critical path, the epilogue's two dependent memory round trips (5-6 us) at the end.  Same tiles,
// fragments, packing and epilogue arithmetic; what differs:
//   * the tap count K is a template parameter and the unit of synchronisation is an ITERATION of G chunks (G K steps):
//     one counted wait and one barrier per iteration, its taps unrolled behind it (fragment reads of a tap run under the
//     MFMAs of the tap before);
//   * no window data in registers: the fp32 window of a chunk is copied by LDS-DMA into a window slot, WI - 1 iterations
//     ahead, laid out so that the two 16-byte pieces a lane converts -- four times of channels 2p and 2p+1 -- sit exactly
//     where that lane's hi and lo image words go: the conversion (mask, leaky-relu, truncating split) is IN PLACE, one
//     iteration ahead of its use;
//   * the weights of an iteration are copied RI - 1 iterations ahead into a ring of RI iteration slots;
//   * every operand of the epilogue is requested before anything else: its round trip runs under the main loop.
// Every wait is a counted vmcnt: the copies of a wave complete in issue order, and the number issued behind the one
// waited for is known (the issue order is W(0..RI-2), X(0..WI-2), then per iteration i: W(i+RI-1), X(i+WI-1)).
constexpr int FR_HALO = 12;        // K > 1: LWP = BN + FR_HALO >= BN + (K-1) dil + 3 (alignment) + 3 (float4 round-up)

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate); n above the 6-bit field waits for 63
#define FR_VM1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define FR_VM8(a, b, c, d, e, f, g, h) FR_VM1(a) FR_VM1(b) FR_VM1(c) FR_VM1(d) FR_VM1(e) FR_VM1(f) FR_VM1(g) FR_VM1(h)
#ifndef FR_DIAG
#define FR_DIAG 0          // timing-only ablations (results wrong): 1 no MFMA, 2 no counted waits, 4 no conversion, 8 no window copies
#endif
__device__ __forceinline__ void fr_vmcnt(int n) {
  if constexpr ((FR_DIAG & 2) != 0) return;
  switch (n) {
    FR_VM8(0, 1, 2, 3, 4, 5, 6, 7) FR_VM8(8, 9, 10, 11, 12, 13, 14, 15) FR_VM8(16, 17, 18, 19, 20, 21, 22, 23)
    FR_VM8(24, 25, 26, 27, 28, 29, 30, 31) FR_VM8(32, 33, 34, 35, 36, 37, 38, 39) FR_VM8(40, 41, 42, 43, 44, 45, 46, 47)
    FR_VM8(48, 49, 50, 51, 52, 53, 54, 55) FR_VM8(56, 57, 58, 59, 60, 61, 62, 63)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
}
// FR_STAMPS (diagnostic build, tools/stamps_frame.py): wave 0 of every block of the VSP_STAMP_FRAME-th conv_frame_f16s
// launch records tagged wall-clock stamps (s_memrealtime, 100 MHz): 1 start | 2 requests out | 3 first window in |
// 4 first window converted | per iteration: 10 top, 11 waited, 12 barrier, 13 copies issued, 14 MFMAs issued, 15 next
// windows converted | 30 loop done | 31 stores issued | 32 retired.
#ifdef FR_STAMPS
constexpr int FR_NSTAMP = 512, FR_NSAMPLE = 64;
__device__ unsigned long long g_fr_stamps[FR_NSAMPLE][FR_NSTAMP];
__device__ unsigned g_fr_stamp_count;
__device__ int g_fr_stamp_on;
#define FR_STAMPT(tag)                                                                  \
  do {                                                                                  \
    if (stamp_slot >= 0 && stamp_n < FR_NSTAMP && lane == 0)                            \
      g_fr_stamps[stamp_slot][stamp_n] = (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull) | ((unsigned long long)(tag) << 56); \
    ++stamp_n;                                                                          \
  } while (0)
extern "C" int vsp_debug_stamps_frame(unsigned long long* host, int max_samples, int reset) {
  unsigned n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fr_stamp_count), sizeof n);
  if ((int)n > max_samples) n = max_samples;
  if (n > (unsigned)FR_NSAMPLE) n = FR_NSAMPLE;
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fr_stamps), (size_t)n * FR_NSTAMP * sizeof(unsigned long long));
  if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fr_stamp_count), &z, sizeof z); }
  return (int)n;
}
#else
#define FR_STAMPT(tag) ((void)0)
#endif
#if FR_DIAG & 1
#define FR_MFMA16(a, b, c) ([&]() { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define FR_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

template <int MT, int NT, int WM, int WN, int K, int G, int RI, int WI>
__global__ void __launch_bounds__(64 * WM * WN, 1) conv_frame_f16s(ConvArgs a) {
  constexpr int HALO = K == 1 ? 0 : FR_HALO;                    // (a 1x1 has no halo and no padding: its window is aligned)
  constexpr int BN = 32 * NT * WN, LWP = BN + HALO, LWP4 = LWP / 4, NW = WM * WN;
  constexpr int SB = MT * WM * 4096, PW = MT * WM * 4 / NW;     // bytes of one step's weights; 1 KiB pieces per wave and step
  constexpr int SI = G * K;                                     // steps of an iteration
  constexpr int NF4 = CONV_CK * LWP4;                           // 16-byte positions of a window slot: hi image | lo image
  constexpr int NPF = (NF4 + 64 * NW - 1) / (64 * NW);          // 1 KiB window pieces per wave and chunk
  constexpr int WSLOT = NPF * NW * 1024;                        // bytes of a window slot
  constexpr int RWPF = (CONV_CK / 2) / NW;                      // channel pairs a wave converts
  static_assert((MT * WM * 4) % NW == 0 && (CONV_CK / 2) % NW == 0, "pieces and channel pairs divide over the waves");
  static_assert(LWP % 4 == 0 && LWP4 <= 64, "one conversion sweep");
  static_assert(WI >= 3 && RI >= 2, "a window is converted one iteration after it landed, one before its use");
  extern __shared__ __attribute__((aligned(16))) float xs[];
  char* const wbase = reinterpret_cast<char*>(xs);
  char* const ring = wbase + WI * G * WSLOT;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * BN;
  const int mtile0 = (blockIdx.y * WM + wm) * MT;
  const int n_mtiles = (a.M + 31) >> 5;
  const int nch = a.nchunks, n_iter = (nch + G - 1) / G;
  const int total_it = nch * K;
#ifdef FR_STAMPS
  int stamp_slot = -1, stamp_n = 0;
  if (wave == 0 && g_fr_stamp_on) {
    unsigned sl_ = 0;
    if (lane == 0) sl_ = atomicAdd(&g_fr_stamp_count, 1u);
    sl_ = __builtin_amdgcn_readfirstlane(sl_);
    stamp_slot = sl_ < (unsigned)FR_NSAMPLE ? (int)sl_ : -1;
  }
  FR_STAMPT(1);
#endif
  // the epilogue's operands first: older than every copy below, they never enter the counted waits
  ConvEpi<MT, NT> epi;
  conv_epi_load<MT, NT>(a, epi, mtile0, n_mtiles, t0, wn, l31, h, b);
  FR_STAMPT(5);

  const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
  // weights of iteration j -> ring slot j % RI: step (g, tap) of the iteration at (g K + tap) SB
  auto w_dma = [&](int j) {
    char* const dst0 = ring + (j % RI) * (SI * SB);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = j * G + g;
      if (c >= nch) break;
#pragma unroll
      for (int tap = 0; tap < K; ++tap) {
#pragma unroll
        for (int u = 0; u < PW; ++u) {
          const int p = u * NW + wave;
          int mtile = blockIdx.y * (WM * MT) + (p >> 2);
          mtile = mtile < n_mtiles ? mtile : n_mtiles - 1;     // (m-tiles past the last re-read it: never stored)
          const float4* src = wp4 + (((size_t)mtile * total_it + c * K + tap) * (CONV_CK / 8) + (p & 3)) * 64 + lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst0 + (g * K + tap) * SB + p * 1024), 16, 0, 0);
        }
      }
    }
  };
  // window geometry (16-byte aligned rows: the launcher checks)
  const float* xb = a.x + (size_t)b * a.x_bs;
  const int LW = BN + (K - 1) * a.dil;
  const int t_start = ((t0 - a.pad) >> 2) << 2;
  const int off = (t0 - a.pad) - t_start;
  const int LW4 = (LW + off + 3) >> 2;
  // piece u of a chunk's window, lane l -> slot position g = (u NW + wave) 64 + l = (image, pair row p, four times c4):
  // channel 2p + image of the chunk.  Positions outside the window (and channels past Cin) copy x[b][0][0..3]: the
  // conversion zeroes what is not there.
  long woff[NPF];
  int wch[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    const int g = (u * NW + wave) * 64 + lane;
    const int img = g / (16 * LWP4), p = (g / LWP4) % 16, c4 = g % LWP4;
    const int t = t_start + 4 * c4;
    const bool ok = g < NF4 && c4 < LW4 && t >= 0 && t < a.T_in;
    wch[u] = ok ? 2 * p + img : 0x40000000;
    woff[u] = (long)(2 * p + img) * a.x_cs + t;
  }
  auto x_dma = [&](int j) {
    if constexpr ((FR_DIAG & 8) != 0) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = j * G + g;
      if (c >= nch) break;
      char* const dst = wbase + ((j % WI) * G + g) * WSLOT;
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const bool ok = c * CONV_CK + wch[u] < a.Cin;
        const float* src = ok ? xb + (size_t)c * CONV_CK * a.x_cs + woff[u] : xb;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + (u * NW + wave) * 1024), 16, 0, 0);
      }
    }
  };
  // requests: weights of the first RI - 1 iterations, windows of the first WI - 1
#pragma unroll
  for (int j = 0; j < RI - 1; ++j) w_dma(j);
  FR_STAMPT(6);
#pragma unroll
  for (int j = 0; j < WI - 1; ++j) x_dma(j);
  FR_STAMPT(2);

  // in-place prologue of the conv (mask, leaky-relu as max(x, slope x)) and truncating split (gen16.hip: g16_split2) of
  // the four times x two channels of a lane: fp32 rows 2p | 2p+1 in, hi | lo image words out, the same two 16-byte places
  const int len = a.lengths ? (int)a.lengths[b] : 0x7fffffff;
  const int t4 = t_start + 4 * lane;
  const bool tin = lane < LW4 && t4 >= 0 && t4 < a.T_in;
  const int lim = a.in_mask ? (len < a.T_in ? len : a.T_in) : a.T_in;
  const bool tfull = tin && t4 + 3 < lim;
  const float slope_eff = a.in_act ? a.in_slope : 1.f;
  // (a block whose whole window lies inside the valid times, on a full chunk of channels, with no activation -- most
  // blocks of most launches -- has nothing to mask or clamp: the conversion is then six instructions per column pair)
  const bool win_inside = t_start >= 0 && t_start + 4 * LW4 <= lim;
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  auto convert = [&](int j) {                 // the windows of iteration j
    if constexpr ((FR_DIAG & 4) != 0) return;
    if (lane >= LW4) return;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int chunk = j * G + g;
      if (chunk >= nch) break;
      char* const slot = wbase + ((j % WI) * G + g) * WSLOT;
      f32x4v va[RWPF], vb[RWPF];
#pragma unroll
      for (int q = 0; q < RWPF; ++q) {
        const int p = wave + q * NW;
        va[q] = *reinterpret_cast<const f32x4v*>(slot + (p * LWP4 + lane) * 16);
        vb[q] = *reinterpret_cast<const f32x4v*>(slot + ((16 + p) * LWP4 + lane) * 16);
      }
      const bool plain = win_inside && (chunk + 1) * CONV_CK <= a.Cin && slope_eff == 1.f;
#pragma unroll
      for (int q = 0; q < RWPF; ++q) {
        const int p = wave + q * NW;
        unsigned wh4[4], wl4[4];
        if (plain) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const f32x2v x = {va[q][u], vb[q][u]};
            vsp_split_pair(x.x, x.y, wh4[u], wl4[u]);       // (kernels.h; unscaled lo parts: ONE accumulator)
          }
        } else {
          const int ci = chunk * CONV_CK + 2 * p;
          const bool ina = tin && ci < a.Cin, inb = tin && ci + 1 < a.Cin;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            f32x2v x = {va[q][u], vb[q][u]};
            if (!(tfull && inb)) {                          // window edge / ragged channel count: zero what is not there
              x.x = (ina && t4 + u < lim) ? x.x : 0.f;
              x.y = (inb && t4 + u < lim) ? x.y : 0.f;
            }
            const f32x2v y = x * slope_eff;
            asm("v_max_f32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(y.x));
            asm("v_max_f32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(y.y));
            vsp_split_pair(x.x, x.y, wh4[u], wl4[u]);       // (kernels.h; unscaled lo parts: ONE accumulator)
          }
        }
        *reinterpret_cast<u32x4*>(slot + (p * LWP4 + lane) * 16) = u32x4{wh4[0], wh4[1], wh4[2], wh4[3]};
        *reinterpret_cast<u32x4*>(slot + ((16 + p) * LWP4 + lane) * 16) = u32x4{wl4[0], wl4[1], wl4[2], wl4[3]};
      }
    }
  };
  // copies a wave issues for iteration j (0 past the end): weights, windows
  auto cin = [&](int j) { const int r = nch - j * G; return j < n_iter ? (r < G ? r : G) : 0; };
  auto Wn = [&](int j) { return cin(j) * (K * PW); };
  auto Xn = [&](int j) { return cin(j) * NPF; };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  {  // the first iteration's windows: behind them, the windows of iterations 1 .. WI - 2
    int n = 0;
#pragma unroll
    for (int j = 1; j < WI - 1; ++j) n += Xn(j);
    fr_vmcnt(n);
  }
  CONV_RAW_BARRIER();
  FR_STAMPT(3);
  convert(0);
  FR_STAMPT(4);

  for (int i = 0; i < n_iter; ++i) {
    FR_STAMPT(10);
    {
      // my pieces of W(i) and, for the conversion below, of X(i + 1): the copies issued behind each
      int nw = 0;
      if (i <= RI - 2) {
#pragma unroll
        for (int j = 1; j < RI - 1; ++j) nw += j > i ? Wn(j) : 0;
#pragma unroll
        for (int j = 0; j < WI - 1; ++j) nw += Xn(j);
        for (int j = 0; j < i; ++j) nw += Wn(j + RI - 1) + Xn(j + WI - 1);
      } else {
        const int j0 = i - RI + 1;
        nw = Xn(j0 + WI - 1);
#pragma unroll
        for (int d = 1; d < RI - 1; ++d) nw += Wn(j0 + d + RI - 1) + Xn(j0 + d + WI - 1);      // j = j0 + 1 .. i - 1
      }
      int n = nw;
      if (i + 1 < n_iter) {
        int nx = 0;
        if (i + 1 <= WI - 2) {
#pragma unroll
          for (int j = 2; j < WI - 1; ++j) nx += j > i + 1 ? Xn(j) : 0;
          for (int j = 0; j < i; ++j) nx += Wn(j + RI - 1) + Xn(j + WI - 1);
        } else {
          const int j1 = i - WI + 2;
#pragma unroll
          for (int d = 1; d < WI - 2; ++d) nx += Wn(j1 + d + RI - 1) + Xn(j1 + d + WI - 1);    // j = j1 + 1 .. i - 1
        }
        n = nx < n ? nx : n;
      }
      fr_vmcnt(n);
    }
    FR_STAMPT(11);
    CONV_RAW_BARRIER();            // this iteration's weights and converted windows are visible; iteration i - 1 is read out
    FR_STAMPT(12);
    w_dma(i + RI - 1);             // (into the slots iteration i - 1 was read from)
    x_dma(i + WI - 1);
    FR_STAMPT(13);
    const char* const rbase = ring + (i % RI) * (SI * SB) + wm * (MT * 4096) + lane * 16;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (i * G + g >= nch) break;
      const unsigned* const ph = reinterpret_cast<const unsigned*>(wbase + ((i % WI) * G + g) * WSLOT);
      const unsigned* const pl = ph + (CONV_CK / 2) * LWP;
#pragma unroll
      for (int tap = 0; tap < K; ++tap) {
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
            const char* pa = rbase + (g * K + tap) * SB + mt * 4096 + ks * 2048;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pa);
            const f16x8 al = *reinterpret_cast<const f16x8*>(pa + 1024);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(ah, bh[nt], acc[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(ah, bl[nt], acc[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(al, bh[nt], acc[mt][nt]);
          }
        }
      }
    }
    FR_STAMPT(14);
    if (i + 1 < n_iter) { convert(i + 1); FR_STAMPT(15); }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= G16_UNSCALE;       // (the weights are packed * G16_WSCALE)
  FR_STAMPT(30);
  conv_epi_store<MT, NT>(a, acc, epi, mtile0, n_mtiles, t0, wn, l31, h, b, len);
#ifdef FR_STAMPS
  FR_STAMPT(31);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FR_STAMPT(32);
#endif
}

#ifdef FR_STAMPS
static int g_fr_launch_no = 0;
static const int g_fr_on_vals[2] = {0, 1};
#endif
template <int MT, int NT, int WM, int WN, int K, int G, int RI, int WI>
static hipError_t launch_frame(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int BN = 32 * NT * WN, BM = 32 * MT * WM, NW = WM * WN;
  constexpr int NF4 = CONV_CK * ((BN + (K == 1 ? 0 : FR_HALO)) / 4);
  constexpr size_t lds = (size_t)WI * G * ((NF4 + 64 * NW - 1) / (64 * NW)) * NW * 1024 + (size_t)RI * G * K * MT * WM * 4096;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = conv_frame_f16s<MT, NT, WM, WN, K, G, RI, WI>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  dim3 grid((a.Nq + BN - 1) / BN, (a.M + BM - 1) / BM, B);
#ifdef FR_STAMPS
  {  // stamps only in the VSP_STAMP_FRAME-th launch (0-based, counted over all conv_frame_f16s launches)
    static int target = -2;
    if (target == -2) { const char* e = getenv("VSP_STAMP_FRAME"); target = e ? atoi(e) : -1; }
    const int on = g_fr_launch_no++ == target ? 1 : 0;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_fr_stamp_on), &g_fr_on_vals[on], sizeof(int), 0, hipMemcpyHostToDevice, s);
    if (on) fprintf(stderr, "[stamps] conv_frame_f16s<%d,%d,%d,%d,K%d,G%d,%d,%d> M %d Cin %d Nq %d grid %u x %u x %u\n", MT, NT, WM, WN, K,
                    G, RI, WI, a.M, a.Cin, a.Nq, grid.x, grid.y, grid.z);
  }
#endif
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}
// the latency form for a tile shape, by tap count (other tap counts: the throughput kernel)
template <int MT, int NT, int WM, int WN>
static bool launch_frame_k(const ConvArgs& a, int B, hipStream_t s, hipError_t& e) {
  switch (a.K) {
    case 1: e = launch_frame<MT, NT, WM, WN, 1, 2, 3, 3>(a, B, s); return true;
    case 3: e = launch_frame<MT, NT, WM, WN, 3, 1, 3, 3>(a, B, s); return true;
    case 5: e = launch_frame<MT, NT, WM, WN, 5, 1, 2, 3>(a, B, s); return true;
    default: return false;
  }
}

template <int MT, int NT, int WM, int WN, bool F16S = false, bool RING = false, bool PRUNE = false>
static hipError_t launch_tile(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int BN = 32 * NT * WN, BM = 32 * MT * WM;
  constexpr size_t lds = (size_t)CONV_CK * (BN + CONV_HALO) * sizeof(float) +
                         (RING ? (size_t)conv_ring_slots(MT, WM) * MT * WM * 4096 : 0);
  static_assert(lds <= 160 * 1024, "LDS budget (<= 80 KiB: two blocks per CU)");
  static std::atomic<uint64_t> attr_done{0};
  auto kern = conv1d_f32_mfma<MT, NT, WM, WN, F16S, RING, PRUNE>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  dim3 grid((a.Nq + BN - 1) / BN, (a.M + BM - 1) / BM, B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_frame_splitk: the form for grids of a FEW DOZEN blocks (one utterance).  There even the kernel above is a chain:
// its four waves walk all K Cin / 32 steps of the tile together (stamps: 41 us for the 30 steps of a k5 WN layer, of
// which 23 us are the steps).  Here the four waves of a block split the INPUT CHANNELS: wave w takes chunks w, w + 4, ...
// of a 64 x 64 tile, with its own window slots, its own weights and no barrier at all until the four partial tiles meet
// in LDS; wave w then finishes tile (w / 2, w % 2) -- the sum of the four partials in the accumulator layout -- with the
// shared epilogue (the gate: waves 0 and 1 take both row tiles of a column tile).
//   * weights: straight from the packed image into registers (a lane's 16 bytes of each fragment), two taps at a time,
//     two register sets: the next group is requested before the current one is multiplied;
//   * window: plain 16-byte loads into registers (all lanes busy: a lane takes (pair row, four times) positions of the
//     chunk's 16 x LW4 grid), converted and written to the wave's LDS slot of the chunk's parity; the next chunk's loads
//     are requested as soon as the registers are free;
//   * every load is unconditional (addresses clamped), so the compiler's counted waits see the same number of
//     operations on every path and leave the prefetches in flight.
// The sum over chunks is taken in a different order than in the kernels above (four partial sums): fp32 rounding only.
template <int K, bool GATE>
__global__ void __launch_bounds__(256, 1) conv_frame_splitk(ConvArgs a) {
  constexpr int MT = 2, NT = 1, NWV = 4;
  constexpr int HALO = K == 1 ? 0 : FR_HALO;
  constexpr int BN = 32 * NT, LWP = BN + HALO, LWP4 = LWP / 4;
  constexpr int WBUF = CONV_CK * LWP * 4;                       // bytes of a window slot (hi image | lo image)
  constexpr int NIT = (16 * LWP4 + 63) / 64;                    // staging sweeps of a wave over the (pair row, four times) grid
  constexpr int GT = 2;                                         // taps per weight group
  constexpr int GPC = (K + GT - 1) / GT;                        // groups per chunk
  extern __shared__ __attribute__((aligned(16))) float xs[];
  char* const wbase = reinterpret_cast<char*>(xs);
  // (dynamic LDS = max(window slots 2 NWV WBUF, the four partial tiles NWV MT NT 16 64 4 bytes): launch_splitk_g)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * BN;
  const int mtile0 = blockIdx.y * MT;
  const int n_mtiles = (a.M + 31) >> 5;
  const int nch = a.nchunks;
  const int total_it = nch * K;
  // ---- the bias / conditioning rows of the tile this wave will finish (the gate: wave 0 both row tiles; else waves
  // 0 and 1 one each), first; its residual / destination values are requested when the main loop is done
  constexpr int FM = GATE ? 2 : 1;
  ConvEpi<FM, 1> epi;
  const bool finisher = wave < (GATE ? 1 : 2);
  const int fmt0 = mtile0 + (GATE ? 0 : wave);
  if (finisher) conv_epi_load<FM, 1>(a, epi, fmt0, n_mtiles, t0, 0, l31, h, b, 1);

  // ---- weights of a group (chunk c, taps [tap0, tap0 + GT)): a lane's 16 bytes of [mt][ks][hi | lo]
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  struct WSet { f32x4v w[GT][MT][4]; };
  const f32x4v* wp4 = reinterpret_cast<const f32x4v*>(a.wp);
  int mt_c[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) mt_c[mt] = mtile0 + mt < n_mtiles ? mtile0 + mt : n_mtiles - 1;
  auto wload = [&](WSet& W, int c, int g) {
    c = c < nch ? c : nch - 1;                                  // (past the end: a harmless re-read)
#pragma unroll
    for (int u = 0; u < GT; ++u) {
      int tap = g * GT + u;
      tap = tap < K ? tap : K - 1;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          W.w[u][mt][q] = wp4[(((size_t)mt_c[mt] * total_it + c * K + tap) * 4 + q) * 64 + lane];
    }
  };
  // ---- window of a chunk: positions idx = sweep 64 + lane of the 16 x LW4 grid (pair row p, four times c4)
  const float* xb = a.x + (size_t)b * a.x_bs;
  const int LW = BN + (K - 1) * a.dil;
  const int t_start = ((t0 - a.pad) >> 2) << 2;
  const int off = (t0 - a.pad) - t_start;
  const int LW4 = (LW + off + 3) >> 2;
  const int len = a.lengths ? (int)a.lengths[b] : 0x7fffffff;
  const int lim = a.in_mask ? (len < a.T_in ? len : a.T_in) : a.T_in;
  const float slope_eff = a.in_act ? a.in_slope : 1.f;
  int xp[NIT], xt[NIT];                 // pair row, first time of this lane's position per sweep (pair row 16 = none)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 64 + lane;
    const int p = idx / LW4, c4 = idx - p * LW4;
    xp[it] = p < 16 ? p : 16;
    xt[it] = t_start + 4 * c4;
  }
  struct XSet { f32x4v va[NIT], vb[NIT]; };
  auto xload = [&](XSet& X, int c) {
    c = c < nch ? c : nch - 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ci = c * CONV_CK + 2 * xp[it];
      const bool tin = xp[it] < 16 && xt[it] >= 0 && xt[it] < a.T_in;
      const float* pa = xb + (size_t)(tin && ci < a.Cin ? ci : 0) * a.x_cs + (tin ? xt[it] : 0);
      const float* pb = xb + (size_t)(tin && ci + 1 < a.Cin ? ci + 1 : 0) * a.x_cs + (tin ? xt[it] : 0);
      X.va[it] = *reinterpret_cast<const f32x4v*>(pa);
      X.vb[it] = *reinterpret_cast<const f32x4v*>(pb);
    }
  };
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  auto xconvert = [&](const XSet& X, int c, int slot) {
    unsigned* const ph = reinterpret_cast<unsigned*>(wbase + (wave * 2 + slot) * WBUF);
    unsigned* const pl = ph + (CONV_CK / 2) * LWP;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (xp[it] >= 16) continue;
      const int ci = c * CONV_CK + 2 * xp[it];
      const bool tin = xt[it] >= 0 && xt[it] < a.T_in;
      const bool ina = tin && ci < a.Cin, inb = tin && ci + 1 < a.Cin;
      unsigned wh4[4], wl4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x2v x = {X.va[it][u], X.vb[it][u]};
        x.x = (ina && xt[it] + u < lim) ? x.x : 0.f;
        x.y = (inb && xt[it] + u < lim) ? x.y : 0.f;
        const f32x2v y = x * slope_eff;
        asm("v_max_f32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(y.y));
        vsp_split_pair(x.x, x.y, wh4[u], wl4[u]);       // (kernels.h; unscaled lo parts: ONE accumulator)
      }
      const int o = xp[it] * LWP + (xt[it] - t_start);
      *reinterpret_cast<u32x4*>(ph + o) = u32x4{wh4[0], wh4[1], wh4[2], wh4[3]};
      *reinterpret_cast<u32x4*>(pl + o) = u32x4{wl4[0], wl4[1], wl4[2], wl4[3]};
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  // MFMAs of a group from the wave's window slot
  auto compute = [&](const WSet& W, int g, int slot) {
    const unsigned* const ph = reinterpret_cast<const unsigned*>(wbase + (wave * 2 + slot) * WBUF);
    const unsigned* const pl = ph + (CONV_CK / 2) * LWP;
#pragma unroll
    for (int u = 0; u < GT; ++u) {
      if (g * GT + u >= K) break;
      const int col = l31 + off + (g * GT + u) * a.dil;
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
          const f16x8 ah = __builtin_bit_cast(f16x8, W.w[u][mt][2 * ks]);
          const f16x8 al = __builtin_bit_cast(f16x8, W.w[u][mt][2 * ks + 1]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(ah, bh[nt], acc[mt][nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(ah, bl[nt], acc[mt][nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = FR_MFMA16(al, bh[nt], acc[mt][nt]);
        }
      }
    }
  };

  // ---- this wave's chunks: c_j = wave + 4 j
  const int nj = nch > wave ? (nch - wave + NWV - 1) / NWV : 0;
  WSet WA, WB;
  XSet XR;
  if (nj > 0) {
    xload(XR, wave);
    wload(WA, wave, 0);
    xconvert(XR, wave, 0);
    xload(XR, wave + NWV);                      // (the second chunk's window, or a harmless re-read)
    // two chunks per iteration: the group sequence (and with it the set each group uses) is static
    for (int j = 0; j < nj; j += 2) {
      const int c0 = wave + NWV * j, c1 = c0 + NWV;
      const bool has1 = j + 1 < nj;
      // chunk c0 from slot 0: groups alternate WA, WB, starting with WA when GPC is even or this is the first chunk
      // of the pair (2 GPC groups per iteration keep the alternation aligned)
      auto chunk_groups = [&](int c, int cnext, int slot, auto FIRST_IS_A) {
        constexpr bool first_a = decltype(FIRST_IS_A)::value;
#pragma unroll
        for (int g = 0; g < GPC; ++g) {
          const bool use_a = first_a ? (g % 2 == 0) : (g % 2 == 1);
          // request the next group (of this chunk, or the first of the next) into the other set
          if (g + 1 < GPC) { if (use_a) wload(WB, c, g + 1); else wload(WA, c, g + 1); }
          else { if (use_a) wload(WB, cnext, 0); else wload(WA, cnext, 0); }
          if (use_a) compute(WA, g, slot); else compute(WB, g, slot);
        }
      };
      chunk_groups(c0, c1, 0, std::true_type{});
      if (has1) xconvert(XR, c1, 1);            // (its loads went out a chunk ago)
      xload(XR, c1 + NWV);
      if (has1) {
        if constexpr (GPC % 2 == 0) chunk_groups(c1, c1 + NWV, 1, std::true_type{});
        else chunk_groups(c1, c1 + NWV, 1, std::false_type{});
        if (j + 2 < nj) xconvert(XR, c1 + NWV, 0);
        xload(XR, c1 + 2 * NWV);
      }
    }
  }
  if (finisher) conv_epi_load<FM, 1>(a, epi, fmt0, n_mtiles, t0, 0, l31, h, b, 2);
  // ---- the four partial tiles meet: [wave][tile][register][lane] floats over the window slots
  __syncthreads();
  float* const part = reinterpret_cast<float*>(wbase);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part[(((wave * MT + mt) * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r] * G16_UNSCALE;
  __syncthreads();
  auto gather = [&](int mt, int nt, f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NWV; ++w2) sum += part[(((w2 * MT + mt) * NT + nt) * 16 + r) * 64 + lane];
      v[r] = sum;
    }
  };
  if (finisher) {
    f32x16 tf[FM][1];
    if constexpr (GATE) { gather(0, 0, tf[0][0]); gather(1, 0, tf[1][0]); }
    else gather(wave, 0, tf[0][0]);
    conv_epi_store<FM, 1>(a, tf, epi, fmt0, n_mtiles, t0, 0, l31, h, b, len);
  }
}

template <int K, bool GATE>
static hipError_t launch_splitk_g(const ConvArgs& a, int B, hipStream_t s) {
  constexpr int LWP = 32 + (K == 1 ? 0 : FR_HALO);
  constexpr size_t win = (size_t)2 * 4 * CONV_CK * LWP * 4, part = (size_t)4 * 2 * 16 * 64 * 4;
  constexpr size_t lds = win > part ? win : part;
  static std::atomic<uint64_t> attr_done{0};
  auto kern = conv_frame_splitk<K, GATE>;
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, attr_done); e != hipSuccess) return e;
  dim3 grid((a.Nq + 31) / 32, (a.M + 63) / 64, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_splitk(const ConvArgs& a, int B, hipStream_t s) {
  return a.act == 2 ? launch_splitk_g<K, true>(a, B, s) : launch_splitk_g<K, false>(a, B, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_t1_gemv: a 1x1 convolution on ONE time step -- the cond_layer(g) / cond(g) projections of the speaker embedding
// (reference modules.py:153-155, models.py:279, 507): out[b][m] = bias[m] + sum_ci w[m][ci] x[b][ci].  A GEMV: one
// thread per output row reads its 16-byte pieces of the packed split-f16 image ((hi + lo) * 2^-8 = the fp32 weight to 22
// bits, exactly what the MFMA path multiplies) and accumulates in fp32 against the full-precision input.  The matrix
// kernels spent 38 us on these (a 64-row tile per block for a single column); this is one memory round trip.
// Round 5: a block = one 32-row tile x EIGHT input-channel groups (group g takes chunks g, g + 8, ..): eight 16-byte loads
// per thread instead of a chain of 64, partial sums through the LDS in a fixed order (25 -> 8 us per launch at any batch).
__global__ void __launch_bounds__(256) conv_t1_gemv(ConvArgs a) {
  __shared__ float xv[2048];
  __shared__ float part[8][32];
  const int tid = threadIdx.x, b = blockIdx.y, rin = tid & 31, grp = tid >> 5, mtile = blockIdx.x;
  for (int i = tid; i < a.Cin; i += 256) xv[i] = a.x[(size_t)b * a.x_bs + (size_t)i * a.x_cs];
  __syncthreads();
  float acc = 0.f;
  const uint4* __restrict__ wp = reinterpret_cast<const uint4*>(a.wp);
  for (int chunk = grp; chunk < a.nchunks; chunk += 8) {
    const size_t blk = ((size_t)mtile * a.nchunks + chunk) * 4;         // 1 KiB sub-images [ks][hi | lo] of the 4 KiB block
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const uint4 qh = wp[(blk + ks * 2) * 64 + rin + 32 * hh], ql = wp[(blk + ks * 2 + 1) * 64 + rin + 32 * hh];
        const f16x8 wh = __builtin_bit_cast(f16x8, qh), wl = __builtin_bit_cast(f16x8, ql);
        const int ci0 = chunk * CONV_CK + ks * 16 + hh * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (ci0 + j < a.Cin) acc = fmaf(((float)wh[j] + (float)wl[j]) * G16_UNSCALE, xv[ci0 + j], acc);
      }
  }
  part[grp][rin] = acc;
  __syncthreads();
  const int m = mtile * 32 + rin;
  if (grp == 0 && m < a.M) {
    float v = a.bias ? a.bias[m] : 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][rin];
    a.out[(size_t)b * a.o_bs + (size_t)m * a.o_cs] = v;
  }
}

// Grids up to this many 64 x 128 tiles (one round of the chip at one block per CU) take the latency kernel; above it the
// throughput kernel's three co-resident blocks per CU hide a block's serial steps better.  Same box, frame-rate
// convolutions per step at thresholds 512 / 256 / 0: the 5168-frame utterance 2.02 / 1.96 / 2.28 ms, C2 (16
// utterances) 2.97 / 2.40 / 2.36, C3 5.17 / 5.14 / -- (88.8 ms per step against 84.8 with the latency kernel everywhere).
static long fr_max_blocks() {
#ifdef VSP_EXPERIMENTS
  static long v = -1;
  if (v < 0) { const char* e = getenv("VSP_FR_BLOCKS"); v = e ? atol(e) : 256; }
  return v;
#else
  return 256;
#endif
}

// (round 5 experiment: a separate limit for 1x1 convolutions, whose throughput-kernel blocks are chains of window round trips)
static long fr_max_blocks_k1() {
#ifdef VSP_EXPERIMENTS
  static long v = -1;
  if (v < 0) { const char* e = getenv("VSP_FR_BLOCKS_K1"); v = e ? atol(e) : fr_max_blocks(); }
  return v;
#else
  return fr_max_blocks();
#endif
}

// Grids up to this many 64 x 32 tiles take the channel-split kernel (same box: one utterance 5.17 -> 3.9 ms, the
// 5168-frame utterance 18.4 -> 17.8, C2 24.5 -> 24.1; thresholds 128 / 256 / 512 / 1024 / 4096 swept).
static long fr_splitk_blocks() {
#ifdef VSP_EXPERIMENTS
  static long v = -1;
  if (v < 0) { const char* e = getenv("VSP_FR_SPLITK"); v = e ? atol(e) : 512; }
  return v;
#else
  return 512;
#endif
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
    // small grids: a 1x1 convolution whose weights have the column-tile image runs every row of a 64-column tile in one
    // block (conv_cols.hip) -- the row-tiled kernels below are 3-9 blocks per column tile, each a chain of window round trips
    // (a few blocks -- one utterance -- keep the row-tiled kernels: a column-tile block is one chain of round trips of ~10 us)
    if (a.wg && conv_cols_supported(a) && (long)B * ((a.Nq + 63) / 64) <= a.wg_max_blocks &&
        (long)B * ((a.Nq + 63) / 64) >= a.wg_min_blocks) return launch_conv_cols(a, B, s);
    // one time step, plain epilogue: the cond(g) projections
    if (a.K == 1 && a.T_in == 1 && a.Nq == 1 && a.Cin <= 2048 && a.act == 0 && !a.res && !a.cond && !a.in_mask && !a.in_act &&
        !a.mask_pre && !a.mask_post && !a.acc_prev && !a.split_row && a.alpha == 1.f && a.div == 1.f) {
      hipLaunchKernelGGL(conv_t1_gemv, dim3((a.M + 31) / 32, B), dim3(256), 0, s, a);
      return hipGetLastError();
    }
    // the latency form (conv_frame_f16s) where the grid does not fill the chip: 64-row tiles, at most two rounds of blocks
    const bool vec = ((a.x_cs & 3) == 0) && ((a.x_bs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    const float sl = a.in_act ? a.in_slope : 1.f;
    if (vec && (a.K - 1) * a.dil + 6 <= FR_HALO && (a.K > 1 || a.pad == 0) && sl >= 0.f && sl <= 1.f && (a.M > 32 || gate)) {
      const long blocks = (long)B * ((a.Nq + 127) / 128) * ((a.M + 63) / 64);
      hipError_t e = hipSuccess;
      // a few dozen blocks: the channel-split form
      const long blocks64 = (long)B * ((a.Nq + 31) / 32) * ((a.M + 63) / 64);
      // (short time axes -- the phoneme-rate encoder of a whole batch -- have nothing better up to a few rounds of blocks)
      if (blocks64 <= (a.Nq <= 96 ? 8 : 1) * fr_splitk_blocks() && a.nchunks >= 2) {
        if (a.K == 1) return launch_splitk<1>(a, B, s);
        if (a.K == 3) return launch_splitk<3>(a, B, s);
        if (a.K == 5) return launch_splitk<5>(a, B, s);
        // (k7 = the generator's conv_pre stays on ONE kernel at every size: the streamed vocoder and the locality
        // tests compare generator outputs of different lengths bit for bit)
      }
      if (a.Nq <= 96) { if (blocks * 2 <= fr_max_blocks() && launch_frame_k<2, 1, 1, 2>(a, B, s, e)) return e; }
      else if (blocks <= (a.K == 1 ? fr_max_blocks_k1() : fr_max_blocks())) { if (launch_frame_k<2, 1, 1, 4>(a, B, s, e)) return e; }
    }
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
    // (what conv1d_f32_mfma's pruned form assumes: 16-byte aligned rows, a leaky-relu slope the max form covers, no polyphase)
    const bool vec_ok = ((a.x_cs & 3) == 0) && ((a.x_bs & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    const float sl_eff = a.in_act ? a.in_slope : 1.f;
    const bool prune_ok = vec_ok && sl_eff >= 0.f && sl_eff <= 1.f && a.ups_s == 0 && !(CONV_DIAG & 2);
    if (a.M <= 64 || (a.M % 128 != 0 && a.M < 256) || !wide) {
      if (a.Nq >= 1024) return launch_tile<2, 2, 1, 4, true, true>(a, B, s);
      if (prune_ok) return launch_tile<2, 1, 1, 4, true, true, true>(a, B, s);
      return launch_tile<2, 1, 1, 4, true, true>(a, B, s);
    }
    if (prune_ok) return launch_tile<2, 2, 2, 2, true, true, true>(a, B, s);
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
