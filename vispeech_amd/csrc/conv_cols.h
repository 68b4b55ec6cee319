// Column-tile form of the 1x1 frame-rate convolutions (round 6): ONE block computes EVERY output row of a 64-column tile
//   tile[row][t] = sum_ci W[row][ci] * (x[ci][t0 + t] * mask)         (K = 1, Cin <= 192, rows <= 384)
// and leaves it in LDS, so that whatever needs all rows of a column can run in the same launch: the LayerNorm behind
// conv_o (reference attentions.py:41-42), the q | k | v operand packing behind the attention projections
// (attentions.py:138-146), both destinations of a WN res_skip layer (modules.py:165-172).  It exists for the SMALL
// grids -- one utterance, the per-rank slice of a sharded batch -- where the row-tiled kernels of conv_mfma.hip are a
// chain of launches of 5-20 us each whose time is ramp + epilogue, not work: 1x1 convolutions with
// 192 input channels are 6 contraction steps.
//
// Arithmetic: the split-f16 form of every other kernel here -- x = hi + lo by vsp_split_pair, weights * 2^8 as hi | lo
// images in 16x16x32 A-fragment order (pack_g16_weights, K = 1: [chunk32][m-tile][hi|lo][lane][8 halfs]), three
// v_mfma_f32_16x16x32_f16 per product into one fp32 accumulator, * 2^-8 on the way out.
// Block = 4 waves and at most 192 rows (more rows = more blocks: the halves of a res_skip layer are independent
// destinations); wave w owns m-tiles [w MW, (w + 1) MW) x all four 16-column n-tiles (the weights stream from L2 ONCE per
// block, a fragment serves four MFMA columns; the activations -- 48 KB as split images -- are shared through LDS).
// A block is a chain of dependent round trips (x -> LDS -> MFMA -> tile -> residual -> store): every global request it
// will ever wait for -- all weight fragments, the activations, the epilogue's operands -- is issued before its first wait.
#pragma once
#include "g16_common.h"

namespace vsp {

constexpr int CC_BT = 64;                    // columns per block
constexpr int CC_TS = CC_BT + 1;             // row stride of the fp32 result tile (floats)
inline constexpr int cc_image_bytes(int Cin) { return (Cin / 32) * 8 * 1024; }          // [chunk][hi|lo][plane][64 t][8 halfs]
inline constexpr int cc_lds_bytes(int Cin, int rows) { return cc_image_bytes(Cin) + rows * CC_TS * 4 + 4 * 64 * 4 + 3 * 192 * 4; }   // images | tile | LayerNorm partials | per-row parameters

// Three pieces, so that a kernel can put every global round trip in flight before the first wait:
//   cc_load_weights  ALL weight fragments of the wave (NC chunks x MW m-tiles x hi | lo: 144 registers at MW = 3, 192 inputs)
//   cc_stage_x       x[b] columns [t0, t0 + 64) -> split images in `img`; columns at and beyond `t_valid` (the tensor's
//                    extent, or the utterance's length when the input is masked) enter as zero; ends with a barrier
//   (the kernel requests its epilogue operands here)
//   cc_contract      the MFMAs; result * G16_UNSCALE -> tile[(local row) * CC_TS + t]; ends with a barrier: every
//                    thread may read the whole tile afterwards
// wg rows [16 mtile0, 16 (mtile0 + MTB)) are the block's; wave w owns m-tiles [w MW, (w + 1) MW) of them.
template <int MW, int CIN>
struct CcWeights { u32x4 a[CIN / 32][MW][2]; };

template <int MW, int CIN>
__device__ __forceinline__ void cc_load_weights(CcWeights<MW, CIN>& W, const uint16_t* __restrict__ wg, int nmt_total, int mtile0,
                                                int MTB) {
  constexpr int NC = CIN / 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4* wp = reinterpret_cast<const u32x4*>(wg);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      const int mt = wave * MW + i;
      // (m-tiles beyond the block's range: a harmless re-read of its first tile, never multiplied)
      const size_t blk = ((size_t)c * nmt_total + mtile0 + (mt < MTB ? mt : 0)) * 2;
      W.a[c][i][0] = wp[blk * 64 + lane];
      W.a[c][i][1] = wp[(blk + 1) * 64 + lane];
    }
}

template <int CIN>
__device__ __forceinline__ void cc_stage_x(const float* __restrict__ xb, long x_cs, int t0, int t_valid, char* __restrict__ img) {
  static_assert(CIN % 32 == 0 && CIN <= 192, "whole 32-channel chunks; the images of 192 channels are 48 KB");
  constexpr int NC = CIN / 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a piece = 8 channels of one time step (one 16-byte fragment row per image); a wave stages the channel groups wave,
  // wave + 4, ..: every load of a thread is in flight before the first conversion
  const int t = lane, tt = t0 + t;
  const bool in = tt < t_valid;
  const float* xp = xb + (in ? tt : 0);
  float v[NC][8];
#pragma unroll
  for (int u = 0; u < NC; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[u][j] = in ? xp[(long)((wave + 4 * u) * 8 + j) * x_cs] : 0.f;
#pragma unroll
  for (int u = 0; u < NC; ++u) {
    const int grp = wave + 4 * u;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vsp_split_pair(v[u][2 * j], v[u][2 * j + 1], hi[j], lo[j]);
    const int c = grp >> 2, q4 = grp & 3;
    *reinterpret_cast<u32x4*>(img + ((c * 2 + 0) * 4 + q4) * 1024 + t * 16) = u32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u32x4*>(img + ((c * 2 + 1) * 4 + q4) * 1024 + t * 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
  }
  __syncthreads();
}

template <int MW, int CIN>
__device__ __forceinline__ void cc_contract(const CcWeights<MW, CIN>& W, int MTB, const char* __restrict__ img,
                                            float* __restrict__ tile) {
  constexpr int NC = CIN / 32;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int mt_first = wave * MW;
  f32x4 acc[MW][4];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f16x8 bh[4], bl[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bh[n] = *reinterpret_cast<const f16x8*>(img + ((c * 2 + 0) * 4 + q) * 1024 + (16 * n + l15) * 16);
      bl[n] = *reinterpret_cast<const f16x8*>(img + ((c * 2 + 1) * 4 + q) * 1024 + (16 * n + l15) * 16);
    }
#pragma unroll
    for (int i = 0; i < MW; ++i) {
      if (mt_first + i < MTB) {                                   // (uniform per wave)
        const f16x8 ah = __builtin_bit_cast(f16x8, W.a[c][i][0]), al = __builtin_bit_cast(f16x8, W.a[c][i][1]);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = G16_MFMA(ah, bh[n], acc[i][n]);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = G16_MFMA(al, bh[n], acc[i][n]);
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = G16_MFMA(ah, bl[n], acc[i][n]);
      }
    }
  }
  // ---- the tile: a lane of a D tile holds rows 4 q .. 4 q + 3 of column l15
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    if (mt_first + i < MTB) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[(16 * (mt_first + i) + 4 * q + r) * CC_TS + 16 * n + l15] = acc[i][n][r] * G16_UNSCALE;
    }
  }
  __syncthreads();
}

}  // namespace vsp
