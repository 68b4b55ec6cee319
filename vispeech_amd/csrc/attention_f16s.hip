// Windowed relative-position self-attention (reference attentions.py:148-179 and the pad / reshape helpers
// :181-243) on the f16 matrix core with SPLIT operands, ONE pass (online softmax).
//
//   S_ij = (q_i / sqrt(dk)) . k_j + [|j-i| <= w] (q_i / sqrt(dk)) . Ek[j-i+w]
//   S_ij = -1e4 where mask_i * mask_j == 0            (attentions.py:166; NOT -inf, gotcha G9)
//   P = softmax_j S ;  O_i = sum_j P_ij v_j + sum_{|j-i|<=w} P_ij Ev[j-i+w]
//
// Round 1 ran this on v_mfma_f32_32x32x2_f32 in two passes (QK^T computed twice, one scalar ds_read_b32 per MFMA
// operand): 25 TFLOP/s on the 60 s utterance.  Here:
//   * attn_pack_f16s (once per layer, O(T)): q | k | v fp32 [3H][T] -> f16 hi / lo images in MFMA FRAGMENT ORDER
//     (q pre-scaled by log2(e) / sqrt(dk) so that the softmax runs on v_exp_f32 = 2^x; power-of-two operand scales keep
//     the lo parts normal f16 numbers), so the attention blocks move operands by LDS-DMA / 16-byte loads only and
//     never convert anything;
//   * attn_relpos_f16s: S^T = K Q^T on v_mfma_f32_16x16x32_f16 (key on the accumulator register, query on the
//     lane: row statistics are in-register + two cross-lane steps), three MFMAs per product into one fp32
//     accumulator (hi hi + lo hi + hi lo: fp32-chain accuracy); the probability tile goes from the accumulator
//     registers straight into the B operand of O^T = V^T P^T (the key permutation this implies is baked into the
//     packed V image); online softmax: QK^T is computed ONCE; 32-key K / V tiles double-buffered in LDS by
//     global_load_lds; the (2w+1)-wide band terms use a per-query LDS table of raw scores (relative values are
//     applied in fp32 at the end, from the final statistics).
// Mask semantics unchanged: -1e4 (times log2 e in the 2^x domain) for masked pairs, so a fully masked query row
// gives the uniform average the reference gives; keys beyond T get -inf.
#include "kernels.h"
#include "conv_cols.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace vsp {


constexpr int AF_RS = 17;             // row stride of the per-query band tables (2w+1 <= 16)
constexpr float AF_QS = 128.f;        // operand scales (powers of two): q, k, v, p
constexpr float AF_KS = 16.f;
constexpr float AF_VS = 16.f;
constexpr float AF_PEXP = 14.f;       // p is carried as p * 2^14 (<= 16384: fits f16, its lo part stays normal)
constexpr float AF_LOG2E = 1.4426950408889634f;


// x -> hi = f16(x), lo = f16(x - hi), two values at a time.  The empty asm pins the ROUNDED hi that is stored as THE
// value the residual is taken against: left to itself hipcc re-derived hi for the subtraction by another conversion
// path that rounds ties differently, which put single elements off by a whole f16 ulp (one q element in ~10^4:
// found as 3e-5 errors in single query rows of the T = 300 parity case).
// Operands beyond the f16 range are clamped to its largest finite value (a scaled |q| > 6.5e4 is |q| > ~3.4e3 before
// scaling; the reference's layer-normed activations are O(1..10)): the result stays finite instead of turning the
// whole row into NaN through inf - inf; the fp32 kernel (VSP_ATT=f32) has no such limit (include/vispeech_hip.h).
__device__ __forceinline__ void af_split2(f32x2 x, f16x2& h, f16x2& l) {
  x.x = __builtin_amdgcn_fmed3f(x.x, -65504.f, 65504.f);
  x.y = __builtin_amdgcn_fmed3f(x.y, -65504.f, 65504.f);
  h = __builtin_convertvector(x, f16x2);
  asm volatile("" : "+v"(h));
  const f32x2 back = __builtin_convertvector(h, f32x2);
  l = __builtin_convertvector(x - back, f16x2);
}

template <class F, int... I>
__device__ __forceinline__ void af_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void af_for(F&& f) {
  af_for_impl(f, std::make_integer_sequence<int, N>{});
}

// bytes of one packed operand image per (utterance, head): Tpad x DK elements, hi + lo halfs
size_t attn_pack_bytes(int B, int n_heads, int DK, int T) {
  const size_t tpad = (size_t)(T + 63) / 64 * 64;
  return (size_t)B * n_heads * tpad * DK * 4;
}

// ---------------------------------------------------------------------------------------------------------
// qkv [B][3H][T] fp32 -> Qp, Kp: [b][head][tile16][dchunk32][hi|lo][lane][8]  (lane = t % 16 + 16 * ((d % 32) / 8))
//                        Vp:     [b][head][group32][dtile16][hi|lo][lane][8]  (lane = d % 16 + 16 * q4, element j =
//                                key 32 g + (j < 4 ? 4 q4 + j : 16 + 4 q4 + j - 4): the order in which a lane of two
//                                16-key S^T accumulator tiles holds its keys)
// One block = 64 time steps of one (b, head) of q, k or v; values beyond T are zero.
template <int DK>
__global__ void __launch_bounds__(256) attn_pack_f16s(const float* __restrict__ qkv, long bs, long cs, int H, int T,
                                                       _Float16* __restrict__ Qp, _Float16* __restrict__ Kp,
                                                       _Float16* __restrict__ Vp) {
  constexpr int NC = DK / 32, ND = DK / 16;
  __shared__ float tile[DK][65];
  const int b = blockIdx.z / 3, hd = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  const int n_heads = gridDim.y;
  const size_t tpad = (size_t)gridDim.x * 64;
  const size_t img = ((size_t)b * n_heads + hd) * tpad * DK * 2;     // halfs per image (hi + lo)
  const float qscale = AF_QS * AF_LOG2E / sqrtf((float)DK);
  {
    const int which = blockIdx.z % 3;            // q, k or v: one block each (a launch of a few dozen blocks is latency)
    const float* src = qkv + (size_t)b * bs + (size_t)(which * H + hd * DK) * cs;
    for (int idx = tid; idx < DK * 64; idx += 256) {
      const int d = idx >> 6, tl = idx & 63;
      tile[d][tl] = t0 + tl < T ? src[(size_t)d * cs + t0 + tl] : 0.f;
    }
    __syncthreads();
    const float sc = which == 0 ? qscale : (which == 1 ? AF_KS : AF_VS);
    if (which < 2) {
      _Float16* dst = (which == 0 ? Qp : Kp) + img;
      for (int u = tid; u < 64 * NC * 4; u += 256) {
        const int tl = u & 63, c = (u >> 6) % NC, q4 = (u >> 6) / NC;
        f16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          f16x2 h2, l2;
          af_split2(f32x2{tile[c * 32 + q4 * 8 + j][tl] * sc, tile[c * 32 + q4 * 8 + j + 1][tl] * sc}, h2, l2);
          vh[j] = h2.x; vh[j + 1] = h2.y;
          vl[j] = l2.x; vl[j + 1] = l2.y;
        }
        const size_t blk = ((size_t)((t0 + tl) >> 4) * NC + c) * 2;
        const int lane = (tl & 15) + 16 * q4;
        *reinterpret_cast<f16x8*>(dst + (blk * 64 + lane) * 8) = vh;
        *reinterpret_cast<f16x8*>(dst + ((blk + 1) * 64 + lane) * 8) = vl;
      }
    } else {
      _Float16* dst = Vp + img;
      for (int u = tid; u < DK * 2 * 4; u += 256) {
        const int d = u % DK, g = (u / DK) & 1, q4 = u / (2 * DK);
        f16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const int key = g * 32 + (j < 4 ? 4 * q4 + j : 16 + 4 * q4 + (j - 4));     // (key + 1 for element j + 1)
          f16x2 h2, l2;
          af_split2(f32x2{tile[d][key] * sc, tile[d][key + 1] * sc}, h2, l2);
          vh[j] = h2.x; vh[j + 1] = h2.y;
          vl[j] = l2.x; vl[j + 1] = l2.y;
        }
        const size_t blk = ((size_t)((t0 >> 5) + g) * ND + (d >> 4)) * 2;
        const int lane = (d & 15) + 16 * q4;
        *reinterpret_cast<f16x8*>(dst + (blk * 64 + lane) * 8) = vh;
        *reinterpret_cast<f16x8*>(dst + ((blk + 1) * 64 + lane) * 8) = vl;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The attention projections and the packing in ONE launch (round 6): conv_q | conv_k | conv_v (reference
// attentions.py:138-146: 1x1 convolutions of x * x_mask) on the column-tile form of conv_cols.h -- a block computes all H
// rows of q, k or v for 64 time steps into an LDS tile -- and the tile goes straight into the packed operand images
// (attn_pack_f16s's code, reading the LDS tile instead of the fp32 tensor): the [B][3H][T] tensor never exists, one launch
// and one HBM round trip less per encoder layer.  grid = (T / 64, 3, B); wg = pack_g16_weights of the 3H x H projection.
template <int DK, int NH>
__global__ void __launch_bounds__(256, 1) attn_qkv_pack_f16s(const float* __restrict__ x, long x_bs, long x_cs,
                                                            const uint16_t* __restrict__ wg, const float* __restrict__ bias,
                                                            const int64_t* __restrict__ lengths, int T,
                                                            _Float16* __restrict__ Qp, _Float16* __restrict__ Kp,
                                                            _Float16* __restrict__ Vp) {
  constexpr int H = DK * NH, NC = DK / 32, ND = DK / 16, MTB = H / 16, MW = (MTB + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char qp_smem[];
  char* const img = qp_smem;
  float* const tile = reinterpret_cast<float*>(qp_smem + cc_image_bytes(H));
  const int b = blockIdx.z, which = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  int len = T;
  if (lengths) { const long l = lengths[b]; len = l < 0 ? 0 : (l < T ? (int)l : T); }
  CcWeights<MW, H> W;
  cc_load_weights<MW, H>(W, wg, 3 * MTB, which * MTB, MTB);
  cc_stage_x<H>(x + (size_t)b * x_bs, x_cs, t0, len, img);
  cc_contract<MW, H>(W, MTB, img, tile);
  // + bias; columns beyond T are zero in the images (attn_pack_f16s: "values beyond T are zero")
  for (int idx = tid; idx < H * 64; idx += 256) {
    const int d = idx >> 6, tl = idx & 63;
    tile[d * CC_TS + tl] = t0 + tl < T ? tile[d * CC_TS + tl] + bias[which * H + d] : 0.f;
  }
  __syncthreads();
  const size_t tpad = (size_t)gridDim.x * 64;
  const float qscale = AF_QS * AF_LOG2E / sqrtf((float)DK);
  const float sc = which == 0 ? qscale : (which == 1 ? AF_KS : AF_VS);
  if (which < 2) {
    for (int u = tid; u < NH * 64 * NC * 4; u += 256) {
      const int hd = u / (64 * NC * 4), uu = u % (64 * NC * 4);
      const int tl = uu & 63, c = (uu >> 6) % NC, q4 = (uu >> 6) / NC;
      const float* th = tile + (size_t)hd * DK * CC_TS;
      _Float16* dst = (which == 0 ? Qp : Kp) + ((size_t)b * NH + hd) * tpad * DK * 2;
      f16x8 vh, vl;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f16x2 h2, l2;
        af_split2(f32x2{th[(c * 32 + q4 * 8 + j) * CC_TS + tl] * sc, th[(c * 32 + q4 * 8 + j + 1) * CC_TS + tl] * sc}, h2, l2);
        vh[j] = h2.x; vh[j + 1] = h2.y;
        vl[j] = l2.x; vl[j + 1] = l2.y;
      }
      const size_t blk = ((size_t)((t0 + tl) >> 4) * NC + c) * 2;
      const int lane = (tl & 15) + 16 * q4;
      *reinterpret_cast<f16x8*>(dst + (blk * 64 + lane) * 8) = vh;
      *reinterpret_cast<f16x8*>(dst + ((blk + 1) * 64 + lane) * 8) = vl;
    }
  } else {
    for (int u = tid; u < NH * DK * 2 * 4; u += 256) {
      const int hd = u / (DK * 2 * 4), uu = u % (DK * 2 * 4);
      const int d = uu % DK, g = (uu / DK) & 1, q4 = uu / (2 * DK);
      const float* th = tile + (size_t)hd * DK * CC_TS;
      _Float16* dst = Vp + ((size_t)b * NH + hd) * tpad * DK * 2;
      f16x8 vh, vl;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const int key = g * 32 + (j < 4 ? 4 * q4 + j : 16 + 4 * q4 + (j - 4));     // (key + 1 for element j + 1)
        f16x2 h2, l2;
        af_split2(f32x2{th[d * CC_TS + key] * sc, th[d * CC_TS + key + 1] * sc}, h2, l2);
        vh[j] = h2.x; vh[j + 1] = h2.y;
        vl[j] = l2.x; vl[j + 1] = l2.y;
      }
      const size_t blk = ((size_t)((t0 >> 5) + g) * ND + (d >> 4)) * 2;
      const int lane = (d & 15) + 16 * q4;
      *reinterpret_cast<f16x8*>(dst + (blk * 64 + lane) * 8) = vh;
      *reinterpret_cast<f16x8*>(dst + ((blk + 1) * 64 + lane) * 8) = vl;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Block = NWV x KS waves: NWV query groups of NQ query tiles of 16, each served by KS waves that take every KS-th key
// tile (a second wave per SIMD hides the first one's MFMA / LDS / exp latencies, and a 5168-frame utterance is only 162
// blocks of 64 queries); their (max, sum, O) partials are merged through LDS at the end.  Keys stream in tiles of 32
// through two LDS buffers of KS tiles.
template <int DK, int NWV, int NQ, int KS>
__global__ void __launch_bounds__(64 * NWV * KS) attn_relpos_f16s(const float* __restrict__ qkv, long bs, long cs,
                                                            const _Float16* __restrict__ Qp, const _Float16* __restrict__ Kp,
                                                            const _Float16* __restrict__ Vp, const float* __restrict__ emb_k,
                                                            const float* __restrict__ emb_v,
                                                            const int64_t* __restrict__ lengths, float* __restrict__ out,
                                                            long o_bs, long o_cs, int H, int T, int w, int n_heads, int n_utt) {
  constexpr int NC = DK / 32, ND = DK / 16;
  constexpr int QB = 16 * NQ * NWV;                  // queries per block
  constexpr int KTILE = 2 * NC * 2 * 1024;           // bytes of a 32-key K tile (2 key tiles x NC chunks x hi/lo)
  constexpr int VTILE = ND * 2 * 1024;               // bytes of a 32-key V tile
  constexpr int STAGE = KTILE + VTILE;
  constexpr int NWT = NWV * KS;                      // waves of the block
  constexpr int NPIECE = KS * STAGE / 1024, NPW = (NPIECE + NWT - 1) / NWT;
  constexpr int MERGE = (KS - 1) * NWV * NQ * (ND * 4 + 2) * 64 * 4;      // bytes of the partials handed over at the end
  static_assert(MERGE <= 2 * KS * STAGE, "the partials reuse the key / value buffers");
  __shared__ __attribute__((aligned(16))) char kv[2 * KS * STAGE];
  __shared__ float Rl[QB * AF_RS];                   // relative-key logits (2^x domain)
  __shared__ float Sb[QB * AF_RS];                   // raw band scores (2^x domain); -inf = no such key
  __shared__ float Evs[16 * DK];

  const int tid = threadIdx.x;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all % NWV, ks = wave_all / NWV;     // query group; which of the KS key-tile streams
  const int lane = tid & 63, l15 = lane & 15, q4 = lane >> 4;
  // XCD-aware block numbering: consecutive workgroup ids go to consecutive XCDs (eight L2s), so with a plain 3-D grid
  // the query blocks of one (utterance, head) -- which all stream the same K / V images -- sat on different L2s (counter
  // table of round 3: L2 hit rate 0.11, waves 53 % in s_waitcnt).  With at least eight (utterance, head) groups, group g
  // lives on XCD g % 8 and its query blocks take consecutive slots there: the first block of a group pulls a tile into
  // that L2, the others hit.  Fewer groups (one long utterance): the plain order, every XCD works on every group.
  const int nqb = (T + QB - 1) / QB;                  // query blocks per group
  const int ngroups = n_heads * n_utt;
  int qb, grp;
  if (ngroups >= 8) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    qb = slot % nqb;
    grp = (slot / nqb) * 8 + xcd;
    if (grp >= ngroups) return;                       // (padding of the group count to a multiple of eight)
  } else {
    qb = blockIdx.x % nqb;
    grp = blockIdx.x / nqb;
  }
  const int b = grp / n_heads, hd = grp - b * n_heads, i0 = qb * QB;
  const int nrel = 2 * w + 1;
  const int len = lengths ? (int)lengths[b] : T;
  const size_t tpad = (size_t)(T + 63) / 64 * 64;
  const size_t img = ((size_t)b * n_heads + hd) * tpad * DK * 2;
  const uint4* Kg = reinterpret_cast<const uint4*>(Kp + img);
  const uint4* Vg = reinterpret_cast<const uint4*>(Vp + img);
  const int ntiles = (T + 31) / 32;

  // ---- K / V tiles tt KS .. tt KS + KS - 1 -> LDS buffer (tt & 1) by LDS-DMA, 1 KiB pieces in fragment order
  // (tiles past the last re-read it: nobody computes on them)
  auto stage = [&](int tt) {
    char* dst = kv + (tt & 1) * (KS * STAGE);
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
      const int pp = u * NWT + wave_all;
      if (NPIECE % NWT == 0 || pp < NPIECE) {
        const int sub = pp / (STAGE / 1024), p = pp % (STAGE / 1024);
        int t = tt * KS + sub;
        t = t < ntiles ? t : ntiles - 1;
        const uint4* src = p < KTILE / 1024 ? Kg + ((size_t)t * (KTILE / 1024) + p) * 64 + lane
                                            : Vg + ((size_t)t * (VTILE / 1024) + (p - KTILE / 1024)) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + pp * 1024), 16, 0, 0);
      }
    }
  };
  stage(0);

  // ---- Q fragments of this wave's query tiles (B operand), straight from the packed image
  const int qt0 = (i0 >> 4) + wave * NQ;            // first 16-query tile of this wave
  f16x8 Qh[NQ][NC], Ql[NQ][NC];
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool ok = (size_t)(qt0 + n) * 16 < tpad;
      const _Float16* p = Qp + img + (((size_t)(ok ? qt0 + n : 0) * NC + c) * 2 * 64 + lane) * 8;
      Qh[n][c] = *reinterpret_cast<const f16x8*>(p);
      Ql[n][c] = *reinterpret_cast<const f16x8*>(p + 64 * 8);
    }

  // ---- relative-key logits of the wave's queries (2^x domain): the 2w+1 rows of emb_k are 2w+1 more keys -- one S^T
  // tile per query tile on the matrix core, split like the keys (round 2 did these T x (2w+1) x DK products as scalar
  // fp32 loops over strided global reads: 15 % of the attention time on utterance-sized inputs) --, band-score table, Ev
  if (ks == 0) {
    f16x8 eh[NC], el[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
      if (l15 < nrel) {
        const float* pe = emb_k + (size_t)l15 * DK + c * 32 + q4 * 8;
        e0 = *reinterpret_cast<const f32x4*>(pe);
        e1 = *reinterpret_cast<const f32x4*>(pe + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; u += 2) {
        f16x2 h2, l2;
        af_split2(f32x2{e0[u] * AF_KS, e0[u + 1] * AF_KS}, h2, l2);
        eh[c][u] = h2.x; eh[c][u + 1] = h2.y; el[c][u] = l2.x; el[c][u + 1] = l2.y;
        af_split2(f32x2{e1[u] * AF_KS, e1[u + 1] * AF_KS}, h2, l2);
        eh[c][4 + u] = h2.x; eh[c][5 + u] = h2.y; el[c][4 + u] = l2.x; el[c][5 + u] = l2.y;
      }
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      f32x4 R = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        R = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[c], Qh[n][c], R, 0, 0, 0);
        R = __builtin_amdgcn_mfma_f32_16x16x32_f16(el[c], Qh[n][c], R, 0, 0, 0);
        R = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[c], Ql[n][c], R, 0, 0, 0);
      }
      const int iq = (wave * NQ + n) * 16 + l15;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (4 * q4 + jj < nrel) Rl[iq * AF_RS + 4 * q4 + jj] = R[jj] * (1.f / (AF_QS * AF_KS));
    }
  }
  for (int idx = tid; idx < QB * AF_RS; idx += 64 * NWT) Sb[idx] = -INFINITY;
  for (int idx = tid; idx < nrel * DK; idx += 64 * NWT) Evs[idx] = emb_v[idx];

  f32x4 O[ND][NQ];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int n = 0; n < NQ; ++n) O[dt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int n = 0; n < NQ; ++n) { m_run[n] = -3.0e38f; l_run[n] = 0.f; }
  const float s_unscale = 1.f / (AF_QS * AF_KS);
  const float masked = -1e4f * AF_LOG2E;

  const int nsteps = (ntiles + KS - 1) / KS;
  for (int tt = 0; tt < nsteps; ++tt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of step tt have landed
    __syncthreads();                                      // everybody's have; everybody is done with step tt - 1
    if (tt + 1 < nsteps) stage(tt + 1);
    const int t = tt * KS + ks;                           // this wave's key tile of the step
    if (t >= ntiles) continue;
    const char* kb = kv + (tt & 1) * (KS * STAGE) + ks * STAGE + lane * 16;
    const char* vb = kb + KTILE;
    const int j0 = t * 32;
    // ---- S^T tiles: [key tile kt][query tile n], lane = query l15, register jj = key 16 kt + 4 q4 + jj
    f32x4 S[2][NQ];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int n = 0; n < NQ; ++n) S[kt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f16x8 kh = *reinterpret_cast<const f16x8*>(kb + ((kt * NC + c) * 2) * 1024);
        const f16x8 kl = *reinterpret_cast<const f16x8*>(kb + ((kt * NC + c) * 2 + 1) * 1024);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
          S[kt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, Qh[n][c], S[kt][n], 0, 0, 0);
          S[kt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, Qh[n][c], S[kt][n], 0, 0, 0);
          S[kt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, Ql[n][c], S[kt][n], 0, 0, 0);
        }
      }
    // ---- scores -> probabilities (online softmax), straight into the B operand of the PV product
    f16x8 Ph[NQ], Pl[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int iq = (wave * NQ + n) * 16 + l15;          // query index inside the block
      const int i = i0 + iq;
      const bool near = j0 + 31 >= i0 + (wave * NQ + n) * 16 - w && j0 <= i0 + (wave * NQ + n) * 16 + 15 + w;   // wave-uniform
      float sv[8];
      // (wave-uniform: a tile away from the band whose 32 keys and 16 queries are all valid has nothing to add or mask)
      const bool inside = !near && j0 + 31 < len && i0 + (wave * NQ + n) * 16 + 15 < len;
      if (inside) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sv[4 * kt + jj] = S[kt][n][jj] * s_unscale;
      } else
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = j0 + 16 * kt + 4 * q4 + jj;
          float s = S[kt][n][jj] * s_unscale;
          if (near) {
            const int rel = j - i + w;
            if (rel >= 0 && rel < nrel) s += Rl[iq * AF_RS + rel];
          }
          if (i >= len || j >= len) s = masked;
          if (j >= T) s = -INFINITY;
          if (near) {
            const int rel = j - i + w;
            if (rel >= 0 && rel < nrel && j < T) Sb[iq * AF_RS + rel] = s;
          }
          sv[4 * kt + jj] = s;
        }
      float tmax = sv[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) tmax = fmaxf(tmax, sv[e]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run[n], tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run[n] - m_new);
      const bool moved = __builtin_amdgcn_ballot_w64(m_new != m_run[n]) != 0;     // (wave-uniform: did any query's running max move?)
      m_run[n] = m_new;
      float part = 0.f;
      f16x8 ph, pl;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const f32x2 p = {__builtin_amdgcn_exp2f(sv[e] - m_new + AF_PEXP), __builtin_amdgcn_exp2f(sv[e + 1] - m_new + AF_PEXP)};   // p * 2^14
        part += p.x + p.y;
        f16x2 h2, l2;
        af_split2(p, h2, l2);
        ph[e] = h2.x; ph[e + 1] = h2.y;
        pl[e] = l2.x; pl[e + 1] = l2.y;
      }
      l_run[n] = l_run[n] * alpha + part;
      Ph[n] = ph;
      Pl[n] = pl;
      if (moved) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) O[dt][n] *= alpha;
      }
    }
    // ---- O^T += V^T P^T  (contraction over the tile's 32 keys in the packed order)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const f16x8 vh = *reinterpret_cast<const f16x8*>(vb + (dt * 2) * 1024);
      const f16x8 vl = *reinterpret_cast<const f16x8*>(vb + (dt * 2 + 1) * 1024);
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        O[dt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, Ph[n], O[dt][n], 0, 0, 0);
        O[dt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, Ph[n], O[dt][n], 0, 0, 0);
        O[dt][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, Pl[n], O[dt][n], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                       // band scores of every wave are in Sb; the K / V buffers are free
  if constexpr (KS > 1) {
    // ---- the key-tile streams of a query group meet: (m, l, O) of streams 1 .. KS - 1 -> LDS -> stream 0
    float* mg = reinterpret_cast<float*>(kv);
    constexpr int NV = ND * 4 + 2;
    if (ks > 0) {
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        float* dst = mg + ((size_t)(((ks - 1) * NWV + wave) * NQ + n) * NV) * 64 + lane;
        dst[0] = m_run[n];
        dst[64] = l_run[n];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) dst[(2 + dt * 4 + jj) * 64] = O[dt][n][jj];
      }
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        const float* src = mg + ((size_t)(((k2 - 1) * NWV + wave) * NQ + n) * NV) * 64 + lane;
        const float m2 = src[0], l2 = src[64];
        const float m_new = fmaxf(m_run[n], m2);
        const float a1 = __builtin_amdgcn_exp2f(m_run[n] - m_new), a2 = __builtin_amdgcn_exp2f(m2 - m_new);
        m_run[n] = m_new;
        l_run[n] = l_run[n] * a1 + l2 * a2;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) O[dt][n][jj] = O[dt][n][jj] * a1 + src[(2 + dt * 4 + jj) * 64] * a2;
      }
  }

  // ---- normalise, add the relative-value term (fp32, from the final statistics), store
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    const int iq = (wave * NQ + n) * 16 + l15, i = i0 + iq;
    float l_tot = l_run[n];
    l_tot += __shfl_xor(l_tot, 16);
    l_tot += __shfl_xor(l_tot, 32);                      // sum of p * 2^14 over all keys
    const float inv_l = 1.f / l_tot;
    float pb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      pb[r] = r < nrel ? __builtin_amdgcn_exp2f(Sb[iq * AF_RS + r] - m_run[n] + AF_PEXP) * inv_l : 0.f;
    if (i < T) {
      float* orow = out + (size_t)b * o_bs + (size_t)(hd * DK) * o_cs + i;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int d = 16 * dt + 4 * q4 + jj;
          float rv = 0.f;
          for (int e = 0; e < nrel; ++e) rv += pb[e] * Evs[e * DK + d];
          orow[(size_t)d * o_cs] = O[dt][n][jj] * (inv_l / AF_VS) + rv;
        }
    }
  }
}

template <int DK, int NWV, int NQ, int KS>
static void launch_attn_f16s(const float* qkv, long qkv_bs, long qkv_cs, const _Float16* Qp, const _Float16* Kp,
                             const _Float16* Vp, const float* emb_k, const float* emb_v, const int64_t* lengths, float* out,
                             long o_bs, long o_cs, int B, int H, int n_heads, int T, int window, hipStream_t s) {
  constexpr int QB = 16 * NQ * NWV;
  const int nqb = (T + QB - 1) / QB, ngroups = n_heads * B;
  // (one-dimensional grid, the kernel maps it: groups padded to a multiple of eight when they are spread over the XCDs)
  const long nblocks = (long)nqb * (ngroups >= 8 ? (ngroups + 7) / 8 * 8 : ngroups);
  dim3 grid((unsigned)nblocks);
  hipLaunchKernelGGL((attn_relpos_f16s<DK, NWV, NQ, KS>), grid, dim3(64 * NWV * KS), 0, s, qkv, qkv_bs, qkv_cs, Qp, Kp, Vp, emb_k,
                     emb_v, lengths, out, o_bs, o_cs, H, T, window, n_heads, B);
}

// workspace: 3 * attn_pack_bytes(B, n_heads, dk, T), 256-byte aligned
// qkv == NULL: the images in `workspace` are already packed (launch_attn_qkv_pack_f16s)
hipError_t launch_attention_f16s(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k, const float* emb_v,
                                 const int64_t* lengths, float* out, long o_bs, long o_cs, int B, int H, int n_heads, int T,
                                 int window, void* workspace, hipStream_t s) {
  if (n_heads <= 0 || H % n_heads != 0 || 2 * window + 1 > 16 || T <= 0 || !workspace) return hipErrorInvalidValue;
  const int dk = H / n_heads;
  const size_t one = attn_pack_bytes(B, n_heads, dk, T);
  _Float16* Qp = static_cast<_Float16*>(workspace);
  _Float16* Kp = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + one);
  _Float16* Vp = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + 2 * one);
  const dim3 pgrid((T + 63) / 64, n_heads, 3 * B);
  // 128-query blocks when they fill the chip -- eight waves of one query tile each (C3 attention 0.66 -> 0.60 ms against
  // four waves of two tiles: same K / V traffic per query, twice the waves to overlap softmax and MFMAs) --; 64-query
  // blocks with two key-tile streams otherwise
  const bool small = (long)((T + 127) / 128) * n_heads * B < 512;
#define VSP_ATTF(DKV)                                                                                                   \
  if (qkv) hipLaunchKernelGGL((attn_pack_f16s<DKV>), pgrid, dim3(256), 0, s, qkv, qkv_bs, qkv_cs, H, T, Qp, Kp, Vp);    \
  if (small) launch_attn_f16s<DKV, 4, 1, 2>(qkv, qkv_bs, qkv_cs, Qp, Kp, Vp, emb_k, emb_v, lengths, out, o_bs, o_cs, B, H, n_heads, T, window, s); \
  else launch_attn_f16s<DKV, 8, 1, 1>(qkv, qkv_bs, qkv_cs, Qp, Kp, Vp, emb_k, emb_v, lengths, out, o_bs, o_cs, B, H, n_heads, T, window, s)
  if (dk == 96) { VSP_ATTF(96); }
  else if (dk == 64) { VSP_ATTF(64); }
  else if (dk == 32) { VSP_ATTF(32); }
  else return hipErrorInvalidValue;
#undef VSP_ATTF
  return hipGetLastError();
}


// the projections + packing launch (attn_qkv_pack_f16s): x [B][H][T], wg / bias of the stacked 3H x H projection
bool attn_qkv_pack_supported(int H, int n_heads) { return n_heads == 2 && (H == 192 || H == 128 || H == 64); }
hipError_t launch_attn_qkv_pack_f16s(const float* x, long x_bs, long x_cs, const uint16_t* wg, const float* bias,
                                     const int64_t* lengths, int B, int H, int n_heads, int T, void* workspace, hipStream_t s) {
  if (!attn_qkv_pack_supported(H, n_heads) || !wg || !bias || !workspace || B <= 0 || T <= 0) return hipErrorInvalidValue;
  const int dk = H / n_heads;
  const size_t one = attn_pack_bytes(B, n_heads, dk, T);
  _Float16* Qp = static_cast<_Float16*>(workspace);
  _Float16* Kp = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + one);
  _Float16* Vp = reinterpret_cast<_Float16*>(static_cast<char*>(workspace) + 2 * one);
  const dim3 grid((T + 63) / 64, 3, B);
  const int lds = cc_image_bytes(H) + H * CC_TS * 4;
#define VSP_QKVP(DKV)                                                                                                    \
  {                                                                                                                      \
    static std::atomic<uint64_t> attr_done{0};                                                                           \
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(attn_qkv_pack_f16s<DKV, 2>), lds, attr_done); e != hipSuccess) return e; \
    hipLaunchKernelGGL((attn_qkv_pack_f16s<DKV, 2>), grid, dim3(256), lds, s, x, x_bs, x_cs, wg, bias, lengths, T, Qp, Kp, Vp); \
  }
  if (dk == 96) VSP_QKVP(96)
  else if (dk == 64) VSP_QKVP(64)
  else VSP_QKVP(32)
#undef VSP_QKVP
  return hipGetLastError();
}

}  // namespace vsp
