// Windowed relative-position self-attention (reference attentions.py:148-179 and the pad/reshape
// helpers :181-243) as a tiled two-pass kernel on v_mfma_f32_32x32x2_f32.  The T x T score matrix
// is never materialised (the reference builds >= 6 [B,heads,T,T] fp32 temporaries per layer).
//
//   S_ij = (q_i / sqrt(dk)) . k_j + [|j-i| <= w] (q_i / sqrt(dk)) . Ek[j-i+w]
//   S_ij = -1e4 where mask_i * mask_j == 0            (attentions.py:166; NOT -inf, gotcha G9)
//   P = softmax_j S ;  O_i = sum_j P_ij v_j + sum_{|j-i|<=w} P_ij Ev[j-i+w]
//
// Layout: qkv [B][3H][T] (time contiguous).  One block = 4 waves = 128 queries of one (b, head);
// each wave owns 32 queries.  Scores are computed TRANSPOSED (S^T = K Q^T: key on the register
// index, query on the lane) so that (a) the row max/sum is an in-register reduction plus one
// cross-half shuffle and (b) the P^T accumulator tile is directly the B operand of O^T = V^T P^T
// with a permuted k order -- no LDS round trip for P.
//   pass 1: row max m_i and sum l_i (online);  pass 2: p = exp(s - m) / l, O^T += V^T p.
// The 2w+1 band probabilities go to a small LDS table and are applied to Ev at the end.
#include "kernels.h"

namespace vsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_QB = 128;   // queries per block
constexpr int ATT_KT = 32;    // keys per tile
constexpr int ATT_RS = 17;    // row stride of the relative tables (2w+1 <= 16)

template <int DK>
__global__ void __launch_bounds__(256) attn_relpos_f32(const float* __restrict__ qkv, long bs, long cs,
                                                       const float* __restrict__ emb_k,
                                                       const float* __restrict__ emb_v,
                                                       const int64_t* __restrict__ lengths,
                                                       float* __restrict__ out, long o_bs, long o_cs, int H,
                                                       int T, int w) {
  constexpr int KS = DK / 2;   // k-steps of the QK^T product
  constexpr int DT = DK / 32;  // 32-row tiles of the head dimension
  __shared__ __attribute__((aligned(16))) float Ks[DK * ATT_KT];
  __shared__ __attribute__((aligned(16))) float Vs[DK * (ATT_KT + 1)];
  __shared__ float Rl[ATT_QB * ATT_RS];
  __shared__ float Pb[ATT_QB * ATT_RS];
  __shared__ float Evs[16 * DK];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, hd = blockIdx.y, i0 = blockIdx.x * ATT_QB;
  const int nrel = 2 * w + 1;
  const int len = lengths ? (int)lengths[b] : T;
  const float* qrow = qkv + (size_t)b * bs + (size_t)(hd * DK) * cs;
  const float* krow = qrow + (size_t)H * cs;
  const float* vrow = qrow + (size_t)(2 * H) * cs;
  const float scale = sqrtf((float)DK);
  const int i = i0 + wave * 32 + l31;  // this lane's query

  // Q fragments (B operand: lane holds q[d = 2s + h][i]) scaled as the reference does
  // (query / sqrt(k_channels), attentions.py:155)
  float qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = i < T ? qrow[(size_t)(2 * s + h) * cs + i] / scale : 0.f;

  // relative-key logits of the block's queries and the Ev table
  for (int idx = tid; idx < ATT_QB * nrel; idx += 256) {
    const int iq = idx % ATT_QB, r = idx / ATT_QB;
    float sum = 0.f;
    if (i0 + iq < T) {
      for (int d = 0; d < DK; ++d) sum += (qrow[(size_t)d * cs + i0 + iq] / scale) * emb_k[r * DK + d];
    }
    Rl[iq * ATT_RS + r] = sum;
  }
  for (int idx = tid; idx < ATT_QB * ATT_RS; idx += 256) Pb[idx] = 0.f;
  for (int idx = tid; idx < nrel * DK; idx += 256) Evs[idx] = emb_v[idx];

  const int iql = wave * 32 + l31;  // query index local to the block
  const int ntiles = (T + ATT_KT - 1) / ATT_KT;
  float m_run = -3.0e38f, l_run = 0.f;

  auto stage = [&](int j0, bool with_v) {
    for (int idx = tid; idx < DK * ATT_KT; idx += 256) {
      const int d = idx / ATT_KT, jl = idx % ATT_KT, j = j0 + jl;
      Ks[idx] = j < T ? krow[(size_t)d * cs + j] : 0.f;
      if (with_v) Vs[d * (ATT_KT + 1) + jl] = j < T ? vrow[(size_t)d * cs + j] : 0.f;
    }
  };
  auto scores = [&](int j0, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * s + h) * ATT_KT + l31], qf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float sv = acc[r];
      const int rel = j - i + w;
      if (rel >= 0 && rel < nrel) sv += Rl[iql * ATT_RS + rel];
      if (i >= len || j >= len) sv = -1e4f;
      if (j >= T) sv = -INFINITY;
      acc[r] = sv;
    }
  };

  // ---- pass 1: row statistics
  for (int jt = 0; jt < ntiles; ++jt) {
    __syncthreads();
    stage(jt * ATT_KT, false);
    __syncthreads();
    f32x16 sT;
    scores(jt * ATT_KT, sT);
    float tmax = sT[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sT[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += expf(sT[r] - m_new);
    l_run = l_run * expf(m_run - m_new) + part;
    m_run = m_new;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);

  // ---- pass 2: probabilities and P.V
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  for (int jt = 0; jt < ntiles; ++jt) {
    __syncthreads();
    stage(jt * ATT_KT, true);
    __syncthreads();
    f32x16 p;
    scores(jt * ATT_KT, p);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jt * ATT_KT + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float pv = expf(p[r] - m_run) / l_tot;
      p[r] = pv;
      const int rel = j - i + w;
      if (rel >= 0 && rel < nrel && j < T) Pb[iql * ATT_RS + rel] = pv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(dt * 32 + l31) * (ATT_KT + 1) + jl], p[r], o[dt], 0, 0, 0);
    }
  }
  __syncthreads();
  // ---- relative-value term and store (lane = query -> coalesced along time)
  if (i < T) {
    float pb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pb[r] = r < nrel ? Pb[iql * ATT_RS + r] : 0.f;
    float* orow = out + (size_t)b * o_bs + (size_t)(hd * DK) * o_cs + i;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float rv = 0.f;
        for (int e = 0; e < nrel; ++e) rv += pb[e] * Evs[e * DK + d];
        orow[(size_t)d * o_cs] = o[dt][r] + rv;
      }
    }
  }
}

hipError_t launch_attention(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k, const float* emb_v,
                            const int64_t* lengths, float* out, long o_bs, long o_cs, int B, int H,
                            int n_heads, int T, int window, hipStream_t s) {
  if (n_heads <= 0 || H % n_heads != 0 || 2 * window + 1 > 16 || T <= 0) return hipErrorInvalidValue;
  const int dk = H / n_heads;
  dim3 grid((T + ATT_QB - 1) / ATT_QB, n_heads, B);
  if (dk == 96) {
    hipLaunchKernelGGL(attn_relpos_f32<96>, grid, dim3(256), 0, s, qkv, qkv_bs, qkv_cs, emb_k, emb_v, lengths,
                       out, o_bs, o_cs, H, T, window);
  } else if (dk == 64) {
    hipLaunchKernelGGL(attn_relpos_f32<64>, grid, dim3(256), 0, s, qkv, qkv_bs, qkv_cs, emb_k, emb_v, lengths,
                       out, o_bs, o_cs, H, T, window);
  } else if (dk == 32) {
    hipLaunchKernelGGL(attn_relpos_f32<32>, grid, dim3(256), 0, s, qkv, qkv_bs, qkv_cs, emb_k, emb_v, lengths,
                       out, o_bs, o_cs, H, T, window);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace vsp
