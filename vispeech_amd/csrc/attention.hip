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

#include <cstdlib>

namespace vsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ATT_KT = 32;    // keys per tile
constexpr int ATT_RS = 17;    // row stride of the relative tables (2w+1 <= 16)

// NW waves per block.  KSPLIT = false: each wave owns 32 queries (128-query blocks) and walks every key tile.
// KSPLIT = true (long single utterances, where 128-query blocks would leave most CUs idle): the block
// owns 32 queries, every staging step brings NW key tiles and wave w multiplies tile w, so the key range is
// split NW ways inside the block; row statistics are merged through LDS after pass 1 and the partial O^T
// tiles are summed through LDS at the end -- NW x the blocks for the same MFMA work.
template <int DK, int NW, bool KSPLIT>
__global__ void __launch_bounds__(64 * NW, KSPLIT ? 1 : 2) attn_relpos_f32(const float* __restrict__ qkv, long bs, long cs,
                                                       const float* __restrict__ emb_k,
                                                       const float* __restrict__ emb_v,
                                                       const int64_t* __restrict__ lengths,
                                                       float* __restrict__ out, long o_bs, long o_cs, int H,
                                                       int T, int w) {
  constexpr int KS = DK / 2;   // k-steps of the QK^T product
  constexpr int DT = DK / 32;  // 32-row tiles of the head dimension
  constexpr int ATT_QB = KSPLIT ? 32 : 32 * NW, NTHR = 64 * NW;
  constexpr int KTA = KSPLIT ? ATT_KT * NW : ATT_KT;     // keys staged per step
  __shared__ __attribute__((aligned(16))) float KV[DK * KTA + DK * (KTA + 1)];
  float* const Ks = KV;
  float* const Vs = KV + DK * KTA;
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
  const int qoff = KSPLIT ? 0 : wave * 32;   // first query of this wave inside the block
  const int koff = KSPLIT ? wave * 32 : 0;   // this wave's key tile inside the staged keys
  const int i = i0 + qoff + l31;  // this lane's query

  // Q fragments (B operand: lane holds q[d = 2s + h][i]) scaled as the reference does
  // (query / sqrt(k_channels), attentions.py:155)
  float qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = i < T ? qrow[(size_t)(2 * s + h) * cs + i] / scale : 0.f;

  // relative-key logits of the block's queries and the Ev table
  for (int idx = tid; idx < ATT_QB * nrel; idx += NTHR) {
    const int iq = idx % ATT_QB, r = idx / ATT_QB;
    float sum = 0.f;
    if (i0 + iq < T) {
      for (int d = 0; d < DK; ++d) sum += (qrow[(size_t)d * cs + i0 + iq] / scale) * emb_k[r * DK + d];
    }
    Rl[iq * ATT_RS + r] = sum;
  }
  for (int idx = tid; idx < ATT_QB * ATT_RS; idx += NTHR) Pb[idx] = 0.f;
  for (int idx = tid; idx < nrel * DK; idx += NTHR) Evs[idx] = emb_v[idx];

  const int iql = qoff + l31;  // query index local to the block
  const int ntiles = (T + KTA - 1) / KTA;
  float m_run = -3.0e38f, l_run = 0.f;

  // K/V tile staging, split in two so that the loads of tile jt+1 are in flight while tile jt is being
  // multiplied: fetch() -> registers (coalesced along time), commit() -> LDS after the barrier that
  // retires the readers of the previous tile.
  constexpr int NST = (DK * KTA + NTHR - 1) / NTHR;
  float kreg[NST];
  [[maybe_unused]] float vreg[NST];
  auto fetch = [&](int j0, bool with_v) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int idx = tid + u * NTHR;
      const int d = idx / KTA, jl = idx % KTA, j = j0 + jl;
      const bool ok = idx < DK * KTA && j < T;
      kreg[u] = ok ? krow[(size_t)d * cs + j] : 0.f;
      if (with_v) vreg[u] = ok ? vrow[(size_t)d * cs + j] : 0.f;
    }
  };
  auto commit = [&](bool with_v) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int idx = tid + u * NTHR;
      if (idx < DK * KTA) {
        const int d = idx / KTA, jl = idx % KTA;
        Ks[idx] = kreg[u];
        if (with_v) Vs[d * (KTA + 1) + jl] = vreg[u];
      }
    }
  };
  auto scores = [&](int j0, f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * s + h) * KTA + koff + l31], qf[s], acc, 0, 0, 0);
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
  fetch(0, false);
  for (int jt = 0; jt < ntiles; ++jt) {
    __syncthreads();
    commit(false);
    __syncthreads();
    if (jt + 1 < ntiles) fetch((jt + 1) * KTA, false);
    f32x16 sT;
    scores(jt * KTA + koff, sT);
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
  float l_tot = l_run + __shfl_xor(l_run, 32);
  if constexpr (KSPLIT) {
    // merge the NW key ranges: m = max_w m_w, l = sum_w l_w * exp(m_w - m)
    __shared__ float Mx[NW * 32], Lx[NW * 32];
    if (h == 0) { Mx[wave * 32 + l31] = m_run; Lx[wave * 32 + l31] = l_tot; }
    __syncthreads();
    float mg = Mx[l31];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) mg = fmaxf(mg, Mx[ww * 32 + l31]);
    float lg = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) lg += Lx[ww * 32 + l31] * expf(Mx[ww * 32 + l31] - mg);
    m_run = mg;
    l_tot = lg;
  }

  // ---- pass 2: probabilities and P.V
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  fetch(0, true);
  for (int jt = 0; jt < ntiles; ++jt) {
    __syncthreads();
    commit(true);
    __syncthreads();
    if (jt + 1 < ntiles) fetch((jt + 1) * KTA, true);
    f32x16 p;
    scores(jt * KTA + koff, p);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jt * KTA + koff + (r & 3) + 8 * (r >> 2) + 4 * h;
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
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(dt * 32 + l31) * (KTA + 1) + koff + jl], p[r], o[dt], 0, 0, 0);
    }
  }
  __syncthreads();
  if constexpr (KSPLIT) {
    // sum the NW partial O^T tiles through LDS (the K/V staging area is free now), add the
    // relative-value term and store: thread -> (d, query), query fastest = coalesced along time
    float* const Ored = KV;                       // [NW][DK][32]
    static_assert(NW * DK * 32 <= DK * KTA + DK * (KTA + 1), "reduction buffer fits the staging area");
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Ored[(wave * DK + dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = o[dt][r];
    __syncthreads();
    for (int idx = tid; idx < DK * 32; idx += NTHR) {
      const int d = idx >> 5, q = idx & 31;
      if (i0 + q >= T) continue;
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) v += Ored[(ww * DK + d) * 32 + q];
      for (int e = 0; e < nrel; ++e) v += Pb[q * ATT_RS + e] * Evs[e * DK + d];
      out[(size_t)b * o_bs + (size_t)(hd * DK + d) * o_cs + i0 + q] = v;
    }
    return;
  }
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

template <int DK, int NW, bool KSPLIT>
static void launch_attn(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k, const float* emb_v,
                        const int64_t* lengths, float* out, long o_bs, long o_cs, int B, int H, int n_heads, int T,
                        int window, hipStream_t s) {
  constexpr int QB = KSPLIT ? 32 : 32 * NW;
  dim3 grid((T + QB - 1) / QB, n_heads, B);
  hipLaunchKernelGGL((attn_relpos_f32<DK, NW, KSPLIT>), grid, dim3(64 * NW), 0, s, qkv, qkv_bs, qkv_cs, emb_k, emb_v,
                     lengths, out, o_bs, o_cs, H, T, window);
}

hipError_t launch_attention(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k, const float* emb_v,
                            const int64_t* lengths, float* out, long o_bs, long o_cs, int B, int H,
                            int n_heads, int T, int window, int ksplit_mode, hipStream_t s) {
  if (n_heads <= 0 || H % n_heads != 0 || 2 * window + 1 > 16 || T <= 0) return hipErrorInvalidValue;
  const int dk = H / n_heads;
  // 128-query blocks (each wave its own queries) when they fill the chip; otherwise 32-query blocks whose
  // waves split the keys (ksplit_mode 0/1 forces one or the other, < 0 = automatic)
  const long blocks = (long)((T + 127) / 128) * n_heads * B;
  const bool ksplit = ksplit_mode >= 0 ? ksplit_mode != 0 : blocks < 256;
#define VSP_ATT(DKV)                                                                                             \
  if (ksplit) launch_attn<DKV, 4, true>(qkv, qkv_bs, qkv_cs, emb_k, emb_v, lengths, out, o_bs, o_cs, B, H, n_heads, T, window, s); \
  else launch_attn<DKV, 4, false>(qkv, qkv_bs, qkv_cs, emb_k, emb_v, lengths, out, o_bs, o_cs, B, H, n_heads, T, window, s)
  if (dk == 96) { VSP_ATT(96); }
  else if (dk == 64) { VSP_ATT(64); }
  else if (dk == 32) { VSP_ATT(32); }
  else return hipErrorInvalidValue;
#undef VSP_ATT
  return hipGetLastError();
}

}  // namespace vsp
