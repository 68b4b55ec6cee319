// Small HBM-bound kernels of the synthesis path: channel LayerNorm, embeddings, variance-adaptor
// formulas, the wave64 prefix-sum length regulator, reparameterisation, conv_post + tanh, and the
// rational-quadratic spline.  All tensors are [B][C][T] float32 with T contiguous; threads run
// along T so that every global access is coalesced.
#include "kernels.h"

#include <math.h>

namespace vsp {

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// LayerNorm over channels (reference modules.py:29-32: transpose, F.layer_norm(eps=1e-5), transpose;
// frame_prior_network.py:80,88 nn.LayerNorm on the transposed tensor is the same arithmetic).
// block = 64 time steps x 4 channel groups.
__global__ void __launch_bounds__(256) layernorm_ct(const float* __restrict__ x, long x_bs, long x_cs,
                                                    const float* __restrict__ res, long r_bs, long r_cs,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ y,
                                                    long y_bs, long y_cs, int C, int T) {
  __shared__ float red[4][64];
  const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + tl;
  const bool ok = t < T;
  const float* xb = x + (size_t)b * x_bs + t;
  const float* rb = res ? res + (size_t)b * r_bs + t : nullptr;
  float s = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 4) s += xb[(size_t)c * x_cs] + (rb ? rb[(size_t)c * r_cs] : 0.f);
  red[cg][tl] = s;
  __syncthreads();
  const float mean = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  __syncthreads();
  float v2 = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 4) {
      const float d = xb[(size_t)c * x_cs] + (rb ? rb[(size_t)c * r_cs] : 0.f) - mean;
      v2 += d * d;
    }
  red[cg][tl] = v2;
  __syncthreads();
  const float var = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (ok) {
    float* yb = y + (size_t)b * y_bs + t;
    for (int c = cg; c < C; c += 4) {
      const float v = xb[(size_t)c * x_cs] + (rb ? rb[(size_t)c * r_cs] : 0.f);
      yb[(size_t)c * y_cs] = (v - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

// The same with the column's values held in registers (C <= 4 * NV): ONE read of x (+ res) instead of three, all
// loads of a thread in flight at once.  Sums run in the same order as above: identical results.
// FULL: C == 4 NV exactly (the 192-channel encoders): no per-channel predicate at all -- with them the kernel kept 48 saved
// exec masks alive across its four loops (113 spilled scalar registers, 100 exec branches); lanes beyond T read column T - 1
// and store nothing.
template <int NV, bool FULL>
__global__ void __launch_bounds__(256) layernorm_ct_reg(const float* __restrict__ x, long x_bs, long x_cs,
                                                        const float* __restrict__ res, long r_bs, long r_cs,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long y_bs, long y_cs, int C, int T) {
  __shared__ float red[4][64];
  const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + tl;
  const bool ok = t < T;
  const int tt = ok ? t : T - 1;
  // (32-bit element offsets within one utterance's tensor: the launcher checks C * stride < 2^31)
  const int xs = (int)x_cs, rs = (int)r_cs, ys = (int)y_cs;
  const float* xb = x + (size_t)b * x_bs + tt;
  const float* rb = res ? res + (size_t)b * r_bs + tt : nullptr;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = cg + 4 * i;
    if constexpr (FULL) v[i] = xb[c * xs];
    else v[i] = c < C ? xb[c * xs] : 0.f;
  }
  if (rb) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = cg + 4 * i;
      if constexpr (FULL) v[i] += rb[c * rs];
      else if (c < C) v[i] += rb[c * rs];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (FULL || cg + 4 * i < C) s += v[i];
  red[cg][tl] = s;
  __syncthreads();
  const float mean = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  __syncthreads();
  float v2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (FULL || cg + 4 * i < C) {
      const float d = v[i] - mean;
      v2 += d * d;
    }
  red[cg][tl] = v2;
  __syncthreads();
  const float var = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (ok) {
    float* yb = y + (size_t)b * y_bs + t;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = cg + 4 * i;
      if (FULL || c < C) yb[c * ys] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

hipError_t launch_layernorm(const float* x, long x_bs, long x_cs, const float* res, long r_bs, long r_cs,
                            const float* gamma, const float* beta, float* y, long y_bs, long y_cs, int B, int C,
                            int T, hipStream_t s) {
  if (C <= 4 * 48 && T > 0 && (long)C * x_cs < (1L << 31) && (long)C * y_cs < (1L << 31) && (!res || (long)C * r_cs < (1L << 31))) {
    if (C == 4 * 48)
      hipLaunchKernelGGL((layernorm_ct_reg<48, true>), dim3(cdiv(T, 64), B), dim3(256), 0, s, x, x_bs, x_cs, res, r_bs, r_cs,
                         gamma, beta, y, y_bs, y_cs, C, T);
    else
      hipLaunchKernelGGL((layernorm_ct_reg<48, false>), dim3(cdiv(T, 64), B), dim3(256), 0, s, x, x_bs, x_cs, res, r_bs, r_cs,
                         gamma, beta, y, y_bs, y_cs, C, T);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(layernorm_ct, dim3(cdiv(T, 64), B), dim3(256), 0, s, x, x_bs, x_cs, res, r_bs, r_cs, gamma,
                     beta, y, y_bs, y_cs, C, T);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// symbol embedding * sqrt(H) (reference models.py:169-170), speaker embedding (models.py:675)
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb, int n_vocab,
                             float scale, float* __restrict__ x, long x_bs, long x_cs, int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  long id = ids[(size_t)b * T + t];
  id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);
  x[(size_t)b * x_bs + (size_t)c * x_cs + t] = emb[(size_t)id * C + c] * scale;
}
hipError_t launch_embed(const int64_t* ids, const float* emb, int n_vocab, float scale, float* x, long x_bs,
                        long x_cs, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(cdiv(T, 64), C, B), dim3(64), 0, s, ids, emb, n_vocab, scale, x, x_bs,
                     x_cs, C, T);
  return hipGetLastError();
}

__global__ void gather_rows_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table, int n_rows,
                                   float* __restrict__ out, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  long r = idx[b];
  r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
  out[(size_t)b * C + c] = table[(size_t)r * C + c];
}
hipError_t launch_gather_rows(const int64_t* idx, const float* table, int n_rows, float* out, int B, int C,
                              hipStream_t s) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(C, 64), B), dim3(64), 0, s, idx, table, n_rows, out, C);
  return hipGetLastError();
}

// x + cond(g) broadcast over time (reference models.py:122, 507; frame_prior_network.py:121)
__global__ void add_cond_kernel(const float* __restrict__ x, long x_bs, long x_cs, const float* __restrict__ cond,
                                long cond_bs, float* __restrict__ y, long y_bs, long y_cs, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  y[(size_t)b * y_bs + (size_t)c * y_cs + t] =
      x[(size_t)b * x_bs + (size_t)c * x_cs + t] + cond[(size_t)b * cond_bs + c];
}
hipError_t launch_add_cond(const float* x, long x_bs, long x_cs, const float* cond, long cond_bs, float* y,
                           long y_bs, long y_cs, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(add_cond_kernel, dim3(cdiv(T, 64), C, B), dim3(64), 0, s, x, x_bs, x_cs, cond, cond_bs, y,
                     y_bs, y_cs, T);
  return hipGetLastError();
}

__global__ void copy3_kernel(const float* __restrict__ x, long x_bs, long x_cs, float* __restrict__ y, long y_bs,
                             long y_cs, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  y[(size_t)b * y_bs + (size_t)c * y_cs + t] = x[(size_t)b * x_bs + (size_t)c * x_cs + t];
}
hipError_t launch_copy3(const float* x, long x_bs, long x_cs, float* y, long y_bs, long y_cs, int B, int C, int T,
                        hipStream_t s) {
  hipLaunchKernelGGL(copy3_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, s, x, x_bs, x_cs, y, y_bs, y_cs, T);
  return hipGetLastError();
}

__global__ void mask3_kernel(float* __restrict__ x, long x_bs, long x_cs, const int64_t* __restrict__ lengths, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T || t < (int)lengths[b]) return;
  x[(size_t)b * x_bs + (size_t)c * x_cs + t] = 0.f;
}
hipError_t launch_mask3(float* x, long x_bs, long x_cs, const int64_t* lengths, int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(mask3_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, s, x, x_bs, x_cs, lengths, T);
  return hipGetLastError();
}

// 1-output-channel 1x1 conv / Linear (reference models.py:131, 513; frame_prior_network.py:107)
__global__ void chan_dot_kernel(const float* __restrict__ x, long x_bs, long x_cs, const float* __restrict__ w,
                                const float* __restrict__ bias, const int64_t* __restrict__ lengths, int mask_in,
                                int mask_out, float* __restrict__ out, int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const bool valid = lengths ? t < (int)lengths[b] : true;
  const float* xb = x + (size_t)b * x_bs + t;
  float s = 0.f;
  if (!(mask_in && !valid))
    for (int c = 0; c < C; ++c) s += w[c] * xb[(size_t)c * x_cs];
  s += bias ? bias[0] : 0.f;
  if (mask_out && !valid) s = 0.f;
  out[(size_t)b * T + t] = s;
}
hipError_t launch_chan_dot(const float* x, long x_bs, long x_cs, const float* w, const float* bias,
                           const int64_t* lengths, int mask_in, int mask_out, float* out, int B, int C, int T,
                           hipStream_t s) {
  hipLaunchKernelGGL(chan_dot_kernel, dim3(cdiv(T, 64), B), dim3(64), 0, s, x, x_bs, x_cs, w, bias, lengths,
                     mask_in, mask_out, out, C, T);
  return hipGetLastError();
}

// x += Conv1d(1, C, 3, padding=1)(sig)  -- pitch_prenet / energy_prenet, UNMASKED (models.py:697, 707)
__global__ void prenet_add_kernel(float* __restrict__ x, long x_bs, long x_cs, const float* __restrict__ w,
                                  const float* __restrict__ bias, const float* __restrict__ sig, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* sb = sig + (size_t)b * T;
  const float sm = t > 0 ? sb[t - 1] : 0.f, s0 = sb[t], sp = t + 1 < T ? sb[t + 1] : 0.f;
  float v = w[c * 3 + 0] * sm;
  v += w[c * 3 + 1] * s0;
  v += w[c * 3 + 2] * sp;
  v += bias[c];
  x[(size_t)b * x_bs + (size_t)c * x_cs + t] += v;
}
hipError_t launch_prenet_add(float* x, long x_bs, long x_cs, const float* w, const float* bias, const float* sig,
                             int B, int C, int T, hipStream_t s) {
  hipLaunchKernelGGL(prenet_add_kernel, dim3(cdiv(T, 64), C, B), dim3(64), 0, s, x, x_bs, x_cs, w, bias, sig, T);
  return hipGetLastError();
}

// duration = ceil((exp(logw) * mask - 1) * duration_control)   (reference models.py:686-688)
__global__ void duration_kernel(const float* __restrict__ logw, const int64_t* __restrict__ lengths, float scale,
                                float* __restrict__ dur, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const float m = t < (int)lengths[b] ? 1.f : 0.f;
  const float e = expf(logw[(size_t)b * T + t]) * m;
  dur[(size_t)b * T + t] = ceilf((e - 1.f) * scale);
}
hipError_t launch_duration_from_logw(const float* logw, const int64_t* lengths, float scale, float* dur, int B,
                                     int T, hipStream_t s) {
  hipLaunchKernelGGL(duration_kernel, dim3(cdiv(T, 64), B), dim3(64), 0, s, logw, lengths, scale, dur, T);
  return hipGetLastError();
}

// LF0 / F0 (reference models.py:691-698; the 2590 in the F0 formula is the reference's constant)
__global__ void pitch_kernel(const float* __restrict__ pitch_ctl, const float* __restrict__ lf0_pred, float scale,
                             float* __restrict__ lf0, float* __restrict__ f0, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float l;
  if (pitch_ctl) {
    l = (2595.f * log10f(1.f + pitch_ctl[i] / 700.f)) / 500.f;
  } else {
    l = lf0_pred[i] * scale;
  }
  lf0[i] = l;
  f0[i] = (powf(10.f, l * 500.f / 2590.f) - 1.f) * 700.f;
}
hipError_t launch_pitch(const float* pitch_ctl, const float* lf0_pred, float scale, float* lf0, float* f0, int n,
                        hipStream_t s) {
  hipLaunchKernelGGL(pitch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, pitch_ctl, lf0_pred, scale, lf0, f0, n);
  return hipGetLastError();
}

// energy (reference models.py:701-708)
__global__ void energy_kernel(const float* __restrict__ energy_ctl, const float* __restrict__ e_pred, float scale,
                              float* __restrict__ norm_e, float* __restrict__ energy, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ne;
  if (energy_ctl) {
    ne = (energy_ctl[i] - 60.f) / 36.f;
  } else {
    ne = (((e_pred[i] * 36.f + 60.f) * scale) - 60.f) / 36.f;
  }
  norm_e[i] = ne;
  energy[i] = ne * 36.f + 60.f;
}
hipError_t launch_energy(const float* energy_ctl, const float* e_pred, float scale, float* norm_e, float* energy,
                         int n, hipStream_t s) {
  hipLaunchKernelGGL(energy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, energy_ctl, e_pred, scale, norm_e, energy,
                     n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Length regulator (reference models.py:398-427).  The reference loops over phonemes in Python
// with one .item() host sync each; here one wave64 per utterance does an inclusive prefix sum of
// reps_i = max(int(d_i), 0) (DPP shuffles, carry across 64-wide chunks), and the expand is a
// per-frame upper_bound into that prefix sum.
__global__ void __launch_bounds__(64) duration_cumsum_kernel(const float* __restrict__ dur, int32_t* __restrict__ cum,
                                                             int64_t* __restrict__ frame_lengths, int Tp,
                                                             unsigned* __restrict__ flags) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < Tp; base += 64) {
    const int i = base + lane;
    int v = 0;
    if (i < Tp) {
      const float d = dur[(size_t)b * Tp + i];
      v = d > 0.f ? (int)d : 0;  // int() truncation, negatives -> 0 (models.py:424)
      // a duration that is not a number, or beyond anything an utterance holds (2^20 frames = 3.4 hours): the predictor's
      // activations left the split-f16 range, or the caller's tensor is broken -- counted as 0 frames and flagged
      // (vsp_status) instead of overflowing the prefix sum into a garbage frame count
      if (!(d <= 1048576.f)) { v = 0; if (flags) vsp_raise_flag(flags, VSP_FLAG_NONFINITE_LATENT); }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int n = __shfl_up(v, off);
      if (lane >= off) v += n;
    }
    v += carry;
    if (i < Tp) cum[(size_t)b * Tp + i] = v;
    carry = __shfl(v, 63);
  }
  if (lane == 0) frame_lengths[b] = carry;
}
hipError_t launch_duration_cumsum(const float* dur, int32_t* cum, int64_t* frame_lengths, int B, int Tp,
                                  hipStream_t s, unsigned* flags) {
  hipLaunchKernelGGL(duration_cumsum_kernel, dim3(B), dim3(64), 0, s, dur, cum, frame_lengths, Tp, flags);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) length_regulate_kernel(const float* __restrict__ x, long x_bs, long x_cs,
                                                              const int32_t* __restrict__ cum,
                                                              float* __restrict__ out, long o_bs, long o_cs, int C,
                                                              int Tp, int Tf, int c_per_block) {
  const int f = blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
  const int c0 = blockIdx.y * c_per_block;
  if (f >= Tf) return;
  const int32_t* cb = cum + (size_t)b * Tp;
  const int total = cb[Tp - 1];
  // upper_bound: first idx with cum[idx] > f
  int lo = 0, hi = Tp;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cb[mid] > f) hi = mid; else lo = mid + 1;
  }
  const bool valid = f < total;
  const int idx = lo < Tp ? lo : Tp - 1;
  const float* xb = x + (size_t)b * x_bs + idx;
  float* ob = out + (size_t)b * o_bs + f;
  const int c1 = c0 + c_per_block < C ? c0 + c_per_block : C;
  for (int c = c0; c < c1; ++c) ob[(size_t)c * o_cs] = valid ? xb[(size_t)c * x_cs] : 0.f;
}
hipError_t launch_length_regulate(const float* x, long x_bs, long x_cs, const int32_t* cum, float* out, long o_bs,
                                  long o_cs, int B, int C, int Tp, int Tf, hipStream_t s) {
  const int cpb = 24;
  hipLaunchKernelGGL(length_regulate_kernel, dim3(cdiv(Tf, 256), cdiv(C, cpb), B), dim3(256), 0, s, x, x_bs, x_cs,
                     cum, out, o_bs, o_cs, C, Tp, Tf, cpb);
  return hipGetLastError();
}

// z_p = m_p + noise * exp(logs_p) * noise_scale  (reference models.py:718)
__global__ void reparam_kernel(const float* __restrict__ m_p, const float* __restrict__ logs_p,
                               const float* __restrict__ noise, float noise_scale, float* __restrict__ z_p, long n,
                               float* __restrict__ copy, unsigned* __restrict__ flags) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float nz = noise ? noise[i] : 0.f;
  const float v = m_p[i] + nz * expf(logs_p[i]) * noise_scale;
  z_p[i] = v;
  if (copy) copy[i] = v;        // (the tensor the inverse flow then transforms in place: saves a copy launch)
  // the frame-rate stages behind m_p / logs_p (text encoder, frame prior network, projection) left the split-f16 range
  // (conv_mfma.hip: an operand beyond +-65504 becomes inf) or the caller's noise is not finite: sticky flag, vsp_status
  if (flags && !(fabsf(v) <= 3.0e38f)) vsp_raise_flag(flags, VSP_FLAG_NONFINITE_LATENT);
}
hipError_t launch_reparam(const float* m_p, const float* logs_p, const float* noise, float noise_scale, float* z_p,
                          long n, hipStream_t s, float* copy, unsigned* flags) {
  hipLaunchKernelGGL(reparam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, m_p, logs_p, noise, noise_scale, z_p, n, copy, flags);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Standard-normal draws for callers without a generator of their own (the torch.randn_like of reference
// models.py:718 / :240): Philox4x32-10 (Salmon et al., SC'11; key = seed lo | hi, counter = (i / 4, 0, 0, 0)) ->
// four 32-bit words -> two Box-Muller pairs, element i = word i % 4 of counter i / 4.  A function of (seed, i) only:
// the same seed gives the same tensor on any grid and any GPU.  NOT torch's generator: a torch.manual_seed(s) draw
// is a different sequence (documented in include/vispeech_hip.h).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// `first` = index of out[0] in the stream: a shard of a larger tensor draws ITS elements (sharded batches do not depend on
// the shard layout).  Uniforms: the top 23 bits, (x + 0.5) / 2^23 -- exact in fp32, strictly inside (0, 1).
__global__ void __launch_bounds__(256) randn_kernel(uint32_t k0, uint32_t k1, long first, long n, float* __restrict__ out) {
  const long q = (first >> 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;   // counter = group of four stream elements
  if (4 * q >= first + n) return;
  uint32_t w[4];
  philox4x32_10((uint32_t)(q & 0xffffffffu), (uint32_t)((unsigned long)q >> 32), 0u, 0u, k0, k1, w);
  float v[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)(w[2 * p] >> 9) + 0.5f) * (1.f / 8388608.f);
    const float u2 = ((float)(w[2 * p + 1] >> 9) + 0.5f) * (1.f / 8388608.f);
    const float rad = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    v[2 * p] = rad * cs;
    v[2 * p + 1] = rad * sn;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long i = 4 * q + j - first;
    if (i >= 0 && i < n) out[i] = v[j];
  }
}
hipError_t launch_randn(uint64_t seed, long first, long n, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (first < 0) return hipErrorInvalidValue;
  const long groups = ((first + n + 3) >> 2) - (first >> 2);
  hipLaunchKernelGGL(randn_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, s, (uint32_t)(seed & 0xffffffffu),
                     (uint32_t)(seed >> 32), first, n, out);
  return hipGetLastError();
}

__global__ void mask_u8_kernel(const int64_t* __restrict__ lengths, uint8_t* __restrict__ mask, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  mask[(size_t)b * T + t] = t < (int)lengths[b] ? 1 : 0;
}
hipError_t launch_mask_u8(const int64_t* lengths, uint8_t* mask, int B, int T, hipStream_t s) {
  hipLaunchKernelGGL(mask_u8_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, s, lengths, mask, T);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// conv_post (C -> 1, k taps, no bias) with the leaky-relu prologue and tanh epilogue
// (reference models.py:286-288; the slope here is F.leaky_relu's default 0.01, gotcha G1).
// One block = 512 outputs; x tile staged (activated) in LDS; each thread produces 4 consecutive
// samples from three ds_read_b128 per channel.
constexpr int CP_TILE = 512;
constexpr int CP_MAXK = 8;
__global__ void __launch_bounds__(128) conv_post_kernel(const float* __restrict__ x, long x_bs, long x_cs,
                                                        const float* __restrict__ w, int C, int K, float slope,
                                                        float* __restrict__ o, long o_bs, int T,
                                                        unsigned* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int LW = CP_TILE + 8;
  float* xs = sm;            // [C][LW]
  float* ws = sm + C * LW;   // [C][CP_MAXK]
  const int b = blockIdx.y, t0 = blockIdx.x * CP_TILE, pad = (K - 1) / 2;
  const float* xb = x + (size_t)b * x_bs;
  for (int idx = threadIdx.x; idx < C * CP_MAXK; idx += 128) {
    const int c = idx / CP_MAXK, j = idx % CP_MAXK;
    ws[idx] = j < K ? w[c * K + j] : 0.f;
  }
  for (int c = 0; c < C; ++c) {
    for (int col = threadIdx.x; col < LW; col += 128) {
      const int t = t0 - pad + col;
      float v = 0.f;
      if (t >= 0 && t < T) {
        v = xb[(size_t)c * x_cs + t];
        v = v > 0.f ? v : v * slope;
      }
      xs[c * LW + col] = v;
    }
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int tl = threadIdx.x * 4;
  for (int c = 0; c < C; ++c) {
    const float4* p = reinterpret_cast<const float4*>(xs + c * LW + tl);
    const float4 v0 = p[0], v1 = p[1], v2 = p[2];
    const float xv[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
    for (int j = 0; j < CP_MAXK; ++j) {
      const float wj = ws[c * CP_MAXK + j];
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] += wj * xv[n + j];
    }
  }
  float* ob = o + (size_t)b * o_bs;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int t = t0 + tl + n;
    if (t < T) {
      ob[t] = tanhf(acc[n]);
      if (flags && !(fabsf(acc[n]) <= 3.0e38f)) vsp_raise_flag(flags, VSP_FLAG_NONFINITE_WAVE);
    }
  }
}
hipError_t launch_conv_post(const float* x, long x_bs, long x_cs, const float* w, int C, int K, float slope,
                            float* o, long o_bs, int B, int T, hipStream_t s, unsigned* flags) {
  if (K > CP_MAXK || C > 32) return hipErrorInvalidValue;
  const size_t lds = ((size_t)C * (CP_TILE + 8) + (size_t)C * CP_MAXK) * sizeof(float);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(conv_post_kernel), 140 * 1024, attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL(conv_post_kernel, dim3(cdiv(T, CP_TILE), B), dim3(128), lds, s, x, x_bs, x_cs, w, C, K, slope,
                     o, o_bs, T, flags);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Monotone rational-quadratic spline with linear tails (reference transforms.py:12-193 as called
// by ConvFlow, modules.py:380-386).  One thread per element; nb <= 16 bins.
constexpr int SP_MAXB = 16;
__global__ void rq_spline_kernel(long n, int nb, const float* __restrict__ x, const float* __restrict__ uw,
                                 const float* __restrict__ uh, const float* __restrict__ ud, int inverse,
                                 float tail_bound, float* __restrict__ y, float* __restrict__ lad) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
  const float xin = x[e];
  const bool inside = xin >= -tail_bound && xin <= tail_bound;
  const float xc = fminf(fmaxf(xin, -tail_bound), tail_bound);
  float cw[SP_MAXB + 1], ch[SP_MAXB + 1], dv[SP_MAXB + 1];
  auto knots = [&](const float* u, float min_sz, float* c) {
    float mx = u[0];
    for (int k = 1; k < nb; ++k) mx = fmaxf(mx, u[k]);
    float ex[SP_MAXB], sum = 0.f;
    for (int k = 0; k < nb; ++k) { ex[k] = expf(u[k] - mx); sum += ex[k]; }
    float run = 0.f;
    c[0] = -tail_bound;
    for (int k = 0; k < nb; ++k) {
      run += min_sz + (1.f - min_sz * nb) * (ex[k] / sum);
      c[k + 1] = 2.f * tail_bound * run + (-tail_bound);
    }
    c[nb] = tail_bound;
  };
  knots(uw + e * nb, min_w, cw);
  knots(uh + e * nb, min_h, ch);
  const float cst = logf(expf(1.f - min_d) - 1.f);  // transforms.py:73
  for (int k = 0; k <= nb; ++k) {
    const float u = (k == 0 || k == nb) ? cst : ud[e * (nb - 1) + k - 1];
    const float sp = u > 20.f ? u : log1pf(expf(u));  // F.softplus
    dv[k] = min_d + sp;
  }
  const float* loc = inverse ? ch : cw;
  int bin = 0;
  for (int k = 0; k <= nb; ++k) {
    const float lk = k == nb ? loc[k] + 1e-6f : loc[k];  // transforms.py:48
    bin += xc >= lk ? 1 : 0;
  }
  bin -= 1;
  bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
  const float in_cw = cw[bin], in_w = cw[bin + 1] - cw[bin];
  const float in_ch = ch[bin], in_h = ch[bin + 1] - ch[bin];
  const float delta = in_h / in_w, d0 = dv[bin], d1 = dv[bin + 1];
  float outv, ladv;
  if (inverse) {
    const float dy = xc - in_ch;
    const float s = d0 + d1 - 2.f * delta;
    const float a = dy * s + in_h * (delta - d0);
    const float bq = in_h * d0 - dy * s;
    const float c = -delta * dy;
    const float disc = bq * bq - 4.f * a * c;
    const float root = (2.f * c) / (-bq - sqrtf(disc));
    outv = root * in_w + in_cw;
    const float tt = root * (1.f - root);
    const float den = delta + s * tt;
    const float num = delta * delta * (d1 * root * root + 2.f * delta * tt + d0 * (1.f - root) * (1.f - root));
    ladv = -(logf(num) - 2.f * logf(den));
  } else {
    const float theta = (xc - in_cw) / in_w;
    const float tt = theta * (1.f - theta);
    const float s = d0 + d1 - 2.f * delta;
    const float den = delta + s * tt;
    outv = in_ch + in_h * (delta * theta * theta + d0 * tt) / den;
    const float num = delta * delta * (d1 * theta * theta + 2.f * delta * tt + d0 * (1.f - theta) * (1.f - theta));
    ladv = logf(num) - 2.f * logf(den);
  }
  y[e] = inside ? outv : xin;
  lad[e] = inside ? ladv : 0.f;
}
hipError_t launch_rq_spline(int64_t n, int nb, const float* x, const float* uw, const float* uh, const float* ud,
                            int inverse, float tail_bound, float* y, float* lad, hipStream_t s) {
  if (nb < 1 || nb > SP_MAXB || n < 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(rq_spline_kernel, dim3(cdiv(n, 128)), dim3(128), 0, s, (long)n, nb, x, uw, uh, ud, inverse,
                     tail_bound, y, lad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// spectrogram front end (reference mel_processing.py:50-69): frames of the reflect-padded signal as the
// [n_fft][T] "channel x time" operand of the DFT conv, then the magnitude of the (re, im) rows.
__global__ void stft_frames_kernel(const float* __restrict__ audio, long a_bs, float* __restrict__ f, long f_bs, long f_cs,
                                   int L, int n_fft, int hop, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const int pad = (n_fft - hop) / 2;
  int i = t * hop + n - pad;
  if (i < 0) i = -i;                       // torch reflect padding (edge sample not repeated)
  if (i >= L) i = 2 * (L - 1) - i;
  f[(size_t)b * f_bs + (size_t)n * f_cs + t] = audio[(size_t)b * a_bs + i];
}
hipError_t launch_stft_frames(const float* audio, long a_bs, float* f, long f_bs, long f_cs, int B, int L, int n_fft,
                              int hop, int T, hipStream_t s) {
  hipLaunchKernelGGL(stft_frames_kernel, dim3((T + 255) / 256, n_fft, B), dim3(256), 0, s, audio, a_bs, f, f_bs, f_cs, L,
                     n_fft, hop, T);
  return hipGetLastError();
}

__global__ void stft_magnitude_kernel(const float* __restrict__ ri, long r_bs, long r_cs, float* __restrict__ spec,
                                      int spec_ch, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float re = ri[(size_t)b * r_bs + (size_t)r * r_cs + t];
  const float im = ri[(size_t)b * r_bs + (size_t)(spec_ch + r) * r_cs + t];
  spec[((size_t)b * spec_ch + r) * T + t] = sqrtf(re * re + im * im + 1e-6f);
}
hipError_t launch_stft_magnitude(const float* ri, long r_bs, long r_cs, float* spec, int B, int spec_ch, int T,
                                 hipStream_t s) {
  hipLaunchKernelGGL(stft_magnitude_kernel, dim3((T + 255) / 256, spec_ch, B), dim3(256), 0, s, ri, r_bs, r_cs, spec,
                     spec_ch, T);
  return hipGetLastError();
}

// mel projection + dynamic range compression (reference mel_processing.py:16-22, 73-82).  A mel filter is a triangle
// over a short run of bins: only [lo, hi) is visited, in increasing bin order.
__global__ void spec_to_mel_kernel(const float* __restrict__ spec, const float* __restrict__ basis,
                                   const int* __restrict__ lo, const int* __restrict__ hi, float* __restrict__ mel,
                                   int n_freq, int n_mels, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* sp = spec + (size_t)b * n_freq * T + t;
  const float* w = basis + (size_t)m * n_freq;
  float acc = 0.f;
  for (int f = lo[m]; f < hi[m]; ++f) acc = fmaf(w[f], sp[(size_t)f * T], acc);
  mel[((size_t)b * n_mels + m) * T + t] = logf(fmaxf(acc, 1e-5f));
}
hipError_t launch_spec_to_mel(const float* spec, const float* basis, const int* lo, const int* hi, float* mel, int B,
                              int n_freq, int n_mels, int T, hipStream_t s) {
  hipLaunchKernelGGL(spec_to_mel_kernel, dim3((T + 255) / 256, n_mels, B), dim3(256), 0, s, spec, basis, lo, hi, mel,
                     n_freq, n_mels, T);
  return hipGetLastError();
}

}  // namespace vsp
