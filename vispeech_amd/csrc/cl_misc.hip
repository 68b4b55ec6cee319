// Small channels-last kernels of the vocoder: the [B][C][T] -> [B][T][C] transpose of conv_pre's output and conv_post.
// (The generator's convolutions are in gen16.hip.)
#include "kernels.h"

namespace vsp {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// [B][C][T] -> [B][T][C] (conv_pre's output enters the channels-last vocoder), 32x32 LDS tiles
__global__ void __launch_bounds__(256) transpose_ct_kernel(const float* __restrict__ x, long x_bs, long x_cs,
                                                           float* __restrict__ y, long y_bs, int y_ts, int C, int T,
                                                           float scale) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ly + 8 * k, t = t0 + lx;
    tile[ly + 8 * k][lx] = (c < C && t < T) ? x[(size_t)b * x_bs + (size_t)c * x_cs + t] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = t0 + ly + 8 * k, c = c0 + lx;
    if (t < T && c < C) y[(size_t)b * y_bs + (size_t)t * y_ts + c] = tile[lx][ly + 8 * k] * scale;
  }
}
hipError_t launch_transpose_ct(const float* x, long x_bs, long x_cs, float* y, long y_bs, int y_ts, int B, int C,
                               int T, hipStream_t s, float scale) {
  hipLaunchKernelGGL(transpose_ct_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, x, x_bs, x_cs, y,
                     y_bs, y_ts, C, T, scale);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// conv_post on a channels-last input: o[b][t] = tanh(sum_j sum_c w[c][j] * lrelu(x[b][t+j-pad][c]))
// (reference models.py:286-288; slope 0.01 = F.leaky_relu's default, gotcha G1).  Block = 256 outputs.
// The (256+K-1) x C window is one contiguous span of HBM: staged with 16-byte buffer loads (rows
// outside the utterance read as 0 = the zero padding), activated once, kept in LDS with rows padded
// to C+4 floats (16-byte aligned, conflict-free ds_read_b128 down a column of rows); the weights are
// read through the scalar unit (uniform addresses).
constexpr int CPL_TILE = 256;
template <int C>
__global__ void __launch_bounds__(256) conv_post_cl_kernel(const float* __restrict__ x, long x_bs,
                                                           const float* __restrict__ wt /* [K][C] */, int K,
                                                           float slope, float* __restrict__ o, long o_bs, int T,
                                                           const int* __restrict__ glen, int grate, float unscale,
                                                           unsigned* __restrict__ flags) {
  constexpr int RSF = C + 4;
  constexpr int C4 = C / 4;
  __shared__ __attribute__((aligned(16))) float xs[(CPL_TILE + 8) * RSF];
  const int b = blockIdx.y, t0 = blockIdx.x * CPL_TILE, pad = (K - 1) / 2;
  if (glen) {                                   // ragged batch: this utterance's tensor ends at glen[b] * grate
    T = __builtin_amdgcn_readfirstlane(glen[b]) * grate;
    if (t0 >= T) return;
  }
  const int rows = CPL_TILE + K - 1;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)b * x_bs, 0,
                                                                      T * C * 4, 0x00020000);
  const int base = (t0 - pad) * C * 4;
  for (int idx = threadIdx.x; idx < rows * C4; idx += 256) {
    const int row = idx / C4, c4 = idx % C4;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, base + idx * 16, 0, 0);
    float e[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = e[k] > 0.f ? e[k] : e[k] * slope;
    *reinterpret_cast<float4*>(xs + row * RSF + 4 * c4) = make_float4(e[0], e[1], e[2], e[3]);
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  float acc = 0.f;
  for (int j = 0; j < K; ++j) {
    const float* xr = xs + (threadIdx.x + j) * RSF;
    const float* wr = wt + j * C;             // uniform -> scalar loads
#pragma unroll
    for (int c4 = 0; c4 < C4; ++c4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * c4);
      acc += wr[4 * c4 + 0] * xv.x;
      acc += wr[4 * c4 + 1] * xv.y;
      acc += wr[4 * c4 + 2] * xv.z;
      acc += wr[4 * c4 + 3] * xv.w;
    }
  }
  // (the generator's activations arrive * the context's activation scale -- every layer between conv_pre and here is
  // positively homogeneous once the biases carry the scale too; conv_post has no bias: the sum is unscaled here, exactly)
  acc *= unscale;
  if (t < T) {
    o[(size_t)b * o_bs + t] = tanhf(acc);
    // an activation beyond the split-f16 range turns into inf / NaN inside the matrix kernels (g16_common.h) and arrives
    // here: raise the context's sticky flag (vsp_status) -- rare, so the atomic costs nothing
    if (flags && !(fabsf(acc) <= 3.0e38f)) vsp_raise_flag(flags, VSP_FLAG_NONFINITE_WAVE);
  }
}
hipError_t launch_conv_post_cl(const float* x, long x_bs, int x_ts, const float* w, int C, int K, float slope,
                               float* o, long o_bs, int B, int T, hipStream_t s, const int* glen, int grate, float unscale,
                               unsigned* flags) {
  if (K > 8 || x_ts != C || (C != 32 && C != 64)) return hipErrorInvalidValue;
  dim3 grid((T + CPL_TILE - 1) / CPL_TILE, B);
  if (C == 32)
    hipLaunchKernelGGL(conv_post_cl_kernel<32>, grid, dim3(256), 0, s, x, x_bs, w, K, slope, o, o_bs, T, glen, grate, unscale, flags);
  else
    hipLaunchKernelGGL(conv_post_cl_kernel<64>, grid, dim3(256), 0, s, x, x_bs, w, K, slope, o, o_bs, T, glen, grate, unscale, flags);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Operand images (kernels.h ClConvArgs::x_img / o_img): the producer writes the data rows [PADF, PADF + T) of every
// plane; the CL_IMG_PADF rows in front and the CL_IMG_PADB rows behind must read as zero (the convolution's zero padding
// and the last tile's overshoot).  One block per plane: planes = B * (C / 32) * 2 * 4.
// Ragged batch: utterance b's data rows end at glen[b] * grate; its CL_IMG_PADB zero rows sit right behind them (the
// rows further back are never read: a tile behind the utterance's end does not run).
__global__ void __launch_bounds__(256) cl_img_zero_pads_kernel(uint4* __restrict__ img, int T, int tpad, int planes_per_utt,
                                                               const int* __restrict__ glen, int grate) {
  uint4* pl = img + (size_t)blockIdx.x * tpad;
  const uint4 z = {0u, 0u, 0u, 0u};
  int end = tpad;
  if (glen) {
    T = glen[blockIdx.x / planes_per_utt] * grate;
    end = CL_IMG_PADF + T + CL_IMG_PADB;
  }
  for (int r = threadIdx.x; r < CL_IMG_PADF; r += 256) pl[r] = z;
  for (int r = CL_IMG_PADF + T + threadIdx.x; r < end; r += 256) pl[r] = z;
}
hipError_t launch_cl_img_zero_pads(uint16_t* img, int B, int C, int T, hipStream_t s, const int* glen, int grate) {
  if (B <= 0 || C % 32 || T <= 0 || (reinterpret_cast<uintptr_t>(img) & 15)) return hipErrorInvalidValue;
  const int ppu = (C / 32) * 8;
  const long planes = (long)B * ppu;
  hipLaunchKernelGGL(cl_img_zero_pads_kernel, dim3((unsigned)planes), dim3(256), 0, s, reinterpret_cast<uint4*>(img), T,
                     cl_img_tpad(T), ppu, glen, grate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Trimmed tails (kernels.h).  gen_plan: the frames the generator computes per utterance.
__global__ void gen_plan_kernel(const int64_t* __restrict__ lengths, int B, int T, int ext, int* __restrict__ glen) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long len = lengths[b];
  len = len < 0 ? 0 : len;
  const long e = len + ext;
  glen[b] = e < T ? (int)e : T;
}
hipError_t launch_gen_plan(const int64_t* lengths, int B, int T, int back, int fwd, int* glen, hipStream_t s) {
  if (!lengths || !glen || B <= 0 || T <= 0 || back < 0 || fwd < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gen_plan_kernel, dim3((B + 63) / 64), dim3(64), 0, s, lengths, B, T, back + 1 + fwd, glen);
  return hipGetLastError();
}
// gen_tail_fill: one block per utterance.  The computed tensor end -- frames [len + back + 1, len + back + 1 + fwd) -- is
// read into LDS first (the steady-state fill overwrites it), then frames [len + back + 1, T - fwd) get the steady-state
// frame len + back and frames [T - fwd, T) the saved tensor end.  Utterances the plan did not trim (glen[b] == T) are skipped.
__global__ void __launch_bounds__(1024) gen_tail_fill_kernel(float* __restrict__ o, long o_bs, const int64_t* __restrict__ lengths,
                                                             const int* __restrict__ glen, int T, int back, int fwd, int up) {
  extern __shared__ float tail[];                 // [fwd * up]
  const int b = blockIdx.x;
  if (glen[b] >= T) return;
  long len = lengths[b];
  len = len < 0 ? 0 : len;
  float* ob = o + (size_t)b * o_bs;
  const long s0 = (len + back) * up;              // the steady-state frame
  const long e0 = s0 + up;                        // the computed tensor end: fwd frames from here
  const int nt = fwd * up;
  for (int i = threadIdx.x; i < nt; i += blockDim.x) tail[i] = ob[e0 + i];
  __syncthreads();
  const long fill_end = (long)(T - fwd) * up;
  for (long i = e0 + threadIdx.x; i < fill_end; i += blockDim.x) ob[i] = ob[s0 + (i - e0) % up];
  for (int i = threadIdx.x; i < nt; i += blockDim.x) ob[fill_end + i] = tail[i];
}
hipError_t launch_gen_tail_fill(float* o, long o_bs, const int64_t* lengths, const int* glen, int B, int T, int back, int fwd,
                                int up, hipStream_t s) {
  if (!o || !lengths || !glen || B <= 0 || T <= 0 || back < 0 || fwd < 0 || up <= 0) return hipErrorInvalidValue;
  const size_t lds = (size_t)fwd * up * sizeof(float);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gen_tail_fill_kernel, dim3(B), dim3(1024), lds, s, o, o_bs, lengths, glen, T, back, fwd, up);
  return hipGetLastError();
}

}  // namespace vsp
