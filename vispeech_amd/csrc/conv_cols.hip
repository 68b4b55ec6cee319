// 1x1 frame-rate convolutions in column-tile form (conv_cols.h): the generic epilogue of conv_mfma.hip's kernels (bias,
// masks, scaling, residual, accumulate, two destinations -- reference modules.py:165-176, 324-343, models.py:526-529)
// and the LayerNorm form (x + conv_o(att) normalised over the channels in the same launch: reference
// attentions.py:40-42, modules.py:29-32).  launch_conv routes here for small grids; run_encoder_masked (api.hip) asks
// for the LayerNorm form directly.
#include "conv_cols.h"

namespace vsp {

constexpr int CC_ROWS = 192;      // rows per block at most: grid.y = ceil(M / 192)

// MW: m-tiles per wave (block rows <= 64 MW <= 192); LN: LayerNorm epilogue over exactly 192 rows
template <int MW, int CIN, bool LN>
__global__ void __launch_bounds__(256, 1) conv_cols_kernel(ConvArgs a, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta) {
  extern __shared__ __attribute__((aligned(16))) char cc_smem[];
  char* const img = cc_smem;
  float* const tile = reinterpret_cast<float*>(cc_smem + cc_image_bytes(CIN));
  const int b = blockIdx.z, t0 = blockIdx.x * CC_BT, row0 = blockIdx.y * CC_ROWS;
  const int tid = threadIdx.x, tl = tid & 63;
  const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: per-row bias / gamma / beta become scalar loads)
  const int T = a.T_in;
  const int rows = a.M - row0 < CC_ROWS ? a.M - row0 : CC_ROWS;       // this block's rows [row0, row0 + rows)
  const int MTB = rows >> 4;
  CcWeights<MW, CIN> W;
  cc_load_weights<MW, CIN>(W, a.wg, a.M >> 4, row0 >> 4, MTB);
  // per-row parameters (bias; gamma, beta of the LayerNorm form): one load per thread now, read back from LDS in the
  // epilogue (as scalar loads behind the contraction they were ~150 dependent s_loads; as per-thread vector loads 48 each)
  float* const par = tile + 64 * MW * CC_TS + 256;    // [3][192]
  if (tid < rows) {
    par[tid] = a.bias ? a.bias[row0 + tid] : 0.f;
    if constexpr (LN) { par[192 + tid] = gamma[tid]; par[384 + tid] = beta[tid]; }
  }
  int len = T;
  if (a.lengths) { const long l = a.lengths[b]; len = l < 0 ? 0 : (l < T ? (int)l : T); }
  cc_stage_x<CIN>(a.x + (size_t)b * a.x_bs, a.x_cs, t0, a.in_mask ? len : T, img);
  const int t = t0 + tl;
  const bool qin = t < a.Nq, valid = t < len;
  const int tt = qin ? t : a.Nq - 1;                  // (lanes beyond the tensor read its last column and store nothing)
  constexpr int NV = 16 * MW;                         // values per thread: rows rg, rg + 4, ..
  if constexpr (LN) {
    // y = LayerNorm_channels(conv + bias + res) * gamma + beta -- layernorm_ct_reg's arithmetic in its order (misc.hip):
    // thread (tl, rg) holds channels rg, rg + 4, ..; partial sums meet through LDS
    float* const red = tile + 64 * MW * CC_TS;        // [4][64]
    const float* rb = a.res ? a.res + (size_t)b * a.r_bs + tt : nullptr;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = rb ? rb[(long)(rg + 4 * i) * a.r_cs] : 0.f;     // (requested before the contraction)
    cc_contract<MW, CIN>(W, MTB, img, tile);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = rg + 4 * i;
      v[i] = (tile[c * CC_TS + tl] + par[c]) + v[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    red[rg * 64 + tl] = s;
    __syncthreads();
    const float mean = (red[tl] + red[64 + tl] + red[128 + tl] + red[192 + tl]) / (float)(64 * MW);
    __syncthreads();
    float v2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float d = v[i] - mean;
      v2 += d * d;
    }
    red[rg * 64 + tl] = v2;
    __syncthreads();
    const float var = (red[tl] + red[64 + tl] + red[128 + tl] + red[192 + tl]) / (float)(64 * MW);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (qin) {
      float* yb = a.out + (size_t)b * a.o_bs + t;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = rg + 4 * i;
        yb[(long)c * a.o_cs] = (v[i] - mean) * rstd * par[192 + c] + par[384 + c];
      }
    }
  } else {
    // conv_epi_store's steps in its order (conv_mfma.hip), one (row, column) per thread and pass: a wave covers 64
    // consecutive columns of one row -- 256-byte segments for every load and store.  A block's rows lie on ONE side of
    // split_row (a multiple of 192 or 0: the launcher checks), so the destination is uniform.
    const bool second = a.split_row && row0 >= a.split_row;
    float* const dst = second ? a.out2 + (size_t)b * a.o2_bs + (long)(row0 - a.split_row) * a.o2_cs
                              : a.out + (size_t)b * a.o_bs + (long)row0 * a.o_cs;
    const long dcs = second ? a.o2_cs : a.o_cs;
    const float* const resb = (a.res && !second) ? a.res + (size_t)b * a.r_bs + (long)row0 * a.r_cs + tt : nullptr;
    const bool prev = second ? a.acc_prev2 != 0 : a.acc_prev != 0;
    // the epilogue's operands, requested before the contraction: one uniform branch per operand KIND around a loop of
    // independent loads (written per element, each value waited for its own bias load: 48 serial round trips per block).
    // rows is a multiple of 16 and a thread's rows are rg + 4 i: row i exists for i < rows / 4, uniformly over the block.
    const int nv = rows >> 2;
    float rv[NV], pv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) rv[i] = pv[i] = 0.f;
    if (resb) {
#pragma unroll
      for (int i = 0; i < NV; ++i) rv[i] = resb[(long)(rg + 4 * (i < nv ? i : 0)) * a.r_cs];
    }
    if (prev) {
#pragma unroll
      for (int i = 0; i < NV; ++i) pv[i] = dst[(long)(rg + 4 * (i < nv ? i : 0)) * dcs + tt];
    }
    cc_contract<MW, CIN>(W, MTB, img, tile);
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = tile[(rg + 4 * (i < nv ? i : 0)) * CC_TS + tl] + par[rg + 4 * (i < nv ? i : 0)];
    if (second) {
      // second destination: out2[row - split_row] = conv + bias (+ out2), masked by mask_post2
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += pv[i];                       // (zeros without acc_prev2)
      if (a.mask_post2 && !valid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
      }
    } else {
      if (a.mask_pre && !valid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
      }
      if (a.alpha != 1.f) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] *= a.alpha;
      }
      if (resb) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += rv[i];
      }
      if (prev) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += pv[i];
      }
      if (a.mask_post && !valid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.f;
      }
    }
    if (qin) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) dst[(long)(rg + 4 * i) * dcs + t] = v[i];
    }
  }
}

bool conv_cols_supported(const ConvArgs& a) {
  return a.wg && a.f16s && a.K == 1 && a.pad == 0 && a.dil == 1 && (a.Cin == 192 || a.Cin == 96) && (a.M & 15) == 0 &&
         a.M >= 64 && a.M <= 576 && !a.cond && a.act == 0 && !a.in_act && a.div == 1.f && a.ups_s == 0 &&
         a.Nq == a.T_in && a.Nq > 0 && (a.split_row == 0 || (a.split_row % CC_ROWS == 0 && a.split_row < a.M && a.out2));
}

template <int MW, int CIN, bool LN>
static hipError_t launch_cols(const ConvArgs& a, int B, hipStream_t s, const float* gamma, const float* beta) {
  const int lds = cc_lds_bytes(CIN, 64 * MW);
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(conv_cols_kernel<MW, CIN, LN>), lds, attr_done); e != hipSuccess)
    return e;
  hipLaunchKernelGGL((conv_cols_kernel<MW, CIN, LN>), dim3((a.Nq + CC_BT - 1) / CC_BT, (a.M + CC_ROWS - 1) / CC_ROWS, B), dim3(256), lds,
                     s, a, gamma, beta);
  return hipGetLastError();
}

hipError_t launch_conv_cols(const ConvArgs& a, int B, hipStream_t s, const float* ln_gamma, const float* ln_beta) {
  if (!conv_cols_supported(a) || B <= 0) return hipErrorInvalidValue;
  const bool ln = ln_gamma != nullptr;
  if (ln) {
    // the LayerNorm form: every row of the tensor in the block, no second destination, no masks (attentions.py:41-42)
    if (!ln_beta || a.split_row || a.M != 192 || a.Cin != 192 || a.mask_pre || a.mask_post || a.acc_prev || a.alpha != 1.f)
      return hipErrorInvalidValue;
    return launch_cols<3, 192, true>(a, B, s, ln_gamma, ln_beta);
  }
  // (rows per block: min(M, 192); two m-tiles per wave cover up to 128 of them)
  const bool two = a.M <= 128;
  if (a.Cin == 192) return two ? launch_cols<2, 192, false>(a, B, s, nullptr, nullptr) : launch_cols<3, 192, false>(a, B, s, nullptr, nullptr);
  return two ? launch_cols<2, 96, false>(a, B, s, nullptr, nullptr) : launch_cols<3, 96, false>(a, B, s, nullptr, nullptr);
}

}  // namespace vsp
