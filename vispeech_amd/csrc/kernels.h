// Internal kernel launch interface of libvispeech_hip (gfx950).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/vispeech_hip.h"   // (VSP_FLAG_*: the sticky status bits the kernels raise)

namespace vsp {

// hipFuncAttributeMaxDynamicSharedMemorySize is set per DEVICE: a process may hold contexts on several GPUs (vsp_create
// takes a device index), so a launcher remembers it per (kernel, device) -- `done` is the launcher's static bit set, one
// bit per device; safe to race (the attribute call is idempotent).
inline hipError_t set_max_dynamic_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// ------------------------------------------------------------------------------------------
// conv1d as implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain).
//   out[b][row][q] = epilogue( sum_{ci,tap} Wp[row][ci][tap] * prologue(x[b][ci][q - pad + tap*dil]) )
// Weights are pre-packed in MFMA A-fragment order by pack_conv_weights() (host).
constexpr int CONV_CK = 32;    // input channels staged per LDS chunk
constexpr int CONV_HALO = 64;  // max (K-1)*dil

struct ConvArgs {
  const float* x; long x_bs, x_cs;
  const float* wp;            // packed weights
  const float* bias;          // [M] per packed row, or null
  float* out; long o_bs, o_cs;
  const float* res; long r_bs, r_cs;
  const float* cond; long cond_bs;   // cond[b*cond_bs + row] added after bias, or null
  const int64_t* lengths;     // [B] valid length (mask = t < lengths[b]), or null
  int Cin, M, K, dil, pad;
  int T_in;                   // input extent (zero padding outside [0,T_in))
  int Nq;                     // output columns computed
  int nchunks;                // ceil(Cin / CONV_CK)
  // prologue on the staged input
  int in_mask;                // x * mask
  int in_act; float in_slope; // leaky_relu(x, slope) (slope 0 = relu)
  // epilogue
  int act;                    // 0 none, 1 relu, 2 gate: tanh(tile 2i) * sigmoid(tile 2i+1)
  int mask_pre;
  float alpha;                // v *= alpha
  int acc_prev;               // v += out (read-modify-write)
  float div;                  // v /= div
  int mask_post;
  int ups_s, ups_p, T_store;  // transposed-conv store: row=(co,r), n = s*q + r - p in [0,T_store)
  int f16s;                   // weights packed by pack_conv_weights_f16s: split-f16 MFMA path
  // two destinations (WN res_skip layer, reference modules.py:165-172, as ONE launch): rows >= split_row (a multiple of
  // 32; 0 = off) go to out2[row - split_row] (+ out2 when acc_prev2) with no residual, mask or scaling; the rows
  // below it keep the epilogue above
  int split_row; float* out2; long o2_bs, o2_cs; int acc_prev2;
  int mask_post2;             // ... except this mask on the second destination (the two halves of one projection)
  // round 6: the same weights in 16x16x32 A-fragment order (pack_g16_weights, K = 1) for the column-tile form
  // (conv_cols.hip) of small-grid 1x1 convolutions; NULL = this convolution has no such image
  const uint16_t* wg;
  long wg_min_blocks, wg_max_blocks;   // launch_conv takes the column-tile kernel for this range of 64-column tiles (B * ceil(Nq / 64))
};

// the tile shape is chosen from M and Nq
hipError_t launch_conv(const ConvArgs& a, int B, hipStream_t s);
// Column-tile form of a 1x1 convolution (conv_cols.hip): every output row of a 64-column tile in ONE block.
// ln_gamma / ln_beta != NULL: out = LayerNorm_channels(conv + bias + res) * gamma + beta (M = Cin = 192, no masks).
bool conv_cols_supported(const ConvArgs& a);
hipError_t launch_conv_cols(const ConvArgs& a, int B, hipStream_t s, const float* ln_gamma = nullptr,
                            const float* ln_beta = nullptr);
size_t packed_conv_floats(int M, int Cin, int K);
// W(row, ci, tap) accessor -> packed buffer (host).  dst has packed_conv_floats(M,Cin,K) floats.
void pack_conv_weights(float* dst, int M, int Cin, int K, const float* dense /* [M][Cin][K] */);
void pack_conv_weights_f16s(float* dst, int M, int Cin, int K, const float* dense /* [M][Cin][K] */);

// The channels-last split-f16 kernels take their weights AND biases * G16_WSCALE and unscale every result by G16_UNSCALE
// (powers of two: exact; g16_common.h "ONE accumulator per tile"): pack_g16_weights scales the weights itself, whoever
// fills a bias array for these kernels (weights.cpp, upload_cl_conv in api.hip) scales the bias.
#ifndef G16_WSCALE_V          // (timing experiments only: -DG16_WSCALE_V=1.f -DG16_UNSCALE_V=1.f is the unscaled form)
#define G16_WSCALE_V 256.f
#define G16_UNSCALE_V (1.f / 256.f)
#endif
constexpr float G16_WSCALE = G16_WSCALE_V, G16_UNSCALE = G16_UNSCALE_V;

#ifdef __HIPCC__
// THE operand split of two fp32 values into f16 hi / lo pairs (hi | lo packed two per register), shared by every split-f16
// kernel (g16_common.h g16_split2; conv_mfma.hip) -- one function, so that the implementations stay bit-identical.
// Round 6 form, FOUR vector instructions per two values (rounds 3-5: six):
//   hi  = v_cvt_pk_f16_f32(x0, x1): both values ROUNDED to f16 (nearest even) in one instruction; +-inf beyond the f16
//         range -- an activation the split cannot represent poisons the product (inf / NaN down to the waveform, where
//         conv_post raises the context's sticky flag: vsp_status) where rounds 2-5's v_cvt_pkrtz saturated silently;
//   lo  = x - hi: v_fma_mix_f32 reads the f16 half directly (hi * -1.0 + x, ONE rounding of an exactly representable
//         difference: exact), so no v_and_b32 truncation and no conversion back -- one instruction per value;
//   lo pair packed by v_cvt_pkrtz_f16_f32 (|lo| <= 2^-11 |x|: 13 significant bits truncated to f16's 11, i.e. x to
//         2^-22 relative; an f16 subnormal -- multiples of 2^-24 -- where |x| < 2^-3).
// Rounds 3-5 cut the mantissa (v_and_b32 0xffffe000) for hi and subtracted in fp32: same precision class (hi truncated:
// |lo| <= 2^-10 |x|), two instructions more.  VSP_SPLIT_FORM=0 builds that form (timing A/B only; it saturates).
// (The operands come from compiler-generated vector instructions, never straight from an MFMA result: see the hazard
// note in g16_common.h.)
// A kernel reports a numeric-range event in the context's status word (vsp_status): pinned host memory mapped into the
// device's address space, so the OR is a SYSTEM-scope atomic (the host reads / clears the same word without synchronising)
__device__ __forceinline__ void vsp_raise_flag(unsigned* flags, unsigned bit) {
  __hip_atomic_fetch_or(flags, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#ifndef VSP_SPLIT_FORM
#define VSP_SPLIT_FORM 1
#endif
__device__ __forceinline__ void vsp_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
#if VSP_SPLIT_FORM == 0
  const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffffe000u);
  float l0, l1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(l0) : "v"(x0), "v"(h0));     // (plain f32 instructions: hipcc would SLP-pack them)
  asm("v_sub_f32 %0, %1, %2" : "=v"(l1) : "v"(x1), "v"(h1));
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
#else
  float l0, l1;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
#endif
}
#endif

// ------------------------------------------------------------------------------------------
// channels-last split-f16 vocoder conv (gen16.hip): x [B][T][Cin], out [B][T][Cout]
struct ClConvArgs {
  const float* x; long x_bs; int x_ts;
  const uint16_t* wh;                       // packed f16 A-fragment image, hi | lo interleaved (pack_g16_weights)
  const float* bias;                        // [Cout] or null
  float* out; long o_bs; int o_ts;
  const float* res; long r_bs; int r_ts;
  int Cin, Cout, K, dil, pad;
  int T_in, Nq;
  int in_act; float in_slope;
  int acc_prev; float div;
  int phases, ups_p, T_store;               // polyphase transposed conv: row n = phases*q + ph - ups_p
  int terms;                                // 3 = fp32-accurate split product (default), 1 = plain f16 operands
  // Operand images (round 4; terms == 3, phases == 1): layout [chunk32][hi | lo][plane][cl_img_tpad(T)][8 halfs] per
  // utterance, CL_IMG_PADF zero rows in front of time 0 and CL_IMG_PADB behind time T - 1 (launch_cl_img_zero_pads).
  //   x_img: read the input from a producer-written image instead of x (K >= 3; in_act / in_slope are then the
  //          PRODUCER's business); o_img: also (out != NULL) or only (out == NULL) write leaky_relu(result, oi_slope)
  //          as the next convolution's image.
  const uint16_t* x_img; long xi_bs; int xi_tpad;   // batch stride in halfs, padded rows per plane
  uint16_t* o_img; long oi_bs; int oi_tpad; float oi_slope;
  int pt_gx, pt_gy, pt_total;               // set by the persistent launcher: time tiles, row groups, tiles in all
  // Ragged batch (round 5, "trimmed tails"): glen[b] (device, int32) = frames of utterance b this launch computes; its
  // extents are then T_in = glen[b] * g_in, Nq = T_in + (Nq - T_in of the launch), T_store = glen[b] * g_store instead of
  // the launch-wide values above (which stay the MAXIMUM: grid size, strides, image padding).  Rows at and beyond an
  // utterance's own extent read as zero and are never stored: the tensor ENDS there for that utterance.  NULL = uniform.
  const int* glen; int g_in, g_store;
};
constexpr int CL_IMG_PADF = 64, CL_IMG_PADB = 320;
inline int cl_img_tpad(int T) { return T + CL_IMG_PADF + CL_IMG_PADB; }
inline size_t cl_img_halfs(int C, int T) { return (size_t)C * 2 * cl_img_tpad(T); }   // per utterance
// zero the pad rows of B utterances' images (the data rows are the producer's); glen != NULL: utterance b's data rows end
// at glen[b] * grate (<= T), the CL_IMG_PADB zero rows follow there
hipError_t launch_cl_img_zero_pads(uint16_t* img, int B, int C, int T, hipStream_t s, const int* glen = nullptr, int grate = 0);

// Fused ResBlock1 conv pair on channels-last activations (gen16.hip):
//   out = x + conv2(lrelu(conv1(lrelu(x), dil) + b1), 1) + b2  [+ out] [/ div];  x != out.
struct ClPairArgs {
  const float* x; long x_bs;                // [B][T][C], batch stride in elements
  const uint16_t* w1h; const float* b1;     // conv1 (dilation dil): packed f16 fragment image, bias
  const uint16_t* w2h; const float* b2;     // conv2 (dilation 1)
  float* out; long o_bs;
  int C, K, dil, T;
  float slope;                              // leaky-relu slope applied to both convs' inputs
  int acc_prev; float div;                  // out = (result + out) / div
  int terms;                                // 3 = split product, 1 = plain f16 operands
  int ring;                                 // 1: the LDS-ring pair kernel (g16_pair) even where the register-weights one exists
  int rw64;                                 // 1: the register-weights kernel for 64-channel kernel-3 pairs (g16_rw64; opt-in, VSP_RW64=1)
  int tiles;                                // set by the launcher: tiles per utterance
  int xrows;                                // set by the launcher: staged window rows
  const int* glen; int grate;               // ragged batch (see ClConvArgs): utterance b's T = glen[b] * grate; NULL = uniform
  int B;                                    // set by the launcher (persistent kernels walk the utterances themselves)
};
// Fused ResBlock1 CHAIN on channels-last activations (gen16.hip): np conv pairs back to back in one launch,
//   x_{p+1} = x_p + conv2_p(lrelu(conv1_p(lrelu(x_p), dil_p) + b1_p), 1) + b2_p,   out = x_np [+ out] [/ div];  x != out.
// The running x_p stays in registers (fp32), the convolution inputs in LDS; a block recomputes the chain's halo.
struct ClChainArgs {
  const float* x; long x_bs;
  float* out; long o_bs;
  const uint16_t* w[6]; const float* b[6];  // conv1, conv2 of pair 0, 1, 2
  int dil[3]; int np;
  int C, K, T;
  float slope;
  int acc_prev; float div;
  int terms;
  int ring;                                 // 1: the LDS-ring chain kernel (g16_chain) even where the register-weights one exists
  int tiles, halo;                          // set by the launcher: tiles per utterance, columns recomputed per side
  const int* glen; int grate;               // ragged batch (see ClConvArgs): utterance b's T = glen[b] * grate; NULL = uniform
  int B;                                    // set by the launcher
};
// the kernel-3 ResBlock of the 32-channel stage as a role pipeline with the weights in registers (gen16_rc.hip);
// launch_g16_chain routes there unless ClChainArgs::ring (VSP_CHAIN_RING=1) asks for the LDS-ring chain (bit-identical)
bool g16_rc_supported(int C, int K, const int* dil, int np, int terms, int acc_prev);
hipError_t launch_g16_rc(const ClChainArgs& a, int B, hipStream_t s);
bool g16_chain_supported(int C, int K, const int* dil, int np);
hipError_t launch_g16_chain(const ClChainArgs& a, int B, hipStream_t s);
hipError_t launch_g16_conv(const ClConvArgs& a, int B, hipStream_t s);
// image-input convolutions on the 128-row tile as persistent blocks pipelined across tiles (gen16_pipe.hip):
// launch_g16_conv routes tiles of at most 28 steps there (VSP_G16_PIPE=0: never, =1: always -- bit-identical)
bool g16_pipe_supported(const ClConvArgs& a);
hipError_t launch_g16_pipe(const ClConvArgs& a, int B, hipStream_t s);
bool g16_pair_supported(int C, int K, int dil);
hipError_t launch_g16_pair(const ClPairArgs& a, int B, hipStream_t s);
// the same pair with the weights held in registers by persistent blocks (gen16_rw.hip: 32 channels, kernel 7 / 11);
// launch_g16_pair routes there unless ClPairArgs::ring (VSP_PAIR=ring) asks for the LDS-ring kernel (second implementation, bit-identical)
bool g16_rw_supported(int C, int K, int dil, int terms);
hipError_t launch_g16_rw(const ClPairArgs& a, int B, hipStream_t s);
// ... and the kernel-3 pairs of the 64-channel stage (gen16_rw64.hip, round 5: two row halves x two column halves per role)
bool g16_rw64_supported(int C, int K, int dil, int terms);
hipError_t launch_g16_rw64(const ClPairArgs& a, int B, hipStream_t s);
// the pair of the 128-channel stage on the ping-pong tile, kernel 3 / 7 (gen16_pp.hip); VSP_PP=0 (read per context): two launches (bit-identical)
bool g16_pp_supported(int C, int K, int dil, int terms);
hipError_t launch_g16_pp(const ClPairArgs& a, int B, hipStream_t s);
size_t packed_g16_halfs(int rows, int Cin, int K);
void pack_g16_weights(uint16_t* dst, int rows, int Cin, int K, const float* dense /* [rows][Cin][K] */);
// mel[b][m][t] = log(max(sum_{f in [lo[m], hi[m])} basis[m][f] * spec[b][f][t], 1e-5))  (reference mel_processing.py:16-22, 73-82)
hipError_t launch_spec_to_mel(const float* spec, const float* basis, const int* lo, const int* hi, float* mel, int B,
                              int n_freq, int n_mels, int T, hipStream_t s);
// y = transpose(x) * scale (the generator's entry: the context's activation scale, model.h)
hipError_t launch_transpose_ct(const float* x, long x_bs, long x_cs, float* y, long y_bs, int y_ts, int B, int C,
                               int T, hipStream_t s, float scale = 1.f);
// o = tanh(unscale * conv(lrelu(x)));  flags != NULL: VSP_FLAG_NONFINITE_WAVE is raised there when a sum is inf / NaN
hipError_t launch_conv_post_cl(const float* x, long x_bs, int x_ts, const float* w, int C, int K, float slope,
                               float* o, long o_bs, int B, int T, hipStream_t s, const int* glen = nullptr, int grate = 0,
                               float unscale = 1.f, unsigned* flags = nullptr);
// Trimmed tails (round 5).  The generator's input behind an utterance's last frame is exactly zero (reference
// models.py:720: z * x_mask), so its output there is a bias-driven signal that depends on the distance to the utterance's
// end and to the tensor's end only: periodic in one frame once the receptive field away from both.
// Output frame F depends on input frames [F - back, F + fwd] (api.hip, generator_frame_dependence).
//   gen_plan:  glen[b] = len[b] + back + 1 + fwd where that is < T (else T): the frames the generator computes for b;
//   gen_tail_fill:  frames [len + back + 1, T - fwd) of o[b] = frame len + back (the steady state), frames
//                   [T - fwd, T) = frames [len + back + 1, len + back + 1 + fwd) (the computed tensor end); up = samples per frame.
hipError_t launch_gen_plan(const int64_t* lengths, int B, int T, int back, int fwd, int* glen, hipStream_t s);
hipError_t launch_gen_tail_fill(float* o, long o_bs, const int64_t* lengths, const int* glen, int B, int T, int back, int fwd,
                                int up, hipStream_t s);

// ------------------------------------------------------------------------------------------
// attention with windowed relative position (reference attentions.py:148-179), f32 MFMA.
// qkv [B][3*H][T]: rows [0,H) = q, [H,2H) = k, [2H,3H) = v; out [B][H][T].
hipError_t launch_attention(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k,
                            const float* emb_v, const int64_t* lengths, float* out, long o_bs,
                            long o_cs, int B, int H, int n_heads, int T, int window, int ksplit_mode, hipStream_t s);

// the same on the f16 matrix core with split operands, one pass (attention_f16s.hip); workspace = 3 *
// attn_pack_bytes(B, n_heads, H / n_heads, T) bytes (the packed q | k | v operand images)
size_t attn_pack_bytes(int B, int n_heads, int DK, int T);
hipError_t launch_attention_f16s(const float* qkv, long qkv_bs, long qkv_cs, const float* emb_k, const float* emb_v,
                                 const int64_t* lengths, float* out, long o_bs, long o_cs, int B, int H, int n_heads, int T,
                                 int window, void* workspace, hipStream_t s);
// conv_q | conv_k | conv_v of x * x_mask AND the packing of their results in one launch: fills `workspace` with the images
// launch_attention_f16s(qkv = NULL, ...) then consumes.  wg: pack_g16_weights of the stacked [3H][H] projection.
bool attn_qkv_pack_supported(int H, int n_heads);
hipError_t launch_attn_qkv_pack_f16s(const float* x, long x_bs, long x_cs, const uint16_t* wg, const float* bias,
                                     const int64_t* lengths, int B, int H, int n_heads, int T, void* workspace, hipStream_t s);

// ------------------------------------------------------------------------------------------
// small kernels (misc.hip)
// y = LN_channels(x (+ res)) * gamma + beta  (reference modules.py:29-32), eps 1e-5
hipError_t launch_layernorm(const float* x, long x_bs, long x_cs, const float* res, long r_bs, long r_cs,
                            const float* gamma, const float* beta, float* y, long y_bs, long y_cs,
                            int B, int C, int T, hipStream_t s);
// x[b][c][t] = emb[ids[b][t]][c] * scale
hipError_t launch_embed(const int64_t* ids, const float* emb, int n_vocab, float scale, float* x,
                        long x_bs, long x_cs, int B, int C, int T, hipStream_t s);
// g[b][c] = table[sid[b]][c]
hipError_t launch_gather_rows(const int64_t* idx, const float* table, int n_rows, float* out, int B, int C,
                              hipStream_t s);
// y[b][c][t] = x[b][c][t] + c[b][c]  (optionally * mask)
hipError_t launch_add_cond(const float* x, long x_bs, long x_cs, const float* cond, long cond_bs, float* y,
                           long y_bs, long y_cs, int B, int C, int T, hipStream_t s);
// out[b][t] = bias + sum_c w[c] * x[b][c][t] (* mask_in on x, * mask_out on result)
hipError_t launch_chan_dot(const float* x, long x_bs, long x_cs, const float* w, const float* bias,
                           const int64_t* lengths, int mask_in, int mask_out, float* out, int B, int C, int T,
                           hipStream_t s);
// x[b][c][t] += bias[c] + sum_j w[c][j] * s[b][t + j - 1]   (Conv1d(1,C,3,padding=1), unmasked)
hipError_t launch_prenet_add(float* x, long x_bs, long x_cs, const float* w, const float* bias,
                             const float* sig, int B, int C, int T, hipStream_t s);
// duration / F0 / energy formulas (reference models.py:681-708)
hipError_t launch_duration_from_logw(const float* logw, const int64_t* lengths, float scale, float* dur,
                                     int B, int T, hipStream_t s);
hipError_t launch_pitch(const float* pitch_ctl, const float* lf0_pred, float scale, float* lf0, float* f0,
                        int n, hipStream_t s);
hipError_t launch_energy(const float* energy_ctl, const float* e_pred, float scale, float* norm_e,
                         float* energy, int n, hipStream_t s);
// cum[b][i] = sum_{k<=i} max(int(d[b][k]),0); frame_lengths[b] = cum[b][Tp-1]
hipError_t launch_duration_cumsum(const float* dur, int32_t* cum, int64_t* frame_lengths, int B, int Tp,
                                  hipStream_t s, unsigned* flags = nullptr);
// out[b][c][f] = f < cum[b][Tp-1] ? x[b][c][upper_bound(cum[b], f)] : 0
hipError_t launch_length_regulate(const float* x, long x_bs, long x_cs, const int32_t* cum, float* out,
                                  long o_bs, long o_cs, int B, int C, int Tp, int Tf, hipStream_t s);
// z_p = m_p + noise * exp(logs_p) * noise_scale ; x_mask[b][t] = t < len[b]
hipError_t launch_reparam(const float* m_p, const float* logs_p, const float* noise, float noise_scale,
                          float* z_p, long n, hipStream_t s, float* copy = nullptr, unsigned* flags = nullptr);
// out[i] = standard normal draw first + i of the Philox4x32-10 stream keyed by `seed` (misc.hip)
hipError_t launch_randn(uint64_t seed, long first, long n, float* out, hipStream_t s);
hipError_t launch_mask_u8(const int64_t* lengths, uint8_t* mask, int B, int T, hipStream_t s);
// o[b][t] = tanh( sum_c sum_j w[c][j] * lrelu(x[b][c][t+j-pad], slope) )  (conv_post, no bias)
hipError_t launch_conv_post(const float* x, long x_bs, long x_cs, const float* w, int C, int K, float slope,
                            float* o, long o_bs, int B, int T, hipStream_t s, unsigned* flags = nullptr);
hipError_t launch_copy3(const float* x, long x_bs, long x_cs, float* y, long y_bs, long y_cs, int B, int C,
                        int T, hipStream_t s);
// x[b][c][t] = 0 for t >= lengths[b]
hipError_t launch_mask3(float* x, long x_bs, long x_cs, const int64_t* lengths, int B, int C, int T, hipStream_t s);
hipError_t launch_rq_spline(int64_t n, int nb, const float* x, const float* uw, const float* uh,
                            const float* ud, int inverse, float tail_bound, float* y, float* lad,
                            hipStream_t s);

hipError_t launch_stft_frames(const float* audio, long a_bs, float* f, long f_bs, long f_cs, int B, int L, int n_fft,
                              int hop, int T, hipStream_t s);
hipError_t launch_stft_magnitude(const float* ri, long r_bs, long r_cs, float* spec, int B, int spec_ch, int T,
                                 hipStream_t s);

}  // namespace vsp
