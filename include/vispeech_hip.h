/*
 * vispeech_hip.h -- C-ABI of the MI355X (gfx950) synthesis path.
 *
 * The reference (innnky/vispeech) has NO native/FFI layer: the only boundary its hot path sits
 * behind is the Python method SynthesizerTrn.infer (reference models.py:672-722) and the
 * checkpoint key schema (reference utils.py:21-51, 67-70).  This header is the C boundary a
 * maintainer binds that method to (ctypes stub in INTEGRATION.md; vispeech_amd/models.py is the
 * shipped binding).  Conventions:
 *   - every entry point returns int: 0 = ok, negative = error class (VSP_ERR_*); nothing throws;
 *     vsp_last_error() gives the message of the last failure on that context.
 *   - plain pointers and sizes only.  "dev" pointers are HIP device pointers owned by the CALLER
 *     (torch tensors' data_ptr()); the library owns only its packed weight arena (unless the
 *     caller supplies one) -- no hidden allocation happens on the infer path.
 *   - all work is enqueued on the caller's stream (void* = hipStream_t); the only host
 *     synchronisation is in vsp_frame_lengths_host (the read of the frame counts, which replaces
 *     the B*T_p .item() syncs of reference models.py:398-427).
 *   - activations are float32, laid out [B][C][T] with T contiguous, exactly like the reference's
 *     tensors; a tensor argument is (ptr, batch_stride, channel_stride) in ELEMENTS.
 *   - one context per device; a context is not thread-safe (the reference serialises infer calls
 *     with a lock, inference_api.py:13,37); different contexts are independent.
 */
#ifndef VISPEECH_HIP_H
#define VISPEECH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 5): no entry point changed, but (a) the PACKED WEIGHT ARENA holds the split-f16 convolutions' weights and biases
 * scaled by 2^8 with unscaled lo parts (one fp32 accumulator per tile) -- an arena packed by an ABI-5 library must not be
 * adopted (vsp_commit_adopted_weights checks this number in the arena header) -- and (b) vsp_frame_lengths_host no longer
 * implies a full stream synchronisation after a vsp_encode with given durations (see there). */
/* 7 (round 6): (a) vsp_status -- sticky numeric-range flags raised by the kernels (see there); (b) the generator's
 * activations are carried * 2^VSP_ACT_SCALE_LOG2 between conv_pre and conv_post and the packed generator biases carry that
 * factor: an arena packed by an ABI-6 library (or under another VSP_ACT_SCALE_LOG2: the arena's configuration hash covers it)
 * must not be adopted; (c) vsp_profile_read_class / vsp_profile_read_families return one more figure per class / family, the
 * bytes a launch moves AS FUSED (a new trailing out-parameter each: callers of ABI <= 6 must be rebuilt); (d) vsp_conv1d
 * accepts split_f16 = 2 (the column-tile kernel as a stand-alone operator). */
#define VSP_ABI_VERSION 7

enum {
  VSP_OK = 0,
  VSP_ERR_ARG = -1,        /* null / out-of-range argument */
  VSP_ERR_STATE = -2,      /* call order (weights not finalised, ...) */
  VSP_ERR_HIP = -3,        /* a HIP runtime call failed */
  VSP_ERR_KEY = -4,        /* unknown state_dict key */
  VSP_ERR_SHAPE = -5,      /* tensor shape differs from the schema */
  VSP_ERR_WORKSPACE = -6,  /* workspace too small */
  VSP_ERR_UNSUPPORTED = -7 /* configuration outside what the kernels cover */
};

#define VSP_MAX_LIST 8

/* Hyper-parameters that fix tensor shapes: the constructor arguments of the reference's
 * SynthesizerTrn (models.py:537-561) plus the constants its code hard-wires
 * (attentions.py:14 window 4; models.py:498 6 pitch layers; :599 duration filter 256;
 * frame_prior_network.py:65 energy filter 768; models.py:597 WN kernel 5 / 4 layers; :184 4 flows). */
typedef struct vsp_config {
  int32_t n_vocab;
  int32_t inter_channels;
  int32_t hidden_channels;
  int32_t filter_channels;
  int32_t n_heads;
  int32_t n_layers;
  int32_t kernel_size;
  int32_t n_resblock_kernels;
  int32_t resblock_kernel_sizes[VSP_MAX_LIST];
  int32_t n_resblock_dilations;                       /* dilations per ResBlock1 (3) */
  int32_t resblock_dilation_sizes[VSP_MAX_LIST][VSP_MAX_LIST];
  int32_t n_upsamples;
  int32_t upsample_rates[VSP_MAX_LIST];
  int32_t upsample_kernel_sizes[VSP_MAX_LIST];
  int32_t upsample_initial_channel;
  int32_t n_speakers;
  int32_t gin_channels;
  int32_t window_size;
  int32_t pitch_layers;
  int32_t dur_filter;
  int32_t energy_filter;
  int32_t flow_kernel;
  int32_t flow_layers;
  int32_t n_flows;
  /* voice conversion only (ABI 2): input channels of the posterior encoder = filter_length/2+1
   * (reference models.py:596) and its WN depth (16, models.py:596).  spec_channels == 0 builds a
   * context without the posterior encoder (enc_q.* tensors are then accepted and ignored). */
  int32_t spec_channels;
  int32_t posterior_layers;
} vsp_config;

typedef struct vsp_ctx vsp_ctx;

/* ---- lifetime --------------------------------------------------------------------------- */
int vsp_abi_version(void);
/* Replaces SynthesizerTrn.__init__ (reference models.py:537-622) for the infer path. */
int vsp_create(const vsp_config* cfg, int device, vsp_ctx** out);
int vsp_destroy(vsp_ctx* ctx);
const char* vsp_last_error(const vsp_ctx* ctx);

/* Sticky numeric-range flags (ABI 7).  The matrix kernels multiply fp32 activations as f16 hi / lo pairs: an activation
 * beyond the f16 range (|x| > 65504; inside the generator |x| > 65504 / 2^VSP_ACT_SCALE_LOG2 = 4094 by default) cannot be
 * split.  It becomes +-inf in the operand (round 5 clamped it silently), so the affected outputs are inf / NaN rather than
 * plausible wrong numbers, and the last kernels of each half raise a flag in a host-visible word of the context:
 *   VSP_FLAG_NONFINITE_LATENT  z_p = m_p + noise * exp(logs_p) * noise_scale is not finite somewhere (phoneme- / frame-rate half)
 *   VSP_FLAG_NONFINITE_WAVE    a waveform sample's pre-tanh sum is not finite (flow or generator)
 * vsp_status copies the word to *flags (no stream synchronisation: it reflects the launches that have COMPLETED; synchronise
 * the stream first for a definitive answer) and clears it when `clear` != 0.  Weights outside the packed range are refused at
 * load time (vsp_finalize_weights); this is the same loudness for activations.  The reference's fp32 path has no such limit.
 * Precision floor of the split, in terms of OUTPUT level: with the default activation scale the waveform stays within
 * 1e-4 * max|o| of the fp32 reference down to a peak level of about -90 dBFS (tests/test_amplitude_floor.py measures it;
 * VSP_ACT_SCALE_LOG2=0, round 5's behaviour, crosses that bound near -68 dBFS). */
#define VSP_FLAG_NONFINITE_LATENT 1u
#define VSP_FLAG_NONFINITE_WAVE 2u
int vsp_status(vsp_ctx* ctx, unsigned* flags, int clear);

/* ---- weights: replaces load_state_dict / utils.load_checkpoint (reference utils.py:21-51) - */
/* One call per state_dict tensor, raw reference keys ("dec.ups.0.weight_v", ...), float32 HOST
 * data (copied).  weight_g/weight_v pairs are folded by vsp_finalize_weights exactly as
 * torch.nn.utils.weight_norm does (reference modules.py:128,135,145,191-206; models.py:255: for
 * ConvTranspose1d the norm is per INPUT channel).  Keys that infer never reads (enc_q.*,
 * enc_p.proj.*, frame_prior_net.emb.*, energy_predictor.predictor.proj.*) are accepted and
 * ignored; unknown keys return VSP_ERR_KEY. */
int vsp_set_weight(vsp_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim);
/* The same for a checkpoint held in another type or on the device (ABI 3; SURVEY 8b's contract): dtype is one of
 * VSP_DTYPE_*, on_device != 0 means `data` is a HIP device pointer (copied to the host first).  Values are
 * widened to float32 exactly (f16 / bf16) or rounded once (f64). */
#define VSP_DTYPE_F32 0
#define VSP_DTYPE_F16 1
#define VSP_DTYPE_BF16 2
#define VSP_DTYPE_F64 3
int vsp_set_weight_typed(vsp_ctx* ctx, const char* key, const void* data, const int64_t* shape, int ndim, int dtype,
                         int on_device);
/* Forget every tensor set so far (start of a new load on a context that was loaded before).  Within one load,
 * setting "<x>.weight" drops an earlier "<x>.weight_g" / "<x>.weight_v" pair and vice versa. */
int vsp_begin_weights(vsp_ctx* ctx);
/* Number of infer-path tensors still missing (0 = ready to finalise). */
int vsp_missing_weights(const vsp_ctx* ctx);
/* Size of the packed device arena (depends on the config only). */
int64_t vsp_weight_arena_bytes(const vsp_ctx* ctx);
/* Fold + pack (MFMA fragment order) + upload.  dev_arena may be NULL (the library allocates).  VSP_ERR_STATE while a
 * required tensor is missing; VSP_ERR_UNSUPPORTED when a folded convolution weight does not fit the packed split-f16 form
 * (|w| < 253: the images hold w * 2^8 as f16 pairs -- ABI 6). */
int vsp_finalize_weights(vsp_ctx* ctx, void* dev_arena);
/* Multi-GPU: a non-root rank adopts an arena that receives rank 0's packed bytes by an RCCL broadcast; no host
 * weights needed.  vsp_adopt_packed_weights only records the pointer (the bytes may still be in flight): the context
 * is NOT ready until vsp_commit_adopted_weights, called after the broadcast has completed, has read the arena's
 * header (one small device-to-host copy + stream sync) and checked magic, ABI version, size and configuration hash.
 * What rank 0 packed travels in that header -- in particular whether the posterior encoder (voice conversion) is
 * there -- instead of being inferred from the local config. */
int vsp_adopt_packed_weights(vsp_ctx* ctx, void* dev_arena);
int vsp_commit_adopted_weights(vsp_ctx* ctx, void* stream);
/* The arena this context reads its weights from (for the broadcast on rank 0). */
int vsp_weight_arena(const vsp_ctx* ctx, void** dev_arena, int64_t* bytes);

/* ---- the path: replaces SynthesizerTrn.infer (reference models.py:672-722) ---------------- */
/* Phoneme-rate half, models.py:674-708 + the prefix sum of LengthRegulator (models.py:398-427):
 *   speaker embedding, TextEncoder, duration / F0 / energy (predicted, or the *_ctl tensor when
 *   non-NULL -- the reference's isinstance(.., torch.Tensor) branches), pitch/energy prenets.
 * Inputs (device): phonemes[B*Tp] int64, lengths[B] int64, sid[B] int64; *_ctl [B*Tp] float or NULL
 *   with the scalar controls used instead (reference defaults 1.0).
 * Outputs (device): x_var [B][H][Tp] (text encoding + prenets, the length regulator's input),
 *   g [B][gin], duration/f0/energy [B*Tp], frame_lengths[B] int64, cum_dur [B*Tp] int32.
 * workspace: >= vsp_encode_workspace_bytes(ctx,B,Tp). */
int64_t vsp_encode_workspace_bytes(const vsp_ctx* ctx, int B, int Tp);
int vsp_encode(vsp_ctx* ctx, void* stream, int B, int Tp,
               const int64_t* phonemes, const int64_t* lengths, const int64_t* sid,
               const float* duration_ctl, const float* pitch_ctl, const float* energy_ctl,
               float duration_scale, float pitch_scale, float energy_scale,
               float* x_var, float* g, float* duration, float* f0, float* energy,
               int64_t* frame_lengths, int32_t* cum_dur,
               void* workspace, int64_t workspace_bytes);
/* Copies frame_lengths[B] to the host and returns max(frame_lengths): the one host wait of an infer call.  After a
 * vsp_encode with duration_ctl != NULL (given durations: the counts depend on nothing vsp_encode computes) it waits only
 * for the copy vsp_encode started before its first launch -- the text encoder may still be running on the stream when it
 * returns, and vsp_decode enqueued behind it finds no idle GPU between the two halves; otherwise (predicted durations,
 * or a frame_lengths_dev that is not the last vsp_encode's) it is a copy + one stream synchronisation.  The pinned
 * B x int64 host buffer and the event behind this are created by the first such vsp_encode and kept with the context. */
int vsp_frame_lengths_host(vsp_ctx* ctx, void* stream, int B, const int64_t* frame_lengths_dev,
                           int64_t* frame_lengths_host, int64_t* max_frames);

/* Frame-rate half, models.py:711-720: length-regulator expand, FramePriorNet, Projection +
 * reparameterisation with the caller's noise (models.py:718), inverse flow, HiFi-GAN generator.
 * Tf = padded frame count of the batch (>= every frame length; the GLOBAL maximum in a sharded
 * run, SURVEY gotcha G6).  max_len < 0 = no truncation; the generator consumes
 * Tdec = min(Tf, max_len) frames and writes o[B][1][Tdec * prod(upsample_rates)].
 * noise [B][inter][Tf], or NULL: then the library draws it on the device -- vsp_randn(noise_seed) over the same
 * [B][inter][Tf] elements (ABI 4; the torch.randn_like(m_p) of models.py:718 for callers without a generator.  It is a
 * Philox4x32-10 stream of its own, NOT bit-compatible with torch.manual_seed(noise_seed); pass the tensor to
 * reproduce a torch run).  noise_seed is ignored when noise != NULL or noise_scale == 0.
 * Outputs (device, contiguous): o, x_mask[B*Tf] uint8, z, z_p, m_p, logs_p [B][inter][Tf]. */
int64_t vsp_decode_workspace_bytes(const vsp_ctx* ctx, int B, int Tp, int Tf);
int vsp_decode(vsp_ctx* ctx, void* stream, int B, int Tp, int Tf, int max_len,
               const float* x_var, const float* g, const int32_t* cum_dur, const int64_t* frame_lengths,
               const float* noise, uint64_t noise_seed, float noise_scale,
               float* o, uint8_t* x_mask, float* z, float* z_p, float* m_p, float* logs_p,
               void* workspace, int64_t workspace_bytes);

/* One-call form of the path for callers that know an upper bound of the frame count up front (supplied
 * durations, or a fixed max_len): vsp_encode + vsp_decode with Tf = tf_pad and NO host synchronisation.
 * tf_pad must be >= every utterance's frame count (frames past it are cut off; frame_lengths is returned
 * so the caller can check afterwards).  Arguments as in vsp_encode / vsp_decode; workspace >=
 * vsp_infer_workspace_bytes(ctx, B, Tp, tf_pad). */
int64_t vsp_infer_workspace_bytes(const vsp_ctx* ctx, int B, int Tp, int tf_pad);
int vsp_infer(vsp_ctx* ctx, void* stream, int B, int Tp, int tf_pad, int max_len,
              const int64_t* phonemes, const int64_t* lengths, const int64_t* sid,
              const float* duration_ctl, const float* pitch_ctl, const float* energy_ctl,
              float duration_scale, float pitch_scale, float energy_scale,
              const float* noise, uint64_t noise_seed, float noise_scale,
              float* o, uint8_t* x_mask, float* z, float* z_p, float* m_p, float* logs_p,
              float* duration, float* f0, float* energy, int64_t* frame_lengths,
              void* workspace, int64_t workspace_bytes);

/* ---- per-stage entry points (unit parity against the oracle) ------------------------------ */
/* MultiHeadAttention.attention (reference attentions.py:148-179 with the relative-position helpers
 * :181-243) of layer `layer` of encoder `which` (0 enc_p.encoder, 1 pitch_predictor.pitch_net,
 * 2 frame_prior_net.fft_block): qkv [B][3H][T] = conv_q | conv_k | conv_v outputs, lengths[B] (attn_mask =
 * mask x mask), out [B][H][T] = the tensor conv_o consumes.  workspace >= vsp_attention_workspace_bytes (the packed
 * f16 operand images; ABI 4: caller-owned like every other workspace -- no allocation behind the ABI).
 * Operand range of the default (split-f16) kernel: q, k, v are packed as f16 hi / lo pairs with fixed power-of-two
 * scales (q * 128 log2(e) / sqrt(d_k), k * 16, v * 16); magnitudes beyond |q| ~ 3.4e3, |k|, |v| ~ 4.0e3 are CLAMPED
 * to the largest finite f16 (the fp32 reference stays exact there; layer-normed encoder activations are O(1..10)).
 * VSP_ATT=f32 selects the f32 kernel, which has no such limit. */
int64_t vsp_attention_workspace_bytes(const vsp_ctx* ctx, int B, int T);
int vsp_attention(vsp_ctx* ctx, void* stream, int which, int layer, int B, int T, const float* qkv,
                  const int64_t* lengths, float* out, void* workspace, int64_t workspace_bytes);
/* attentions.Encoder.forward (reference attentions.py:35-47). which: 0 = enc_p.encoder,
 * 1 = pitch_predictor.pitch_net, 2 = frame_prior_net.fft_block.  x [B][H][T] in, y [B][H][T] out. */
int64_t vsp_encoder_workspace_bytes(const vsp_ctx* ctx, int B, int T);
int vsp_encoder(vsp_ctx* ctx, void* stream, int which, int B, int T, const float* x,
                const int64_t* lengths, float* y, void* workspace, int64_t workspace_bytes);
/* LengthRegulator (reference models.py:398-427) from a ready prefix sum. */
int vsp_length_regulate(vsp_ctx* ctx, void* stream, int B, int C, int Tp, int Tf, const float* x,
                        const int32_t* cum_dur, float* x_frame);
/* ResidualCouplingBlock.forward(reverse=True) (reference models.py:202-209). z_p -> z, in place
 * semantics on a copy: z is written, z_p is read. */
int64_t vsp_flow_workspace_bytes(const vsp_ctx* ctx, int B, int Tf);
int vsp_flow_reverse(vsp_ctx* ctx, void* stream, int B, int Tf, const float* z_p, const float* g,
                     const int64_t* frame_lengths, float* z, void* workspace, int64_t workspace_bytes);
/* One layer of modules.WN.forward (reference modules.py:148-176, dilation_rate 1; the gate is
 * commons.fused_add_tanh_sigmoid_multiply, commons.py:100-107):
 *   a = in_layers[layer](x) + cond_layer(g)[layer * 2h : (layer + 1) * 2h];  acts = tanh(a[:h]) * sigmoid(a[h:])
 *   rs = res_skip_layers[layer](acts)
 *   layer < n - 1:  x = (x + rs[:h]) * mask;  skip (+)= rs[h:]
 *   layer == n - 1: skip = (skip (+) rs) * mask        (the `output * x_mask` of modules.py:176 is folded in)
 * which: 0 .. n_flows-1 = flow.flows[2 * which].enc, -1 = enc_q.enc (needs the posterior-encoder weights).
 * x [B][h][T] is updated in place; skip [B][h][T] is accumulated into when accumulate != 0, else overwritten;
 * g [B][gin]; lengths [B]. */
int64_t vsp_wn_layer_workspace_bytes(const vsp_ctx* ctx, int which, int B, int T);
int vsp_wn_layer(vsp_ctx* ctx, void* stream, int which, int layer, int B, int T, float* x, const float* g,
                 const int64_t* lengths, float* skip, int accumulate, void* workspace, int64_t workspace_bytes);
/* Generator.forward (reference models.py:271-290). z [B][inter][T] (already masked/truncated). */
int64_t vsp_generator_workspace_bytes(const vsp_ctx* ctx, int B, int T);
int vsp_generator(vsp_ctx* ctx, void* stream, int B, int T, const float* z, const float* g,
                  float* o, void* workspace, int64_t workspace_bytes);

/* Streamed vocoder (BASELINE config 5; the chunked output loop of reference inference_api.py:50-60 applied to the
 * vocoder itself): the waveform samples of frames [f0, f1) of z [B][inter][T], computed from those frames plus
 * vsp_generator_halo_frames() frames on each side -- bit-identical to the same samples of one vsp_generator call.
 * o_chunk [B][1][(f1 - f0) * prod(upsample_rates)] contiguous; workspace >= vsp_generator_stream_workspace_bytes
 * for the largest f1 - f0 used. */
int vsp_generator_halo_frames(const vsp_ctx* ctx);
/* ABI 5: the SAMPLE-EXACT dependence of the vocoder's output frames on its input frames, from the configuration's kernel
 * sizes, strides and dilations: output frame F depends on input frames [F - *back, F + *fwd] (13 / 13 for
 * configs/config.json; the halo above is the coarser per-stage bound).  It is what the trimmed tails of a ragged batch
 * rest on (VSP_TRIM_TAILS): an utterance of L frames is computed to L + back + 1 + fwd frames -- frames [0, L + back) as in
 * the padded run, frame L + back the steady state of the zero input behind the utterance (periodic in one frame), the
 * last fwd frames the tensor end's -- and the rest of the padded waveform is filled from those, bit for bit. */
int vsp_generator_frame_dependence(const vsp_ctx* ctx, int* back, int* fwd);
int64_t vsp_generator_stream_workspace_bytes(const vsp_ctx* ctx, int B, int chunk_frames);
int vsp_generator_stream_chunk(vsp_ctx* ctx, void* stream, int B, int T, const float* z, const float* g, int f0, int f1,
                               float* o_chunk, void* workspace, int64_t workspace_bytes);

/* ---- voice conversion: replaces SynthesizerTrn.voice_conversion (reference models.py:724-732) */
/* Needs cfg.spec_channels > 0 and every enc_q.* tensor set before vsp_finalize_weights
 * (VSP_ERR_STATE otherwise).  y [B][spec][T] linear spectrogram, y_lengths[B], sid_src/sid_tgt[B]
 * int64, noise [B][inter][T] (the torch.randn_like of models.py:230).  Outputs (device,
 * contiguous): o_hat [B][1][T*prod(upsample_rates)], y_mask [B*T] uint8, z / z_p / z_hat
 * [B][inter][T]; m_q / logs_q [B][inter][T] may be NULL. */
int64_t vsp_voice_conversion_workspace_bytes(const vsp_ctx* ctx, int B, int T);
int vsp_voice_conversion(vsp_ctx* ctx, void* stream, int B, int T, const float* y, const int64_t* y_lengths,
                         const int64_t* sid_src, const int64_t* sid_tgt, const float* noise,
                         float* o_hat, uint8_t* y_mask, float* z, float* z_p, float* z_hat,
                         float* m_q, float* logs_q, void* workspace, int64_t workspace_bytes);
/* PosteriorEncoder.forward (reference models.py:212-241): y, lengths, g [B][gin], noise -> z, m, logs. */
int64_t vsp_posterior_workspace_bytes(const vsp_ctx* ctx, int B, int T);
int vsp_posterior_encoder(vsp_ctx* ctx, void* stream, int B, int T, const float* y, const int64_t* y_lengths,
                          const float* g, const float* noise, float* z, float* m, float* logs,
                          void* workspace, int64_t workspace_bytes);
/* ResidualCouplingBlock.forward(reverse=False) (reference models.py:202-206). z -> z_p. */
int vsp_flow_forward(vsp_ctx* ctx, void* stream, int B, int Tf, const float* z, const float* g,
                     const int64_t* frame_lengths, float* z_p, void* workspace, int64_t workspace_bytes);
/* Linear spectrogram of mel_processing.spectrogram_torch (reference mel_processing.py:50-69), the input of
 * voice_conversion: reflect-pad (n_fft - hop)/2 on both sides, periodic Hann window of n_fft, one-sided DFT
 * (n_fft = 2 * (cfg.spec_channels - 1)), magnitude sqrt(re^2 + im^2 + 1e-6).  audio [B][L] (device),
 * spec [B][spec_channels][T] with T = vsp_spectrogram_frames(L, hop); needs cfg.spec_channels > 0. */
int vsp_spectrogram_frames(const vsp_ctx* ctx, int L, int hop);
int64_t vsp_spectrogram_workspace_bytes(const vsp_ctx* ctx, int B, int L, int hop);
int vsp_spectrogram(vsp_ctx* ctx, void* stream, int B, int L, int hop, const float* audio, float* spec,
                    void* workspace, int64_t workspace_bytes);
/* 1 if the posterior-encoder weights are loaded (voice conversion available), else 0. */
int vsp_has_voice_conversion(const vsp_ctx* ctx);

/* piecewise_rational_quadratic_transform with tails='linear' (reference transforms.py:12-193),
 * n elements, nb bins; uw/uh [n][nb], ud [n][nb-1]; outputs y[n], logabsdet[n]. */
int vsp_rq_spline(void* stream, int64_t n, int nb, const float* x, const float* uw, const float* uh,
                  const float* ud, int inverse, float tail_bound, float* y, float* logabsdet);

/* n standard-normal draws (device, float32): element i of the Philox4x32-10 stream keyed by `seed` (counter i / 4,
 * word i % 4, Box-Muller pairs).  A function of (seed, i) only.  This is the draw vsp_decode / vsp_infer make when
 * their noise argument is NULL; it replaces torch.randn_like (reference models.py:718, 240) for C callers and is not
 * bit-compatible with torch's generator.
 * STREAM CHANGE (late round 4, first released under ABI 5): the uniforms behind Box-Muller are built from 23 random bits
 * ((x + 0.5) / 2^23, strictly inside (0, 1)) instead of 24 ((x + 0.5) / 2^24 could round to exactly 1.0 in fp32 and
 * give log(1) = 0 radii).  EVERY element of the stream differs from what ABI <= 4 libraries of rounds 1-3 drew for the
 * same seed: audio reproduced from a stored seed is not bit-reproducible across that boundary. */
int vsp_randn(void* stream, uint64_t seed, int64_t n, float* out);
/* The same stream from element `first` on: out[i] = element first + i.  A shard [lo, hi) of a batch whose noise is
 * drawn by the library passes first = lo * inter * Tf and gets exactly the elements the unsharded call would draw for
 * those utterances. */
int vsp_randn_at(void* stream, uint64_t seed, int64_t first, int64_t n, float* out);
/* Where in that stream the noise tensor of vsp_decode / vsp_infer (noise == NULL) starts: element 0 of the context's
 * [B][inter][Tf] tensor is stream element `first_element` (default 0).  A rank that synthesises utterances [lo, hi) of a
 * global batch sets lo * inter * Tf (Tf = the GLOBAL padded frame count): the result no longer depends on the shard
 * layout.  Sticky until changed; 0 restores the default.  (Added in round 4; callers that never set it are unaffected.) */
int vsp_set_noise_offset(vsp_ctx* ctx, int64_t first_element);

/* ---- mel spectrogram (reference mel_processing.py:73-112) --------------------------------- */
/* The mel basis the reference takes from librosa.filters.mel(sampling_rate, n_fft, n_mels, fmin, fmax) with that
 * function's defaults (Slaney mel scale: linear below 1 kHz, logarithmic above; triangles between successive mel
 * points; Slaney area normalisation): basis_host[n_mels][n_fft / 2 + 1], computed on the host in double precision.
 * librosa is a third-party dependency of the reference (requirements.txt, no version pinned) and is not vendored:
 * this restates its published algorithm.  fmax <= 0 means sampling_rate / 2.
 * Pinned (round 5) to a third party's implementation of the same routine: transformers.audio_utils.mel_filter_bank(norm =
 * "slaney", mel_scale = "slaney") -- "adapted from torchaudio and librosa" -- agrees to fp32 rounding for the reference's
 * configuration and three other shapes (tests/test_oracle_golden.py, on the CPU).  librosa itself and torchaudio are not
 * in the build image: the reference's own call has never been run beside it. */
int vsp_mel_filterbank(int sampling_rate, int n_fft, int n_mels, float fmin, float fmax, float* basis_host);
/* spec_to_mel_torch (reference mel_processing.py:73-82): mel = log(clamp(basis @ spec, min = 1e-5)).
 * spec [B][n_fft / 2 + 1][T] and mel [B][n_mels][T] are device pointers; the basis is built and uploaded inside the
 * call.  vsp_spectrogram followed by this call = mel_spectrogram_torch (mel_processing.py:85-112).
 * NOT on the infer path, and unlike it this entry allocates: every call hipMallocs the basis + band tables, uploads
 * them, synchronises the stream and frees them again (the header's "no hidden allocation, no sync" promise covers
 * vsp_encode / vsp_decode / vsp_infer and the per-stage entries that take a workspace, not this metric helper). */
int vsp_spec_to_mel(void* stream, int B, int T, int n_fft, int n_mels, int sampling_rate, float fmin, float fmax,
                    const float* spec, float* mel);

/* ---- vocoder operators, stand-alone (no context) ------------------------------------------ */
/* The channels-last split-f16 convolution kernels of the generator as plain operators, for unit parity
 * and for callers that hold their own weights.  Activations are device pointers, fp32, channels-last
 * [B][T][C]; weights and biases are HOST pointers in the reference's dense layout (torch.nn.Conv1d
 * weight [Cout][Cin][K], weight-norm already folded); they are packed into fragment order and uploaded
 * inside the call, which synchronises the stream before it returns (these are not the fast path:
 * vsp_generator keeps its weights packed in the arena).  terms = 3: fp32-accurate split products,
 * 1: plain f16 operands.
 *
 * vsp_cl_conv1d: out = conv1d(lrelu(x, in_slope), w, dilation, padding = dilation (K - 1) / 2) + bias
 * [+ res]; in_slope = 1 applies no activation (reference modules.py:214-221: F.leaky_relu + Conv1d).
 * Cin % 32 == 0, Cout % 32 == 0, K odd, (K - 1) * dilation <= 64. */
int vsp_cl_conv1d(void* stream, int B, int T, int Cin, int Cout, int K, int dilation, const float* x,
                  const float* w_host, const float* bias_host, float in_slope, const float* res, int terms,
                  float* out);
/* vsp_conv1d: the frame- / phoneme-rate convolution kernel (conv1d_f32_mfma) as a plain operator on the reference's
 * layout, x [B][Cin][T] -> out [B][Cout][T] (device, fp32, T contiguous):
 *   out = epilogue(conv1d(prologue(x), w, dilation, padding = dilation (K - 1) / 2) + bias)
 * prologue: x * mask (lengths != NULL and mask_in) then leaky_relu(x, in_slope) when in_act; epilogue: act 0 none,
 * 1 relu, 2 WN gate tanh(rows [0, Cout/2)) * sigmoid(rows [Cout/2, Cout)) of interleaved 32-row tiles (then the output
 * has Cout / 2 rows; commons.py:100-107; no residual with the gate); then + res [B][rows][T], then * mask when mask_out.  K odd,
 * (K - 1) * dilation + 3 <= 64, T % 4 == 0 or T == 1 (the cond(g) projections of one time step).  split_f16 = 1: fp32-accurate split-f16 MFMA (the default path of the library),
 * 0: f32 MFMA, 2 (round 6): the split-f16 COLUMN-TILE kernel -- every output row of a 64-column tile in one block, what the
 * path uses for the 1x1 convolutions of mid-size grids -- K = 1, Cin 96 or 192, Cout a multiple of 16 in [64, 576], act 0,
 * no in_act (VSP_ERR_UNSUPPORTED otherwise).  Reference: torch.nn.Conv1d as used in attentions.py:138-145, 277-285,
 * modules.py:148-176. */
int vsp_conv1d(void* stream, int B, int T, int Cin, int Cout, int K, int dilation, const float* x, const float* w_host,
               const float* bias_host, const int64_t* lengths, int mask_in, int in_act, float in_slope, int act,
               const float* res, int mask_out, int split_f16, float* out);
/* vsp_cl_resblock: ResBlock1.forward without the mask (reference modules.py:210-223):
 *   for p < n_pairs:  x = x + conv2_p(lrelu(conv1_p(lrelu(x), dilations[p]))), slope 0.1
 * w_host[2 p], w_host[2 p + 1] = conv1_p, conv2_p dense [C][C][K]; bias_host likewise [C].
 * mode 0: one launch per convolution (g16_conv), any C % 32 == 0;
 * mode 1: one launch per pair (g16_pair), C = 32 or 64;  mode 2: ONE launch (g16_chain), C = 32 or 64,
 * n_pairs <= 3.  The three modes return identical bits.  x != out. */
int vsp_cl_resblock(void* stream, int B, int T, int C, int K, int n_pairs, const int* dilations, const float* x,
                    const float* const* w_host, const float* const* bias_host, int mode, int terms, float* out);

/* ---- measurement -------------------------------------------------------------------------- */
/* When enabled, every launch of a profiled class is bracketed by a HIP event pair on the launch
 * stream.  vsp_profile_read_class synchronises those events and returns, since the last reset, for
 * one class: the number of launches, their summed duration in milliseconds, their summed
 * algorithmic FLOPs and their summed algorithmic bytes:
 *   VSP_PROF_GENERATOR  the generator's convolutions (conv_pre, ups, ResBlock convs / fused pairs):
 *                       FLOPs 2 * Cout * Cin * taps * columns * B; bytes = SURVEY.md 8d's
 *                       layer-boundary model, input once + output once per convolution, fp32 (a fused
 *                       pair is charged the two convolutions it replaces = 4 passes); bytes_ext adds
 *                       the residual / accumulate operand reads (5-6 passes per pair); bytes_moved
 *                       (ABI 7) is what the launch must move through HBM AS FUSED -- input, output,
 *                       residual and previous sum once each: 2-3 passes per fused pair or chain.
 *                       The layer-boundary model is the contract's algorithmic figure; it is NOT a
 *                       bound of a fused launch (single launches exceed the HBM peak under it), bytes_moved is.
 *                       A ragged batch (trimmed tails) is charged the frames each launch COMPUTES
 *                       (ABI 7; the plan is read back inside profiled passes: one host wait per generator call);
 *   VSP_PROF_ATTENTION  relative-position attention launches (reference attentions.py:148-179):
 *                       FLOPs 4 * H * T^2 (QK^T and PV) + 4 * H * T * (2 window + 1) (banded
 *                       relative terms) per utterance; bytes = q|k|v in + out;
 *   VSP_PROF_FRAME      every other conv1d launch (encoders, predictors, flow, projection).
 * Profiling costs two hipEventRecord per launch: time the headline with it OFF.
 * vsp_profile_read is the class VSP_PROF_GENERATOR (kept from ABI version 2). */
#define VSP_PROF_GENERATOR 0
#define VSP_PROF_ATTENTION 1
#define VSP_PROF_FRAME 2
#define VSP_PROF_CLASSES 3
int vsp_profile_enable(vsp_ctx* ctx, int on);
int vsp_profile_read(vsp_ctx* ctx, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes,
                     int reset);
int vsp_profile_read_class(vsp_ctx* ctx, int cls, int64_t* launches, double* total_ms, double* total_flops,
                           double* total_bytes, double* total_bytes_ext, double* total_bytes_moved, int reset);
/* ABI 5: the same measurement per KERNEL FAMILY of one class (so that a bench line can name its dominant kernel and a
 * reader can recompute its figures from a rocprofv3 kernel-stats table).  family = kind | log2(channels / 32) << 3 for
 * the generator class (channels = the launch's OUTPUT channels), 0 elsewhere:
 *   VSP_FAM_CONV   one launch per convolution (g16_conv / g16_convp)      VSP_FAM_UPS    a transposed up-convolution
 *   VSP_FAM_PAIR   one launch per ResBlock conv pair (g16_pp at 128 channels, g16_pair at 64, g16_rw at 32)
 *   VSP_FAM_CHAIN  one launch per ResBlock (g16_rc / g16_chain)           VSP_FAM_PRE    conv_pre (+ cond)
 * Fills up to `max_families` slots (families that saw no launch since the last reset are skipped) and returns the
 * number filled (>= 0) or a negative error code.  Call it BEFORE the vsp_profile_read_class(..., reset = 1) of the
 * same class; it never resets. */
#define VSP_FAM_OTHER 0
#define VSP_FAM_CONV 1
#define VSP_FAM_UPS 2
#define VSP_FAM_PAIR 3
#define VSP_FAM_CHAIN 4
#define VSP_FAM_PRE 5
int vsp_profile_read_families(vsp_ctx* ctx, int cls, int max_families, int* family, int64_t* launches, double* total_ms,
                              double* total_flops, double* total_bytes, double* total_bytes_moved /* ABI 7, may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* VISPEECH_HIP_H */
