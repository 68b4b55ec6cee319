// Host-side model description of libvispeech_hip: packed-arena plan, schema, context.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/vispeech_hip.h"
#include "kernels.h"

namespace vsp {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

// One conv layer in the packed arena (offsets in floats from the arena base).
struct Conv {
  int M = 0, Cin = 0, K = 1, dil = 1, pad = 0;
  int ups_s = 0, ups_p = 0;
  size_t w = 0;
  long b = -1;
  bool f16s = false;   // packed for the split-f16 MFMA path of conv_mfma.hip (same bytes as the f32 packing)
  // round 6: a second image of a small 1x1 convolution's weights, 16x16x32 A-fragment order (pack_g16_weights, K = 1), for
  // the column-tile kernels of conv_cols.hip / attn_qkv_pack_f16s (has_wg == false: none planned)
  size_t wg = 0;
  bool has_wg = false;
};

// One channels-last split-f16 conv (gen16.hip); offsets in floats (2 halfs per float).
struct ClConv {
  int Cout = 0, Cin = 0, K = 1, dil = 1, pad = 0, phases = 1, ups_p = 0;
  size_t wg = 0;           // 16x16x32 interleaved hi|lo A-fragment image (pack_g16_weights)
  long b = -1;
};

struct EncLayer {
  Conv qkv, o, f1, f2;
  size_t ek = 0, ev = 0, g1 = 0, b1 = 0, g2 = 0, b2 = 0;
};
struct EncoderW {
  std::string prefix;
  std::vector<EncLayer> layers;
};
struct FlowW {
  bool flipped = false;
  Conv pre, cond, post;
  std::vector<Conv> in, res, skip;
};
// PosteriorEncoder (reference models.py:212-241): pre 1x1, WN (k5, dilation 1, n layers), proj 1x1.
struct PosteriorW {
  Conv pre, cond, proj;  // proj: m rows, then logs rows (one launch, two destinations)
  std::vector<Conv> in, res, skip;
};
struct ResBlockW {
  int k = 0;
  std::vector<int> dil;
  std::vector<Conv> c1, c2;        // f32-MFMA packing (generator mode 0)
  std::vector<ClConv> h1, h2;      // split-f16 packing (generator mode 1)
};

struct Model {
  size_t emb_sym = 0, emb_g = 0;
  EncoderW enc[3];  // 0 enc_p.encoder, 1 pitch_predictor.pitch_net, 2 frame_prior_net.fft_block
  Conv dur_cond, dur_c1, dur_c2;
  size_t dur_g1 = 0, dur_b1 = 0, dur_g2 = 0, dur_b2 = 0, dur_pw = 0, dur_pb = 0;
  Conv pit_cond;
  size_t pit_pw = 0, pit_pb = 0;
  Conv en_cond, en_c1, en_c2;
  size_t en_g1 = 0, en_b1 = 0, en_g2 = 0, en_b2 = 0, en_lw = 0, en_lb = 0;
  size_t ppre_w = 0, ppre_b = 0, epre_w = 0, epre_b = 0;
  Conv proj;            // project.proj: rows [0, inter) = m_p, [inter, 2 inter) = logs_p -- ONE launch with two destinations
  Conv flow_cond_all;   // the cond_layer of every coupling layer's WN as one 1x1 convolution on g (FlowW::cond = its rows)
  std::vector<FlowW> flows;  // index = flow layer i (applied in order n_flows-1 .. 0)
  Conv stft;            // windowed one-sided DFT basis (rows: cos 0..spec-1, then -sin), planned when cfg.spec_channels > 0
  PosteriorW enc_q;     // planned when cfg.spec_channels > 0
  bool has_vc = false;  // enc_q weights were packed (voice conversion available)
  Conv g_pre, g_cond;
  std::vector<Conv> ups;
  std::vector<ClConv> ups_h;
  std::vector<ResBlockW> rbs;
  size_t post_w = 0;    // conv_post weight [C][K] (reference layout)
  size_t post_wt = 0;   // the same, tap-major [K][C] for the channels-last kernel
  int post_k = 7, post_c = 32;
  size_t total_floats = 0;
  bool has_cl = false;  // split-f16 channels-last generator weights present
};

// The packed arena starts with a header so that a rank that ADOPTS rank 0's bytes (vsp_adopt_packed_weights +
// vsp_commit_adopted_weights) can check what it received instead of inferring it from its own config:
//   word 0 magic, 1 ABI version, 2/3 total floats (lo/hi), 4 flags (bit 0: posterior-encoder weights packed),
//   5 hash of the vsp_config bytes + the packing switches.
constexpr size_t ARENA_HEADER_FLOATS = 64;
constexpr uint32_t ARENA_MAGIC = 0x41505356u;   // "VSPA"
constexpr uint32_t ARENA_FLAG_VC = 1u;

struct SchemaEntry {
  std::vector<int64_t> shape;
  bool used;
  bool optional = false;  // kept when set, but not required by vsp_finalize_weights (enc_q.*)
};

}  // namespace vsp

struct vsp_ctx {
  vsp_config cfg;
  int device = 0;
  std::string err;
  std::map<std::string, vsp::SchemaEntry> schema;
  std::map<std::string, vsp::HostTensor> raw;
  vsp::Model model;
  float* arena = nullptr;
  bool arena_owned = false;
  bool ready = false;
  bool t_img = true;              // ResBlock intermediates of the per-convolution stages as operand images (VSP_TIMG=0: fp32)
  bool pp_pairs = true;           // k3 / k7 conv pairs of the 128-channel stage as one launch (g16_pp; VSP_PP=0: two launches)
  bool pair_ring = false;         // VSP_PAIR=ring: the LDS-ring pair kernel on the 32-channel stage instead of g16_rw
  bool chain_ring = false;        // VSP_CHAIN_RING=1: the LDS-ring chain kernel (g16_chain) instead of g16_rc
  bool rw64 = false;              // VSP_RW64=1: the 64-channel k3 pairs on g16_rw64 (register weights; measured slower: opt-in)
  // Activation scale of the channels-last generator (round 6; VSP_ACT_SCALE_LOG2 = 0 .. 8, default 4): the fp32 activations
  // between conv_pre and conv_post are carried * act_scale (a power of two: every layer in between is positively
  // homogeneous -- leaky-relu, convolutions, sums -- once the packed biases carry the factor too, weights.cpp), so that the
  // operand split's UNSCALED lo parts (g16_common.h) fall under the f16 subnormal granularity 2^-24 only for
  // |x| < 2^-4 / act_scale: the level below which the waveform leaves the 1e-4 gate moves from -68 to about -92 dBFS; the
  // price is the upper end, |x| < 65504 / act_scale (beyond it: inf, flagged -- vsp_status).  Part of the arena's hash.
  float act_scale = 16.f;
  unsigned* flags_host = nullptr;   // pinned + mapped status word the kernels raise VSP_FLAG_* in (vsp_status)
  unsigned* flags_dev = nullptr;    // its device address
  long cols_min_blocks = 32;      // (VSP_COLS_MIN_BLOCKS)
  long cols_blocks = 256;         // ... launch_conv routes a 1x1 convolution there up to this many 64-column tiles (VSP_COLS_BLOCKS)
  bool cols = true;               // small-grid 1x1 convolutions, conv_o + LayerNorm and q | k | v + packing on the column-tile
                                  // kernels (conv_cols.hip; VSP_COLS=0: the row-tiled kernels + separate LayerNorm / pack launches)
  bool trim_tails = true;         // ragged batches: the generator runs each utterance to length + 2 halo + 1 frames and fills the
                                  // padded tail from the steady state (VSP_TRIM_TAILS=0: to the padded length; bit-identical)
  // round 5, measured and NOT adopted (profiles/r05_resblock_chains_on_side_streams.txt): the ResBlocks of a generator stage
  // (independent until their sum, reference models.py:276-285) on the caller's stream and two side streams of the context,
  // forked / joined by events, so that the ramp and the partial last round of one launch are covered by another chain's
  // blocks.  VSP_RB_STREAMS=<stage mask> (bit i = stage i; 0 = one stream, the product); bit-identical either way; profiled
  // steps always run on one stream so that the per-launch events time one kernel each.
  int rb_streams = 0;
  // the frame counts of a batch whose durations are GIVEN (duration_control tensor) depend on nothing vsp_encode computes:
  // it derives them first, copies them to this pinned buffer and records fl_ev, so that vsp_frame_lengths_host waits for
  // that copy only while the text encoder still runs (no idle GPU between the two halves of an infer call)
  bool early_fl = true;              // VSP_EARLY_FL=0: vsp_frame_lengths_host always synchronises the stream
  int64_t* fl_pinned = nullptr;
  int fl_cap = 0, fl_n = 0;
  hipEvent_t fl_ev = nullptr;
  const int64_t* fl_src = nullptr;   // device tensor the pending early copy was taken from (nullptr: none pending)
  // what the last vsp_frame_lengths_host returned, valid until the next vsp_encode / vsp_infer: lets vsp_decode see on
  // the HOST that no utterance of the batch can be trimmed (a uniform batch, one long utterance) and run the generator
  // without per-utterance extents -- the only use: a stale match can only turn trimming off or on, both of which give
  // the reference's output
  std::vector<int64_t> fl_known;
  const int64_t* fl_known_src = nullptr;
  hipStream_t side[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> sync_ev;
  int64_t noise_first = 0;        // stream index of element 0 of a library-drawn noise tensor (vsp_set_noise_offset)
  bool adopted_pending = false;   // an adopted arena whose header has not been checked yet (vsp_commit_adopted_weights)
  int gen_mode = 1;  // 0: f32 MFMA channel-major generator, 1: split-f16 (fp32-accurate) channels-last generator,
                     // 2: same kernels with plain f16 operands (VSP_GENERATOR=f16, opt-in reduced precision)
  bool frame_f16s = true;  // frame/phoneme-rate convs on the split-f16 matrix path (VSP_FRAME=f32: f32 MFMA)
  bool att_f16s = true;    // attention on the split-f16 matrix path, one pass (VSP_ATT=f32: the two-pass f32 MFMA kernel)
  int att_ksplit = -1;     // attention key-split blocks: -1 automatic (under-filled grids), 0 never, 1 always (VSP_ATT_KSPLIT)
  int chain_mask = 0x1;    // ResBlock chains (all dilation pairs of a ResBlock in one launch): bit 0 = k3, 1 = k7, 2 = k11 (VSP_CHAIN=<mask>; measured: only k3 pays)
  int chain128_mask = 0;   // conv PAIRS of the 128-channel stage as one launch each (g16_chain, one pair): bit 0 = k3, 1 = k7, 2 = k11 (VSP_CHAIN128)
  int chain_ch = 32;       // widest stage that runs chains (VSP_CHAIN_CH)
  bool fuse_pairs = true;  // ResBlock conv pairs of the 32/64-channel stages as one launch (VSP_FUSE_PAIRS=0: two launches)
  double chunk_mb = 0.0;   // generator batch chunk in MiB per activation tensor (VSP_CHUNK_MB; 0 = whole batch: measured faster)
  // profiling: HIP event pairs around the launches of a class (VSP_PROF_* in vispeech_hip.h)
  bool prof_on = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_cls;           // class of event pair i (events 2i, 2i+1)
  std::vector<int> ev_fam;           // kernel family of event pair i (VSP_FAM_* in vispeech_hip.h)
  std::vector<double> ev_flops, ev_bytes;   // algorithmic work of the launch of event pair i
  std::vector<double> ev_moved;             // bytes that launch must MOVE through HBM as fused (each operand once)
  size_t ev_used = 0;
  int64_t prof_launches[VSP_PROF_CLASSES] = {};
  double prof_flops[VSP_PROF_CLASSES] = {};
  double prof_bytes[VSP_PROF_CLASSES] = {};       // input once + output once per convolution
  double prof_bytes_ext[VSP_PROF_CLASSES] = {};   // the same plus residual / accumulate operand reads
  double prof_bytes_moved[VSP_PROF_CLASSES] = {}; // what the launches move as FUSED: input, output, residual, previous sum once each

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

namespace vsp {
void build_schema(const vsp_config& c, std::map<std::string, SchemaEntry>& out);
int plan_model(vsp_ctx* ctx);                         // fills ctx->model offsets from cfg
int fill_model(vsp_ctx* ctx, std::vector<float>& host_arena);  // folds + packs ctx->raw
}  // namespace vsp
