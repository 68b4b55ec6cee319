// Checkpoint schema, weight-norm folding and packing into the device arena.
//
// The reference's weight wire format is its state_dict key schema (reference models.py:537-622,
// utils.py:21-51).  build_schema() re-derives every key/shape from the constructor arguments;
// plan_model() lays the packed tensors out in one arena (layout depends on the config only, so
// a non-root rank can adopt rank 0's bytes after an RCCL broadcast); fill_model() folds
// weight-norm (w = g * v / ||v||, norm over all axes but 0: reference modules.py:128,135,145,
// 191-206, models.py:255) and writes every conv in MFMA A-fragment order.
#include <cmath>
#include <cstring>
#include <functional>

#include "model.h"

namespace vsp {

using Shape = std::vector<int64_t>;

static void add(std::map<std::string, SchemaEntry>& s, const std::string& k, Shape shape, bool used = true) {
  s[k] = SchemaEntry{std::move(shape), used, false};
}

static void schema_encoder(std::map<std::string, SchemaEntry>& s, const std::string& p, int n_layers,
                           const vsp_config& c) {
  const int64_t h = c.hidden_channels, f = c.filter_channels, k = c.kernel_size, dk = h / c.n_heads;
  for (int i = 0; i < n_layers; ++i) {
    const std::string a = p + ".attn_layers." + std::to_string(i);
    add(s, a + ".emb_rel_k", {1, 2 * c.window_size + 1, dk});
    add(s, a + ".emb_rel_v", {1, 2 * c.window_size + 1, dk});
    for (const char* nm : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
      add(s, a + "." + nm + ".weight", {h, h, 1});
      add(s, a + "." + nm + ".bias", {h});
    }
    for (const char* n : {"norm_layers_1.", "norm_layers_2."}) {
      add(s, p + "." + n + std::to_string(i) + ".gamma", {h});
      add(s, p + "." + n + std::to_string(i) + ".beta", {h});
    }
    const std::string q = p + ".ffn_layers." + std::to_string(i);
    add(s, q + ".conv_1.weight", {f, h, k});
    add(s, q + ".conv_1.bias", {f});
    add(s, q + ".conv_2.weight", {h, f, k});
    add(s, q + ".conv_2.bias", {h});
  }
}

static void schema_wn(std::map<std::string, SchemaEntry>& s, const std::string& p, int64_t hidden, int64_t kernel,
                      int n_layers, int64_t gin, bool used) {
  for (int i = 0; i < n_layers; ++i) {
    const std::string a = p + ".in_layers." + std::to_string(i);
    add(s, a + ".bias", {2 * hidden}, used);
    add(s, a + ".weight_g", {2 * hidden, 1, 1}, used);
    add(s, a + ".weight_v", {2 * hidden, hidden, kernel}, used);
    const int64_t rs = i < n_layers - 1 ? 2 * hidden : hidden;
    const std::string r = p + ".res_skip_layers." + std::to_string(i);
    add(s, r + ".bias", {rs}, used);
    add(s, r + ".weight_g", {rs, 1, 1}, used);
    add(s, r + ".weight_v", {rs, hidden, 1}, used);
  }
  if (gin) {
    add(s, p + ".cond_layer.bias", {2 * hidden * n_layers}, used);
    add(s, p + ".cond_layer.weight_g", {2 * hidden * n_layers, 1, 1}, used);
    add(s, p + ".cond_layer.weight_v", {2 * hidden * n_layers, gin, 1}, used);
  }
}

void build_schema(const vsp_config& c, std::map<std::string, SchemaEntry>& s) {
  s.clear();
  const int64_t h = c.hidden_channels, gin = c.gin_channels, inter = c.inter_channels;
  add(s, "enc_p.symbol_emb.weight", {c.n_vocab, h});
  schema_encoder(s, "enc_p.encoder", c.n_layers, c);
  add(s, "enc_p.proj.weight", {2 * inter, h, 1}, false);
  add(s, "enc_p.proj.bias", {2 * inter}, false);
  const int64_t c0 = c.upsample_initial_channel;
  add(s, "dec.conv_pre.weight", {c0, inter, 7});
  add(s, "dec.conv_pre.bias", {c0});
  int64_t ch = c0;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int64_t cin = c0 >> i, cout = c0 >> (i + 1);
    const std::string p = "dec.ups." + std::to_string(i);
    add(s, p + ".bias", {cout});
    add(s, p + ".weight_g", {cin, 1, 1});
    add(s, p + ".weight_v", {cin, cout, c.upsample_kernel_sizes[i]});
    ch = cout;
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      const std::string rb = "dec.resblocks." + std::to_string(i * c.n_resblock_kernels + j);
      for (const char* grp : {".convs1.", ".convs2."})
        for (int m = 0; m < c.n_resblock_dilations; ++m) {
          const std::string q = rb + grp + std::to_string(m);
          add(s, q + ".bias", {ch});
          add(s, q + ".weight_g", {ch, 1, 1});
          add(s, q + ".weight_v", {ch, ch, c.resblock_kernel_sizes[j]});
        }
    }
  }
  add(s, "dec.conv_post.weight", {1, ch, 7});
  add(s, "dec.cond.weight", {c0, gin, 1});
  add(s, "dec.cond.bias", {c0});
  // posterior encoder (reference models.py:596): in checkpoints, never read by infer; kept as
  // OPTIONAL tensors for voice conversion when the config names spec_channels, otherwise
  // accepted unchecked and ignored (see vsp_set_weight)
  if (c.spec_channels > 0) {
    std::map<std::string, SchemaEntry> q;
    add(q, "enc_q.pre.weight", {h, c.spec_channels, 1});
    add(q, "enc_q.pre.bias", {h});
    schema_wn(q, "enc_q.enc", h, c.flow_kernel, c.posterior_layers, gin, true);
    add(q, "enc_q.proj.weight", {2 * inter, h, 1});
    add(q, "enc_q.proj.bias", {2 * inter});
    for (auto& kv : q) { kv.second.optional = true; s[kv.first] = kv.second; }
  }
  for (int i = 0; i < c.n_flows; ++i) {
    const std::string p = "flow.flows." + std::to_string(2 * i);
    add(s, p + ".pre.weight", {h, inter / 2, 1});
    add(s, p + ".pre.bias", {h});
    schema_wn(s, p + ".enc", h, c.flow_kernel, c.flow_layers, gin, true);
    add(s, p + ".post.weight", {inter / 2, h, 1});
    add(s, p + ".post.bias", {inter / 2});
  }
  const int64_t f = c.dur_filter;
  add(s, "duration_predictor.conv_1.weight", {f, h, 3});
  add(s, "duration_predictor.conv_1.bias", {f});
  add(s, "duration_predictor.norm_1.gamma", {f});
  add(s, "duration_predictor.norm_1.beta", {f});
  add(s, "duration_predictor.conv_2.weight", {f, f, 3});
  add(s, "duration_predictor.conv_2.bias", {f});
  add(s, "duration_predictor.norm_2.gamma", {f});
  add(s, "duration_predictor.norm_2.beta", {f});
  add(s, "duration_predictor.proj.weight", {1, f, 1});
  add(s, "duration_predictor.proj.bias", {1});
  add(s, "duration_predictor.cond.weight", {h, gin, 1});
  add(s, "duration_predictor.cond.bias", {h});
  add(s, "frame_prior_net.emb.weight", {121, h}, false);
  schema_encoder(s, "frame_prior_net.fft_block", c.n_layers, c);
  schema_encoder(s, "pitch_predictor.pitch_net", c.pitch_layers, c);
  add(s, "pitch_predictor.proj_f0.weight", {1, h, 1});
  add(s, "pitch_predictor.proj_f0.bias", {1});
  add(s, "pitch_predictor.cond.weight", {h, gin, 1});
  add(s, "pitch_predictor.cond.bias", {h});
  const int64_t e = c.energy_filter;
  const std::string ep = "energy_predictor.predictor";
  add(s, ep + ".conv_layer.conv_1.conv.weight", {e, h, 3});
  add(s, ep + ".conv_layer.conv_1.conv.bias", {e});
  add(s, ep + ".conv_layer.layer_norm_1.weight", {e});
  add(s, ep + ".conv_layer.layer_norm_1.bias", {e});
  add(s, ep + ".conv_layer.conv_2.conv.weight", {e, e, 3});
  add(s, ep + ".conv_layer.conv_2.conv.bias", {e});
  add(s, ep + ".conv_layer.layer_norm_2.weight", {e});
  add(s, ep + ".conv_layer.layer_norm_2.bias", {e});
  add(s, ep + ".linear_layer.weight", {1, e});
  add(s, ep + ".linear_layer.bias", {1});
  add(s, ep + ".proj.weight", {h, 1}, false);
  add(s, ep + ".proj.bias", {h}, false);
  add(s, "energy_predictor.cond.weight", {h, gin, 1});
  add(s, "energy_predictor.cond.bias", {h});
  add(s, "project.proj.weight", {2 * inter, h, 1});
  add(s, "project.proj.bias", {2 * inter});
  add(s, "pitch_prenet.weight", {h, 1, 3});
  add(s, "pitch_prenet.bias", {h});
  add(s, "energy_prenet.weight", {h, 1, 3});
  add(s, "energy_prenet.bias", {h});
  add(s, "emb_g.weight", {c.n_speakers, gin});
}

// ------------------------------------------------------------------------------------------
struct Planner {
  size_t cur = 0;
  bool f16s = false;   // packing of the convs planned from here on
  size_t raw(size_t n) {
    const size_t o = cur;
    cur += (n + 63) / 64 * 64;  // 256-byte granules
    return o;
  }
  Conv conv(int M, int Cin, int K, int dil, int pad, bool bias) {
    Conv c;
    c.M = M; c.Cin = Cin; c.K = K; c.dil = dil; c.pad = pad;
    c.w = raw(packed_conv_floats(M, Cin, K));
    c.b = bias ? (long)raw((size_t)M) : -1;
    c.f16s = f16s;
    // (conv_cols.hip: 1x1, whole 32-channel chunks up to 192 inputs, whole 16-row tiles up to the 576 rows of q | k | v)
    if (f16s && K == 1 && Cin % 32 == 0 && Cin <= 192 && M % 16 == 0 && M >= 64 && M <= 576) {
      c.wg = raw(packed_g16_halfs(M, Cin, 1) / 2);
      c.has_wg = true;
    }
    return c;
  }
  ClConv clconv(int Cout, int Cin, int K, int dil, int pad, int phases, int ups_p) {
    ClConv c;
    c.Cout = Cout; c.Cin = Cin; c.K = K; c.dil = dil; c.pad = pad; c.phases = phases; c.ups_p = ups_p;
    c.wg = raw(packed_g16_halfs(phases * Cout, Cin, K) / 2);
    c.b = (long)raw((size_t)Cout);
    return c;
  }
};

static void plan_encoder(Planner& p, EncoderW& e, const std::string& prefix, int n_layers, const vsp_config& c) {
  const int h = c.hidden_channels, f = c.filter_channels, k = c.kernel_size, dk = h / c.n_heads;
  const int nrel = 2 * c.window_size + 1;
  e.prefix = prefix;
  e.layers.resize(n_layers);
  for (auto& L : e.layers) {
    L.qkv = p.conv(3 * h, h, 1, 1, 0, true);
    L.o = p.conv(h, h, 1, 1, 0, true);
    L.ek = p.raw((size_t)nrel * dk);
    L.ev = p.raw((size_t)nrel * dk);
    L.g1 = p.raw(h); L.b1 = p.raw(h);
    L.f1 = p.conv(f, h, k, 1, (k - 1) / 2, true);
    L.f2 = p.conv(h, f, k, 1, (k - 1) / 2, true);
    L.g2 = p.raw(h); L.b2 = p.raw(h);
  }
}

int plan_model(vsp_ctx* ctx) {
  const vsp_config& c = ctx->cfg;
  Model& m = ctx->model;
  m = Model();
  const int h = c.hidden_channels, gin = c.gin_channels, inter = c.inter_channels;
  if (h <= 0 || c.n_heads <= 0 || h % c.n_heads) return ctx->fail(VSP_ERR_ARG, "hidden_channels %% n_heads != 0");
  const int dk = h / c.n_heads;
  if (dk != 96 && dk != 64 && dk != 32)
    return ctx->fail(VSP_ERR_UNSUPPORTED, "head dim %d: attention kernel covers 32/64/96", dk);
  if (h % 32) return ctx->fail(VSP_ERR_UNSUPPORTED, "hidden_channels must be a multiple of 32 (WN gate tiles)");
  if (gin <= 0 || c.n_speakers <= 1)
    return ctx->fail(VSP_ERR_UNSUPPORTED, "n_speakers > 1 and gin_channels > 0 are required (the reference's "
                                          "EnergyPredictor dereferences g unconditionally, frame_prior_network.py:120)");
  if (inter % 2) return ctx->fail(VSP_ERR_ARG, "inter_channels must be even");
  if (inter % 32) return ctx->fail(VSP_ERR_UNSUPPORTED, "inter_channels must be a multiple of 32 (the projection's two halves are row tiles of one launch)");
  if (2 * c.window_size + 1 > 16) return ctx->fail(VSP_ERR_UNSUPPORTED, "window_size > 7");
  if (c.n_upsamples < 1 || c.n_upsamples > VSP_MAX_LIST || c.n_resblock_kernels < 1 ||
      c.n_resblock_kernels > VSP_MAX_LIST || c.n_resblock_dilations < 1 || c.n_resblock_dilations > VSP_MAX_LIST)
    return ctx->fail(VSP_ERR_ARG, "list sizes out of range");
  Planner p;
  p.raw(ARENA_HEADER_FLOATS);  // arena header (model.h): magic, ABI, size, flags, config hash
  p.f16s = ctx->frame_f16s;   // everything up to the generator: encoders, predictors, projection, flows, posterior
  m.emb_sym = p.raw((size_t)c.n_vocab * h);
  m.emb_g = p.raw((size_t)c.n_speakers * gin);
  plan_encoder(p, m.enc[0], "enc_p.encoder", c.n_layers, c);
  plan_encoder(p, m.enc[1], "pitch_predictor.pitch_net", c.pitch_layers, c);
  plan_encoder(p, m.enc[2], "frame_prior_net.fft_block", c.n_layers, c);
  const int f = c.dur_filter;
  m.dur_cond = p.conv(h, gin, 1, 1, 0, true);
  m.dur_c1 = p.conv(f, h, 3, 1, 1, true);
  m.dur_c2 = p.conv(f, f, 3, 1, 1, true);
  m.dur_g1 = p.raw(f); m.dur_b1 = p.raw(f); m.dur_g2 = p.raw(f); m.dur_b2 = p.raw(f);
  m.dur_pw = p.raw(f); m.dur_pb = p.raw(1);
  m.pit_cond = p.conv(h, gin, 1, 1, 0, true);
  m.pit_pw = p.raw(h); m.pit_pb = p.raw(1);
  const int e = c.energy_filter;
  m.en_cond = p.conv(h, gin, 1, 1, 0, true);
  m.en_c1 = p.conv(e, h, 3, 1, 1, true);
  m.en_c2 = p.conv(e, e, 3, 1, 1, true);
  m.en_g1 = p.raw(e); m.en_b1 = p.raw(e); m.en_g2 = p.raw(e); m.en_b2 = p.raw(e);
  m.en_lw = p.raw(e); m.en_lb = p.raw(1);
  m.ppre_w = p.raw((size_t)h * 3); m.ppre_b = p.raw(h);
  m.epre_w = p.raw((size_t)h * 3); m.epre_b = p.raw(h);
  m.proj = p.conv(2 * inter, h, 1, 1, 0, true);
  // flows: applied in order n_flows-1 .. 0, each preceded by a Flip (reference models.py:202-209);
  // the flip is folded into channel order: before layer i the tensor has seen n_flows - i flips.
  m.flows.resize(c.n_flows);
  const int half = inter / 2, fk = c.flow_kernel, fl = c.flow_layers;
  // every coupling layer's cond_layer(g) in ONE launch: the packed image is m-tile major, so the rows of flow i are a
  // contiguous run of it (and of the bias vector) -- FlowW::cond is a view
  m.flow_cond_all = p.conv(c.n_flows * 2 * h * fl, gin, 1, 1, 0, true);
  for (int i = 0; i < c.n_flows; ++i) {
    FlowW& F = m.flows[i];
    F.flipped = ((c.n_flows - i) % 2) == 1;
    F.pre = p.conv(h, half, 1, 1, 0, true);
    F.cond = m.flow_cond_all;
    F.cond.M = 2 * h * fl;
    F.cond.w = m.flow_cond_all.w + (size_t)i * packed_conv_floats(2 * h * fl, gin, 1);
    F.cond.b = m.flow_cond_all.b + (long)i * 2 * h * fl;
    for (int l = 0; l < fl; ++l) {
      F.in.push_back(p.conv(2 * h, h, fk, 1, (fk - 1) / 2, true));
      if (l < fl - 1) F.res.push_back(p.conv(2 * h, h, 1, 1, 0, true));   // residual rows, then skip rows: one launch
      F.skip.push_back(p.conv(h, h, 1, 1, 0, true));
    }
    F.post = p.conv(half, h, 1, 1, 0, true);
  }
  if (c.n_flows % 2)
    return ctx->fail(VSP_ERR_UNSUPPORTED, "odd n_flows leaves the latent channel-flipped; only even counts are folded");
  if (c.spec_channels > 0) {
    if (c.posterior_layers < 1) return ctx->fail(VSP_ERR_ARG, "posterior_layers < 1");
    // spectrogram front end of voice conversion: DFT as a 1x1 conv over the n_fft samples of a frame
    const int n_fft = 2 * (c.spec_channels - 1);
    m.stft = p.conv(2 * c.spec_channels, n_fft, 1, 1, 0, false);
    PosteriorW& Q = m.enc_q;
    const int ql = c.posterior_layers;
    Q.pre = p.conv(h, c.spec_channels, 1, 1, 0, true);
    Q.cond = p.conv(2 * h * ql, gin, 1, 1, 0, true);
    for (int l = 0; l < ql; ++l) {
      Q.in.push_back(p.conv(2 * h, h, fk, 1, (fk - 1) / 2, true));
      if (l < ql - 1) Q.res.push_back(p.conv(2 * h, h, 1, 1, 0, true));
      Q.skip.push_back(p.conv(h, h, 1, 1, 0, true));
    }
    Q.proj = p.conv(2 * inter, h, 1, 1, 0, true);
  }
  // generator (its channel-major f32 form is the second implementation kept for VSP_GENERATOR=f32).
  // ctx->gen_mode holds the requested mode here (vsp_create parses VSP_GENERATOR before planning); it falls back to 0
  // when the split-f16 channels-last kernels do not cover the configuration, BEFORE conv_pre / cond are planned.
  const int c0 = c.upsample_initial_channel;
  int ch = c0;
  // the split-f16 channels-last generator covers channel counts of 32 or multiples of 64
  m.has_cl = true;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int cin = c0 >> i, cout = c0 >> (i + 1);
    if (cin % 32 || cout % 32 || (cin != 32 && cin % 64) || (cout != 32 && cout % 64)) m.has_cl = false;
  }
  if ((c0 >> c.n_upsamples) > 64) m.has_cl = false;  // conv_post_cl covers <= 64 channels
  if (!m.has_cl) ctx->gen_mode = 0;
  // (conv_pre and the speaker conditioning run on the split-f16 path whenever the vocoder does)
  p.f16s = ctx->frame_f16s && ctx->gen_mode != 0;
  m.g_pre = p.conv(c0, inter, 7, 1, 3, true);
  m.g_cond = p.conv(c0, gin, 1, 1, 0, true);
  p.f16s = false;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const int s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    if (s < 1 || k % s || (k - s) % 2)
      return ctx->fail(VSP_ERR_UNSUPPORTED, "upsample kernel %d / stride %d: polyphase form needs k %% s == 0", k, s);
    const int cin = c0 >> i, cout = c0 >> (i + 1), kt = k / s;
    if (cout < 1) return ctx->fail(VSP_ERR_ARG, "too many upsamples for upsample_initial_channel");
    Conv u = p.conv(cout * s, cin, kt, 1, kt - 1, true);
    u.ups_s = s;
    u.ups_p = (k - s) / 2;
    m.ups.push_back(u);
    const bool cl_ok = m.has_cl;
    if (cl_ok) m.ups_h.push_back(p.clconv(cout, cin, kt, 1, kt - 1, s, (k - s) / 2));
    ch = cout;
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      ResBlockW rb;
      rb.k = c.resblock_kernel_sizes[j];
      for (int q = 0; q < c.n_resblock_dilations; ++q) {
        const int d = c.resblock_dilation_sizes[j][q];
        if ((rb.k - 1) * d + 3 > CONV_HALO) return ctx->fail(VSP_ERR_UNSUPPORTED, "resblock halo (k-1)*d > %d", CONV_HALO);
        rb.dil.push_back(d);
        rb.c1.push_back(p.conv(ch, ch, rb.k, d, (rb.k * d - d) / 2, true));
        rb.c2.push_back(p.conv(ch, ch, rb.k, 1, (rb.k - 1) / 2, true));
        if (cl_ok) {
          rb.h1.push_back(p.clconv(ch, ch, rb.k, d, (rb.k * d - d) / 2, 1, 0));
          rb.h2.push_back(p.clconv(ch, ch, rb.k, 1, (rb.k - 1) / 2, 1, 0));
        }
      }
      m.rbs.push_back(rb);
    }
  }
  if (ch > 32) return ctx->fail(VSP_ERR_UNSUPPORTED, "conv_post kernel covers <= 32 input channels (got %d)", ch);
  m.post_c = ch;
  m.post_k = 7;
  m.post_w = p.raw((size_t)ch * 7);
  m.post_wt = p.raw((size_t)ch * 7);
  m.total_floats = p.cur;
  return VSP_OK;
}

// ------------------------------------------------------------------------------------------
struct Filler {
  vsp_ctx* ctx;
  std::vector<float>& arena;
  std::map<std::string, HostTensor> folded;
  bool ok = true;
  std::string missing;
  float too_large = 0.f;       // a split-f16 weight whose packed form (* G16_WSCALE) leaves the f16 range

  // the split-f16 kernels take their weights * G16_WSCALE as f16 pairs (kernels.h): |w| must stay below 65504 / 256
  void check_f16_range(const std::vector<float>& dense) {
    for (float w : dense)
      if (!(std::fabs(w) * G16_WSCALE < 65000.f)) too_large = std::max(too_large, std::isfinite(w) ? std::fabs(w) : INFINITY);
  }

  const HostTensor* get(const std::string& name) {
    auto it = ctx->raw.find(name);
    if (it != ctx->raw.end()) return &it->second;
    auto f = folded.find(name);
    if (f != folded.end()) return &f->second;
    // "<x>.weight" from "<x>.weight_g" / "<x>.weight_v"
    auto iv = ctx->raw.find(name + "_v"), ig = ctx->raw.find(name + "_g");
    if (iv == ctx->raw.end() || ig == ctx->raw.end()) {
      if (ok) missing = name;
      ok = false;
      return nullptr;
    }
    const HostTensor& v = iv->second;
    const HostTensor& g = ig->second;
    HostTensor w;
    w.shape = v.shape;
    w.data.resize(v.data.size());
    const size_t rows = (size_t)v.shape[0], inner = v.data.size() / rows;
    for (size_t r = 0; r < rows; ++r) {
      double nrm = 0.0;
      for (size_t i = 0; i < inner; ++i) { const double x = v.data[r * inner + i]; nrm += x * x; }
      nrm = std::sqrt(nrm);
      const double sc = (double)g.data[r] / nrm;
      for (size_t i = 0; i < inner; ++i) w.data[r * inner + i] = (float)((double)v.data[r * inner + i] * sc);
    }
    return &(folded[name] = std::move(w));
  }
  void copy_raw(size_t off, const std::string& name, size_t n) {
    const HostTensor* t = get(name);
    if (!t) return;
    std::memcpy(arena.data() + off, t->data.data(), n * sizeof(float));
  }
  // conv from accessor functions
  void conv(const Conv& c, const std::function<float(int, int, int)>& w, const std::function<float(int)>& b) {
    if (!ok) return;
    std::vector<float> dense((size_t)c.M * c.Cin * c.K);
    for (int r = 0; r < c.M; ++r)
      for (int ci = 0; ci < c.Cin; ++ci)
        for (int t = 0; t < c.K; ++t) dense[((size_t)r * c.Cin + ci) * c.K + t] = w(r, ci, t);
    if (c.f16s) check_f16_range(dense);
    if (c.f16s) pack_conv_weights_f16s(arena.data() + c.w, c.M, c.Cin, c.K, dense.data());
    else pack_conv_weights(arena.data() + c.w, c.M, c.Cin, c.K, dense.data());
    if (c.has_wg) pack_g16_weights(reinterpret_cast<uint16_t*>(arena.data() + c.wg), c.M, c.Cin, 1, dense.data());
    if (c.b >= 0)
      for (int r = 0; r < c.M; ++r) arena[c.b + r] = b(r);
  }
  // channels-last split-f16 conv from an accessor W(phase, co, ci, tap)
  void clconv(const ClConv& c, const std::function<float(int, int, int, int)>& w, const std::function<float(int)>& b) {
    if (!ok) return;
    std::vector<float> dense((size_t)c.phases * c.Cout * c.Cin * c.K);
    for (int ph = 0; ph < c.phases; ++ph)
      for (int co = 0; co < c.Cout; ++co)
        for (int ci = 0; ci < c.Cin; ++ci)
          for (int t = 0; t < c.K; ++t) dense[(((size_t)ph * c.Cout + co) * c.Cin + ci) * c.K + t] = w(ph, co, ci, t);
    if (ctx->gen_mode != 0) check_f16_range(dense);       // (VSP_GENERATOR=f32 never multiplies this image)
    // [phase][co][ci][tap] is [row = phase * Cout + co][ci][tap]: the stacked-phase form gen16.hip multiplies
    pack_g16_weights(reinterpret_cast<uint16_t*>(arena.data() + c.wg), c.phases * c.Cout, c.Cin, c.K, dense.data());
    // (kernels.h: the bias rides in the scaled accumulator; model.h: ... of activations that are carried * act_scale)
    for (int co = 0; co < c.Cout; ++co) arena[c.b + co] = b(co) * G16_WSCALE * ctx->act_scale;
  }
  void clconv_plain(const ClConv& c, const std::string& wname, const std::string& bname) {
    const HostTensor* W = get(wname);
    const HostTensor* B = get(bname);
    if (!ok) return;
    const float* wd = W->data.data();
    const float* bd = B->data.data();
    const int Cin = c.Cin, K = c.K;
    clconv(c, [=](int, int co, int ci, int t) { return wd[((size_t)co * Cin + ci) * K + t]; },
           [=](int co) { return bd[co]; });
  }
  // plain Conv1d weight [M][Cin][K] + bias, rows taken from [row0, row0+M)
  void conv_plain(const Conv& c, const std::string& wname, const std::string& bname, int row0 = 0) {
    const HostTensor* W = get(wname);
    const HostTensor* B = bname.empty() ? nullptr : get(bname);
    if (!ok) return;
    const float* wd = W->data.data();
    const float* bd = B ? B->data.data() : nullptr;
    const int Cin = c.Cin, K = c.K;
    conv(c, [=](int r, int ci, int t) { return wd[((size_t)(row0 + r) * Cin + ci) * K + t]; },
         [=](int r) { return bd ? bd[row0 + r] : 0.f; });
  }
};

static void fill_encoder(Filler& f, const EncoderW& e, const vsp_config& c) {
  const int h = c.hidden_channels, dk = h / c.n_heads, nrel = 2 * c.window_size + 1;
  for (size_t i = 0; i < e.layers.size(); ++i) {
    const EncLayer& L = e.layers[i];
    const std::string a = e.prefix + ".attn_layers." + std::to_string(i);
    const HostTensor* wq = f.get(a + ".conv_q.weight");
    const HostTensor* wk = f.get(a + ".conv_k.weight");
    const HostTensor* wv = f.get(a + ".conv_v.weight");
    const HostTensor* bq = f.get(a + ".conv_q.bias");
    const HostTensor* bk = f.get(a + ".conv_k.bias");
    const HostTensor* bv = f.get(a + ".conv_v.bias");
    if (!f.ok) return;
    const float* ws[3] = {wq->data.data(), wk->data.data(), wv->data.data()};
    const float* bs[3] = {bq->data.data(), bk->data.data(), bv->data.data()};
    f.conv(L.qkv, [=](int r, int ci, int) { return ws[r / h][(size_t)(r % h) * h + ci]; },
           [=](int r) { return bs[r / h][r % h]; });
    f.conv_plain(L.o, a + ".conv_o.weight", a + ".conv_o.bias");
    f.copy_raw(L.ek, a + ".emb_rel_k", (size_t)nrel * dk);
    f.copy_raw(L.ev, a + ".emb_rel_v", (size_t)nrel * dk);
    const std::string n1 = e.prefix + ".norm_layers_1." + std::to_string(i);
    const std::string n2 = e.prefix + ".norm_layers_2." + std::to_string(i);
    f.copy_raw(L.g1, n1 + ".gamma", h); f.copy_raw(L.b1, n1 + ".beta", h);
    f.copy_raw(L.g2, n2 + ".gamma", h); f.copy_raw(L.b2, n2 + ".beta", h);
    const std::string q = e.prefix + ".ffn_layers." + std::to_string(i);
    f.conv_plain(L.f1, q + ".conv_1.weight", q + ".conv_1.bias");
    f.conv_plain(L.f2, q + ".conv_2.weight", q + ".conv_2.bias");
  }
}

// packed row -> original row of a WN in_layer / cond block with `h` tanh + `h` sigmoid rows:
// 32-row tiles alternate tanh rows [32t,32t+32) and the matching sigmoid rows.
static inline int gate_row(int pr, int h) {
  const int t = pr / 64, within = pr % 64;
  return within < 32 ? 32 * t + within : h + 32 * t + (within - 32);
}

// modules.WN (reference modules.py:113-176): cond_layer rows and in_layer rows are permuted into
// tanh/sigmoid tile pairs (gate_row); a res_skip layer is one conv of 2 h rows with two destinations (ConvArgs::split_row).
static void fill_wn(Filler& f, const std::string& p, const Conv& cond, const std::vector<Conv>& in,
                    const std::vector<Conv>& res, const std::vector<Conv>& skip, int nl, int h, int gin, int fk) {
  {
    const HostTensor* W = f.get(p + ".cond_layer.weight");
    const HostTensor* B = f.get(p + ".cond_layer.bias");
    if (!f.ok) return;
    const float* wd = W->data.data();
    const float* bd = B->data.data();
    auto orig = [=](int r) { return (r / (2 * h)) * 2 * h + gate_row(r % (2 * h), h); };
    f.conv(cond, [=](int r, int ci, int) { return wd[(size_t)orig(r) * gin + ci]; },
           [=](int r) { return bd[orig(r)]; });
  }
  for (int l = 0; l < nl && f.ok; ++l) {
    const std::string il = p + ".in_layers." + std::to_string(l);
    const HostTensor* W = f.get(il + ".weight");
    const HostTensor* B = f.get(il + ".bias");
    if (!f.ok) return;
    const float* wd = W->data.data();
    const float* bd = B->data.data();
    f.conv(in[l], [=](int r, int ci, int t) { return wd[((size_t)gate_row(r, h) * h + ci) * fk + t]; },
           [=](int r) { return bd[gate_row(r, h)]; });
    const std::string rs = p + ".res_skip_layers." + std::to_string(l);
    if (l < nl - 1) {
      f.conv_plain(res[l], rs + ".weight", rs + ".bias", 0);      // all 2 h rows: residual half, skip half
      f.conv_plain(skip[l], rs + ".weight", rs + ".bias", h);     // (the skip half alone: the two-launch form)
    } else {
      f.conv_plain(skip[l], rs + ".weight", rs + ".bias", 0);
    }
  }
}

int fill_model(vsp_ctx* ctx, std::vector<float>& arena) {
  const vsp_config& c = ctx->cfg;
  Model& m = ctx->model;
  arena.assign(m.total_floats, 0.f);
  Filler f{ctx, arena};
  const int h = c.hidden_channels, gin = c.gin_channels, inter = c.inter_channels;
  f.copy_raw(m.emb_sym, "enc_p.symbol_emb.weight", (size_t)c.n_vocab * h);
  f.copy_raw(m.emb_g, "emb_g.weight", (size_t)c.n_speakers * gin);
  for (int i = 0; i < 3; ++i) fill_encoder(f, m.enc[i], c);
  f.conv_plain(m.dur_cond, "duration_predictor.cond.weight", "duration_predictor.cond.bias");
  f.conv_plain(m.dur_c1, "duration_predictor.conv_1.weight", "duration_predictor.conv_1.bias");
  f.conv_plain(m.dur_c2, "duration_predictor.conv_2.weight", "duration_predictor.conv_2.bias");
  f.copy_raw(m.dur_g1, "duration_predictor.norm_1.gamma", c.dur_filter);
  f.copy_raw(m.dur_b1, "duration_predictor.norm_1.beta", c.dur_filter);
  f.copy_raw(m.dur_g2, "duration_predictor.norm_2.gamma", c.dur_filter);
  f.copy_raw(m.dur_b2, "duration_predictor.norm_2.beta", c.dur_filter);
  f.copy_raw(m.dur_pw, "duration_predictor.proj.weight", c.dur_filter);
  f.copy_raw(m.dur_pb, "duration_predictor.proj.bias", 1);
  f.conv_plain(m.pit_cond, "pitch_predictor.cond.weight", "pitch_predictor.cond.bias");
  f.copy_raw(m.pit_pw, "pitch_predictor.proj_f0.weight", h);
  f.copy_raw(m.pit_pb, "pitch_predictor.proj_f0.bias", 1);
  const std::string ep = "energy_predictor.predictor";
  f.conv_plain(m.en_cond, "energy_predictor.cond.weight", "energy_predictor.cond.bias");
  f.conv_plain(m.en_c1, ep + ".conv_layer.conv_1.conv.weight", ep + ".conv_layer.conv_1.conv.bias");
  f.conv_plain(m.en_c2, ep + ".conv_layer.conv_2.conv.weight", ep + ".conv_layer.conv_2.conv.bias");
  f.copy_raw(m.en_g1, ep + ".conv_layer.layer_norm_1.weight", c.energy_filter);
  f.copy_raw(m.en_b1, ep + ".conv_layer.layer_norm_1.bias", c.energy_filter);
  f.copy_raw(m.en_g2, ep + ".conv_layer.layer_norm_2.weight", c.energy_filter);
  f.copy_raw(m.en_b2, ep + ".conv_layer.layer_norm_2.bias", c.energy_filter);
  f.copy_raw(m.en_lw, ep + ".linear_layer.weight", c.energy_filter);
  f.copy_raw(m.en_lb, ep + ".linear_layer.bias", 1);
  f.copy_raw(m.ppre_w, "pitch_prenet.weight", (size_t)h * 3);
  f.copy_raw(m.ppre_b, "pitch_prenet.bias", h);
  f.copy_raw(m.epre_w, "energy_prenet.weight", (size_t)h * 3);
  f.copy_raw(m.epre_b, "energy_prenet.bias", h);
  f.conv_plain(m.proj, "project.proj.weight", "project.proj.bias", 0);

  const int half = inter / 2, fl = c.flow_layers, fk = c.flow_kernel;
  for (int i = 0; i < c.n_flows && f.ok; ++i) {
    const FlowW& F = m.flows[i];
    const std::string p = "flow.flows." + std::to_string(2 * i);
    {
      const HostTensor* W = f.get(p + ".pre.weight");
      const HostTensor* B = f.get(p + ".pre.bias");
      if (!f.ok) break;
      const float* wd = W->data.data();
      const float* bd = B->data.data();
      const bool flip = F.flipped;
      // flipped: the coupling's x0 (logical channels 0..half-1) lives in physical channels
      // inter-1 .. half, i.e. physical (half + ci') holds logical half-1-ci'.
      f.conv(F.pre, [=](int r, int ci, int) { return wd[(size_t)r * half + (flip ? half - 1 - ci : ci)]; },
             [=](int r) { return bd[r]; });
    }
    fill_wn(f, p + ".enc", F.cond, F.in, F.res, F.skip, fl, h, gin, fk);
    if (!f.ok) break;
    {
      const HostTensor* W = f.get(p + ".post.weight");
      const HostTensor* B = f.get(p + ".post.bias");
      if (!f.ok) break;
      const float* wd = W->data.data();
      const float* bd = B->data.data();
      const bool flip = F.flipped;
      // flipped: logical x1 channel cl (logical index half+cl) is physical half-1-cl.
      f.conv(F.post, [=](int r, int ci, int) { return wd[(size_t)(flip ? half - 1 - r : r) * h + ci]; },
             [=](int r) { return bd[flip ? half - 1 - r : r]; });
    }
  }

  // windowed DFT basis of the spectrogram front end (config-derived, no checkpoint tensor): row r < spec is
  // cos(2 pi r n / N) * hann[n], row spec + r is -sin(...) * hann[n]; periodic Hann (torch.hann_window)
  if (c.spec_channels > 0 && f.ok) {
    const int spec = c.spec_channels, N = 2 * (spec - 1);
    std::vector<double> hann(N);
    for (int n = 0; n < N; ++n) hann[n] = 0.5 - 0.5 * std::cos(2.0 * M_PI * n / N);
    std::vector<double> cs(N), sn(N);
    for (int k = 0; k < N; ++k) { cs[k] = std::cos(2.0 * M_PI * k / N); sn[k] = std::sin(2.0 * M_PI * k / N); }
    f.conv(m.stft, [&](int row, int n, int) {
      const int r = row < spec ? row : row - spec;
      const int k = (int)(((long long)r * n) % N);        // exact angle reduction
      return (float)((row < spec ? cs[k] : -sn[k]) * hann[n]);
    }, [](int) { return 0.f; });
  }
  // posterior encoder (optional): packed only when every enc_q tensor was supplied
  m.has_vc = false;
  if (c.spec_channels > 0 && f.ok) {
    bool all = true;
    for (const auto& kv : ctx->schema)
      if (kv.second.optional && !ctx->raw.count(kv.first)) {
        const std::string& k = kv.first;
        const bool gv = k.size() > 2 && (k.compare(k.size() - 2, 2, "_v") == 0 || k.compare(k.size() - 2, 2, "_g") == 0);
        if (!(gv && ctx->raw.count(k.substr(0, k.size() - 2)))) { all = false; break; }
      }
    if (all) {
      const PosteriorW& Q = m.enc_q;
      f.conv_plain(Q.pre, "enc_q.pre.weight", "enc_q.pre.bias");
      fill_wn(f, "enc_q.enc", Q.cond, Q.in, Q.res, Q.skip, c.posterior_layers, h, gin, fk);
      f.conv_plain(Q.proj, "enc_q.proj.weight", "enc_q.proj.bias", 0);
      m.has_vc = f.ok;
    }
  }

  f.conv_plain(m.g_pre, "dec.conv_pre.weight", "dec.conv_pre.bias");
  f.conv_plain(m.g_cond, "dec.cond.weight", "dec.cond.bias");
  const int nk = c.n_resblock_kernels;
  for (int i = 0; i < c.n_upsamples && f.ok; ++i) {
    const Conv& U = m.ups[i];
    const std::string p = "dec.ups." + std::to_string(i);
    const HostTensor* W = f.get(p + ".weight");  // folded [Cin][Cout][k], norm per input channel
    const HostTensor* B = f.get(p + ".bias");
    if (!f.ok) break;
    const float* wd = W->data.data();
    const float* bd = B->data.data();
    const int s = U.ups_s, kt = U.K, cout = U.M / s, k = kt * s;
    // out[co][s*q + r - p] = sum_ci sum_m x[ci][q - m] * w[ci][co][s*m + r]; tap = kt-1-m
    f.conv(U, [=](int row, int ci, int tap) {
      const int co = row / s, r = row % s, mm = kt - 1 - tap;
      return wd[((size_t)ci * cout + co) * k + s * mm + r];
    }, [=](int row) { return bd[row / s]; });
    if (m.has_cl) {
      // polyphase: out[s*q + r - p][co] = sum_m sum_ci x[q - m][ci] * w[ci][co][s*m + r], tap = kt-1-m
      f.clconv(m.ups_h[i], [=](int r, int co, int ci, int tap) {
        return wd[((size_t)ci * cout + co) * k + s * (kt - 1 - tap) + r];
      }, [=](int co) { return bd[co]; });
    }
    for (int j = 0; j < nk && f.ok; ++j) {
      const ResBlockW& rb = m.rbs[i * nk + j];
      const std::string q = "dec.resblocks." + std::to_string(i * nk + j);
      for (size_t d = 0; d < rb.dil.size(); ++d) {
        f.conv_plain(rb.c1[d], q + ".convs1." + std::to_string(d) + ".weight", q + ".convs1." + std::to_string(d) + ".bias");
        f.conv_plain(rb.c2[d], q + ".convs2." + std::to_string(d) + ".weight", q + ".convs2." + std::to_string(d) + ".bias");
        if (m.has_cl) {
          f.clconv_plain(rb.h1[d], q + ".convs1." + std::to_string(d) + ".weight", q + ".convs1." + std::to_string(d) + ".bias");
          f.clconv_plain(rb.h2[d], q + ".convs2." + std::to_string(d) + ".weight", q + ".convs2." + std::to_string(d) + ".bias");
        }
      }
    }
  }
  f.copy_raw(m.post_w, "dec.conv_post.weight", (size_t)m.post_c * m.post_k);
  if (f.ok)
    for (int c2 = 0; c2 < m.post_c; ++c2)
      for (int j = 0; j < m.post_k; ++j) arena[m.post_wt + (size_t)j * m.post_c + c2] = arena[m.post_w + (size_t)c2 * m.post_k + j];
  if (!f.ok) return ctx->fail(VSP_ERR_STATE, "missing weight: %s", f.missing.c_str());
  if (f.too_large > 0.f)
    return ctx->fail(VSP_ERR_UNSUPPORTED, "a convolution weight of magnitude %g (after the weight-norm fold) exceeds what the "
                                          "split-f16 matrix path represents (|w| < %g); VSP_GENERATOR=f32 / VSP_FRAME=f32 "
                                          "select the f32 matrix kernels", (double)f.too_large, 65000.0 / G16_WSCALE);
  return VSP_OK;
}

}  // namespace vsp
