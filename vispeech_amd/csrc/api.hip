// C-ABI of libvispeech_hip (include/vispeech_hip.h): context, weights, and the launch sequences
// that restate SynthesizerTrn.infer (reference models.py:672-722) as HIP kernel launches on the
// caller's stream.  No allocation and no host synchronisation happens on these paths except
// vsp_frame_lengths_host.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "model.h"

using namespace vsp;

namespace {

void generator_frame_dependence(const vsp_config& c, int pre_k, int post_k, int& back, int& fwd);   // (defined below)
long total_upsample(const vsp_config& c);

struct T3 {
  float* p = nullptr;
  long bs = 0, cs = 0;
  T3 chan(int c) const { return T3{p ? p + (size_t)c * cs : nullptr, bs, cs}; }
};

// Bump allocator over the caller's workspace; in dry mode it only measures.
struct Ws {
  char* base;
  size_t cap;
  size_t cur = 0;
  bool dry;
  bool overflow = false;
  Ws(void* b, size_t c, bool d) : base((char*)b), cap(c), dry(d) {}
  void* bytes(size_t n) {
    const size_t o = cur;
    cur += (n + 255) / 256 * 256;
    if (dry) return nullptr;
    if (cur > cap) { overflow = true; return nullptr; }
    return base + o;
  }
  float* f(size_t n) { return (float*)bytes(n * sizeof(float)); }
  T3 t3(int B, int C, int T) {
    const long ts = (T + 63) / 64 * 64;
    T3 t;
    t.p = f((size_t)B * C * ts);
    t.cs = ts;
    t.bs = (long)C * ts;
    return t;
  }
};

inline T3 ext(const float* p, int C, int T) { return T3{const_cast<float*>(p), (long)C * T, (long)T}; }

struct Run {
  vsp_ctx* ctx;
  hipStream_t s;
  Ws& ws;
  int rc = VSP_OK;
  // ragged batch of the channels-last generator (kernels.h, ClConvArgs::glen): frames per utterance of the batch chunk
  // being launched, and the columns per frame of the current stage's input / output tensors
  const int64_t* host_lengths = nullptr;   // the batch's frame counts where the host knows them (vsp_ctx::fl_known), else NULL
  const int* glen = nullptr;
  int grate_in = 0, grate_out = 0;
  // share of the padded frames the launches of the current generator chunk compute (trimmed tails): the profiled work
  // of a launch is charged for the frames it processes, not for the padded tensor (set while profiling only)
  double work_frac = 1.0;
  bool dry() const { return ws.dry; }
  const float* A(size_t off) const { return ctx->arena + off; }
  bool ok() const { return rc == VSP_OK && !ws.overflow; }
  void chk(hipError_t e, const char* what) {
    if (e != hipSuccess && rc == VSP_OK) rc = ctx->fail(VSP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  }

  static int fam(int kind, int channels) {
    int l = 0;
    while ((32 << l) < channels && l < 7) ++l;
    return kind | (l << 3);
  }
  ConvArgs args(const Conv& L, T3 x, T3 out, int T_in, int Nq) const {
    ConvArgs a;
    std::memset(&a, 0, sizeof a);
    a.x = x.p; a.x_bs = x.bs; a.x_cs = x.cs;
    a.wp = A(L.w);
    a.bias = L.b >= 0 ? A((size_t)L.b) : nullptr;
    a.out = out.p; a.o_bs = out.bs; a.o_cs = out.cs;
    a.Cin = L.Cin; a.M = L.M; a.K = L.K; a.dil = L.dil; a.pad = L.pad;
    a.T_in = T_in; a.Nq = Nq;
    a.nchunks = (L.Cin + CONV_CK - 1) / CONV_CK;
    a.alpha = 1.f; a.div = 1.f;
    a.ups_s = L.ups_s; a.ups_p = L.ups_p;
    a.f16s = L.f16s ? 1 : 0;
    a.wg = (L.has_wg && ctx->cols) ? reinterpret_cast<const uint16_t*>(A(L.wg)) : nullptr;
    a.wg_max_blocks = ctx->cols_blocks;
    a.wg_min_blocks = ctx->cols_min_blocks;
    return a;
  }
  // event pair around one launch of a profiled class; end() books the launch's algorithmic work
  bool prof_begin(int cls, int fam = VSP_FAM_OTHER) {
    if (!ctx->prof_on) return false;
    while (ctx->ev_pool.size() < ctx->ev_used + 2) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { rc = ctx->fail(VSP_ERR_HIP, "hipEventCreate"); return false; }
      ctx->ev_pool.push_back(e);
    }
    if (ctx->ev_cls.size() < ctx->ev_pool.size() / 2) {
      ctx->ev_cls.resize(ctx->ev_pool.size() / 2, 0);
      ctx->ev_fam.resize(ctx->ev_pool.size() / 2, 0);
      ctx->ev_flops.resize(ctx->ev_pool.size() / 2, 0.0);
      ctx->ev_bytes.resize(ctx->ev_pool.size() / 2, 0.0);
      ctx->ev_moved.resize(ctx->ev_pool.size() / 2, 0.0);
    }
    ctx->ev_cls[ctx->ev_used / 2] = cls;
    ctx->ev_fam[ctx->ev_used / 2] = fam;
    (void)hipEventRecord(ctx->ev_pool[ctx->ev_used], s);
    return true;
  }
  // bytes: SURVEY 8d's layer-boundary model (input + output of every CONVOLUTION the launch replaces); bytes_ext: the same
  // plus residual / accumulate reads; bytes_moved: what the launch moves through HBM as fused (each operand once)
  void prof_end(int cls, double flops, double bytes, double bytes_ext, double bytes_moved) {
    (void)hipEventRecord(ctx->ev_pool[ctx->ev_used + 1], s);
    const double f = cls == VSP_PROF_GENERATOR ? work_frac : 1.0;
    ctx->ev_flops[ctx->ev_used / 2] = flops * f;
    ctx->ev_bytes[ctx->ev_used / 2] = bytes * f;
    ctx->ev_moved[ctx->ev_used / 2] = bytes_moved * f;
    ctx->ev_used += 2;
    ctx->prof_launches[cls] += 1;
    ctx->prof_flops[cls] += flops * f;
    ctx->prof_bytes[cls] += bytes * f;
    ctx->prof_bytes_ext[cls] += bytes_ext * f;
    ctx->prof_bytes_moved[cls] += bytes_moved * f;
  }
  // fused ResBlock1 chain (g16_chain, gen16.hip): all dilation pairs of one ResBlock in one launch
  // (pairs [p0, p0 + np) of the ResBlock: the whole block by default)
  void clchain(const ResBlockW& rb, int ch, const float* x, float* out, long bs, int T, bool acc_prev, float div, int B,
               int p0 = 0, int np = -1) {
    if (dry() || !ok()) return;
    ClChainArgs a;
    std::memset(&a, 0, sizeof a);
    if (np < 0) np = (int)rb.dil.size();
    a.x = x; a.x_bs = bs; a.out = out; a.o_bs = bs;
    for (int p = 0; p < np; ++p) {
      a.w[2 * p] = reinterpret_cast<const uint16_t*>(A(rb.h1[p0 + p].wg));
      a.w[2 * p + 1] = reinterpret_cast<const uint16_t*>(A(rb.h2[p0 + p].wg));
      a.b[2 * p] = A((size_t)rb.h1[p0 + p].b); a.b[2 * p + 1] = A((size_t)rb.h2[p0 + p].b);
      a.dil[p] = rb.dil[p0 + p];
    }
    a.np = np; a.C = ch; a.K = rb.k; a.T = T;
    a.slope = 0.1f;                                    // modules.LRELU_SLOPE (reference modules.py:17)
    a.acc_prev = acc_prev ? 1 : 0; a.div = div;
    a.terms = ctx->gen_mode == 2 ? 1 : 3;
    a.ring = ctx->chain_ring ? 1 : 0;
    a.glen = glen; a.grate = grate_out;
    const bool prof = prof_begin(VSP_PROF_GENERATOR, fam(VSP_FAM_CHAIN, ch));
    chk(launch_g16_chain(a, B, s), "g16_chain");
    if (prof) {
      // the 2 np convolutions this launch replaces, each charged its input and its output
      const double el = (double)T * ch;
      prof_end(VSP_PROF_GENERATOR, np * 2.0 * 2.0 * ch * ch * rb.k * (double)T * B, 4.0 * B * el * 4.0 * np,
               4.0 * B * el * (5.0 * np + (acc_prev ? 1.0 : 0.0)), 4.0 * B * el * (2.0 + (acc_prev ? 1.0 : 0.0)));
    }
  }
  void conv(const ConvArgs& a, int B, bool generator = false) {
    if (dry() || !ok()) return;
    const int cls = generator ? VSP_PROF_GENERATOR : VSP_PROF_FRAME;
    const bool prof = prof_begin(cls, generator ? (a.ups_s > 0 ? fam(VSP_FAM_UPS, a.M / a.ups_s) : fam(VSP_FAM_PRE, a.M)) : VSP_FAM_OTHER);
    chk(launch_conv(a, B, s), "conv1d_f32_mfma");
    if (prof) {
      const double in_el = (double)a.T_in * a.Cin, out_el = (double)(a.ups_s > 0 ? a.T_store * (a.M / a.ups_s) : a.Nq * a.M);
      const double ext = 4.0 * B * (in_el + out_el * (1.0 + (a.res ? 1.0 : 0.0) + (a.acc_prev ? 1.0 : 0.0)));
      prof_end(cls, 2.0 * a.M * a.Cin * a.K * (double)a.Nq * B, 4.0 * B * (in_el + out_el), ext, ext);
    }
  }
  // channels-last split-f16 conv: x [B][T_in][Cin] -> out rows of Cout
  // x_img / o_img (round 4): the input read from / the result (also, or with out == NULL only) written as an OPERAND
  // IMAGE (kernels.h ClConvArgs): a ResBlock's intermediate lives in HBM as the next convolution's split, activated
  // window planes and reaches its LDS by LDS-DMA, without conversion arithmetic in the consumer.
  void clconv(const ClConv& L, const float* x, long x_bs, float* out, long o_bs, const float* res, long r_bs, int T_in,
              int Nq, int T_store, float in_slope, bool acc_prev, float div, int B, const uint16_t* x_img = nullptr,
              uint16_t* o_img = nullptr) {
    if (dry() || !ok()) return;
    ClConvArgs a;
    std::memset(&a, 0, sizeof a);
    a.x_img = x_img; a.xi_bs = (long)cl_img_halfs(L.Cin, T_in); a.xi_tpad = cl_img_tpad(T_in);
    a.o_img = o_img; a.oi_bs = (long)cl_img_halfs(L.Cout, T_store); a.oi_tpad = cl_img_tpad(T_store);
    a.oi_slope = 0.1f;                                 // modules.LRELU_SLOPE: what every ResBlock convolution applies to its input
    a.x = x; a.x_bs = x_bs; a.x_ts = L.Cin;
    a.wh = reinterpret_cast<const uint16_t*>(A(L.wg));
    a.bias = A((size_t)L.b);
    a.out = out; a.o_bs = o_bs; a.o_ts = L.Cout;
    a.res = res; a.r_bs = r_bs; a.r_ts = L.Cout;
    a.Cin = L.Cin; a.Cout = L.Cout; a.K = L.K; a.dil = L.dil; a.pad = L.pad;
    a.T_in = T_in; a.Nq = Nq;
    a.in_act = 1; a.in_slope = in_slope;
    a.acc_prev = acc_prev ? 1 : 0; a.div = div;
    a.phases = L.phases; a.ups_p = L.ups_p; a.T_store = T_store;
    a.terms = ctx->gen_mode == 2 ? 1 : 3;
    a.glen = glen; a.g_in = L.phases > 1 ? grate_in : grate_out; a.g_store = grate_out;
    const bool prof = prof_begin(VSP_PROF_GENERATOR, fam(L.phases > 1 ? VSP_FAM_UPS : VSP_FAM_CONV, L.Cout));
    chk(launch_g16_conv(a, B, s), "g16_conv");
    if (prof) {
      // input once + output once; the residual / accumulate reads go to bytes_ext
      const double in_el = (double)T_in * L.Cin, out_el = (double)T_store * L.Cout;
      const double ext = 4.0 * B * (in_el + out_el * (1.0 + (res ? 1.0 : 0.0) + (acc_prev ? 1.0 : 0.0)));
      prof_end(VSP_PROF_GENERATOR, 2.0 * L.Cout * L.Cin * L.K * L.phases * (double)Nq * B, 4.0 * B * (in_el + out_el), ext,
               ext + (out && o_img ? 4.0 * B * out_el : 0.0));
    }
  }
  // fused ResBlock1 pair (g16_pair, gen16.hip): out = x + conv2(lrelu(conv1(lrelu(x)))) [+ out] [/ div]
  void clpair(const ClConv& L1, const ClConv& L2, const float* x, float* out, long bs, int T, bool acc_prev, float div,
              int B) {
    if (dry() || !ok()) return;
    ClPairArgs a;
    std::memset(&a, 0, sizeof a);
    a.x = x; a.x_bs = bs; a.out = out; a.o_bs = bs;
    a.w1h = reinterpret_cast<const uint16_t*>(A(L1.wg));
    a.w2h = reinterpret_cast<const uint16_t*>(A(L2.wg));
    a.b1 = A((size_t)L1.b); a.b2 = A((size_t)L2.b);
    a.C = L1.Cout; a.K = L1.K; a.dil = L1.dil; a.T = T;
    a.slope = 0.1f;                                    // modules.LRELU_SLOPE (reference modules.py:17)
    a.acc_prev = acc_prev ? 1 : 0; a.div = div;
    a.terms = ctx->gen_mode == 2 ? 1 : 3;
    a.ring = ctx->pair_ring ? 1 : 0;
    a.rw64 = ctx->rw64 ? 1 : 0;
    a.glen = glen; a.grate = grate_out;
    const bool prof = prof_begin(VSP_PROF_GENERATOR, fam(VSP_FAM_PAIR, L1.Cout));
    chk(launch_g16_pair(a, B, s), "g16_pair");
    if (prof) {
      // the two convolutions this launch replaces: SURVEY.md 8d charges each its input and its output (4 passes of
      // T x C); with conv2's residual read (and the accumulate read of a ResBlock's last pair): 5 (6) -> bytes_ext
      const double el = (double)T * L1.Cout;
      prof_end(VSP_PROF_GENERATOR, 2.0 * 2.0 * L1.Cout * L1.Cin * L1.K * (double)T * B, 4.0 * B * el * 4.0,
               4.0 * B * el * (5.0 + (acc_prev ? 1.0 : 0.0)), 4.0 * B * el * (2.0 + (acc_prev ? 1.0 : 0.0)));
    }
  }
  // cond(g): 1x1 conv on g [B][gin] (T = 1) -> out [B][M]
  void cond(const Conv& L, const float* g, float* out, int B) {
    const int gin = L.Cin;
    ConvArgs a = args(L, T3{const_cast<float*>(g), (long)gin, 1}, T3{out, (long)L.M, 1}, 1, 1);
    conv(a, B);
  }
  void ln(T3 x, T3 res, size_t gamma, size_t beta, T3 y, int B, int C, int T) {
    if (dry() || !ok()) return;
    chk(launch_layernorm(x.p, x.bs, x.cs, res.p, res.bs, res.cs, A(gamma), A(beta), y.p, y.bs, y.cs, B, C, T, s),
        "layernorm");
  }
};

// attentions.Encoder.forward (reference attentions.py:35-47) including the x * x_mask the reference applies at its exit
// (:46).  x_in must already be masked by the caller's semantics (every reference call site passes x * x_mask); it is
// read, never written: layer 0 takes it as its input and residual directly (round 5: no entry copy), and the last
// layer's second LayerNorm writes y_out, masked in place behind it (no exit copy; a mask folded into the LayerNorm
// kernel's store was measured: its scalar spills made EVERY LayerNorm launch 9 us slower).
void run_encoder_masked(Run& r, const EncoderW& E, int B, int T, T3 x_in, const int64_t* lengths, T3 y_out) {
  const vsp_config& c = r.ctx->cfg;
  const int h = c.hidden_channels, f = c.filter_channels;
  T3 X = r.ws.t3(B, h, T), S = r.ws.t3(B, h, T), QKV = r.ws.t3(B, 3 * h, T), AT = r.ws.t3(B, h, T),
     FF = r.ws.t3(B, f, T);
  void* AP = r.ctx->att_f16s ? r.ws.bytes(3 * attn_pack_bytes(B, c.n_heads, h / c.n_heads, T)) : nullptr;   // packed q | k | v images
  if (E.layers.empty()) {                       // (no layer: y = x * mask)
    if (!r.dry() && r.ok()) {
      r.chk(launch_copy3(x_in.p, x_in.bs, x_in.cs, y_out.p, y_out.bs, y_out.cs, B, h, T, r.s), "copy");
      if (lengths) r.chk(launch_mask3(y_out.p, y_out.bs, y_out.cs, lengths, B, h, T, r.s), "mask");
    }
    return;
  }
  // round 6, the column-tile kernels (conv_cols.h): the projections write the packed attention operands themselves
  // (no [B][3h][T] tensor, no pack launch) and conv_o normalises in its own launch (no LayerNorm launch): 8 -> 6 launches
  // per layer.  VSP_COLS=0: the separate launches (second implementation, tests/test_hip_parity.py).
  // (not for a handful of column tiles -- one utterance --: a column-tile block is one chain of round trips of 10+ us, the
  // row-tiled kernels' blocks are shorter than the launch they save: measured 3.12 against 3.14 ms for one utterance)
  const long col_tiles = (long)B * ((T + 63) / 64);
  const bool cols_size = col_tiles >= r.ctx->cols_min_blocks;
  const bool fuse_qkv = r.ctx->cols && cols_size && r.ctx->att_f16s && r.ctx->frame_f16s && E.layers[0].qkv.has_wg &&
                        E.layers[0].qkv.Cin == h && attn_qkv_pack_supported(h, c.n_heads);
  const bool fuse_ln = r.ctx->cols && cols_size && r.ctx->frame_f16s && E.layers[0].o.has_wg && h == 192;
  for (size_t i = 0; i < E.layers.size(); ++i) {
    const EncLayer& L = E.layers[i];
    const T3 Xi = i == 0 ? x_in : X;            // this layer's input
    const bool last = i + 1 == E.layers.size();
    ConvArgs a = r.args(L.qkv, Xi, QKV, T, T);
    a.lengths = lengths; a.in_mask = 1;  // x * x_mask feeds the attention (attentions.py:38)
    if (!fuse_qkv) r.conv(a, B);
    else if (!r.dry() && r.ok()) {
      const bool prof = r.prof_begin(VSP_PROF_FRAME);
      r.chk(launch_attn_qkv_pack_f16s(Xi.p, Xi.bs, Xi.cs, a.wg, a.bias, lengths, B, h, c.n_heads, T, AP, r.s), "q | k | v + pack");
      if (prof) {
        const double in_el = (double)T * h, out_el = (double)T * 3 * h;
        r.prof_end(VSP_PROF_FRAME, 2.0 * 3 * h * h * (double)T * B, 4.0 * B * (in_el + out_el), 4.0 * B * (in_el + out_el),
                   4.0 * B * (in_el + out_el));
      }
    }
    if (!r.dry() && r.ok()) {
      const bool prof = r.prof_begin(VSP_PROF_ATTENTION);
      if (r.ctx->att_f16s)
        r.chk(launch_attention_f16s(fuse_qkv ? nullptr : QKV.p, QKV.bs, QKV.cs, r.A(L.ek), r.A(L.ev), lengths, AT.p, AT.bs, AT.cs, B, h,
                                    c.n_heads, T, c.window_size, AP, r.s), "attention (split f16)");
      else
        r.chk(launch_attention(QKV.p, QKV.bs, QKV.cs, r.A(L.ek), r.A(L.ev), lengths, AT.p, AT.bs, AT.cs, B, h,
                               c.n_heads, T, c.window_size, r.ctx->att_ksplit, r.s), "attention");
      // QK^T and PV: 2 * h * T^2 MAC per utterance; banded relative logits and values: 2 * h * T * (2w+1) MAC
      if (prof) r.prof_end(VSP_PROF_ATTENTION, (double)B * (4.0 * h * (double)T * T + 4.0 * h * (double)T * (2 * c.window_size + 1)),
                           4.0 * B * 4.0 * h * (double)T, 4.0 * B * 4.0 * h * (double)T, 4.0 * B * 4.0 * h * (double)T);
    }
    // S = x + conv_o(att);  X = LayerNorm(S)
    a = r.args(L.o, AT, S, T, T);
    a.res = Xi.p; a.r_bs = Xi.bs; a.r_cs = Xi.cs;
    if (fuse_ln && a.wg) {
      // one launch: the block holds every channel of its columns (Xi may be X: a block reads and writes its own columns only)
      a.out = X.p; a.o_bs = X.bs; a.o_cs = X.cs;
      if (!r.dry() && r.ok()) {
        const bool prof = r.prof_begin(VSP_PROF_FRAME);
        r.chk(launch_conv_cols(a, B, r.s, r.A(L.g1), r.A(L.b1)), "conv_o + LayerNorm");
        if (prof) {
          const double el = (double)T * h;
          r.prof_end(VSP_PROF_FRAME, 2.0 * h * h * (double)T * B, 4.0 * B * 2.0 * el, 4.0 * B * 3.0 * el, 4.0 * B * 3.0 * el);
        }
      }
    } else {
      r.conv(a, B);
      r.ln(S, T3{}, L.g1, L.b1, X, B, h, T);
    }
    // FFN (attentions.py:277-285)
    a = r.args(L.f1, X, FF, T, T);
    a.lengths = lengths; a.in_mask = 1; a.act = 1;
    r.conv(a, B);
    a = r.args(L.f2, FF, S, T, T);
    a.lengths = lengths; a.in_mask = 1; a.mask_pre = 1;
    a.res = X.p; a.r_bs = X.bs; a.r_cs = X.cs;
    r.conv(a, B);
    r.ln(S, T3{}, L.g2, L.b2, last ? y_out : X, B, h, T);
  }
  // y = x * mask (attentions.py:46)
  if (lengths && !r.dry() && r.ok()) r.chk(launch_mask3(y_out.p, y_out.bs, y_out.cs, lengths, B, h, T, r.s), "mask");
}

void mask3(Run& r, T3 x, const int64_t* lengths, int B, int C, int T) {
  if (r.dry() || !r.ok() || !lengths) return;
  r.chk(launch_mask3(x.p, x.bs, x.cs, lengths, B, C, T, r.s), "mask");
}

// modules.WN.forward (reference modules.py:148-176; dilation_rate 1) on H [B][h][T]: H is the running
// residual stream (destroyed), OUT receives the masked skip sum.  gc: nl * 2h conditioning rows.
// cond != null: cond_layer(g) is evaluated here into gc [B][nl * 2h]; null: gc already holds it, batch stride gc_bs.
void run_wn(Run& r, const Conv* cond, const std::vector<Conv>& in, const std::vector<Conv>& res,
            const std::vector<Conv>& skip, int nl, int B, int T, T3 H, T3 ACT, T3 OUT, float* gc, long gc_bs, const float* g,
            const int64_t* lengths) {
  const int h = r.ctx->cfg.hidden_channels;
  if (cond) r.cond(*cond, g, gc, B);
  for (int l = 0; l < nl; ++l) {
    ConvArgs a = r.args(in[l], H, ACT, T, T);
    a.act = 2; a.cond = gc ? gc + (size_t)l * 2 * h : nullptr; a.cond_bs = gc_bs;
    r.conv(a, B);
    if (l < nl - 1) {
      // res_skip_layers[l] (modules.py:165-172) as one launch with two destinations: rows [0, h) are the in-place
      // residual update H = (H + res) * mask, rows [h, 2 h) accumulate into the skip sum
      a = r.args(res[l], ACT, H, T, T);
      a.res = H.p; a.r_bs = H.bs; a.r_cs = H.cs;
      a.lengths = lengths; a.mask_post = 1;
      a.split_row = h; a.out2 = OUT.p; a.o2_bs = OUT.bs; a.o2_cs = OUT.cs; a.acc_prev2 = l > 0;
      r.conv(a, B);
    } else {
      a = r.args(skip[l], ACT, OUT, T, T);
      a.acc_prev = l > 0;
      a.lengths = lengths; a.mask_post = 1;
      r.conv(a, B);
    }
  }
}

// ResidualCouplingBlock.forward in place on z [B][inter][T] (reference models.py:202-209,
// modules.py:324-343, 148-176).  reverse: x1 = (x1 - m) * mask, layers n-1 .. 0 (infer);
// forward: x1 = m + x1 * mask, layers 0 .. n-1 (voice conversion).  With an even number of flows
// layer i sees the same channel flip in both directions (i and n - i flips).
void run_flow(Run& r, int B, int T, T3 z, const float* g, const int64_t* lengths, bool reverse = true) {
  const vsp_config& c = r.ctx->cfg;
  const Model& m = r.ctx->model;
  const int h = c.hidden_channels, half = c.inter_channels / 2, fl = c.flow_layers;
  T3 H = r.ws.t3(B, h, T), ACT = r.ws.t3(B, h, T), OUT = r.ws.t3(B, h, T);
  // cond_layer(g) of all coupling layers: one launch (reference modules.py:153-155, once per WN.forward)
  const long gcs = (long)c.n_flows * 2 * h * fl;
  float* gc = r.ws.f((size_t)B * gcs);
  r.cond(m.flow_cond_all, g, gc, B);
  for (int n = 0; n < c.n_flows; ++n) {
    const int i = reverse ? c.n_flows - 1 - n : n;
    const FlowW& F = m.flows[i];
    const T3 x0 = F.flipped ? z.chan(half) : z;
    const T3 x1 = F.flipped ? z : z.chan(half);
    ConvArgs a = r.args(F.pre, x0, H, T, T);
    a.lengths = lengths; a.mask_post = 1;
    r.conv(a, B);
    run_wn(r, nullptr, F.in, F.res, F.skip, fl, B, T, H, ACT, OUT, gc ? gc + (size_t)i * 2 * h * fl : nullptr, gcs, g, lengths);
    // m = post(out) * mask ; x1 = (x1 -/+ m) * mask
    a = r.args(F.post, OUT, x1, T, T);
    a.lengths = lengths; a.mask_pre = 1; a.alpha = reverse ? -1.f : 1.f;
    a.res = x1.p; a.r_bs = x1.bs; a.r_cs = x1.cs;
    a.mask_post = 1;
    r.conv(a, B);
  }
}

// PosteriorEncoder.forward (reference models.py:233-241): y [B][spec][T] -> m, logs, z [B][inter][T].
void run_posterior(Run& r, int B, int T, T3 y, const int64_t* lengths, const float* g, const float* noise, T3 Z,
                   T3 M, T3 LOGS) {
  const vsp_config& c = r.ctx->cfg;
  const PosteriorW& Q = r.ctx->model.enc_q;
  const int h = c.hidden_channels, ql = c.posterior_layers;
  T3 H = r.ws.t3(B, h, T), ACT = r.ws.t3(B, h, T), OUT = r.ws.t3(B, h, T);
  float* gc = r.ws.f((size_t)B * 2 * h * ql);
  if (r.dry()) return;
  ConvArgs a = r.args(Q.pre, y, H, T, T);
  a.lengths = lengths; a.mask_post = 1;
  r.conv(a, B);
  run_wn(r, &Q.cond, Q.in, Q.res, Q.skip, ql, B, T, H, ACT, OUT, gc, 2L * h * ql, g, lengths);
  // proj: m and logs rows in one launch, two destinations (reference models.py:238-239: stats = proj(x) * mask, split)
  a = r.args(Q.proj, OUT, M, T, T);
  a.lengths = lengths; a.mask_post = 1;
  a.split_row = c.inter_channels; a.out2 = LOGS.p; a.o2_bs = LOGS.bs; a.o2_cs = LOGS.cs; a.mask_post2 = 1;
  r.conv(a, B);
  if (r.ok()) {
    // z = (m + eps * exp(logs)) * mask   (contiguous [B][inter][T] outputs)
    r.chk(launch_reparam(M.p, LOGS.p, noise, 1.f, Z.p, (long)B * c.inter_channels * T, r.s), "reparam");
    r.chk(launch_mask3(Z.p, Z.bs, Z.cs, lengths, B, c.inter_channels, T, r.s), "mask");
  }
}

// Generator.forward (reference models.py:271-290).  z [B][inter][T]; in_lengths != null applies
// the (z * x_mask) of models.py:720 while staging conv_pre's input.
void run_generator(Run& r, int B, int T, T3 z, const int64_t* in_lengths, const float* g, float* o) {
  const vsp_config& c = r.ctx->cfg;
  const Model& m = r.ctx->model;
  const int c0 = c.upsample_initial_channel, nk = c.n_resblock_kernels;
  float* gc = r.ws.f((size_t)B * c0);
  r.cond(m.g_cond, g, gc, B);
  // buffer sizes: max over stages of C * T
  size_t mx = (size_t)c0 * ((T + 63) / 64 * 64);
  {
    long t = T;
    for (int i = 0; i < c.n_upsamples; ++i) {
      t *= c.upsample_rates[i];
      mx = std::max(mx, (size_t)(c0 >> (i + 1)) * (size_t)((t + 63) / 64 * 64));
    }
  }
  float* buf[5];
  for (auto& b : buf) b = r.ws.f((size_t)B * mx);
  auto view = [&](int k, int C, long Tn) {
    const long ts = (Tn + 63) / 64 * 64;
    return T3{buf[k], (long)C * ts, ts};
  };
  int cur = 0;  // buffer holding the stage input
  T3 X = view(cur, c0, T);
  ConvArgs a = r.args(m.g_pre, z, X, T, T);
  a.lengths = in_lengths; a.in_mask = in_lengths ? 1 : 0;
  a.cond = gc; a.cond_bs = c0;
  r.conv(a, B, true);
  long Tn = T;
  int ch = c0;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const Conv& U = m.ups[i];
    const long Tout = Tn * U.ups_s;
    ch = c0 >> (i + 1);
    // free buffers: all but `cur`
    int fb[4], nf = 0;
    for (int k = 0; k < 5; ++k) if (k != cur) fb[nf++] = k;
    T3 XU = view(fb[0], ch, Tout), T1 = view(fb[1], ch, Tout), YA = view(fb[2], ch, Tout), XS = view(fb[3], ch, Tout);
    a = r.args(U, X, XU, (int)Tn, (int)Tn + 1);
    a.in_act = 1; a.in_slope = 0.1f;
    a.T_store = (int)Tout;
    r.conv(a, B, true);
    for (int j = 0; j < nk; ++j) {
      const ResBlockW& rb = m.rbs[i * nk + j];
      const int nd = (int)rb.dil.size();
      for (int d = 0; d < nd; ++d) {
        const T3 yin = d == 0 ? XU : YA;
        a = r.args(rb.c1[d], yin, T1, (int)Tout, (int)Tout);
        a.in_act = 1; a.in_slope = 0.1f;
        r.conv(a, B, true);
        const bool last = d == nd - 1;
        a = r.args(rb.c2[d], T1, last ? XS : YA, (int)Tout, (int)Tout);
        a.in_act = 1; a.in_slope = 0.1f;
        a.res = yin.p; a.r_bs = yin.bs; a.r_cs = yin.cs;
        if (last) {
          a.acc_prev = j > 0;
          if (j == nk - 1) a.div = (float)nk;
        }
        r.conv(a, B, true);
      }
    }
    cur = fb[3];
    X = XS;
    Tn = Tout;
  }
  if (!r.dry() && r.ok())
    r.chk(launch_conv_post(X.p, X.bs, X.cs, r.A(m.post_w), m.post_c, m.post_k, 0.01f, o, Tn, B, (int)Tn, r.s, r.ctx->flags_dev),
          "conv_post");
}

// Generator.forward on the split-f16 channels-last kernels (gen16.hip): conv_pre stays on the
// f32 kernel (input z is channel-major and tiny), its output is transposed once to [B][T][C].
// The context's two side streams and n fork / join events (created on first use, on the device the caller's stream
// belongs to; destroyed with the context).
bool ensure_side_streams(Run& r, size_t n_events) {
  vsp_ctx* ctx = r.ctx;
  // side[0]: the device's greatest stream priority, side[1]: its least (the caller's stream is assumed to sit between)
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  for (int k = 0; k < 2; ++k)
    if (!ctx->side[k] &&
        hipStreamCreateWithPriority(&ctx->side[k], hipStreamNonBlocking, k == 0 ? greatest : least) != hipSuccess) {
      ctx->side[k] = nullptr;
      return false;
    }
  while (ctx->sync_ev.size() < n_events) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return false;
    ctx->sync_ev.push_back(e);
  }
  return true;
}

void run_generator_cl(Run& r, int B, int T, T3 z, const int64_t* in_lengths, const float* g, float* o) {
  const vsp_config& c = r.ctx->cfg;
  const Model& m = r.ctx->model;
  const int c0 = c.upsample_initial_channel, nk = c.n_resblock_kernels;
  float* gc = r.ws.f((size_t)B * c0);
  r.cond(m.g_cond, g, gc, B);
  T3 X0 = r.ws.t3(B, c0, T);
  size_t mx = (size_t)c0 * T;
  {
    long t = T;
    for (int i = 0; i < c.n_upsamples; ++i) {
      t *= c.upsample_rates[i];
      mx = std::max(mx, (size_t)(c0 >> (i + 1)) * (size_t)t);
    }
  }
  float* buf[5];
  for (auto& bptr : buf) bptr = r.ws.f((size_t)B * mx);
  // ResBlocks 1 .. nk-1 of a stage on side streams (vsp_ctx::rb_streams): their own ping-pong tensors (and operand image)
  std::vector<float*> side_buf;
  if (r.ctx->rb_streams)
    for (int j = 1; j < nk; ++j)
      for (int u = 0; u < 2; ++u) side_buf.push_back(r.ws.f((size_t)B * mx));
  // the operand image of a ResBlock intermediate on the stages that run one launch per convolution (>= 128 channels)
  size_t mx_img = 0;
  {
    long t = T;
    for (int i = 0; i < c.n_upsamples; ++i) {
      t *= c.upsample_rates[i];
      const int chi = c0 >> (i + 1);
      if (chi >= 128 && chi % 32 == 0) mx_img = std::max(mx_img, cl_img_halfs(chi, (int)t) / 2);
    }
  }
  const bool want_img = r.ctx->t_img && mx_img && r.ctx->gen_mode == 1;
  uint16_t* timg = want_img ? reinterpret_cast<uint16_t*>(r.ws.f((size_t)B * mx_img)) : nullptr;
  std::vector<uint16_t*> side_img;
  if (want_img && r.ctx->rb_streams)
    for (int j = 1; j < nk; ++j) side_img.push_back(reinterpret_cast<uint16_t*>(r.ws.f((size_t)B * mx_img)));
  // Trimmed tails (round 5; VSP_TRIM_TAILS=0: off, second implementation, bit-identical).  Behind an utterance's last frame
  // the input is exactly zero (z * x_mask, reference models.py:720), so the output there depends on the distance to the
  // utterance's end and to the tensor's end only: every kernel below treats utterance b's tensor as ENDING after
  // len_b + back + 1 + fwd frames (kernels.h, ClConvArgs::glen) -- frames [0, len + back) come out as in the padded run,
  // frame len + back is the steady state (periodic in one frame), the last fwd frames are the tensor end's -- and
  // gen_tail_fill writes the rest of the padded tensor from those.  The reference's full padded output, bit for bit.
  // (back / fwd: output frame F depends on input frames [F - back, F + fwd], sample-exact: 13 / 13 for configs/config.json.
  // A tensor that ends after E frames is exact up to frame E - 1 - fwd; frame len + back is the first steady one.)
  int back = 0, fwd = 0;
  generator_frame_dependence(c, m.g_pre.K, m.post_k, back, fwd);
  int* glen_all = r.ctx->trim_tails ? reinterpret_cast<int*>(r.ws.bytes((size_t)B * sizeof(int))) : nullptr;
  // (gen_tail_fill keeps the computed tensor end -- fwd frames of the waveform -- in 64 KB of LDS: other configurations run untrimmed)
  bool trim = glen_all && in_lengths && T > back + fwd + 1 && (size_t)fwd * total_upsample(c) * sizeof(float) <= 64 * 1024;
  if (trim && r.host_lengths) {
    // the host knows the frame counts (vsp_frame_lengths_host of this batch): when no utterance ends early enough to be
    // trimmed (a uniform batch, one long utterance) the kernels run WITHOUT per-utterance extents -- same output, and
    // the persistent pipelined g16_convp serves the uniform tiles again
    bool any = false;
    for (int b = 0; b < B; ++b) any = any || std::max<int64_t>(r.host_lengths[b], 0) + back + 1 + fwd < T;
    trim = any;
  }
  if (trim && !r.dry() && r.ok()) r.chk(launch_gen_plan(in_lengths, B, T, back, fwd, glen_all, r.s), "gen_plan");
  // profiled (untimed) passes charge every launch the frames it COMPUTES: read the plan back (a host wait -- profiling only)
  std::vector<int> glen_host;
  if (trim && r.ctx->prof_on && !r.dry() && r.ok()) {
    glen_host.resize(B);
    hipError_t e = hipMemcpyAsync(glen_host.data(), glen_all, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, r.s);
    if (e == hipSuccess) e = hipStreamSynchronize(r.s);
    r.chk(e, "gen_plan read-back (profiling)");
  }
  // the channels-last kernels address one utterance's tensor with 32-bit byte offsets (buffer descriptors)
  // (the operand images -- C * 4 * (T + 384) bytes per utterance -- are addressed the same way)
  if (mx * sizeof(float) >= (size_t)1 << 31 || (timg && mx_img * sizeof(float) >= (size_t)1 << 31)) {
    if (r.rc == VSP_OK)
      r.rc = r.ctx->fail(VSP_ERR_UNSUPPORTED, "generator: %d frames per call exceed the 2 GiB per-utterance activation "
                                              "limit; synthesise in chunks (Engine.generator_stream)", T);
    return;
  }
  ConvArgs a = r.args(m.g_pre, z, X0, T, T);
  a.lengths = in_lengths; a.in_mask = in_lengths ? 1 : 0;
  a.cond = gc; a.cond_bs = c0;
  r.conv(a, B, true);
  int cur = 0;
  if (!r.dry() && r.ok())
    r.chk(launch_transpose_ct(X0.p, X0.bs, X0.cs, buf[cur], (long)T * c0, c0, B, c0, T, r.s, r.ctx->act_scale), "transpose");
  long Tn = T;
  int ch = c0;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const ClConv& U = m.ups_h[i];
    const long Tout = Tn * U.phases;
    const int cin = ch;
    ch = c0 >> (i + 1);
    int fb[4], nf = 0;
    for (int k = 0; k < 5; ++k) if (k != cur) fb[nf++] = k;
    float *XU = buf[fb[0]], *T1 = buf[fb[1]], *YA = buf[fb[2]], *XS = buf[fb[3]];
    const long bs = Tout * ch, xbs = Tn * cin;
    // Optional cache blocking over the batch (VSP_CHUNK_MB, default off): a stage runs chunk by chunk
    // so that the tensors two consecutive launches exchange are <= chunk_mb and could be served by the
    // 256 MiB Infinity Cache.  Measured on MI355X (profiles/r01_tile_experiments.txt): 32-192 MiB
    // chunks are 3-10 % SLOWER than whole-batch launches (more launches and tails, no shorter kernels),
    // so the product runs whole-batch; the knob stays for experiments.
    int Bc = B;
    if (r.ctx->chunk_mb > 0) {
      const double per_utt_mb = (double)bs * 4.0 / (1024.0 * 1024.0);
      Bc = (int)std::max(1.0, std::floor(r.ctx->chunk_mb / per_utt_mb));
      const long tiles_per_utt = (Tout + 255) / 256 * std::max(1, ch / 128);
      const int min_b = (int)((2 * 256 * (ch >= 128 ? 1 : 2) + tiles_per_utt - 1) / tiles_per_utt);
      Bc = std::min(B, std::max(Bc, min_b));
    }
    for (int b0 = 0; b0 < B; b0 += Bc) {
      const int nb = std::min(Bc, B - b0);
      r.glen = trim ? glen_all + b0 : nullptr;
      r.grate_in = (int)(Tn / T); r.grate_out = (int)(Tout / T);
      r.work_frac = 1.0;
      if (!glen_host.empty() && r.ok()) {
        double fr = 0.0;
        for (int b = b0; b < b0 + nb; ++b) fr += glen_host[b];
        r.work_frac = fr / ((double)nb * T);
      }
      const float* xin = buf[cur] + (size_t)b0 * xbs;
      float *xu = XU + (size_t)b0 * bs, *xs = XS + (size_t)b0 * bs;
      r.clconv(U, xin, xbs, xu, bs, nullptr, 0, (int)Tn, (int)Tn + 1, (int)Tout, 0.1f, false, 1.f, nb);
      // The stage's ResBlocks are independent chains until their sum (reference models.py:276-285): chain 0 runs on the
      // caller's stream, chains 1 .. on the context's side streams with their own intermediates, forked after the
      // up-convolution; a chain's LAST launch accumulates into xs and therefore waits for the previous chain's last
      // launch (the reference's order of the sum, bit for bit), the caller's stream joins after the last chain.
      const bool conc = ((r.ctx->rb_streams >> i) & 1) && !r.ctx->prof_on && !r.dry() && r.ok() && Bc == B && nk > 1 &&
                        (int)side_buf.size() == 2 * (nk - 1) && ensure_side_streams(r, (size_t)c.n_upsamples * (nk + 1));
      hipStream_t const main_s = r.s;
      hipEvent_t* const ev = conc ? r.ctx->sync_ev.data() + (size_t)i * (nk + 1) : nullptr;   // [0] fork, [1 + j] chain j's last launch
      // stages that run one launch per convolution hand the pair's intermediate over as an operand image
      const bool use_img = timg && ch >= 128 && ch % 32 == 0;
      auto chain_img = [&](int j) -> uint16_t* {
        if (!use_img) return nullptr;
        uint16_t* base = (conc && j != 1 && (int)side_img.size() == nk - 1) ? side_img[j == 0 ? 0 : j - 1] : timg;
        return base + (size_t)b0 * cl_img_halfs(ch, (int)Tout);
      };
      if (use_img && !r.dry() && r.ok())
        for (int j = (conc && !side_img.empty()) ? 0 : 1; j < ((conc && !side_img.empty()) ? nk : 2); ++j)
          r.chk(launch_cl_img_zero_pads(chain_img(j), nb, ch, (int)Tout, r.s, r.glen, r.grate_out), "cl_img_zero_pads");
      if (conc) r.chk(hipEventRecord(ev[0], main_s), "fork event");
      for (int j = 0; j < nk; ++j) {
        const ResBlockW& rb = m.rbs[i * nk + j];
        const int nd = (int)rb.dil.size();
        // chain j's stream and intermediates: chain 0 on the high-priority side stream, chain 1 on the caller's, the rest
        // on the low-priority one -- the hardware then dispatches a chain's blocks into the slots the chains before it
        // leave free (ramps, partial last rounds) instead of sharing the CUs launch by launch in lockstep, and a chain's
        // last launch finds the previous chain's long finished
        float *t1 = T1 + (size_t)b0 * bs, *ya = YA + (size_t)b0 * bs;
        const int side_set = j == 0 ? 0 : j - 1;           // (chain 1 keeps the caller-stream tensors)
        if (conc && j != 1) {
          r.s = r.ctx->side[j == 0 ? 0 : 1];
          t1 = side_buf[2 * side_set] + (size_t)b0 * bs;
          ya = side_buf[2 * side_set + 1] + (size_t)b0 * bs;
          r.chk(hipStreamWaitEvent(r.s, ev[0], 0), "fork wait");
        }
        uint16_t* const ti = chain_img(j);
        auto before_last = [&]() { if (conc && j > 0) r.chk(hipStreamWaitEvent(r.s, ev[j], 0), "sum order wait"); };
        auto after_last = [&]() { if (conc) r.chk(hipEventRecord(ev[1 + j], r.s), "chain end event"); r.s = main_s; };
        bool fuse = r.ctx->fuse_pairs;
        const int terms = r.ctx->gen_mode == 2 ? 1 : 3;
        for (int d = 0; d < nd; ++d)
          fuse = fuse && (g16_pair_supported(ch, rb.k, rb.dil[d]) || (r.ctx->pp_pairs && g16_pp_supported(ch, rb.k, rb.dil[d], terms)));
        const int kbit = rb.k <= 3 ? 1 : rb.k <= 7 ? 2 : 4;
        if (fuse && (r.ctx->chain_mask & kbit) && ch <= r.ctx->chain_ch && nd <= 3 &&
            g16_chain_supported(ch, rb.k, rb.dil.data(), nd)) {
          before_last();
          r.clchain(rb, ch, xu, xs, bs, (int)Tout, j > 0, j == nk - 1 ? (float)nk : 1.f, nb);
          after_last();
          continue;
        }
        for (int d = 0; d < nd; ++d) {
          const bool last = d == nd - 1;
          const float div = (last && j == nk - 1) ? (float)nk : 1.f;
          if (fuse) {
            // one launch per pair; the running y ping-pongs between ya and t1 (a block reads halo rows
            // that a neighbour writes, so a pair cannot run in place)
            const float* yin = d == 0 ? xu : ((d & 1) ? ya : t1);
            float* yout = last ? xs : ((d & 1) ? t1 : ya);
            if (last) before_last();
            r.clpair(rb.h1[d], rb.h2[d], yin, yout, bs, (int)Tout, last && j > 0, div, nb);
          } else if (ch == 128 && (r.ctx->chain128_mask & kbit) && g16_chain_supported(ch, rb.k, &rb.dil[d], 1)) {
            // 128-channel pair as ONE launch (g16_chain, 128-column blocks): the intermediate never reaches HBM
            const float* yin = d == 0 ? xu : ((d & 1) ? ya : t1);
            float* yout = last ? xs : ((d & 1) ? t1 : ya);
            if (last) before_last();
            r.clchain(rb, ch, yin, yout, bs, (int)Tout, last && j > 0, div, nb, d, 1);
          } else {
            const float* yin = d == 0 ? xu : ya;
            const bool img = ti != nullptr && rb.k >= 3;
            r.clconv(rb.h1[d], yin, bs, img ? nullptr : t1, bs, nullptr, 0, (int)Tout, (int)Tout, (int)Tout, 0.1f, false, 1.f,
                     nb, nullptr, img ? ti : nullptr);
            if (last) before_last();
            r.clconv(rb.h2[d], t1, bs, last ? xs : ya, bs, yin, bs, (int)Tout, (int)Tout, (int)Tout, 0.1f,
                     last && j > 0, div, nb, img ? ti : nullptr, nullptr);
          }
        }
        after_last();
      }
      r.s = main_s;
      if (conc) r.chk(hipStreamWaitEvent(main_s, ev[nk], 0), "join wait");
    }
    cur = fb[3];
    Tn = Tout;
  }
  r.glen = nullptr;
  r.work_frac = 1.0;
  if (!r.dry() && r.ok()) {
    r.chk(launch_conv_post_cl(buf[cur], Tn * ch, ch, r.A(m.post_wt), m.post_c, m.post_k, 0.01f, o, Tn, B, (int)Tn, r.s,
                              trim ? glen_all : nullptr, (int)(Tn / T), 1.f / r.ctx->act_scale, r.ctx->flags_dev), "conv_post_cl");
    if (trim) r.chk(launch_gen_tail_fill(o, Tn, in_lengths, glen_all, B, T, back, fwd, (int)(Tn / T), r.s), "gen_tail_fill");
  }
}

void run_gen(Run& r, int B, int T, T3 z, const int64_t* in_lengths, const float* g, float* o) {
  if (r.ctx->gen_mode >= 1 && r.ctx->model.has_cl) run_generator_cl(r, B, T, z, in_lengths, g, o);
  else run_generator(r, B, T, z, in_lengths, g, o);
}

int check_ready(vsp_ctx* ctx) {
  if (!ctx) return VSP_ERR_ARG;
  if (!ctx->ready) return ctx->fail(VSP_ERR_STATE, "weights not finalised");
  return VSP_OK;
}

// Exact dependence of the generator's output FRAMES on its input frames, from the configuration (sample-exact supports;
// vsp_generator_halo_frames below is the coarser per-stage bound the streamed vocoder uses): output frame F depends on
// input frames [F - back, F + fwd].  An impulse at input position 0 reaches output samples [lo, hi]: conv_pre widens the
// support by its padding, a transposed convolution (k, s, p) maps [lo, hi] to [lo s - p, hi s - p + k - 1], a stage's
// ResBlocks widen it by max_k sum_d ((k - 1) d / 2 + (k - 1) / 2), conv_post by its padding.
void generator_frame_dependence(const vsp_config& c, int pre_k, int post_k, int& back, int& fwd) {
  long lo = -(pre_k - 1) / 2, hi = (pre_k - 1) / 2, up = 1;
  for (int i = 0; i < c.n_upsamples; ++i) {
    const long s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i], p = (k - s) / 2;
    lo = lo * s - p; hi = hi * s - p + k - 1; up *= s;
    long rb = 0;
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      long acc = 0;
      const long kk = c.resblock_kernel_sizes[j];
      for (int d = 0; d < c.n_resblock_dilations; ++d) acc += (kk - 1) * c.resblock_dilation_sizes[j][d] / 2 + (kk - 1) / 2;
      rb = std::max(rb, acc);
    }
    lo -= rb; hi += rb;
  }
  lo -= (post_k - 1) / 2; hi += (post_k - 1) / 2;
  // output frame F = samples [up F, up F + up - 1] depends on input frames f with up f + lo <= n <= up f + hi
  back = (int)(hi / up);                     // F - floor(hi / up)
  fwd = (int)((up - 1 - lo) / up);           // F + floor((up - 1 - lo) / up)
}

long total_upsample(const vsp_config& c) {
  long u = 1;
  for (int i = 0; i < c.n_upsamples; ++i) u *= c.upsample_rates[i];
  return u;
}

}  // namespace

// ============================================================================================
extern "C" {

int vsp_abi_version(void) { return VSP_ABI_VERSION; }

int vsp_create(const vsp_config* cfg, int device, vsp_ctx** out) {
  if (!cfg || !out) return VSP_ERR_ARG;
  vsp_ctx* ctx = new (std::nothrow) vsp_ctx();
  if (!ctx) return VSP_ERR_ARG;
  ctx->cfg = *cfg;
  ctx->device = device;
  // Second implementations kept under test (tests/test_hip_parity.py): VSP_FRAME=f32 / VSP_ATT=f32 / VSP_GENERATOR=f32
  // (f32 matrix core), VSP_FUSE_PAIRS=0 (one launch per convolution), VSP_CHAIN=<mask> (whole-ResBlock launches:
  // bit 0 = k3, 1 = k7, 2 = k11), and the opt-in reduced precision VSP_GENERATOR=f16.  Everything else that was a
  // knob while the kernels were being tuned is compiled out of the product build (-DVSP_EXPERIMENTS brings it back).
  if (const char* e = getenv("VSP_FRAME")) ctx->frame_f16s = strcmp(e, "f32") != 0;
  if (const char* e = getenv("VSP_ATT")) ctx->att_f16s = strcmp(e, "f32") != 0;
  // the generator mode is part of the plan (conv_pre / cond packing): parse it BEFORE plan_model
  bool want_f16 = false;
  if (const char* e = getenv("VSP_GENERATOR")) {
    if (!strcmp(e, "f32")) ctx->gen_mode = 0;
    want_f16 = !strcmp(e, "f16");
  }
  if (const char* e = getenv("VSP_FUSE_PAIRS")) ctx->fuse_pairs = atoi(e) != 0;
  if (const char* e = getenv("VSP_TIMG")) ctx->t_img = atoi(e) != 0;   // 0: ResBlock intermediates as fp32 tensors (second implementation)
  if (const char* e = getenv("VSP_PP")) ctx->pp_pairs = atoi(e) != 0;  // 0: the 128-channel stage's k3 / k7 pairs as two launches
  if (const char* e = getenv("VSP_PAIR")) ctx->pair_ring = !strcmp(e, "ring");
  if (const char* e = getenv("VSP_CHAIN_RING")) ctx->chain_ring = atoi(e) != 0;
  if (const char* e = getenv("VSP_CHAIN")) ctx->chain_mask = atoi(e);
  if (const char* e = getenv("VSP_RW64")) ctx->rw64 = atoi(e) != 0;               // 1: g16_rw64 for the 64-channel k3 pairs (opt-in)
  if (const char* e = getenv("VSP_TRIM_TAILS")) ctx->trim_tails = atoi(e) != 0;   // 0: every utterance runs to the padded length
  if (const char* e = getenv("VSP_COLS")) ctx->cols = atoi(e) != 0;               // 0: no column-tile kernels (second implementation)
  if (const char* e = getenv("VSP_COLS_BLOCKS")) ctx->cols_blocks = atol(e);      // size limits of launch_conv's routing to them
  if (const char* e = getenv("VSP_COLS_MIN_BLOCKS")) ctx->cols_min_blocks = atol(e);
  if (const char* e = getenv("VSP_ACT_SCALE_LOG2")) {                             // model.h: the generator's activation scale
    const int l = atoi(e);
    ctx->act_scale = std::ldexp(1.f, l < 0 ? 0 : l > 8 ? 8 : l);
  }
  if (const char* e = getenv("VSP_EARLY_FL")) ctx->early_fl = atoi(e) != 0;
  if (const char* e = getenv("VSP_RB_STREAMS")) ctx->rb_streams = atoi(e);   // stage mask: ResBlock chains on side streams (opt-in, measured slower)
#ifdef VSP_EXPERIMENTS
  if (const char* e = getenv("VSP_ATT_KSPLIT")) ctx->att_ksplit = atoi(e);
  if (const char* e = getenv("VSP_CHUNK_MB")) ctx->chunk_mb = atof(e);
  if (const char* e = getenv("VSP_CHAIN_CH")) ctx->chain_ch = atoi(e);
  if (const char* e = getenv("VSP_CHAIN128")) ctx->chain128_mask = atoi(e);
#endif
  build_schema(ctx->cfg, ctx->schema);
  const int rc = plan_model(ctx);            // (falls back to gen_mode 0 when the channels-last kernels do not cover the config)
  if (want_f16 && ctx->gen_mode == 1) ctx->gen_mode = 2;   // opt-in reduced precision: same packing as mode 1
  *out = ctx;  // returned even on failure so that vsp_last_error can be read; caller destroys it
  return rc;
}

int vsp_destroy(vsp_ctx* ctx) {
  if (!ctx) return VSP_ERR_ARG;
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  for (auto e : ctx->sync_ev) (void)hipEventDestroy(e);
  if (ctx->fl_ev) (void)hipEventDestroy(ctx->fl_ev);
  if (ctx->fl_pinned) (void)hipHostFree(ctx->fl_pinned);
  if (ctx->flags_host) (void)hipHostFree(ctx->flags_host);
  for (auto st : ctx->side) if (st) (void)hipStreamDestroy(st);
  if (ctx->arena && ctx->arena_owned) (void)hipFree(ctx->arena);
  delete ctx;
  return VSP_OK;
}

const char* vsp_last_error(const vsp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vsp_status(vsp_ctx* ctx, unsigned* flags, int clear) {
  if (!ctx || !flags) return VSP_ERR_ARG;
  *flags = 0u;
  if (!ctx->flags_host) return VSP_OK;                 // (nothing has run on this context yet)
  volatile unsigned* w = ctx->flags_host;
  *flags = clear ? __atomic_exchange_n(ctx->flags_host, 0u, __ATOMIC_ACQ_REL) : *w;
  return VSP_OK;
}

int vsp_begin_weights(vsp_ctx* ctx) {
  if (!ctx) return VSP_ERR_ARG;
  ctx->raw.clear();
  ctx->ready = false;
  ctx->adopted_pending = false;
  return VSP_OK;
}

int vsp_set_weight(vsp_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim) {
  if (!ctx || !key || !host_data || !shape || ndim < 0 || ndim > 8) return ctx ? ctx->fail(VSP_ERR_ARG, "null argument") : VSP_ERR_ARG;
  const std::string k(key);
  // posterior encoder: voice conversion only; ignored by a context built without spec_channels
  if (k.rfind("enc_q.", 0) == 0 && ctx->cfg.spec_channels <= 0) return VSP_OK;
  auto it = ctx->schema.find(k);
  bool folded_form = false;
  if (it == ctx->schema.end()) {
    // accept a pre-folded "<x>.weight" where the schema has "<x>.weight_v" (remove_weight_norm'ed checkpoint)
    auto iv = ctx->schema.find(k + "_v");
    if (iv == ctx->schema.end()) return ctx->fail(VSP_ERR_KEY, "unknown state_dict key '%s'", key);
    it = iv;
    folded_form = true;
  }
  const SchemaEntry& e = it->second;
  bool same = (int)e.shape.size() == ndim;
  for (int i = 0; same && i < ndim; ++i) same = e.shape[i] == shape[i];
  if (!same) return ctx->fail(VSP_ERR_SHAPE, "shape mismatch for '%s'", key);
  if (!e.used) return VSP_OK;
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(host_data, host_data + t.numel());
  // one form per layer: a folded "<x>.weight" replaces an earlier weight_g / weight_v pair and vice versa (a second
  // load on the same context must not keep the other form's tensors)
  if (folded_form) {
    ctx->raw.erase(k + "_v");
    ctx->raw.erase(k + "_g");
  } else if (k.size() > 9 && (k.compare(k.size() - 9, 9, ".weight_v") == 0 || k.compare(k.size() - 9, 9, ".weight_g") == 0)) {
    ctx->raw.erase(k.substr(0, k.size() - 2));
  }
  ctx->raw[k] = std::move(t);
  ctx->ready = false;
  return VSP_OK;
}

// f16 / bf16 bit patterns -> float (host)
static float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {  // subnormal
      int e = -1;
      uint32_t m = man;
      do { ++e; m <<= 1; } while (!(m & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 112u) << 23) | (man << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

int vsp_set_weight_typed(vsp_ctx* ctx, const char* key, const void* data, const int64_t* shape, int ndim, int dtype,
                         int on_device) {
  if (!ctx || !key || !data || !shape || ndim < 0 || ndim > 8) return ctx ? ctx->fail(VSP_ERR_ARG, "null argument") : VSP_ERR_ARG;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) return ctx->fail(VSP_ERR_ARG, "negative dimension");
    n *= (size_t)shape[i];
  }
  size_t esz = 0;
  switch (dtype) {
    case VSP_DTYPE_F32: esz = 4; break;
    case VSP_DTYPE_F16: case VSP_DTYPE_BF16: esz = 2; break;
    case VSP_DTYPE_F64: esz = 8; break;
    default: return ctx->fail(VSP_ERR_ARG, "vsp_set_weight_typed: unknown dtype %d", dtype);
  }
  std::vector<unsigned char> staged;
  const unsigned char* src = static_cast<const unsigned char*>(data);
  if (on_device) {
    staged.resize(n * esz);
    hipError_t e = hipMemcpy(staged.data(), data, n * esz, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "vsp_set_weight_typed(%s): %s", key, hipGetErrorString(e));
    src = staged.data();
  }
  if (dtype == VSP_DTYPE_F32) return vsp_set_weight(ctx, key, reinterpret_cast<const float*>(src), shape, ndim);
  std::vector<float> f(n);
  for (size_t i = 0; i < n; ++i) {
    if (dtype == VSP_DTYPE_F64) { double d; std::memcpy(&d, src + 8 * i, 8); f[i] = (float)d; }
    else {
      uint16_t h; std::memcpy(&h, src + 2 * i, 2);
      if (dtype == VSP_DTYPE_BF16) { const uint32_t b = (uint32_t)h << 16; std::memcpy(&f[i], &b, 4); }
      else f[i] = half_bits_to_float(h);
    }
  }
  return vsp_set_weight(ctx, key, f.data(), shape, ndim);
}

int vsp_missing_weights(const vsp_ctx* ctx) {
  if (!ctx) return VSP_ERR_ARG;
  int n = 0;
  for (const auto& kv : ctx->schema) {
    if (!kv.second.used || kv.second.optional) continue;
    if (ctx->raw.count(kv.first)) continue;
    const std::string& k = kv.first;
    // weight_g / weight_v are satisfied by a pre-folded weight
    if (k.size() > 2 && (k.compare(k.size() - 2, 2, "_v") == 0 || k.compare(k.size() - 2, 2, "_g") == 0) &&
        ctx->raw.count(k.substr(0, k.size() - 2)))
      continue;
    ++n;
  }
  return n;
}

int64_t vsp_weight_arena_bytes(const vsp_ctx* ctx) {
  return ctx ? (int64_t)(ctx->model.total_floats * sizeof(float)) : VSP_ERR_ARG;
}

// The status word (vsp_status): pinned host memory mapped into the device's address space, so that reading it costs no
// stream synchronisation; the kernels touch it only when they have something to report.
static int ensure_flags(vsp_ctx* ctx) {
  if (ctx->flags_host) return VSP_OK;
  void* h = nullptr;
  void* d = nullptr;
  hipError_t e = hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable);
  if (e == hipSuccess) { *static_cast<unsigned*>(h) = 0u; e = hipHostGetDevicePointer(&d, h, 0); }
  if (e != hipSuccess) {
    if (h) (void)hipHostFree(h);
    return ctx->fail(VSP_ERR_HIP, "status word (hipHostMalloc): %s", hipGetErrorString(e));
  }
  ctx->flags_host = static_cast<unsigned*>(h);
  ctx->flags_dev = static_cast<unsigned*>(d);
  return VSP_OK;
}

static int set_arena(vsp_ctx* ctx, void* dev_arena) {
  if (int rc = ensure_flags(ctx)) return rc;
  if (ctx->arena && ctx->arena_owned && ctx->arena != dev_arena) (void)hipFree(ctx->arena);
  if (dev_arena) {
    ctx->arena = (float*)dev_arena;
    ctx->arena_owned = false;
  } else if (!ctx->arena || !ctx->arena_owned) {
    void* p = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&p, ctx->model.total_floats * sizeof(float));
    if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "hipMalloc(weight arena): %s", hipGetErrorString(e));
    ctx->arena = (float*)p;
    ctx->arena_owned = true;
  }
  return VSP_OK;
}

static uint32_t config_hash(const vsp_ctx* ctx) {
  // FNV-1a over the config bytes and the switches that change the packing
  uint32_t h = 2166136261u;
  auto mix = [&](const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 16777619u; }
  };
  mix(&ctx->cfg, sizeof ctx->cfg);
  const int sw[3] = {ctx->frame_f16s ? 1 : 0, ctx->model.has_cl ? 1 : 0, ctx->gen_mode != 0 ? 1 : 0};
  mix(sw, sizeof sw);
  mix(&ctx->act_scale, sizeof ctx->act_scale);   // (the packed generator biases carry it)
  return h;
}

int vsp_finalize_weights(vsp_ctx* ctx, void* dev_arena) {
  if (!ctx) return VSP_ERR_ARG;
  if (ctx->model.total_floats == 0) return ctx->fail(VSP_ERR_STATE, "context was not planned (vsp_create failed)");
  const int miss = vsp_missing_weights(ctx);
  if (miss) return ctx->fail(VSP_ERR_STATE, "%d infer-path tensors missing", miss);
  std::vector<float> host;
  int rc = fill_model(ctx, host);
  if (rc) return rc;
  {
    const uint64_t tf = ctx->model.total_floats;
    const uint32_t hdr[6] = {ARENA_MAGIC, (uint32_t)VSP_ABI_VERSION, (uint32_t)(tf & 0xffffffffu), (uint32_t)(tf >> 32),
                             ctx->model.has_vc ? ARENA_FLAG_VC : 0u, config_hash(ctx)};
    std::memcpy(host.data(), hdr, sizeof hdr);
  }
  rc = set_arena(ctx, dev_arena);
  if (rc) return rc;
  hipError_t e = hipMemcpy(ctx->arena, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "hipMemcpy(weight arena): %s", hipGetErrorString(e));
  ctx->adopted_pending = false;
  ctx->ready = true;
  return VSP_OK;
}

int vsp_adopt_packed_weights(vsp_ctx* ctx, void* dev_arena) {
  if (!ctx || !dev_arena) return ctx ? ctx->fail(VSP_ERR_ARG, "null arena") : VSP_ERR_ARG;
  if (ctx->model.total_floats == 0) return ctx->fail(VSP_ERR_STATE, "context was not planned");
  const int rc = set_arena(ctx, dev_arena);
  if (rc) return rc;
  // the bytes may not have arrived yet (the broadcast follows): nothing is known about them until
  // vsp_commit_adopted_weights has read the header -- fail closed until then
  ctx->model.has_vc = false;
  ctx->ready = false;
  ctx->adopted_pending = true;
  return VSP_OK;
}

int vsp_commit_adopted_weights(vsp_ctx* ctx, void* stream) {
  if (!ctx) return VSP_ERR_ARG;
  if (!ctx->adopted_pending || !ctx->arena) return ctx->fail(VSP_ERR_STATE, "no adopted arena to commit");
  uint32_t hdr[6] = {0, 0, 0, 0, 0, 0};
  hipError_t e = hipMemcpyAsync(hdr, ctx->arena, sizeof hdr, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "arena header read: %s", hipGetErrorString(e));
  const uint64_t tf = (uint64_t)hdr[2] | ((uint64_t)hdr[3] << 32);
  if (hdr[0] != ARENA_MAGIC) return ctx->fail(VSP_ERR_STATE, "adopted arena has no header (bytes not broadcast yet, or not a packed arena)");
  if (hdr[1] != (uint32_t)VSP_ABI_VERSION) return ctx->fail(VSP_ERR_STATE, "adopted arena was packed by ABI %u, this library is ABI %d", hdr[1], VSP_ABI_VERSION);
  if (tf != ctx->model.total_floats || hdr[5] != config_hash(ctx))
    return ctx->fail(VSP_ERR_STATE, "adopted arena was packed for a different configuration");
  ctx->model.has_vc = (hdr[4] & ARENA_FLAG_VC) != 0;
  ctx->adopted_pending = false;
  ctx->ready = true;
  return VSP_OK;
}

int vsp_has_voice_conversion(const vsp_ctx* ctx) { return ctx && ctx->ready && ctx->model.has_vc ? 1 : 0; }

int vsp_weight_arena(const vsp_ctx* ctx, void** dev_arena, int64_t* bytes) {
  if (!ctx || !dev_arena || !bytes) return VSP_ERR_ARG;
  *dev_arena = ctx->arena;
  *bytes = (int64_t)(ctx->model.total_floats * sizeof(float));
  return VSP_OK;
}

// -------------------------------------------------------------------------------------------- encode
// pinned host buffer + event of the early frame-count copy (created on first use, grown by doubling; kept with the context)
static bool early_frame_lengths_ready(vsp_ctx* ctx, int B) {
  if (!ctx->fl_ev && hipEventCreateWithFlags(&ctx->fl_ev, hipEventDisableTiming) != hipSuccess) {
    ctx->fl_ev = nullptr;
    return false;
  }
  if (ctx->fl_cap < B) {
    if (ctx->fl_pinned) (void)hipHostFree(ctx->fl_pinned);
    ctx->fl_pinned = nullptr;
    int cap = std::max(64, ctx->fl_cap);
    while (cap < B) cap *= 2;
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->fl_pinned), (size_t)cap * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
      ctx->fl_pinned = nullptr;
      ctx->fl_cap = 0;
      return false;
    }
    ctx->fl_cap = cap;
  }
  return true;
}

static int encode_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int B, int Tp, const int64_t* phonemes,
                       const int64_t* lengths, const int64_t* sid, const float* dctl, const float* pctl,
                       const float* ectl, float dscale, float pscale, float escale, float* x_var, float* g,
                       float* duration, float* f0, float* energy, int64_t* frame_lengths, int32_t* cum_dur,
                       bool early_copy = true) {
  const vsp_config& c = ctx->cfg;
  const Model& m = ctx->model;
  const int h = c.hidden_channels, gin = c.gin_channels;
  Run r{ctx, s, ws};
  const T3 XV = ext(x_var, h, Tp);
  T3 XE = ws.t3(B, h, Tp), TMP = ws.t3(B, h, Tp);
  float* cvec = ws.f((size_t)B * h);
  float* lf0 = ws.f((size_t)B * Tp);
  float* pred = ws.f((size_t)B * Tp);
  float* norm_e = ws.f((size_t)B * Tp);
  // duration-predictor / energy-predictor activations
  const int fmax_ = std::max(c.dur_filter, c.energy_filter);
  T3 P1 = ws.t3(B, fmax_, Tp), P2 = ws.t3(B, fmax_, Tp);
  const bool live = !ws.dry && !ws.overflow;
  if (live) {
    r.chk(launch_gather_rows(sid, r.A(m.emb_g), c.n_speakers, g, B, gin, s), "emb_g");
    r.chk(launch_embed(phonemes, r.A(m.emb_sym), c.n_vocab, sqrtf((float)h), XE.p, XE.bs, XE.cs, B, h, Tp, s),
          "symbol_emb");
  }
  // ---- given durations (models.py:681): the frame counts need nothing computed here -- derive them FIRST and start their
  // copy to the host, so that vsp_frame_lengths_host returns while the text encoder runs (vsp_ctx::fl_pinned)
  if (live) { ctx->fl_src = nullptr; ctx->fl_known_src = nullptr; }
  if (dctl && live) {
    r.chk(hipMemcpyAsync(duration, dctl, (size_t)B * Tp * sizeof(float), hipMemcpyDeviceToDevice, s), "dur copy");
    r.chk(launch_duration_cumsum(duration, cum_dur, frame_lengths, B, Tp, s, ctx->flags_dev), "duration cumsum");
    // (early_copy == false: the one-call form vsp_infer never reads the counts back -- nothing would consume the copy)
    if (r.ok() && early_copy && ctx->early_fl && early_frame_lengths_ready(ctx, B)) {
      hipError_t e = hipMemcpyAsync(ctx->fl_pinned, frame_lengths, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipEventRecord(ctx->fl_ev, s);
      if (e == hipSuccess) { ctx->fl_src = frame_lengths; ctx->fl_n = B; }
    }
  }
  mask3(r, XE, lengths, B, h, Tp);  // TextEncoder passes x * x_mask (models.py:173)
  run_encoder_masked(r, m.enc[0], B, Tp, XE, lengths, XV);   // XV = x_enc
  // ---- duration (models.py:681-688, 119-133)
  if (!dctl) {
    const int f = c.dur_filter;
    T3 A1 = P1, A2 = P2;
    r.cond(m.dur_cond, g, cvec, B);
    if (live) r.chk(launch_add_cond(XV.p, XV.bs, XV.cs, cvec, h, TMP.p, TMP.bs, TMP.cs, B, h, Tp, s), "add_cond");
    ConvArgs a = r.args(m.dur_c1, TMP, A1, Tp, Tp);
    a.lengths = lengths; a.in_mask = 1; a.act = 1;
    r.conv(a, B);
    r.ln(A1, T3{}, m.dur_g1, m.dur_b1, A1, B, f, Tp);
    a = r.args(m.dur_c2, A1, A2, Tp, Tp);
    a.lengths = lengths; a.in_mask = 1; a.act = 1;
    r.conv(a, B);
    r.ln(A2, T3{}, m.dur_g2, m.dur_b2, A2, B, f, Tp);
    if (live) {
      r.chk(launch_chan_dot(A2.p, A2.bs, A2.cs, r.A(m.dur_pw), r.A(m.dur_pb), lengths, 1, 1, pred, B, f, Tp, s), "dur proj");
      r.chk(launch_duration_from_logw(pred, lengths, dscale, duration, B, Tp, s), "duration");
    }
  }
  // ---- pitch (models.py:691-698, 505-514)
  if (!pctl) {
    T3 PI = ws.t3(B, h, Tp);
    r.cond(m.pit_cond, g, cvec, B);
    if (live) r.chk(launch_add_cond(XV.p, XV.bs, XV.cs, cvec, h, TMP.p, TMP.bs, TMP.cs, B, h, Tp, s), "add_cond");
    mask3(r, TMP, lengths, B, h, Tp);
    run_encoder_masked(r, m.enc[1], B, Tp, TMP, lengths, PI);
    if (live) r.chk(launch_chan_dot(PI.p, PI.bs, PI.cs, r.A(m.pit_pw), r.A(m.pit_pb), lengths, 1, 0, pred, B, h, Tp, s), "proj_f0");
  }
  if (live) {
    r.chk(launch_pitch(pctl, pred, pscale, lf0, f0, B * Tp, s), "pitch");
    r.chk(launch_prenet_add(XV.p, XV.bs, XV.cs, r.A(m.ppre_w), r.A(m.ppre_b), lf0, B, h, Tp, s), "pitch_prenet");
  }
  // ---- energy (models.py:701-708; frame_prior_network.py:104-124: no mask anywhere)
  if (!ectl) {
    const int e = c.energy_filter;
    T3 A1 = P1, A2 = P2;
    r.cond(m.en_cond, g, cvec, B);
    if (live) r.chk(launch_add_cond(XV.p, XV.bs, XV.cs, cvec, h, TMP.p, TMP.bs, TMP.cs, B, h, Tp, s), "add_cond");
    ConvArgs a = r.args(m.en_c1, TMP, A1, Tp, Tp);
    a.act = 1;
    r.conv(a, B);
    r.ln(A1, T3{}, m.en_g1, m.en_b1, A1, B, e, Tp);
    a = r.args(m.en_c2, A1, A2, Tp, Tp);
    a.act = 1;
    r.conv(a, B);
    r.ln(A2, T3{}, m.en_g2, m.en_b2, A2, B, e, Tp);
    if (live) r.chk(launch_chan_dot(A2.p, A2.bs, A2.cs, r.A(m.en_lw), r.A(m.en_lb), nullptr, 0, 0, pred, B, e, Tp, s), "energy linear");
  }
  if (live) {
    r.chk(launch_energy(ectl, pred, escale, norm_e, energy, B * Tp, s), "energy");
    r.chk(launch_prenet_add(XV.p, XV.bs, XV.cs, r.A(m.epre_w), r.A(m.epre_b), norm_e, B, h, Tp, s), "energy_prenet");
    if (!dctl) r.chk(launch_duration_cumsum(duration, cum_dur, frame_lengths, B, Tp, s, ctx->flags_dev), "duration cumsum");
  }
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "encode workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

int64_t vsp_encode_workspace_bytes(const vsp_ctx* ctx, int B, int Tp) {
  if (!ctx || B <= 0 || Tp <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  encode_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, B, Tp, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1,
              1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_encode(vsp_ctx* ctx, void* stream, int B, int Tp, const int64_t* phonemes, const int64_t* lengths,
               const int64_t* sid, const float* duration_ctl, const float* pitch_ctl, const float* energy_ctl,
               float duration_scale, float pitch_scale, float energy_scale, float* x_var, float* g, float* duration,
               float* f0, float* energy, int64_t* frame_lengths, int32_t* cum_dur, void* workspace,
               int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || Tp <= 0 || !phonemes || !lengths || !sid || !x_var || !g || !duration || !f0 || !energy ||
      !frame_lengths || !cum_dur || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_encode: null or non-positive argument");
  const int64_t need = vsp_encode_workspace_bytes(ctx, B, Tp);
  if (workspace_bytes < need)
    return ctx->fail(VSP_ERR_WORKSPACE, "encode workspace too small: %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return encode_impl(ctx, (hipStream_t)stream, ws, B, Tp, phonemes, lengths, sid, duration_ctl, pitch_ctl, energy_ctl,
                     duration_scale, pitch_scale, energy_scale, x_var, g, duration, f0, energy, frame_lengths, cum_dur);
}

int vsp_frame_lengths_host(vsp_ctx* ctx, void* stream, int B, const int64_t* frame_lengths_dev,
                           int64_t* frame_lengths_host, int64_t* max_frames) {
  if (!ctx || B <= 0 || !frame_lengths_dev || !frame_lengths_host || !max_frames)
    return ctx ? ctx->fail(VSP_ERR_ARG, "vsp_frame_lengths_host: bad argument") : VSP_ERR_ARG;
  hipError_t e;
  if (ctx->fl_src == frame_lengths_dev && ctx->fl_n == B) {
    // vsp_encode already started this copy (given durations): wait for IT, not for the rest of the stream
    e = hipEventSynchronize(ctx->fl_ev);
    if (e == hipSuccess) std::memcpy(frame_lengths_host, ctx->fl_pinned, (size_t)B * sizeof(int64_t));
    ctx->fl_src = nullptr;
  } else {
    e = hipMemcpyAsync(frame_lengths_host, frame_lengths_dev, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToHost,
                       (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  }
  if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "frame length read: %s", hipGetErrorString(e));
  int64_t mx = 0;
  for (int b = 0; b < B; ++b) mx = std::max(mx, frame_lengths_host[b]);
  *max_frames = mx;
  ctx->fl_known.assign(frame_lengths_host, frame_lengths_host + B);   // (model.h: until the next vsp_encode)
  ctx->fl_known_src = frame_lengths_dev;
  return VSP_OK;
}

// -------------------------------------------------------------------------------------------- decode
static int decode_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int B, int Tp, int Tf, int max_len, const float* x_var,
                       const float* g, const int32_t* cum_dur, const int64_t* frame_lengths, const float* noise,
                       uint64_t noise_seed, float noise_scale, float* o, uint8_t* x_mask, float* z, float* z_p, float* m_p,
                       float* logs_p) {
  const vsp_config& c = ctx->cfg;
  const Model& m = ctx->model;
  const int h = c.hidden_channels, inter = c.inter_channels;
  Run r{ctx, s, ws};
  if (frame_lengths && ctx->fl_known_src == frame_lengths && (int)ctx->fl_known.size() == B) r.host_lengths = ctx->fl_known.data();
  T3 XF = ws.t3(B, h, Tf), HF = ws.t3(B, h, Tf);
  // noise == NULL: the library draws it (Philox4x32-10 keyed by noise_seed) -- the torch.randn_like of models.py:718
  float* drawn = ws.f((size_t)B * inter * Tf);
  const bool live = !ws.dry && !ws.overflow;
  if (live && !noise && noise_scale != 0.f) {
    r.chk(launch_randn(noise_seed, (long)ctx->noise_first, (long)B * inter * Tf, drawn, s), "randn");
    noise = drawn;
  }
  if (live) {
    r.chk(launch_length_regulate(x_var, (long)h * Tp, Tp, cum_dur, XF.p, XF.bs, XF.cs, B, h, Tp, Tf, s), "length_regulate");
    r.chk(launch_mask_u8(frame_lengths, x_mask, B, Tf, s), "x_mask");
  }
  run_encoder_masked(r, m.enc[2], B, Tf, XF, frame_lengths, HF);
  const T3 MP = ext(m_p, inter, Tf), LP = ext(logs_p, inter, Tf), Z = ext(z, inter, Tf);
  // Projection (reference models.py:526-529): one 1x1 convolution, m_p and logs_p rows to their own tensors
  ConvArgs a = r.args(m.proj, HF, MP, Tf, Tf);
  a.lengths = frame_lengths; a.mask_post = 1;
  a.split_row = inter; a.out2 = LP.p; a.o2_bs = LP.bs; a.o2_cs = LP.cs; a.mask_post2 = 1;
  r.conv(a, B);
  if (live) {
    const long n = (long)B * inter * Tf;
    r.chk(launch_reparam(m_p, logs_p, noise, noise_scale, z_p, n, s, z, ctx->flags_dev), "reparam");   // (z = z_p: the flow transforms z in place)
  }
  run_flow(r, B, Tf, Z, g, frame_lengths);
  const int Tdec = max_len < 0 ? Tf : std::min(Tf, max_len);
  if (Tdec > 0) run_gen(r, B, Tdec, Z, frame_lengths, g, o);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "decode workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

int64_t vsp_decode_workspace_bytes(const vsp_ctx* ctx, int B, int Tp, int Tf) {
  if (!ctx || B <= 0 || Tp <= 0 || Tf <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  decode_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, B, Tp, Tf, -1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_decode(vsp_ctx* ctx, void* stream, int B, int Tp, int Tf, int max_len, const float* x_var, const float* g,
               const int32_t* cum_dur, const int64_t* frame_lengths, const float* noise, uint64_t noise_seed,
               float noise_scale, float* o, uint8_t* x_mask, float* z, float* z_p, float* m_p, float* logs_p,
               void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || Tp <= 0 || Tf <= 0 || !x_var || !g || !cum_dur || !frame_lengths || !o || !x_mask || !z || !z_p ||
      !m_p || !logs_p || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_decode: null or non-positive argument");
  const int64_t need = vsp_decode_workspace_bytes(ctx, B, Tp, Tf);
  if (workspace_bytes < need)
    return ctx->fail(VSP_ERR_WORKSPACE, "decode workspace too small: %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return decode_impl(ctx, (hipStream_t)stream, ws, B, Tp, Tf, max_len, x_var, g, cum_dur, frame_lengths, noise,
                     noise_seed, noise_scale, o, x_mask, z, z_p, m_p, logs_p);
}

// one-call form: encode + decode with a caller-supplied frame padding, no host synchronisation
static int infer_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int B, int Tp, int Tf, int max_len, const int64_t* phonemes,
                      const int64_t* lengths, const int64_t* sid, const float* dctl, const float* pctl, const float* ectl,
                      float dsc, float psc, float esc, const float* noise, uint64_t noise_seed, float noise_scale, float* o,
                      uint8_t* x_mask, float* z, float* z_p, float* m_p, float* logs_p, float* duration, float* f0,
                      float* energy, int64_t* frame_lengths) {
  const vsp_config& c = ctx->cfg;
  float* x_var = ws.f((size_t)B * c.hidden_channels * Tp);
  float* g = ws.f((size_t)B * c.gin_channels);
  int32_t* cum = (int32_t*)ws.bytes((size_t)B * Tp * sizeof(int32_t));
  const size_t mark = ws.cur;
  int rc = encode_impl(ctx, s, ws, B, Tp, phonemes, lengths, sid, dctl, pctl, ectl, dsc, psc, esc, x_var, g, duration,
                       f0, energy, frame_lengths, cum, /*early_copy=*/false);
  const size_t after_enc = ws.cur;
  ws.cur = mark;                 // the two halves run one after the other on one stream: shared scratch
  if (rc == VSP_OK)
    rc = decode_impl(ctx, s, ws, B, Tp, Tf, max_len, x_var, g, cum, frame_lengths, noise, noise_seed, noise_scale, o, x_mask,
                     z, z_p, m_p, logs_p);
  ws.cur = std::max(ws.cur, after_enc);
  return rc;
}

int64_t vsp_infer_workspace_bytes(const vsp_ctx* ctx, int B, int Tp, int tf_pad) {
  if (!ctx || B <= 0 || Tp <= 0 || tf_pad <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  infer_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, B, Tp, tf_pad, -1, nullptr, nullptr, nullptr, nullptr, nullptr,
             nullptr, 1.f, 1.f, 1.f, nullptr, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
             nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_infer(vsp_ctx* ctx, void* stream, int B, int Tp, int tf_pad, int max_len, const int64_t* phonemes,
              const int64_t* lengths, const int64_t* sid, const float* duration_ctl, const float* pitch_ctl,
              const float* energy_ctl, float duration_scale, float pitch_scale, float energy_scale, const float* noise,
              uint64_t noise_seed, float noise_scale, float* o, uint8_t* x_mask, float* z, float* z_p, float* m_p, float* logs_p,
              float* duration, float* f0, float* energy, int64_t* frame_lengths, void* workspace,
              int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || Tp <= 0 || tf_pad <= 0 || !phonemes || !lengths || !sid || !o || !x_mask || !z || !z_p || !m_p ||
      !logs_p || !duration || !f0 || !energy || !frame_lengths || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_infer: null or non-positive argument");
  const int64_t need = vsp_infer_workspace_bytes(ctx, B, Tp, tf_pad);
  if (workspace_bytes < need)
    return ctx->fail(VSP_ERR_WORKSPACE, "infer workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
                     (long long)need);
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return infer_impl(ctx, (hipStream_t)stream, ws, B, Tp, tf_pad, max_len, phonemes, lengths, sid, duration_ctl, pitch_ctl,
                    energy_ctl, duration_scale, pitch_scale, energy_scale, noise, noise_seed, noise_scale, o, x_mask, z, z_p,
                    m_p, logs_p, duration, f0, energy, frame_lengths);
}

// -------------------------------------------------------------------------------------------- stages
int64_t vsp_attention_workspace_bytes(const vsp_ctx* ctx, int B, int T) {
  if (!ctx || B <= 0 || T <= 0) return VSP_ERR_ARG;
  const vsp_config& c = ctx->cfg;
  // the packed q | k | v operand images of the split-f16 kernel; the f32 kernel (VSP_ATT=f32) needs none
  return ctx->att_f16s ? (int64_t)(3 * attn_pack_bytes(B, c.n_heads, c.hidden_channels / c.n_heads, T)) : 256;
}

int vsp_attention(vsp_ctx* ctx, void* stream, int which, int layer, int B, int T, const float* qkv, const int64_t* lengths,
                  float* out, void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (which < 0 || which > 2 || B <= 0 || T <= 0 || !qkv || !lengths || !out || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_attention: bad argument");
  const EncoderW& E = ctx->model.enc[which];
  if (layer < 0 || layer >= (int)E.layers.size()) return ctx->fail(VSP_ERR_ARG, "vsp_attention: no such layer");
  if (workspace_bytes < vsp_attention_workspace_bytes(ctx, B, T))
    return ctx->fail(VSP_ERR_WORKSPACE, "attention workspace too small (need %lld bytes)",
                     (long long)vsp_attention_workspace_bytes(ctx, B, T));
  const vsp_config& c = ctx->cfg;
  const int h = c.hidden_channels;
  const EncLayer& L = E.layers[layer];
  hipError_t e;
  if (ctx->att_f16s)
    e = launch_attention_f16s(qkv, 3L * h * T, T, ctx->arena + L.ek, ctx->arena + L.ev, lengths, out, (long)h * T, T, B, h,
                              c.n_heads, T, c.window_size, workspace, (hipStream_t)stream);
  else
    e = launch_attention(qkv, 3L * h * T, T, ctx->arena + L.ek, ctx->arena + L.ev, lengths, out, (long)h * T, T,
                         B, h, c.n_heads, T, c.window_size, ctx->att_ksplit, (hipStream_t)stream);
  return e == hipSuccess ? VSP_OK : ctx->fail(VSP_ERR_HIP, "attention: %s", hipGetErrorString(e));
}

int64_t vsp_encoder_workspace_bytes(const vsp_ctx* ctx, int B, int T) {
  if (!ctx || B <= 0 || T <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  Run r{const_cast<vsp_ctx*>(ctx), nullptr, ws};
  ws.t3(B, ctx->cfg.hidden_channels, T);
  run_encoder_masked(r, ctx->model.enc[0], B, T, T3{}, nullptr, T3{});
  return (int64_t)ws.cur;
}

int vsp_encoder(vsp_ctx* ctx, void* stream, int which, int B, int T, const float* x, const int64_t* lengths, float* y,
                void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (which < 0 || which > 2 || B <= 0 || T <= 0 || !x || !lengths || !y || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_encoder: bad argument");
  if (workspace_bytes < vsp_encoder_workspace_bytes(ctx, B, T))
    return ctx->fail(VSP_ERR_WORKSPACE, "encoder workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  Run r{ctx, (hipStream_t)stream, ws};
  const int h = ctx->cfg.hidden_channels;
  // the reference call sites pass x * x_mask (models.py:173, 469, 511): mask a private copy
  T3 XI = ws.t3(B, h, T);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "encoder workspace too small");
  r.chk(launch_copy3(x, (long)h * T, T, XI.p, XI.bs, XI.cs, B, h, T, r.s), "copy");
  mask3(r, XI, lengths, B, h, T);
  run_encoder_masked(r, ctx->model.enc[which], B, T, XI, lengths, ext(y, h, T));
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "encoder workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

int vsp_length_regulate(vsp_ctx* ctx, void* stream, int B, int C, int Tp, int Tf, const float* x, const int32_t* cum_dur,
                        float* x_frame) {
  if (!ctx || B <= 0 || C <= 0 || Tp <= 0 || Tf <= 0 || !x || !cum_dur || !x_frame)
    return ctx ? ctx->fail(VSP_ERR_ARG, "vsp_length_regulate: bad argument") : VSP_ERR_ARG;
  hipError_t e = launch_length_regulate(x, (long)C * Tp, Tp, cum_dur, x_frame, (long)C * Tf, Tf, B, C, Tp, Tf,
                                        (hipStream_t)stream);
  return e == hipSuccess ? VSP_OK : ctx->fail(VSP_ERR_HIP, "length_regulate: %s", hipGetErrorString(e));
}

int64_t vsp_flow_workspace_bytes(const vsp_ctx* ctx, int B, int Tf) {
  if (!ctx || B <= 0 || Tf <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  Run r{const_cast<vsp_ctx*>(ctx), nullptr, ws};
  run_flow(r, B, Tf, T3{}, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_flow_reverse(vsp_ctx* ctx, void* stream, int B, int Tf, const float* z_p, const float* g,
                     const int64_t* frame_lengths, float* z, void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || Tf <= 0 || !z_p || !g || !frame_lengths || !z || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_flow_reverse: bad argument");
  if (workspace_bytes < vsp_flow_workspace_bytes(ctx, B, Tf))
    return ctx->fail(VSP_ERR_WORKSPACE, "flow workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  Run r{ctx, (hipStream_t)stream, ws};
  const long n = (long)B * ctx->cfg.inter_channels * Tf;
  r.chk(hipMemcpyAsync(z, z_p, n * sizeof(float), hipMemcpyDeviceToDevice, r.s), "z copy");
  run_flow(r, B, Tf, ext(z, ctx->cfg.inter_channels, Tf), g, frame_lengths);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "flow workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

int vsp_flow_forward(vsp_ctx* ctx, void* stream, int B, int Tf, const float* z, const float* g,
                     const int64_t* frame_lengths, float* z_p, void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || Tf <= 0 || !z || !g || !frame_lengths || !z_p || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_flow_forward: bad argument");
  if (workspace_bytes < vsp_flow_workspace_bytes(ctx, B, Tf))
    return ctx->fail(VSP_ERR_WORKSPACE, "flow workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  Run r{ctx, (hipStream_t)stream, ws};
  const long n = (long)B * ctx->cfg.inter_channels * Tf;
  r.chk(hipMemcpyAsync(z_p, z, n * sizeof(float), hipMemcpyDeviceToDevice, r.s), "z copy");
  run_flow(r, B, Tf, ext(z_p, ctx->cfg.inter_channels, Tf), g, frame_lengths, false);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "flow workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

static int check_vc(vsp_ctx* ctx);

// One layer of modules.WN.forward (reference modules.py:148-176), for unit parity: which 0 .. n_flows-1 = the WN of
// flow.flows[2 * which], -1 = enc_q.enc.
static int wn_layer_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int which, int layer, int B, int T, float* x, const float* g,
                         const int64_t* lengths, float* skip, int accumulate) {
  const vsp_config& c = ctx->cfg;
  const Model& m = ctx->model;
  const int h = c.hidden_channels;
  const bool post = which < 0;
  const Conv& cond = post ? m.enc_q.cond : m.flows[which].cond;
  const std::vector<Conv>& in = post ? m.enc_q.in : m.flows[which].in;
  const std::vector<Conv>& res = post ? m.enc_q.res : m.flows[which].res;
  const std::vector<Conv>& sk = post ? m.enc_q.skip : m.flows[which].skip;
  const int nl = post ? c.posterior_layers : c.flow_layers;
  Run r{ctx, s, ws};
  T3 ACT = ws.t3(B, h, T);
  float* gc = ws.f((size_t)B * 2 * h * nl);
  if (ws.dry) return VSP_OK;
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "wn layer workspace too small");
  const T3 H = ext(x, h, T), OUT = ext(skip, h, T);
  r.cond(cond, g, gc, B);                              // cond_layer(g): all layers' rows, this layer's slice is used
  ConvArgs a = r.args(in[layer], H, ACT, T, T);
  a.act = 2; a.cond = gc + (size_t)layer * 2 * h; a.cond_bs = 2L * h * nl;
  r.conv(a, B);
  if (layer < nl - 1) {
    a = r.args(res[layer], ACT, H, T, T);
    a.res = H.p; a.r_bs = H.bs; a.r_cs = H.cs;
    a.lengths = lengths; a.mask_post = 1;
    a.split_row = h; a.out2 = OUT.p; a.o2_bs = OUT.bs; a.o2_cs = OUT.cs; a.acc_prev2 = accumulate ? 1 : 0;
    r.conv(a, B);
  } else {
    a = r.args(sk[layer], ACT, OUT, T, T);
    a.acc_prev = accumulate ? 1 : 0;
    a.lengths = lengths; a.mask_post = 1;
    r.conv(a, B);
  }
  return r.rc;
}

int64_t vsp_wn_layer_workspace_bytes(const vsp_ctx* ctx, int which, int B, int T) {
  if (!ctx || B <= 0 || T <= 0 || which < -1 || which >= ctx->cfg.n_flows) return VSP_ERR_ARG;
  if (which < 0 && ctx->cfg.spec_channels <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  wn_layer_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, which, 0, B, T, nullptr, nullptr, nullptr, nullptr, 0);
  return (int64_t)ws.cur;
}

int vsp_wn_layer(vsp_ctx* ctx, void* stream, int which, int layer, int B, int T, float* x, const float* g,
                 const int64_t* lengths, float* skip, int accumulate, void* workspace, int64_t workspace_bytes) {
  int rc = which < 0 ? check_vc(ctx) : check_ready(ctx);
  if (rc) return rc;
  if (which < -1 || which >= ctx->cfg.n_flows || B <= 0 || T <= 0 || !x || !g || !lengths || !skip || !workspace || x == skip)
    return ctx->fail(VSP_ERR_ARG, "vsp_wn_layer: bad argument");
  const int nl = which < 0 ? ctx->cfg.posterior_layers : ctx->cfg.flow_layers;
  if (layer < 0 || layer >= nl) return ctx->fail(VSP_ERR_ARG, "vsp_wn_layer: no such layer");
  if (workspace_bytes < vsp_wn_layer_workspace_bytes(ctx, which, B, T))
    return ctx->fail(VSP_ERR_WORKSPACE, "wn layer workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return wn_layer_impl(ctx, (hipStream_t)stream, ws, which, layer, B, T, x, g, lengths, skip, accumulate);
}

int vsp_randn(void* stream, uint64_t seed, int64_t n, float* out) { return vsp_randn_at(stream, seed, 0, n, out); }

int vsp_randn_at(void* stream, uint64_t seed, int64_t first, int64_t n, float* out) {
  if (n < 0 || first < 0 || (n > 0 && !out)) return VSP_ERR_ARG;
  return launch_randn(seed, (long)first, (long)n, out, (hipStream_t)stream) == hipSuccess ? VSP_OK : VSP_ERR_HIP;
}

int vsp_set_noise_offset(vsp_ctx* ctx, int64_t first_element) {
  if (!ctx || first_element < 0) return ctx ? ctx->fail(VSP_ERR_ARG, "vsp_set_noise_offset: negative offset") : VSP_ERR_ARG;
  ctx->noise_first = first_element;
  return VSP_OK;
}

// ---------------------------------------------------------------------------------- spectrogram
int vsp_spectrogram_frames(const vsp_ctx* ctx, int L, int hop) {
  if (!ctx || ctx->cfg.spec_channels <= 1 || hop <= 0) return VSP_ERR_ARG;
  const int n_fft = 2 * (ctx->cfg.spec_channels - 1), pad = (n_fft - hop) / 2;
  if (L <= pad || n_fft < hop) return VSP_ERR_ARG;      // reflect padding needs pad < L
  return 1 + (L + 2 * pad - n_fft) / hop;
}

static int spectrogram_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int B, int L, int hop, int T, const float* audio,
                            float* spec) {
  const vsp_config& c = ctx->cfg;
  const int n_fft = 2 * (c.spec_channels - 1);
  Run r{ctx, s, ws};
  T3 F = ws.t3(B, n_fft, T), RI = ws.t3(B, 2 * c.spec_channels, T);
  if (ws.dry) return VSP_OK;
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "spectrogram workspace too small");
  r.chk(launch_stft_frames(audio, L, F.p, F.bs, F.cs, B, L, n_fft, hop, T, s), "stft frames");
  ConvArgs a = r.args(ctx->model.stft, F, RI, T, T);
  r.conv(a, B);
  if (r.ok()) r.chk(launch_stft_magnitude(RI.p, RI.bs, RI.cs, spec, B, c.spec_channels, T, s), "stft magnitude");
  return r.rc;
}

int64_t vsp_spectrogram_workspace_bytes(const vsp_ctx* ctx, int B, int L, int hop) {
  const int T = vsp_spectrogram_frames(ctx, L, hop);
  if (T <= 0 || B <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  spectrogram_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, B, L, hop, T, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_spectrogram(vsp_ctx* ctx, void* stream, int B, int L, int hop, const float* audio, float* spec, void* workspace,
                    int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (ctx->cfg.spec_channels <= 1) return ctx->fail(VSP_ERR_STATE, "context was created without spec_channels");
  const int T = vsp_spectrogram_frames(ctx, L, hop);
  if (T <= 0 || B <= 0 || !audio || !spec || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_spectrogram: bad argument (the signal must be longer than (n_fft - hop) / 2)");
  if (workspace_bytes < vsp_spectrogram_workspace_bytes(ctx, B, L, hop))
    return ctx->fail(VSP_ERR_WORKSPACE, "spectrogram workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return spectrogram_impl(ctx, (hipStream_t)stream, ws, B, L, hop, T, audio, spec);
}

// ---------------------------------------------------------------------------------- voice conversion
static int check_vc(vsp_ctx* ctx) {
  const int rc = check_ready(ctx);
  if (rc) return rc;
  if (ctx->cfg.spec_channels <= 0) return ctx->fail(VSP_ERR_STATE, "context was created without spec_channels");
  if (!ctx->model.has_vc) return ctx->fail(VSP_ERR_STATE, "enc_q.* tensors were not loaded before vsp_finalize_weights");
  return VSP_OK;
}

int64_t vsp_posterior_workspace_bytes(const vsp_ctx* ctx, int B, int T) {
  if (!ctx || B <= 0 || T <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  Run r{const_cast<vsp_ctx*>(ctx), nullptr, ws};
  run_posterior(r, B, T, T3{}, nullptr, nullptr, nullptr, T3{}, T3{}, T3{});
  return (int64_t)ws.cur;
}

int vsp_posterior_encoder(vsp_ctx* ctx, void* stream, int B, int T, const float* y, const int64_t* y_lengths,
                          const float* g, const float* noise, float* z, float* m, float* logs, void* workspace,
                          int64_t workspace_bytes) {
  int rc = check_vc(ctx);
  if (rc) return rc;
  if (B <= 0 || T <= 0 || !y || !y_lengths || !g || !noise || !z || !m || !logs || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_posterior_encoder: bad argument");
  if (workspace_bytes < vsp_posterior_workspace_bytes(ctx, B, T))
    return ctx->fail(VSP_ERR_WORKSPACE, "posterior workspace too small");
  Ws ws(workspace, (size_t)workspace_bytes, false);
  Run r{ctx, (hipStream_t)stream, ws};
  const int inter = ctx->cfg.inter_channels;
  run_posterior(r, B, T, ext(y, ctx->cfg.spec_channels, T), y_lengths, g, noise, ext(z, inter, T), ext(m, inter, T),
                ext(logs, inter, T));
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "posterior workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

// SynthesizerTrn.voice_conversion (reference models.py:724-732)
static int vc_impl(vsp_ctx* ctx, hipStream_t s, Ws& ws, int B, int T, const float* y, const int64_t* y_lengths,
                   const int64_t* sid_src, const int64_t* sid_tgt, const float* noise, float* o_hat, uint8_t* y_mask,
                   float* z, float* z_p, float* z_hat, float* m_q, float* logs_q) {
  const vsp_config& c = ctx->cfg;
  const Model& m = ctx->model;
  const int inter = c.inter_channels, gin = c.gin_channels;
  Run r{ctx, s, ws};
  float* g_src = ws.f((size_t)B * gin);
  float* g_tgt = ws.f((size_t)B * gin);
  T3 MQ = m_q ? ext(m_q, inter, T) : ws.t3(B, inter, T);
  T3 LQ = logs_q ? ext(logs_q, inter, T) : ws.t3(B, inter, T);
  if (!m_q || !logs_q) {
    // launch_reparam walks contiguous tensors: the scratch copies must be dense too
    MQ.cs = LQ.cs = T; MQ.bs = LQ.bs = (long)inter * T;
  }
  const bool live = !ws.dry && !ws.overflow;
  if (live) {
    r.chk(launch_gather_rows(sid_src, r.A(m.emb_g), c.n_speakers, g_src, B, gin, s), "emb_g");
    r.chk(launch_gather_rows(sid_tgt, r.A(m.emb_g), c.n_speakers, g_tgt, B, gin, s), "emb_g");
    r.chk(launch_mask_u8(y_lengths, y_mask, B, T, s), "y_mask");
  }
  const size_t mark = ws.cur;
  run_posterior(r, B, T, ext(y, c.spec_channels, T), y_lengths, g_src, noise, ext(z, inter, T), MQ, LQ);
  const size_t after_post = ws.cur;
  ws.cur = mark;  // the three stages run one after another on one stream: they share scratch
  const long n = (long)B * inter * T;
  if (live && r.ok()) r.chk(hipMemcpyAsync(z_p, z, n * sizeof(float), hipMemcpyDeviceToDevice, s), "z_p copy");
  run_flow(r, B, T, ext(z_p, inter, T), g_src, y_lengths, false);
  if (live && r.ok()) r.chk(hipMemcpyAsync(z_hat, z_p, n * sizeof(float), hipMemcpyDeviceToDevice, s), "z_hat copy");
  ws.cur = mark;
  run_flow(r, B, T, ext(z_hat, inter, T), g_tgt, y_lengths, true);
  const size_t after_flow = ws.cur;
  ws.cur = mark;
  // dec(z_hat * y_mask, g_tgt): z_hat is already masked by the flow's last update on x1 ... but x0
  // passes through unmasked only if the input was unmasked; z is masked, so z_hat * mask == z_hat.
  run_gen(r, B, T, ext(z_hat, inter, T), y_lengths, g_tgt, o_hat);
  ws.cur = std::max(ws.cur, std::max(after_post, after_flow));
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "voice conversion workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

int64_t vsp_voice_conversion_workspace_bytes(const vsp_ctx* ctx, int B, int T) {
  if (!ctx || B <= 0 || T <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  vc_impl(const_cast<vsp_ctx*>(ctx), nullptr, ws, B, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
          nullptr, nullptr, nullptr, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_voice_conversion(vsp_ctx* ctx, void* stream, int B, int T, const float* y, const int64_t* y_lengths,
                         const int64_t* sid_src, const int64_t* sid_tgt, const float* noise, float* o_hat,
                         uint8_t* y_mask, float* z, float* z_p, float* z_hat, float* m_q, float* logs_q,
                         void* workspace, int64_t workspace_bytes) {
  int rc = check_vc(ctx);
  if (rc) return rc;
  if (B <= 0 || T <= 0 || !y || !y_lengths || !sid_src || !sid_tgt || !noise || !o_hat || !y_mask || !z || !z_p ||
      !z_hat || !workspace)
    return ctx->fail(VSP_ERR_ARG, "vsp_voice_conversion: null or non-positive argument");
  const int64_t need = vsp_voice_conversion_workspace_bytes(ctx, B, T);
  if (workspace_bytes < need)
    return ctx->fail(VSP_ERR_WORKSPACE, "voice conversion workspace too small: %lld < %lld bytes",
                     (long long)workspace_bytes, (long long)need);
  Ws ws(workspace, (size_t)workspace_bytes, false);
  return vc_impl(ctx, (hipStream_t)stream, ws, B, T, y, y_lengths, sid_src, sid_tgt, noise, o_hat, y_mask, z, z_p, z_hat,
                 m_q, logs_q);
}

int64_t vsp_generator_workspace_bytes(const vsp_ctx* ctx, int B, int T) {
  if (!ctx || B <= 0 || T <= 0) return VSP_ERR_ARG;
  Ws ws(nullptr, 0, true);
  Run r{const_cast<vsp_ctx*>(ctx), nullptr, ws};
  run_gen(r, B, T, T3{}, nullptr, nullptr, nullptr);
  return (int64_t)ws.cur;
}

int vsp_generator(vsp_ctx* ctx, void* stream, int B, int T, const float* z, const float* g, float* o, void* workspace,
                  int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || T <= 0 || !z || !g || !o || !workspace) return ctx->fail(VSP_ERR_ARG, "vsp_generator: bad argument");
  if (workspace_bytes < vsp_generator_workspace_bytes(ctx, B, T))
    return ctx->fail(VSP_ERR_WORKSPACE, "generator workspace too small (need %lld bytes)",
                     (long long)vsp_generator_workspace_bytes(ctx, B, T));
  Ws ws(workspace, (size_t)workspace_bytes, false);
  Run r{ctx, (hipStream_t)stream, ws};
  run_gen(r, B, T, ext(z, ctx->cfg.inter_channels, T), nullptr, g, o);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "generator workspace too small (need %zu bytes)", ws.cur);
  return r.rc;
}

// Receptive field of the generator in input frames (one side), from the configuration: walking back from conv_post,
// every ResBlock1 stage adds max_k sum_d ((k-1) d / 2 + (k-1) / 2) positions at its rate, every transposed conv maps
// w positions to ceil((w + (k + s) / 2 - 1) / s), conv_pre adds 3.  (14 for configs/config.json; measured 12.33.)
int vsp_generator_halo_frames(const vsp_ctx* ctx) {
  if (!ctx) return VSP_ERR_ARG;
  const vsp_config& c = ctx->cfg;
  long w = 3;   // conv_post k7
  for (int i = c.n_upsamples - 1; i >= 0; --i) {
    long rb = 0;
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      long acc = 0;
      const int k = c.resblock_kernel_sizes[j];
      for (int d = 0; d < c.n_resblock_dilations; ++d) acc += (long)(k - 1) * c.resblock_dilation_sizes[j][d] / 2 + (k - 1) / 2;
      rb = std::max(rb, acc);
    }
    w += rb;
    const int s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    w = (w + (k + s) / 2 - 1 + s - 1) / s;
  }
  return (int)(w + 3);   // conv_pre k7
}

int vsp_generator_frame_dependence(const vsp_ctx* ctx, int* back, int* fwd) {
  if (!ctx || !back || !fwd) return VSP_ERR_ARG;
  generator_frame_dependence(ctx->cfg, ctx->model.g_pre.K > 0 ? ctx->model.g_pre.K : 7, ctx->model.post_k, *back, *fwd);
  return VSP_OK;
}

int64_t vsp_generator_stream_workspace_bytes(const vsp_ctx* ctx, int B, int chunk_frames) {
  if (!ctx || B <= 0 || chunk_frames <= 0) return VSP_ERR_ARG;
  const int span = chunk_frames + 2 * vsp_generator_halo_frames(ctx);
  const int64_t g = vsp_generator_workspace_bytes(ctx, B, span);
  if (g < 0) return g;
  return g + (((int64_t)B * span * total_upsample(ctx->cfg) * (int64_t)sizeof(float) + 255) / 256) * 256;
}

// Streamed vocoder (BASELINE config 5): the waveform of frames [f0, f1) of z [B][inter][T], computed from those frames
// plus the halo on both sides (zero padding only at the true ends) -- bit-identical to the same samples of one
// vsp_generator call over all T frames.  o_chunk [B][1][(f1 - f0) * prod(upsample_rates)], contiguous.
int vsp_generator_stream_chunk(vsp_ctx* ctx, void* stream, int B, int T, const float* z, const float* g, int f0, int f1,
                               float* o_chunk, void* workspace, int64_t workspace_bytes) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (B <= 0 || T <= 0 || !z || !g || !o_chunk || !workspace || f0 < 0 || f1 <= f0 || f1 > T)
    return ctx->fail(VSP_ERR_ARG, "vsp_generator_stream_chunk: bad argument");
  const int halo = vsp_generator_halo_frames(ctx);
  const int lo = std::max(0, f0 - halo), hi = std::min(T, f1 + halo), span = hi - lo;
  const long up = total_upsample(ctx->cfg);
  const int64_t gen_bytes = vsp_generator_workspace_bytes(ctx, B, span);
  const int64_t scratch = (((int64_t)B * span * up * (int64_t)sizeof(float) + 255) / 256) * 256;
  if (gen_bytes < 0 || workspace_bytes < gen_bytes + scratch)
    return ctx->fail(VSP_ERR_WORKSPACE, "stream chunk workspace too small (need %lld bytes)", (long long)(gen_bytes + scratch));
  float* o_span = static_cast<float*>(workspace);
  Ws ws(static_cast<char*>(workspace) + scratch, (size_t)(workspace_bytes - scratch), false);
  Run r{ctx, (hipStream_t)stream, ws};
  const int inter = ctx->cfg.inter_channels;
  // the frames [lo, hi) of every channel row: same strides as z, shifted start
  run_gen(r, B, span, T3{const_cast<float*>(z) + lo, (long)inter * T, (long)T}, nullptr, g, o_span);
  if (ws.overflow) return ctx->fail(VSP_ERR_WORKSPACE, "stream chunk workspace too small (need %zu bytes)", ws.cur);
  if (r.rc != VSP_OK) return r.rc;
  hipError_t e = hipMemcpy2DAsync(o_chunk, (size_t)(f1 - f0) * up * sizeof(float), o_span + (size_t)(f0 - lo) * up,
                                  (size_t)span * up * sizeof(float), (size_t)(f1 - f0) * up * sizeof(float), (size_t)B,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return e == hipSuccess ? VSP_OK : ctx->fail(VSP_ERR_HIP, "stream chunk copy: %s", hipGetErrorString(e));
}

int vsp_rq_spline(void* stream, int64_t n, int nb, const float* x, const float* uw, const float* uh, const float* ud,
                  int inverse, float tail_bound, float* y, float* logabsdet) {
  if (n < 0 || !x || !uw || !uh || !ud || !y || !logabsdet) return VSP_ERR_ARG;
  hipError_t e = launch_rq_spline(n, nb, x, uw, uh, ud, inverse, tail_bound, y, logabsdet, (hipStream_t)stream);
  return e == hipSuccess ? VSP_OK : (e == hipErrorInvalidValue ? VSP_ERR_UNSUPPORTED : VSP_ERR_HIP);
}

// -------------------------------------------------------------------------------------------- mel spectrogram
namespace {
// librosa.filters.mel with its defaults (htk = False, norm = 'slaney'), in double precision
bool mel_basis(int sr, int n_fft, int n_mels, double fmin, double fmax, std::vector<float>& w) {
  if (sr <= 0 || n_fft < 2 || (n_fft & 1) || n_mels < 1) return false;
  if (fmax <= 0.0) fmax = sr / 2.0;
  if (fmin < 0.0 || fmax <= fmin) return false;
  const int nf = n_fft / 2 + 1;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
  auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
  std::vector<double> mel_f(n_mels + 2);
  const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
  for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
  w.assign((size_t)n_mels * nf, 0.f);
  for (int m = 0; m < n_mels; ++m) {
    const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
    for (int k = 0; k < nf; ++k) {
      const double f = (sr / 2.0) * k / (nf - 1);
      const double lower = (f - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - f) / (mel_f[m + 2] - mel_f[m + 1]);
      const double v = std::max(0.0, std::min(lower, upper));
      w[(size_t)m * nf + k] = (float)(v * enorm);
    }
  }
  return true;
}
}  // namespace

int vsp_mel_filterbank(int sampling_rate, int n_fft, int n_mels, float fmin, float fmax, float* basis_host) {
  if (!basis_host) return VSP_ERR_ARG;
  std::vector<float> w;
  if (!mel_basis(sampling_rate, n_fft, n_mels, fmin, fmax, w)) return VSP_ERR_ARG;
  std::memcpy(basis_host, w.data(), w.size() * sizeof(float));
  return VSP_OK;
}

int vsp_spec_to_mel(void* stream, int B, int T, int n_fft, int n_mels, int sampling_rate, float fmin, float fmax,
                    const float* spec, float* mel) {
  if (!spec || !mel || B < 0 || T < 0) return VSP_ERR_ARG;
  std::vector<float> w;
  if (!mel_basis(sampling_rate, n_fft, n_mels, fmin, fmax, w)) return VSP_ERR_ARG;
  if (B == 0 || T == 0) return VSP_OK;
  const int nf = n_fft / 2 + 1;
  std::vector<int> lo(n_mels), hi(n_mels);
  for (int m = 0; m < n_mels; ++m) {
    int a = nf, b = 0;
    for (int k = 0; k < nf; ++k)
      if (w[(size_t)m * nf + k] != 0.f) { a = std::min(a, k); b = k + 1; }
    lo[m] = a < b ? a : 0; hi[m] = a < b ? b : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  void *dw = nullptr, *dr = nullptr;
  hipError_t e = hipMalloc(&dw, w.size() * 4);
  if (e == hipSuccess) e = hipMalloc(&dr, (size_t)2 * n_mels * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(dr, lo.data(), (size_t)n_mels * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(static_cast<int*>(dr) + n_mels, hi.data(), (size_t)n_mels * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess)
    e = launch_spec_to_mel(spec, static_cast<const float*>(dw), static_cast<const int*>(dr), static_cast<const int*>(dr) + n_mels,
                           mel, B, nf, n_mels, T, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);      // (the host vectors and the scratch die with this frame)
  if (dw) (void)hipFree(dw);
  if (dr) (void)hipFree(dr);
  return e == hipSuccess ? VSP_OK : VSP_ERR_HIP;
}

// -------------------------------------------------------------------------------------------- stand-alone vocoder operators
namespace {
struct DevBuf {           // hipMalloc'd scratch of one stand-alone call
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
// dense [Cout][Cin][K] host weights -> packed fragment image on the device; bias -> device
hipError_t upload_cl_conv(const float* w_host, const float* bias_host, int Cout, int Cin, int K, DevBuf& w, DevBuf& bias,
                          hipStream_t s) {
  std::vector<uint16_t> packed(packed_g16_halfs(Cout, Cin, K));
  pack_g16_weights(packed.data(), Cout, Cin, K, w_host);
  hipError_t e = w.alloc(packed.size() * 2);
  if (e == hipSuccess) e = bias.alloc((size_t)Cout * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(w.p, packed.data(), packed.size() * 2, hipMemcpyHostToDevice, s);
  std::vector<float> scaled(Cout, 0.f);                  // (kernels.h: these kernels take the bias * G16_WSCALE)
  if (bias_host) for (int i = 0; i < Cout; ++i) scaled[i] = bias_host[i] * G16_WSCALE;
  if (e == hipSuccess) e = hipMemcpyAsync(bias.p, scaled.data(), (size_t)Cout * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);     // the host vectors die with this frame
  return e;
}
ClConvArgs cl_conv_args(const float* x, int T, int Cin, int Cout, int K, int dil, const DevBuf& w, const DevBuf& bias,
                        float in_slope, const float* res, int terms, float* out) {
  ClConvArgs a;
  std::memset(&a, 0, sizeof a);
  a.x = x; a.x_bs = (long)T * Cin; a.x_ts = Cin;
  a.wh = static_cast<const uint16_t*>(w.p); a.bias = static_cast<const float*>(bias.p);
  a.out = out; a.o_bs = (long)T * Cout; a.o_ts = Cout;
  a.res = res; a.r_bs = (long)T * Cout; a.r_ts = Cout;
  a.Cin = Cin; a.Cout = Cout; a.K = K; a.dil = dil; a.pad = dil * (K - 1) / 2;
  a.T_in = T; a.Nq = T; a.T_store = T;
  a.in_act = 1; a.in_slope = in_slope;
  a.acc_prev = 0; a.div = 1.f; a.phases = 1; a.ups_p = 0;
  a.terms = terms;
  return a;
}
int op_rc(hipError_t e) { return e == hipSuccess ? VSP_OK : (e == hipErrorInvalidValue ? VSP_ERR_UNSUPPORTED : VSP_ERR_HIP); }
}  // namespace

int vsp_cl_conv1d(void* stream, int B, int T, int Cin, int Cout, int K, int dilation, const float* x, const float* w_host,
                  const float* bias_host, float in_slope, const float* res, int terms, float* out) {
  if (!x || !w_host || !out || B < 0 || T < 0 || (terms != 1 && terms != 3)) return VSP_ERR_ARG;
  if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || K < 1 || !(K & 1) || dilation < 1 || (K - 1) * dilation > 64 ||
      (size_t)T * std::max(Cin, Cout) * 4 >= (size_t)1 << 31)
    return VSP_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VSP_OK;
  hipStream_t s = (hipStream_t)stream;
  DevBuf w, bias;
  hipError_t e = upload_cl_conv(w_host, bias_host, Cout, Cin, K, w, bias, s);
  if (e == hipSuccess) e = launch_g16_conv(cl_conv_args(x, T, Cin, Cout, K, dilation, w, bias, in_slope, res, terms, out), B, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return op_rc(e);
}

int vsp_conv1d(void* stream, int B, int T, int Cin, int Cout, int K, int dilation, const float* x, const float* w_host,
               const float* bias_host, const int64_t* lengths, int mask_in, int in_act, float in_slope, int act,
               const float* res, int mask_out, int split_f16, float* out) {
  if (!x || !w_host || !out || B < 0 || T < 0 || act < 0 || act > 2 || ((mask_in || mask_out) && !lengths) || (act == 2 && res))
    return VSP_ERR_ARG;
  if (Cin <= 0 || Cout <= 0 || K < 1 || !(K & 1) || dilation < 1 || (K - 1) * dilation + 3 > CONV_HALO ||
      (act == 2 && Cout % 64) || ((T & 3) && T != 1))   // (rows of T floats must stay 16-byte aligned for the vector staging;
                                                        //  T = 1: the one-time-step projections, conv_t1_gemv)
    return VSP_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VSP_OK;
  hipStream_t s = (hipStream_t)stream;
  // the gate reads its tanh / sigmoid halves from interleaved 32-row tiles (weights.cpp packs WN in_layers the same way)
  std::vector<float> wd((size_t)Cout * Cin * K), bd(Cout, 0.f);
  for (int r = 0; r < Cout; ++r) {
    int src = r;
    if (act == 2) { const int tile = r / 32, in = r % 32; src = (tile & 1) * (Cout / 2) + (tile >> 1) * 32 + in; }
    std::memcpy(&wd[(size_t)r * Cin * K], w_host + (size_t)src * Cin * K, (size_t)Cin * K * sizeof(float));
    if (bias_host) bd[r] = bias_host[src];
  }
  std::vector<float> packed(packed_conv_floats(Cout, Cin, K));
  if (split_f16) pack_conv_weights_f16s(packed.data(), Cout, Cin, K, wd.data());
  else pack_conv_weights(packed.data(), Cout, Cin, K, wd.data());
  DevBuf w, bias;
  hipError_t e = w.alloc(packed.size() * 4);
  if (e == hipSuccess) e = bias.alloc((size_t)Cout * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(w.p, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(bias.p, bd.data(), (size_t)Cout * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return op_rc(e);
  const int rows_out = act == 2 ? Cout / 2 : Cout;
  ConvArgs a;
  std::memset(&a, 0, sizeof a);
  a.x = x; a.x_bs = (long)Cin * T; a.x_cs = T;
  a.wp = static_cast<const float*>(w.p); a.bias = static_cast<const float*>(bias.p);
  a.out = out; a.o_bs = (long)rows_out * T; a.o_cs = T;
  a.res = res; a.r_bs = (long)rows_out * T; a.r_cs = T;
  a.lengths = lengths;
  a.Cin = Cin; a.M = Cout; a.K = K; a.dil = dilation; a.pad = dilation * (K - 1) / 2;
  a.T_in = T; a.Nq = T; a.nchunks = (Cin + CONV_CK - 1) / CONV_CK;
  a.in_mask = mask_in ? 1 : 0; a.in_act = in_act ? 1 : 0; a.in_slope = in_slope;
  a.act = act; a.alpha = 1.f; a.div = 1.f; a.mask_post = mask_out ? 1 : 0;
  a.f16s = split_f16 ? 1 : 0;
  if (split_f16 == 2) {
    // the column-tile form (conv_cols.hip: every output row of a 64-column tile in one block) as a stand-alone operator:
    // the same weights in 16x16x32 A-fragment order; refused where that kernel does not apply
    std::vector<uint16_t> wgh(packed_g16_halfs(Cout, Cin, 1));
    DevBuf wgd;
    if (K != 1 || Cin % 32 || Cout % 16) return VSP_ERR_UNSUPPORTED;
    pack_g16_weights(wgh.data(), Cout, Cin, 1, wd.data());
    e = wgd.alloc(wgh.size() * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(wgd.p, wgh.data(), wgh.size() * 2, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return op_rc(e);
    a.wg = static_cast<const uint16_t*>(wgd.p);
    if (!conv_cols_supported(a)) return VSP_ERR_UNSUPPORTED;
    e = launch_conv_cols(a, B, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return op_rc(e);
  }
  e = launch_conv(a, B, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return op_rc(e);
}

int vsp_cl_resblock(void* stream, int B, int T, int C, int K, int n_pairs, const int* dilations, const float* x,
                    const float* const* w_host, const float* const* bias_host, int mode, int terms, float* out) {
  if (!x || !w_host || !bias_host || !dilations || !out || x == out || B < 0 || T < 0 || n_pairs < 1 || n_pairs > 8 ||
      mode < 0 || mode > 2 || (terms != 1 && terms != 3))
    return VSP_ERR_ARG;
  if (C <= 0 || C % 32 || K < 1 || !(K & 1) || (size_t)T * C * 4 >= (size_t)1 << 31) return VSP_ERR_UNSUPPORTED;
  for (int p = 0; p < n_pairs; ++p) {
    if (dilations[p] < 1 || (K - 1) * dilations[p] > 64) return VSP_ERR_UNSUPPORTED;
    if (mode == 1 && !g16_pair_supported(C, K, dilations[p]) && !g16_pp_supported(C, K, dilations[p], terms)) return VSP_ERR_UNSUPPORTED;
    if (!w_host[2 * p] || !w_host[2 * p + 1]) return VSP_ERR_ARG;
  }
  if (mode == 2 && (n_pairs > 3 || !g16_chain_supported(C, K, dilations, n_pairs))) return VSP_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VSP_OK;
  hipStream_t s = (hipStream_t)stream;
  std::vector<DevBuf> w(2 * n_pairs), bias(2 * n_pairs);
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 * n_pairs && e == hipSuccess; ++i) e = upload_cl_conv(w_host[i], bias_host[i], C, C, K, w[i], bias[i], s);
  const size_t el = (size_t)B * T * C;
  DevBuf t1, ya, yb;
  if (e == hipSuccess && mode != 2) e = t1.alloc(el * 4);
  if (e == hipSuccess && mode != 2) e = ya.alloc(el * 4);
  if (e == hipSuccess && mode != 2) e = yb.alloc(el * 4);
  DevBuf timg;
  if (e == hipSuccess && mode == 0 && terms == 3 && K >= 3) {
    e = timg.alloc((size_t)B * cl_img_halfs(C, T) * 2);
    if (e == hipSuccess) e = launch_cl_img_zero_pads(static_cast<uint16_t*>(timg.p), B, C, T, s);
  }
  if (e != hipSuccess) return op_rc(e);
  if (mode == 2) {
    ClChainArgs a;
    std::memset(&a, 0, sizeof a);
    a.x = x; a.x_bs = (long)T * C; a.out = out; a.o_bs = (long)T * C;
    for (int i = 0; i < 2 * n_pairs; ++i) { a.w[i] = static_cast<const uint16_t*>(w[i].p); a.b[i] = static_cast<const float*>(bias[i].p); }
    for (int p = 0; p < n_pairs; ++p) a.dil[p] = dilations[p];
    a.np = n_pairs; a.C = C; a.K = K; a.T = T; a.slope = 0.1f; a.acc_prev = 0; a.div = 1.f; a.terms = terms;
    if (const char* ev = getenv("VSP_CHAIN_RING")) a.ring = atoi(ev) != 0;   // (read per call: the test API has no context)
    e = launch_g16_chain(a, B, s);
  } else {
    // the running y ping-pongs between two buffers (a tile reads halo rows its neighbour writes)
    const float* yin = x;
    for (int p = 0; p < n_pairs && e == hipSuccess; ++p) {
      float* yout = p == n_pairs - 1 ? out : static_cast<float*>((p & 1) ? yb.p : ya.p);
      if (mode == 1) {
        ClPairArgs a;
        std::memset(&a, 0, sizeof a);
        a.x = yin; a.x_bs = (long)T * C; a.out = yout; a.o_bs = (long)T * C;
        a.w1h = static_cast<const uint16_t*>(w[2 * p].p); a.w2h = static_cast<const uint16_t*>(w[2 * p + 1].p);
        a.b1 = static_cast<const float*>(bias[2 * p].p); a.b2 = static_cast<const float*>(bias[2 * p + 1].p);
        a.C = C; a.K = K; a.dil = dilations[p]; a.T = T; a.slope = 0.1f; a.acc_prev = 0; a.div = 1.f; a.terms = terms;
        if (const char* ev = getenv("VSP_PAIR")) a.ring = !strcmp(ev, "ring");     // (read per call: the test API has no context)
        if (const char* ev = getenv("VSP_RW64")) a.rw64 = atoi(ev) != 0;
        e = launch_g16_pair(a, B, s);
      } else {
        // one launch per convolution; the intermediate as an operand image where the kernels take one (terms 3, K >= 3):
        // the path the >= 128-channel stages of the generator run
        const bool img = terms == 3 && K >= 3 && timg.p;
        ClConvArgs a1 = cl_conv_args(yin, T, C, C, K, dilations[p], w[2 * p], bias[2 * p], 0.1f, nullptr, terms,
                                     img ? nullptr : static_cast<float*>(t1.p));
        ClConvArgs a2 = cl_conv_args(static_cast<const float*>(t1.p), T, C, C, K, 1, w[2 * p + 1], bias[2 * p + 1], 0.1f, yin,
                                     terms, yout);
        if (img) {
          a1.o_img = static_cast<uint16_t*>(timg.p); a1.oi_bs = (long)cl_img_halfs(C, T); a1.oi_tpad = cl_img_tpad(T); a1.oi_slope = 0.1f;
          a2.x_img = a1.o_img; a2.xi_bs = a1.oi_bs; a2.xi_tpad = a1.oi_tpad;
        }
        e = launch_g16_conv(a1, B, s);
        if (e == hipSuccess) e = launch_g16_conv(a2, B, s);
      }
      yin = yout;
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return op_rc(e);
}

// -------------------------------------------------------------------------------------------- profiling
int vsp_profile_enable(vsp_ctx* ctx, int on) {
  if (!ctx) return VSP_ERR_ARG;
  if (on && !ctx->prof_on) {          // a fresh measurement: forget whatever an earlier one left unread
    ctx->ev_used = 0;
    for (int c = 0; c < VSP_PROF_CLASSES; ++c) {
      ctx->prof_launches[c] = 0;
      ctx->prof_flops[c] = ctx->prof_bytes[c] = ctx->prof_bytes_ext[c] = ctx->prof_bytes_moved[c] = 0.0;
    }
  }
  ctx->prof_on = on != 0;
  return VSP_OK;
}

int vsp_profile_read_class(vsp_ctx* ctx, int cls, int64_t* launches, double* total_ms, double* total_flops,
                           double* total_bytes, double* total_bytes_ext, double* total_bytes_moved, int reset) {
  if (!ctx || cls < 0 || cls >= VSP_PROF_CLASSES || !launches || !total_ms || !total_flops || !total_bytes)
    return ctx ? ctx->fail(VSP_ERR_ARG, "vsp_profile_read_class: bad argument") : VSP_ERR_ARG;
  double ms = 0.0;
  for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
    if (ctx->ev_cls[i / 2] != cls) continue;
    hipError_t e = hipEventSynchronize(ctx->ev_pool[i + 1]);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, ctx->ev_pool[i], ctx->ev_pool[i + 1]);
    if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "profile events: %s", hipGetErrorString(e));
    ms += t;
  }
  *launches = ctx->prof_launches[cls];
  *total_ms = ms;
  *total_flops = ctx->prof_flops[cls];
  *total_bytes = ctx->prof_bytes[cls];
  if (total_bytes_ext) *total_bytes_ext = ctx->prof_bytes_ext[cls];
  if (total_bytes_moved) *total_bytes_moved = ctx->prof_bytes_moved[cls];
  if (reset) {
    // the event pool is shared: drop every class's events only when the LAST class has been read; a reset of one
    // class zeroes its counters and marks its pairs as consumed
    ctx->prof_launches[cls] = 0;
    ctx->prof_flops[cls] = ctx->prof_bytes[cls] = ctx->prof_bytes_ext[cls] = ctx->prof_bytes_moved[cls] = 0.0;
    bool any = false;
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
      if (ctx->ev_cls[i / 2] == cls) ctx->ev_cls[i / 2] = -1;
      else if (ctx->ev_cls[i / 2] >= 0) any = true;
    }
    if (!any) ctx->ev_used = 0;
  }
  return VSP_OK;
}

int vsp_profile_read_families(vsp_ctx* ctx, int cls, int max_families, int* family, int64_t* launches, double* total_ms,
                              double* total_flops, double* total_bytes, double* total_bytes_moved) {
  if (!ctx || cls < 0 || cls >= VSP_PROF_CLASSES || max_families < 0 || !family || !launches || !total_ms || !total_flops ||
      !total_bytes)
    return ctx ? ctx->fail(VSP_ERR_ARG, "vsp_profile_read_families: bad argument") : VSP_ERR_ARG;
  int n = 0;
  for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
    if (ctx->ev_cls[i / 2] != cls) continue;
    hipError_t e = hipEventSynchronize(ctx->ev_pool[i + 1]);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, ctx->ev_pool[i], ctx->ev_pool[i + 1]);
    if (e != hipSuccess) return ctx->fail(VSP_ERR_HIP, "profile events: %s", hipGetErrorString(e));
    const int f = ctx->ev_fam[i / 2];
    int k = 0;
    while (k < n && family[k] != f) ++k;
    if (k == n) {
      if (n == max_families) continue;
      family[n] = f; launches[n] = 0; total_ms[n] = total_flops[n] = total_bytes[n] = 0.0;
      if (total_bytes_moved) total_bytes_moved[n] = 0.0;
      ++n;
    }
    launches[k] += 1;
    total_ms[k] += t;
    total_flops[k] += ctx->ev_flops[i / 2];
    total_bytes[k] += ctx->ev_bytes[i / 2];
    if (total_bytes_moved) total_bytes_moved[k] += ctx->ev_moved[i / 2];
  }
  return n;
}

int vsp_profile_read(vsp_ctx* ctx, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes,
                     int reset) {
  return vsp_profile_read_class(ctx, VSP_PROF_GENERATOR, launches, total_ms, total_flops, total_bytes, nullptr, nullptr, reset);
}

}  // extern "C"
