// Vocoder convolution on the f16 matrix core with SPLIT operands -- fp32-accurate at ~4.6x the
// native f32 MFMA rate (measured on MI355X, profiles/r01_exp_split_f16.log: v_mfma_f32_32x32x16_f16
// 2.06 PFLOP/s vs v_mfma_f32_32x32x2_f32 0.148; error of the 3-term split 6e-8 * sum|ab|, i.e. no
// worse than an f32 fmaf chain).
//
//   x = xh + xl * 2^-11,  xh = f16(x),  xl = f16((x - xh) * 2^11)      (exact residual, scaled so the
//   w = wh + wl * 2^-11                                                  low part never underflows)
//   x*w ~= xh*wh + (xh*wl + xl*wh) * 2^-11        (dropped xl*wl term <= 2^-24 |x w|)
//   -> two f32 accumulators per output tile: HH and CROSS; result = HH + CROSS * 2^-11.
//
// Layout: the vocoder's activations live CHANNELS-LAST in HBM, [B][T][C] fp32, so that
//   * a block's input window (rows t0-pad .. t0+BT+halo, one ci chunk) is read with 16-byte loads
//     that cover whole 128/256-byte rows (perfectly coalesced), activated (leaky-relu), split and
//     written once to LDS as two f16 images [row][CKC+8] (row stride 80/144 B: conflict-free
//     ds_read_b128 for the A fragments: lane = time row, 8 consecutive input channels);
//   * GEMM orientation M = time, N = output channel, K = (tap, ci): the D tile has the output
//     channel on the lane, so every store instruction writes two full contiguous rows;
//   * weights are pre-split and pre-packed on the host in B-fragment order ([phase][tap][k-step]
//     [n-tile][lane][8 f16]); a wave fetches a fragment with one 1 KiB global_load_dwordx4 (L2).
// ConvTranspose1d runs as `phases` = stride independent polyphase GEMMs (blockIdx.y), each writing
// rows n = s*q + r - p, which are again whole contiguous rows.
//
// Reference call sites: modules.py:210-223 (ResBlock1 convs), models.py:255-257, 276-285 (ups,
// resblock averaging).
#include "kernels.h"

#include <cstdlib>
#include <cstring>

namespace vsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int CL_HALO = 64;  // max (K-1)*dil

// VSP_STAMPS (diagnostic build, tools/stamps.py): wave 0 of every 61st block of the VSP_STAMP_LAUNCH-th launch
// records wall-clock stamps (s_memrealtime, 100 MHz) at its phase boundaries; read back by vsp_debug_stamps_cl.
#ifdef VSP_STAMPS
constexpr int CL_NSTAMP = 160, CL_NSAMPLE = 128;
__device__ unsigned long long g_cl_stamps[CL_NSAMPLE][CL_NSTAMP];
__device__ unsigned long long g_cl_cycles[CL_NSAMPLE][CL_NSTAMP];   // s_memtime (shader clock) at the same points
__device__ unsigned g_cl_stamp_count;
#define CL_STAMP()                                                          \
  do {                                                                      \
    if (stamp_slot >= 0 && stamp_n < CL_NSTAMP && lane == 0) {              \
      g_cl_stamps[stamp_slot][stamp_n] = __builtin_amdgcn_s_memrealtime();  \
      g_cl_cycles[stamp_slot][stamp_n] = __builtin_readcyclecounter();      \
    }                                                                       \
    ++stamp_n;                                                              \
  } while (0)
#else
#define CL_STAMP() ((void)0)
#endif

size_t packed_cl_halfs(int Cout, int Cin, int K, int phases) {
  return (size_t)phases * K * (Cin / 16) * (Cout / 32) * 64 * 8;
}

// dense: W[phase][co][ci][tap] fp32 -> hi / lo f16 fragment images (each packed_cl_halfs halfs)
void pack_cl_weights(uint16_t* hi, uint16_t* lo, int Cout, int Cin, int K, int phases, const float* dense) {
  const int nks = Cin / 16, nnt = Cout / 32;
  for (int ph = 0; ph < phases; ++ph)
    for (int co = 0; co < Cout; ++co)
      for (int ci = 0; ci < Cin; ++ci)
        for (int tap = 0; tap < K; ++tap) {
          const float w = dense[(((size_t)ph * Cout + co) * Cin + ci) * K + tap];
          const _Float16 h = (_Float16)w;
          const _Float16 l = (_Float16)((w - (float)h) * 2048.f);
          const int ks = ci / 16, hh = (ci % 16) / 8, j = ci % 8, nt = co / 32, lane = (co % 32) + 32 * hh;
          const size_t idx = (((((size_t)ph * K + tap) * nks + ks) * nnt + nt) * 64 + lane) * 8 + j;
          std::memcpy(hi + idx, &h, 2);
          std::memcpy(lo + idx, &l, 2);
        }
}

// Block = 8 waves, 256 time rows x (NT*WN*32) output channels.  Per ci-chunk the activated, split
// input window sits in LDS (one buffer); the weights stream through a double-buffered LDS ring in
// slices of G taps x one chunk (32 KiB hi+lo), fetched from L2 ONCE per block, one slice ahead: by
// LDS-DMA on the 128-column tile (GLDS), through staging registers on the small tiles.  Every wave then
// reads both operands' fragments from LDS (ds_read_b128, conflict-free); where registers allow (PF) the
// fragments of the next k-step are requested before the MFMAs of the current one.
//
// Addressing: rocprofv3 --pmc showed the first versions issuing 18-57 VALU instructions per MFMA,
// nearly all 64-bit index arithmetic and per-element bounds branches (the epilogue alone was ~1100
// VALU + 650 SALU per tile).  Every activation access therefore goes through a per-utterance BUFFER
// descriptor: the hardware range check returns 0 for rows before/after the utterance (= the conv's
// zero padding) and drops stores outside it (tile edges, polyphase rows), so there is no bounds
// code at all; a lane's offset is (per-tile lane constant) + (wave-uniform row term) = one v_add.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// VSP_DIAG: timing-only ablation builds (tools/ablate.sh); bit 0 no MFMA, bit 1 no weight-slice loads,
// bit 2 no activation loads, bit 3 no epilogue memory traffic, bit 4 no barriers.  0 in the product build.
#ifndef VSP_DIAG
#define VSP_DIAG 0
#endif
#if VSP_DIAG & 16
#define CL_SYNC() ((void)0)      // timing-only: no barriers
#else
#define CL_SYNC() __syncthreads()
#endif

// TERMS = 3: fp32-accurate split product (default).  TERMS = 1: plain f16 operands (hi images only,
// one MFMA per product) -- the opt-in reduced-precision mode VSP_GENERATOR=f16 (BASELINE config 3's
// low-precision variant), NOT used by bench.py or the parity gates.
// GLDS: the weight slices go global -> LDS directly (global_load_lds_dwordx4: one 1 KiB fragment block per
// wave-instruction, exactly the lane-linear layout of the ring), one slice ahead; no staging registers,
// no ds_write -- the registers this frees pay for the fragment double-buffer (PF) of the 128-column tile.
template <int MT, int NT, int WM, int WN, int CKC, int G, bool PF, int WD, int TERMS, bool GLDS = false>
__global__ void __launch_bounds__(64 * WM * WN) cl_conv_f16s(ClConvArgs a) {
  constexpr int BT = 32 * MT * WM;
  constexpr int RS = CKC + 8;                 // LDS row stride of the activation images (halfs)
  constexpr int NTH = 64 * WM * WN;
  constexpr int C4 = CKC / 4;                 // float4 per staged activation row
  constexpr int ROWS_PER_U = NTH / C4;        // rows covered by one staging sweep of the block
  constexpr int NL = (BT + CL_HALO + ROWS_PER_U - 1) / ROWS_PER_U;
  constexpr int WMAX = NL * ROWS_PER_U;       // staged rows (>= BT + halo, whole sweeps)
  constexpr int KS = CKC / 16;                // k-steps per tap and chunk
  constexpr int NTB = NT * WN;                // 32-channel output tiles per block
  constexpr int XIMG = WMAX * RS;             // halfs per activation image
  constexpr int WIMG = G * KS * NTB * 64 * 8; // halfs per weight-slice image
  constexpr int NWV = NTH / 64;
  constexpr int NBLK = (TERMS == 3 ? 2 : 1) * G * KS * NTB;   // 1-KiB fragment blocks per slice (hi [+ lo])
  constexpr int NWL = (NBLK + NWV - 1) / NWV; // blocks per wave per slice
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const Xh = lds;                   // [WMAX][RS] hi, then lo
  _Float16* const Wb = lds + 2 * XIMG;        // 2 x (hi image, lo image)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int nnt = a.Cout >> 5, nks = a.Cin >> 4;
  const int ncb = nnt / NTB;
  // XCD-aware block numbering (workgroup ids go round-robin over the 8 XCDs): XCD k gets the k-th contiguous eighth
  // of the sequence (utterance, time tile, column group / phase), column group fastest -- the blocks that read the
  // same input window (all column groups of a time tile) and the neighbouring tiles (shared halo) share an L2
  const int gy = gridDim.y, gx = gridDim.x;
  const int nwg = gx * gy * gridDim.z, orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int xcd = orig & 7, qd = nwg >> 3, rem = nwg & 7;
  const int id = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (orig >> 3);
  const int by = id % gy, bx = (id / gy) % gx, b = id / (gy * gx);
  const int t0 = bx * BT;
  const int ph = by / ncb, cb = by - ph * ncb;
  const int row0 = wm * MT * 32;
  const int nchunks = a.Cin / CKC;
  const int ns = (a.K + G - 1) / G;           // weight slices per chunk
  const int nsteps = nchunks * ns;

  // per-utterance buffer descriptors (wave-uniform)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)b * a.x_bs, 0, a.T_in * a.x_ts * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      a.out + (size_t)b * a.o_bs, 0, a.T_store * a.o_ts * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res : a.out) + (size_t)b * (a.res ? a.r_bs : a.o_bs), 0,
      a.T_store * (a.res ? a.r_ts : a.o_ts) * 4, 0x00020000);

  // accumulators start at the bias (lane = output channel in the D layout)
  f32x16 hh[MT][NT], cr[MT][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float bias = a.bias ? a.bias[(cb * NTB + wn * NT + nt) * 32 + l31] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[mt][nt][r] = bias; cr[mt][nt][r] = 0.f; }
  }

#ifdef VSP_STAMPS
  int stamp_slot = -1, stamp_n = 0;
  if (wave == ((a.terms >> 12) & 7) && (blockIdx.x + 7 * blockIdx.z) % 61 == 3 && blockIdx.y == 0 && (a.terms & 0x100)) {
    unsigned sl_ = 0;
    if (lane == 0) sl_ = atomicAdd(&g_cl_stamp_count, 1u);
    sl_ = __builtin_amdgcn_readfirstlane(sl_);
    stamp_slot = sl_ < (unsigned)CL_NSAMPLE ? (int)sl_ : -1;
  }
  CL_STAMP();                                   // 0: start
#endif
  const uint4* WHg = reinterpret_cast<const uint4*>(a.wh);
  const uint4* WLg = reinterpret_cast<const uint4*>(a.wl);

  // ---- activation staging (issue early / convert + write late); per-lane parts computed once
  const int st_row = tid / C4, st_c4 = tid % C4;
  const int st_voff = (st_row * a.x_ts + 4 * st_c4) * 4;                   // bytes
  const int st_loff = st_row * RS + 4 * st_c4;                             // halfs
  u32x4 sv[NL];
  const float slope = a.in_act ? a.in_slope : 1.f;
  auto x_issue = [&](int chunk) {
    const int base = ((t0 - a.pad) * a.x_ts + chunk * CKC) * 4;            // uniform, may be negative
#pragma unroll
    for (int u = 0; u < NL; ++u)
      sv[u] = (VSP_DIAG & 4) ? u32x4{1u, 2u, 3u, 4u}
                             : __builtin_amdgcn_raw_buffer_load_b128(rx, st_voff + (base + u * ROWS_PER_U * a.x_ts * 4), 0, 0);
  };
  auto x_write = [&]() {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      // two elements at a time so that the multiplies / subtract / converts can use the packed forms
      // (v_pk_mul_f32, v_pk_add_f32, v_cvt_pk_f16_f32); leaky-relu = max(x, slope*x), 0 <= slope <= 1
      f16x4 eh, el;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f32x2 x = {__uint_as_float(k == 0 ? sv[u].x : sv[u].z), __uint_as_float(k == 0 ? sv[u].y : sv[u].w)};
        const f32x2 y = x * slope;
        asm("v_max_f32 %0, %1, %2" : "=v"(x.x) : "v"(x.x), "v"(y.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(x.y) : "v"(x.y), "v"(y.y));
        const f16x2 xh = __builtin_convertvector(x, f16x2);
        const f32x2 back = __builtin_convertvector(xh, f32x2);
        const f16x2 xl = __builtin_convertvector((x - back) * 2048.f, f16x2);
        eh[2 * k] = xh.x; eh[2 * k + 1] = xh.y;
        el[2 * k] = xl.x; el[2 * k + 1] = xl.y;
      }
      _Float16* dst = Xh + st_loff + u * (ROWS_PER_U * RS);  // compile-time stride
      *reinterpret_cast<f16x4*>(dst) = eh;
      if constexpr (TERMS == 3) *reinterpret_cast<f16x4*>(dst + XIMG) = el;
    }
  };
  // ---- weight-slice staging: slice (chunk, sl) = taps [sl*G, sl*G+G) x k-steps of the chunk x the
  //      block's NTB output tiles; a wave copies whole 1-KiB fragment blocks, block index
  //      ((img*G + g)*KS + ks)*NTB + ntl is wave-uniform
  // Slices are requested WD steps ahead of their use into WD rotating register sets (one slice is
  // 32 KiB from L2: with a single slice in flight the weight stream is latency-bound -- measured:
  // the kernel WITHOUT its MFMAs still took 70 % of the time); the LDS ring stays at two slots,
  // slice s+1 is written at the end of step s.
  static_assert(WD >= 1 && WD <= 3, "prefetch distance");
  static_assert(!GLDS || WD == 1, "the LDS-DMA ring runs one slice ahead");
  uint4 wq[WD][NWL];
  // LDS-DMA variant of w_issue + w_write: slice `step` -> ring slot `buf`
  // per-wave constants of the slice copy: block u*NWV + wave of a slice is fragment block (img, g, ks, ntl);
  // its source offset (in 16-byte units) is wslice(chunk, tap0) + wblk[u], so a step costs one 64-bit
  // multiply-add instead of four index chains (PMC: 3.4 scalar instructions per MFMA before this)
  const size_t w_tap = (size_t)nks * nnt * 64;                       // one tap
  const size_t w_base = ((size_t)ph * a.K * nks * nnt + (size_t)cb * NTB) * 64;
  size_t wblk[NWL];
  int wg[NWL];
  bool wlo[NWL];
#pragma unroll
  for (int u = 0; u < NWL; ++u) {
    const int blk = u * NWV + wave;
    const int ntl = blk % NTB, ks = (blk / NTB) % KS, g = (blk / (NTB * KS)) % G, img = blk / (NTB * KS * G);
    wblk[u] = w_base + (size_t)g * w_tap + ((size_t)ks * nnt + ntl) * 64;
    wg[u] = g;
    wlo[u] = img != 0;
  }
  auto w_dma = [&](int chunk, int sl, int buf) {
    const size_t wslice = (size_t)(sl * G) * w_tap + (size_t)chunk * KS * nnt * 64;
#pragma unroll
    for (int u = 0; u < NWL; ++u) {
      const int blk = u * NWV + wave;
      // (NBLK % NWV == 0 and one tap per slice make both tests compile-time true: straight-line issue)
      if ((NBLK % NWV == 0 || blk < NBLK) && (G == 1 || sl * G + wg[u] < a.K) && (VSP_DIAG & 2) == 0) {
        const uint4* gp = (wlo[u] ? WLg : WHg) + (wslice + wblk[u]) + lane;
        _Float16* lp = Wb + buf * 2 * WIMG + blk * 512;     // 1 KiB per fragment block, wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
      }
    }
  };
  auto w_issue_cs = [&](int chunk, int sl, uint4(&wv)[NWL]) {
    const size_t wslice = (size_t)(sl * G) * w_tap + (size_t)chunk * KS * nnt * 64;
#pragma unroll
    for (int u = 0; u < NWL; ++u) {
      const int blk = u * NWV + wave;
      wv[u] = make_uint4(0u, 0u, 0u, 0u);
      if (blk < NBLK && sl * G + wg[u] < a.K && (VSP_DIAG & 2) == 0) wv[u] = (wlo[u] ? WLg : WHg)[wslice + wblk[u] + lane];
    }
  };
  auto w_issue = [&](int step, uint4(&wv)[NWL]) {      // step-indexed form of the deeper (WD >= 2) queues
    const int chunk = step / ns;
    w_issue_cs(chunk, step - chunk * ns, wv);
  };
  auto w_write = [&](int buf, const uint4(&wv)[NWL]) {
    uint4* dst = reinterpret_cast<uint4*>(Wb + buf * 2 * WIMG) + tid;
#pragma unroll
    for (int u = 0; u < NWL; ++u)
      if (u * NWV + wave < NBLK) dst[u * NTH] = wv[u];
  };

  // ---- fragment loads (LDS): per-lane bases once, wave-uniform (tap, k-step) offsets per call
  const int xf_lane = (row0 + l31) * RS + 8 * h;               // halfs
  const int wf_lane = (wn * NT * 64 + lane) * 8;               // halfs
  auto load_frags = [&](const _Float16* Wc, int tap0, int it, f16x8(&xh)[MT], f16x8(&xl)[MT], f16x8(&wh)[NT],
                        f16x8(&wl)[NT]) {
    const int g = it / KS, ks = it % KS;
    const _Float16* px = Xh + xf_lane + ((tap0 + g) * a.dil * RS + ks * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xh[mt] = *reinterpret_cast<const f16x8*>(px + mt * 32 * RS);
      if constexpr (TERMS == 3) xl[mt] = *reinterpret_cast<const f16x8*>(px + mt * 32 * RS + XIMG);
    }
    const _Float16* pw = Wc + wf_lane + (g * KS + ks) * NTB * 512;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      wh[nt] = *reinterpret_cast<const f16x8*>(pw + nt * 512);
      if constexpr (TERMS == 3) wl[nt] = *reinterpret_cast<const f16x8*>(pw + nt * 512 + WIMG);
    }
  };
  auto mma = [&](const f16x8(&xh)[MT], const f16x8(&xl)[MT], const f16x8(&wh)[NT], const f16x8(&wl)[NT]) {
    if constexpr ((VSP_DIAG & 1) != 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(xh[mt]), "v"(xl[mt]));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(wh[nt]), "v"(wl[nt]));
      return;
    }
    // the three products are issued tile-interleaved so that the two MFMAs that chain on the same
    // CROSS accumulator are never back to back
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        hh[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[mt], wh[nt], hh[mt][nt], 0, 0, 0);
    if constexpr (TERMS == 3) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          cr[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[mt], wl[nt], cr[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          cr[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[mt], wh[nt], cr[mt][nt], 0, 0, 0);
    }
  };

  x_issue(0);
  if constexpr (GLDS) w_dma(0, 0, 0);
  else w_issue_cs(0, 0, wq[0]);
  x_write();
  if constexpr (!GLDS) w_write(0, wq[0]);
#pragma unroll
  for (int d = 1; d < WD; ++d)
    if (d < nsteps) w_issue(d, wq[d % WD]);
  if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CL_STAMP();                                   // 1: first window + slice staged
  CL_SYNC();
  CL_STAMP();                                   // 2: barrier
  f16x8 xhA[MT], xlA[MT], whA[NT], wlA[NT];
  [[maybe_unused]] f16x8 xhB[MT], xlB[MT], whB[NT], wlB[NT];
  // one step; `ld` receives slice step+WD, `st` holds slice step+1
  auto step_body = [&](int step, uint4(&ld)[NWL], const uint4(&st)[NWL]) {
    const int chunk = step / ns, sl = step - chunk * ns;
    const bool more = step + 1 < nsteps;
    const bool new_chunk = more && sl == ns - 1;
    if (step + WD < nsteps) w_issue(step + WD, ld);
    if (new_chunk) x_issue(chunk + 1);
    const _Float16* Wc = Wb + (step & 1) * 2 * WIMG;
    const int tap0 = sl * G;
    const int nit = ((a.K - tap0) < G ? (a.K - tap0) : G) * KS;
    if constexpr (PF) {
      // fragments one k-step ahead of the MFMAs (two named register sets)
      load_frags(Wc, tap0, 0, xhA, xlA, whA, wlA);
      for (int it = 0; it < nit; it += 2) {
        if (it + 1 < nit) load_frags(Wc, tap0, it + 1, xhB, xlB, whB, wlB);
        mma(xhA, xlA, whA, wlA);
        if (it + 1 < nit) {
          if (it + 2 < nit) load_frags(Wc, tap0, it + 2, xhA, xlA, whA, wlA);
          mma(xhB, xlB, whB, wlB);
        }
      }
    } else {
      // register-tight tile: the SIMD's second wave covers the LDS latency
      for (int it = 0; it < nit; ++it) {
        load_frags(Wc, tap0, it, xhA, xlA, whA, wlA);
        mma(xhA, xlA, whA, wlA);
      }
    }
    CL_STAMP();                                 // step: MFMAs done
    if (more) {
      if (new_chunk) {
        CL_SYNC();          // every wave is done reading the activation window
        x_write();
      }
      w_write((step + 1) & 1, st);  // the other ring slot: last read one barrier ago
      CL_STAMP();                               // step: next slice (and window) written
      CL_SYNC();
      CL_STAMP();                               // step: barrier
    }
  };
  // slice j lives in register set j % WD: step s loads into set s % WD (freed at the end of step
  // s-1) and stores from set (s+1) % WD
  if constexpr (WD == 1) {
    // plain loop (kept literally separate from step_body: the register allocation of this form
    // fits the 128-VGPR budget of the two-blocks-per-CU tiles)
    int chunk = 0, sl = 0;                        // (chunk, slice) of the current step, kept incrementally
    for (int step = 0; step < nsteps; ++step) {
      const bool more = step + 1 < nsteps;
      const bool last_sl = sl == ns - 1;
      const bool new_chunk = more && last_sl;
      const int chunk_n = last_sl ? chunk + 1 : chunk, sl_n = last_sl ? 0 : sl + 1;   // of step + 1
      if (more) {
        if constexpr (GLDS) w_dma(chunk_n, sl_n, (step + 1) & 1);   // that slot was last read one barrier ago
        else w_issue_cs(chunk_n, sl_n, wq[0]);
      }
      if (new_chunk) x_issue(chunk + 1);
      const _Float16* Wc = Wb + (step & 1) * 2 * WIMG;
      const int tap0 = sl * G;
      const int nit = ((a.K - tap0) < G ? (a.K - tap0) : G) * KS;
      if constexpr (PF && G == 1 && KS == 4) {
        // one tap per slice: the four k-steps are a fixed, branch-free schedule (fragments of k-step i+1
        // requested before the MFMAs of k-step i, so the compiler can count its LDS waits)
        load_frags(Wc, tap0, 0, xhA, xlA, whA, wlA);
        load_frags(Wc, tap0, 1, xhB, xlB, whB, wlB);
        __builtin_amdgcn_sched_barrier(0);
        mma(xhA, xlA, whA, wlA);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(Wc, tap0, 2, xhA, xlA, whA, wlA);
        __builtin_amdgcn_sched_barrier(0);
        mma(xhB, xlB, whB, wlB);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(Wc, tap0, 3, xhB, xlB, whB, wlB);
        __builtin_amdgcn_sched_barrier(0);
        mma(xhA, xlA, whA, wlA);
        __builtin_amdgcn_sched_barrier(0);
        mma(xhB, xlB, whB, wlB);
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (PF) {
        load_frags(Wc, tap0, 0, xhA, xlA, whA, wlA);
        for (int it = 0; it < nit; it += 2) {
          if (it + 1 < nit) load_frags(Wc, tap0, it + 1, xhB, xlB, whB, wlB);
          mma(xhA, xlA, whA, wlA);
          if (it + 1 < nit) {
            if (it + 2 < nit) load_frags(Wc, tap0, it + 2, xhA, xlA, whA, wlA);
            mma(xhB, xlB, whB, wlB);
          }
        }
      } else {
        for (int it = 0; it < nit; ++it) {
          load_frags(Wc, tap0, it, xhA, xlA, whA, wlA);
          mma(xhA, xlA, whA, wlA);
        }
      }
      CL_STAMP();                               // step: MFMAs done
      if (more) {
        if (new_chunk) {
          CL_SYNC();
          x_write();
        }
        if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA'd slice has landed
        else w_write((step + 1) & 1, wq[0]);
        CL_STAMP();                             // step: next slice (and window) in LDS
        CL_SYNC();
        CL_STAMP();                             // step: barrier
      }
      chunk = chunk_n;
      sl = sl_n;
    }
  } else if constexpr (WD == 2) {
    for (int step = 0; step < nsteps; step += 2) {
      step_body(step, wq[0], wq[1]);
      if (step + 1 < nsteps) step_body(step + 1, wq[1], wq[0]);
    }
  } else {
    for (int step = 0; step < nsteps; step += 3) {
      step_body(step, wq[0], wq[1]);
      if (step + 1 < nsteps) step_body(step + 1, wq[1], wq[2]);
      if (step + 2 < nsteps) step_body(step + 2, wq[2], wq[0]);
    }
  }

  CL_STAMP();                                   // epilogue start
  // ---- epilogue.  D tile: register r is time row rr(r) = (r&3) + 8*(r>>2) (+4h), lane = channel.
  //      Offsets (bytes, 32-bit) = lane part once per (mt, nt) + wave-uniform row part; rows outside
  //      [0, T_store) fall outside the descriptor: loads give 0, stores are dropped.
  const int n_wave = a.phases > 1 ? a.phases * (t0 + row0) + ph - a.ups_p : (t0 + row0);  // uniform, may be < 0
  const int ostep = a.phases * a.o_ts * 4, rstep = a.phases * (a.res ? a.r_ts : a.o_ts) * 4;  // bytes per D row
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = (cb * NTB + wn * NT + nt) * 32 + l31;
      const int lo = (4 * h + mt * 32) * ostep + co * 4 + n_wave * a.o_ts * 4;
      const int lr = (4 * h + mt * 32) * rstep + co * 4 + n_wave * (a.res ? a.r_ts : a.o_ts) * 4;
      // uniform conditions hoisted: each optional operand is one block of 16 loads + 16 adds
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = hh[mt][nt][r] + cr[mt][nt][r] * (1.f / 2048.f);
      if (a.res && (VSP_DIAG & 8) == 0) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr_, lr + ((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rv[r];
      }
      if (a.acc_prev && (VSP_DIAG & 8) == 0) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, lo + ((r & 3) + 8 * (r >> 2)) * ostep, 0, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += pv[r];
      }
      if (a.div != 1.f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] /= a.div;
      }
      if constexpr ((VSP_DIAG & 8) != 0) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += v[r];
        if (sum == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sum), ro, lo, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, lo + ((r & 3) + 8 * (r >> 2)) * ostep, 0, 0);
      }
    }
  }
#ifdef VSP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CL_STAMP();                                   // end (stores retired)
#endif
}

#ifdef VSP_STAMPS
extern "C" int vsp_debug_stamps_cl(unsigned long long* host, int max_samples, int reset) {
  unsigned n = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_cl_stamp_count), sizeof n);
  if ((int)n > max_samples) n = max_samples;
  if (n > (unsigned)CL_NSAMPLE) n = CL_NSAMPLE;
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cl_stamps), (size_t)n * CL_NSTAMP * sizeof(unsigned long long));
  if (max_samples >= 2 * CL_NSAMPLE)   // caller's buffer has room for the cycle counters behind the stamps
    (void)hipMemcpyFromSymbol(host + (size_t)CL_NSAMPLE * CL_NSTAMP, HIP_SYMBOL(g_cl_cycles),
                              (size_t)n * CL_NSTAMP * sizeof(unsigned long long));
  if (reset) { const unsigned z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cl_stamp_count), &z, sizeof z); }
  return (int)n;
}
#endif

template <int MT, int NT, int WM, int WN, int CKC, int G, bool PF, int WD, int TERMS, bool GLDS = false>
static hipError_t launch_cl_tile(const ClConvArgs& a_in, int B, hipStream_t s) {
  ClConvArgs a = a_in;
#ifdef VSP_STAMPS
  {  // stamps only in the VSP_STAMP_CL-th launch of this tile type (0-based)
    static int launch_no = 0;
    static int target = -2;
    if (target == -2) { const char* e = getenv("VSP_STAMP_CL"); target = e ? atoi(e) : -1; }
    if (launch_no++ == target) {
      const char* w = getenv("VSP_STAMP_WAVE");
      a.terms |= 0x100 | ((w ? atoi(w) : 0) << 12);
    }
  }
#endif
  constexpr int BT = 32 * MT * WM;
  constexpr int RPU = (64 * WM * WN) / (CKC / 4);
  constexpr int WMAXL = (BT + CL_HALO + RPU - 1) / RPU * RPU;
  constexpr size_t lds = ((size_t)2 * WMAXL * (CKC + 8) + (size_t)4 * G * (CKC / 16) * NT * WN * 512) *
                         sizeof(_Float16);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = cl_conv_f16s<MT, NT, WM, WN, CKC, G, PF, WD, TERMS, GLDS>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nnt = a.Cout / 32;
  if (nnt % (NT * WN) || a.Cin % CKC) return hipErrorInvalidValue;
  dim3 grid((a.Nq + BT - 1) / BT, a.phases * (nnt / (NT * WN)), B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_cl_conv(const ClConvArgs& a, int B, hipStream_t s) {
  if ((a.K - 1) * a.dil > CL_HALO || a.K < 1 || a.Nq <= 0 || B <= 0 || a.Cout % 32 || a.Cin % 32 || a.phases < 1 ||
      (a.x_ts & 3) || (a.x_bs & 3) || (reinterpret_cast<uintptr_t>(a.x) & 15))
    return hipErrorInvalidValue;
  // <MT, NT, WM, WN, CKC, G, PF, WD, TERMS, GLDS>; tile choices measured on MI355X (profiles/r01_tile_experiments.txt):
  //  * >= 128 output channels: 8 waves x (64 rows x 64 channels), 64-channel chunks, one block per CU
  //    (two blocks of 4 waves, one fat wave per SIMD, 64-column tiles at two blocks per CU: all measured slower or equal);
  //  * 64 / 32 output channels: LDS kept under 80 KiB (32-channel chunks, small weight ring) so that
  //    TWO blocks share a CU.  With the ResBlock pairs of those stages fused (respair_f16s.hip) only the
  //    transposed convs into them still use these two tiles.
  if (a.terms == 1) {
    if (a.Cout % 128 == 0 && a.Cin % 64 == 0) return launch_cl_tile<2, 2, 4, 2, 64, 1, false, 2, 1>(a, B, s);
    if (a.Cout % 64 == 0) return launch_cl_tile<2, 1, 4, 2, 32, 1, true, 1, 1>(a, B, s);
    return launch_cl_tile<1, 1, 8, 1, 32, 2, true, 1, 1>(a, B, s);
  }
  // 128-column tile: weight ring filled by LDS-DMA, fragments double-buffered on a fixed schedule (default);
  // VSP_BIG_TILE=0 selects the register-staged ring (two slices ahead, no fragment prefetch), 1 the DMA ring alone
  static int big = -1;
  if (big < 0) { const char* e = getenv("VSP_BIG_TILE"); big = e ? atoi(e) : 2; }
  if (a.Cout % 128 == 0 && a.Cin % 64 == 0) {
    if (big == 1) return launch_cl_tile<2, 2, 4, 2, 64, 1, false, 1, 3, true>(a, B, s);
    if (big == 0) return launch_cl_tile<2, 2, 4, 2, 64, 1, false, 2, 3>(a, B, s);
    return launch_cl_tile<2, 2, 4, 2, 64, 1, true, 1, 3, true>(a, B, s);
  }
  if (a.Cout % 64 == 0) return launch_cl_tile<2, 1, 4, 2, 32, 1, true, 1, 3>(a, B, s);
  return launch_cl_tile<1, 1, 8, 1, 32, 2, true, 1, 3>(a, B, s);
}

// ------------------------------------------------------------------------------------------
// [B][C][T] -> [B][T][C] (conv_pre's output enters the channels-last vocoder), 32x32 LDS tiles
__global__ void __launch_bounds__(256) transpose_ct_kernel(const float* __restrict__ x, long x_bs, long x_cs,
                                                           float* __restrict__ y, long y_bs, int y_ts, int C, int T) {
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
    if (t < T && c < C) y[(size_t)b * y_bs + (size_t)t * y_ts + c] = tile[lx][ly + 8 * k];
  }
}
hipError_t launch_transpose_ct(const float* x, long x_bs, long x_cs, float* y, long y_bs, int y_ts, int B, int C,
                               int T, hipStream_t s) {
  hipLaunchKernelGGL(transpose_ct_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, x, x_bs, x_cs, y,
                     y_bs, y_ts, C, T);
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
                                                           float slope, float* __restrict__ o, long o_bs, int T) {
  constexpr int RSF = C + 4;
  constexpr int C4 = C / 4;
  __shared__ __attribute__((aligned(16))) float xs[(CPL_TILE + 8) * RSF];
  const int b = blockIdx.y, t0 = blockIdx.x * CPL_TILE, pad = (K - 1) / 2;
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
  if (t < T) o[(size_t)b * o_bs + t] = tanhf(acc);
}
hipError_t launch_conv_post_cl(const float* x, long x_bs, int x_ts, const float* w, int C, int K, float slope,
                               float* o, long o_bs, int B, int T, hipStream_t s) {
  if (K > 8 || x_ts != C || (C != 32 && C != 64)) return hipErrorInvalidValue;
  dim3 grid((T + CPL_TILE - 1) / CPL_TILE, B);
  if (C == 32)
    hipLaunchKernelGGL(conv_post_cl_kernel<32>, grid, dim3(256), 0, s, x, x_bs, w, K, slope, o, o_bs, T);
  else
    hipLaunchKernelGGL(conv_post_cl_kernel<64>, grid, dim3(256), 0, s, x, x_bs, w, K, slope, o, o_bs, T);
  return hipGetLastError();
}

}  // namespace vsp
