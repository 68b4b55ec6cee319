// Micro-benchmark: the inner loop of the 128-column tile of cl_conv_f16s in isolation -- per k-step 8 ds_read_b128
// (two A tiles, two B tiles, hi and lo) and 12 v_mfma_f32_32x32x16_f16 (tile-interleaved HH / CROSS / CROSS) --
// with 8 waves per block (two per SIMD), no global traffic, no barriers.  Prints the matrix-core issue rate so
// that the product kernel's 38-44 % can be compared with what this instruction mix can reach at all.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds_loop.hip -o /tmp/mfma_lds_loop && /tmp/mfma_lds_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: load -> mfma per k-step; 1: fragments of the next k-step requested first (pinned); 2: no LDS (register
// operands); 3: MODE 0 + one barrier per 4 k-steps (a weight-slice step); 4: MODE 3 + a 32 KiB LDS-DMA per step
template <int MODE>
__global__ void __launch_bounds__(512) loop_kernel(float* out, int iters, const uint4* __restrict__ wsrc) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pseudo-random operands (zero-filled or regular data reads high: less switching power, higher clock)
  for (int i = tid; i < 40 * 1024; i += 512) {
    unsigned hsh = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    lds[i] = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 32768.0f));
  }
  __syncthreads();
  const _Float16* xa = lds + (wave & 3) * 2048 + lane * 8;     // conflict-free lane-linear fragments
  const _Float16* wb = lds + 16384 + (wave >> 2) * 2048 + lane * 8;
  f32x16 hh[2][2], cr[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) { hh[a][b][r] = 0.f; cr[a][b][r] = 0.f; }
  auto load = [&](int k, f16x8(&xh)[2], f16x8(&xl)[2], f16x8(&wh)[2], f16x8(&wl)[2]) {
    const int o = (k & 3) * 512;
    for (int m = 0; m < 2; ++m) {
      xh[m] = *reinterpret_cast<const f16x8*>(xa + o + m * 8192 / 8);
      xl[m] = *reinterpret_cast<const f16x8*>(xa + o + m * 8192 / 8 + 4096);
      wh[m] = *reinterpret_cast<const f16x8*>(wb + o + m * 8192 / 8);
      wl[m] = *reinterpret_cast<const f16x8*>(wb + o + m * 8192 / 8 + 4096);
    }
  };
  auto mma = [&](const f16x8(&xh)[2], const f16x8(&xl)[2], const f16x8(&wh)[2], const f16x8(&wl)[2]) {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[a], wh[b], hh[a][b], 0, 0, 0);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) cr[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[a], wl[b], cr[a][b], 0, 0, 0);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) cr[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[a], wh[b], cr[a][b], 0, 0, 0);
  };
  f16x8 xhA[2], xlA[2], whA[2], wlA[2], xhB[2], xlB[2], whB[2], wlB[2];
  if (MODE == 2) load(0, xhA, xlA, whA, wlA);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3 || MODE == 4) {
      if (MODE == 4) {
        // 32 fragment blocks of 1 KiB per step into a scratch area of LDS, as the ring fill does
#pragma unroll
        for (int u = 0; u < 4; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ((it & 7) * 32 + u * 8 + wave) * 64 + lane),
                                           (__attribute__((address_space(3))) void*)(lds + 24576 + (u * 8 + wave) * 512), 16, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { load(k, xhA, xlA, whA, wlA); mma(xhA, xlA, whA, wlA); }
      if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE >= 3) __syncthreads();
    } else if (MODE == 1) {
      load(0, xhA, xlA, whA, wlA);
      load(1, xhB, xlB, whB, wlB);
      __builtin_amdgcn_sched_barrier(0);
      mma(xhA, xlA, whA, wlA);
      __builtin_amdgcn_sched_barrier(0);
      load(2, xhA, xlA, whA, wlA);
      __builtin_amdgcn_sched_barrier(0);
      mma(xhB, xlB, whB, wlB);
      __builtin_amdgcn_sched_barrier(0);
      load(3, xhB, xlB, whB, wlB);
      __builtin_amdgcn_sched_barrier(0);
      mma(xhA, xlA, whA, wlA);
      __builtin_amdgcn_sched_barrier(0);
      mma(xhB, xlB, whB, wlB);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) mma(xhA, xlA, whA, wlA);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += hh[a][b][r] + cr[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu) {
  const int iters = 2000, nblk = 256 * blocks_per_cu;
  float* out;
  hipMalloc(&out, (size_t)nblk * 512 * 4);
  uint4* wsrc;
  hipMalloc(&wsrc, 8 * 32 * 1024);
  hipMemset(wsrc, 0x11, 8 * 32 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(nblk), dim3(512), 96 * 1024, 0, out, iters, wsrc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)nblk * 8 * iters * 48;
  const double flops = mfma * 2.0 * 32 * 32 * 16;
  printf("%-34s %d block(s)/CU: %.3f ms, %.0f TFLOP/s f16 MFMA = %.1f %% of 2.5 PF nominal (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
         name, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500 * 100, ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
  hipFree(out);
}

int main() {
  run<2>("register operands (no LDS)", 1);
  run<0>("load -> mfma per k-step", 1);
  run<1>("next fragments first (pinned)", 1);
  run<3>("+ barrier every 4 k-steps", 1);
  run<4>("+ barrier + 32 KiB LDS-DMA per step", 1);
  return 0;
}
