// Micro-benchmark: what does replacing the two CROSS products of a split-f16 product (wl*xh, wh*xl) by int8 MFMAs buy
// under the board's power limit?  Per 64-deep step of a 64x64 wave tile:
//   V0  6 x v_mfma_f32_16x16x32_f16 per tile (hh, lh, hl for both 32-deep halves)      -- the product's arithmetic
//   V1  2 x f16 (hh) + 2 x v_mfma_i32_16x16x64_i8 (the two cross terms, 64 deep each)   -- the int8-cross scheme
//   V2  2 x f16 only (one term)                                                         -- lower bound
//   V3  6 x i8 (rate / power of the i8 instruction alone)
// Operands are re-read from LDS every step (16 ds_read_b128 per 32-deep f16 step, as in the kernels), pseudo-random
// data, 8 waves per block, one block per CU, no global traffic.  Each variant runs ~1 s so that the power management
// settles; prints steps/s (a step = 64 deep on a 64x64 tile per wave) -- the quantity the generator needs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_mix.hip -o build/mfma_mix && build/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ void __launch_bounds__(512) mix_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 48 * 1024; i += 512) {
    unsigned hsh = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    const _Float16 v = (_Float16)(((int)(hsh & 0xffff) - 32768) * (1.0f / 32768.0f));
    lds[i] = __builtin_bit_cast(unsigned short, v);
  }
  __syncthreads();
  const unsigned short* xa = lds + (wave & 3) * 4096 + lane * 8;
  const unsigned short* wb = lds + 24576 + (wave >> 2) * 4096 + lane * 8;
  f32x4 hh[4][4];
  i32x4 cr[4][4];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 4; ++r) { hh[a][b][r] = 0.f; cr[a][b][r] = 0; }
  for (int it = 0; it < iters; ++it) {
    // two 32-deep halves of a 64-deep step
    f16x8 xh[2][4], wh[2][4];
    i32x4 x8[2][4], w8[2][4];     // [0] = "hi as int8 x 64 deep", [1] = "lo as int8 x 64 deep" (same LDS bytes as the f16 lo images)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      for (int m = 0; m < 4; ++m) {
        xh[k][m] = *reinterpret_cast<const f16x8*>(xa + m * 512 + k * 8192);
        wh[k][m] = *reinterpret_cast<const f16x8*>(wb + m * 512 + k * 8192);
        x8[k][m] = *reinterpret_cast<const i32x4*>(xa + m * 512 + 2048 + k * 8192);
        w8[k][m] = *reinterpret_cast<const i32x4*>(wb + m * 512 + 2048 + k * 8192);
      }
    if constexpr (V == 0) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k][a], wh[k][b], hh[a][b], 0, 0, 0);
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k][a], __builtin_bit_cast(f16x8, w8[k][b]), hh[a][b], 0, 0, 0);
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x8[k][a]), wh[k][b], hh[a][b], 0, 0, 0);
      }
    } else if constexpr (V == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k][a], wh[k][b], hh[a][b], 0, 0, 0);
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) cr[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x8[0][a], w8[1][b], cr[a][b], 0, 0, 0);
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) cr[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x8[1][a], w8[0][b], cr[a][b], 0, 0, 0);
    } else if constexpr (V == 2) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) hh[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k][a], wh[k][b], hh[a][b], 0, 0, 0);
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) cr[a][b] += x8[a & 1][a] + w8[b & 1][b];   // keep the loads alive
    } else {
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
              cr[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(r3 ? x8[k][a] : __builtin_bit_cast(i32x4, xh[k][a]), r3 == 1 ? __builtin_bit_cast(i32x4, wh[k][b]) : w8[k][b], cr[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < 4; ++r) s += hh[a][b][r] + (float)cr[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V>
static void run(const char* name, int reps) {
  const int iters = 4000, nblk = 256;
  float* out;
  hipMalloc(&out, (size_t)nblk * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL((mix_kernel<V>), dim3(nblk), dim3(512), 96 * 1024, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix_kernel<V>), dim3(nblk), dim3(512), 96 * 1024, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)nblk * 8 * iters * reps;      // 64-deep steps of a 64x64 wave tile
  printf("%-58s %8.1f ms  %7.2f G tile-steps/s  (= %6.0f TFLOP/s of fp32-accurate products at 64x64x64x2 per step)\n", name, ms,
         steps / ms / 1e6, steps * 64.0 * 64 * 64 * 2 / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("warm-up (3 f16 MFMAs per product)", 20);
  run<0>("V0  3 x f16 per product (the product's arithmetic)", 60);
  run<1>("V1  1 x f16 + cross terms as 2 x i8 16x16x64 per 64 deep", 60);
  run<2>("V2  1 x f16 only", 60);
  run<3>("V3  i8 16x16x64 only, 6 per step", 60);
  run<0>("V0 again", 60);
  run<1>("V1 again", 60);
  return 0;
}
