// LDS read bandwidth of one CU on gfx950: NW waves per block (one block per CU), each issuing ds_read_b128 back to back
// (conflict-free: lane * 16 bytes + immediate offsets).  Prints bytes per clock per CU at the measured time and an assumed clock.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_read_bw.hip -o build/lds_read_bw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WIDTH>
__global__ void __launch_bounds__(1024) k(int iters, float* out, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * WIDTH +
                        ((threadIdx.x >> 6) & 3) * 8192;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (WIDTH == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(base), "n"(u * 1024));
      else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 t;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(base), "n"(u * 512));
        v[u] = f32x4{t.x, t.y, 0.f, 0.f};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(v[u]));
    acc += v[0];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  if (acc.x == 12345.678f) out[0] = acc.x;
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int width : {16, 8})
    for (int nw : {1, 2, 4, 8, 16}) {
      auto kern = width == 16 ? k<16> : k<8>;
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(kern, dim3(256), dim3(64 * nw), 65536, 0, 100, out, clk);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(64 * nw), 65536, 0, iters, out, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
      const double bytes = (double)iters * 8 * 64 * width * nw;   // per CU
      printf("ds_read_b%d  %2d waves/CU: %.3f ms  %.1f B/ns/CU = %.1f B/clk at 2.4 GHz | s_memtime ticks %llu (100 MHz: %.3f ms)\n",
             width * 8, nw, ms, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4, c, c / 1e5);
    }
  return 0;
}
