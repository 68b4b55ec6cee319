// Does v_mfma_f32_16x16x32_f16 keep f16 DENORMAL inputs, or flush them to zero?  (Decides whether the operand split's lo
// parts could stay unscaled -- x = hi + lo with lo as an f16 denormal where |x| is small -- so that HH and the two CROSS
// products of the 3-term scheme could share ONE fp32 accumulator; today lo is carried * 2^11 to stay a normal number and
// the cross products have their own accumulator.)  Prints D[0][0] of A = a (all rows, k) x B = b for denormal a.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f16_denorm.hip -o build/mfma_f16_denorm ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const _Float16* av, const _Float16* bv, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = av[i]; b[j] = bv[i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[i] = c[0];
  }
}
int main() {
  const int n = 6;
  // a: 2^-24 (smallest denormal), 2^-20, 2^-15 (largest denormal region), 2^-14 (smallest normal), 3 * 2^-24, 1.0
  const float af[n] = {5.9604645e-8f, 9.5367432e-7f, 3.0517578e-5f, 6.1035156e-5f, 1.7881393e-7f, 1.f};
  const float bf[n] = {1024.f, 1024.f, 1024.f, 1024.f, 1024.f, 1024.f};
  _Float16 ah[n], bh[n];
  for (int i = 0; i < n; ++i) { ah[i] = (_Float16)af[i]; bh[i] = (_Float16)bf[i]; }
  _Float16 *da, *db; float* dout;
  hipMalloc(&da, sizeof ah); hipMalloc(&db, sizeof bh); hipMalloc(&dout, n * 4);
  hipMemcpy(da, ah, sizeof ah, hipMemcpyHostToDevice); hipMemcpy(db, bh, sizeof bh, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, n);
  float o[n]; hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("a = %.9g (as f16 %.9g) x b = %g, k = 32:  D = %.9g   expected %.9g   %s\n", af[i], (float)ah[i], bf[i], o[i],
           32.0 * (double)(float)ah[i] * bf[i], o[i] == (float)(32.0 * (double)(float)ah[i] * bf[i]) ? "kept" : "DIFFERENT");
  return 0;
}
