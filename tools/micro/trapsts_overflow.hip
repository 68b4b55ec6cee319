// Does the wave's TRAPSTS.EXCP sticky field record the overflow of v_cvt_pkrtz_f16_f32 (the operand split's packing, which
// SATURATES at the largest finite f16) with exceptions DISABLED (the default MODE)?  If so, a kernel can surface activation
// saturation by reading ONE hardware register at its end instead of comparing every value it splits.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/trapsts_overflow.hip -o build/trapsts_overflow ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
  // one value per launch-block so that sticky bits do not leak between cases: block b converts in[b]
  const float x = in[blockIdx.x];
  unsigned before, after;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(before));
  const auto h = __builtin_amdgcn_cvt_pkrtz(x, 0.f);
  unsigned bits = __builtin_bit_cast(unsigned, h);
  asm volatile("s_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(after) : "v"(bits));
  float y = x * x;                       // fp32 overflow for |x| > 1.8e19
  unsigned after2;
  asm volatile("s_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(after2) : "v"(y));
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = before; out[blockIdx.x * 4 + 1] = after; out[blockIdx.x * 4 + 2] = bits & 0xffff;
    out[blockIdx.x * 4 + 3] = after2;
  }
}
int main() {
  const int n = 6;
  const float v[n] = {1.0f, 65504.f, 65520.f, 1.0e6f, -3.0e7f, 1.0e30f};
  float* din; unsigned* dout;
  hipMalloc(&din, sizeof v); hipMalloc(&dout, n * 16);
  hipMemcpy(din, v, sizeof v, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, din, dout, n);
  unsigned o[n * 4]; hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("x = %-10g  f16 bits %04x  TRAPSTS before %08x  after cvt %08x (EXCP %03x)  after x*x %08x (EXCP %03x)\n", v[i], o[i * 4 + 2],
           o[i * 4], o[i * 4 + 1], o[i * 4 + 1] & 0x1ff, o[i * 4 + 3], o[i * 4 + 3] & 0x1ff);
  return 0;
}
