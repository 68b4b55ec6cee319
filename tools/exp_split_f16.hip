// Experiment (not product code): accuracy and rate of an fp32-accurate GEMM built from f16 MFMAs
// on split operands, against the native f32 MFMA chain.  x = hi + lo * 2^-11 with hi = f16(x),
// lo = f16((x - hi) * 2^11); x*w ~= hi*hi + (hi*lo + lo*hi) * 2^-11  (3 MFMAs, 2 accumulators).
//   hipcc --offload-arch=gfx950 -O3 tools/exp_split_f16.hip -o /tmp/exp && /tmp/exp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 352;  // 32 channels x 11 taps

__global__ void gemm_f32(const float* A, const float* B, float* C) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

__device__ inline void split16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * 2048.f);
}

__global__ void gemm_split_f16(const float* A, const float* B, float* C, int nterms) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 hh = {0}, cr = {0};
  for (int k = 0; k < K; k += 16) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      _Float16 x, y;
      split16(A[r * K + k + 8 * h + j], x, y); ah[j] = x; al[j] = y;
      split16(B[(k + 8 * h + j) * 32 + r], x, y); bh[j] = x; bl[j] = y;
    }
    hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hh, 0, 0, 0);
    if (nterms >= 3) {
      cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cr, 0, 0, 0);
      cr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cr, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = hh[i] + cr[i] * (1.f / 2048.f);
}

__device__ inline short bf16_rn(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFF + ((u >> 16) & 1);
  return (short)(u >> 16);
}
__device__ inline float bf16_f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }

__global__ void gemm_split_bf16(const float* A, const float* B, float* C, int nterms) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 16) {
    bf16x8 a0, a1, a2, b0, b1, b2;
    for (int j = 0; j < 8; ++j) {
      float x = A[r * K + k + 8 * h + j];
      short s0 = bf16_rn(x); float r1 = x - bf16_f(s0); short s1 = bf16_rn(r1); short s2 = bf16_rn(r1 - bf16_f(s1));
      a0[j] = s0; a1[j] = s1; a2[j] = s2;
      x = B[(k + 8 * h + j) * 32 + r];
      s0 = bf16_rn(x); r1 = x - bf16_f(s0); s1 = bf16_rn(r1); s2 = bf16_rn(r1 - bf16_f(s1));
      b0[j] = s0; b1[j] = s1; b2[j] = s2;
    }
    // smallest terms first
    if (nterms >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
    }
    if (nterms >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// rate: back-to-back MFMAs, 4 waves per block, 1 block per CU x 256 x several
template <int MODE>
__global__ void rate(float* out, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const float s = threadIdx.x * 1e-3f;
  if (MODE == 0) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(s, s, c3, 0, 0, 0);
    }
  } else {
    f16x8 a;
    for (int j = 0; j < 8; ++j) a[j] = (_Float16)(s + j);
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  std::vector<float> A(32 * K), B(K * 32);
  srand(1);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2 - 1); };
  for (auto& x : A) x = rnd() * 3.f;     // activations O(1)
  for (auto& x : B) x = rnd() * 0.05f;   // weights O(1e-2)
  std::vector<double> ref(32 * 32), mag(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0, m = 0;
      for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
      ref[i * 32 + j] = s; mag[i * 32 + j] = m;
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> C(32 * 32);
  auto report = [&](const char* name) {
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, mx = 0;
    for (int i = 0; i < 32 * 32; ++i) { mx = std::fmax(mx, std::fabs(ref[i])); }
    for (int i = 0; i < 32 * 32; ++i) {
      e1 = std::fmax(e1, std::fabs(C[i] - ref[i]) / mx);
      e2 = std::fmax(e2, std::fabs(C[i] - ref[i]) / mag[i]);
    }
    printf("%-28s max|err|/max|C| = %.3e   max|err|/sum|ab| = %.3e\n", name, e1, e2);
  };
  gemm_f32<<<1, 64>>>(dA, dB, dC); report("f32 mfma 32x32x2");
  gemm_split_f16<<<1, 64>>>(dA, dB, dC, 1); report("f16 1-term");
  gemm_split_f16<<<1, 64>>>(dA, dB, dC, 3); report("f16 split 3-term");
  gemm_split_bf16<<<1, 64>>>(dA, dB, dC, 1); report("bf16 1-term");
  gemm_split_bf16<<<1, 64>>>(dA, dB, dC, 3); report("bf16 split 3-term");
  gemm_split_bf16<<<1, 64>>>(dA, dB, dC, 6); report("bf16 split 6-term");
  // host fp32 sequential for scale
  {
    double e2 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 32 + j], s);
      e2 = std::fmax(e2, std::fabs(s - ref[i * 32 + j]) / mag[i * 32 + j]);
    }
    printf("%-28s max|err|/sum|ab| = %.3e\n", "host fmaf chain", e2);
  }
  // rates
  float* dO; hipMalloc(&dO, 1024 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) rate<0><<<1024, 256>>>(dO, iters); else rate<1><<<1024, 256>>>(dO, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 1024.0 * 4 * iters * 4 * (mode == 0 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2);
      printf("rate mode %d: %.2f ms  %.1f TFLOP/s\n", mode, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
