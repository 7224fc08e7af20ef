// MFMA-only streams on random register operands: does the instruction shape change the power-limited rate?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const uint16_t* X, int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  bf16x8 f[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) f[i] = *(const bf16x8*)(X + ((size_t)(blockIdx.x * 512 + threadIdx.x) * 12 + i) * 8);
  float r = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        acc[t & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[t % 12], f[(t + 5) % 12], acc[t & 7], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][lane & 15];
  } else {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 32; ++t)   // same flops per iteration: 32 x (16x16x32) == 16 x (32x32x16)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[t % 12], f[(t + 5) % 12], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i][lane & 3];
  }
  if (r == 12345.678f) sink[0] = r;
}

extern "C" int run_mfma(const void* X, int iters, int shape, float* sink, void* st) {
  if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, (hipStream_t)st, (const uint16_t*)X, iters, sink);
  else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, (hipStream_t)st, (const uint16_t*)X, iters, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
