// Issue rate of v_exp_f32 / v_rcp_f32 against v_fma_f32 on gfx950 at 1 / 2 / 4 waves per SIMD, 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cyc) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + threadIdx.x * 1e-3f + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
      if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[(blockIdx.x * 1024 + threadIdx.x) & 65535] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; long long h;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  const char* names[3] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32"};
  for (int wps = 1; wps <= 4; wps *= 2)
  for (int op = 0; op < 3; ++op) {
    const int nthr = 256 * wps;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(nthr), 0, 0, out, iters, cyc);
      if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(nthr), 0, 0, out, iters, cyc);
      if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(nthr), 0, 0, out, iters, cyc);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d wave(s) per SIMD: %.3f ms for %d x 8 instructions per wave: %.2f ns per SIMD-instruction, %.2f counter ticks "
           "per SIMD-instruction\n", names[op], wps, ms, iters, ms * 1e6 / (iters * 8.0 * wps), (double)h / (iters * 8.0 * wps));
  }
  return 0;
}
