// Load-only probe: the GEMM's LDS-DMA traffic without MFMA/ds_read.  Patterns:
//   0: row gather, XOR-swizzled chunks (current GEMM)   1: row gather, linear chunks
//   2: pre-tiled operand, each slab one contiguous 32 KiB block (1 KiB per wave-instruction)
#include <hip/hip_runtime.h>
#include <stdint.h>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)g, (AS3 void*)l, 16, 0, 0);
}
template <int PAT, int DEPTH>
__global__ __launch_bounds__(512, 2) void probe(const uint16_t* X, const uint16_t* W, int M, int N, int K, int* sink) {
  __shared__ __attribute__((aligned(1024))) uint16_t smem[2][512 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nt_n = N / 256, nt_m = M / 256, ntiles = nt_n * nt_m, nk = K / 64;
  const int pc = lane & 7, rsub = lane >> 3;
  const int lc = PAT == 0 ? (pc ^ ((lane >> 4) | ((wave & 1) << 2))) : pc;
  int it = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
    const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
    const int tm = id / nt_n, tn = id % nt_n;
    for (int kt = 0; kt < nk; ++kt, ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint16_t* sx; const uint16_t* sw;
        if (PAT == 2) {
          sx = X + ((size_t)(tm * nk + kt) * 256 * 64) + ((i * 8 + wave) * 64 + lane) * 8;
          sw = W + ((size_t)(tn * nk + kt) * 256 * 64) + ((i * 8 + wave) * 64 + lane) * 8;
        } else {
          const int row = (i * 8 + wave) * 8 + rsub;
          sx = X + (size_t)(tm * 256 + row) * K + kt * 64 + lc * 8;
          sw = W + (size_t)(tn * 256 + row) * K + kt * 64 + lc * 8;
        }
        glds16(sx, &smem[it & 1][(i * 8 + wave) * 512]);
        glds16(sw, &smem[it & 1][256 * 64 + (i * 8 + wave) * 512]);
      }
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = smem[0][lane];
}
extern "C" int run_probe(int pat, int depth, const void* X, const void* W, int M, int N, int K, int* sink, int grid, void* st) {
#define L(P, D) hipLaunchKernelGGL((probe<P, D>), dim3(grid), dim3(512), 0, (hipStream_t)st, (const uint16_t*)X, (const uint16_t*)W, M, N, K, sink)
  if (pat == 0 && depth == 1) L(0, 1); else if (pat == 1 && depth == 1) L(1, 1); else if (pat == 2 && depth == 1) L(2, 1);
  else if (pat == 0) L(0, 2); else if (pat == 1) L(1, 2); else L(2, 2);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
