// Memory side of the producer epilogue of gemm_pp_kernel<0, false, 2> WITHOUT the GEMM: per 256 x 256 tile a workgroup of 8
// waves reads the fp32 residual tile, writes the fp32 tile back and a bf16 copy.  How fast can ONE CU move these bytes, and how
// does the rate change with the number of CUs doing it at once (per-CU bound vs chip-wide HBM bound), with the look-ahead
// depth, and with the lane -> address mapping (32 x 32 blocks, 128 B per row and instruction -- the kernel's -- or 16 x 64
// blocks, 256 B per row)?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define AS1 __attribute__((address_space(1)))

__device__ __forceinline__ unsigned pk(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  b2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return *reinterpret_cast<unsigned*>(&v);
}

// MAP 0: wave tile 128 x 64 in 8 blocks of 32 rows x 32 columns: lane (srow = lane / 8, sch = lane % 8) owns 4 columns of rows
//        srow + 8 i: every 16-byte access instruction touches 8 rows x 128 B.
// MAP 1: blocks of 16 rows x 64 columns: lane (srow = lane / 16, sch = lane % 16): 4 rows x 256 B per instruction.
// NT: non-temporal loads / stores.  DEPTH: blocks whose loads are in flight before the first use.  BF: also write the bf16 copy.
template <int MAP, int DEPTH, bool NT, bool BF>
__global__ __launch_bounds__(512) void epi_probe(float* __restrict__ x, uint16_t* __restrict__ xb, int ld, int ntm, int ntn, int tiles) {
  extern __shared__ char lds_hold[];                 // 150 KB requested: one workgroup per CU
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  constexpr int NB = 8;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int m0 = (t % ntm) * 256 + wm, n0 = (t / ntm) * 256 + wn;
    f32x4 rv[NB][4];
    auto addr = [&](int b, int i, size_t& row, int& col) {
      if (MAP == 0) { row = (size_t)(m0 + (b >> 1) * 32 + i * 8 + (lane >> 3)); col = n0 + (b & 1) * 32 + (lane & 7) * 4; }
      else { row = (size_t)(m0 + b * 16 + i * 4 + (lane >> 4)); col = n0 + (lane & 15) * 4; }
    };
    auto load = [&](int b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        size_t row; int col;
        addr(b, i, row, col);
        const AS1 f32x4* p = (const AS1 f32x4*)(x + row * ld + col);
        rv[b][i] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    };
#pragma unroll
    for (int b = 0; b < DEPTH; ++b) load(b);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        size_t row; int col;
        addr(b, i, row, col);
        f32x4 v = rv[b][i];
        v[0] += 1.f; v[1] += 1.f; v[2] += 1.f; v[3] += 1.f;
        AS1 f32x4* p = (AS1 f32x4*)(x + row * ld + col);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        if (BF) {
          u32x2 o = {pk(v[0], v[1]), pk(v[2], v[3])};
          *(AS1 u32x2*)(xb + row * ld + col) = o;
        }
      }
      if (b + DEPTH < NB) load(b + DEPTH);
    }
  }
}

template <int MAP, int DEPTH, bool NT, bool BF>
static void go(float* x, uint16_t* xb, int ld, int ntm, int ntn, int grid, hipStream_t st) {
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&epi_probe<MAP, DEPTH, NT, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); done = true; }
  hipLaunchKernelGGL((epi_probe<MAP, DEPTH, NT, BF>), dim3(grid), dim3(512), 150 * 1024, st, x, xb, ld, ntm, ntn, ntm * ntn);
}

extern "C" int run_epi(float* x, uint16_t* xb, int ld, int ntm, int ntn, int grid, int map, int depth, int nt, int bf, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define CASE(M, D, N, B) if (map == M && depth == D && nt == N && bf == B) { go<M, D, N, B>(x, xb, ld, ntm, ntn, grid, st); return 0; }
  CASE(0, 2, 1, 1) CASE(0, 8, 1, 1) CASE(1, 2, 1, 1) CASE(1, 8, 1, 1) CASE(0, 2, 0, 1) CASE(0, 8, 0, 1) CASE(0, 4, 1, 1)
  CASE(0, 2, 1, 0) CASE(0, 8, 1, 0) CASE(1, 8, 0, 1) CASE(1, 8, 1, 0)
  return -1;
}
