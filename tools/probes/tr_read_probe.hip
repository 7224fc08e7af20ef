// RESULT (MI355X, measured with this probe): inside each 16-lane group, lane s passes the address of 4 consecutive 16-bit
// elements vec[s][0..3]; lane i receives out[i][j] = vec[4*j + i/4][i % 4], j = 0..3 -- a transpose of the group's 16 four-
// vectors seen as a [4][16] block whose row j is the concatenation of the vectors of lanes 4j .. 4j+3.  So with an LDS
// image img[k][n] (k = contraction index, e.g. token) and lane s -> &img[k0 + s/4][n0 + 4*(s%4)], lane i gets
// img[k0 .. k0+3][n0 + i]: four CONSECUTIVE k for column n0 + i, i.e. half of a v_mfma_f32_16x16x32_bf16 operand fragment
// (lane (i, quad) needs k = 8*quad .. 8*quad+7: two reads, k0 = kb + 8*quad and + 4).  The row stride of img is free.
// That is what a weight-gradient GEMM over ROW-MAJOR dY [tokens][N] and X [tokens][K] needs (no operand transposes).
//
// What does ds_read_b64_tr_b16 deliver?  LDS holds img[r][c] = r * 64 + c (as raw 16-bit integers, 64 columns per row).
// Every lane passes its own byte address; the probe prints, for a few address patterns, which (r, c) each lane received.
//   hipcc --offload-arch=gfx950 -O2 -o tr_read_probe tr_read_probe.hip && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(int mode, uint16_t* out) {
  __shared__ uint16_t img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  int r, c;
  if (mode == 0) { r = 0; c = 0; }                                   // uniform address
  else if (mode == 1) { r = (l & 15) >> 2; c = (l & 3) * 4 + (l >> 4) * 16; }   // 16-lane group: 4 rows x 16 columns, lane -> (row i/4, 4 columns)
  else if (mode == 2) { r = l & 15; c = (l >> 4) * 4; }              // lane -> its own row, 4 columns per group
  else { r = (l & 3) + 4 * (l >> 4); c = ((l & 15) >> 2) * 4; }      // lane -> row i%4 (+4 per group), column block i/4
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)img + (unsigned)(r * 64 + c) * 2u;
  unsigned lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&lo), "=v"(hi) : "v"(addr));
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = (uint16_t)(v & 0xffff); out[l * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
  out[l * 4 + 2] = (uint16_t)((v >> 32) & 0xffff); out[l * 4 + 3] = (uint16_t)(v >> 48);
}
int main() {
  uint16_t* d; uint16_t h[256];
  hipMalloc(&d, 512);
  const char* names[4] = {"uniform address (0,0)", "group = [4 rows][16 cols], lane i -> (row i/4, cols 4*(i%4)..), groups 16 cols apart",
                          "lane i -> (row i, cols 4*group..)", "lane i -> (row i%4 + 4*group, cols 4*(i/4)..)"};
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d: %s\n", mode, names[mode]);
    for (int l = 0; l < 64; ++l) {
      if (l % 16 == 0) printf("  lanes %2d..%2d:", l, l + 15);
      if (l % 16 < 6 || l % 16 == 15) { printf("  l%d:", l); for (int j = 0; j < 4; ++j) printf("(%d,%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64); }
      if (l % 16 == 15) printf("\n");
    }
  }
  return 0;
}
