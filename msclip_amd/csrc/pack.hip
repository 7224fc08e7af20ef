// Re-packing the convolutional side's weights after an optimizer step in ONE launch (round 5).
//
// The engine's operands of the conv stem, the parallel branch and the lateral adapters are derived tensors: eval-mode
// BatchNorm folded into the filter (M.py:1825-1861, 1920-1936), the stem stages' 1x1 shortcut merged into the 3x3 centre tap,
// filters laid out [cout][KH*KW*Cin padded to 64] in bf16 for the gathering GEMM or transposed in fp32 for the fused front /
// adapter kernels, biases = BatchNorm shifts.  msclip_amd/packing.py states these folds as tensor algebra (device-agnostic: the
// CPU tests check them against the oracle) -- ~170 small ATen launches per re-pack, i.e. per training step, with the host
// behind the GPU at the step boundary.  Here a device-resident table of items describes every derived tensor once
// (msclip_amd/engine.py builds it after the first full pack; sources are the module's own parameter storage, destinations the
// engine's persistent operand tensors), and one kernel rewrites all of them in place.  The arithmetic is the tensor algebra's,
// operation by operation in IEEE fp32 (this file is compiled without fast-math and without contraction: build.sh): bitwise what
// packing.py computes on the CPU; torch's GPU division / square root are not correctly rounded, so against the GPU tensor
// algebra a BatchNorm scale can differ by one ulp.
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

// BatchNorm(x) = scale x + shift in eval mode: scale = gamma / sqrt(var + eps), shift = beta - mean scale
__device__ __forceinline__ float bn_scale(const float* g, const float* var, float eps, int c) {
  return __fdiv_rn(g[c], __fsqrt_rn(__fadd_rn(var[c], eps)));
}
__device__ __forceinline__ float bn_shift(const float* g, const float* b, const float* mu, const float* var, float eps, int c) {
  return __fsub_rn(b[c], __fmul_rn(mu[c], bn_scale(g, var, eps, c)));
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const msclip_pack_item* __restrict__ items,
                                                           const int* __restrict__ blk_start, int n_items) {
  // block -> item (blk_start is ascending, n_items + 1 entries)
  int lo = 0, hi = n_items;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const msclip_pack_item it = items[lo];
  const int b = blockIdx.x - blk_start[lo];
  const int taps = it.kh * it.kw, kreal = taps * it.ci;
  // a workgroup takes 1024 consecutive elements of ONE output-channel row (row length kpad / kreal): one division per workgroup
  // for the row, 32-bit tap arithmetic per element (the first version decoded a flat 64-bit index per element: 0.5 ms per launch)
  if (it.out) {
    const int rowlen = it.mode == 0 ? it.kpad : kreal;
    const int cpr = (rowlen + 1023) >> 10;           // chunks per row
    const int o = b / cpr, k0 = (b - o * cpr) << 10;
    if (o < it.co) {
      const float sc = it.g ? bn_scale(it.g, it.var, it.eps, o) : 1.f;
      const float sc2 = it.w2 ? bn_scale(it.g2, it.var2, it.eps2, o) : 0.f;
      const float* wrow = it.w + (size_t)o * kreal;
      for (int k = k0 + threadIdx.x; k < rowlen && k < k0 + 1024; k += 256) {
        if (it.mode == 0) {                          // bf16 [co][kpad], k = (kh, kw, ci) (the NHWC gather order), zero padded
          float v = 0.f;
          if (k < kreal) {
            const int tap = k / it.ci, c = k - tap * it.ci, y = tap / it.kw, x = tap - y * it.kw;
            v = wrow[(c * it.kh + y) * it.kw + x];
            if (it.g) v = __fmul_rn(v, sc);
            if (it.w2 && y == it.kh / 2 && x == it.kw / 2)    // the stem stage's 1x1 shortcut samples the 3x3 window's centre tap
              v = __fadd_rn(v, __fmul_rn(it.w2[(size_t)o * it.ci + c], sc2));
          }
          ((bf16_t*)it.out)[(size_t)o * it.kpad + k] = f32_to_bf16(v);
        } else {                                     // fp32, transposed: out[k][col0 + o], k = (ci, kh, kw) as the filter lies in memory
          float v = wrow[k];
          if (it.g) v = __fmul_rn(v, sc);
          ((float*)it.out)[(size_t)k * it.ld + it.col0 + o] = v;
        }
      }
    }
  }
  if (it.bias_out && it.bias_mode == 3) {
    // pointwise conv behind a BatchNorm (adapter): bias[o] = sum_c W[o][c] shift[c], shift over the INPUT channels (ci <= 1024).  A
    // workgroup takes 8 outputs (the item has ceil(co / 8) workgroups): the shifts once into LDS, a wave per two outputs.  (First
    // version: ONE workgroup per item walked all co x ci products with a division and a square root each -- the 0.5 ms tail of
    // the launch.)
    __shared__ float sh[1024];
    for (int c = threadIdx.x; c < it.ci; c += 256) sh[c] = bn_shift(it.g, it.b, it.mu, it.var, it.eps, c);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = b * 8 + wave * 2 + j;
      if (o < it.co) {
        float v = 0.f;
        for (int c = lane; c < it.ci; c += 64) v += it.w[(size_t)o * it.ci + c] * sh[c];
        v = wave_sum(v);
        if (lane == 0) it.bias_out[it.bias_col0 + o] = v;
      }
    }
  } else if (b == 0 && it.bias_out) {
    for (int o = threadIdx.x; o < it.co; o += 256) {
      float v = bn_shift(it.g, it.b, it.mu, it.var, it.eps, o);
      if (it.bias_mode == 2) v = __fadd_rn(v, bn_shift(it.g2, it.b2, it.mu2, it.var2, it.eps2, o));
      it.bias_out[it.bias_col0 + o] = v;
    }
  }
}

}  // namespace

extern "C" int msclip_pack_weights(const msclip_pack_item* items_dev, const int* blk_start_dev, int n_items, int n_blocks,
                                   void* stream) {
  MSCLIP_PLAN_HOOK(msclip_pack_weights, stream, items_dev, blk_start_dev, n_items, n_blocks);
  if (!items_dev || !blk_start_dev || n_items <= 0 || n_blocks <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, blk_start_dev, n_items);
  return msclip_launch_status();
}
