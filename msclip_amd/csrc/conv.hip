// HBM-bound convolution front-ends that do not fit the implicit-GEMM loader:
//  * the two Cin=3 3x3/s2 convolutions that open the stem and the parallel
//    branch (reference lib/models/clip_openai_pe_res_v1.py:1993-1995 and
//    2260-2273 via 2436) fused into ONE pass over the NCHW image: both read the
//    same pixels, BatchNorm is folded into weights/bias, ReLU fused, output is
//    two NHWC bf16 tensors;
//  * the lateral adapters' non-overlapping depthwise "patch pooling" conv
//    (kernel == stride, ibid. 1573-1581 with the PRALLEL_T2B_* geometry).
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

template <typename InT> __device__ __forceinline__ float ld_px(const InT* p);
template <> __device__ __forceinline__ float ld_px<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_px<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// One thread per output pixel, all 2*C1 output channels in registers; the weights are
// wave-uniform (compile-time indices) so they stream through the scalar cache.
template <typename InT, int C1>
__global__ __launch_bounds__(256) void stem_dual_kernel(const InT* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out_a,
                                                        bf16_t* __restrict__ out_b, int B, int H, int W, int Ho,
                                                        int Wo) {
  constexpr int CO = 2 * C1;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Ho * Wo;
  if (p >= total) return;
  const int b = (int)(p / (Ho * Wo));
  const int r = (int)(p - (long long)b * Ho * Wo);
  const int ho = r / Wo, wo = r - ho * Wo;

  float xin[27];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = 2 * ho - 1 + kh, iw = 2 * wo - 1 + kw;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        xin[ci * 9 + kh * 3 + kw] = ok ? ld_px<InT>(img + (((size_t)b * 3 + ci) * H + ih) * W + iw) : 0.f;
      }
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = bias[c];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = fmaf(xin[t], w[t * CO + c], acc[c]);

  bf16_t* oa = out_a + (size_t)p * C1;
  bf16_t* ob = out_b + (size_t)p * C1;
#pragma unroll
  for (int c = 0; c < C1; c += 8) {
    uint4 u, v;
    u.x = pack_bf16x2(fmaxf(acc[c + 0], 0.f), fmaxf(acc[c + 1], 0.f));
    u.y = pack_bf16x2(fmaxf(acc[c + 2], 0.f), fmaxf(acc[c + 3], 0.f));
    u.z = pack_bf16x2(fmaxf(acc[c + 4], 0.f), fmaxf(acc[c + 5], 0.f));
    u.w = pack_bf16x2(fmaxf(acc[c + 6], 0.f), fmaxf(acc[c + 7], 0.f));
    v.x = pack_bf16x2(fmaxf(acc[C1 + c + 0], 0.f), fmaxf(acc[C1 + c + 1], 0.f));
    v.y = pack_bf16x2(fmaxf(acc[C1 + c + 2], 0.f), fmaxf(acc[C1 + c + 3], 0.f));
    v.z = pack_bf16x2(fmaxf(acc[C1 + c + 4], 0.f), fmaxf(acc[C1 + c + 5], 0.f));
    v.w = pack_bf16x2(fmaxf(acc[C1 + c + 6], 0.f), fmaxf(acc[C1 + c + 7], 0.f));
    *(uint4*)(oa + c) = u;
    *(uint4*)(ob + c) = v;
  }
}


// MFMA version of the same pass (what ships; the scalar kernel above stays as the cross-check of the parity test).
// out[p, 0..95] = relu(bias + sum_{k<27} x[p, k] * w[k, :]) is a GEMM with K = 27 (padded to 32): a wave takes 32
// consecutive output pixels, builds their im2col rows directly in the MFMA operand layout (lane = pixel, 16 of the
// 32 taps per lane: 16 scalar loads from the NCHW image, requested one block ahead), multiplies them with the
// 96 x 32 filter bank held in registers as bf16 (6 v_mfma_f32_32x32x16_bf16) and writes the two NHWC outputs through
// a 6-KiB staging block as two contiguous 3-KiB runs (consecutive pixels are adjacent in NHWC).
// The pass is HBM-bound: 4 B/pixel/channel in, 2 x 96 B per output pixel out.
// RAW (msclip_stem_conv3x3s2_dual_raw; the training step under train-mode BatchNorm, which normalises the RAW convolution outputs
// with batch statistics): no bias, no ReLU, fp32 outputs [pixels][48] -- the staging planes and the output runs are twice as long.
// BN_STATS / BN_NORM (round 6; msclip_stem_conv3x3s2_dual_stats / _norm): the two-pass form of the same train-mode BatchNorm.  The raw
// fp32 maps are never written: pass 1 runs the convolutions for their per-channel sums only (sum x, sum x^2 per wave ->
// part [waves][conv][2][48], folded by msclip_colsum), pass 2 runs them again and normalises in the epilogue -- y = relu(x scale +
// shift), the same expression msclip_bn_apply evaluates, and xhat = x a + b (a = rstd, b = -mean rstd), both bf16: xhat is what the
// backward reads instead of the raw map.  18 -> 8 bytes per map element over the forward (the image is read twice: 0.3 GB).
enum { STEM_FOLDED = 0, STEM_RAW = 1, STEM_BN_STATS = 2, STEM_BN_NORM = 3 };
template <typename InT, int MODE = STEM_FOLDED>
__global__ __launch_bounds__(256) void stem_dual_mfma_kernel(const InT* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, void* __restrict__ out_a_,
                                                             void* __restrict__ out_b_, int B, int H, int W, int Ho,
                                                             int Wo, void* __restrict__ out_c_ = nullptr,
                                                             void* __restrict__ out_d_ = nullptr, float* __restrict__ part = nullptr,
                                                             const float* __restrict__ bias_b = nullptr) {
  constexpr bool RAW = MODE == STEM_RAW;
  constexpr int PLANE = RAW ? 6144 : 3072;            // bytes of one staged output: 32 pixels x 48 channels
  constexpr int NPLANE = MODE == STEM_BN_NORM ? 4 : 2;
  __shared__ __attribute__((aligned(16))) char stg_all[MODE == STEM_BN_STATS ? 16 : 4 * NPLANE * PLANE];
  // FOLDED: bias [96]; BN_NORM: [4][96] = scale, shift, a, b per channel (conv a's 48 channels, then conv b's)
  __shared__ __attribute__((aligned(16))) float bias_l[MODE == STEM_BN_NORM ? 384 : 96];
  if (MODE == STEM_BN_NORM) {
    // bias / bias_b = msclip_bn_finish's rows [5][48] (mean, var, rstd, scale, shift) of conv a / conv b
    if (threadIdx.x < 96) {
      const int c = threadIdx.x;
      const float* k = (c < 48 ? bias : bias_b) + (c < 48 ? c : c - 48);
      bias_l[c] = k[3 * 48];
      bias_l[96 + c] = k[4 * 48];
      bias_l[192 + c] = k[2 * 48];
      bias_l[288 + c] = -k[0] * k[2 * 48];
    }
  } else if (threadIdx.x < 96) {
    bias_l[threadIdx.x] = MODE == STEM_FOLDED ? bias[threadIdx.x] : 0.f;
  }
  __syncthreads();
  char* out_a = (char*)out_a_;
  char* out_b = (char*)out_b_;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* stg = stg_all + (MODE == STEM_BN_STATS ? 0 : wave * NPLANE * PLANE);
  const int fr = lane & 31, fhi = lane >> 5;

  // filter bank as A fragments: tile t (32 output channels), k-step s (16 taps): lane (fr, fhi) holds taps
  // s*16 + fhi*8 .. +8 of channel t*32 + fr
  bf16x8 wf[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = s2 * 16 + fhi * 8 + e;
        wf[t][s2][e] = (__bf16)(k < 27 ? w[k * 96 + t * 32 + fr] : 0.f);
      }
  // this lane's 16 taps: j < 8 -> tap fhi*8 + j, j >= 8 -> tap 16 + fhi*8 + (j - 8); offset inside the image + (kh, kw)
  int toff[16];
  unsigned top = 0, left = 0;                        // bit j: tap j sits in the window's top row / left column
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = (j < 8 ? 0 : 16) + fhi * 8 + (j & 7);
    const int ci = k / 9, kh = (k % 9) / 3, kw = k % 3;
    toff[j] = k < 27 ? (ci * H + kh) * W + kw : -1;
    top |= (unsigned)(kh == 0) << j;
    left |= (unsigned)(kw == 0) << j;
  }

  const long long total = (long long)B * Ho * Wo;
  const long long nblk = (total + 31) / 32;
  const long long stride = (long long)gridDim.x * 4;
  auto load_block = [&](long long blk, float (&x)[16]) {
    long long p = blk * 32 + fr;
    p = p < total ? p : total - 1;
    const int b = (int)(p / (Ho * Wo));
    const int r = (int)(p - (long long)b * Ho * Wo);
    const int ho = r / Wo, wo = r - ho * Wo;
    const InT* base = img + ((size_t)b * 3 * H + (2 * ho - 1)) * W + (2 * wo - 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // stride 2, pad 1, even H and W: only the top row at ho == 0 and the left column at wo == 0 leave the image
      const bool ok = toff[j] >= 0 && !(ho == 0 && ((top >> j) & 1)) && !(wo == 0 && ((left >> j) & 1));
      x[j] = ok ? ld_px<InT>(base + toff[j]) : 0.f;
    }
  };

  // BN_STATS: this lane's running sums over its pixel column, per owned channel (3 tiles x 16)
  float ssum[MODE == STEM_BN_STATS ? 48 : 1], qsum[MODE == STEM_BN_STATS ? 48 : 1];
  if constexpr (MODE == STEM_BN_STATS) {
#pragma unroll
    for (int i = 0; i < 48; ++i) ssum[i] = qsum[i] = 0.f;
  }

  long long blk = (long long)blockIdx.x * 4 + wave;
  float xc[16], xn[16];
  if (blk < nblk) load_block(blk, xc);
  for (; blk < nblk; blk += stride) {
    if (blk + stride < nblk) load_block(blk + stride, xn);
    bf16x8 xf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[s2][e] = (__bf16)xc[s2 * 8 + e];
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == STEM_FOLDED) b4 = *(const float4*)(bias_l + t * 32 + g * 8 + fhi * 4);
        acc[t][g * 4 + 0] = b4.x; acc[t][g * 4 + 1] = b4.y; acc[t][g * 4 + 2] = b4.z; acc[t][g * 4 + 3] = b4.w;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][s2], xf[s2], acc[t], 0, 0, 0);
    }
    const long long p0 = blk * 32;
    if constexpr (MODE == STEM_BN_STATS) {
      if (p0 + fr < total) {                                       // (the clamped pixels of the last block are duplicates)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            ssum[t * 16 + r] += acc[t][r];
            qsum[t * 16 + r] = fmaf(acc[t][r], acc[t][r], qsum[t * 16 + r]);
          }
      }
    } else {
      // stage: plane 0 = out_a rows [32][48], plane 1 = out_b rows; lane (fr, fhi) owns channels t*32 + g*8 + fhi*4 .. +4
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = t * 32 + g * 8 + fhi * 4;                  // never straddles 48
          const int plane = c >= 48, cc = c - plane * 48;
          if constexpr (RAW) {
            *(float4*)(stg + plane * PLANE + fr * 192 + cc * 4) =
                make_float4(acc[t][g * 4 + 0], acc[t][g * 4 + 1], acc[t][g * 4 + 2], acc[t][g * 4 + 3]);
          } else if constexpr (MODE == STEM_BN_NORM) {
            const float4 sc = *(const float4*)(bias_l + c), sh = *(const float4*)(bias_l + 96 + c);
            const float4 ka = *(const float4*)(bias_l + 192 + c), kb = *(const float4*)(bias_l + 288 + c);
            uint2 o, xh;
            o.x = pack_bf16x2(fmaxf(fmaf(acc[t][g * 4 + 0], sc.x, sh.x), 0.f), fmaxf(fmaf(acc[t][g * 4 + 1], sc.y, sh.y), 0.f));
            o.y = pack_bf16x2(fmaxf(fmaf(acc[t][g * 4 + 2], sc.z, sh.z), 0.f), fmaxf(fmaf(acc[t][g * 4 + 3], sc.w, sh.w), 0.f));
            xh.x = pack_bf16x2(fmaf(acc[t][g * 4 + 0], ka.x, kb.x), fmaf(acc[t][g * 4 + 1], ka.y, kb.y));
            xh.y = pack_bf16x2(fmaf(acc[t][g * 4 + 2], ka.z, kb.z), fmaf(acc[t][g * 4 + 3], ka.w, kb.w));
            *(uint2*)(stg + plane * PLANE + fr * 96 + cc * 2) = o;
            *(uint2*)(stg + (2 + plane) * PLANE + fr * 96 + cc * 2) = xh;
          } else {
            uint2 o;
            o.x = pack_bf16x2(fmaxf(acc[t][g * 4 + 0], 0.f), fmaxf(acc[t][g * 4 + 1], 0.f));
            o.y = pack_bf16x2(fmaxf(acc[t][g * 4 + 2], 0.f), fmaxf(acc[t][g * 4 + 3], 0.f));
            *(uint2*)(stg + plane * PLANE + fr * 96 + cc * 2) = o;
          }
        }
      constexpr int PXB = RAW ? 192 : 96;                          // bytes per pixel of one output; PXB / 16 chunks per pixel
#pragma unroll
      for (int i = 0; i < PLANE / 1024; ++i) {
        const int ch = i * 64 + lane;                              // 16-byte chunk of the plane's contiguous run
        if (p0 + ch / (PXB / 16) < total) {
          *(uint4*)(out_a + (size_t)p0 * PXB + ch * 16) = *(const uint4*)(stg + ch * 16);
          *(uint4*)(out_b + (size_t)p0 * PXB + ch * 16) = *(const uint4*)(stg + PLANE + ch * 16);
          if constexpr (MODE == STEM_BN_NORM) {
            *(uint4*)((char*)out_c_ + (size_t)p0 * PXB + ch * 16) = *(const uint4*)(stg + 2 * PLANE + ch * 16);
            *(uint4*)((char*)out_d_ + (size_t)p0 * PXB + ch * 16) = *(const uint4*)(stg + 3 * PLANE + ch * 16);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) xc[j] = xn[j];
  }
  if constexpr (MODE == STEM_BN_STATS) {
    // fold the 32 pixel columns that share fhi; lanes 0 / 32 write their 48 channels: part [wave][conv][2][48]
#pragma unroll
    for (int i = 0; i < 48; ++i) {
      float a = ssum[i], q = qsum[i];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        q += __shfl_xor(q, m, 64);
      }
      ssum[i] = a;
      qsum[i] = q;
    }
    if (fr == 0) {
      float* pw = part + ((size_t)blockIdx.x * 4 + wave) * 192;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;   // accumulator row r of tile t in this half-wave
          const int conv = c >= 48, cc = c - conv * 48;
          pw[conv * 96 + cc] = ssum[t * 16 + r];
          pw[conv * 96 + 48 + cc] = qsum[t * 16 + r];
        }
    }
  }
}

// out[b, gy, gx, c] = sum_{ky,kx<k} w[ky*k+kx][c] * top[b, gy*k+ky, gx*k+kx, c]   (NHWC bf16, k == stride, pad 0)
__global__ __launch_bounds__(256) void dwpool_kernel(const bf16_t* __restrict__ top, const float* __restrict__ w,
                                                     bf16_t* __restrict__ out, int ldo, int B, int H, int W, int C,
                                                     int k, int g) {
  const int nch = C >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * g * g * nch;
  if (idx >= total) return;
  const int cc = (int)(idx % nch);
  const long long pidx = idx / nch;
  const int b = (int)(pidx / (g * g));
  const int pr = (int)(pidx - (long long)b * g * g);
  const int gy = pr / g, gx = pr - gy * g;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const bf16_t* rowp = top + (((size_t)b * H + gy * k + ky) * W + (size_t)gx * k) * C + cc * 8;
    const float* wr = w + (size_t)(ky * k) * C + cc * 8;
#pragma unroll 4
    for (int kx = 0; kx < k; ++kx) {
      const uint4 u = *(const uint4*)(rowp + (size_t)kx * C);
      const float4 w0 = *(const float4*)(wr + (size_t)kx * C);
      const float4 w1 = *(const float4*)(wr + (size_t)kx * C + 4);
      float f[8];
      unpack_bf16x8(u, f);
      acc[0] = fmaf(f[0], w0.x, acc[0]); acc[1] = fmaf(f[1], w0.y, acc[1]);
      acc[2] = fmaf(f[2], w0.z, acc[2]); acc[3] = fmaf(f[3], w0.w, acc[3]);
      acc[4] = fmaf(f[4], w1.x, acc[4]); acc[5] = fmaf(f[5], w1.y, acc[5]);
      acc[6] = fmaf(f[6], w1.z, acc[6]); acc[7] = fmaf(f[7], w1.w, acc[7]);
    }
  }
  uint4 o;
  o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
  o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
  *(uint4*)(out + (size_t)pidx * ldo + cc * 8) = o;
}


// Row-streaming version of the same pooling (what ships when k * C <= 768: 768 for the 32-pixel patch grid, 384 for
// the 16-pixel one).  A window row top[b, gy*k + ky, gx*k .. gx*k + k, :] is k * C contiguous bf16 whatever the stage,
// and the weights of that row, w[ky*k .. ky*k + k][:], are the matching k * C contiguous floats: a wave takes one output
// position, lane l multiplies 16-byte chunks l and l + 64 of every window row (coalesced full-line loads; the
// thread-per-channel-chunk kernel above reads 96-byte pieces 1.5 KB apart and reaches 2.4 TB/s at k = 16) against
// the weight table held in LDS, and the k partial sums of a channel are added up through a 3-KiB LDS block.
__global__ __launch_bounds__(256) void dwpool_rows_kernel(const bf16_t* __restrict__ top, const float* __restrict__ w,
                                                          bf16_t* __restrict__ out, int ldo, int B, int H, int W,
                                                          int C, int k, int g) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int RL = k * C;                              // window-row length in elements (<= 768)
  float* wl = dsm;                                   // [k][RL]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* red = dsm + k * RL + wave * RL;             // [RL / 8 chunks][8]
  for (int i = threadIdx.x; i < k * RL / 4; i += 256) *(float4*)(wl + i * 4) = *(const float4*)(w + i * 4);
  __syncthreads();
  const int npos = B * g * g;
  const bool one = lane * 8 < RL, two = (lane + 64) * 8 < RL;   // chunks l and l + 64 of the row
  for (int pos = blockIdx.x * 4 + wave; pos < npos; pos += gridDim.x * 4) {
    const int b = pos / (g * g), pr = pos - b * g * g;
    const int gy = pr / g, gx = pr - gy * g;
    const bf16_t* base = top + (((size_t)b * H + (size_t)gy * k) * W + (size_t)gx * k) * C;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
    // window rows in groups of four: the eight 16-byte loads of a group are issued together (unconditional, clamped
    // chunk index: no exec-mask branch between them) before the group's multiply-adds
    const int c0 = one ? lane * 8 : 0, c1 = two ? (lane + 64) * 8 : 0;
    for (int ky0 = 0; ky0 < k; ky0 += 4) {
      uint4 q0[4], q1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q0[j] = q1[j] = make_uint4(0, 0, 0, 0);
        if (ky0 + j < k) {                             // wave-uniform (k = 1 and 2 have shorter groups)
          const bf16_t* row = base + (size_t)(ky0 + j) * W * C;
          q0[j] = *(const uint4*)(row + c0);
          q1[j] = *(const uint4*)(row + c1);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
      if (ky0 + j < k) {
      const int ky = ky0 + j;
      const float* wr = wl + ky * RL;
      const uint4 u0 = one ? q0[j] : make_uint4(0, 0, 0, 0);
      const uint4 u1 = q1[j];
      float f[8];
      unpack_bf16x8(u0, f);
      const int wo = one ? lane * 8 : 0;
      const float4 w0 = *(const float4*)(wr + wo), w1 = *(const float4*)(wr + wo + 4);
      a0[0] = fmaf(f[0], w0.x, a0[0]); a0[1] = fmaf(f[1], w0.y, a0[1]); a0[2] = fmaf(f[2], w0.z, a0[2]);
      a0[3] = fmaf(f[3], w0.w, a0[3]); a0[4] = fmaf(f[4], w1.x, a0[4]); a0[5] = fmaf(f[5], w1.y, a0[5]);
      a0[6] = fmaf(f[6], w1.z, a0[6]); a0[7] = fmaf(f[7], w1.w, a0[7]);
      if (two) {
        unpack_bf16x8(u1, f);
        const float4 x0 = *(const float4*)(wr + (lane + 64) * 8), x1 = *(const float4*)(wr + (lane + 64) * 8 + 4);
        a1[0] = fmaf(f[0], x0.x, a1[0]); a1[1] = fmaf(f[1], x0.y, a1[1]); a1[2] = fmaf(f[2], x0.z, a1[2]);
        a1[3] = fmaf(f[3], x0.w, a1[3]); a1[4] = fmaf(f[4], x1.x, a1[4]); a1[5] = fmaf(f[5], x1.y, a1[5]);
        a1[6] = fmaf(f[6], x1.z, a1[6]); a1[7] = fmaf(f[7], x1.w, a1[7]);
      }
      }
      }
    }
    if (one) {
      *(float4*)(red + lane * 8) = make_float4(a0[0], a0[1], a0[2], a0[3]);
      *(float4*)(red + lane * 8 + 4) = make_float4(a0[4], a0[5], a0[6], a0[7]);
    }
    if (two) {
      *(float4*)(red + (lane + 64) * 8) = make_float4(a1[0], a1[1], a1[2], a1[3]);
      *(float4*)(red + (lane + 64) * 8 + 4) = make_float4(a1[4], a1[5], a1[6], a1[7]);
    }
    __builtin_amdgcn_wave_barrier();
    // element x of the window row belongs to channel x % C; channel c = sum over kx of red[kx * C + c]
    bf16_t* orow = out + (size_t)pos * ldo;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int kx = 0; kx < k; ++kx) {
        const float4 v = *(const float4*)(red + kx * C + c0);
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
      }
      uint2 o;
      o.x = pack_bf16x2(s4.x, s4.y);
      o.y = pack_bf16x2(s4.z, s4.w);
      *(uint2*)(orow + c0) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// Patch matrix of a kernel == stride == P convolution over an NCHW image (the plain patch conv of M.py:2502-2508, 2657:
// BASELINE config C5's ViT-L/14): out[b*g*g + py*g + px][c*P*P + kh*P + kw] = bf16(img[b][c][py*P + kh][px*P + kw]), columns
// [3*P*P, kpad) zero.  One workgroup per (image, patch row): it reads the 3*P image rows of that patch row whole (W contiguous
// pixels each: coalesced), lays the g output rows out in LDS and writes them as one contiguous run of g * kpad bf16 (the g patches
// of a patch row are consecutive rows of the matrix).  HBM-bound: image once in, matrix once out.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ img, bf16_t* __restrict__ out, int kpad, int H, int W,
                                                       int P, int g) {
  extern __shared__ bf16_t tile[];                 // [g][kpad]
  const int b = blockIdx.x / g, py = blockIdx.x - b * g;
  const int kp = 3 * P * P;
  for (int i = threadIdx.x; i < g * (kpad - kp); i += 256) {
    const int r = i / (kpad - kp);
    tile[r * kpad + kp + (i - r * (kpad - kp))] = 0;
  }
  for (int cr = 0; cr < 3 * P; ++cr) {             // (channel, row inside the patch)
    const int c = cr / P, kh = cr - c * P;
    const T* src = img + (((size_t)b * 3 + c) * H + py * P + kh) * W;
    for (int x = threadIdx.x; x < g * P; x += 256) {
      const int px = x / P, kw = x - px * P;
      float v;
      if constexpr (sizeof(T) == 2) v = bf16_to_f32(src[x]);
      else v = src[x];
      tile[px * kpad + c * P * P + kh * P + kw] = f32_to_bf16(v);
    }
  }
  __syncthreads();
  const uint4* t4 = (const uint4*)tile;
  uint4* o4 = (uint4*)(out + ((size_t)b * g * g + (size_t)py * g) * kpad);
  for (int i = threadIdx.x; i < g * kpad / 8; i += 256) o4[i] = t4[i];
}
}  // namespace

extern "C" int msclip_stem_conv3x3s2_dual_raw(const void* img, int img_is_bf16, const float* w, float* out_a, float* out_b, int B,
                                              int H, int W, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_stem_conv3x3s2_dual_raw, stream, img, img_is_bf16, w, out_a, out_b, B, H, W);
  if (!img || !w || !out_a || !out_b || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return MSCLIP_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long nblk = ((long long)B * Ho * Wo + 31) / 32;
  long long g = (nblk + 3) / 4;
  if (g > 256 * 3) g = 256 * 3;                                  // 48 KB of staging per workgroup: three per CU
  const dim3 grid((unsigned)g), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (img_is_bf16)
    hipLaunchKernelGGL((stem_dual_mfma_kernel<bf16_t, STEM_RAW>), grid, blk, 0, st, (const bf16_t*)img, w, (const float*)nullptr,
                       (void*)out_a, (void*)out_b, B, H, W, Ho, Wo);
  else
    hipLaunchKernelGGL((stem_dual_mfma_kernel<float, STEM_RAW>), grid, blk, 0, st, (const float*)img, w, (const float*)nullptr,
                       (void*)out_a, (void*)out_b, B, H, W, Ho, Wo);
  return msclip_launch_status();
}

// Two-pass train-mode BatchNorm over the two image convolutions (include/msclip_hip.h).  Pass 1: part [waves][2 convs][2][48].
extern "C" int msclip_stem_conv3x3s2_dual_stats(const void* img, int img_is_bf16, const float* w, float* part, int part_waves, int B,
                                                int H, int W, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_stem_conv3x3s2_dual_stats, stream, img, img_is_bf16, w, part, part_waves, B, H, W);
  if (!img || !w || !part || part_waves < 4 || (part_waves & 3) || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return MSCLIP_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const dim3 grid((unsigned)(part_waves / 4)), blk(256);          // every wave writes its row (zeros when it had no block)
  hipStream_t st = (hipStream_t)stream;
  if (img_is_bf16)
    hipLaunchKernelGGL((stem_dual_mfma_kernel<bf16_t, STEM_BN_STATS>), grid, blk, 0, st, (const bf16_t*)img, w, (const float*)nullptr,
                       (void*)nullptr, (void*)nullptr, B, H, W, Ho, Wo, (void*)nullptr, (void*)nullptr, part);
  else
    hipLaunchKernelGGL((stem_dual_mfma_kernel<float, STEM_BN_STATS>), grid, blk, 0, st, (const float*)img, w, (const float*)nullptr,
                       (void*)nullptr, (void*)nullptr, B, H, W, Ho, Wo, (void*)nullptr, (void*)nullptr, part);
  return msclip_launch_status();
}

// Pass 2: stats_a / stats_b [5][48] = msclip_bn_finish's rows (mean, var, rstd, scale, shift) of conv a / conv b; y_* = relu(x scale +
// shift), xhat_* = (x - mean) rstd (bf16 [pixels][48] each).
extern "C" int msclip_stem_conv3x3s2_dual_norm(const void* img, int img_is_bf16, const float* w, const float* stats_a,
                                               const float* stats_b, void* y_a, void* y_b, void* xhat_a, void* xhat_b, int B, int H,
                                               int W, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_stem_conv3x3s2_dual_norm, stream, img, img_is_bf16, w, stats_a, stats_b, y_a, y_b, xhat_a, xhat_b, B, H, W);
  if (!img || !w || !stats_a || !stats_b || !y_a || !y_b || !xhat_a || !xhat_b || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1))
    return MSCLIP_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long nblk = ((long long)B * Ho * Wo + 31) / 32;
  long long g = (nblk + 3) / 4;
  if (g > 256 * 3) g = 256 * 3;                                  // 48 KB of staging per workgroup: three per CU
  const dim3 grid((unsigned)g), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (img_is_bf16)
    hipLaunchKernelGGL((stem_dual_mfma_kernel<bf16_t, STEM_BN_NORM>), grid, blk, 0, st, (const bf16_t*)img, w, stats_a, y_a, y_b, B, H, W,
                       Ho, Wo, xhat_a, xhat_b, (float*)nullptr, stats_b);
  else
    hipLaunchKernelGGL((stem_dual_mfma_kernel<float, STEM_BN_NORM>), grid, blk, 0, st, (const float*)img, w, stats_a, y_a, y_b, B, H, W,
                       Ho, Wo, xhat_a, xhat_b, (float*)nullptr, stats_b);
  return msclip_launch_status();
}

extern "C" int msclip_stem_conv3x3s2_dual(const void* img, int img_is_bf16, const float* w, const float* bias,
                                          void* out_a, void* out_b, int B, int H, int W, int C1, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_stem_conv3x3s2_dual, stream, img, img_is_bf16, w, bias, out_a, out_b, B, H, W, C1);
  if (!img || !w || !bias || !out_a || !out_b || B <= 0 || (C1 != 48 && C1 != 64) || H < 2 || W < 2 || (H & 1) || (W & 1)) return MSCLIP_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)B * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  if (C1 == 64) {                                               // width 1024 (the ViT-L-width stand-in): 2 x 64 channels on the VALU kernel
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    if (img_is_bf16)
      hipLaunchKernelGGL((stem_dual_kernel<bf16_t, 64>), grid, blk, 0, st, (const bf16_t*)img, w, bias, (bf16_t*)out_a,
                         (bf16_t*)out_b, B, H, W, Ho, Wo);
    else
      hipLaunchKernelGGL((stem_dual_kernel<float, 64>), grid, blk, 0, st, (const float*)img, w, bias, (bf16_t*)out_a,
                         (bf16_t*)out_b, B, H, W, Ho, Wo);
    return msclip_launch_status();
  }
  const char* scalar = getenv("MSCLIP_STEM_SCALAR");            // the VALU kernel, for cross-checks only
  if (scalar && scalar[0] == '1') {
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    if (img_is_bf16)
      hipLaunchKernelGGL((stem_dual_kernel<bf16_t, 48>), grid, blk, 0, st, (const bf16_t*)img, w, bias, (bf16_t*)out_a,
                         (bf16_t*)out_b, B, H, W, Ho, Wo);
    else
      hipLaunchKernelGGL((stem_dual_kernel<float, 48>), grid, blk, 0, st, (const float*)img, w, bias, (bf16_t*)out_a,
                         (bf16_t*)out_b, B, H, W, Ho, Wo);
    return msclip_launch_status();
  }
  const long long nblk = (total + 31) / 32;
  long long g = (nblk + 3) / 4;
  if (g > 256 * 8) g = 256 * 8;                                  // persistent: up to 8 workgroups of 4 waves per CU
  const dim3 grid((unsigned)g), blk(256);
  if (img_is_bf16)
    hipLaunchKernelGGL((stem_dual_mfma_kernel<bf16_t>), grid, blk, 0, st, (const bf16_t*)img, w, bias, out_a, out_b, B, H, W, Ho,
                       Wo);
  else
    hipLaunchKernelGGL((stem_dual_mfma_kernel<float>), grid, blk, 0, st, (const float*)img, w, bias, (bf16_t*)out_a,
                       (bf16_t*)out_b, B, H, W, Ho, Wo);
  return msclip_launch_status();
}

extern "C" int msclip_dwpool(const void* top, const float* w, void* out, int ldo, int B, int H, int W, int C, int k,
                             void* stream) {
  MSCLIP_PLAN_HOOK(msclip_dwpool, stream, top, w, out, ldo, B, H, W, C, k);
  if (!top || !w || !out || B <= 0 || k <= 0 || (C % 8) || (ldo % 8) || (H % k) || H != W) return MSCLIP_EINVAL;
  const int g = H / k;
  const char* oldk = getenv("MSCLIP_DWPOOL_SCALAR");            // the thread-per-chunk kernel, for cross-checks only
  if (k >= 4 && k * C <= 768 && !(oldk && oldk[0] == '1')) {   // k <= 2: the per-chunk kernel is as fast
    const int npos = B * g * g;
    int grid = (npos + 3) / 4;
    if (grid > 256 * 4) grid = 256 * 4;
    const size_t lds = (size_t)(k * k * C + 4 * k * C) * sizeof(float);
    if (lds > 65536) return MSCLIP_EINVAL;
    hipLaunchKernelGGL(dwpool_rows_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)top, w,
                       (bf16_t*)out, ldo, B, H, W, C, k, g);
    return msclip_launch_status();
  }
  const long long total = (long long)B * g * g * (C / 8);
  hipLaunchKernelGGL(dwpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)top, w, (bf16_t*)out, ldo, B, H, W, C, k, g);
  return msclip_launch_status();
}

extern "C" int msclip_patchify(const void* img, int img_is_bf16, void* out, int kpad, int B, int H, int W, int P, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_patchify, stream, img, img_is_bf16, out, kpad, B, H, W, P);
  if (!img || !out || B <= 0 || P <= 0 || H != W || (H % P) || kpad < 3 * P * P || (kpad % 64)) return MSCLIP_EINVAL;
  const int g = H / P;
  const size_t lds = (size_t)g * kpad * sizeof(bf16_t);
  if (lds > 65536) return MSCLIP_EINVAL;
  const dim3 grid((unsigned)(B * g)), blk(256);
  if (img_is_bf16)
    hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, blk, lds, (hipStream_t)stream, (const bf16_t*)img, (bf16_t*)out, kpad, H, W, P, g);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, grid, blk, lds, (hipStream_t)stream, (const float*)img, (bf16_t*)out, kpad, H, W, P, g);
  return msclip_launch_status();
}
