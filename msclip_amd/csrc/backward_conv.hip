// Backward of the convolutional side of the vision tower: the EarlyconvRes stem, the parallel
// convolutional branch and the top-down / depthwise halves of the lateral adapters (reference M.py:1898-2000,
// 1812-1895, 1752-1778).  The contractions themselves run on the MFMA GEMM (msclip_gemm): a convolution's weight
// gradient is dY^T . im2col(X) and its input gradient is col2im(dY . W), so what lives here is the data movement around
// those GEMMs and the small depthwise reductions:
//
//   msclip_im2col          NHWC bf16 (or the NCHW fp32 / bf16 input image) -> patch matrix [pixels, Kp] bf16,
//                          K index (kh*KW + kw)*C + ci like the packed forward weights (packing.conv_weight_matrix)
//   msclip_col2im          patch-matrix gradient -> NHWC bf16 input gradient (gather form: no atomics, deterministic)
//   msclip_relu_bwd        (dy [+ dy2]) * (y > 0)
//   msclip_dwpool_bwd      input gradient of the adapters' kernel == stride depthwise conv
//   msclip_dwpool_wgrad    its filter gradient, per-slab partial sums [S][k*k][C] (the caller folds them with msclip_colsum)
//   msclip_dw3x3_wgrad     filter gradient of the depthwise 3x3 over the token grid, per-slab partials [S][9][C]
//
// All of them are HBM-bound streaming kernels: 16-byte accesses, consecutive lanes on consecutive channel chunks.
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

static int grid_for(size_t n, int per_block, int cap = 1 << 20) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

// one thread = one (pixel, 8-wide K chunk); grid-stride over pixels*Kp/8
template <int KIND>   // 0: NHWC bf16 with C % 8 == 0 (vector path), 1: NCHW fp32 image, 2: NCHW bf16 image, 3: NHWC bf16 any C
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ xin, bf16_t* __restrict__ col, int B, int H,
                                                     int W, int C, int KH, int KW, int stride, int pad, int Ho, int Wo,
                                                     int Kp) {
  const int kc = Kp >> 3, K = KH * KW * C;
  const size_t total = (size_t)B * Ho * Wo * kc;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c8 = (int)(idx % kc);
    const size_t pix = idx / kc;
    const int b = (int)(pix / ((size_t)Ho * Wo));
    const int rem = (int)(pix - (size_t)b * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (KIND == 0) {
      const int k0 = c8 * 8;
      if (k0 < K) {
        const int tap = k0 / C, ci = k0 - tap * C;
        const int kh = tap / KW, kw = tap - kh * KW;
        const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
          v = *(const uint4*)((const bf16_t*)xin + (((size_t)b * H + ih) * W + iw) * C + ci);
      }
    } else {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = c8 * 8 + j;
        e[j] = 0;
        if (k < K) {
          const int tap = k / C, ci = k - tap * C;
          const int kh = tap / KW, kw = tap - kh * KW;
          const int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
          if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
            if (KIND == 1) e[j] = f32_to_bf16(((const float*)xin)[(((size_t)b * C + ci) * H + ih) * W + iw]);
            else if (KIND == 2) e[j] = ((const bf16_t*)xin)[(((size_t)b * C + ci) * H + ih) * W + iw];
            else e[j] = ((const bf16_t*)xin)[(((size_t)b * H + ih) * W + iw) * C + ci];
          }
        }
      }
      v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
    }
    *(uint4*)(col + pix * Kp + (size_t)c8 * 8) = v;
  }
}

// The NCHW input image (KIND 1 fp32 / 2 bf16; 3 channels, 3 x 3 / stride 2 in the stems): one workgroup per (image, output row).
// The C * KH input rows it reads go through LDS with coalesced loads (the per-thread gather above issues 8 scattered 4-byte
// loads per 16-byte store: 0.8 ms for the 512 x 224 x 224 batch, 1.4 TB/s of its bytes); a thread then assembles (pixel, 8-wide K
// chunk) items from LDS: consecutive threads write consecutive 16-byte pieces of the patch matrix.  Same values, bit for bit.
template <int KIND>
__global__ __launch_bounds__(256) void im2col_image_rows_kernel(const void* __restrict__ xin, bf16_t* __restrict__ col, int H,
                                                                int W, int C, int KH, int KW, int stride, int pad, int Ho,
                                                                int Wo, int Kp) {
  extern __shared__ float img_rows[];                  // [C * KH][W]
  const int b = blockIdx.x / Ho, oh = blockIdx.x - b * Ho;
  const int ih0 = oh * stride - pad;
  const int nrow = C * KH;
  if (KIND == 1 && !(W & 3) && !((size_t)xin & 15)) {
    const int w4 = W >> 2;
    for (int i = threadIdx.x; i < nrow * w4; i += 256) {
      const int r = i / w4, x4 = i - r * w4;
      const int ci = r / KH, kh = r - ci * KH, ih = ih0 + kh;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)ih < (unsigned)H) v = *(const float4*)((const float*)xin + (((size_t)b * C + ci) * H + ih) * W + x4 * 4);
      *(float4*)(img_rows + r * W + x4 * 4) = v;
    }
  } else {
    for (int i = threadIdx.x; i < nrow * W; i += 256) {
      const int r = i / W, x = i - r * W;
      const int ci = r / KH, kh = r - ci * KH, ih = ih0 + kh;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)H) {
        const size_t a = (((size_t)b * C + ci) * H + ih) * W + x;
        v = KIND == 1 ? ((const float*)xin)[a] : bf16_to_f32(((const bf16_t*)xin)[a]);
      }
      img_rows[i] = v;
    }
  }
  __syncthreads();
  const int kc = Kp >> 3, K = KH * KW * C;
  bf16_t* out = col + ((size_t)b * Ho + oh) * Wo * Kp;
  for (int it = threadIdx.x; it < Wo * kc; it += 256) {
    const int ow = it / kc, c8 = it - ow * kc;
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = c8 * 8 + j;
      e[j] = 0;
      if (k < K) {
        const int tap = k / C, ci = k - tap * C;
        const int kh = tap / KW, kw = tap - kh * KW;
        const int iw = ow * stride - pad + kw;
        // (rows outside the image were stored as zeros: bf16(0) = 0, what the gather kernel writes for them)
        if ((unsigned)iw < (unsigned)W) e[j] = f32_to_bf16(img_rows[(ci * KH + kh) * W + iw]);
      }
    }
    uint4 v;
    v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
    v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
    *(uint4*)(out + (size_t)it * 8) = v;
  }
}

// ---- Weight (and bias) gradient of a 3 x 3 / stride 2 / pad 1 convolution ON THE INPUT IMAGE, without a patch matrix
// (msclip_image_conv_wgrad; the stem's conv1 and parallel stage 0, M.py:1898-1905 / 1812-1830: 3 input channels, 224 x 224 ->
// 112 x 112, 6.4 M output pixels at batch 512).  dW[co][tap] = sum_pix dY[pix][co] col[pix][tap] is a 48 x 27 result over a
// 6.4 M-deep contraction: a streaming reduction whose floor is one read of dY.  Through the generic path it was three passes --
// the patch matrix written (im2col, 0.32 ms) and read by a split-K GEMM built for wide operands (256-byte row pieces of a
// 64-byte row: 1.15 ms for 48 channels), the bias sums a third (0.19 ms) -- all on the main queue at the end of the backward.
// Here a workgroup takes (image, output row) units: the row's dY [Wo][co] and the three image rows x three channels it reads go
// to LDS; wave w contracts pixels 32 w .. 32 w + 31 (Wo <= 128, padded with zero dY rows) on v_mfma_f32_16x16x32_bf16 --
// A = dY^T fragments by ds_read_b64_tr_b16 from the row-major tile, B[tap][pixel] gathered from the fp32 image rows (stride-2
// LDS reads, rounded to bf16 as the patch matrix was); tap 27 is the constant 1, so column 27 of the result is sum dY, the bias
// gradient.  Accumulators stay in registers across a workgroup's units; part[block][co_pad][32] fp32, folded by msclip_colsum
// (fixed order).  ~20 KB of LDS and < 96 VGPRs: 5-6 workgroups per CU hide the load -> LDS -> MFMA chain of each other.
typedef __attribute__((ext_vector_type(4))) __bf16 icw_bf16x4;
__device__ __forceinline__ bf16x8 icw_ld_tr8(const char* p0, const char* p1) {
  const icw_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 icw_bf16x4*)p0);
  const icw_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 icw_bf16x4*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MT>                                      // co padded to 16 * MT output channels
__global__ __launch_bounds__(256) void image_conv_wgrad_kernel(const float* __restrict__ img, const bf16_t* __restrict__ dy, int lddy,
                                                               float* __restrict__ part, int B, int S, int Ho, int co,
                                                               int units_per_block) {
  constexpr int CP = 16 * MT, RS = CP + 8;             // dY tile row stride (elements): 16-byte aligned rows, tr-read friendly
  constexpr int IW = 264;                              // image row stride (floats): pixel iw sits at [iw + 4]
  __shared__ __attribute__((aligned(16))) bf16_t dyt[128 * RS];
  __shared__ __attribute__((aligned(16))) float rows[9 * IW];
  __shared__ float red[CP * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, quad = lane >> 4;
  const int Wo = Ho, W = S, H = S;
  // zero once: the dY rows of pixels >= Wo (and the channel padding), the image rows' borders
  for (int i = tid; i < 128 * RS / 2; i += 256) ((unsigned*)dyt)[i] = 0;
  for (int i = tid; i < 9 * IW; i += 256) rows[i] = 0.f;
  __syncthreads();
  f32x4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's two taps n = r16, 16 + r16: column (kh * 3 + kw) * 3 + ci of the patch matrix; 27 = the constant 1; 28.. = 0
  int roff[2], kind[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = nt * 16 + r16;
    const int tap = n / 3, ci = n - tap * 3, kh = tap / 3, kw = tap - kh * 3;
    kind[nt] = n < 27 ? 0 : (n == 27 ? 1 : 2);
    roff[nt] = n < 27 ? (ci * 3 + kh) * IW + kw - 1 + 4 : 0;
  }
  const int co8 = co >> 3, w4 = W >> 2;
  const int total = B * Ho;
  const int u0 = blockIdx.x * units_per_block, u1 = min(total, u0 + units_per_block);
  for (int u = u0; u < u1; ++u) {
    const int b = u / Ho, oh = u - b * Ho;
    // ---- this unit's dY row block and image rows -> LDS
    const bf16_t* dsrc = dy + (size_t)u * Wo * lddy;
    for (int i = tid; i < Wo * co8; i += 256) {
      const int px = i / co8, c8 = i - px * co8;
      *(uint4*)(dyt + px * RS + c8 * 8) = *(const uint4*)(dsrc + (size_t)px * lddy + c8 * 8);
    }
    const int ih0 = oh * 2 - 1;
    for (int i = tid; i < 9 * w4; i += 256) {
      const int r = i / w4, x4 = i - r * w4;
      const int ci = r / 3, kh = r - ci * 3, ih = ih0 + kh;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)ih < (unsigned)H) v = *(const float4*)(img + (((size_t)b * 3 + ci) * H + ih) * W + x4 * 4);
      *(float4*)(rows + r * IW + 4 + x4 * 4) = v;
    }
    __syncthreads();
    // ---- wave w: pixels 32 w .. 32 w + 31
    bf16x8 bf[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float f[8];
      const float* rp = rows + roff[nt] + 2 * (wave * 32 + quad * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = kind[nt] == 0 ? rp[2 * e] : (kind[nt] == 1 ? 1.f : 0.f);
      uint4 pk;
      pk.x = pack_bf16x2(f[0], f[1]); pk.y = pack_bf16x2(f[2], f[3]); pk.z = pack_bf16x2(f[4], f[5]); pk.w = pack_bf16x2(f[6], f[7]);
      bf[nt] = *reinterpret_cast<bf16x8*>(&pk);
    }
    const char* abase = (const char*)(dyt + (wave * 32 + quad * 8 + (r16 >> 2)) * RS + 4 * (r16 & 3));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 a = icw_ld_tr8(abase + mt * 32, abase + mt * 32 + 4 * RS * 2);
      acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[0], acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf[1], acc[mt][1], 0, 0, 0);
    }
    __syncthreads();                                   // every wave is done with the tiles before the next unit's are written
  }
  // ---- the four waves' accumulators, added in wave order; acc[mt][nt][r] = dW[mt * 16 + quad * 4 + r][nt * 16 + r16]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = red + (mt * 16 + quad * 4 + r) * 32 + nt * 16 + r16;
            *q = w ? *q + acc[mt][nt][r] : acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  for (int i = tid; i < CP * 32; i += 256) part[(size_t)blockIdx.x * CP * 32 + i] = red[i];
}

// one thread = one input pixel's 8-channel chunk; sums the patch-matrix entries of every (output pixel, tap) that read it
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, int ld, bf16_t* __restrict__ dx, int B,
                                                     int H, int W, int C, int KH, int KW, int stride, int pad, int Ho, int Wo,
                                                     int accumulate) {
  const int nch = C >> 3;
  const size_t total = (size_t)B * H * W * nch;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cc = (int)(idx % nch);
    const size_t ipix = idx / nch;
    const int b = (int)(ipix / ((size_t)H * W));
    const int rem = (int)(ipix - (size_t)b * H * W);
    const int ih = rem / W, iw = rem - ih * W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (accumulate) {
      float f[8];
      unpack_bf16x8(*(const uint4*)(dx + ipix * C + cc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = f[e];
    }
    for (int kh = 0; kh < KH; ++kh) {
      const int th = ih + pad - kh;
      if (th < 0 || th % stride) continue;
      const int oh = th / stride;
      if (oh >= Ho) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int tw = iw + pad - kw;
        if (tw < 0 || tw % stride) continue;
        const int ow = tw / stride;
        if (ow >= Wo) continue;
        float f[8];
        unpack_bf16x8(*(const uint4*)(dcol + (((size_t)b * Ho + oh) * Wo + ow) * ld + (size_t)(kh * KW + kw) * C + cc * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    }
    uint4 o;
    o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
    o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
    *(uint4*)(dx + ipix * C + cc * 8) = o;
  }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ dy2,
                                                       const uint4* __restrict__ y, uint4* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float d[8], yy[8];
    unpack_bf16x8(dy[i], d);
    unpack_bf16x8(y[i], yy);
    if (dy2) {
      float d2[8];
      unpack_bf16x8(dy2[i], d2);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] += d2[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = yy[e] > 0.f ? d[e] : 0.f;
    uint4 o;
    o.x = pack_bf16x2(d[0], d[1]); o.y = pack_bf16x2(d[2], d[3]);
    o.z = pack_bf16x2(d[4], d[5]); o.w = pack_bf16x2(d[6], d[7]);
    out[i] = o;
  }
}

// dtop[b, ih, iw, c] (+)= dpool[b, ih/k, iw/k, c] * w[(ih%k)*k + iw%k][c]    (windows do not overlap: k == stride)
__global__ __launch_bounds__(256) void dwpool_bwd_kernel(const bf16_t* __restrict__ dpool, int ldp, const float* __restrict__ w,
                                                         bf16_t* __restrict__ dtop, int B, int H, int W, int C, int k,
                                                         int accumulate) {
  const int nch = C >> 3, g = H / k;
  const size_t total = (size_t)B * H * W * nch;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int cc = (int)(idx % nch);
    const size_t ipix = idx / nch;
    const int b = (int)(ipix / ((size_t)H * W));
    const int rem = (int)(ipix - (size_t)b * H * W);
    const int ih = rem / W, iw = rem - ih * W;
    const int gy = ih / k, gx = iw / k, ky = ih - gy * k, kx = iw - gx * k;
    float d[8], acc[8];
    unpack_bf16x8(*(const uint4*)(dpool + (((size_t)b * g + gy) * g + gx) * ldp + cc * 8), d);
    const float* wr = w + (size_t)(ky * k + kx) * C + cc * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = d[e] * wr[e];
    if (accumulate) {
      float f[8];
      unpack_bf16x8(*(const uint4*)(dtop + ipix * C + cc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
    uint4 o;
    o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
    o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
    *(uint4*)(dtop + ipix * C + cc * 8) = o;
  }
}

// part[s][tap][c] = sum over the output positions of slab s of dpool[pos][c] * top[window(pos) + tap][c].
// grid (k, S): a block owns one ROW ky of the k x k window.  For a fixed output position the k taps of that row are k
// neighbouring pixels = k*C contiguous bf16 of the NHWC map (768 for every adapter: 16 x 48 ... 1 x 768), so thread e of
// the block reads element e of that run -- full cache lines whatever the channel count -- and accumulates tap e / C,
// channel e % C; three elements per thread and two positions in flight keep six loads outstanding.  (The first version, a
// block per tap with one serial chain of dependent loads per channel and 48 of 256 lanes busy at C = 48, ran at
// 0.2-0.9 TB/s: 3.6 ms per step at batch 512.)
__global__ __launch_bounds__(256) void dwpool_wgrad_kernel(const bf16_t* __restrict__ dpool, int ldp,
                                                           const bf16_t* __restrict__ top, float* __restrict__ part, int B,
                                                           int H, int W, int C, int k, int S) {
  constexpr int EPT = 3;
  const int ky = blockIdx.x, s = blockIdx.y, g = H / k, g2 = g * g, E = k * C;
  const size_t npos = (size_t)B * g2;
  const size_t p0 = npos * s / S, p1 = npos * (s + 1) / S;
  for (int e0 = threadIdx.x; e0 < E; e0 += 256 * EPT) {
    int ce[EPT];
    bool on[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = e0 + 256 * i;
      on[i] = e < E;
      ce[i] = on[i] ? e % C : 0;
    }
    float acc[2][EPT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < EPT; ++i) acc[u][i] = 0.f;
    auto visit = [&](size_t p, float* a) {
      const int b = (int)(p / (size_t)g2);
      const int rem = (int)(p - (size_t)b * g2);
      const int gy = rem / g, gx = rem - gy * g;
      const bf16_t* trow = top + (((size_t)b * H + gy * k + ky) * W + gx * k) * C + e0;
      const bf16_t* drow = dpool + p * ldp;
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (on[i]) a[i] = fmaf(bf16_to_f32(drow[ce[i]]), bf16_to_f32(trow[256 * i]), a[i]);
    };
    size_t p = p0;
    for (; p + 1 < p1; p += 2) {
      visit(p, acc[0]);
      visit(p + 1, acc[1]);
    }
    if (p < p1) visit(p, acc[0]);
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (on[i]) part[((size_t)s * k * k + (size_t)ky * k) * C + e0 + 256 * i] = acc[0][i] + acc[1][i];
  }
}

// The same sums with 16-byte loads (C % 8 == 0, 16-byte aligned rows): a thread owns 8 consecutive elements of the k*C run
// (one tap, 8 channels), 256 / (k*C/8) positions are walked side by side, four positions per thread in flight (eight 16-byte
// loads), the position lanes are folded through LDS.  2-byte loads made a wave's request 128 bytes: request-rate-bound.
__global__ __launch_bounds__(256) void dwpool_wgrad_vec_kernel(const bf16_t* __restrict__ dpool, int ldp,
                                                               const bf16_t* __restrict__ top, float* __restrict__ part, int B,
                                                               int H, int W, int C, int k, int S) {
  __shared__ float red[256 * 8];
  const int ky = blockIdx.x, s = blockIdx.y, g = H / k, g2 = g * g, E = k * C, VL = E / 8;      // VL <= 256 (host)
  const int PL = 256 / VL;
  const int pl = threadIdx.x / VL, vl = threadIdx.x - pl * VL;
  const size_t npos = (size_t)B * g2;
  const size_t p0 = npos * s / S, p1 = npos * (s + 1) / S;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (pl < PL) {
    const int c0 = (vl * 8) % C;
    auto fetch = [&](size_t p, uint4& t, uint4& d) {
      const int b = (int)(p / (size_t)g2);
      const int rem = (int)(p - (size_t)b * g2);
      const int gy = rem / g, gx = rem - gy * g;
      t = *(const uint4*)(top + (((size_t)b * H + gy * k + ky) * W + gx * k) * C + vl * 8);
      d = *(const uint4*)(dpool + p * ldp + c0);
    };
    auto fma8 = [&](const uint4& t, const uint4& d) {
      float tf[8], df[8];
      unpack_bf16x8(t, tf);
      unpack_bf16x8(d, df);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(df[i], tf[i], acc[i]);
    };
    size_t p = p0 + pl;
    for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {
      uint4 t0, t1, t2, t3, d0, d1, d2, d3;
      fetch(p, t0, d0);
      fetch(p + PL, t1, d1);
      fetch(p + 2 * (size_t)PL, t2, d2);
      fetch(p + 3 * (size_t)PL, t3, d3);
      fma8(t0, d0);
      fma8(t1, d1);
      fma8(t2, d2);
      fma8(t3, d3);
    }
    for (; p < p1; p += PL) {
      uint4 t0, d0;
      fetch(p, t0, d0);
      fma8(t0, d0);
    }
  }
  if (PL > 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
    __syncthreads();
    if (pl == 0)
      for (int q = 1; q < PL; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += red[(q * VL + vl) * 8 + i];
  }
  if (pl == 0) {
    float* o = part + ((size_t)s * k * k + (size_t)ky * k) * C + vl * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = acc[i];
  }
}

// part[s][tap][c] = sum over samples of slab s and grid positions of dsum[b, 1+pos, c] * x[b, 1 + neighbour(pos, tap), c]
// (generic grid size: one serial chain per (tap, channel); the 7 x 7 and 14 x 14 grids of the two configs use the row form)
__global__ __launch_bounds__(256) void dw3x3_wgrad_kernel(const float* __restrict__ dsum, int lds, const float* __restrict__ x,
                                                          int ldx, float* __restrict__ part, int B, int L, int g, int C, int S) {
  const int tap = blockIdx.x, s = blockIdx.y, ky = tap / 3, kx = tap - ky * 3;
  const int b0 = (int)((long long)B * s / S), b1 = (int)((long long)B * (s + 1) / S);
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int b = b0; b < b1; ++b)
      for (int py = 0; py < g; ++py) {
        const int yy = py + ky - 1;
        if (yy < 0 || yy >= g) continue;
        for (int px = 0; px < g; ++px) {
          const int xx = px + kx - 1;
          if (xx < 0 || xx >= g) continue;
          acc += dsum[((size_t)b * L + 1 + py * g + px) * lds + c] * x[((size_t)b * L + 1 + yy * g + xx) * ldx + c];
        }
      }
    part[((size_t)s * 9 + tap) * C + c] = acc;
  }
}

// Row form for a G x G grid known at compile time: a thread owns one channel and ALL nine taps; per grid row it loads the
// gradient row and the three neighbouring input rows (G + 3G independent, coalesced loads) and does the 9 x G products from
// registers.  grid (ceil(C / 256), S).  The tap-per-block form above read both matrices nine times through chains of
// dependent loads: 0.85 ms per adapter at batch 512 (0.2 TB/s) against ~0.05 ms for the 157 MB it has to read.
template <int G>
__global__ __launch_bounds__(256) void dw3x3_wgrad_rows_kernel(const float* __restrict__ dsum, int lds,
                                                               const float* __restrict__ x, int ldx, float* __restrict__ part,
                                                               int B, int L, int C, int S) {
  const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (c >= C) return;
  const int b0 = (int)((long long)B * s / S), b1 = (int)((long long)B * (s + 1) / S);
  float acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* db = dsum + ((size_t)b * L + 1) * lds + c;
    const float* xb = x + ((size_t)b * L + 1) * ldx + c;
#pragma unroll
    for (int py = 0; py < G; ++py) {
      float d[G];
#pragma unroll
      for (int px = 0; px < G; ++px) d[px] = db[(size_t)(py * G + px) * lds];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = py + ky - 1;
        if (yy < 0 || yy >= G) continue;
        float xr[G];
#pragma unroll
        for (int xx = 0; xx < G; ++xx) xr[xx] = xb[(size_t)(yy * G + xx) * ldx];
#pragma unroll
        for (int px = 0; px < G; ++px)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = px + kx - 1;
            if (xx >= 0 && xx < G) acc[ky][kx] = fmaf(d[px], xr[xx], acc[ky][kx]);
          }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) part[((size_t)s * 9 + t) * C + c] = acc[t / 3][t % 3];
}

// ---- train-mode BatchNorm (per-GPU batch statistics; reference nn.BatchNorm2d in train(), M.py:1825-1861, 1920-1936) ----
// x is the RAW convolution output [M, C] (bf16 activation matrix, or the fp32 token-grid matrix of the adapters' depthwise
// 3x3).  Statistics and both backward reductions are produced as per-chunk partial rows [chunks][2][C] that the caller
// folds with msclip_colsum (fixed order: deterministic).
template <typename T>
__device__ __forceinline__ float ld_f(const T* p) {
  if constexpr (sizeof(T) == 2) return bf16_to_f32(*p);
  else return *p;
}
// part[chunk][0][c] = sum x, part[chunk][1][c] = sum x^2 over the chunk's rows.  block = 256 threads = 4 row groups x 64 columns
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int ld, float* __restrict__ part, int M, int C,
                                                       int rows_per_chunk) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f, q = 0.f;
  if (c < C)
#pragma unroll 4
    for (int m = m0 + w; m < m1; m += 4) {
      const float v = ld_f(x + (size_t)m * ld + c);
      s += v;
      q = fmaf(v, v, q);
    }
  red[0][w][threadIdx.x & 63] = s;
  red[1][w][threadIdx.x & 63] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    const int l = threadIdx.x;
    part[((size_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    part[((size_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
  }
}
// part[chunk][0][c] = sum dy, part[chunk][1][c] = sum dy * xhat,  xhat = (x - mean) * rstd
template <typename T, typename TD>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const TD* __restrict__ dy, int lddy, const T* __restrict__ x, int ld,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ part, int M, int C, int rows_per_chunk) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f, q = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
#pragma unroll 4
    for (int m = m0 + w; m < m1; m += 4) {
      const float d = ld_f(dy + (size_t)m * lddy + c);
      s += d;
      q = fmaf(d, (ld_f(x + (size_t)m * ld + c) - mu) * rs, q);
    }
  }
  red[0][w][threadIdx.x & 63] = s;
  red[1][w][threadIdx.x & 63] = q;
  __syncthreads();
  if (w == 0 && c < C) {
    const int l = threadIdx.x;
    part[((size_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
    part[((size_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
  }
}
// y = act(x * scale[c] + shift[c] [+ resid]) -> bf16 or fp32 (TO).  thread = one column, blockIdx.y = row chunk: the
// per-channel constants are loaded once, consecutive lanes read consecutive columns, no index division per element.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ld, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const bf16_t* __restrict__ resid,
                                                       int ldr, TO* __restrict__ y, int ldy, int M, int C, int relu,
                                                       int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = scale[c], sh = shift[c];
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    float v = fmaf(ld_f(x + (size_t)m * ld + c), sc, sh);
    if (resid) v += bf16_to_f32(resid[(size_t)m * ldr + c]);
    if (relu) v = fmaxf(v, 0.f);
    if constexpr (sizeof(TO) == 2) y[(size_t)m * ldy + c] = f32_to_bf16(v);
    else y[(size_t)m * ldy + c] = v;
  }
}
// The same for the shape the training step has -- fp32 raw map in, bf16 out, C / leading dimensions multiples of 4 --: a thread
// owns 4 consecutive columns (16-byte loads of x, 8-byte residual loads / output stores; the scalar kernel above moves 4 + 2 bytes
// per lane and instruction: 3.4-4.2 TB/s isolated).
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const float* __restrict__ x, int ld, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const bf16_t* __restrict__ resid,
                                                           int ldr, bf16_t* __restrict__ y, int ldy, int M, int C, int relu,
                                                           int rows_per_chunk) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    const float4 xv = *(const float4*)(x + (size_t)m * ld + c);
    float4 v = make_float4(fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w));
    if (resid) {
      const uint2 r = *(const uint2*)(resid + (size_t)m * ldr + c);
      v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
      v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *(uint2*)(y + (size_t)m * ldy + c) = o;
  }
}
// dx = gamma * rstd * (dy - dbeta / n - xhat * dgamma / n), n = the number of rows the statistics were taken over
template <typename T, typename TD>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const TD* __restrict__ dy, int lddy, const T* __restrict__ x, int ld,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                                        const float* __restrict__ dgamma, TD* __restrict__ dx, int lddx, int M,
                                                        int C, float inv_n, int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c], rs = rstd[c], g = gamma[c] * rs, kb = dbeta[c] * inv_n, kg = dgamma[c] * inv_n;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    const float xh = (ld_f(x + (size_t)m * ld + c) - mu) * rs;
    const float v = g * (ld_f(dy + (size_t)m * lddy + c) - kb - xh * kg);
    if constexpr (sizeof(TD) == 2) dx[(size_t)m * lddx + c] = f32_to_bf16(v);
    else dx[(size_t)m * lddx + c] = v;
  }
}


// ---- train-mode BatchNorm backward, the shape the training step has: bf16 upstream gradient, fp32 raw maps, C and the leading
// dimensions multiples of 4.  Three things the scalar kernels above do not do: (1) a thread owns 4 consecutive columns (16-byte
// loads of the raw map, 8-byte loads / stores of the bf16 gradients); (2) the ReLU in front of the BatchNorm(s) is applied on the
// fly -- d = bf16(dy [+ dy2]) * (y > 0), exactly what msclip_relu_bwd would have written -- so that pass and its map are gone;
// (3) two BatchNorms that receive the SAME upstream gradient (a residual block's main path and its shortcut) share one pass.
struct BnBwdSide {
  const void* x;         // [M][ld]: the raw convolution output (fp32), or xhat (bf16: mean 0 / rstd 1 / gamma := scale; two-pass forward)
  int ld;
  const float* mean;     // [C] (tiled with the row fold)
  const float* rstd;
  const float* gamma;    // dx pass
  const float* dbeta;
  const float* dgamma;
  bf16_t* dx;
  int lddx;
  float* part;           // reduce pass: [chunks][2][C]
};

__device__ __forceinline__ float4 bn_ld_bf16x4(const bf16_t* p) {
  const uint2 r = *(const uint2*)p;
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ float bn_round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <typename XT>
__device__ __forceinline__ float4 bn_ld_x4(const void* x, size_t off) {
  if constexpr (sizeof(XT) == 4) return *(const float4*)((const float*)x + off);
  else return bn_ld_bf16x4((const bf16_t*)x + off);
}

// the upstream gradient of one (row, column quad) as the unfused path's msclip_relu_bwd output would hold it
__device__ __forceinline__ float4 bn_upstream(const bf16_t* dy, const bf16_t* dy2, const bf16_t* y, size_t o_dy, size_t o_dy2,
                                              size_t o_y) {
  float4 d = bn_ld_bf16x4(dy + o_dy);
  if (dy2) {
    const float4 e = bn_ld_bf16x4(dy2 + o_dy2);
    d = make_float4(bn_round_bf16(d.x + e.x), bn_round_bf16(d.y + e.y), bn_round_bf16(d.z + e.z), bn_round_bf16(d.w + e.w));
  }
  if (y) {
    const float4 a = bn_ld_bf16x4(y + o_y);
    d.x = a.x > 0.f ? d.x : 0.f; d.y = a.y > 0.f ? d.y : 0.f; d.z = a.z > 0.f ? d.z : 0.f; d.w = a.w > 0.f ? d.w : 0.f;
  }
  return d;
}

// part[chunk][0][c] = sum d, part[chunk][1][c] = sum d * xhat per side.  block = 4 row groups x 64 column quads (the scalar
// kernel's row order: group w adds rows m0 + w, m0 + w + 4, ...; the groups are added in order)
template <int NB, typename XT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ dy2,
                                                                int lddy2, const bf16_t* __restrict__ y, int ldy, BnBwdSide a,
                                                                BnBwdSide b, int M, int C, int rows_per_chunk) {
  __shared__ float4 red[NB][2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
  float4 s[NB], q[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) s[k] = q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    float4 mu[NB], rs[NB];
    mu[0] = *(const float4*)(a.mean + c);
    rs[0] = *(const float4*)(a.rstd + c);
    if constexpr (NB == 2) {
      mu[1] = *(const float4*)(b.mean + c);
      rs[1] = *(const float4*)(b.rstd + c);
    }
#pragma unroll 4
    for (int m = m0 + w; m < m1; m += 4) {
      const float4 d = bn_upstream(dy, dy2, y, (size_t)m * lddy + c, (size_t)m * lddy2 + c, (size_t)m * ldy + c);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const float4 xv = bn_ld_x4<XT>(k ? b.x : a.x, (size_t)m * (k ? b.ld : a.ld) + c);
        s[k].x += d.x; s[k].y += d.y; s[k].z += d.z; s[k].w += d.w;
        q[k].x = fmaf(d.x, (xv.x - mu[k].x) * rs[k].x, q[k].x);
        q[k].y = fmaf(d.y, (xv.y - mu[k].y) * rs[k].y, q[k].y);
        q[k].z = fmaf(d.z, (xv.z - mu[k].z) * rs[k].z, q[k].z);
        q[k].w = fmaf(d.w, (xv.w - mu[k].w) * rs[k].w, q[k].w);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    red[k][0][w][lane] = s[k];
    red[k][1][w][lane] = q[k];
  }
  __syncthreads();
  if (w < 2 * NB && c < C) {                     // wave (side, which sum)
    const int k = w >> 1, h = w & 1;
    const float4 r0 = red[k][h][0][lane], r1 = red[k][h][1][lane], r2 = red[k][h][2][lane], r3 = red[k][h][3][lane];
    float* part = (k ? b.part : a.part) + ((size_t)blockIdx.y * 2 + h) * C + c;
    *(float4*)part = make_float4(r0.x + r1.x + r2.x + r3.x, r0.y + r1.y + r2.y + r3.y, r0.z + r1.z + r2.z + r3.z,
                                 r0.w + r1.w + r2.w + r3.w);
  }
}

// dx = gamma * rstd * (d - dbeta / n - xhat * dgamma / n) per side, bf16
template <int NB, typename XT>
__global__ __launch_bounds__(256) void bn_bwd_dx_vec_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ dy2,
                                                            int lddy2, const bf16_t* __restrict__ y, int ldy, BnBwdSide a, BnBwdSide b,
                                                            int M, int C, float inv_n, int rows_per_chunk) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 4;
  if (c >= C) return;
  float4 mu[NB], rs[NB], g[NB], kb[NB], kg[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const BnBwdSide& sd = k ? b : a;
    mu[k] = *(const float4*)(sd.mean + c);
    rs[k] = *(const float4*)(sd.rstd + c);
    const float4 ga = *(const float4*)(sd.gamma + c), db = *(const float4*)(sd.dbeta + c), dg = *(const float4*)(sd.dgamma + c);
    g[k] = make_float4(ga.x * rs[k].x, ga.y * rs[k].y, ga.z * rs[k].z, ga.w * rs[k].w);
    kb[k] = make_float4(db.x * inv_n, db.y * inv_n, db.z * inv_n, db.w * inv_n);
    kg[k] = make_float4(dg.x * inv_n, dg.y * inv_n, dg.z * inv_n, dg.w * inv_n);
  }
  const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
#pragma unroll 4
  for (int m = m0 + w; m < m1; m += 4) {
    const float4 d = bn_upstream(dy, dy2, y, (size_t)m * lddy + c, (size_t)m * lddy2 + c, (size_t)m * ldy + c);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const BnBwdSide& sd = k ? b : a;
      const float4 xv = bn_ld_x4<XT>(sd.x, (size_t)m * sd.ld + c);
      const float v0 = g[k].x * (d.x - kb[k].x - (xv.x - mu[k].x) * rs[k].x * kg[k].x);
      const float v1 = g[k].y * (d.y - kb[k].y - (xv.y - mu[k].y) * rs[k].y * kg[k].y);
      const float v2 = g[k].z * (d.z - kb[k].z - (xv.z - mu[k].z) * rs[k].z * kg[k].z);
      const float v3 = g[k].w * (d.w - kb[k].w - (xv.w - mu[k].w) * rs[k].w * kg[k].w);
      uint2 o;
      o.x = pack_bf16x2(v0, v1);
      o.y = pack_bf16x2(v2, v3);
      *(uint2*)(sd.dx + (size_t)m * sd.lddx + c) = o;
    }
  }
}

}  // namespace

extern "C" int msclip_im2col(const void* x, int x_kind, void* col, int B, int H, int W, int C, int KH, int KW, int stride,
                             int pad, int Ho, int Wo, int Kp, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_im2col, stream, x, x_kind, col, B, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp);
  if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0)
    return MSCLIP_EINVAL;
  if ((Kp % 8) || Kp < KH * KW * C || x_kind < 0 || x_kind > 2) return MSCLIP_EINVAL;
  const size_t total = (size_t)B * Ho * Wo * (Kp / 8);
  const int grid = grid_for(total, 256);
  hipStream_t st = (hipStream_t)stream;
  const size_t row_lds = (size_t)C * KH * W * sizeof(float);
  if (x_kind != 0 && row_lds <= 49152 && (long long)B * Ho < (1ll << 31)) {    // the input image: rows through LDS
    if (x_kind == 1)
      hipLaunchKernelGGL(im2col_image_rows_kernel<1>, dim3(B * Ho), dim3(256), row_lds, st, x, (bf16_t*)col, H, W, C, KH, KW, stride,
                         pad, Ho, Wo, Kp);
    else
      hipLaunchKernelGGL(im2col_image_rows_kernel<2>, dim3(B * Ho), dim3(256), row_lds, st, x, (bf16_t*)col, H, W, C, KH, KW, stride,
                         pad, Ho, Wo, Kp);
    return msclip_launch_status();
  }
#define IM2COL(KIND)                                                                                                   \
  hipLaunchKernelGGL(im2col_kernel<KIND>, dim3(grid), dim3(256), 0, st, x, (bf16_t*)col, B, H, W, C, KH, KW, stride, pad, \
                     Ho, Wo, Kp)
  if (x_kind == 1) IM2COL(1);
  else if (x_kind == 2) IM2COL(2);
  else if (C % 8 == 0) IM2COL(0);
  else IM2COL(3);
#undef IM2COL
  return msclip_launch_status();
}

extern "C" int msclip_image_conv_wgrad(const float* img, const void* dy, int lddy, float* part, int part_blocks, int B, int S,
                                       int co, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_image_conv_wgrad, stream, img, dy, lddy, part, part_blocks, B, S, co);
  const int Ho = (S + 2 - 3) / 2 + 1;
  if (!img || !dy || !part || B <= 0 || S <= 0 || (S & 3) || S > 256 || Ho > 128 || co <= 0 || co > 64 || (co & 7) || (lddy & 7) ||
      lddy < co || part_blocks < 1 || ((size_t)img & 15) || ((size_t)dy & 15))
    return MSCLIP_EINVAL;
  const int total = B * Ho;
  const int upb = (total + part_blocks - 1) / part_blocks;
  hipStream_t st = (hipStream_t)stream;
#define ICW(MT)                                                                                                              \
  hipLaunchKernelGGL(image_conv_wgrad_kernel<MT>, dim3(part_blocks), dim3(256), 0, st, img, (const bf16_t*)dy, lddy, part, B, S, \
                     Ho, co, upb)
  if (co <= 16) ICW(1);
  else if (co <= 32) ICW(2);
  else if (co <= 48) ICW(3);
  else ICW(4);
#undef ICW
  return msclip_launch_status();
}

extern "C" int msclip_col2im(const void* dcol, int ld, void* dx, int B, int H, int W, int C, int KH, int KW, int stride,
                             int pad, int Ho, int Wo, int accumulate, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_col2im, stream, dcol, ld, dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, accumulate);
  if (!dcol || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
    return MSCLIP_EINVAL;
  if ((ld % 8) || ld < KH * KW * C || Ho <= 0 || Wo <= 0) return MSCLIP_EINVAL;
  const size_t total = (size_t)B * H * W * (C / 8);
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, ld,
                     (bf16_t*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, accumulate);
  return msclip_launch_status();
}

extern "C" int msclip_relu_bwd(const void* dy, const void* dy2, const void* y, void* out, long long n, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_relu_bwd, stream, dy, dy2, y, out, n);
  if (!dy || !y || !out || n <= 0 || (n % 8)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((size_t)n / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)dy, (const uint4*)dy2, (const uint4*)y, (uint4*)out, (size_t)n / 8);
  return msclip_launch_status();
}

extern "C" int msclip_dwpool_bwd(const void* dpool, int ldp, const float* w, void* dtop, int B, int H, int W, int C, int k,
                                 int accumulate, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_dwpool_bwd, stream, dpool, ldp, w, dtop, B, H, W, C, k, accumulate);
  if (!dpool || !w || !dtop || B <= 0 || k <= 0 || H <= 0 || H != W || (H % k) || (C % 8) || (ldp % 8) || ldp < C)
    return MSCLIP_EINVAL;
  const size_t total = (size_t)B * H * W * (C / 8);
  hipLaunchKernelGGL(dwpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpool,
                     ldp, w, (bf16_t*)dtop, B, H, W, C, k, accumulate);
  return msclip_launch_status();
}

extern "C" int msclip_dwpool_wgrad(const void* dpool, int ldp, const void* top, float* part, int B, int H, int W, int C,
                                   int k, int slabs, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_dwpool_wgrad, stream, dpool, ldp, top, part, B, H, W, C, k, slabs);
  if (!dpool || !top || !part || B <= 0 || k <= 0 || H <= 0 || H != W || (H % k) || C <= 0 || ldp < C || slabs <= 0 ||
      slabs > 65535)
    return MSCLIP_EINVAL;
  if (!(C % 8) && !(ldp % 8) && k * C <= 2048 && !((size_t)dpool % 16) && !((size_t)top % 16))
    hipLaunchKernelGGL(dwpool_wgrad_vec_kernel, dim3(k, slabs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpool, ldp,
                       (const bf16_t*)top, part, B, H, W, C, k, slabs);
  else
  hipLaunchKernelGGL(dwpool_wgrad_kernel, dim3(k, slabs), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dpool, ldp,
                     (const bf16_t*)top, part, B, H, W, C, k, slabs);
  return msclip_launch_status();
}

extern "C" int msclip_dw3x3_wgrad(const float* dsum, int lds, const float* x, int ldx, float* part, int B, int L, int g, int C,
                                  int slabs, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_dw3x3_wgrad, stream, dsum, lds, x, ldx, part, B, L, g, C, slabs);
  if (!dsum || !x || !part || B <= 0 || L != g * g + 1 || C <= 0 || lds < C || ldx < C || slabs <= 0 || slabs > 65535)
    return MSCLIP_EINVAL;
  const dim3 rows_grid((C + 255) / 256, slabs);
  if (g == 7)
    hipLaunchKernelGGL(dw3x3_wgrad_rows_kernel<7>, rows_grid, dim3(256), 0, (hipStream_t)stream, dsum, lds, x, ldx, part, B, L, C, slabs);
  else if (g == 14)
    hipLaunchKernelGGL(dw3x3_wgrad_rows_kernel<14>, rows_grid, dim3(256), 0, (hipStream_t)stream, dsum, lds, x, ldx, part, B, L, C, slabs);
  else
    hipLaunchKernelGGL(dw3x3_wgrad_kernel, dim3(9, slabs), dim3(256), 0, (hipStream_t)stream, dsum, lds, x, ldx, part, B, L, g, C,
                       slabs);
  return msclip_launch_status();
}

// ---- train-mode BatchNorm entry points.  x_f32 / dy_f32: 0 = bf16, 1 = fp32 matrices.
extern "C" int msclip_bn_stats(const void* x, int ld, int x_f32, float* part, int M, int C, int chunks, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_stats, stream, x, ld, x_f32, part, M, C, chunks);
  if (!x || !part || M <= 0 || C <= 0 || ld < C || chunks < 1 || chunks > 65535) return MSCLIP_EINVAL;
  const int rpc = (M + chunks - 1) / chunks;
  const dim3 grid((C + 63) / 64, chunks);
  if (x_f32) hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, part, M, C, rpc);
  else hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, part, M, C, rpc);
  return msclip_launch_status();
}

extern "C" int msclip_bn_apply(const void* x, int ld, int x_f32, const float* scale, const float* shift, const void* resid,
                               int ldr, void* y, int ldy, int y_f32, int M, int C, int relu, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_apply, stream, x, ld, x_f32, scale, shift, resid, ldr, y, ldy, y_f32, M, C, relu);
  if (!x || !scale || !shift || !y || M <= 0 || C <= 0 || ld < C || ldy < C || (resid && ldr < C)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* r = (const bf16_t*)resid;
  static int scalar_only = -1;
  if (scalar_only < 0) {
    const char* e = getenv("MSCLIP_BN_APPLY_SCALAR");
    scalar_only = e && e[0] == '1';
  }
  if (x_f32 && !y_f32 && !scalar_only && !(C & 3) && !(ld & 3) && !(ldy & 3) && !(resid && (ldr & 3)) &&
      !(((size_t)x | (size_t)scale | (size_t)shift) & 15) && !(((size_t)y | (size_t)resid) & 7)) {
    const int cbv = (C / 4 + 255) / 256;               // 4 columns per thread
    int ch = (M + 31) / 32;                            // >= 32 rows per block, at most ~8192 blocks
    if (ch * cbv > 8192) ch = 8192 / cbv > 0 ? 8192 / cbv : 1;
    const int rp = (M + ch - 1) / ch;
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(cbv, (M + rp - 1) / rp), dim3(256), 0, st, (const float*)x, ld, scale, shift, r, ldr,
                       (bf16_t*)y, ldy, M, C, relu, rp);
    return msclip_launch_status();
  }
  int chunks = (M + 63) / 64;                          // >= 64 rows per block, at most ~4096 blocks
  const int cb = (C + 255) / 256;
  if (chunks * cb > 4096) chunks = 4096 / cb > 0 ? 4096 / cb : 1;
  const int rpc = (M + chunks - 1) / chunks;
  const dim3 grid(cb, (M + rpc - 1) / rpc);
#define BN_APPLY(T, TO) \
  hipLaunchKernelGGL((bn_apply_kernel<T, TO>), grid, dim3(256), 0, st, (const T*)x, ld, scale, shift, r, ldr, (TO*)y, ldy, M, C, relu, rpc)
  if (x_f32 && y_f32) BN_APPLY(float, float);
  else if (x_f32) BN_APPLY(float, bf16_t);
  else if (y_f32) BN_APPLY(bf16_t, float);
  else BN_APPLY(bf16_t, bf16_t);
#undef BN_APPLY
  return msclip_launch_status();
}

extern "C" int msclip_bn_bwd_reduce(const void* dy, int lddy, int dy_f32, const void* x, int ld, int x_f32, const float* mean,
                                    const float* rstd, float* part, int M, int C, int chunks, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_bwd_reduce, stream, dy, lddy, dy_f32, x, ld, x_f32, mean, rstd, part, M, C, chunks);
  if (!dy || !x || !mean || !rstd || !part || M <= 0 || C <= 0 || ld < C || lddy < C || chunks < 1 || chunks > 65535)
    return MSCLIP_EINVAL;
  const int rpc = (M + chunks - 1) / chunks;
  const dim3 grid((C + 63) / 64, chunks);
  hipStream_t st = (hipStream_t)stream;
#define BN_RED(T, TD) \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, TD>), grid, dim3(256), 0, st, (const TD*)dy, lddy, (const T*)x, ld, mean, rstd, part, M, C, rpc)
  if (x_f32 && dy_f32) BN_RED(float, float);
  else if (x_f32) BN_RED(float, bf16_t);
  else if (dy_f32) BN_RED(bf16_t, float);
  else BN_RED(bf16_t, bf16_t);
#undef BN_RED
  return msclip_launch_status();
}

extern "C" int msclip_bn_bwd_dx(const void* dy, int lddy, int dy_f32, const void* x, int ld, int x_f32, const float* mean,
                                const float* rstd, const float* gamma, const float* dbeta, const float* dgamma, void* dx,
                                int lddx, int M, int C, long long n_stat, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_bwd_dx, stream, dy, lddy, dy_f32, x, ld, x_f32, mean, rstd, gamma, dbeta, dgamma, dx, lddx, M, C, n_stat);
  if (!dy || !x || !mean || !rstd || !gamma || !dbeta || !dgamma || !dx || M <= 0 || C <= 0 || ld < C || lddy < C || lddx < C ||
      n_stat <= 0)
    return MSCLIP_EINVAL;
  int chunks = (M + 63) / 64;
  const int cb = (C + 255) / 256;
  if (chunks * cb > 4096) chunks = 4096 / cb > 0 ? 4096 / cb : 1;
  const int rpc = (M + chunks - 1) / chunks;
  const dim3 grid(cb, (M + rpc - 1) / rpc);
  const float inv_n = 1.f / (float)n_stat;
  hipStream_t st = (hipStream_t)stream;
#define BN_DX(T, TD)                                                                                                      \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<T, TD>), grid, dim3(256), 0, st, (const TD*)dy, lddy, (const T*)x, ld, mean, rstd, gamma, \
                     dbeta, dgamma, (TD*)dx, lddx, M, C, inv_n, rpc)
  if (x_f32 && dy_f32) BN_DX(float, float);
  else if (x_f32) BN_DX(float, bf16_t);
  else if (dy_f32) BN_DX(bf16_t, float);
  else BN_DX(bf16_t, bf16_t);
#undef BN_DX
  return msclip_launch_status();
}

// Fused form of (msclip_relu_bwd ->) msclip_bn_bwd_reduce / msclip_bn_bwd_dx for one or two BatchNorms behind the same upstream
// gradient (include/msclip_hip.h).  pass 0 = the reduce (fills part1 / part2 [chunks][2][C]), pass 1 = the dx pass.
extern "C" int msclip_bn_bwd_fused(int pass, const void* dy, int lddy, const void* dy2, int lddy2, const void* y, int ldy,
                                   const msclip_bn_bwd_side* s1, const msclip_bn_bwd_side* s2, int M, int C, int chunks,
                                   long long n_stat, void* stream) {
  MSCLIP_PLAN_UNSUPPORTED(msclip_bn_bwd_fused)      // (side descriptors are host structs; the training step is not a plan)
  if ((pass != 0 && pass != 1) || !dy || !s1 || M <= 0 || C <= 0 || (C & 3) || (lddy & 3) || lddy < C || ((size_t)dy & 7) ||
      (dy2 && ((lddy2 & 3) || lddy2 < C || ((size_t)dy2 & 7))) || (y && ((ldy & 3) || ldy < C || ((size_t)y & 7))))
    return MSCLIP_EINVAL;
  BnBwdSide sd[2] = {};
  for (int k = 0; k < 2; ++k) {
    const msclip_bn_bwd_side* s = k ? s2 : s1;
    if (!s) continue;
    if (!s->x || !s->mean || !s->rstd || (s->ld & 3) || s->ld < C || ((size_t)s->x & (s->x_bf16 ? 7 : 15)) || ((size_t)s->mean & 15) ||
        ((size_t)s->rstd & 15) || (s->x_bf16 != 0) != (s1->x_bf16 != 0))
      return MSCLIP_EINVAL;
    if (pass == 0 && (!s->part || ((size_t)s->part & 15))) return MSCLIP_EINVAL;
    if (pass == 1 && (!s->gamma || !s->dbeta || !s->dgamma || !s->dx || (s->lddx & 3) || s->lddx < C || ((size_t)s->dx & 7) ||
                      ((size_t)s->gamma & 15) || ((size_t)s->dbeta & 15) || ((size_t)s->dgamma & 15)))
      return MSCLIP_EINVAL;
    sd[k] = BnBwdSide{s->x, s->ld, s->mean, s->rstd, s->gamma, s->dbeta, s->dgamma, (bf16_t*)s->dx, s->lddx, s->part};
  }
  hipStream_t st = (hipStream_t)stream;
  const int cb = (C / 4 + 63) / 64;
  if (pass == 0) {
    if (chunks < 1 || chunks > 65535) return MSCLIP_EINVAL;
    const int rpc = (M + chunks - 1) / chunks;
    const dim3 grid(cb, chunks);
#define BN_RED_VEC(NB, XT, B2)                                                                                               \
  hipLaunchKernelGGL((bn_bwd_reduce_vec_kernel<NB, XT>), grid, dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)dy2, lddy2, \
                     (const bf16_t*)y, ldy, sd[0], sd[B2], M, C, rpc)
    if (s2 && s1->x_bf16) BN_RED_VEC(2, bf16_t, 1);
    else if (s2) BN_RED_VEC(2, float, 1);
    else if (s1->x_bf16) BN_RED_VEC(1, bf16_t, 0);
    else BN_RED_VEC(1, float, 0);
#undef BN_RED_VEC
    return msclip_launch_status();
  }
  if (n_stat <= 0) return MSCLIP_EINVAL;
  int nch = chunks > 0 ? chunks : (M + 31) / 32;
  if ((long long)nch * cb > 16384) nch = 16384 / cb > 0 ? 16384 / cb : 1;
  const int rpc = (M + nch - 1) / nch;
  const dim3 grid(cb, (M + rpc - 1) / rpc);
  const float inv_n = 1.f / (float)n_stat;
#define BN_DX_VEC(NB, XT, B2)                                                                                                \
  hipLaunchKernelGGL((bn_bwd_dx_vec_kernel<NB, XT>), grid, dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)dy2, lddy2, \
                     (const bf16_t*)y, ldy, sd[0], sd[B2], M, C, inv_n, rpc)
  if (s2 && s1->x_bf16) BN_DX_VEC(2, bf16_t, 1);
  else if (s2) BN_DX_VEC(2, float, 1);
  else if (s1->x_bf16) BN_DX_VEC(1, bf16_t, 0);
  else BN_DX_VEC(1, float, 0);
#undef BN_DX_VEC
  return msclip_launch_status();
}

namespace {
// Chain rule of a folded (frozen-statistics) BatchNorm back to the module's own parameters, one block per output channel c:
//   s = gamma rstd, rstd = 1 / sqrt(var + eps);  W_f = W s, shift = beta - mean s
//   dW[c][k] = G[c][k] s[c];  dgamma[c] = (sum_k G[c][k] W[c][k] - mean[c] dshift[c]) rstd[c];  dbeta[c] = dshift[c]
// (G = dL/dW_f, dshift = dL/dshift).  Replaces a dozen per-channel ATen launches per BatchNorm of the frozen-statistics backward.
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float* __restrict__ G, long long ldg, const float* __restrict__ w,
                                                          int K, const float* __restrict__ dshift, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                          float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float rstd = 1.f / sqrtf(var[c] + eps);
  const float sc = gamma[c] * rstd;
  const float* g = G + (size_t)c * ldg;
  const float* wr = w + (size_t)c * K;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float gv = g[k];
    acc += gv * wr[k];
    dW[(size_t)c * K + k] = gv * sc;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ds = (red[0] + red[1]) + (red[2] + red[3]) - mean[c] * dshift[c];
    dgamma[c] = ds * rstd;
    dbeta[c] = dshift[c];
  }
}
}  // namespace

extern "C" int msclip_bn_fold_bwd(const float* G, long long ldg, const float* w_raw, int cout, int K, const float* dshift,
                                  const float* gamma, const float* mean, const float* var, float eps, float* dW, float* dgamma,
                                  float* dbeta, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_fold_bwd, stream, G, ldg, w_raw, cout, K, dshift, gamma, mean, var, eps, dW, dgamma, dbeta);
  if (!G || !w_raw || !dshift || !gamma || !mean || !var || !dW || !dgamma || !dbeta || cout <= 0 || K <= 0 || ldg < K) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream, G, ldg, w_raw, K, dshift, gamma, mean, var,
                     eps, dW, dgamma, dbeta);
  return msclip_launch_status();
}

namespace {
// Train-mode BatchNorm, the per-channel tail of the statistics pass in one launch: sums[0][j][c] = sum x, sums[1][j][c] = sum x^2
// over the rows of fold j (r folds: narrow maps are reduced as r rows per wide row, hip.py::_bn_fold_rows)
// -> out[0] = mean, out[1] = biased variance, out[2] = rstd, out[3] = scale = gamma rstd, out[4] = shift = beta - mean scale.
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ sums, int r, int C, float inv_n,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ out, int rep) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int j = 0; j < r; ++j) {
    s1 += sums[(size_t)j * C + c];
    s2 += sums[(size_t)(r + j) * C + c];
  }
  const float mean = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  // every vector is written `rep` times back to back (row stride rep * C): the passes over a narrow map read r rows as one wide
  // row and take their per-channel vectors tiled r times -- from here instead of five ATen repeat launches per BatchNorm
  for (int j = 0; j < rep; ++j) {
    const size_t o = (size_t)j * C + c, ld = (size_t)rep * C;
    out[o] = mean;
    out[ld + o] = var;
    out[2 * ld + o] = rstd;
    out[3 * ld + o] = sc;
    out[4 * ld + o] = beta[c] - mean * sc;
  }
}

// Tail of the train-mode BatchNorm backward: part [chunks][2][r * C] (msclip_bn_bwd_reduce over r-folded rows) ->
// out[0] = dbeta, out[1] = dgamma, out[2] = gamma, each [C] tiled `r` times (row stride r * C): what msclip_bn_bwd_dx reads.
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ part, int chunks, int r, int C,
                                                            const float* __restrict__ gamma, float* __restrict__ out) {
  // one workgroup per channel: up to 512 chunks x 16 folds of partials per channel -- 256 threads stride over them, then a
  // fixed-order tree in LDS (deterministic)
  __shared__ float red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const size_t ld = (size_t)r * C;
  float sb = 0.f, sg = 0.f;
  for (int i = tid; i < chunks * r; i += 256) {
    const int k = i / r, j = i - k * r;
    sb += part[(size_t)k * 2 * ld + (size_t)j * C + c];
    sg += part[(size_t)k * 2 * ld + ld + (size_t)j * C + c];
  }
  red[0][tid] = sb;
  red[1][tid] = sg;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      red[0][tid] += red[0][tid + w];
      red[1][tid] += red[1][tid + w];
    }
    __syncthreads();
  }
  sb = red[0][0];
  sg = red[1][0];
  const float g = gamma[c];
  for (int j = tid; j < r; j += 256) {
    const size_t o = (size_t)j * C + c;
    out[o] = sb;
    out[ld + o] = sg;
    out[2 * ld + o] = g;
  }
}
}  // namespace

extern "C" int msclip_bn_finish(const float* sums, int r, int C, long long n, const float* gamma, const float* beta, float eps,
                                float* out, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_finish, stream, sums, r, C, n, gamma, beta, eps, out);
  if (!sums || !gamma || !beta || !out || r <= 0 || C <= 0 || n <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, r, C, 1.f / (float)n, gamma,
                     beta, eps, out, 1);
  return msclip_launch_status();
}

extern "C" int msclip_bn_finish_tiled(const float* sums, int r, int C, long long n, const float* gamma, const float* beta, float eps,
                                      float* out, int rep, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_finish_tiled, stream, sums, r, C, n, gamma, beta, eps, out, rep);
  if (!sums || !gamma || !beta || !out || r <= 0 || C <= 0 || n <= 0 || rep <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, r, C, 1.f / (float)n, gamma,
                     beta, eps, out, rep);
  return msclip_launch_status();
}

extern "C" int msclip_bn_bwd_finish(const float* part, int chunks, int r, int C, const float* gamma, float* out, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_bn_bwd_finish, stream, part, chunks, r, C, gamma, out);
  if (!part || !gamma || !out || chunks <= 0 || r <= 0 || C <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, chunks, r, C, gamma, out);
  return msclip_launch_status();
}
