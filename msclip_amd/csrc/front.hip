// Fused front of the two convolutional branches (HBM-bound, 112x112 maps with 48 channels):
//
//   stem:      image --3x3/s2 (3->48, BN, ReLU)--> S1 --3x3/s2 (48->96, BN(+1x1 shortcut), ReLU)--> 96@56x56
//              (reference lib/models/clip_openai_pe_res_v1.py:1993-1995 then resnet_stage.conv_0, 1920-1936)
//   parallel:  P0 --1x1 (48->48, BN, ReLU)--> t1 --3x3/s2 (48->48, BN, ReLU)--> 48@56x56
//              (ConvResBlock conv1/bn1/relu -> conv2/bn2/relu, ibid. 1825-1840)
//
// Unfused, S1 / t1 (616 MB each at batch 512) are written and then gathered back through 64-byte segments at
// 2.5-2.9 TB/s.  Here a workgroup owns an 8 x 8 tile of the 56 x 56 output: the PRODUCER computes the 17 x 17
// window of the intermediate map that tile needs (halo included, 13 % recomputation of a cheap convolution) with
// MFMAs straight into LDS, the CONSUMER runs the 3x3/s2 convolution out of LDS against weights that stay resident
// in LDS.  The intermediate map never exists in HBM.  The stem producer also emits the parallel branch's stage-0
// map (same image taps, the other 48 filters) for the tile's 16 x 16 interior, staged so it leaves as full lines.
//
// MFMA operand convention (v_mfma_f32_16x16x32_bf16, weights as A, pixels as B): lane (p = l % 16, q = l / 16) holds
// 8 K-values q*8.. of channel p (A) / of pixel p (B); the result holds channels 4q..4q+3 of pixel p.
#include <stdlib.h>
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int CM = 48;                    // channels of the intermediate map
constexpr int TS = 8;                     // output tile edge
constexpr int MS = 2 * TS + 1;            // intermediate window edge (17)
constexpr int NMID = MS * MS;             // 289 intermediate pixels
constexpr int NMT = (NMID + 15) / 16;     // 19 producer pixel-tiles
constexpr int PMT = (NMT + 3) / 4;        // pixel-tiles per wave (5)
constexpr int PS = CM * 2 + 16;           // LDS bytes per intermediate pixel (112)
constexpr int MIDB = ((NMID * PS + 127) / 128) * 128 + 128;
constexpr int KP = 448;                   // 9 * 48 = 432 padded to the packed weight's row length
constexpr int KS = KP / 32;               // 14 K-steps
constexpr int WS = KP * 2 + 16;           // weight row stride in LDS (912: conflict-free 16-byte reads over 16 rows)
constexpr int PSTB = 4 * TS * TS * CM * 2;  // stage-0 map of the tile interior (16 x 16 pixels)
constexpr int PATCHB = 3 * (4 * TS + 3) * (4 * TS + 4) * 2 + 8;   // bf16 image patch [3][35][36]

struct FrontArgs {
  const void* x;        // PROD 0: NCHW image (fp32 / bf16); PROD 1: NHWC bf16 [B, Hm, Wm, 48]
  const void* w1;       // PROD 0: fp32 [27][96]; PROD 1: bf16 [48][64]
  const float* b1;      // PROD 0: [96]; PROD 1: [48]
  const bf16_t* w2;     // bf16 [Cout][448], K = (kh*3 + kw)*48 + c
  const float* b2;      // [Cout]
  bf16_t* side;         // PROD 0: stage-0 map NHWC bf16 [B, Hm, Wm, 48]
  bf16_t* out;          // NHWC bf16 [B, Ho, Wo, Cout]
  unsigned xbytes;      // extent of x (< 2^31: offsets beyond it mark padding)
  int B, Hm, Wm, Ho, Wo, Himg, Wimg, ntile, tyn, txn;
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int PROD, int NTT, typename InT>
__global__ __launch_bounds__(256) void front_kernel(FrontArgs a) {
  extern __shared__ __attribute__((aligned(128))) char lds[];
  char* const wl = lds;
  char* const mid = lds + NTT * 16 * WS;
  char* const pst = mid + MIDB;
  char* const patch = pst + PSTB;                       // PROD 0: bf16 [3][35][36] image patch
  constexpr int COUT = NTT * 16;
  constexpr int MT = NTT / 3;                           // consumer pixel-tiles per wave (4 pixel-tiles per 8x8 tile)
  constexpr int NP = PROD == 0 ? 6 : 3;                 // producer channel tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;

  // ---- consumer weights -> LDS, once per (persistent) workgroup
  for (int i = tid; i < COUT * (KP / 8); i += 256) {
    const int r = i / (KP / 8), c = i - r * (KP / 8);
    *(u32x4*)(wl + r * WS + c * 16) = *(const u32x4*)(a.w2 + (size_t)r * KP + c * 8);
  }
  for (int i = tid; i < MIDB / 16; i += 256) *(u32x4*)(mid + i * 16) = u32x4{0, 0, 0, 0};

  // ---- producer weights / biases in registers
  bf16x8 wf[NP][PROD == 0 ? 1 : 2];
  float pb[NP][4];
  if constexpr (PROD == 0) {
    const float* w = (const float*)a.w1;
#pragma unroll
    for (int nt = 0; nt < NP; ++nt)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = q * 8 + e;
        wf[nt][0][e] = (__bf16)(k < 27 ? w[k * 96 + nt * 16 + p] : 0.f);
      }
  } else {
    const bf16_t* w = (const bf16_t*)a.w1;
#pragma unroll
    for (int nt = 0; nt < NP; ++nt)
#pragma unroll
      for (int s = 0; s < 2; ++s) wf[nt][s] = as_bf16x8(*(const u32x4*)(w + (nt * 16 + p) * 64 + (4 * s + q) * 8));
  }
#pragma unroll
  for (int nt = 0; nt < NP; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) pb[nt][r] = a.b1[nt * 16 + 4 * q + r];

  // ---- consumer geometry
  const int cm0 = NTT == 6 ? (wave & 1) * 2 : wave;     // first pixel-tile of this wave
  const int cn0 = NTT == 6 ? (wave >> 1) * 3 : 0;       // first channel-tile
  float cb[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cb[j][r] = a.b2[(cn0 + j) * 16 + 4 * q + r];
  int tapoff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 4 * s + q;
    const int tap = c / 6, c8 = c - tap * 6;
    const int kh = tap / 3, kw = tap - kh * 3;
    tapoff[s] = c < 54 ? (kh * MS + kw) * PS + c8 * 16 : 0;   // chunks 54, 55 meet zero weights: any finite data
  }
  int pixbase[MT], oyl[MT], oxl[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int op = (cm0 + m) * 16 + p;
    oyl[m] = op >> 3;
    oxl[m] = op & 7;
    pixbase[m] = (2 * oyl[m] * MS + 2 * oxl[m]) * PS;
  }
  const int wrow = p * WS + q * 16;

  // ---- producer inputs, requested one tile ahead.  Buffer-addressed loads: a pixel outside the image gets an offset
  // beyond the descriptor's range and reads as zero -- no branch and no select on the loaded value, so nothing waits
  // on a load before the NEXT iteration uses it.  The stem producer's source is the 35 x 35 x 3 image patch under the
  // window: it is fetched as coalesced row pieces (15 elements per thread), parked in LDS as bf16 and the MFMA
  // operands (8 of the 27 taps per lane) are picked out of LDS.
  constexpr unsigned OOB = 0x80000000u;
  constexpr int PW = 4 * TS + 3, PWS = PW + 1;          // patch edge (35), padded row (36 elements)
  constexpr int NPATCH = 3 * PW * PW, NPL = (NPATCH + 255) / 256;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.xbytes);
  u32x4 pre1[PMT][2];
  unsigned praw[NPL];
  int prow[NPL], pcol[NPL], poff[NPL], plds[NPL];       // patch row / column / image offset / LDS byte offset of piece i
  int tapc[8];
  if constexpr (PROD == 0) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      int j = i * 256 + tid;
      j = j < NPATCH ? j : NPATCH - 1;
      const int ci = j / (PW * PW), r = (j - ci * PW * PW) / PW, c = j - ci * PW * PW - r * PW;
      prow[i] = r;
      pcol[i] = c;
      poff[i] = (ci * a.Himg + r) * a.Wimg + c;
      plds[i] = ((ci * PW + r) * PWS + c) * 2;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = q * 8 + e;
      const int ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
      tapc[e] = k < 27 ? ((ci * PW + kh) * PWS + kw) * 2 : 0;   // taps 27..31 meet zero weights
    }
  }
  auto coords = [&](int t, int& b, int& oy0, int& ox0) {
    b = t / (a.tyn * a.txn);
    const int r = t - b * (a.tyn * a.txn);
    const int ty = r / a.txn;
    oy0 = ty * TS;
    ox0 = (r - ty * a.txn) * TS;
  };
  auto request = [&](int t) {
    int b, oy0, ox0;
    coords(t, b, oy0, ox0);
    if constexpr (PROD == 1) {
#pragma unroll
      for (int i = 0; i < PMT; ++i) {
        const int mt = wave + 4 * i;
        if (mt < NMT) {
          int mp = mt * 16 + p;
          mp = mp < NMID ? mp : NMID - 1;
          const int my = mp / MS, mx = mp - my * MS;
          const int gy = 2 * oy0 - 1 + my, gx = 2 * ox0 - 1 + mx;
          const bool in = (unsigned)gy < (unsigned)a.Hm && (unsigned)gx < (unsigned)a.Wm;
          const unsigned src = (unsigned)(((b * a.Hm + gy) * a.Wm + gx) * (CM * 2));
          unsigned o0 = in ? src + q * 16 : OOB, o1 = (in && q < 2) ? src + 64 + q * 16 : OOB;
          asm volatile("" : "+v"(o0), "+v"(o1));                 // opaque: keeps the loads unconditional
          pre1[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, o0, 0, 0);
          pre1[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, o1, 0, 0);
        }
      }
    } else {
      const int py0 = 4 * oy0 - 3, px0 = 4 * ox0 - 3;
      const int base = (b * 3 * a.Himg + py0) * a.Wimg + px0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const bool ok = (unsigned)(py0 + prow[i]) < (unsigned)a.Himg && (unsigned)(px0 + pcol[i]) < (unsigned)a.Wimg;
        unsigned off = ok ? (unsigned)(base + poff[i]) * (unsigned)sizeof(InT) : OOB;
        asm volatile("" : "+v"(off));                            // opaque: keeps the load unconditional
        if constexpr (sizeof(InT) == 4) praw[i] = __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0);
        else praw[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rx, off, 0, 0);
      }
    }
  };

  // every load of the set-up is complete here: otherwise the compiler's wait for them lands inside the loop, where it
  // would also drain the prefetch of the next tile
  __builtin_amdgcn_s_waitcnt(0x0070);                   // vmcnt(0) lgkmcnt(0)
  __syncthreads();
  int t = blockIdx.x;
  if (t < a.ntile) request(t);

  for (; t < a.ntile; t += gridDim.x) {
    int b, oy0, ox0;
    coords(t, b, oy0, ox0);
    if constexpr (PROD == 0) {
      // image patch -> LDS (bf16)
#pragma unroll
      for (int i = 0; i < NPL; ++i)
        if (i * 256 + tid < NPATCH) {
          bf16_t h;
          if constexpr (sizeof(InT) == 4) h = f32_to_bf16(__uint_as_float(praw[i]));
          else h = (bf16_t)praw[i];
          *(bf16_t*)(patch + plds[i]) = h;
        }
      __syncthreads();
    }
    // ================= producer: intermediate window -> LDS
#pragma unroll
    for (int i = 0; i < PMT; ++i) {
      const int mt = wave + 4 * i;
      if (mt < NMT) {
        const int mp = mt * 16 + p;
        const int mpc = mp < NMID ? mp : NMID - 1;
        const int my = mpc / MS, mx = mpc - my * MS;
        const int gy = 2 * oy0 - 1 + my, gx = 2 * ox0 - 1 + mx;
        const bool in = (unsigned)gy < (unsigned)a.Hm && (unsigned)gx < (unsigned)a.Wm;
        bf16x8 xb[2];
        if constexpr (PROD == 1) {
          xb[0] = as_bf16x8(pre1[i][0]);
          xb[1] = as_bf16x8(pre1[i][1]);
        } else {
          const char* src = patch + (2 * my * PWS + 2 * mx) * 2;
          unsigned short h[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = *(const unsigned short*)(src + tapc[e]);
          u32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = (unsigned)h[2 * e] | ((unsigned)h[2 * e + 1] << 16);
          xb[0] = as_bf16x8(u);
        }
#pragma unroll
        for (int nt = 0; nt < NP; ++nt) {
          f32x4 acc = {pb[nt][0], pb[nt][1], pb[nt][2], pb[nt][3]};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][0], xb[0], acc, 0, 0, 0);
          if constexpr (PROD == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][1], xb[1], acc, 0, 0, 0);
          uint2 o;
          o.x = pack_bf16x2(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f));
          o.y = pack_bf16x2(fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
          if (nt < 3) {
            if (!in) o.x = o.y = 0u;                            // zero padding of the 3x3 that follows
            if (mp < NMID) *(uint2*)(mid + mp * PS + (nt * 16 + 4 * q) * 2) = o;
          } else {                                              // stage-0 map of the parallel branch: interior only
            if (mp < NMID && my >= 1 && mx >= 1)
              *(uint2*)(pst + ((my - 1) * 16 + (mx - 1)) * (CM * 2) + ((nt - 3) * 16 + 4 * q) * 2) = o;
          }
        }
      }
    }
    __syncthreads();
#ifndef FRONT_NOREQ
    if (t + (int)gridDim.x < a.ntile) request(t + gridDim.x);
#endif
#ifndef FRONT_NOSIDE
    if constexpr (PROD == 0) {
      // the 16 x 16 interior leaves as 16 runs of 1536 contiguous bytes
#pragma unroll
      for (int i = 0; i < PSTB / 16 / 256; ++i) {
        const int ch = i * 256 + tid;
        const int row = ch / 96, within = ch - row * 96;
        const int gy = 2 * oy0 + row, gx = 2 * ox0 + within / 6;
        if (gy < a.Hm && gx < a.Wm)
          *(u32x4*)((char*)(a.side + (((size_t)b * a.Hm + gy) * a.Wm + 2 * ox0) * CM) + within * 16) =
              *(const u32x4*)(pst + ch * 16);
      }
    }
#endif
    // ================= consumer: 3x3 / stride 2 out of LDS
    f32x4 acc[MT][3];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[m][j] = f32x4{cb[j][0], cb[j][1], cb[j][2], cb[j][3]};
#ifdef FRONT_NOCONS
#pragma unroll
    for (int s = 0; s < 1; ++s) {
#else
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#endif
      bf16x8 wa[3], xb[MT];
#pragma unroll
      for (int j = 0; j < 3; ++j) wa[j] = as_bf16x8(*(const u32x4*)(wl + (cn0 + j) * 16 * WS + wrow + s * 64));
#pragma unroll
      for (int m = 0; m < MT; ++m) xb[m] = as_bf16x8(*(const u32x4*)(mid + pixbase[m] + tapoff[s]));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j], xb[m], acc[m][j], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oy = oy0 + oyl[m], ox = ox0 + oxl[m];
#ifdef FRONT_NOOUT
      if (oy < a.Ho && ox < a.Wo && acc[m][0][0] == 12345.f) {
#else
      if (oy < a.Ho && ox < a.Wo) {
#endif
        bf16_t* dst = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * COUT + cn0 * 16 + 4 * q;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          uint2 o;
          o.x = pack_bf16x2(fmaxf(acc[m][j][0], 0.f), fmaxf(acc[m][j][1], 0.f));
          o.y = pack_bf16x2(fmaxf(acc[m][j][2], 0.f), fmaxf(acc[m][j][3], 0.f));
          *(uint2*)(dst + j * 16) = o;
        }
      }
    }
    __syncthreads();
  }
}

template <int PROD, int NTT, typename InT>
int launch_front(FrontArgs a, hipStream_t st) {
  const size_t lds = (size_t)NTT * 16 * WS + MIDB + (PROD == 0 ? PSTB + PATCHB : 0);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)front_kernel<PROD, NTT, InT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return MSCLIP_ELAUNCH;
    attr_done = true;
  }
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    ncu = 256;
  const int per_cu = (int)(160 * 1024 / lds) > 2 ? 2 : (int)(160 * 1024 / lds);
  int grid = ncu * (per_cu < 1 ? 1 : per_cu);
  if (grid > a.ntile) grid = a.ntile;
  hipLaunchKernelGGL((front_kernel<PROD, NTT, InT>), dim3(grid), dim3(256), lds, st, a);
  return msclip_launch_status();
}

int fill_geometry(FrontArgs& a, int B, int Hm, int Wm) {
  a.B = B;
  a.Hm = Hm;
  a.Wm = Wm;
  a.Ho = (Hm + 2 - 3) / 2 + 1;
  a.Wo = (Wm + 2 - 3) / 2 + 1;
  a.tyn = (a.Ho + TS - 1) / TS;
  a.txn = (a.Wo + TS - 1) / TS;
  const long long nt = (long long)B * a.tyn * a.txn;
  if (nt > 0x7fffffffLL) return MSCLIP_EINVAL;
  a.ntile = (int)nt;
  return MSCLIP_OK;
}

}  // namespace

extern "C" int msclip_conv1x1_conv3x3s2(const void* x, const void* w1, const float* b1, const void* w2,
                                        const float* b2, void* out, int B, int H, int W, int Cout, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !out || B <= 0 || H <= 0 || W <= 0 || (Cout != 48 && Cout != 96))
    return MSCLIP_EINVAL;
  FrontArgs a{};
  a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.side = nullptr; a.out = (bf16_t*)out;
  a.Himg = a.Wimg = 0;
  if (fill_geometry(a, B, H, W) != MSCLIP_OK) return MSCLIP_EINVAL;
  const unsigned long long xb = (unsigned long long)B * H * W * CM * 2;
  if (xb >= 0x80000000ull) return MSCLIP_EINVAL;
  a.xbytes = (unsigned)xb;
  return Cout == 48 ? launch_front<1, 3, bf16_t>(a, (hipStream_t)stream) : launch_front<1, 6, bf16_t>(a, (hipStream_t)stream);
}

extern "C" int msclip_stem_dual_conv3x3s2(const void* img, int img_is_bf16, const float* w, const float* bias,
                                          void* out_b, const void* w2, const float* b2, void* out2, int B, int H,
                                          int W, int Cout, void* stream) {
  if (!img || !w || !bias || !out_b || !w2 || !b2 || !out2 || B <= 0 || H <= 0 || W <= 0 || (Cout != 48 && Cout != 96))
    return MSCLIP_EINVAL;
  FrontArgs a{};
  a.x = img; a.w1 = w; a.b1 = bias; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.side = (bf16_t*)out_b; a.out = (bf16_t*)out2;
  a.Himg = H; a.Wimg = W;
  if (fill_geometry(a, B, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1) != MSCLIP_OK) return MSCLIP_EINVAL;
  const unsigned long long xb = (unsigned long long)B * 3 * H * W * (img_is_bf16 ? 2 : 4);
  if (xb >= 0x80000000ull) return MSCLIP_EINVAL;
  a.xbytes = (unsigned)xb;
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 48) return img_is_bf16 ? launch_front<0, 3, bf16_t>(a, st) : launch_front<0, 3, float>(a, st);
  return img_is_bf16 ? launch_front<0, 6, bf16_t>(a, st) : launch_front<0, 6, float>(a, st);
}
