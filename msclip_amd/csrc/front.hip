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
// MFMAs straight into LDS, the CONSUMER runs the 3x3/s2 convolution out of LDS.  The intermediate map never exists
// in HBM.  The stem producer also emits the parallel branch's stage-0 map (same image taps, the other 48 filters)
// for the tile's 16 x 16 interior, staged so it leaves as full lines.
//
// Two forms: front_ws_kernel (8 waves, producer and consumer wave groups running concurrently on two window
// buffers; what ships) and front_kernel (4 waves taking the phases in turn; image widths that are not a multiple of
// 4, and MSCLIP_FRONT_4WAVE=1 for A/B runs).  Batch 512, 224 x 224: stem 700 -> 360 us (4-wave form 545), bottleneck
// conv1->conv2 515 -> 187 us; the write-heavy traffic of the fused stem pass runs at 3.2 TB/s where the unfused
// first pass reaches 4 TB/s, i.e. within 30 % of what this traffic mix gets from HBM.
//
// MFMA operand convention (v_mfma_f32_16x16x32_bf16, weights as A, pixels as B): lane (p = l % 16, q = l / 16) holds
// 8 K-values q*8.. of channel p (A) / of pixel p (B); the result holds channels 4q..4q+3 of pixel p.
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

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
constexpr int KREG = 7;                   // consumer K-steps whose weights live in registers
constexpr int WS2 = (KS - KREG) * 64 + 16;  // LDS row stride of the other K-steps (464: conflict-free over 16 rows)
constexpr int RS = 2 * CM * 2 + 16;        // TAIL staging row: 48 + 48 channels (208 bytes: conflict-free over 16 rows)
constexpr int PSTB = 4 * TS * TS * CM * 2;  // stage-0 map of the tile interior (16 x 16 pixels)
constexpr int PATCHB = 3 * (4 * TS + 3) * (4 * TS + 4) * 2 + 8;   // bf16 image patch [3][35][36]

struct FrontArgs {
  const void* x;        // PROD 0: NCHW image (fp32 / bf16); PROD 1: NHWC bf16 [B, Hm, Wm, 48]
  const void* w1;       // PROD 0: fp32 [27][96]; PROD 1: bf16 [48][64]
  const float* b1;      // PROD 0: [96]; PROD 1: [48]
  const bf16_t* w2;     // bf16 [Cout][448], K = (kh*3 + kw)*48 + c
  const float* b2;      // [Cout]
  bf16_t* side;         // PROD 0: stage-0 map NHWC bf16 [B, Hm, Wm, 48]
  const bf16_t* w3;     // TAIL: conv3 [96][64] and shortcut [96][64] weights (K = 48 padded), b3 = conv3 + shortcut bias
  const bf16_t* wr;
  const float* b3;
  bf16_t* out;          // NHWC bf16 [B, Ho, Wo, Cout]
  unsigned xbytes;      // extent of x (< 2^31: offsets beyond it mark padding)
  unsigned sbytes;      // extent of side
  int B, Hm, Wm, Ho, Wo, Himg, Wimg, ntile, tyn, txn;
};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// relu(bf16(lo)), relu(bf16(hi)) packed: the ReLU runs on the packed pair as a signed 16-bit max with 0 (one
// v_pk_max_i16 for two values; a negative bf16 is a negative int16, -0 becomes +0) -- fmaxf costs two VALU ops per
// value here (NaN canonicalisation), which made the producer issue-bound.
__device__ __forceinline__ unsigned relu_pack_bf16x2(float lo, float hi) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);          // one v_cvt_pk_bf16_f32
  const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, h), s16x2{0, 0});
  return __builtin_bit_cast(unsigned, r);
}

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
    if (t + (int)gridDim.x < a.ntile) request(t + gridDim.x);
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
    // ================= consumer: 3x3 / stride 2 out of LDS
    f32x4 acc[MT][3];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[m][j] = f32x4{cb[j][0], cb[j][1], cb[j][2], cb[j][3]};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
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
      if (oy < a.Ho && ox < a.Wo) {
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

#ifdef WS_TRACE
__device__ unsigned long long g_front_trace[32];
#define WS_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && iter == WS_TRACE) g_front_trace[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised form (what ships): 8 waves.  Waves 0-3 PRODUCE the window of tile i+1 into one of two window
// buffers while waves 4-7 CONSUME tile i out of the other, so the producer's load latency and MFMAs run under the
// consumer's LDS reads and MFMAs instead of in turn.  Two workgroup barriers per tile keep the groups in step
// (B1: image patch of tile i+1 parked | stage-0 interior of tile i written out, first half of the K loop;
//  B2: window i+1 complete | tile i stored).
//  * The consumer keeps its 48 x 448 slice of the 3x3 weights in REGISTERS (42 MFMA A-fragments, loaded once by the
//    persistent workgroup): its K loop reads only the two pixel fragments per step from LDS.
//  * Only the consumer group stores to HBM (the producer's stage-0 map goes through an LDS staging block and leaves
//    as full lines from the consumer waves); only the producer group loads.  A wave that both loads and stores has
//    to wait for its slowly acknowledged stores before the VM counter tells it that an older prefetch has landed.
template <int PROD, int NTT, typename InT, bool TAIL = false>
__global__ __launch_bounds__(512) void front_ws_kernel(FrontArgs a) {
  static_assert(!TAIL || (PROD == 1 && NTT == 3), "the fused block tail belongs to the 48-channel bottleneck");
  extern __shared__ __attribute__((aligned(128))) char lds[];
  char* const mid0 = lds;
  char* const pst = mid0 + 2 * MIDB;                    // PROD 0: stage-0 map of the tile interior [16 x 16][48]
  char* const patch = pst + (PROD == 0 ? PSTB : 0);     // PROD 0: bf16 [3][35][36] image patch
  char* const wl = patch + (PROD == 0 ? PATCHB : 0);    // 3x3 weights, K-steps KREG.. : [COUT][WS2]
  char* const trash = wl + NTT * 16 * WS2;              // 128 bytes that masked-out lanes write to
  char* const rows = trash + 128;                       // TAIL: [2][64 pixels][RS]: conv2 output | centre pixel of x
  char* const wt = rows + 2 * 64 * RS;                  // TAIL: [96][RS]: conv3 weights | shortcut weights
  constexpr int COUT = NTT * 16;
  constexpr int MT = NTT / 3;
  constexpr int NP = PROD == 0 ? 6 : 3;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int PW = 4 * TS + 3, PWS = PW + 1;
  constexpr int PCH = PWS / 4;                          // 16-byte chunks (4 pixels) per patch row: columns 4*ox0-4 .. +35
  constexpr int NPATCH = 3 * PW * PCH, NPL = (NPATCH + 255) / 256;   // 945 chunks, 4 per thread

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  const bool producer = wave < 4;
  const int gw = wave & 3;                              // wave inside its group
  const int gtid = tid & 255;
  const int G = gridDim.x;
  auto coords = [&](int t, int& b, int& oy0, int& ox0) {
    b = t / (a.tyn * a.txn);
    const int r = t - b * (a.tyn * a.txn);
    const int ty = r / a.txn;
    oy0 = ty * TS;
    ox0 = (r - ty * a.txn) * TS;
  };

  if (producer) {
    // =========================================================================================== producer group
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.xbytes);
    bf16x8 wf[NP][PROD == 0 ? 1 : 2];
    float pb[NP][4];
    if constexpr (PROD == 0) {
      const float* w = (const float*)a.w1;
#pragma unroll
      for (int nt = 0; nt < NP; ++nt)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = q * 8 + e;
          wf[nt][0][e] = (__bf16)(k < 27 ? w[(k < 27 ? k : 0) * 96 + nt * 16 + p] : 0.f);
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

    u32x4 pre1[PMT][2];
    u32x4 prawA[NPL], prawB[NPL];                                 // two tiles of patch pieces in flight (fp32 image: 4 pixels; bf16: .x/.y)
    int prow[NPL], pcol[NPL], poff[NPL], plds[NPL];
    int tapc[8];
    if constexpr (PROD == 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        int j = i * 256 + gtid;
        j = j < NPATCH ? j : NPATCH - 1;
        const int ci = j / (PW * PCH), r = (j - ci * PW * PCH) / PCH, c = j - ci * PW * PCH - r * PCH;
        prow[i] = r;
        pcol[i] = 4 * c;
        poff[i] = (ci * a.Himg + r) * a.Wimg + 4 * c;
        plds[i] = ((ci * PW + r) * PWS + 4 * c) * 2;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = q * 8 + e;
        const int ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
        tapc[e] = k < 27 ? ((ci * PW + kh) * PWS + kw + 1) * 2 : 0;   // +1: the patch starts one column early (alignment)
      }
    }
    // Buffer-addressed loads: a pixel outside the image gets an offset beyond the descriptor's range and reads as
    // zero -- no branch and no select on the loaded value, so nothing waits on a load before its use an iteration on.
    auto request = [&](int t, u32x4 (&praw)[NPL]) __attribute__((always_inline)) {
      int b, oy0, ox0;
      coords(t, b, oy0, ox0);
      if constexpr (PROD == 1) {
#pragma unroll
        for (int i = 0; i < PMT; ++i) {
          int mp = (gw + 4 * i) * 16 + p;                         // wave 3's fifth tile is past the window: all masked
          const bool inw = mp < NMID;
          mp = inw ? mp : NMID - 1;
          const int my = mp / MS, mx = mp - my * MS;
          const int gy = 2 * oy0 - 1 + my, gx = 2 * ox0 - 1 + mx;
          const bool in = inw && (unsigned)gy < (unsigned)a.Hm && (unsigned)gx < (unsigned)a.Wm;
          const unsigned src = (unsigned)(((b * a.Hm + gy) * a.Wm + gx) * (CM * 2));
          unsigned o0 = in ? src + q * 16 : OOB, o1 = (in && q < 2) ? src + 64 + q * 16 : OOB;
          asm volatile("" : "+v"(o0), "+v"(o1));
          pre1[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, o0, 0, 0);
          pre1[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, o1, 0, 0);
        }
      } else {
        // rows 4*oy0-3 .. +34, columns 4*ox0-4 .. +35: a 16-byte chunk is inside the image or outside as a whole
        // (image width is a multiple of 4)
        const int py0 = 4 * oy0 - 3, px0 = 4 * ox0 - 4;
        const int base = (b * 3 * a.Himg + py0) * a.Wimg + px0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const bool ok = (unsigned)(py0 + prow[i]) < (unsigned)a.Himg && (unsigned)(px0 + pcol[i]) < (unsigned)a.Wimg;
          unsigned off = ok ? (unsigned)(base + poff[i]) * (unsigned)sizeof(InT) : OOB;
          asm volatile("" : "+v"(off));
          if constexpr (sizeof(InT) == 4) praw[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
          else {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
            praw[i] = u32x4{v[0], v[1], 0u, 0u};
          }
        }
      }
    };
    auto park_patch = [&](u32x4 (&praw)[NPL]) __attribute__((always_inline)) {
      if constexpr (PROD == 0) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          u32x2 h;
          if constexpr (sizeof(InT) == 4) {
            h[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{__uint_as_float(praw[i][0]), __uint_as_float(praw[i][1])}, bf16x2));
            h[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{__uint_as_float(praw[i][2]), __uint_as_float(praw[i][3])}, bf16x2));
          } else {
            h[0] = praw[i][0];
            h[1] = praw[i][1];
          }
          // the last piece of threads past the end repeats chunk NPATCH-1 with the same data: harmless
          *(u32x2*)(patch + plds[i]) = h;
        }
      }
    };
    auto produce = [&](int t, int wb) __attribute__((always_inline)) {
      char* const mid = mid0 + wb * MIDB;
      int b, oy0, ox0;
      coords(t, b, oy0, ox0);
#pragma unroll
      for (int i = 0; i < PMT; ++i) {
        const int mp = (gw + 4 * i) * 16 + p;
        const int mpc = mp < NMID ? mp : NMID - 1;
        const int my = mpc / MS, mx = mpc - my * MS;
        const int gy = 2 * oy0 - 1 + my, gx = 2 * ox0 - 1 + mx;
        const bool in = (unsigned)gy < (unsigned)a.Hm && (unsigned)gx < (unsigned)a.Wm;
        char* const mdst = mp < NMID ? mid + mp * PS : trash;
        [[maybe_unused]] char* const sdst = (mp < NMID && my >= 1 && mx >= 1) ? pst + ((my - 1) * 16 + (mx - 1)) * (CM * 2) : trash;
        bf16x8 xb[2];
        if constexpr (PROD == 1) {
          xb[0] = as_bf16x8(pre1[i][0]);
          xb[1] = as_bf16x8(pre1[i][1]);
          if constexpr (TAIL) {
            // the strided 1x1 shortcut samples x at the window's odd positions (= input pixel (2*oy, 2*ox)): park
            // those pixels beside the slot where the consumer will put conv2's output for the same output pixel
            const bool ctr = mp < NMID && (my & 1) && (mx & 1);
            char* const cdst = rows + (wb * 64 + ((my - 1) >> 1) * 8 + ((mx - 1) >> 1)) * RS + CM * 2;
            *(u32x4*)(ctr ? cdst + q * 16 : trash + q * 16) = pre1[i][0];
            *(u32x4*)((ctr && q < 2) ? cdst + 64 + q * 16 : trash + q * 16) = pre1[i][1];
          }
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
          o.x = relu_pack_bf16x2(acc[0], acc[1]);
          o.y = relu_pack_bf16x2(acc[2], acc[3]);
          // lanes with nothing to write aim at a scratch row instead of branching around the write
          if (nt < 3) {
            if (!in) o.x = o.y = 0u;                              // zero padding of the 3x3 that follows
            *(uint2*)(mdst + (nt * 16 + 4 * q) * 2) = o;
          } else {                                                // stage-0 map of the parallel branch: interior only
            *(uint2*)(sdst + ((nt - 3) * 16 + 4 * q) * 2) = o;
          }
        }
      }
    };

    __builtin_amdgcn_s_waitcnt(0x0070);
    int t = blockIdx.x;
    int buf = 0;
    [[maybe_unused]] int iter = 0;
    if constexpr (PROD == 0) {
      // HBM read latency under this kernel's write load is several microseconds: two tiles of patch pieces stay in
      // flight (registers A / B alternate; a tile's pieces are requested two iterations before they are parked)
      request(t, prawA);
      park_patch(prawA);
      __syncthreads();
      if (t + G < a.ntile) request(t + G, prawA);
      if (t + 2 * G < a.ntile) request(t + 2 * G, prawB);
      produce(t, 0);
      __syncthreads();
      auto step = [&](u32x4 (&praw)[NPL]) __attribute__((always_inline)) {   // one iteration: tile t consumed, t+G produced
        const int tn = t + G;
        if (gw == 0) WS_STAMP(0);
        if (tn < a.ntile) {
          park_patch(praw);
          if (tn + 2 * G < a.ntile) request(tn + 2 * G, praw);
        }
        if (gw == 0) WS_STAMP(1);
        __syncthreads();                                          // B1
        if (gw == 0) WS_STAMP(2);
        if (tn < a.ntile) produce(tn, buf ^ 1);
        if (gw == 0) WS_STAMP(3);
        __syncthreads();                                          // B2
        if (gw == 0) WS_STAMP(4);
        buf ^= 1;
        t += G;
        ++iter;
      };
      while (t < a.ntile) {
        step(prawA);
        if (t >= a.ntile) break;
        step(prawB);
      }
    } else {
      request(t, prawA);
      __syncthreads();
      produce(t, 0);
      if (t + G < a.ntile) request(t + G, prawA);
      __syncthreads();
      for (; t < a.ntile; t += G) {
        const int tn = t + G;
        __syncthreads();                                          // B1
        if (tn < a.ntile) {
          produce(tn, buf ^ 1);
          if (tn + G < a.ntile) request(tn + G, prawA);
        }
        __syncthreads();                                          // B2
        buf ^= 1;
      }
    }
  } else {
    // =========================================================================================== consumer group
    const int cm0 = NTT == 6 ? (gw & 1) * 2 : gw;
    const int cn0 = NTT == 6 ? (gw >> 1) * 3 : 0;
    // this wave's 48 output channels: K-steps 0..KREG-1 in registers (84), the rest in LDS (conflict-free rows)
    bf16x8 wa[KREG][3];
#pragma unroll
    for (int s = 0; s < KREG; ++s)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        wa[s][j] = as_bf16x8(*(const u32x4*)(a.w2 + (size_t)((cn0 + j) * 16 + p) * KP + (4 * s + q) * 8));
    for (int i = gtid; i < COUT * (KS - KREG) * 4; i += 256) {
      const int r = i / ((KS - KREG) * 4), c = i - r * ((KS - KREG) * 4);
      *(u32x4*)(wl + r * WS2 + c * 16) = *(const u32x4*)(a.w2 + (size_t)r * KP + KREG * 32 + c * 8);
    }
    const char* const wrow = wl + (cn0 * 16 + p) * WS2 + q * 16;
    [[maybe_unused]] float tb[6][4];
    if constexpr (TAIL) {
      for (int i = gtid; i < 96 * 12; i += 256) {                 // [W3 | Wr] rows of 96 K-values
        const int n = i / 12, c = i - n * 12;
        const bf16_t* src = c < 6 ? a.w3 + n * 64 + c * 8 : a.wr + n * 64 + (c - 6) * 8;
        *(u32x4*)(wt + n * RS + c * 16) = *(const u32x4*)src;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[j][r] = a.b3[j * 16 + 4 * q + r];
    }
    float cb[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[j][r] = a.b2[(cn0 + j) * 16 + 4 * q + r];
    int pixbase[MT], oyl[MT], oxl[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int op = (cm0 + m) * 16 + p;
      oyl[m] = op >> 3;
      oxl[m] = op & 7;
      pixbase[m] = (2 * oyl[m] * MS + 2 * oxl[m]) * PS;
    }
    int tapoff[KS];                                               // chunk 4s+q of K = (tap, 8 channels)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 4 * s + q;
      const int tap = c / 6, c8 = c - tap * 6;
      const int kh = tap / 3, kw = tap - kh * 3;
      tapoff[s] = c < 54 ? (kh * MS + kw) * PS + c8 * 16 : 0;     // chunks 54, 55 meet zero weights: any finite data
    }
    // stage-0 write-out: piece i of this thread = 16 bytes of row (ch / 96) of the 16 x 16 interior
    constexpr int NSO = PSTB / 16 / 256;
    int so_off[NSO], so_row[NSO], so_col[NSO];
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(PROD == 0 ? (const void*)a.side : (const void*)a.out, PROD == 0 ? a.sbytes : 16u);
    if constexpr (PROD == 0) {
#pragma unroll
      for (int i = 0; i < NSO; ++i) {
        const int ch = i * 256 + gtid;
        const int row = ch / 96, within = ch - row * 96;
        so_row[i] = row;
        so_col[i] = within / 6;
        so_off[i] = row * a.Wm * (CM * 2) + within * 16;
      }
    }

    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    __syncthreads();
    int buf = 0;
    [[maybe_unused]] int iter = 0;
    for (int t = blockIdx.x; t < a.ntile; t += G, ++iter) {
      int b, oy0, ox0;
      if (gw == 0) WS_STAMP(8);
      coords(t, b, oy0, ox0);
      const char* mid = mid0 + buf * MIDB;
      // the stage-0 interior of this tile leaves as 16 runs of 1536 contiguous bytes: six 16-byte pieces per thread,
      // read out of the staging block now (the producer rewrites it after B1) and stored one per two K-steps --
      // back to back they stall the wave on the store queue (the CU's store path moves ~12 B/clk)
      [[maybe_unused]] const unsigned tbase = (unsigned)(((b * a.Hm + 2 * oy0) * a.Wm + 2 * ox0) * (CM * 2));   // wave-uniform
      [[maybe_unused]] u32x4 sreg[NSO];
      if constexpr (PROD == 0) {
#pragma unroll
        for (int i = 0; i < NSO; ++i) sreg[i] = *(const u32x4*)(pst + (i * 256 + gtid) * 16);
      }
      auto side_piece = [&](int i) {
        if constexpr (PROD == 0) {
          const bool ok = 2 * oy0 + so_row[i] < a.Hm && 2 * ox0 + so_col[i] < a.Wm;
          __builtin_amdgcn_raw_buffer_store_b128(sreg[i], rs, ok ? (unsigned)so_off[i] : OOB, tbase, 0);
          // gfx950 / ROCm 7.2: a buffer store of more than 8 bytes WITH an SGPR offset reads its data VGPRs after issue, and
          // hipcc re-used the first data register for the next piece's address one instruction later (v_or v170 right behind
          // buffer_store_dwordx4 v[170:173] ... s50 offen) without the two wait states the ISA asks for: under contention (text
          // block 0 on another stream) single dwords of the stage-0 map came out wrong -- rare, timing-dependent, 1e-3 on the
          // image features (round 4, tools/probes/repeat_probe.py).  The data stays live across two idle states instead.
          asm volatile("s_nop 1" : "+v"(sreg[i]));
        }
      };
      f32x4 acc[MT][3];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[m][j] = f32x4{cb[j][0], cb[j][1], cb[j][2], cb[j][3]};
      auto ksteps = [&](int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
          bf16x8 xb[MT], wk[3];
          const int to = tapoff[s];
          if ((s & 1) == 0 && s / 2 < NSO) side_piece(s / 2);
#pragma unroll
          for (int m = 0; m < MT; ++m) xb[m] = as_bf16x8(*(const u32x4*)(mid + pixbase[m] + to));
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if (s < KREG) wk[j] = wa[s < KREG ? s : 0][j];
            else wk[j] = as_bf16x8(*(const u32x4*)(wrow + j * 16 * WS2 + (s - KREG) * 64));
          }
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[j], xb[m], acc[m][j], 0, 0, 0);
        }
      };
      ksteps(0, KS / 2);
      if (gw == 0) WS_STAMP(9);
      __syncthreads();                                            // B1
      if (gw == 0) WS_STAMP(10);
      ksteps(KS / 2, KS);
      if (gw == 0) WS_STAMP(11);
      if constexpr (TAIL) {
        // Block tail: out = relu([W3 | Wr] . [relu(conv2) ; x(2oy, 2ox)] + (b3 + br)) -- conv3, the strided shortcut
        // and the residual add as ONE K = 96 contraction per pixel.  This wave's 16 pixels of conv2 output go to its
        // own rows of the staging block (beside the centre pixels the producer parked there), come back in the MFMA
        // operand layout and meet the 96 x 96 weight block in LDS; neither conv2's output nor the shortcut's reaches HBM.
        char* const rr = rows + (buf * 64 + gw * 16 + p) * RS;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          uint2 o;
          o.x = relu_pack_bf16x2(acc[0][j][0], acc[0][j][1]);
          o.y = relu_pack_bf16x2(acc[0][j][2], acc[0][j][3]);
          *(uint2*)(rr + (j * 16 + 4 * q) * 2) = o;
        }
        f32x4 acc2[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc2[j] = f32x4{tb[j][0], tb[j][1], tb[j][2], tb[j][3]};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const bf16x8 xb = as_bf16x8(*(const u32x4*)(rr + (4 * s + q) * 16));
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const bf16x8 wk = as_bf16x8(*(const u32x4*)(wt + (j * 16 + p) * RS + (4 * s + q) * 16));
            acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk, xb, acc2[j], 0, 0, 0);
          }
        }
        const int oy = oy0 + oyl[0], ox = ox0 + oxl[0];
        const bool ok = oy < a.Ho && ox < a.Wo;
        bf16_t* const dst = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * 96;
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {                           // channel-tile pairs: 16 contiguous bytes per lane
          uint2 lo, hi;
          lo.x = relu_pack_bf16x2(acc2[2 * jp][0], acc2[2 * jp][1]);
          lo.y = relu_pack_bf16x2(acc2[2 * jp][2], acc2[2 * jp][3]);
          hi.x = relu_pack_bf16x2(acc2[2 * jp + 1][0], acc2[2 * jp + 1][1]);
          hi.y = relu_pack_bf16x2(acc2[2 * jp + 1][2], acc2[2 * jp + 1][3]);
          auto sw = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
          lo.x = sw[0]; hi.x = sw[1];
          sw = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
          lo.y = sw[0]; hi.y = sw[1];
          if (ok) *(u32x4*)(dst + jp * 32 + (q & 1) * 16 + (q >> 1) * 8) = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int oy = oy0 + oyl[m], ox = ox0 + oxl[m];
          // channel tiles cn0 and cn0+1: after swapping tile 0's odd lane rows with tile 1's even rows a lane holds 16
          // contiguous bytes (rows 0, 2 -> channels 8*(q/2).. of tile 0, rows 1, 3 -> 16 + 8*(q/2).. of tile 1)
          uint2 o[3];
  #pragma unroll
          for (int j = 0; j < 3; ++j) {
            o[j].x = relu_pack_bf16x2(acc[m][j][0], acc[m][j][1]);
            o[j].y = relu_pack_bf16x2(acc[m][j][2], acc[m][j][3]);
          }
          uint2 lo = o[0], hi = o[1];
          auto sw = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
          lo.x = sw[0]; hi.x = sw[1];
          sw = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
          lo.y = sw[0]; hi.y = sw[1];
          if (oy < a.Ho && ox < a.Wo) {
            bf16_t* dst = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * COUT + cn0 * 16;
            *(u32x4*)(dst + (q & 1) * 16 + (q >> 1) * 8) = u32x4{lo.x, lo.y, hi.x, hi.y};
            *(uint2*)(dst + 32 + 4 * q) = o[2];
          }
        }
      }
      if (gw == 0) WS_STAMP(12);
      __syncthreads();                                            // B2
      if (gw == 0) WS_STAMP(13);
      buf ^= 1;
    }
  }
}

template <int PROD, int NTT, typename InT>
int launch_front(FrontArgs a, hipStream_t st) {
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    ncu = 256;
  const char* old4 = getenv("MSCLIP_FRONT_4WAVE");              // the single-group kernel, for A/B runs only
  // the 8-wave stem kernel fetches the image in aligned 4-pixel pieces: other widths take the single-group kernel
  if ((old4 && old4[0] == '1') || (PROD == 0 && a.Wimg % 4 != 0)) {
    const size_t lds = (size_t)NTT * 16 * WS + MIDB + (PROD == 0 ? PSTB + PATCHB : 0);
    bool attr_ok = true;
    MSCLIP_LDS_ATTR((front_kernel<PROD, NTT, InT>), lds, attr_ok);
    if (!attr_ok) return MSCLIP_ELAUNCH;
    const int per_cu = (int)(160 * 1024 / lds) > 2 ? 2 : (int)(160 * 1024 / lds);
    int grid = ncu * (per_cu < 1 ? 1 : per_cu);
    if (grid > a.ntile) grid = a.ntile;
    hipLaunchKernelGGL((front_kernel<PROD, NTT, InT>), dim3(grid), dim3(256), lds, st, a);
    return msclip_launch_status();
  }
  const size_t lds = (size_t)2 * MIDB + (PROD == 0 ? PSTB + PATCHB : 0) + (size_t)NTT * 16 * WS2 + 128;
  bool attr_ok = true;
  MSCLIP_LDS_ATTR((front_ws_kernel<PROD, NTT, InT>), lds, attr_ok);
  if (!attr_ok) return MSCLIP_ELAUNCH;
  int grid = ncu;                                               // one 8-wave workgroup per CU, persistent
  if (grid > a.ntile) grid = a.ntile;
  hipLaunchKernelGGL((front_ws_kernel<PROD, NTT, InT>), dim3(grid), dim3(512), lds, st, a);
  return msclip_launch_status();
}

int launch_block_tail(FrontArgs a, hipStream_t st) {
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    ncu = 256;
  const size_t lds = (size_t)2 * MIDB + (size_t)3 * 16 * WS2 + 128 + 2 * 64 * RS + 96 * RS;
  bool attr_ok = true;
  MSCLIP_LDS_ATTR((front_ws_kernel<1, 3, bf16_t, true>), lds, attr_ok);
  if (!attr_ok) return MSCLIP_ELAUNCH;
  int grid = ncu;
  if (grid > a.ntile) grid = a.ntile;
  hipLaunchKernelGGL((front_ws_kernel<1, 3, bf16_t, true>), dim3(grid), dim3(512), lds, st, a);
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

#ifdef WS_TRACE
extern "C" int msclip_front_trace_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_front_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int msclip_conv1x1_conv3x3s2(const void* x, const void* w1, const float* b1, const void* w2,
                                        const float* b2, void* out, int B, int H, int W, int Cout, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_conv1x1_conv3x3s2, stream, x, w1, b1, w2, b2, out, B, H, W, Cout);
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

extern "C" int msclip_convresblock48_s2(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                        const void* w3, const void* wr, const float* b3r, void* out, int B, int H,
                                        int W, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_convresblock48_s2, stream, x, w1, b1, w2, b2, w3, wr, b3r, out, B, H, W);
  if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !wr || !b3r || !out || B <= 0 || H <= 0 || W <= 0) return MSCLIP_EINVAL;
  FrontArgs a{};
  a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.side = nullptr; a.out = (bf16_t*)out;
  a.w3 = (const bf16_t*)w3; a.wr = (const bf16_t*)wr; a.b3 = b3r;
  a.Himg = a.Wimg = 0;
  if (fill_geometry(a, B, H, W) != MSCLIP_OK) return MSCLIP_EINVAL;
  const unsigned long long xb = (unsigned long long)B * H * W * CM * 2;
  if (xb >= 0x80000000ull) return MSCLIP_EINVAL;
  a.xbytes = (unsigned)xb;
  return launch_block_tail(a, (hipStream_t)stream);
}

extern "C" int msclip_stem_dual_conv3x3s2(const void* img, int img_is_bf16, const float* w, const float* bias,
                                          void* out_b, const void* w2, const float* b2, void* out2, int B, int H,
                                          int W, int Cout, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_stem_dual_conv3x3s2, stream, img, img_is_bf16, w, bias, out_b, w2, b2, out2, B, H, W, Cout);
  if (!img || !w || !bias || !out_b || !w2 || !b2 || !out2 || B <= 0 || H <= 0 || W <= 0 || (Cout != 48 && Cout != 96))
    return MSCLIP_EINVAL;
  FrontArgs a{};
  a.x = img; a.w1 = w; a.b1 = bias; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.side = (bf16_t*)out_b; a.out = (bf16_t*)out2;
  a.Himg = H; a.Wimg = W;
  if (fill_geometry(a, B, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1) != MSCLIP_OK) return MSCLIP_EINVAL;
  const unsigned long long xb = (unsigned long long)B * 3 * H * W * (img_is_bf16 ? 2 : 4);
  if (xb >= 0x80000000ull) return MSCLIP_EINVAL;
  a.xbytes = (unsigned)xb;
  const unsigned long long sb = (unsigned long long)B * a.Hm * a.Wm * CM * 2;
  if (sb >= 0x80000000ull) return MSCLIP_EINVAL;
  a.sbytes = (unsigned)sb;
  hipStream_t st = (hipStream_t)stream;
  if (Cout == 48) return img_is_bf16 ? launch_front<0, 3, bf16_t>(a, st) : launch_front<0, 3, float>(a, st);
  return img_is_bf16 ? launch_front<0, 6, bf16_t>(a, st) : launch_front<0, 6, float>(a, st);
}
