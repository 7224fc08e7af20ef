// bf16 MFMA GEMM for gfx950 with a gathering X-loader and fused epilogues.
//
//   out[row(m), n] = epilogue( alpha * sum_k X[m, k] * W[n, k] )
//
// X is either a dense row-major [M, K] matrix or an NHWC activation gathered on
// the fly as the implicit-GEMM view of a KHxKW/stride/pad convolution (every
// 16-byte K-chunk of a row is one LDS-DMA source address, padding taps read a
// zero page).  W is always the pre-packed [N][Kpad] bf16 weight (Kpad % 64 == 0).
//
// Tile 128(m) x 128(n) x 64(k), 4 waves as 2x2, each wave 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16 tiles.  Operands are staged with global_load_lds
// (16 B/lane, lane-linear LDS image); the LDS image is the XOR-swizzled
// [row][8 chunks] layout: physical chunk = logical chunk ^ ((row >> 1) & 7),
// applied on the per-lane SOURCE address and again on the ds_read_b128 address
// (cdna_hip_programming.md s5.4 rule 21), conflict-free for the 32x32x16
// fragment read.  MFMA operands are swapped (A = W rows, B = X rows) so every
// lane owns 4 consecutive output columns of one output row.
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

struct RowSrc {          // per staged X row (conv mode)
  long long pix;         // element offset of the (ih0, iw0) tap pixel (may be negative)
  int ih0, iw0;
  int ok;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const msclip_gemm_desc a) {
  __shared__ __attribute__((aligned(1024))) bf16_t smem[2][2][BM * BK];  // [buf][X|W][row*64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int nt_n = (a.N + BN - 1) / BN;
  int id;
  {  // XCD-aware bijective remap: consecutive ids stay on one XCD (shared X rows in its L2)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  const int m0 = (id / nt_n) * BM;
  const int n0 = (id % nt_n) * BN;

  const bf16_t* __restrict__ X = (const bf16_t*)a.X;
  const bf16_t* __restrict__ W = (const bf16_t*)a.W;
  const bf16_t* __restrict__ Z = (const bf16_t*)a.zero;

  // ---- loader state: lane owns (row = (i*4+wave)*8 + lane/8, physical chunk = lane%8), i = 0..3
  const int pc = lane & 7;
  const int lc = pc ^ ((lane >> 4) | ((wave & 1) << 2));  // logical chunk; == pc ^ ((row>>1)&7)
  const int rsub = lane >> 3;

  const bf16_t* xrow[4];
  RowSrc xr[4];
  const bf16_t* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 8 + rsub;
    const int m = m0 + r, n = n0 + r;
    wrow[i] = (n < a.N) ? W + (size_t)n * a.ldw + lc * 8 : nullptr;
    if (MODE == 0) {
      xrow[i] = (m < a.M) ? X + (size_t)m * a.ldx + lc * 8 : nullptr;
    } else {
      const int hw = a.Ho * a.Wo;
      const int bi = m / hw, p = m - bi * hw;
      const int ho = p / a.Wo, wo = p - ho * a.Wo;
      xr[i].ih0 = ho * a.stride - a.pad;
      xr[i].iw0 = wo * a.stride - a.pad;
      xr[i].pix = (((long long)bi * a.H + xr[i].ih0) * a.Wd + xr[i].iw0) * a.Cin;
      xr[i].ok = m < a.M;
    }
  }

  auto stage = [&](int buf, int kt) {
    int e = 0;
    if (MODE == 1) e = a.ktab[kt * 8 + lc];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src;
      if (MODE == 0) {
        src = xrow[i] ? xrow[i] + kt * BK : Z;
      } else {
        const int kh = (e >> 20) & 15, kw = (e >> 24) & 15;
        const int ih = xr[i].ih0 + kh, iw = xr[i].iw0 + kw;
        const bool ok = xr[i].ok && e >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.Wd;
        src = ok ? X + (xr[i].pix + (e & 0xFFFFF)) : Z;
      }
      glds16(src, &smem[buf][0][(i * 4 + wave) * 8 * BK]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src = wrow[i] ? wrow[i] + kt * BK : Z;
      glds16(src, &smem[buf][1][(i * 4 + wave) * 8 * BK]);
    }
  };

  // ---- fragment read state
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fr = lane & 31;
  const int fsw = (lane >> 1) & 7;
  const int fhi = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = a.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const bf16_t* xs = smem[kt & 1][0];
    const bf16_t* ws = smem[kt & 1][1];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ph = ((kk * 2 + fhi) ^ fsw) * 8;
      bf16x8 wf[2], xf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        wf[t] = *(const bf16x8*)(ws + (wn + t * 32 + fr) * BK + ph);
        xf[t] = *(const bf16x8*)(xs + (wm + t * 32 + fr) * BK + ph);
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns row m = .. + (lane&31), columns n = .. + 8g + 4*(lane>>5) + 0..3
  const float* __restrict__ bias = a.bias;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 3);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm + tm * 32 + fr;
    if (m >= a.M) continue;
    const int grp = m / a.rpg;
    const size_t orow = (size_t)(m + grp * a.radd + a.roff);
    size_t rrow = (size_t)m;
    if (a.resid_kind == 3) rrow = (size_t)(m - grp * a.rpg + a.roff);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn + tn * 32 + g * 8 + fhi * 4;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][g * 4 + j] * a.alpha;
        if (vec) {
          if (bias) {
            const float4 bv = *(const float4*)(bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (a.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
          }
          if (a.resid_kind == 1 || a.resid_kind == 3) {
            const float4 rv = *(const float4*)((const float*)a.resid + rrow * a.ldr + n);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
          } else if (a.resid_kind == 2) {
            const uint2 rv = *(const uint2*)((const bf16_t*)a.resid + rrow * a.ldr + n);
            v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
            v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
          }
          if (a.act == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          if (a.out_kind == 1) {
            *(float4*)((float*)a.out + orow * a.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((bf16_t*)a.out + orow * a.ldo + n) = o;
          }
        } else {  // ragged N or unaligned leading dimensions: element-wise tail path
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (n + j >= a.N) break;
            float y = v[j];
            if (bias) y += bias[n + j];
            if (a.act == 1) y = y / (1.f + __expf(-1.702f * y));
            if (a.resid_kind == 1 || a.resid_kind == 3) y += ((const float*)a.resid)[rrow * a.ldr + n + j];
            else if (a.resid_kind == 2) y += bf16_to_f32(((const bf16_t*)a.resid)[rrow * a.ldr + n + j]);
            if (a.act == 2) y = fmaxf(y, 0.f);
            if (a.out_kind == 1) ((float*)a.out)[orow * a.ldo + n + j] = y;
            else ((bf16_t*)a.out)[orow * a.ldo + n + j] = f32_to_bf16(y);
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int msclip_gemm(const msclip_gemm_desc* d, void* stream) {
  if (!d || !d->X || !d->W || !d->out || !d->zero) return MSCLIP_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % BK)) return MSCLIP_EINVAL;
  if (d->mode == 0 && (d->ldx % 8)) return MSCLIP_EINVAL;
  if (d->ldw < d->K || (d->ldw % 8)) return MSCLIP_EINVAL;
  if (d->mode == 1 && (!d->ktab || (d->Cin % 8))) return MSCLIP_EINVAL;
  if (d->rpg <= 0) return MSCLIP_EINVAL;
  const int grid = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  if (d->mode == 0)
    hipLaunchKernelGGL(gemm_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
  else
    hipLaunchKernelGGL(gemm_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
  return msclip_launch_status();
}
