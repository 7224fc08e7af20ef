// Streaming bf16 MFMA GEMM for small K (64 .. 192): the pointwise convolutions of the conv branch and the lateral
// adapters, out[M, N] = epilogue(alpha * X[M, K] . W[N, K]^T) with M in the 10^5 .. 10^7 range.  These launches are
// pure HBM traffic (a row of X is read once, a row of out written once), so nothing of X goes through LDS:
//   * a workgroup (4 waves) keeps its NW = 32*NT weight rows resident in LDS (row stride K*2 + 16 B: an odd number
//     of 16-byte slots, conflict-free for the 32x32x16 fragment read without a swizzle) and its bias slice;
//   * a wave owns whole 32-row blocks: the X fragments are loaded straight from HBM into the MFMA operand layout
//     (lane = row, 16 B per k-step; the 2K-byte row is consumed entirely by this wave, the partial-line requests
//     hit L1), one 64-deep chunk ahead of the MFMAs, and the next block's first chunk before the epilogue;
//   * epilogue per 32x32 tile through a 4-KiB per-wave staging block (fp32, padded rows): afterwards a lane owns 8
//     consecutive columns of one row -> bias / activation / residual / conversion -> 16-byte stores (bf16) or two
//     16-byte stores (fp32) that cover whole 64 / 128-byte row segments.
// N wider than NW is covered by several column chunks (blockIdx.y); X is then re-read from L2.
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int SW = 4;                       // waves per workgroup
constexpr int SSTG_ROW = 144;               // staging row: 32 fp32 + 16 B pad (9 slots: conflict-free both ways)
constexpr int SSTG_BYTES = 32 * SSTG_ROW;   // per wave

template <int NT>
__global__ __launch_bounds__(SW * 64) void gemm_stream_kernel(const msclip_gemm_desc a) {
  extern __shared__ __attribute__((aligned(16))) char slds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NT * 32;
  const int K = a.K;
  const int wstride = K * 2 + 16;                               // bytes per weight row in LDS
  char* wl = slds;                                              // [NW][wstride]
  float* bl = (float*)(slds + NW * wstride);                    // [NW]
  char* stg = slds + NW * wstride + NW * 4 + wave * SSTG_BYTES;
  const int n0 = blockIdx.y * NW;

  // ---- resident operands
  const bf16_t* __restrict__ W = (const bf16_t*)a.W;
  const int cpr = K / 8;                                        // 16-byte chunks per weight row
  for (int i = tid; i < NW * cpr; i += SW * 64) {
    const int r = i / cpr, c = i - r * cpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (n0 + r < a.N) v = *(const uint4*)(W + (size_t)(n0 + r) * a.ldw + c * 8);
    *(uint4*)(wl + r * wstride + c * 16) = v;
  }
  for (int i = tid; i < NW; i += SW * 64) bl[i] = (a.bias && n0 + i < a.N) ? a.bias[n0 + i] : 0.f;
  __syncthreads();

  const int fr = lane & 31, fhi = lane >> 5;
  const bf16_t* __restrict__ X = (const bf16_t*)a.X;
  const int nblk = (a.M + 31) / 32;
  const int nkc = K / 64;
  const int wpc = gridDim.x * SW;                               // waves per column chunk
  auto xrow = [&](int blk) {
    int m = blk * 32 + fr;
    m = m < a.M ? m : a.M - 1;
    return X + (size_t)m * a.ldx + fhi * 8;
  };
  auto load_chunk = [&](const bf16_t* p, int kc, uint4 (&d)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) d[ks] = *(const uint4*)(p + kc * 64 + ks * 16);
  };

  // read-back mapping of the staging block: pass p covers rows 16p + lane/4, the lane owns columns 8*(lane%4) .. +8
  const int rr = lane >> 2, rc = lane & 3;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 7);

  int blk = blockIdx.x * SW + wave;
  uint4 xa[4], xb[4];
  const bf16_t* xp = nullptr;
  if (blk < nblk) {
    xp = xrow(blk);
    load_chunk(xp, 0, xa);
  }
  for (; blk < nblk; blk += wpc) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int nxt = blk + wpc;
    const bf16_t* xn = nxt < nblk ? xrow(nxt) : xp;

    // ---- K loop: chunk kc lives in xa (even) / xb (odd), the following chunk is requested before the MFMAs
    for (int kc = 0; kc < nkc; kc += 2) {
      if (kc + 1 < nkc) load_chunk(xp, kc + 1, xb);
      else load_chunk(xn, 0, xb);                               // last chunk of the block: next block's first one
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&xa[ks]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bf16x8 wf = *(const bf16x8*)(wl + (t * 32 + fr) * wstride + (kc * 8 + ks * 2 + fhi) * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
        }
      }
      if (kc + 1 < nkc) {
        if (kc + 2 < nkc) load_chunk(xp, kc + 2, xa);
        else load_chunk(xn, 0, xa);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&xb[ks]);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8 wf = *(const bf16x8*)(wl + (t * 32 + fr) * wstride + ((kc + 1) * 8 + ks * 2 + fhi) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xa[ks] = xb[ks];         // odd chunk count: the prefetched chunk moves to xa
      }
    }
    xp = xn;

    // ---- epilogue, one 32x32 tile at a time
    const int m0 = blk * 32;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt0 = n0 + t * 32;
      if (nt0 >= a.N) break;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = acc[t][g * 4 + 0] * a.alpha; v.y = acc[t][g * 4 + 1] * a.alpha;
        v.z = acc[t][g * 4 + 2] * a.alpha; v.w = acc[t][g * 4 + 3] * a.alpha;
        *(float4*)(stg + fr * SSTG_ROW + (g * 2 + fhi) * 16) = v;
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = p * 16 + rr;
        const int m = m0 + row;
        const int n = nt0 + rc * 8;
        const float4 lo = *(const float4*)(stg + row * SSTG_ROW + rc * 32);
        const float4 hi = *(const float4*)(stg + row * SSTG_ROW + rc * 32 + 16);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const float* bp = bl + t * 32 + rc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bp[j];
        if (a.act == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
        }
        if (m < a.M && n < a.N) {
          if (vec) {
            if (a.resid_kind == 1) {
              const float4 r0 = *(const float4*)((const float*)a.resid + (size_t)m * a.ldr + n);
              const float4 r1 = *(const float4*)((const float*)a.resid + (size_t)m * a.ldr + n + 4);
              v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
              v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            } else if (a.resid_kind == 2) {
              const uint4 u = *(const uint4*)((const bf16_t*)a.resid + (size_t)m * a.ldr + n);
              float f[8];
              unpack_bf16x8(u, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += f[j];
            }
            if (a.act == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (a.out_kind == 1) {
              float* o = (float*)a.out + (size_t)m * a.ldo + n;
              *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              uint4 o;
              o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
              o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
              *(uint4*)((bf16_t*)a.out + (size_t)m * a.ldo + n) = o;
            }
          } else {                                              // ragged N / unaligned leading dimensions
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (n + j >= a.N) break;
              float y = v[j];
              if (a.resid_kind == 1) y += ((const float*)a.resid)[(size_t)m * a.ldr + n + j];
              else if (a.resid_kind == 2) y += bf16_to_f32(((const bf16_t*)a.resid)[(size_t)m * a.ldr + n + j]);
              if (a.act == 2) y = fmaxf(y, 0.f);
              if (a.out_kind == 1) ((float*)a.out)[(size_t)m * a.ldo + n + j] = y;
              else ((bf16_t*)a.out)[(size_t)m * a.ldo + n + j] = f32_to_bf16(y);
            }
          }
        }
      }
    }
  }
}

template <int NT>
void launch_stream(const msclip_gemm_desc* d, hipStream_t st, int ncu) {
  constexpr int NW = NT * 32;
  const int chunks = (d->N + NW - 1) / NW;
  const size_t lds = (size_t)NW * (d->K * 2 + 16) + NW * 4 + SW * SSTG_BYTES;
  const int nblk = (d->M + 31) / 32;
  int gx = (2 * ncu + chunks - 1) / chunks;                     // ~2 workgroups per CU over all column chunks
  const int need = (nblk + SW - 1) / SW;
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(gemm_stream_kernel<NT>, dim3(gx, chunks), dim3(SW * 64), lds, st, *d);
}

}  // namespace

// Takes the launch if the problem is a plain dense GEMM with a short K; returns false otherwise.
bool msclip_gemm_small_try(const msclip_gemm_desc* d, hipStream_t st, int ncu) {
  if (d->mode != 0 || d->K > 192 || (d->K % 64) || d->M < 4096) return false;
  if (d->rpg != 0x7fffffff || d->radd || d->roff || d->resid_kind == 3) return false;
  if ((d->ldx % 8) || (d->ldw % 8)) return false;
  // columns per workgroup: weights (K*2 + 16 B per row) + staging must leave room for two workgroups per CU
  // columns per workgroup: the resident weights (K*2 + 16 B per row) stay under 40 KiB so that, with the staging
  // blocks, two workgroups fit a CU: up to 192 columns at K = 64, 96 beyond
  const int n32 = (d->N + 31) / 32;
  int nt;
  if (n32 <= 2) nt = 2;
  else if (d->K == 64 && n32 % 6 == 0) nt = 6;
  else if (n32 % 3 == 0) nt = 3;
  else nt = 2;
  if (nt == 2) launch_stream<2>(d, st, ncu);
  else if (nt == 3) launch_stream<3>(d, st, ncu);
  else launch_stream<6>(d, st, ncu);
  return true;
}
