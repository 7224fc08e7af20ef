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
// The same kernel runs the implicit-GEMM convolutions whose weights fit LDS (CONV = true): the 1x1 stride-2
// shortcuts and the two 3x3 stride-2 layers with 48 input channels (K = 432).  A lane's row is an output pixel and
// k-step s reads 16 bytes at pixel base + ktab[2s + lane/32] (the table of include/msclip_hip.h, mode 1; taps outside
// the image and table padding read as zero).
#include <type_traits>
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int SSTG_ROW = 144;               // staging row: 32 fp32 + 16 B pad (an odd number of 16-byte slots)
constexpr int SSTG_BYTES = 32 * SSTG_ROW;   // per wave

// NT: 32-column tiles per wave, NKC: K / 64, SWV: waves per workgroup
// BN (msclip_gemm_desc.bn_mode; instantiations of their own so that the standard kernel stays what it was): 1 = column sums of the
// accumulators instead of an output (train-mode BatchNorm statistics: a lane keeps the sums of its 8 columns over its rows of every
// block, four exchanges fold the 16 lanes that share the columns at the end, the wave writes its row of `part`), 2 = normalise
// epilogue (out = act(x scale + shift [+ resid]), out2 = x a + b, constants per column in LDS).
template <int NT, int NKC, bool CONV, int SWV, int BN = 0>
__global__ __launch_bounds__(SWV * 64) void gemm_stream_kernel(const msclip_gemm_desc a) {
  extern __shared__ __attribute__((aligned(16))) char slds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NT * 32, K = NKC * 64;
  constexpr int wstride = K * 2 + 16;                           // bytes per weight row in LDS
  char* wl = slds;                                              // [NW][wstride]
  constexpr int NBL = BN == 2 ? 4 * NW : NW;
  float* bl = (float*)(slds + NW * wstride);                    // [NW] bias; BN 2: [4][NW] = scale, shift, a, b
  char* stg = slds + NW * wstride + NBL * 4 + wave * SSTG_BYTES;
  const int n0 = blockIdx.y * NW;

  // ---- resident operands
  const bf16_t* __restrict__ W = (const bf16_t*)a.W;
  constexpr int cpr = K / 8;                                    // 16-byte chunks per weight row
  for (int i = tid; i < NW * cpr; i += SWV * 64) {
    const int r = i / cpr, c = i - r * cpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (n0 + r < a.N) v = *(const uint4*)(W + (size_t)(n0 + r) * a.ldw + c * 8);
    *(uint4*)(wl + r * wstride + c * 16) = v;
  }
  if constexpr (BN == 2) {
    // bn_consts = msclip_bn_finish's rows (mean, var, rstd, scale, shift) -> (scale, shift, a = rstd, b = -mean rstd) per column
    for (int c = tid; c < NW; c += SWV * 64) {
      const bool in = n0 + c < a.N;
      const float* k = a.bn_consts + n0 + c;
      const float mean = in ? k[0] : 0.f, rstd = in ? k[2 * (size_t)a.N] : 0.f;
      bl[c] = in ? k[3 * (size_t)a.N] : 0.f;
      bl[NW + c] = in ? k[4 * (size_t)a.N] : 0.f;
      bl[2 * NW + c] = rstd;
      bl[3 * NW + c] = -mean * rstd;
    }
  } else {
    for (int i = tid; i < NW; i += SWV * 64) bl[i] = (a.bias && n0 + i < a.N) ? a.bias[n0 + i] : 0.f;
  }
  __syncthreads();

  const int fr = lane & 31, fhi = lane >> 5;
  const bf16_t* __restrict__ X = (const bf16_t*)a.X;
  const int nblk = (a.M + 31) / 32;
  const int wpc = gridDim.x * SWV;                              // waves per column chunk

  // conv mode: this lane's chunk-table entries, k-step s -> entry 2s + fhi
  int ent[CONV ? NKC * 4 : 1];
  if (CONV) {
#pragma unroll
    for (int s2 = 0; s2 < NKC * 4; ++s2) ent[s2] = a.ktab[s2 * 2 + fhi];
  }
  struct Row { const bf16_t* p; int ih0, iw0; };
  auto xrow = [&](int blk) {
    int m = blk * 32 + fr;
    m = m < a.M ? m : a.M - 1;
    Row r;
    if (CONV) {
      const int hw = a.Ho * a.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      r.ih0 = ho * a.stride - a.pad;
      r.iw0 = wo * a.stride - a.pad;
      r.p = X + (((long long)b * a.H + r.ih0) * a.Wd + r.iw0) * a.Cin;   // may point before the image: guarded below
    } else {
      r.p = X + (size_t)m * a.ldx + fhi * 8;
      r.ih0 = r.iw0 = 0;
    }
    return r;
  };
  auto load_chunk = [&](const Row& r, auto kcc, uint4 (&d)[4]) {
    constexpr int kc = decltype(kcc)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (CONV) {
        const int e = ent[kc * 4 + ks];
        const int ih = r.ih0 + ((e >> 20) & 15), iw = r.iw0 + ((e >> 24) & 15);
        const bool ok = e >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.Wd;
        const uint4 v = *(const uint4*)(ok ? r.p + (e & 0xfffff) : X);
        d[ks] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      } else {
        d[ks] = *(const uint4*)(r.p + kc * 64 + ks * 16);
      }
    }
  };

  // read-back mapping of the staging block: pass p covers rows 16p + lane/4, the lane owns columns 8*(lane%4) .. +8
  const int rr = lane >> 2, rc = lane & 3;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 7);

  float csum[BN == 1 ? NT * 8 : 1], csq[BN == 1 ? NT * 8 : 1];   // BN 1: this lane's column sums (tile t, column rc * 8 + j)
  if constexpr (BN == 1) {
#pragma unroll
    for (int i = 0; i < NT * 8; ++i) csum[i] = csq[i] = 0.f;
  }

  int blk = blockIdx.x * SWV + wave;
  uint4 xq[2][4];
  Row xp = xrow(blk < nblk ? blk : 0);
  if (blk < nblk) load_chunk(xp, std::integral_constant<int, 0>{}, xq[0]);
  for (; blk < nblk; blk += wpc) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int nxt = blk + wpc;
    const Row xn = xrow(nxt < nblk ? nxt : blk);

    // ---- K loop: chunk kc sits in xq[kc & 1]; the following chunk (or the next block's first) is requested first
    auto kstep = [&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      if (kc + 1 < NKC) load_chunk(xp, std::integral_constant<int, (kc + 1 < NKC ? kc + 1 : 0)>{}, xq[(kc + 1) & 1]);
      else load_chunk(xn, std::integral_constant<int, 0>{}, xq[(kc + 1) & 1]);
      if constexpr (NT <= 3) {
        // narrow outputs: the chunk's 4 * NT weight fragments are requested from LDS before its MFMAs (and pinned there:
        // the scheduler otherwise sinks each ds_read to its MFMA behind an lgkmcnt(0), one exposed LDS round trip per MFMA)
        bf16x8 wf[4][NT];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            wf[ks][t] = *(const bf16x8*)(wl + (t * 32 + fr) * wstride + (kc * 8 + ks * 2 + fhi) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&xq[kc & 1][ks]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][t], xf, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(&xq[kc & 1][ks]);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8 wf = *(const bf16x8*)(wl + (t * 32 + fr) * wstride + (kc * 8 + ks * 2 + fhi) * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
          }
        }
      }
    };
    kstep(std::integral_constant<int, 0>{});
    if constexpr (NKC > 1) kstep(std::integral_constant<int, 1>{});
    if constexpr (NKC > 2) kstep(std::integral_constant<int, 2>{});
    if constexpr (NKC > 3) kstep(std::integral_constant<int, 3>{});
    if constexpr (NKC > 4) kstep(std::integral_constant<int, 4>{});
    if constexpr (NKC > 5) kstep(std::integral_constant<int, 5>{});
    if constexpr (NKC > 6) kstep(std::integral_constant<int, 6>{});
    static_assert(NKC <= 7, "unrolled K chunks");
    if constexpr (NKC & 1) {                                    // the prefetched chunk moves to the even buffer
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xq[0][ks] = xq[1][ks];
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
        if constexpr (BN == 1) {
          if (m < a.M) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              csum[t * 8 + j] += v[j];
              csq[t * 8 + j] = fmaf(v[j], v[j], csq[t * 8 + j]);
            }
          }
          continue;
        }
        const float* bp = bl + t * 32 + rc * 8;
        if constexpr (BN == 2) {
          if (m < a.M && n < a.N) {
            float y[8], xh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              y[j] = fmaf(v[j], bp[j], bp[NW + j]);
              xh[j] = fmaf(v[j], bp[2 * NW + j], bp[3 * NW + j]);
            }
            if (a.resid_kind == 2) {
              float f[8];
              unpack_bf16x8(*(const uint4*)((const bf16_t*)a.resid + (size_t)m * a.ldr + n), f);
#pragma unroll
              for (int j = 0; j < 8; ++j) y[j] += f[j];
            }
            if (a.act == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) y[j] = fmaxf(y[j], 0.f);
            }
            uint4 o, h;
            o.x = pack_bf16x2(y[0], y[1]); o.y = pack_bf16x2(y[2], y[3]);
            o.z = pack_bf16x2(y[4], y[5]); o.w = pack_bf16x2(y[6], y[7]);
            h.x = pack_bf16x2(xh[0], xh[1]); h.y = pack_bf16x2(xh[2], xh[3]);
            h.z = pack_bf16x2(xh[4], xh[5]); h.w = pack_bf16x2(xh[6], xh[7]);
            *(uint4*)((bf16_t*)a.out + (size_t)m * a.ldo + n) = o;
            *(uint4*)((bf16_t*)a.out2 + (size_t)m * a.ldo + n) = h;
          }
          continue;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bp[j];
        if (a.act == 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
        }
        // store row: m + (m / rpg) * radd + roff (include/msclip_hip.h; the parity-class input gradients of the stride-2
        // convolutions scatter their rows into dX this way, train_conv.py)
        const size_t orow = a.rpg == 0x7fffffff ? (size_t)m : (size_t)(m + (m / a.rpg) * a.radd + a.roff);
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
            } else if (a.resid_kind == 6) {                     // accumulate into the bf16 map at the store row
              const uint4 u = *(const uint4*)((const bf16_t*)a.resid + orow * a.ldr + n);
              float f[8];
              unpack_bf16x8(u, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += f[j];
            } else if (a.resid_kind == 5) {                     // ReLU backward: mask by the saved activation at the store row
              const uint4 u = *(const uint4*)((const bf16_t*)a.resid + orow * a.ldr + n);
              float f[8];
              unpack_bf16x8(u, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = f[j] > 0.f ? v[j] : 0.f;
            }
            if (a.act == 2) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (a.out_kind == 1) {
              float* o = (float*)a.out + orow * a.ldo + n;
              *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              uint4 o;
              o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
              o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
              *(uint4*)((bf16_t*)a.out + orow * a.ldo + n) = o;
            }
          } else {                                              // ragged N / unaligned leading dimensions
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (n + j >= a.N) break;
              float y = v[j];
              if (a.resid_kind == 1) y += ((const float*)a.resid)[(size_t)m * a.ldr + n + j];
              else if (a.resid_kind == 2) y += bf16_to_f32(((const bf16_t*)a.resid)[(size_t)m * a.ldr + n + j]);
              else if (a.resid_kind == 5) y = bf16_to_f32(((const bf16_t*)a.resid)[orow * a.ldr + n + j]) > 0.f ? y : 0.f;
              else if (a.resid_kind == 6) y += bf16_to_f32(((const bf16_t*)a.resid)[orow * a.ldr + n + j]);
              if (a.act == 2) y = fmaxf(y, 0.f);
              if (a.out_kind == 1) ((float*)a.out)[orow * a.ldo + n + j] = y;
              else ((bf16_t*)a.out)[orow * a.ldo + n + j] = f32_to_bf16(y);
            }
          }
        }
      }
    }
  }
  if constexpr (BN == 1) {
    // the 16 lanes that share (lane & 3) hold the same columns: fold them, lanes 0-3 write the wave's row
    float* prow = a.part + ((size_t)blockIdx.x * SWV + wave) * 2 * a.N;
#pragma unroll
    for (int i = 0; i < NT * 8; ++i) {
      float u = csum[i], q = csq[i];
#pragma unroll
      for (int msk = 4; msk < 64; msk <<= 1) {
        u += __shfl_xor(u, msk, 64);
        q += __shfl_xor(q, msk, 64);
      }
      const int n = n0 + (i >> 3) * 32 + rc * 8 + (i & 7);
      if (rr == 0 && n < a.N) {
        prow[n] = u;
        prow[a.N + n] = q;
      }
    }
  }
}

template <int NT, int NKC, bool CONV, int SWV, int BN = 0>
void launch_stream_bn(const msclip_gemm_desc* d, hipStream_t st, int ncu, int wg_per_cu) {
  constexpr int NW = NT * 32;
  const int chunks = (d->N + NW - 1) / NW;
  const size_t lds = (size_t)NW * (NKC * 128 + 16) + (BN == 2 ? 4 : 1) * NW * 4 + SWV * SSTG_BYTES;
  if (lds > 65536) {
    bool attr_ok = true;                                        // (per instantiation and device)
    MSCLIP_LDS_ATTR((&gemm_stream_kernel<NT, NKC, CONV, SWV, BN>), lds, attr_ok);
    (void)attr_ok;
  }
  const int nblk = (d->M + 31) / 32;
  int gx = (wg_per_cu * ncu + chunks - 1) / chunks;
  const int need = (nblk + SWV - 1) / SWV;
  if (gx > need) gx = need;
  if (BN == 1 && gx > d->part_rows / SWV) gx = d->part_rows / SWV;      // (every wave owns a row of `part`)
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((gemm_stream_kernel<NT, NKC, CONV, SWV, BN>), dim3(gx, chunks), dim3(SWV * 64), lds, st, *d);
}

template <int NT, int NKC, bool CONV, int SWV>
void launch_stream(const msclip_gemm_desc* d, hipStream_t st, int ncu, int wg_per_cu) {
  if (d->bn_mode == 1) launch_stream_bn<NT, NKC, CONV, SWV, 1>(d, st, ncu, wg_per_cu);
  else if (d->bn_mode == 2) launch_stream_bn<NT, NKC, CONV, SWV, 2>(d, st, ncu, wg_per_cu);
  else launch_stream_bn<NT, NKC, CONV, SWV, 0>(d, st, ncu, wg_per_cu);
}

template <bool CONV>
bool dispatch_stream(const msclip_gemm_desc* d, hipStream_t st, int ncu) {
  const int n32 = (d->N + 31) / 32;
  const int nkc = d->K / 64;
  if (nkc == 6) {                                               // four taps over 96 channels (a parity class of a stride-2 dgrad)
    if (!CONV || n32 > 3) return false;
    if (n32 == 3) launch_stream<3, 6, CONV, 4>(d, st, ncu, 2);
    else launch_stream<2, 6, CONV, 4>(d, st, ncu, 2);
    return true;
  }
  if (nkc == 7) {                                               // 3x3, 48 input channels: all of W resident, 8 waves
    if (!CONV || n32 > 3) return false;
    if (n32 == 3) launch_stream<3, 7, CONV, 8>(d, st, ncu, 1);
    else launch_stream<2, 7, CONV, 8>(d, st, ncu, 1);
    return true;
  }
  // the resident weights (K*2 + 16 B per row) stay under 40 KiB so that, with the staging blocks, two workgroups fit
  // a CU: up to 192 columns at K = 64, 96 beyond
  int nt;
  if (n32 <= 2) nt = 2;
  else if (nkc == 1 && n32 % 6 == 0) nt = 6;
  else if (n32 % 3 == 0) nt = 3;
  else nt = 2;
  if (nkc == 1) {
    if (nt == 2) launch_stream<2, 1, CONV, 4>(d, st, ncu, 2);
    else if (nt == 3) launch_stream<3, 1, CONV, 4>(d, st, ncu, 2);
    else launch_stream<6, 1, CONV, 4>(d, st, ncu, 2);
  } else if (nkc == 2) {
    if (nt == 2) launch_stream<2, 2, CONV, 4>(d, st, ncu, 2);
    else launch_stream<3, 2, CONV, 4>(d, st, ncu, 2);
  } else {
    if (nt == 2) launch_stream<2, 3, CONV, 4>(d, st, ncu, 2);
    else launch_stream<3, 3, CONV, 4>(d, st, ncu, 2);
  }
  return true;
}

}  // namespace

// Does the problem stream (short K, weights resident in LDS)?  The one predicate both the launch and
// msclip_gemm_variant() use.
bool msclip_gemm_small_eligible(const msclip_gemm_desc* d) {
  if ((d->K % 64) || d->M < 4096) return false;
  if (d->bn_mode) {                                             // train-mode BatchNorm passes: aligned, unscattered launches only
    const int SWVmax = 8;
    if (d->bn_mode < 1 || d->bn_mode > 2 || d->bias || d->alpha != 1.f || (d->N & 7) || d->rpg != 0x7fffffff) return false;
    if (d->bn_mode == 1 && (!d->part || d->part_rows < SWVmax)) return false;
    if (d->bn_mode == 2 && (!d->bn_consts || !d->out2 || d->out_kind != 0 || (d->ldo & 7) || (d->resid_kind != 0 && d->resid_kind != 2) ||
                            (d->resid_kind == 2 && (!d->resid || (d->ldr & 7))) || (d->act != 0 && d->act != 2)))
      return false;
  }
  if (d->resid_kind == 3 || d->resid_kind == 4 || (d->rpg != 0x7fffffff && d->resid_kind && d->resid_kind < 5)) return false;    // (row scatter: plain stores or the ReLU mask)
  if (d->ldw % 8) return false;
  if (d->mode == 0) return d->K <= 192 && !(d->ldx % 8);
  if (d->mode == 1) {
    if (!d->ktab || (d->Cin % 8) || !(d->K <= 192 || d->K == 448 || d->K == 384)) return false;
    return d->K <= 192 || (d->N + 31) / 32 <= 3;               // 3x3 over 48 channels / four taps over 96: at most 96 output channels
  }
  return false;
}

// Takes the launch if the problem streams; returns false otherwise.
bool msclip_gemm_small_try(const msclip_gemm_desc* d, hipStream_t st, int ncu) {
  if (!msclip_gemm_small_eligible(d)) return false;
  return d->mode == 0 ? dispatch_stream<false>(d, st, ncu) : dispatch_stream<true>(d, st, ncu);
}
