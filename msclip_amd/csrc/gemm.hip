// bf16 MFMA GEMM for gfx950 with a gathering X-loader and fused epilogues.
//
//   out[row(m), n] = epilogue( alpha * sum_k X[m, k] * W[n, k] )
//
// X is either a dense row-major [M, K] matrix or an NHWC activation gathered on
// the fly as the implicit-GEMM view of a KHxKW/stride/pad convolution (every
// 16-byte K-chunk of a row is one LDS-DMA source address, padding taps read a
// zero page).  W is the pre-packed [N][ldw] bf16 weight (K % 64 == 0, zero padded).
//
// Structure (measured choices, see DESIGN.md "GEMM"):
//  * persistent workgroups walk tiles t = blockIdx.x, += gridDim.x in an XCD-aware order;
//  * two 64-deep K-slabs in LDS, filled by global_load_lds (16 B/lane, lane-linear image);
//    the image is XOR-swizzled: physical chunk = logical ^ ((row >> 1) & 7), applied on the
//    per-lane SOURCE address and on the ds_read_b128 address (conflict-free, PMC-verified);
//  * the LDS-DMA of slab s+1 (and, on the last slab, of the NEXT tile's first slab) and the
//    fragment reads of k-step kk+1 are issued BETWEEN the MFMAs of k-step kk
//    (sched_group_barrier), so memory instructions hide in MFMA issue gaps instead of
//    forming a serial burst after each barrier;
//  * MFMA operands are swapped (A = W rows, B = X rows): a lane owns 4 consecutive output
//    columns of one row; interior tiles are transposed through the just-consumed LDS slab
//    so every global store / residual load instruction moves full 128-byte lines.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;

struct RowSrc {          // per staged X row (conv mode)
  long long pix;         // element offset of the (ih0, iw0) tap pixel (may be negative)
  int ih0, iw0;
  int ok;
};



// Interior tile (no guards): transpose the wave's TM x TN 32x32 accumulator tiles through LDS so that 8 lanes
// cover one 128-byte line of one output row; bias/QuickGELU are applied in accumulator layout, the residual
// add / ReLU / conversion after the transpose (row-contiguous, full-line residual loads).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_interior(f32x16 (&acc)[TN][TM], const msclip_gemm_desc& a, char* stg,
                                                  int mw0, int nw0, int lane) {
  const int fr = lane & 31, fhi = lane >> 5;
  const int srow = lane >> 3, sch = lane & 7;      // read-back mapping: row i*8 + srow, 16-byte chunk sch
  const float* __restrict__ bias = a.bias;
  float4 bv[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bv[tn][g] = (bias && nw0 + fhi * 4 + tn * 32 + g * 8 < a.N) ? *(const float4*)(bias + nw0 + fhi * 4 + tn * 32 + g * 8)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool direct16 = TN == 2 && a.out_kind == 0 && a.resid_kind == 0;   // bf16 out, nothing to add after the transpose
  char* wr = stg + fr * 128;
  const int wsw = fr & 7;
  const char* rd = stg + srow * 128 + ((sch ^ srow) << 4);      // + i * 1024 for row block i
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int mrow0 = mw0 + tm * 32;
    if (direct16) {                                  // 64 bf16 columns per staged row: TN == 2 only
#pragma unroll
      for (int tn = 0; tn < (TN == 2 ? TN : 0); ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v0 = acc[tn][tm][g * 4 + 0] * a.alpha + bv[tn][g].x;
          float v1 = acc[tn][tm][g * 4 + 1] * a.alpha + bv[tn][g].y;
          float v2 = acc[tn][tm][g * 4 + 2] * a.alpha + bv[tn][g].z;
          float v3 = acc[tn][tm][g * 4 + 3] * a.alpha + bv[tn][g].w;
          if (a.act == 1) {
            v0 = v0 / (1.f + __expf(-1.702f * v0)); v1 = v1 / (1.f + __expf(-1.702f * v1));
            v2 = v2 / (1.f + __expf(-1.702f * v2)); v3 = v3 / (1.f + __expf(-1.702f * v3));
          } else if (a.act == 2) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          uint2 o;
          o.x = pack_bf16x2(v0, v1);
          o.y = pack_bf16x2(v2, v3);
          const int c = tn * 4 + g;                                // 16-byte chunk; this lane owns half fhi of it
          *(uint2*)(wr + ((c ^ wsw) << 4) + fhi * 8) = o;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4 u = *(const uint4*)(rd + i * 1024);
        if (nw0 + sch * 8 < a.N)   // N % 8 == 0 on this path
          *(uint4*)((bf16_t*)a.out + (size_t)(mrow0 + i * 8 + srow) * a.ldo + nw0 + sch * 8) = u;
      }
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = acc[tn][tm][g * 4 + 0] * a.alpha + bv[tn][g].x;
          v.y = acc[tn][tm][g * 4 + 1] * a.alpha + bv[tn][g].y;
          v.z = acc[tn][tm][g * 4 + 2] * a.alpha + bv[tn][g].z;
          v.w = acc[tn][tm][g * 4 + 3] * a.alpha + bv[tn][g].w;
          if (a.act == 1) {
            v.x = v.x / (1.f + __expf(-1.702f * v.x)); v.y = v.y / (1.f + __expf(-1.702f * v.y));
            v.z = v.z / (1.f + __expf(-1.702f * v.z)); v.w = v.w / (1.f + __expf(-1.702f * v.w));
          }
          const int c = g * 2 + fhi;                               // 16-byte chunk = 4 fp32 columns
          *(float4*)(wr + ((c ^ wsw) << 4)) = v;
        }
        float4 rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const size_t row = (size_t)(mrow0 + i * 8 + srow);
          const int n = nw0 + tn * 32 + sch * 4;
          if (n >= a.N) {
            rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          } else if (a.resid_kind == 1) {
            rv[i] = *(const float4*)((const float*)a.resid + row * a.ldr + n);
          } else if (a.resid_kind == 2) {
            const uint2 u = *(const uint2*)((const bf16_t*)a.resid + row * a.ldr + n);
            rv[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
          } else {
            rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 v = *(const float4*)(rd + i * 1024);
          v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
          if (a.act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          const size_t row = (size_t)(mrow0 + i * 8 + srow);
          const int n = nw0 + tn * 32 + sch * 4;
          if (n >= a.N) {
          } else if (a.out_kind == 1) {
            *(float4*)((float*)a.out + row * a.ldo + n) = v;
          } else {
            uint2 o;
            o.x = pack_bf16x2(v.x, v.y);
            o.y = pack_bf16x2(v.z, v.w);
            *(uint2*)((bf16_t*)a.out + row * a.ldo + n) = o;
          }
        }
      }
    }
  }
}

// Edge tiles, ragged N, unaligned leading dimensions, row scatter / table residual: guarded, straight from the
// accumulator layout (lane owns row .. + (lane&31), columns .. + 8g + 4*(lane>>5) + 0..3).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_generic(f32x16 (&acc)[TN][TM], const msclip_gemm_desc& a, bool vec, int mw0,
                                                 int nw0, int lane) {
  const int fr = lane & 31, fhi = lane >> 5;
  const float* __restrict__ bias = a.bias;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = mw0 + tm * 32 + fr;
    if (m >= a.M) continue;
    const int grp = m / a.rpg;
    const size_t orow = (size_t)(m + grp * a.radd + a.roff);
    size_t rrow = (size_t)m;
    if (a.resid_kind == 3) rrow = (size_t)(m - grp * a.rpg + a.roff);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw0 + tn * 32 + g * 8 + fhi * 4;
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
          } else if (a.resid_kind == 6) {        // accumulate: + the bf16 value already at the STORE row (in-place out += v)
            const uint2 rv = *(const uint2*)((const bf16_t*)a.resid + orow * a.ldr + n);
            v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
            v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
          } else if (a.resid_kind == 5) {        // ReLU backward: keep v where the saved activation (bf16, at the STORE row) is > 0
            const uint2 rv = *(const uint2*)((const bf16_t*)a.resid + orow * a.ldr + n);
            v[0] = __uint_as_float(rv.x << 16) > 0.f ? v[0] : 0.f; v[1] = __uint_as_float(rv.x & 0xffff0000u) > 0.f ? v[1] : 0.f;
            v[2] = __uint_as_float(rv.y << 16) > 0.f ? v[2] : 0.f; v[3] = __uint_as_float(rv.y & 0xffff0000u) > 0.f ? v[3] : 0.f;
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

// Tile configuration: BM x BN output tile, WM x WN waves, each wave owns TM x TN 32x32 MFMA tiles.
//   big  : 256 x 256, 8 waves (2 x 4), 128 KiB LDS, one workgroup per CU  (transformer projections)
//   small: 128 x 128, 4 waves (2 x 2),  64 KiB LDS, two workgroups per CU (narrow convolutions, tiny heads)
template <int MODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_kernel(const msclip_gemm_desc a_in) {
  // Split-K launches (msclip_gemm_splitk): blockIdx.y = K slice; slice s contracts columns [s*K/S, (s+1)*K/S) of both
  // operands into its own fp32 output matrix out[s][M][ldo] (the caller folds the S partials in a fixed order).
  msclip_gemm_desc a = a_in;
  if (MODE == 0 && gridDim.y > 1) {
    const int kc = a.K / (int)gridDim.y;
    a.X = (const bf16_t*)a.X + (size_t)blockIdx.y * kc;
    a.W = (const bf16_t*)a.W + (size_t)blockIdx.y * kc;
    a.out = (float*)a.out + (size_t)blockIdx.y * a.M * a.ldo;
    a.K = kc;
  }
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int XI = BM * 8 / NT, WI = BN * 8 / NT;   // LDS-DMA pieces per lane per K-slab
  constexpr int NPC = XI + WI;                        // pieces per lane per K-slab, spread over the 4 k-steps
  static_assert(XI >= 1 && WI >= 1 && (NW % 2) == 0 && NPC >= 4 && (TN == 2 || TN == 3), "tile config");
  static_assert((BM + BN) * BK * 2 >= NW * STG_BYTES, "staging region per wave");
  __shared__ __attribute__((aligned(1024))) bf16_t smem[2][(BM + BN) * BK];  // [buf][X rows | W rows][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int nt_n = (a.N + BN - 1) / BN;
  const int nt_m = (a.M + BM - 1) / BM;
  const int ntiles = nt_n * nt_m;

  const bf16_t* __restrict__ X = (const bf16_t*)a.X;
  const bf16_t* __restrict__ W = (const bf16_t*)a.W;
  const bf16_t* __restrict__ Z = (const bf16_t*)a.zero;

  // ---- loader state: lane owns (row = (i*NW+wave)*8 + lane/8, physical chunk = lane%8)
  const int pc = lane & 7;
  const int lc = pc ^ ((lane >> 4) | ((wave & 1) << 2));  // logical chunk; == pc ^ ((row>>1)&7)
  const int rsub = lane >> 3;

  const bf16_t* xrow[XI];
  RowSrc xr[XI];
  const bf16_t* wrow[WI];

  auto tile_origin = [&](int t, int& m0, int& n0) {
    // XCD-aware bijective remap: tiles with consecutive ids run on one XCD (they share X rows in its L2)
    const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
    const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
    m0 = (id / nt_n) * BM;
    n0 = (id % nt_n) * BN;
  };

  auto setup_rows = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int n = n0 + (i * NW + wave) * 8 + rsub;
      wrow[i] = (n < a.N) ? W + (size_t)n * a.ldw + lc * 8 : nullptr;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int m = m0 + (i * NW + wave) * 8 + rsub;
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
  };

  // source addresses of one K-slab for this lane (e = chunk-table entry of the slab in conv mode)
  auto slab_sources = [&](int kt, int e, const bf16_t* (&src)[XI + WI]) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if (MODE == 0) {
        src[i] = xrow[i] ? xrow[i] + kt * BK : Z;
      } else {
        const int kh = (e >> 20) & 15, kw = (e >> 24) & 15;
        const int ih = xr[i].ih0 + kh, iw = xr[i].iw0 + kw;
        const bool ok = xr[i].ok && e >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.Wd;
        src[i] = ok ? X + (xr[i].pix + (e & 0xFFFFF)) : Z;
      }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) src[XI + i] = wrow[i] ? wrow[i] + kt * BK : Z;
  };
  auto piece_dst = [&](int buf, int j) -> bf16_t* {   // LDS destination (wave-uniform) of piece j
    return j < XI ? &smem[buf][(j * NW + wave) * 8 * BK] : &smem[buf][BM * BK + ((j - XI) * NW + wave) * 8 * BK];
  };

  // ---- fragment read state
  const int wm = (wave / WN) * (TM * 32), wn = (wave % WN) * (TN * 32);
  const int fr = lane & 31;
  const int fsw = (lane >> 1) & 7;
  const int fhi = lane >> 5;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 3);
  const bool plain_rows = a.rpg == 0x7fffffff && a.resid_kind != 3 && a.resid_kind < 5;
  const int nk = a.K / BK;

  int t = blockIdx.x;
  int m0 = 0, n0 = 0;
  int e_next = 0;   // conv: chunk-table entry of the slab that will be ISSUED during the coming iteration
  if (t < ntiles) {
    tile_origin(t, m0, n0);
    setup_rows(m0, n0);
    const bf16_t* src[XI + WI];
    slab_sources(0, MODE == 1 ? a.ktab[lc] : 0, src);
#pragma unroll
    for (int j = 0; j < XI + WI; ++j) glds16(src[j], piece_dst(0, j));
    if (MODE == 1) e_next = a.ktab[(nk > 1 ? 8 : 0) + lc];
  }
  int it = 0;         // running K-slab counter: slab `it` lives in buffer it & 1
  bool landed = false;  // the first slab of this tile was already waited for (and published) by the previous epilogue
  for (; t < ntiles; t += gridDim.x) {
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int cm0 = m0, cn0 = n0;
    const bool has_next = t + (int)gridDim.x < ntiles;
    for (int kt = 0; kt < nk; ++kt, ++it) {
      if (kt == 0 && landed) {
        // slab already published by the epilogue's barrier; this one only orders the staging reads of all waves
        // before the LDS-DMA into that buffer -- raw barrier, so this tile's stores keep draining in the background
        __builtin_amdgcn_s_barrier();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      // ---- what to issue during this slab's MFMAs: slab kt+1 of this tile, or slab 0 of the next tile
      const bf16_t* src[XI + WI];
      const bool last = kt + 1 == nk;
      if (last && has_next) {
        tile_origin(t + gridDim.x, m0, n0);
        setup_rows(m0, n0);
      }
      if (!last || has_next) {
        slab_sources(last ? 0 : kt + 1, e_next, src);
      } else {
#pragma unroll
        for (int j = 0; j < XI + WI; ++j) src[j] = Z;   // tail of the tile list: harmless dummy pieces
      }
      if (MODE == 1) {   // chunk-table entry for the slab after that one (used next iteration)
        int kn = last ? 1 : kt + 2;
        if (kn >= nk) kn -= nk;
        e_next = a.ktab[kn * 8 + lc];
      }
      const int nb = (it + 1) & 1;
      const bf16_t* xs = smem[it & 1] + (wm + fr) * BK;
      const bf16_t* ws = smem[it & 1] + BM * BK + (wn + fr) * BK;
      bf16x8 wf[2][TN], xf[2][TM];
      {
        const int ph = (fhi ^ fsw) * 8;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[0][i] = *(const bf16x8*)(ws + i * 32 * BK + ph);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[0][j] = *(const bf16x8*)(xs + j * 32 * BK + ph);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
          const int ph = (((kk + 1) * 2 + fhi) ^ fsw) * 8;
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[(kk + 1) & 1][i] = *(const bf16x8*)(ws + i * 32 * BK + ph);
#pragma unroll
          for (int j = 0; j < TM; ++j) xf[(kk + 1) & 1][j] = *(const bf16x8*)(xs + j * 32 * BK + ph);
        }
        constexpr int NV0 = NPC / 4, NVR = NPC % 4;  // k-step kk issues NV0 (+1 while kk < NVR) pieces
        const int nvk = NV0 + (kk < NVR ? 1 : 0);
        const int pv0 = kk * NV0 + (kk < NVR ? kk : NVR);
#pragma unroll
        for (int v = 0; v < NV0 + 1; ++v)
          if (v < nvk) glds16(src[pv0 + v], piece_dst(nb, pv0 + v));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], xf[kk & 1][j], acc[i][j], 0, 0, 0);
          }
        // interleave: one memory instruction behind each MFMA (fragment reads first, then LDS-DMA pieces)
#pragma unroll
        for (int q = 0; q < TM * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
          if (kk < 3 && q < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // DS read
          if (q >= TM * TN - nvk) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- the prefetched slab (next tile) lands under the epilogue math; wait for it BEFORE the stores are issued
    //      so that the next tile's first barrier does not have to drain this tile's stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                  // also: every wave is done reading slab (it-1)&1 -> it becomes the staging area
    landed = true;

    // ---- epilogue
    if (vec && plain_rows && cm0 + BM <= a.M && (cn0 + BN <= a.N || !(a.N & 7)))
      epilogue_interior<TM, TN>(acc, a, (char*)smem[(it + 1) & 1] + wave * STG_BYTES, cm0 + wm, cn0 + wn, lane);
    else
      epilogue_generic<TM, TN>(acc, a, vec, cm0 + wm, cn0 + wn, lane);
  }
}


// ------------------------------------------------------------------------------------------------------------
// Ping-pong kernel for the dense projections: 256 x 256 tile, 8 waves (2 x 4), K-tiles of 64.  LDS is a ring of
// ten 16-KiB regions; a region holds one half-operand of one K-tile (128 rows x 128 B, full cache lines):
// W rows 0-127, W rows 128-255, X rows 0-127, X rows 128-255, in that order (2 LDS-DMA pieces per wave and region,
// buffer-addressed: SGPR descriptor + 32-bit lane offset + scalar K offset).  A K-tile is computed in 2 phases (one
// 64-row half of the wave's 128 x 64 block over the whole K-tile each):
//     phase A:  [ ds_reads: both W fragment sets + X rows 0-63 (16) ; 4 DMA pieces ]
//     phase B:  [ ds_reads: X rows 64-127 (8) ; 4 DMA pieces ; s_waitcnt vmcnt(8) ]
//     each followed by   s_barrier ; lgkmcnt(0) ; 32 MFMAs at raised priority ; s_barrier
// Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its memory section while the other
// owns the MFMA pipe, and a barrier resolves under the tail of the other wave's last MFMA.  (The first version split
// the K-tile into four quadrant phases: twice the barrier hand-overs for the same work; two phases are +1.7 % on the
// step.  -DPP_4PHASE rebuilds it.)
// The wait of phase B leaves the 8 pieces of this K-tile's issues in flight: the regions of the NEXT K-tile (issued
// one K-tile earlier) have landed, and they are first read after the barrier that follows every wave's wait.  Region
// r+10 overwrites the ring slot of region r: phase A of K-tile t re-issues the slots of K-tile t-1's X regions (last
// read in its phase B), phase B those of K-tile t's W regions (last read in phase A).  The last reader's ds_reads
// were issued before the barrier the issuing wave has just passed (the other group waits for them right behind that
// barrier), and the wave issues its own fragment reads before its DMA pieces, whose data is a memory round trip
// away.  At a tile's first K-tile the phase-A pieces move to phase B: their slots are the epilogue's staging until
// the first barrier of the tile, which every wave reaches only after its epilogue.  The DMA stream runs across tile
// boundaries (eight regions of the next tile fly under the epilogue).  Rows past M / N are out of range of the
// tile's buffer descriptor (read as zero).
// ------------------------------------------------------------------------------------------------------------
constexpr int PSLOTS = 10, PREG = 128 * 64;   // ring regions, bf16 elements per region
#ifndef PP_NOEPI
#define PP_NOEPI 0
#endif

__device__ __forceinline__ bf16x8 pp_ld(const void* p) { return *(const bf16x8*)p; }

// one 16 x 16 accumulator tile += W fragment x X fragment.  bf16: k-step ks of the K-tile (16x16x32); fp8: the two 16-byte
// halves of a fragment form ONE 32-byte e4m3 operand of the MX instruction (block scales 2^0 = E8M0 127), ks is always 0.
typedef __attribute__((ext_vector_type(8))) int i32x8;
template <bool F8>
__device__ __forceinline__ f32x4 pp_mma(const bf16x8 (&w)[2], const bf16x8 (&x)[2], int ks, f32x4 c) {
  if constexpr (F8) {
    i32x8 a, b;
    __builtin_memcpy(&a, &w[0], 32);
    __builtin_memcpy(&b, &x[0], 32);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], x[ks], c, 0, 0, 0);
  }
}

// MODE 1: implicit GEMM of a convolution whose input-channel count is a multiple of 64 (a K-tile never straddles a
// filter tap).  X rows are output pixels: the lane keeps the byte offset of its four pixels' window origin and their
// (ih0, iw0); a K-tile adds the tap's offset (chunk table entry 8*kti, fetched a phase ahead as a scalar load) and taps
// outside the image -- or rows past M -- get an offset beyond the descriptor's range (read as zero).
// F8 (msclip_gemm_f8, dense operands only): X and W are OCP e4m3 bytes, a K-tile is 128 elements -- the same 128-byte LDS
// rows, regions, ring and LDS-DMA stream -- contracted by ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 x 16 tile and K-tile
// (twice the bf16 rate; block scales 2^0: the operands carry per-row scales instead, row_scale[m] for X rows and col_scale[n]
// for W rows, fp32, applied to the accumulators in front of the epilogue).
// EPI (dense bf16 only): the LayerNorm fold's two epilogue kinds are instantiations of their own, so that the standard kernel's
// register allocation stays what it was (with all forms in one kernel hipcc spilled inside the K loop): 1 = consumer (the
// projection behind a folded LayerNorm: per-row rstd / mean, per-column weight sums, two row segments), 2 = producer
// (out_proj / c_proj: residual update + bf16 centred copy + per-row partial statistics).  3 = the training step's dgrad behind
// QuickGELU (resid_kind 4: out = bf16(acc * QuickGELU'(resid)), optionally + the stored values' column sums per 128 rows).
// TNL (dense bf16, msclip_gemm_splitk_tn: the weight gradients dW = dY^T X of the training step): both operands are TOKEN-major,
// X [K, ldx] holds the output rows m as COLUMNS and W [K, ldw] the output columns n as columns; the contraction runs over rows.
// A region is 64 token rows x 128 channels (256-byte row segments: full lines), four 1-KiB DMA pieces of 4 rows per wave, and a
// fragment (16 channels x 32 tokens) is built by ds_read_b64_tr_b16 -- the LDS transpose read: 16 lanes read a 4-token x
// 16-channel block, lane i receives channel i's 4 tokens -- two reads per 16 x 16 x 32 operand instead of one ds_read_b128.
// 32-byte channel pairs of a row are XOR-swizzled by (row & 3) | (row >> 3 & 1) << 2: the 8 token rows a half-wave reads in one
// instruction then cover all 64 banks.  Ring, phases, barriers, MFMA order and epilogues are the NT kernel's.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
__device__ __forceinline__ bf16x8 pp_ld_tr(const char* p) {         // tokens k0 .. k0 + 7 of this lane's channel
  const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)p);
  const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)(p + 1024));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MODE, bool F8 = false, int EPI = 0, bool TNL = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const msclip_gemm_desc a_in, const float* __restrict__ row_scale,
                                                      const float* __restrict__ col_scale) {
  static_assert(!(F8 && MODE == 1), "fp8 operands: dense GEMM only");
  static_assert(EPI == 0 || (MODE == 0 && (!F8 || EPI == 2)), "LayerNorm fold: dense GEMM; fp8 operands as the producer only");
  static_assert(EPI != 3 || !F8, "training dgrad epilogue: bf16 operands");
  static_assert(!TNL || (MODE == 0 && !F8 && EPI == 0), "token-major operands: dense bf16 GEMM, plain epilogues");
  // Split-K launches (msclip_gemm_splitk with tile = 4; the weight gradients of the training step): blockIdx.y = K slice;
  // slice s contracts columns [s*K/S, (s+1)*K/S) of both operands into its own fp32 matrix out[s][M][ldo].  A weight
  // gradient is 9-36 output tiles over a 65 024-deep contraction: tiles x slices workgroups of ONE launch fill the chip.
  msclip_gemm_desc a = a_in;
  if (a.M_dev) a.M = min(a.M, *a.M_dev);           // device-side row count (packed captions): a_in.M is the launch's upper bound
  unsigned kbase = 0;                              // TNL: first token row of this K slice
  const int ktot = a.K;                            // TNL: rows of the operands (tokens beyond it read as zero)
  if (TNL) {
    const int kc = ((a.K + 63) / 64 + (int)gridDim.y - 1) / (int)gridDim.y * 64;      // whole K-tiles per slice
    kbase = blockIdx.y * (unsigned)kc;
    a.out = (float*)a.out + (size_t)blockIdx.y * a.M * a.ldo;
    a.K = kc;
  } else
  if (MODE == 0 && !F8 && gridDim.y > 1) {
    const int kc = a.K / (int)gridDim.y;
    a.X = (const bf16_t*)a.X + (size_t)blockIdx.y * kc;
    a.W = (const bf16_t*)a.W + (size_t)blockIdx.y * kc;
    a.out = (float*)a.out + (size_t)blockIdx.y * a.M * a.ldo;
    a.K = kc;
  }
  constexpr unsigned ES = F8 ? 1u : 2u;            // operand element size in bytes
  constexpr int KT = F8 ? 128 : 64;                // elements per K-tile (128 bytes)
  constexpr int TM = 4, TN = 2;
  constexpr bool TE = MODE == 0 && !F8 && EPI == 0;              // training-step forward form (out2) compiled in
  __shared__ __attribute__((aligned(1024))) bf16_t smem[PSLOTS * PREG];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // 0: leading group, 1: one barrier behind
  const int nt_n = (a.N + 255) / 256;
  const int nt_m = (a.M + 255) / 256;
  const int ntiles = nt_n * nt_m;
  const int nk = a.K / KT;

  // Tile id -> origin runs twice per tile in every wave (issue side and compute side).  Its two divisions by launch
  // constants go through multiply-high with reciprocals made once here (exact while id * divisor < 2^32; the host
  // keeps ntiles below 2^15): scalar ALU work instead of two ~40-instruction VALU division sequences, which cost
  // ~2 k of a 54 k-cycle QKV tile in the phase trace.
  constexpr int CG = 4;
  const unsigned tper = (unsigned)(nt_m * CG);
  const unsigned tper_rcp = 0xffffffffu / tper + 1u;
  const int wg_tail = nt_n - (nt_n - 1) / CG * CG;                 // width of the last column group (1..CG)
  const unsigned wgt_rcp = 0xffffffffu / (unsigned)wg_tail + 1u;   // unused when wg_tail == 1
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
    const unsigned id = (unsigned)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3));
    // column groups of 4 tiles, rows fastest inside a group: the ~32 tiles an XCD runs together then cover 8 row blocks
    // x 4 column slices (12 distinct operand slices in its L2) instead of 2.7 x 12 (14.7); +1 % on the step
    const unsigned g = (unsigned)(((unsigned long long)id * tper_rcp) >> 32);
    const unsigned idg = id - g * tper;
    const bool tail = (int)(g * CG + CG) > nt_n;
    const unsigned wg = tail ? (unsigned)wg_tail : (unsigned)CG;
    const unsigned row = !tail ? idg >> 2 : wg_tail == 1 ? idg : (unsigned)(((unsigned long long)idg * wgt_rcp) >> 32);
    m0 = (int)(row * 256u);
    n0 = (int)((g * CG + (idg - row * wg)) * 256u);
  };

  // ---- issue side.  Piece p (rows 8p .. 8p+7 of a region) is issued by wave p & 7; lane -> row 8p + lane/8,
  // physical 16-byte chunk lane%8 holds logical chunk (lane%8) ^ ((row >> 1) & 7).
  unsigned vx[2], vw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (TNL) {                                     // piece = 4 token rows x 256 B; lane -> row 4p + lane / 16, physical 16-byte chunk lane % 16
      const int row = (wave + 8 * i) * 4 + (lane >> 4);
      const int g = (row & 3) | (((row >> 3) & 1) << 2);
      const unsigned ch = (unsigned)(((((lane & 15) >> 1) ^ g) << 1) | (lane & 1)) << 4;
      vx[i] = (unsigned)row * (unsigned)a.ldx * 2u + ch;
      vw[i] = (unsigned)row * (unsigned)a.ldw * 2u + ch;
      continue;
    }
    const int row = (wave + 8 * i) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) << 4;
    vx[i] = (unsigned)row * (unsigned)a.ldx * ES + ch;
    vw[i] = (unsigned)row * (unsigned)a.ldw * ES + ch;
  }
  const unsigned xhalf = TNL ? 256u : 128u * (unsigned)a.ldx * ES, whalf = TNL ? 256u : 128u * (unsigned)a.ldw * ES;
  int ti = blockIdx.x, kti = 0, islot = 0;
  __amdgpu_buffer_rsrc_t rx, rw;
  // conv mode: window origin of this lane's pixel rows [half][piece]: byte offset + 16-byte chunk, and (ih0, iw0)
  int cpix[2][2], chw[2][2];                       // (ih0 in the low half, iw0 in the high half)
  int ce = 0;                                      // chunk-table entry of the K-tile whose X regions are issued next
  auto set_tile = [&](int t) {
    if (MODE == 1) {
      int m0 = 0, n0 = 0;
      if (t < ntiles) tile_origin(t, m0, n0);
      const int hw = a.Ho * a.Wo;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (wave + 8 * i) * 8 + (lane >> 3);
          const int m = m0 + h * 128 + row;
          const int b = m / hw, rem = m - b * hw;
          const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
          const int ih0 = ho * a.stride - a.pad, iw0 = wo * a.stride - a.pad;
          const int ihs = (t < ntiles && m < a.M) ? ih0 : -(1 << 14);     // rows past M: every tap is "outside"
          chw[h][i] = (ihs & 0xffff) | (iw0 << 16);
          cpix[h][i] = (((b * a.H + ih0) * a.Wd + iw0) * a.Cin) * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
      const unsigned long long wb = (unsigned long long)(a.N - n0) * (unsigned long long)a.ldw * ES;
      rw = t < ntiles ? make_rsrc((const char*)a.W + (size_t)n0 * a.ldw * ES, wb > 0xffffffffull ? 0xffffffffu : (unsigned)wb)
                      : make_rsrc(a.W, 0);
      return;
    }
    if (TNL) {
      int m0 = 0, n0 = 0;
      if (t < ntiles) tile_origin(t, m0, n0);
      const unsigned long long xb = (unsigned long long)ktot * a.ldx * 2ull - (unsigned long long)m0 * 2ull;
      const unsigned long long wb = (unsigned long long)ktot * a.ldw * 2ull - (unsigned long long)n0 * 2ull;
      rx = t < ntiles ? make_rsrc((const char*)a.X + (size_t)m0 * 2, (unsigned)xb) : make_rsrc(a.X, 0);
      rw = t < ntiles ? make_rsrc((const char*)a.W + (size_t)n0 * 2, (unsigned)wb) : make_rsrc(a.W, 0);
      return;
    }
    if (t < ntiles) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const unsigned long long xb = (unsigned long long)(a.M - m0) * (unsigned long long)a.ldx * ES;
      const unsigned long long wb = (unsigned long long)(a.N - n0) * (unsigned long long)a.ldw * ES;
      rx = make_rsrc((const char*)a.X + (size_t)m0 * a.ldx * ES, xb > 0xffffffffull ? 0xffffffffu : (unsigned)xb);
      const void* wseg = (EPI == 1 && a.W2 && m0 >= a.seg_split) ? a.W2 : a.W;   // second row segment: the other modality's folded weight
      rw = make_rsrc((const char*)wseg + (size_t)n0 * a.ldw * ES, wb > 0xffffffffull ? 0xffffffffu : (unsigned)wb);
    } else {                                       // past the tile list: empty descriptors, the counts stay exact
      rx = make_rsrc(a.X, 0);
      rw = make_rsrc(a.W, 0);
    }
  };
  auto issue = [&](auto jc) {                      // region j of K-tile (ti, kti) -> ring slot islot
    constexpr int J = decltype(jc)::value;
    bf16_t* dst = smem + islot * PREG + wave * 512;
    const unsigned ko = (unsigned)kti * 128u;
    if (TNL) {                                     // K-tile = 64 token rows down the matrix, region half = 128 columns to the right
      // (the row offset goes into the LANE offset: the descriptor's range check -- token rows past the matrix read as zero,
      //  which a ragged last K-tile / an over-hanging last slice rely on -- covers the lane offset, not the scalar one)
      const unsigned kr = kbase + (unsigned)kti * 64u;
      if (J < 2) {
        const unsigned ro = kr * (unsigned)a.ldw * 2u;
        blds16(rw, vw[0] + ro, (J & 1) * 256u, dst);
        blds16(rw, vw[1] + ro, (J & 1) * 256u, dst + 8 * 512);
      } else {
        const unsigned ro = kr * (unsigned)a.ldx * 2u;
        blds16(rx, vx[0] + ro, (J & 1) * 256u, dst);
        blds16(rx, vx[1] + ro, (J & 1) * 256u, dst + 8 * 512);
      }
    } else
    if (J < 2) {
      blds16(rw, vw[0], ko + (J & 1) * whalf, dst);
      blds16(rw, vw[1], ko + (J & 1) * whalf, dst + 8 * 512);
    } else if (MODE == 1) {
      const int kh = (ce >> 20) & 15, kw = (ce >> 24) & 15, doff = (ce & 0xfffff) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ih = ((chw[J & 1][i] << 16) >> 16) + kh, iw = (chw[J & 1][i] >> 16) + kw;
        const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.Wd;
        blds16(rx, ok ? (unsigned)(cpix[J & 1][i] + doff) : 0x80000000u, 0, dst + i * 8 * 512);
      }
    } else {
      blds16(rx, vx[0], ko + (J & 1) * xhalf, dst);
      blds16(rx, vx[1], ko + (J & 1) * xhalf, dst + 8 * 512);
    }
    if (MODE == 1 && J == 1) {
      // entry for the X regions of this K-tile, issued two phases on.  A scalar load in inline asm (with its own wait,
      // so the value is valid wherever the compiler keeps or moves it): as a tracked vector load it would sit in the
      // VM counter's in-order queue and its use would drain every DMA piece in flight.
      const int* tp = a.ktab + kti * 8;
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ce) : "s"(tp));
    }
    islot = islot == PSLOTS - 1 ? 0 : islot + 1;
    if (J == 3) {
      if (++kti == nk) {
        kti = 0;
        ti += gridDim.x;
        set_tile(ti);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- compute side
  const int wm = grp * 128, wn = (wave & 3) * 64;
  // v_mfma_f32_16x16x32_bf16 (on random operands it sustains 2.03 PF against 1.73-1.82 PF for 32x32x16 at the same
  // power limit, tools/probes/mfma_probe): a fragment is 16 rows x 32 K, lane l holds row l % 16, 16-byte chunk l / 16
  // (+ 4 for the second k-step); la[ks] = byte offset of that chunk in a 128-byte LDS row (same swizzle as the image)
  const int r16 = lane & 15, quad = lane >> 4;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 3);
  const bool plain_rows = a.rpg == 0x7fffffff && a.resid_kind != 3 && a.resid_kind < 5;
  // bf16: la[ks] = k-step ks (16-byte chunk ks*4 + quad of the 128-byte row); fp8: the lane's 32 bytes of the one k-step are
  // chunks 2*quad (la[0]) and 2*quad + 1 (la[1]) -- A and B use the same lane -> k assignment, so the contraction is exact
  int la[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    la[ks] = r16 * 128 + (((F8 ? (quad << 1) | ks : (ks << 2) | quad) ^ ((r16 >> 1) & 7)) << 4);
  const int wsub = ((wave >> 1) & 1);              // W half-region of this wave
  const int woff = (wave & 1) * 64 * 128;          // byte offset of its 64 rows inside that region
  // TNL: lane l of a 16-lane group reads the 8 bytes at (token row 8 * quad + (l >> 2) [+ 4 for the second read, + 32 for the
  // second k-step], channels 4 * (l & 3) .. + 3 of a 16-channel block); block cb sits at 32-byte pair cb ^ g(row) of the row
  const int lt = (8 * quad + (r16 >> 2)) * 256 + ((((r16 >> 2) & 3) | ((quad & 1) << 2)) << 5) + (r16 & 3) * 8;
  const int ltw = lt ^ ((wave & 1) * 128);         // the wave's 64 W channels: pairs 4 * (wave & 1) + 0..3
  const char* lds = (const char*)smem;

  if (MODE == 1)                                   // one descriptor over the whole NHWC input (< 2 GiB, checked by the host)
    rx = make_rsrc(a.X, (unsigned)((long long)(a.M / (a.Ho * a.Wo)) * a.H * a.Wd * a.Cin * 2));
  set_tile(ti);
  issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
  issue(I0{}); issue(I1{}); issue(I2{});
  issue(I3{});
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp) __builtin_amdgcn_s_barrier();           // stagger

  int cslot = 0;                                   // ring slot of region 0 of the K-tile being computed
  bf16x8 w0[2][2], w1[2][2], xf[4][2];             // [16-row tile][k-step]: W rows 0-31 / 32-63 of the wave, X rows of a sub-block
  int epi_stores = 0;                              // stores this wave is known to have issued in the previous tile's epilogue
  for (int tc = blockIdx.x; tc < ntiles; tc += gridDim.x) {
    int cm0, cn0;
    tile_origin(tc, cm0, cn0);
    f32x4 acc[4][8];                               // [16-column tile][16-row tile] of the wave's 128 x 64 block
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // Bias of this tile: requested here, first touched after the first vmcnt wait of the K loop.  The loads are
    // unconditional (clamped index, zero page when there is no bias) and the lane id is recomputed: a predicated load
    // or a reloaded spill at this point would bring an s_waitcnt vmcnt(0), i.e. a wait for the previous tile's stores.
    int lane_s;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
    const bool seg2 = EPI == 1 && a.W2 && cm0 >= a.seg_split;
    const float* bseg = seg2 ? a.bias2 : a.bias;
    const float* bsrc = bseg ? bseg : (const float*)a.zero;
    const int nlast = bseg ? a.N - 1 : 0;
    // bias4[i]: this lane's 4 epilogue columns of 32-column block i (see epilogue_rows).
    // The loads are inline asm: the compiler must not know they are pending (it would wait for them -- and so for the
    // previous tile's stores, the VM counter retires in order -- before entering the K loop).  They are older than the
    // eight DMA pieces the first vmcnt(8) of the K loop leaves in flight, and first read in the epilogue.
    float bcol;                                    // packed bf16 epilogue: lane j holds the bias of the wave's column j
    {
      const int n = cn0 + wn + lane_s;
      const float* p = bsrc + (n < nlast ? n : nlast);
      asm volatile("global_load_dword %0, %1, off" : "=&v"(bcol) : "v"(p));
    }
    // LayerNorm fold: the consumer asks for the column sums of its segment's weight like the bias (lane j: column j) and for its
    // wave's 128 rows' (rstd, mean * rstd) -- lane j holds rows j (.x, .y) and j + 64 (.z, .w); the producer for its rows'
    // centres (.x, .z).  M % 256 == 0 on these paths (host-checked): no clamping.
    float ccol = 0.f;
    f32x4 rstat4 = {1.f, 0.f, 1.f, 0.f};
    if constexpr (EPI == 1) {
      const float* p = (seg2 ? a.csum2 : a.csum) + cn0 + wn + lane_s;
      asm volatile("global_load_dword %0, %1, off" : "=&v"(ccol) : "v"(p));
      const float* q0 = a.rowstat + 2 * (size_t)(cm0 + wm + lane_s);
      f32x2 lo, hi;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(lo) : "v"(q0));
      asm volatile("global_load_dwordx2 %0, %1, off offset:512" : "=&v"(hi) : "v"(q0));
      rstat4 = f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else if constexpr (EPI == 2) {
      const float* q0 = a.center + cm0 + wm + lane_s;
      float lo, hi;
      asm volatile("global_load_dword %0, %1, off" : "=&v"(lo) : "v"(q0));
      asm volatile("global_load_dword %0, %1, off offset:256" : "=&v"(hi) : "v"(q0));
      rstat4 = f32x4{lo, 0.f, hi, 0.f};
    }

    float pf_sink = 0.f;                           // destination of the next tile's statistics prefetch (kept allocated until it has landed)
    for (int kt = 0; kt < nk; ++kt) {
      int s1 = cslot + wsub, s2 = cslot + 2 + grp;
      if (s1 >= PSLOTS) s1 -= PSLOTS;
      if (s2 >= PSLOTS) s2 -= PSLOTS;
      const char* wreg = lds + s1 * (PREG * 2) + woff;
      const char* xreg = lds + s2 * (PREG * 2);
      cslot = cslot + 4 >= PSLOTS ? cslot + 4 - PSLOTS : cslot + 4;

#define PP_PRIO(x) __builtin_amdgcn_s_setprio(x)
#define PP_LGK() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_SYNC_IN()                                   \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  PP_LGK();                                            \
  __builtin_amdgcn_sched_barrier(0);                   \
  PP_PRIO(1)
#define PP_SYNC_OUT()                                  \
  PP_PRIO(0);                                          \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  asm volatile("" ::: "memory");                       \
  __builtin_amdgcn_sched_barrier(0)

      // ---- phase 0: W sub 0 (32 rows), X sub 0 (64 rows) -> quadrant (0, 0)
      const size_t wtn = (size_t)(lds + s1 * (PREG * 2) + ltw), xtn = (size_t)(lds + s2 * (PREG * 2) + lt);     // (TNL)
      if constexpr (TNL) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) w0[i][ks] = pp_ld_tr((const char*)(wtn ^ (size_t)(i * 32)) + ks * 8192);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) xf[j][ks] = pp_ld_tr((const char*)(xtn ^ (size_t)(j * 32)) + ks * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) w1[i][ks] = pp_ld_tr((const char*)(wtn ^ (size_t)((2 + i) * 32)) + ks * 8192);
      } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w0[i][ks] = pp_ld(wreg + i * 2048 + la[ks]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[j][ks] = pp_ld(xreg + j * 2048 + la[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w1[i][ks] = pp_ld(wreg + 4096 + i * 2048 + la[ks]);
      }
      if (kt) {                                    // K-tile 0: the slots are still the epilogue's staging until the first barrier
        issue(I0{});
        issue(I1{});
      }
      PP_SYNC_IN();
#pragma unroll
      for (int ks = 0; ks < (F8 ? 1 : 2); ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][j] = pp_mma<F8>(w0[i], xf[j], ks, acc[i][j]);
#pragma unroll
      for (int ks = 0; ks < (F8 ? 1 : 2); ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[2 + i][j] = pp_mma<F8>(w1[i], xf[j], ks, acc[2 + i][j]);
      PP_SYNC_OUT();

      // ---- phase 2: X sub 1 -> quadrant (1, 1)
      if constexpr (TNL) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) xf[j][ks] = pp_ld_tr((const char*)(xtn ^ (size_t)((4 + j) * 32)) + ks * 8192);
      } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[j][ks] = pp_ld(xreg + 8192 + j * 2048 + la[ks]);
      }
      if (!kt) {
        issue(I0{});
        issue(I1{});
      }
      issue(I2{});
      issue(I3{});
      if constexpr (EPI == 1 || EPI == 2) {
        // The row statistics / centres a tile asks for at its start were written by the previous kernel on other XCDs: a miss
        // to the MALL takes longer than the 0.65 us to the first counted wait of the K loop, which retires in order.  So each
        // wave touches the NEXT tile's lines here (lanes 0-7: one 128-byte line each; value discarded), a K-tile away from the
        // wait that covers it; at the next tile's start they are L2 hits like the bias.
        if (kt == 2) {
          int nm0, nn0;
          tile_origin(tc + (int)gridDim.x < ntiles ? tc + (int)gridDim.x : tc, nm0, nn0);
          const float* q = (EPI == 1 ? a.rowstat + 2 * (size_t)(nm0 + wm) : a.center + (nm0 + wm)) + (lane_s & (EPI == 1 ? 7 : 3)) * 32;
          asm volatile("global_load_dword %0, %1, off" : "=&v"(pf_sink) : "v"(q));
        }
      }
      if ((EPI == 1 || EPI == 2) && kt == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else
      if (kt == 0 && nk >= 3 && epi_stores == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (kt == 0 && nk >= 3 && epi_stores == 32) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (EPI == 3 && kt == 0 && nk >= 3 && epi_stores == 34) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
      else if (kt == 0 && nk >= 3 && epi_stores == 52) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      PP_SYNC_IN();
#pragma unroll
      for (int ks = 0; ks < (F8 ? 1 : 2); ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[2 + i][4 + j] = pp_mma<F8>(w1[i], xf[j], ks, acc[2 + i][4 + j]);

      // ---- phase 3: nothing new to read -> quadrant (1, 0); the next K-tile's regions are waited for here
#pragma unroll
      for (int ks = 0; ks < (F8 ? 1 : 2); ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][4 + j] = pp_mma<F8>(w0[i], xf[j], ks, acc[i][4 + j]);
      PP_SYNC_OUT();
    }

    // Both groups run the epilogue together: the leading group waits one barrier, the trailing one re-staggers after.
    // Staging = the ring slots of the last K-tile's X regions (dead since its phase 2; re-issued in phase 1 of the
    // next tile, behind a barrier every wave reaches only after its epilogue).
    if (!grp) __builtin_amdgcn_s_barrier();
    asm volatile("" : "+v"(pf_sink)::"memory");
    {
      int ss = cslot + 8 + (wave >> 2);            // cslot already points 4 ahead: X regions of the last K-tile = cslot - 2, - 1
      while (ss >= PSLOTS) ss -= PSLOTS;
      const unsigned stg = (unsigned)(size_t)(AS3 bf16_t*)smem + ss * (PREG * 2) + (wave & 3) * STG_BYTES;
      // lane id recomputed from scratch: the epilogue's lane constants must not live (spilled) across the main loop
      int lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      epi_stores = 0;
#if PP_NOEPI
      // probe builds only (tools/probes/build_ablations.sh noepi "-DPP_NOEPI=1"): the tile's epilogue is skipped -- no loads, no
      // stores -- with the accumulators kept alive by a store that never happens: what is left of a launch is the main loop +
      // tile transitions, i.e. launch time minus this = the epilogue's share (profiles/r05_gemm_epilogue_share.md)
      {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) sacc += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        if (sacc == 1.2345678e-30f) ((float*)a.zero)[lane_e] = sacc;
      }
#else
      if (F8) {
        // per-row operand scales: acc[ni][mi][r] = C[row mi*16 + lane%16][column ni*16 + 4*(lane/16) + r]
        const int er = lane_e & 15, eq = lane_e >> 4;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int m = cm0 + wm + mi * 16 + er;
          const float sx = row_scale[m < a.M ? m : a.M - 1];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ni][mi][r] *= sx;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = cn0 + wn + ni * 16 + eq * 4 + r;
            const float sw = col_scale[n < a.N ? n : a.N - 1];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi][r] *= sw;
          }
      }
      if constexpr (EPI == 1) {                    // whole tiles, bf16 output, act 0 / 1 (host-checked)
        const float4 rstat = make_float4(rstat4[0], rstat4[1], rstat4[2], rstat4[3]);
        epi_stores = 4 * TM;
        if (a.act == 0)
          epilogue_pack16<TM, TN, 0, true>(acc, a, stg, cm0 + wm, cn0 + wn, lane_e, bcol, a.out, ccol, rstat);   // QKV behind a folded LayerNorm
        else
          epilogue_pack16<TM, TN, 1, true>(acc, a, stg, cm0 + wm, cn0 + wn, lane_e, bcol, a.out, ccol, rstat);   // c_fc + QuickGELU
      } else if constexpr (EPI == 3) {             // whole 256-row tiles, bf16 output (host-checked)
        const bool full_n = cn0 + 256 <= a.N;
        if (full_n) epi_stores = a.part ? 4 * TM * TN + TN : 4 * TM * TN;    // 32 row stores (+ 2 of the column-sum partials) per wave
        epilogue_rows<TM, TN, 4, 0, 0, true>(acc, a, stg, cm0 + wm, cn0 + wn, lane_e, bcol);
      } else if constexpr (EPI == 2) {             // whole tiles, in-place fp32 residual update (host-checked)
        epi_stores = 52;                           // 32 fp32 + 16 bf16 full-line stores + 4 stores of row partials per wave
        // (a tile lies in one row segment: seg_split % 256 == 0; the second segment's residual stream may sit elsewhere)
        const float* rs = a.resid2 && cm0 >= a.seg_split ? (const float*)a.resid2 : (const float*)a.resid;
        epilogue_rows_stats<TM, TN>(acc, a, stg, cm0 + wm, cn0 + wn, lane_e, bcol, rstat4[0], rstat4[2], rs);
      } else
      if (vec && plain_rows && cm0 + 256 <= a.M)
      {
        const bool full_n = cn0 + 256 <= a.N;      // no lane's store is predicated off
        const int mw0 = cm0 + wm, nw0 = cn0 + wn;
        const bool pack16 = a.resid_kind == 0 && a.out_kind == 0 && !((a.N | a.ldo) & 7);
        if (F8 && a.out_kind == 2) {               // e4m3 output with a static scale (c_fc -> the fp8 operand of c_proj)
          if (full_n) epi_stores = 2 * TM;
          if (a.act == 1) epilogue_pack8<TM, TN, 1>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out_scale);
          else epilogue_pack8<TM, TN, 0>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out_scale);
        } else {
        if (full_n) epi_stores = pack16 && a.act <= 2 ? (a.out2 ? 8 * TM : 4 * TM) : 4 * TM * TN;   // 16 / 32 store instructions per wave
        if (TE && pack16 && a.out2)                // training forward of c_fc: the pre-activation first, then the activation
          epilogue_pack16<TM, TN, 0>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out2);
        if (pack16 && a.act == 0)
          epilogue_pack16<TM, TN, 0>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);        // QKV
        else if (pack16 && a.act == 1)
          epilogue_pack16<TM, TN, 1>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);        // c_fc + QuickGELU
        else if (pack16 && a.act == 2)
          epilogue_pack16<TM, TN, 2>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);        // convolution + ReLU
        else if (a.resid_kind == 1 && a.act == 0 && a.out_kind == 1)
          epilogue_rows<TM, TN, 1, 0, 1>(acc, a, stg, mw0, nw0, lane_e, bcol);           // out_proj / c_proj into the fp32 stream
        else
          epilogue_rows<TM, TN, -1, -1, -1>(acc, a, stg, mw0, nw0, lane_e, bcol);        // pointwise convolutions, heads
        }
      }
      else
        epilogue_generic16<TM, TN>(acc, a, vec, cm0 + wm, cn0 + wn, lane_e);
#endif
    }
    if (grp) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing empty pieces must retire before the LDS is released
#undef PP_SYNC_IN
#undef PP_SYNC_OUT
#undef PP_PRIO
#undef PP_LGK
}

}  // namespace

template <int MODE, int BM, int BN, int WM, int WN>
static void launch_cfg(const msclip_gemm_desc* d, hipStream_t st, int blocks_per_cu) {
  const int ncu = msclip_device_cus();
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  const int cap = ncu * blocks_per_cu;
  const int grid = tiles < cap ? tiles : cap;
  hipLaunchKernelGGL((gemm_kernel<MODE, BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), 0, st, *d);
}


bool msclip_gemm_small_try(const msclip_gemm_desc* d, hipStream_t st, int ncu);   // gemm_small.hip
bool msclip_gemm_small_eligible(const msclip_gemm_desc* d);

static int device_cus() { return msclip_device_cus(); }

// ---- kernel choice: ONE function decides, msclip_gemm launches what it says and msclip_gemm_variant reports it
enum GemmVariant { GV_INVALID = 0, GV_STREAM, GV_PP, GV_DENSE128, GV_PPCONV, GV_CONV192, GV_CONV128, GV_DENSE192 };
static const char* const kVariantName[] = {"invalid", "stream", "pp", "dense128", "ppconv", "conv192", "conv128", "dense192"};

static GemmVariant pick_variant(const msclip_gemm_desc* d) {
  if (!d || !d->X || !d->W || !d->out || !d->zero) return GV_INVALID;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % BK)) return GV_INVALID;
  if (d->mode == 0 && (d->ldx % 8)) return GV_INVALID;
  if (d->ldw < d->K || (d->ldw % 8)) return GV_INVALID;
  if (d->mode == 1 && (!d->ktab || (d->Cin % 8))) return GV_INVALID;
  if (d->mode != 0 && d->mode != 1) return GV_INVALID;
  if (d->rpg <= 0) return GV_INVALID;
  if (d->resid_kind < 0 || d->resid_kind > 6 || d->out_kind < 0 || d->out_kind > 1) return GV_INVALID;   // (e4m3 outputs: msclip_gemm_f8)
  // resid_kind 6 (+ the bf16 value at the store row: a scattered launch accumulating into an existing map): generic / streaming kernels
  if (d->resid_kind == 6 && (!d->resid || (d->ldr & 3) || d->tile == 4 || d->out2 || d->rowstat || d->W2 || d->xb || d->out_kind != 0))
    return GV_INVALID;
  // resid_kind 5 (ReLU mask at the store row; the conv side's input gradients): implicit-conv launches of the generic / streaming
  // kernels only
  if (d->resid_kind == 5 && (d->mode != 1 || !d->resid || (d->ldr & 3) || d->tile == 4 || d->out2 || d->rowstat || d->W2 || d->xb))
    return GV_INVALID;
  // train-mode BatchNorm passes (bn_mode 1 / 2): the streaming kernel or nothing (no other kernel carries the epilogues)
  if (d->bn_mode)
    return (d->tile == 0 || d->tile == 5) && !d->rowstat && !d->W2 && !d->xb && !d->M_dev && msclip_gemm_small_eligible(d) ? GV_STREAM
                                                                                                                         : GV_INVALID;
  // training-step epilogue forms (second pre-activation output; multiply by QuickGELU'(resid)): ping-pong kernels only,
  // whole 256-row tiles only (the guarded edge-tile epilogue does not carry them)
  const bool train_epi = d->out2 || d->resid_kind == 4;
  if (train_epi && (d->mode != 0 || d->out_kind != 0 || ((d->N | d->ldo) & 7) || (d->M % 256) || (d->out2 && d->resid_kind) ||
                    (d->resid_kind == 4 && (!d->resid || (d->ldr & 3) || d->act)) || d->rpg != 0x7fffffff))
    return GV_INVALID;
  if (train_epi && d->tile != 4 && d->tile != 0) return GV_INVALID;
  if (d->part && !d->xb && (d->resid_kind != 4 || (d->N & 3))) return GV_INVALID;   // column-sum partials: the resid_kind 4 epilogue only
  // LayerNorm fold (consumer: rowstat / W2; producer: xb): whole 256 x 256 tiles of the dense ping-pong kernel only
  const bool fold_c = d->rowstat || d->W2, fold_p = d->xb != nullptr;
  if (fold_c && (d->mode != 0 || !d->rowstat || !d->csum || d->out_kind != 0 || d->resid_kind || d->act > 1 || d->out2 ||
                 d->alpha != 1.f || (d->M % 256) || (d->N % 256) || (d->ldo % 8) || d->rpg != 0x7fffffff ||
                 (d->W2 && (!d->csum2 || d->seg_split <= 0 || (d->seg_split % 256) || d->seg_split >= d->M)) || fold_p))
    return GV_INVALID;
  if (fold_p && (d->mode != 0 || d->resid_kind != 1 || d->out_kind != 1 || d->act || d->out2 || !d->center || !d->part ||
                 (d->M % 256) || (d->N % 256) || (d->ldxb % 4) || (d->ldo % 4) || (d->ldr % 4) || d->rpg != 0x7fffffff ||
                 (d->resid2 && (d->seg_split <= 0 || (d->seg_split % 256) || d->seg_split >= d->M))))
    return GV_INVALID;
  if (d->resid2 && !fold_p) return GV_INVALID;
  if ((fold_c || fold_p) && d->tile != 4 && d->tile != 0) return GV_INVALID;
  // tile choice: the 256x256 / 8-wave config whenever the problem fills the chip with it, else 128x128
  const long long big_tiles = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
  if (d->tile < 0 || d->tile > 6 || d->tile == 2 || d->tile == 3) return GV_INVALID;   // 2, 3 (and 7, 8 of rounds 2-3): retired kernels
  const bool big = d->tile == 4 || (d->tile == 0 && ((d->N >= 192 && big_tiles >= 128) || train_epi || fold_c || fold_p));
  if ((d->tile == 0 || d->tile == 5) && !train_epi && !fold_c && !fold_p && msclip_gemm_small_eligible(d)) return GV_STREAM;
  // (rounds 2-3 built two kernels that hide the epilogue -- a 4-wave kernel carrying it under the next tile's K loop and two
  //  4-wave workgroups per CU on 256 x 128 tiles; both measured slower than the ping-pong kernel and were retired in round 4:
  //  DESIGN.md "The epilogue problem", "Two workgroups per CU"; sources in the git history up to round 3)
  if (d->mode == 0) {
    // 256-row tile offsets must fit the 32-bit buffer offsets of the ping-pong kernel's loads
    // ... and tile id x (4 row-tile counts) below 2^32 for the kernel's reciprocal-multiply tile mapping
    const bool pp_ok = (long long)d->ldx * 2 * 256 + (long long)d->K * 2 < (1ll << 31) &&
                       (long long)d->ldw * 2 * 256 + (long long)d->K * 2 < (1ll << 31) &&
                       big_tiles * ((d->M + 255) / 256) * 4 < (1ll << 32);
    if (pp_ok && d->resid_kind < 5 && (d->tile == 4 || (d->tile == 0 && big))) return GV_PP;     // ping-pong kernel (default for the projections)
    if (train_epi || fold_c || fold_p) return GV_INVALID;   // (offsets beyond the ping-pong kernel's 32-bit addressing)
    if (d->tile == 6) return GV_DENSE192;      // 256 x 192 two-buffer tiles on dense operands (calibration of the fused QKV + attention kernel's main loop)
    return GV_DENSE128;                        // small problems, heads, logits (and tile 1)
  }
  // input channels a multiple of 64 (a K-tile stays inside one filter tap): the ping-pong kernel gathers the rows itself
  const long long in_bytes = (long long)(d->M / (d->Ho * d->Wo)) * d->H * d->Wd * d->Cin * 2;
  if ((d->tile == 0 || d->tile == 4) && d->Cin % 64 == 0 && d->K % d->Cin == 0 && in_bytes < (1ll << 31) &&
      (long long)d->ldw * 2 * 256 + (long long)d->K * 2 < (1ll << 31) && d->rpg == 0x7fffffff && d->resid_kind != 3 &&
      d->resid_kind < 5 && d->M % (d->Ho * d->Wo) == 0 && (big_tiles >= 128 || d->tile == 4) && big_tiles * ((d->M + 255) / 256) * 4 < (1ll << 32))
    return GV_PPCONV;
  // N a multiple of 192 (192 / 384 / 768 output channels): 256 x 192 tiles leave no idle columns
  const long long t192 = (long long)((d->M + 255) / 256) * ((d->N + 191) / 192);
  if (d->tile == 6 || (d->tile == 0 && d->N % 192 == 0 && t192 >= 128)) return GV_CONV192;
  return GV_CONV128;
}

extern "C" const char* msclip_gemm_variant(const msclip_gemm_desc* d) { return kVariantName[pick_variant(d)]; }

// Split-K form for contractions that are deep and narrow (weight gradients: a few output tiles over 10^5 - 10^7 tokens or
// pixels): `slices` workgroup rows, each contracting K / slices columns into out[slice][M][ldo] (fp32), 128 x 128 tiles.
extern "C" int msclip_gemm_splitk(const msclip_gemm_desc* d, int slices, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_gemm_splitk, stream, d, slices);
  if (!d || slices < 1 || slices > 65535 || pick_variant(d) == GV_INVALID || d->M_dev) return MSCLIP_EINVAL;
  if (d->mode != 0 || d->out_kind != 1 || d->bias || d->resid || d->resid_kind || d->act || d->rpg != 0x7fffffff || d->radd ||
      d->roff || (d->K % (BK * slices)))
    return MSCLIP_EINVAL;
  if (d->tile == 4) {                                    // the ping-pong kernel, one 256 x 256 tile list per slice
    if (pick_variant(d) != GV_PP) return MSCLIP_EINVAL;
    const int t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    const int ncu = device_cus();
    hipLaunchKernelGGL((gemm_pp_kernel<0, false>), dim3(t256 < ncu ? t256 : ncu, slices), dim3(512), 0, (hipStream_t)stream, *d,
                       nullptr, nullptr);
    return msclip_launch_status();
  }
  const int tiles = ((d->M + 127) / 128) * ((d->N + 127) / 128);
  const int cap = device_cus() * 2;
  hipLaunchKernelGGL((gemm_kernel<0, 128, 128, 2, 2>), dim3(tiles < cap ? tiles : cap, slices), dim3(256), 0,
                     (hipStream_t)stream, *d);
  return msclip_launch_status();
}

// dW [M, N] (fp32, `slices` partial matrices) = X^T W for TOKEN-major operands X [T, ldx] (columns = output rows m) and
// W [T, ldw] (columns = output columns n): the weight gradients of the training step without the operand transposes
// (gemm_pp_kernel<0, false, 0, true>).  desc: X, W, out, zero, M, N, K = T, ldx, ldw, ldo; everything else unset.
extern "C" int msclip_gemm_splitk_tn(const msclip_gemm_desc* d, int slices, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_gemm_splitk_tn, stream, d, slices);
  if (!d || !d->X || !d->W || !d->out || !d->zero || slices < 1 || slices > 65535 || d->M_dev) return MSCLIP_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->ldx % 8) || (d->ldw % 8) || d->ldx < d->M || d->ldw < d->N || (d->ldo % 4) ||
      d->ldo < d->N || d->mode != 0 || d->out_kind != 1 || d->bias || d->resid || d->resid_kind || d->act || d->out2 || d->xb ||
      d->rowstat || d->W2 || d->rpg != 0x7fffffff || d->radd || d->roff || d->alpha != 1.f)
    return MSCLIP_EINVAL;
  const long long t256 = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
  // 32-bit lane offsets over the whole operand -- INCLUDING the rows an over-hanging last slice addresses (its row index reaches
  // K rounded up to a K-tile plus one K-tile per slice; they must fall outside the descriptor's range, not wrap around into
  // valid memory); tile-map reciprocals
  const long long krows = ((long long)d->K + 63) / 64 * 64 + 64ll * (slices + 1);
  if (krows * d->ldx * 2 + 4096 >= (1ll << 32) || krows * d->ldw * 2 + 4096 >= (1ll << 32) ||
      t256 * ((d->M + 255) / 256) * 4 >= (1ll << 32))
    return MSCLIP_EINVAL;
  const int ncu = device_cus();
  hipLaunchKernelGGL((gemm_pp_kernel<0, false, 0, true>), dim3(t256 < ncu ? (int)t256 : ncu, slices), dim3(512), 0,
                     (hipStream_t)stream, *d, nullptr, nullptr);
  return msclip_launch_status();
}

// fp8 (OCP e4m3) operands on the ping-pong kernel: X [M, K] and W [N, K] bytes (ldx / ldw in elements = bytes, multiples of 16),
// K a multiple of 128; out = epilogue(alpha * row_scale[m] * col_scale[n] * sum_k X[m, k] W[n, k]) with the usual bias /
// activation / residual epilogues.  Dense operands only.
extern "C" int msclip_gemm_f8(const msclip_gemm_desc* d, const float* row_scale, const float* col_scale, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_gemm_f8, stream, d, row_scale, col_scale);
  if (!d || !d->X || !d->W || !d->out || !d->zero || !row_scale || !col_scale) return MSCLIP_EINVAL;
  if (d->mode != 0 || d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % 128) || (d->ldx % 16) || (d->ldw % 16) || d->ldx < d->K ||
      d->ldw < d->K || d->rpg != 0x7fffffff || d->out2 || d->out_kind < 0 || d->out_kind > 2 || d->resid_kind < 0 || d->resid_kind > 2 ||
      d->rowstat || d->W2)
    return MSCLIP_EINVAL;
  if (d->out_kind == 2 && ((d->M % 256) || (d->N % 16) || (d->ldo % 16) || d->resid_kind || !(d->out_scale > 0.f)))
    return MSCLIP_EINVAL;
  const long long tiles = (long long)((d->M + 255) / 256) * ((d->N + 255) / 256);
  if ((long long)d->ldx * 256 + d->K >= (1ll << 31) || (long long)d->ldw * 256 + d->K >= (1ll << 31) ||
      tiles * ((d->M + 255) / 256) * 4 >= (1ll << 32))
    return MSCLIP_EINVAL;
  const int ncu = device_cus();
  if (d->xb) {   // c_proj under PRECISION fp8 as the producer of the next block's folded ln_1 (whole tiles, in-place fp32 residual update)
    if (d->resid_kind != 1 || d->out_kind != 1 || d->act || !d->center || !d->part || (d->M % 256) || (d->N % 256) || (d->ldxb % 4) ||
        (d->ldo % 4) || (d->ldr % 4) || (d->resid2 && (d->seg_split <= 0 || (d->seg_split % 256) || d->seg_split >= d->M)))
      return MSCLIP_EINVAL;
    hipLaunchKernelGGL((gemm_pp_kernel<0, true, 2>), dim3(tiles < ncu ? (int)tiles : ncu), dim3(512), 0, (hipStream_t)stream, *d,
                       row_scale, col_scale);
    return msclip_launch_status();
  }
  hipLaunchKernelGGL((gemm_pp_kernel<0, true>), dim3(tiles < ncu ? (int)tiles : ncu), dim3(512), 0, (hipStream_t)stream, *d,
                     row_scale, col_scale);
  return msclip_launch_status();
}

extern "C" int msclip_gemm(const msclip_gemm_desc* d, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_gemm, stream, d);
  const GemmVariant v = pick_variant(d);
  if (v == GV_INVALID) return MSCLIP_EINVAL;
  if (d->M_dev && v != GV_PP) return MSCLIP_EINVAL;   // device-side row counts: the persistent dense ping-pong kernel only
  hipStream_t st = (hipStream_t)stream;
  const int ncu = device_cus();
  const int tiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
  const int grid = tiles < ncu ? tiles : ncu;
  switch (v) {
    case GV_STREAM: if (!msclip_gemm_small_try(d, st, ncu)) return MSCLIP_EINVAL; break;
    case GV_PP:
      if (d->rowstat) hipLaunchKernelGGL((gemm_pp_kernel<0, false, 1>), dim3(grid), dim3(512), 0, st, *d, nullptr, nullptr);
      else if (d->xb) hipLaunchKernelGGL((gemm_pp_kernel<0, false, 2>), dim3(grid), dim3(512), 0, st, *d, nullptr, nullptr);
      else if (d->resid_kind == 4) hipLaunchKernelGGL((gemm_pp_kernel<0, false, 3>), dim3(grid), dim3(512), 0, st, *d, nullptr, nullptr);
      else hipLaunchKernelGGL((gemm_pp_kernel<0, false>), dim3(grid), dim3(512), 0, st, *d, nullptr, nullptr);
      break;
    case GV_PPCONV: hipLaunchKernelGGL((gemm_pp_kernel<1, false>), dim3(grid), dim3(512), 0, st, *d, nullptr, nullptr); break;
    case GV_DENSE128: launch_cfg<0, 128, 128, 2, 2>(d, st, 2); break;
    case GV_CONV192: launch_cfg<1, 256, 192, 4, 2>(d, st, 1); break;
    case GV_DENSE192: launch_cfg<0, 256, 192, 4, 2>(d, st, 1); break;
    case GV_CONV128: launch_cfg<1, 128, 128, 2, 2>(d, st, 2); break;
    default: return MSCLIP_EINVAL;
  }
  return msclip_launch_status();
}
