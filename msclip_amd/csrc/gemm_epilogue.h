// Epilogues of the 128 x 64 wave tile of the dense ping-pong GEMMs (gemm.hip: gemm_pp_kernel, gemm_pp2.hip: gemm_pp2_kernel):
// accumulators in the v_mfma_f32_16x16x32 layout -> bias / activation / residual -> full-line global stores, staged through a
// 4-KiB LDS block per wave.
#pragma once
#include "common.h"
#include "../../include/msclip_hip.h"

#ifndef PP_EPI_NT
#define PP_EPI_NT 1        // the producer epilogue's fp32 residual stream is loaded / stored non-temporally (0: plain: probe builds)
#endif
#ifndef PP_EPI_GROW
#define PP_EPI_GROW 0      // 1: the residual look-ahead of the producer epilogue grows as accumulator registers die (round 5: measured neutral -- the epilogue is HBM-bound chip-wide, tools/probes/epi_probe.py -- and 13 more VGPRs)
#endif

namespace {

#ifndef PP_QKV_TEMPORAL
#define PP_QKV_TEMPORAL 0
#endif
constexpr int STG_BYTES = 4096;   // per-wave staging: 32 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7)

// Interior tile of the ping-pong kernel.  Every 32x32 accumulator tile goes through the wave's 4-KiB staging block
// as fp32 (32 rows x 128 B, chunks XOR-swizzled by row & 7) so that afterwards lane (srow = lane/8, sch = lane%8)
// owns 4 consecutive columns sch*4.. of rows srow, srow+8, ...: bias (8 registers, loaded at tile start by the
// caller), activation and residual are applied there.  Nothing is loaded in the epilogue for the bf16 outputs
// (a load would sit behind the in-flight LDS-DMA of the next tile: the VM counter retires in order), the fp32
// residual rows of two 32-row blocks are requested up front and re-requested two blocks ahead.
// fp32 outputs leave as full 128-byte lines, bf16 outputs as aligned 64-byte half lines.
// RK / ACT / OUTK: resid_kind / act / out_kind known at compile time, or -1 = read the descriptor.
// The staging traffic is inline asm: a compiler-visible LDS access after an LDS-DMA gets an s_waitcnt vmcnt(0) in front of
// it (the DMA is a pending LDS write), i.e. a stall on the next tile's in-flight prefetch.  A wave's DS instructions
// execute in issue order, so block b+1 may be written over block b as soon as b's reads are issued.
__device__ __forceinline__ void stg_write16(unsigned addr, f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4 stg_read16(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// Global-memory accesses of these epilogues carry address space 1 explicitly: a store through a generic pointer "may
// alias LDS", and while it is pending the compiler guards the K loop's fragment reads with s_waitcnt vmcnt(0).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void st_global(void* p, float4 v) { *(AS1 f32x4*)p = f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st_global(void* p, uint4 v) { *(AS1 u32x4*)p = u32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st_global(void* p, uint2 v) { *(AS1 u32x2*)p = u32x2{v.x, v.y}; }
__device__ __forceinline__ float4 ld_global_f4(const void* p) {
  const f32x4 v = *(const AS1 f32x4*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 ld_global_u2(const void* p) {
  const u32x2 v = *(const AS1 u32x2*)p;
  return make_uint2(v[0], v[1]);
}

// d/dh [h sigma(1.702 h)] = s + 1.702 h s (1 - s), s = sigma(1.702 h)   (M.py:222-224)
__device__ __forceinline__ float quickgelu_grad(float h) {
  const float s = 1.f / (1.f + __expf(-1.702f * h));
  return s + 1.702f * h * s * (1.f - s);
}

// bias4[tn] = this lane's 4 epilogue columns (lane & 7) * 4 .. + 3 of 32-column block tn, from the wave's one-register bias
// (lane j: column j; requested at tile start): 8 lane-crossbar reads here instead of 8 registers live across the K loop.
template <int TN>
__device__ __forceinline__ void bias4_from_bcol(float bcol, int lane, float4 (&bias4)[TN]) {
  const int src = __float_as_int(bcol);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int base = (tn * 32 + (lane & 7) * 4) * 4;
    bias4[tn].x = __int_as_float(__builtin_amdgcn_ds_bpermute(base, src));
    bias4[tn].y = __int_as_float(__builtin_amdgcn_ds_bpermute(base + 4, src));
    bias4[tn].z = __int_as_float(__builtin_amdgcn_ds_bpermute(base + 8, src));
    bias4[tn].w = __int_as_float(__builtin_amdgcn_ds_bpermute(base + 12, src));
  }
  // all of them are back before the first hand-counted LDS operation of the staging code is issued
  static_assert(TN == 2, "wave tile = 64 columns");
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(bias4[0].x), "+v"(bias4[0].y), "+v"(bias4[0].z), "+v"(bias4[0].w), "+v"(bias4[1].x), "+v"(bias4[1].y),
                 "+v"(bias4[1].z), "+v"(bias4[1].w)
               :
               : "memory");
}

template <int TM, int TN, int RK, int ACT, int OUTK, bool TE = false>     // TE: the training-step forms are compiled in
__device__ __forceinline__ void epilogue_rows(f32x4 (&acc)[2 * TN][2 * TM], const msclip_gemm_desc& a, unsigned stg,
                                              int mw0, int nw0, int lane, float bcol) {
  float4 bias4[TN];
  bias4_from_bcol<TN>(bcol, lane, bias4);
  // accumulator layout of v_mfma_f32_16x16x32 with swapped operands: acc[ni][mi][r] = C[mi*16 + lane%16][ni*16 + 4*(lane/16) + r]
  const int r16 = lane & 15, quad = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  const unsigned wr = stg + r16 * 128;
  const int wsw = r16 & 7;
  const unsigned rd = stg + srow * 128 + ((sch ^ srow) << 4);    // + i * 1024 for rows i*8 + srow
  const int rk = RK >= 0 ? RK : a.resid_kind;
  const int act = ACT >= 0 ? ACT : a.act;
  const int outk = OUTK >= 0 ? OUTK : a.out_kind;
  constexpr int RAHEAD = 2;                                      // residual blocks (32 x 32) requested ahead ...
  constexpr int NRV = RAHEAD;                                    // (the growing look-ahead of epilogue_rows_stats spills here: 129 VGPRs in the standard kernel)
  float4 rv[NRV][4];                                             // raw: fp32 x 4, or bf16 x 4 in .x/.y (unpacked at use)
  auto load_res = [&](int b, float4 (&dst)[4]) {
    const int tm = b / TN, tn = b % TN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t row = (size_t)(mw0 + tm * 32 + i * 8 + srow);
      const int n = nw0 + tn * 32 + sch * 4;
      if (n >= a.N) {
        dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (rk == 1) {
        dst[i] = ld_global_f4((const float*)a.resid + row * a.ldr + n);
      } else {
        const uint2 u = ld_global_u2((const bf16_t*)a.resid + row * a.ldr + n);
        dst[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
      }
    }
  };
  if (rk) {
#pragma unroll
    for (int b = 0; b < RAHEAD; ++b) load_res(b, rv[b]);
  }
  f32x4 x[2][4];
  // resid_kind 4 with a.part: this lane's share of the COLUMN sums of the stored values over the wave's 128 rows (the bias
  // gradient of the projection whose output gradient this launch writes: no second pass over the [M, N] tensor)
  float4 cs[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) cs[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage = [&](int b, f32x4 (&dst)[4]) {                     // block b = tm * TN + tn -> LDS -> row-major registers
    const int tm = b / TN, tn = b % TN;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {                              // the four 16 x 16 tiles of this 32 x 32 block
        const f32x4 v = acc[2 * tn + ni][2 * tm + mi] * a.alpha;
        stg_write16(wr + mi * 2048 + (((ni * 4 + quad) ^ wsw) << 4), v);
      }
    dst[0] = stg_read16<0>(rd); dst[1] = stg_read16<1024>(rd);
    dst[2] = stg_read16<2048>(rd); dst[3] = stg_read16<3072>(rd);
  };
  stage(0, x[0]);
#pragma unroll
  for (int b = 0; b < TM * TN; ++b) {
    const int tm = b / TN, tn = b % TN;
    f32x4(&xb)[4] = x[b & 1];
    if (b + 1 < TM * TN) {
      stage(b + 1, x[(b + 1) & 1]);
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
    }
    const int n = nw0 + tn * 32 + sch * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = make_float4(xb[i][0] + bias4[tn].x, xb[i][1] + bias4[tn].y, xb[i][2] + bias4[tn].z, xb[i][3] + bias4[tn].w);
      if (act == 1) {
        v.x = v.x / (1.f + __expf(-1.702f * v.x)); v.y = v.y / (1.f + __expf(-1.702f * v.y));
        v.z = v.z / (1.f + __expf(-1.702f * v.z)); v.w = v.w / (1.f + __expf(-1.702f * v.w));
      }
      if (rk == 1) {
        const float4 r = rv[b % NRV][i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      } else if (TE && rk == 4) {                                    // v *= QuickGELU'(h), h = the bf16 "residual" (training dgrad)
        const unsigned ux = __float_as_uint(rv[b % NRV][i].x), uy = __float_as_uint(rv[b % NRV][i].y);
        v.x *= quickgelu_grad(__uint_as_float(ux << 16)); v.y *= quickgelu_grad(__uint_as_float(ux & 0xffff0000u));
        v.z *= quickgelu_grad(__uint_as_float(uy << 16)); v.w *= quickgelu_grad(__uint_as_float(uy & 0xffff0000u));
        cs[tn].x += v.x; cs[tn].y += v.y; cs[tn].z += v.z; cs[tn].w += v.w;
      } else if (rk) {
        const unsigned ux = __float_as_uint(rv[b % NRV][i].x), uy = __float_as_uint(rv[b % NRV][i].y);
        v.x += __uint_as_float(ux << 16); v.y += __uint_as_float(ux & 0xffff0000u);
        v.z += __uint_as_float(uy << 16); v.w += __uint_as_float(uy & 0xffff0000u);
      }
      if (act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const size_t row = (size_t)(mw0 + tm * 32 + i * 8 + srow);
      if (n >= a.N) {
      } else if (outk == 1) {
        st_global((float*)a.out + row * a.ldo + n, v);
      } else {
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y);
        o.y = pack_bf16x2(v.z, v.w);
        st_global((bf16_t*)a.out + row * a.ldo + n, o);
      }
    }
    if (rk && b + RAHEAD < TM * TN) load_res(b + RAHEAD, rv[b % RAHEAD]);
  }
  if (TE && rk == 4 && a.part) {
    // fold the 8 lanes that hold the same 4 columns (srow = lane / 8: lane bits 3..5), fixed order; lanes 0..7 write the wave's
    // row of the partials: part[mw0 / 128][N] fp32, a full 128-byte line per 32-column block.  (The staging code's hand-counted
    // LDS-pipe operations are all retired here: the last block waited lgkmcnt(0).)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int sh = 8; sh <= 32; sh <<= 1) {
        cs[tn].x += __shfl_xor(cs[tn].x, sh); cs[tn].y += __shfl_xor(cs[tn].y, sh);
        cs[tn].z += __shfl_xor(cs[tn].z, sh); cs[tn].w += __shfl_xor(cs[tn].w, sh);
      }
      const int n = nw0 + tn * 32 + sch * 4;
      if (srow == 0 && n < a.N) st_global(a.part + (size_t)(mw0 >> 7) * a.N + n, cs[tn]);
    }
  }
}

// out_proj / c_proj as the PRODUCER of the next LayerNorm's operands (DESIGN "LayerNorm fold"): besides the in-place fp32
// residual update v = resid + alpha acc + bias it writes xb[m][n] = bf16(v - center[m]) -- the operand the next projection reads
// instead of a LayerNorm output; center[m] is the row's mean at the previous LayerNorm point, so the bf16 rounding is relative
// to the deviation from the mean as it is for a LayerNorm output -- and part[m][column group][2] = (sum, sum of squares) of
// v - center[m] over the wave's 64 columns, fp32 (msclip_rowstat_finalize folds a row's N / 64 groups in a fixed order: mean,
// rstd).  Same staging / residual look-ahead as epilogue_rows<.., 1, 0, 1>.
// sum over each aligned group of 8 lanes, in every lane of the group: DPP moves (quad xor 1, quad xor 2, half-row mirror), no
// LDS-pipe traffic (a ds_bpermute-based shuffle would sit in the hand-counted lgkmcnt sequence of the staging code)
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue_rows_stats(f32x4 (&acc)[2 * TN][2 * TM], const msclip_gemm_desc& a, unsigned stg,
                                                    int mw0, int nw0, int lane, float bcol, float cen_lo, float cen_hi,
                                                    const float* __restrict__ resid) {
  static_assert(TN == 2 && TM == 4, "wave tile = 128 rows x 64 columns = one statistics group per row");
  float4 bias4[TN];
  bias4_from_bcol<TN>(bcol, lane, bias4);
  const int r16 = lane & 15, quad = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  const unsigned wr = stg + r16 * 128;
  const int wsw = r16 & 7;
  const unsigned rd = stg + srow * 128 + ((sch ^ srow) << 4);
  // Residual prefetch depth.  The epilogue is a chain of memory round trips: with the residual rows of only two 32 x 32 blocks
  // in flight per wave (all the registers there are while the 128 accumulator registers are live) a CU has 64 KB outstanding
  // and waits out ~4 latencies per tile.  The accumulators of finished blocks are dead registers, so the depth GROWS: blocks 0, 1
  // up front, then two more behind every finished block (2b + 2, 2b + 3): five blocks in flight by the middle of the tile.
  // PP_EPI_GROW=0 rebuilds the fixed two-ahead form.
  constexpr int RAHEAD = 2;
  constexpr int NRV = PP_EPI_GROW ? TM * TN : RAHEAD;
  float4 rv[NRV][4];
  auto load_res = [&](int b, float4 (&dst)[4]) {
    const int tm = b / TN, tn = b % TN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t row = (size_t)(mw0 + tm * 32 + i * 8 + srow);
      // non-temporal: with the fold nobody re-reads the fp32 stream soon (no LayerNorm pass behind this launch), so it should
      // not take the MALL from the bf16 operand the next projection reads (same-box A/B, loads + stores: +0.5-0.9 % on the step)
#if PP_EPI_NT
      const f32x4 t = __builtin_nontemporal_load((const AS1 f32x4*)(resid + row * a.ldr + nw0 + tn * 32 + sch * 4));
#else
      const f32x4 t = *(const AS1 f32x4*)(resid + row * a.ldr + nw0 + tn * 32 + sch * 4);
#endif
      dst[i] = make_float4(t[0], t[1], t[2], t[3]);
    }
  };
#pragma unroll
  for (int b = 0; b < RAHEAD; ++b) load_res(b, rv[b]);
  // centre of row r of the wave's 128: lane r & 63 of cen_lo (r < 64) / cen_hi (requested at tile start like the bias).  The
  // lane-crossbar reads are inline asm (the compiler must not count them: its own wait would also drain the staging reads
  // issued behind them); LDS-pipe operations return in order, so the hand-counted waits below cover them.
  auto issue_cen = [&](int tm, float (&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int addr = ((tm & 1) * 32 + i * 8 + srow) * 4;
      asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(dst[i]) : "v"(addr), "v"(tm < 2 ? cen_lo : cen_hi) : "memory");
    }
  };
  float cen[4], cenn[4];
  issue_cen(0, cen);
  f32x4 x[2][4];
  auto stage = [&](int b, f32x4 (&dst)[4]) {
    const int tm = b / TN, tn = b % TN;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const f32x4 v = acc[2 * tn + ni][2 * tm + mi] * a.alpha;
        stg_write16(wr + mi * 2048 + (((ni * 4 + quad) ^ wsw) << 4), v);
      }
    dst[0] = stg_read16<0>(rd); dst[1] = stg_read16<1024>(rd);
    dst[2] = stg_read16<2048>(rd); dst[3] = stg_read16<3072>(rd);
  };
  stage(0, x[0]);
  float ps[4], pq[4];                                              // this lane's share of the row sums over the block row's 64 columns
  uint2 hold[4];
  const int ngrp = a.N >> 6, grp = nw0 >> 6;
#pragma unroll
  for (int b = 0; b < TM * TN; ++b) {
    const int tm = b / TN, tn = b % TN;
    f32x4(&xb)[4] = x[b & 1];
    const bool next_cen = tn == TN - 1 && tm + 1 < TM;
    if (next_cen) issue_cen(tm + 1, cenn);                         // older than the staging of block b + 1: covered by its wait
    if (b + 1 < TM * TN) {
      stage(b + 1, x[(b + 1) & 1]);
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]), "+v"(cen[0]), "+v"(cen[1]), "+v"(cen[2]), "+v"(cen[3]),
                     "+v"(cenn[0]), "+v"(cenn[1]), "+v"(cenn[2]), "+v"(cenn[3])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
    }
    const int n = nw0 + tn * 32 + sch * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 r = rv[b % NRV][i];
      const float4 v = make_float4(xb[i][0] + bias4[tn].x + r.x, xb[i][1] + bias4[tn].y + r.y, xb[i][2] + bias4[tn].z + r.z,
                                   xb[i][3] + bias4[tn].w + r.w);
      const size_t row = (size_t)(mw0 + tm * 32 + i * 8 + srow);
#if PP_EPI_NT
      __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (AS1 f32x4*)((float*)a.out + row * a.ldo + n));
#else
      *(AS1 f32x4*)((float*)a.out + row * a.ldo + n) = f32x4{v.x, v.y, v.z, v.w};
#endif
      const float c = cen[i];
      const float d0 = v.x - c, d1 = v.y - c, d2 = v.z - c, d3 = v.w - c;
      uint2 o;
      o.x = pack_bf16x2(d0, d1);
      o.y = pack_bf16x2(d2, d3);
      if (tn == 0) {
        hold[i] = o;                                                // the row's first 32 columns wait for the other 32
      } else {
        // one full 128-byte line per row and store instruction: neighbouring lanes swap 8-byte pieces, even lanes write the
        // line's first half (16 bytes of the first block: own + neighbour's), odd lanes the second (neighbour's + own)
        const bool odd = sch & 1;
        const uint2 send = odd ? hold[i] : o;
        uint2 recv;
        recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xF, 0xF, true);     // quad_perm [1, 0, 3, 2]
        recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xF, 0xF, true);
        const uint4 q = odd ? make_uint4(recv.x, recv.y, o.x, o.y) : make_uint4(hold[i].x, hold[i].y, recv.x, recv.y);
        st_global((bf16_t*)a.xb + row * a.ldxb + nw0 + (odd ? 32 + (sch - 1) * 4 : sch * 4), q);
      }
      const float s4 = (d0 + d1) + (d2 + d3), q4 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      ps[i] = tn == 0 ? s4 : ps[i] + s4;
      pq[i] = tn == 0 ? q4 : pq[i] + q4;
    }
    if (PP_EPI_GROW) {
      if (2 * b + 2 < TM * TN) load_res(2 * b + 2, rv[(2 * b + 2) % NRV]);
      if (2 * b + 3 < TM * TN) load_res(2 * b + 3, rv[(2 * b + 3) % NRV]);
    } else if (b + RAHEAD < TM * TN) {
      load_res(b + RAHEAD, rv[b % RAHEAD]);
    }
    if (tn == TN - 1) {                                             // the block row's 64 columns are complete: fold the 8 lanes of a row
      float s[4], q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i] = sum8_dpp(ps[i]);
        q[i] = sum8_dpp(pq[i]);
      }
      // lane sch = i of a row group writes row i * 8 + srow: the block row's 32 rows leave in ONE store instruction
      const float ss = sch == 0 ? s[0] : sch == 1 ? s[1] : sch == 2 ? s[2] : s[3];
      const float qq = sch == 0 ? q[0] : sch == 1 ? q[1] : sch == 2 ? q[2] : q[3];
      if (sch < 4) {
        const size_t row = (size_t)(mw0 + tm * 32 + sch * 8 + srow);
        *(AS1 f32x2*)(a.part + (row * ngrp + grp) * 2) = f32x2{ss, qq};
      }
      if (next_cen) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cen[i] = cenn[i];
      }
    }
  }
}

// bf16 outputs without a residual (QKV, c_fc).  The staged fp32 epilogue above is bound by its LDS-write and
// store-instruction issue (8-byte stores, 16-byte LDS writes), so here bias and activation are applied in the
// accumulator layout, the tile is packed to bf16 BEFORE staging (8-byte LDS writes, half the bytes) and a whole
// 32-row x 64-column block of the wave is staged at once: the read-back hands every lane 16 bytes and a store
// instruction writes 8 full 128-byte lines.  The wave's 64 bias values live in ONE register (lane j: column j,
// requested at tile start) and reach the accumulator layout through v_readlane.
__device__ __forceinline__ void stg_write8(unsigned addr, unsigned lo, unsigned hi) {
  u32x2 v = {lo, hi};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ u32x4 stg_read16u(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

// FOLD (DESIGN "LayerNorm fold"): the X operand was (x - center) in bf16 and W carries the LayerNorm's gamma, so the LayerNorm
// output's projection is  rstd[m] * acc - (mean[m] * rstd[m]) * csum[n] + bias'[n]  (M.py:204-219 applied to the GEMM's result:
// csum[n] = sum_k W'[n][k], bias' = bias + W beta).  rstat = this wave's 128 rows' (rstd, mean * rstd), lane j: rows j (.x, .y)
// and j + 64 (.z, .w), requested at tile start like the bias; ccol = csum of the wave's column j.
template <int TM, int TN, int ACT, bool FOLD = false>
__device__ __forceinline__ void epilogue_pack16(f32x4 (&acc)[2 * TN][2 * TM], const msclip_gemm_desc& a, unsigned stg,
                                                int mw0, int nw0, int lane, float bcol, void* outp, float ccol = 0.f,
                                                float4 rstat = make_float4(1.f, 0.f, 1.f, 0.f)) {
  static_assert(TN == 2, "64 bf16 columns = one 128-byte staged row");
  const int r16 = lane & 15, quad = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  const unsigned wr = stg + r16 * 128 + (quad & 1) * 8;
  const int wsw = r16 & 7;
  const unsigned rd = stg + srow * 128 + ((sch ^ srow) << 4);    // + i * 1024 for rows i*8 + srow
  // bias of columns ni*16 + 4*quad + r: lane c of bcol holds column c's bias; 16 lane-crossbar reads (ds_bpermute
  // touches no LDS memory).  The v_readlane + three-way select this replaces compiled into divergent control flow
  // (one exec-mask branch nest per value: ~400 scalar-heavy instructions, 3-4 k cycles per tile in the phase trace).
  float b[4][4];
  {
    const int src = __float_as_int(bcol), base = quad * 16;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        b[ni][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(base + (ni * 16 + r) * 4, src));
    // all sixteen are back before the first hand-counted LDS operation of the staging code is issued
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]),
                   "+v"(b[1][3]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[2][2]), "+v"(b[2][3]), "+v"(b[3][0]), "+v"(b[3][1]),
                   "+v"(b[3][2]), "+v"(b[3][3])
                 :
                 : "memory");
  }
  float cs[FOLD ? 4 : 1][4], rs[FOLD ? 2 * TM : 1], sh[FOLD ? 2 * TM : 1];
  if constexpr (FOLD) {
    const int src = __float_as_int(ccol), base = quad * 16;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cs[ni][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(base + (ni * 16 + r) * 4, src));
#pragma unroll
    for (int mi = 0; mi < 2 * TM; ++mi) {                            // row mi*16 + r16 of the wave's 128: lane (mi & 3)*16 + r16, half mi >> 2
      const int sl = ((mi & 3) * 16 + r16) * 4;
      rs[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(mi < 4 ? rstat.x : rstat.z)));
      sh[mi] = __int_as_float(__builtin_amdgcn_ds_bpermute(sl, __float_as_int(mi < 4 ? rstat.y : rstat.w)));
    }
    static_assert(!FOLD || TM == 4, "128 rows per wave");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(cs[0][0]), "+v"(cs[0][1]), "+v"(cs[0][2]), "+v"(cs[0][3]), "+v"(cs[1][0]), "+v"(cs[1][1]), "+v"(cs[1][2]),
                   "+v"(cs[1][3]), "+v"(cs[2][0]), "+v"(cs[2][1]), "+v"(cs[2][2]), "+v"(cs[2][3]), "+v"(cs[3][0]), "+v"(cs[3][1]),
                   "+v"(cs[3][2]), "+v"(cs[3][3])
                 :
                 : "memory");
    asm volatile("" : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]), "+v"(rs[4]), "+v"(rs[5]), "+v"(rs[6]), "+v"(rs[7]),
                      "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]), "+v"(sh[4]), "+v"(sh[5]), "+v"(sh[6]), "+v"(sh[7]));
  }
  // Block tm = 32 rows x 64 columns of the wave.  Its values are computed in four QUARTERS (8 values + one 16-byte staging
  // write pair each); between the quarters of block tm+1 the four 1-KiB store instructions of block tm go out one at a
  // time instead of back to back.  Measured same-box against the back-to-back form: QKV 226.6 -> 223.1 us, c_fc unchanged
  // (334 -> 333 us), ping-pong launches in the model step 187.0 -> 185.6 us on average.  (The probe builds say the c_fc
  // epilogue is 57 us of activation math + staging and 32 us that vanish without the stores, nearly additive; spreading
  // the stores within the epilogue does not recover them.  Nor are they lost in the next tile's counted waits: a timing-only
  // probe whose K-tile 1 / K-tile 2 waits also leave the epilogue's stores in flight -- wrong results, -DPP_RELAX2 / 3 of
  // tools/probes/gemm_pp_probes.hip -- runs every shape within noise of the shipped kernel.  What is left is the CU's
  // store path itself: ~128 KiB per tile at ~16 B/clk is ~8 k cycles during which all eight waves sit in the epilogue and
  // nobody issues MFMAs.)
  u32x4 x[2][4];
  auto quarter = [&](int tm, int q) {                            // (mi, ni) = (q >> 1, 2 * (q & 1) + {0, 1})
    const int mi = q >> 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ni = 2 * (q & 1) + h;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (FOLD) v[e] = acc[ni][2 * tm + mi][e] * rs[2 * tm + mi] + (b[ni][e] - sh[2 * tm + mi] * cs[ni][e]);
        else v[e] = acc[ni][2 * tm + mi][e] * a.alpha + b[ni][e];
        if (ACT == 1) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));
        if (ACT == 2) v[e] = fmaxf(v[e], 0.f);
      }
      stg_write8(wr + mi * 2048 + (((ni * 2 + (quad >> 1)) ^ wsw) << 4), pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  };
  auto fetch = [&](u32x4 (&dst)[4]) {                            // the staged block, 4 rows of 128 B per lane group
    dst[0] = stg_read16u<0>(rd); dst[1] = stg_read16u<1024>(rd);
    dst[2] = stg_read16u<2048>(rd); dst[3] = stg_read16u<3072>(rd);
  };
  auto store = [&](int tm, int i, const u32x4& v) {
    const int n = nw0 + sch * 8;
    const size_t row = (size_t)(mw0 + tm * 32 + i * 8 + srow);
    // non-temporal: the tile leaves faster (QKV 262 -> 249 us, c_fc 378 -> 365 us; +0.7 % on the step), the fp32
    // stream of out_proj / c_proj stays cacheable for the LayerNorm that follows
#if PP_QKV_TEMPORAL   // probe builds only: the q|k|v matrix of the packed step (199 MB) fits the 256 MB MALL -- does attention find it there?
    if (ACT == 0) { if (n < a.N) *(AS1 u32x4*)((bf16_t*)outp + row * a.ldo + n) = v; } else
#endif
    if (n < a.N) __builtin_nontemporal_store(v, (AS1 u32x4*)((bf16_t*)outp + row * a.ldo + n));
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) quarter(0, q);
  fetch(x[0]);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    u32x4(&xb)[4] = x[tm & 1];
    if (tm + 1 < TM) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        quarter(tm + 1, q);
        // the 4 reads of block tm are older than the 2 staging writes of this quarter (LDS returns in order)
        if (q == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
        store(tm, q, xb[q]);
        __builtin_amdgcn_sched_barrier(0);
      }
      fetch(x[(tm + 1) & 1]);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3])::"memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) store(tm, i, xb[i]);
    }
  }
}

// e4m3 outputs (the fp8 instantiation only: c_fc under MODEL.SPEC.PRECISION fp8 writes the MLP hidden matrix as the fp8 operand
// of c_proj).  out[m][n] = fp8(act(alpha acc + bias) * qscale) with ONE static scale per tensor (calibrated, engine.py); values
// beyond the calibrated range saturate at the format's largest finite value.  A 32-row x 64-column block of the wave is 2 KiB of
// bytes: staged with an 80-byte row stride (16-byte aligned read-back, 2-way bank aliasing on the 4-byte writes costs nothing),
// read back as 16 bytes per lane, stored as 64-byte row segments (the wave's 64 columns of a row).
__device__ __forceinline__ void stg_write4(unsigned addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int TM, int TN, int ACT>
__device__ __forceinline__ void epilogue_pack8(f32x4 (&acc)[2 * TN][2 * TM], const msclip_gemm_desc& a, unsigned stg,
                                               int mw0, int nw0, int lane, float bcol, float qscale) {
  static_assert(TN == 2, "64 e4m3 columns = one 64-byte staged row");
  const int r16 = lane & 15, quad = lane >> 4;
  const unsigned wr = stg + r16 * 80 + quad * 4;                 // + mi * 1280 + ni * 16
  const int srow = lane >> 2, sch = lane & 3;                    // read-back: rows srow, srow + 16; 16-byte chunk sch
  const unsigned rd = stg + srow * 80 + sch * 16;
  float b[4][4];
  {
    const int src = __float_as_int(bcol), base = quad * 16;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        b[ni][r] = __int_as_float(__builtin_amdgcn_ds_bpermute(base + (ni * 16 + r) * 4, src));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]),
                   "+v"(b[1][3]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[2][2]), "+v"(b[2][3]), "+v"(b[3][0]), "+v"(b[3][1]),
                   "+v"(b[3][2]), "+v"(b[3][3])
                 :
                 : "memory");
  }
  // Software-pipelined like epilogue_pack16 (round 6; round 5 staged, waited and stored one 32-row block at a time: 148 us for
  // c_fc's 306 MB at M = 74 752 where the bf16 form moves twice the bytes in 125 us): block tm + 1 is computed in four QUARTERS
  // (8 values -> two 4-byte staging writes each) while the two 1-KiB stores of block tm go out between them; the staging block
  // is reused as soon as its read-back has been ISSUED (a wave's DS instructions execute in order).
  u32x4 x[2][2];
  auto quarter = [&](int tm, int q) {                            // (mi, ni) = (q >> 1, 2 * (q & 1) + {0, 1})
    const int mi = q >> 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ni = 2 * (q & 1) + h;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[ni][2 * tm + mi][e] * a.alpha + b[ni][e];
        if (ACT == 1) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));
        if (ACT == 2) v[e] = fmaxf(v[e], 0.f);
        v[e] = fminf(fmaxf(v[e] * qscale, -448.f), 448.f);
      }
      int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
      p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true);
      stg_write4(wr + mi * 1280 + ni * 16, p);
    }
  };
  auto fetch = [&](u32x4 (&dst)[2]) {
    dst[0] = stg_read16u<0>(rd);
    dst[1] = stg_read16u<1280>(rd);
  };
  const int n = nw0 + sch * 16;
  auto store = [&](int tm, int i, const u32x4& v) {              // rows srow (i = 0) / srow + 16 (i = 1) of block tm
    const size_t row = (size_t)(mw0 + tm * 32 + srow + 16 * i);
    if (n < a.N) __builtin_nontemporal_store(v, (AS1 u32x4*)((unsigned char*)a.out + row * a.ldo + n));
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) quarter(0, q);
  fetch(x[0]);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    u32x4(&xb)[2] = x[tm & 1];
    if (tm + 1 < TM) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        quarter(tm + 1, q);
        // the 2 reads of block tm are older than the 2 staging writes of this quarter (LDS returns in order)
        if (q == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xb[0]), "+v"(xb[1])::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (q == 0) store(tm, 0, xb[0]);
        if (q == 2) store(tm, 1, xb[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      fetch(x[(tm + 1) & 1]);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[0]), "+v"(xb[1])::"memory");
      store(tm, 0, xb[0]);
      store(tm, 1, xb[1]);
    }
  }
}

// Edge tiles of the ping-pong kernel (rows past M, ragged N, unaligned leading dimensions): guarded, straight from the
// 16 x 16 accumulator layout (lane owns row mi*16 + lane%16, columns ni*16 + 4*(lane/16) + 0..3).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_generic16(f32x4 (&acc)[2 * TN][2 * TM], const msclip_gemm_desc& a, bool vec,
                                                   int mw0, int nw0, int lane) {
  const int r16 = lane & 15, quad = lane >> 4;
  const float* __restrict__ bias = a.bias;
#pragma unroll
  for (int mi = 0; mi < 2 * TM; ++mi) {
    const int m = mw0 + mi * 16 + r16;
    if (m >= a.M) continue;
    const int grp = m / a.rpg;
    const size_t orow = (size_t)(m + grp * a.radd + a.roff);       // row scatter (stem -> token rows)
    size_t row = (size_t)m;                                         // residual row
    if (a.resid_kind == 3) row = (size_t)(m - grp * a.rpg + a.roff);
#pragma unroll
    for (int ni = 0; ni < 2 * TN; ++ni) {
      const int n = nw0 + ni * 16 + quad * 4;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][j] * a.alpha;
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
          const float4 rv = *(const float4*)((const float*)a.resid + row * a.ldr + n);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        } else if (a.resid_kind == 2) {
          const uint2 rv = *(const uint2*)((const bf16_t*)a.resid + row * a.ldr + n);
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
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + j >= a.N) break;
          float y = v[j];
          if (bias) y += bias[n + j];
          if (a.act == 1) y = y / (1.f + __expf(-1.702f * y));
          if (a.resid_kind == 1 || a.resid_kind == 3) y += ((const float*)a.resid)[row * a.ldr + n + j];
          else if (a.resid_kind == 2) y += bf16_to_f32(((const bf16_t*)a.resid)[row * a.ldr + n + j]);
          if (a.act == 2) y = fmaxf(y, 0.f);
          if (a.out_kind == 1) ((float*)a.out)[orow * a.ldo + n + j] = y;
          else ((bf16_t*)a.out)[orow * a.ldo + n + j] = f32_to_bf16(y);
        }
      }
    }
  }
}

}  // namespace
