// Two co-resident workgroups per CU for the dense projections (QKV / out_proj / c_fc / c_proj, reference
// lib/models/clip_openai_pe_res_v1.py:612,747,794-798): gemm_pp2_kernel.
//
// Why: in gemm_pp_kernel (gemm.hip) the single resident 8-wave workgroup reaches its epilogue with all eight waves at
// once -- for 18 % (QKV) to 51 % (out_proj) of a tile nobody issues MFMAs while the CU pushes its output tile out
// (DESIGN.md "The epilogue problem").  Here a workgroup is 4 waves, ONE per SIMD, on a 256 x 128 tile (the same
// 128 x 64 wave tile, accumulators and epilogues as the ping-pong kernel), 80 KiB of LDS, so TWO workgroups share a
// CU: on every SIMD sit two waves of DIFFERENT workgroups that share no barrier.  The second workgroup of a CU starts
// half a tile late, so that one workgroup's epilogue (VALU work + store push) runs under the other's main loop on the
// same SIMDs.  The price: W is no longer shared across 256 output columns -- 384 operand rows per 256 x 128 x 64
// instead of 512 per 256 x 256 x 64, i.e. 1.5 x the L2 -> LDS traffic and LDS-DMA pieces per FLOP.
//
// LDS: ring of ten 8-KiB regions (64 rows x 128 B, full lines, XOR-swizzled like gemm_pp_kernel's).  A K-tile (64 deep)
// is six regions: W0 | W1 (tile columns 0-63 | 64-127), XA0 | XA1 (tile rows 0-63 | 128-191), XB0 | XB1 (rows 64-127 |
// 192-255).  Wave (wm, wn) owns rows wm*128.., columns wn*64..: phase A of a K-tile contracts W(wn) with XA(wm) (its
// first 64 rows), phase B W(wn) (kept in registers) with XB(wm).  Issue order per K-tile step t, region sequence number
// s = 6t + ..:
//     phase A(t):  XA(t+1)            -> the slots of XB(t-1)   (last read in phase B(t-1))
//     phase B(t):  XB(t+1), W(t+2)    -> the slots of W(t), XA(t) (last read in phase A(t))
// slot = s mod 10, so every region is in flight for a whole K-tile before its first read and each phase's wait is the
// same counted vmcnt(8): the wave's 8 youngest pieces (2 per region) may still be in flight.  A phase is
//     s_waitcnt vmcnt(8) ; s_barrier ; DMA issue ; fragment reads ; 32 MFMAs
// -- the barrier publishes every wave's landed pieces and orders all fragment reads of the regions that are re-issued
// behind it.  No ping-pong inside the workgroup: the partner on each SIMD is the OTHER workgroup's wave.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/msclip_hip.h"
#include "gemm_epilogue.h"

namespace {

constexpr int P2_SLOTS = 10, P2_REG = 64 * 64;   // ring regions, bf16 elements per region
constexpr int P2_CG = 8;                         // column group of the tile order, in 128-column tiles

__device__ __forceinline__ bf16x8 p2_ld(const void* p) { return *(const bf16x8*)p; }

#ifdef P2_TRACE   // probe builds only (tools/probes): cycle stamps of wave 0 of every workgroup, scalar stores (no vmcnt traffic)
constexpr int P2_TRACE_N = 128;
__device__ unsigned long long p2_trace_buf[1024 * P2_TRACE_N];
#define P2_STAMP(id)                                                                                   \
  if (wave == 0 && tr_n < P2_TRACE_N) {                                                                \
    unsigned long long t_;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    t_ = (t_ & 0x00ffffffffffffffull) | ((unsigned long long)(id) << 56);                              \
    asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(t_), "s"(tr_p + tr_n) : "memory");               \
    ++tr_n;                                                                                            \
  }
#else
#define P2_STAMP(id)
#endif

__global__ __launch_bounds__(256, 2) void gemm_pp2_kernel(const msclip_gemm_desc a, const int delay_mode, const int delay_unit) {
  constexpr int TM = 4, TN = 2;
  constexpr bool TE = true;                          // training-step epilogue forms (out2, resid_kind 4) compiled in
  __shared__ __attribute__((aligned(1024))) bf16_t smem[P2_SLOTS * P2_REG];   // 80 KiB: two workgroups per CU

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nt_n = (a.N + 127) / 128;
  const int nt_m = (a.M + 255) / 256;
  const int ntiles = nt_n * nt_m;
  const int nk = a.K / 64;

  // ---- tile order (same idea as gemm_pp_kernel): consecutive ids run on one XCD; column groups of eight 128-column tiles,
  // rows fastest inside a group, so the 64 tiles an XCD's 32 CUs x 2 workgroups run together cover 8 row blocks x 8 column
  // slices = the same 2048 x 1024 super-tile (12 distinct 256-row operand slices in its L2) as 32 tiles of 256 x 256.
  const unsigned tper = (unsigned)(nt_m * P2_CG);
  const unsigned tper_rcp = 0xffffffffu / tper + 1u;
  const int wg_tail = nt_n - (nt_n - 1) / P2_CG * P2_CG;            // width of the last column group (1..CG)
  const unsigned wgt_rcp = 0xffffffffu / (unsigned)wg_tail + 1u;    // unused when wg_tail == 1
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
    const unsigned id = (unsigned)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3));
    const unsigned g = (unsigned)(((unsigned long long)id * tper_rcp) >> 32);
    const unsigned idg = id - g * tper;
    const bool tail = (int)(g * P2_CG + P2_CG) > nt_n;
    const unsigned wg = tail ? (unsigned)wg_tail : (unsigned)P2_CG;
    const unsigned row = !tail ? idg >> 3 : wg_tail == 1 ? idg : (unsigned)(((unsigned long long)idg * wgt_rcp) >> 32);
    m0 = (int)(row * 256u);
    n0 = (int)((g * P2_CG + (idg - row * wg)) * 128u);
  };

  // ---- issue side.  A region is 8 pieces (8 rows x 128 B each); piece p is issued by wave p & 3: lane -> row 8p + lane/8,
  // physical 16-byte chunk lane%8 holds logical chunk (lane%8) ^ ((row >> 1) & 7).
  unsigned vx[2], vw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) << 4;
    vx[i] = (unsigned)row * (unsigned)a.ldx * 2u + ch;
    vw[i] = (unsigned)row * (unsigned)a.ldw * 2u + ch;
  }
  const unsigned x64 = 64u * (unsigned)a.ldx * 2u, w64 = 64u * (unsigned)a.ldw * 2u;
  int xti = blockIdx.x, xk = 0, wti = blockIdx.x, wk = 0, islot = 2;
  __amdgpu_buffer_rsrc_t rx, rw;
  auto set_x_tile = [&](int t) {
    if (t < ntiles) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const unsigned long long xb = (unsigned long long)(a.M - m0) * (unsigned long long)a.ldx * 2ull;
      rx = make_rsrc((const bf16_t*)a.X + (size_t)m0 * a.ldx, xb > 0xffffffffull ? 0xffffffffu : (unsigned)xb);
    } else {
      rx = make_rsrc(a.X, 0);                      // past the tile list: empty descriptor, the counts stay exact
    }
  };
  auto set_w_tile = [&](int t) {
    if (t < ntiles) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const unsigned long long wb = (unsigned long long)(a.N - n0) * (unsigned long long)a.ldw * 2ull;
      rw = make_rsrc((const bf16_t*)a.W + (size_t)n0 * a.ldw, wb > 0xffffffffull ? 0xffffffffu : (unsigned)wb);
    } else {
      rw = make_rsrc(a.W, 0);
    }
  };
  auto next_slot = [&]() { islot = islot == P2_SLOTS - 1 ? 0 : islot + 1; };
  auto issue_x = [&](int h) {                      // XA (h = 0: rows 0-63 | 128-191) or XB (h = 1: rows 64-127 | 192-255) of K-tile (xti, xk)
    const unsigned ko = (unsigned)xk * 128u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16_t* dst = smem + islot * P2_REG + wave * 512;
      const unsigned so = ko + (unsigned)(2 * j + h) * x64;
      blds16(rx, vx[0], so, dst);
      blds16(rx, vx[1], so, dst + 4 * 512);
      next_slot();
    }
    if (h) {
      if (++xk == nk) {
        xk = 0;
        xti += gridDim.x;
        set_x_tile(xti);
      }
    }
  };
  auto issue_w = [&]() {                           // W0 | W1 of K-tile (wti, wk)
    const unsigned ko = (unsigned)wk * 128u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16_t* dst = smem + islot * P2_REG + wave * 512;
      const unsigned so = ko + (unsigned)j * w64;
      blds16(rw, vw[0], so, dst);
      blds16(rw, vw[1], so, dst + 4 * 512);
      next_slot();
    }
    if (++wk == nk) {
      wk = 0;
      wti += gridDim.x;
      set_w_tile(wti);
    }
  };

  // ---- de-phasing: the workgroup that shares its CU with an older one starts `delay_unit * nk` x 64 cycles late (about
  // half a tile), so that the two reach their epilogues half a tile apart.  Which workgroup that is: mode 1 = the upper
  // half of the grid (the dispatcher hands the first gridDim/2 workgroups one per CU), mode 2 = the wave landed in an odd
  // wave slot of its SIMD (HW_ID.WAVE_ID).  A wrong guess costs speed only.
  {
    bool late = false;
    if (delay_mode == 1) late = (int)blockIdx.x * 2 >= (int)gridDim.x;
    if (delay_mode == 2) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      late = hw & 1u;
    }
    if (late && (int)gridDim.x * 1 > 256) {
      for (int i = 0, n = delay_unit * nk; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }
  }

#ifdef P2_TRACE
  unsigned long long* tr_p = p2_trace_buf + (size_t)blockIdx.x * P2_TRACE_N;
  int tr_n = 0;
  {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (wave == 0) {
      unsigned long long t_ = (unsigned long long)hw | ((unsigned long long)xcc << 32);
      asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(t_), "s"(tr_p) : "memory");
      ++tr_n;
    }
  }
#endif
#ifdef P2_PRIO
  if ((int)blockIdx.x * 2 >= (int)gridDim.x) __builtin_amdgcn_s_setprio(3);   // probe: static priority for one of the two co-resident workgroups
#endif
  // ---- compute side
  const int r16 = lane & 15, quad = lane >> 4;
  const bool vec = !((a.N | a.ldo | (a.resid_kind ? a.ldr : 0)) & 3);
  const bool plain_rows = a.rpg == 0x7fffffff && a.resid_kind != 3;
  int la[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) la[ks] = r16 * 128 + ((((ks << 2) | quad) ^ ((r16 >> 1) & 7)) << 4);
  const char* lds = (const char*)smem;

  set_x_tile(xti);
  set_w_tile(wti);
  issue_w();                                       // W(0)  -> slots 2, 3
  issue_x(0);                                      // XA(0) -> 4, 5
  issue_x(1);                                      // XB(0) -> 6, 7
  issue_w();                                       // W(1)  -> 8, 9; the ring counter wraps to 0

  int c = 0;                                       // 6 t mod 10 of the K-tile being computed
  int epi_extra = 3;                               // vmcnt slack of a tile's first K-tile: epilogue stores + the 3 bias loads
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
    // bias of this tile (see gemm_pp_kernel: unconditional inline-asm loads the compiler does not track; they are older
    // than every DMA piece issued in the K loop and first read in the epilogue)
    int lane_s;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
    const float* bsrc = a.bias ? a.bias : (const float*)a.zero;
    const int nlast = a.bias ? a.N - 1 : 0;
    float bcol;
    {
      const int n = cn0 + wn * 64 + lane_s;
      const float* p = bsrc + (n < nlast ? n : nlast);
      asm volatile("global_load_dword %0, %1, off" : "=&v"(bcol) : "v"(p));
    }
    float4 bias4[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      int n = cn0 + wn * 64 + i * 32 + (lane_s & 7) * 4;
      n = n + 3 < nlast ? n : (nlast & ~3);
      const float* p = bsrc + (vec ? n : 0);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p));
      bias4[i] = make_float4(v[0], v[1], v[2], v[3]);
    }

    for (int kt = 0; kt < nk; ++kt) {
      int sW = c + 2 + wn, sXA = c + 4 + wm, sXB = c + 6 + wm;
      if (sW >= P2_SLOTS) sW -= P2_SLOTS;
      if (sXA >= P2_SLOTS) sXA -= P2_SLOTS;
      if (sXB >= P2_SLOTS) sXB -= P2_SLOTS;
      c = c + 6 >= P2_SLOTS ? c + 6 - P2_SLOTS : c + 6;
      const char* wreg = lds + sW * (P2_REG * 2);
      const char* xa = lds + sXA * (P2_REG * 2);
      const char* xb = lds + sXB * (P2_REG * 2);
      bf16x8 w[4][2], xf[4][2];

#define P2_WAIT()                                                                          \
  if (kt == 0 && epi_extra == 19) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");        \
  else if (kt == 0 && epi_extra == 35) asm volatile("s_waitcnt vmcnt(43)" ::: "memory");   \
  else if (kt == 0 && epi_extra == 3) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");    \
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                    \
  __builtin_amdgcn_sched_barrier(0);                                                       \
  __builtin_amdgcn_s_barrier();                                                            \
  asm volatile("" ::: "memory");                                                           \
  __builtin_amdgcn_sched_barrier(0)

      // ---- phase A: W(wn) x XA(wm) -> rows 0-63 of the wave's block
      P2_STAMP(1);
      P2_WAIT();
      P2_STAMP(2);
#ifndef P2_NODMA
      issue_x(0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w[i][ks] = p2_ld(wreg + i * 2048 + la[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[j][ks] = p2_ld(xa + j * 2048 + la[ks]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][ks], xf[j][ks], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);

      // ---- phase B: W(wn) (still in registers) x XB(wm) -> rows 64-127
      P2_STAMP(3);
      P2_WAIT();
      P2_STAMP(4);
#ifndef P2_NODMA
      issue_x(1);
      issue_w();
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xf[j][ks] = p2_ld(xb + j * 2048 + la[ks]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i][ks], xf[j][ks], acc[i][4 + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef P2_WAIT

    // ---- epilogue.  Staging = the slots of the last K-tile's XB regions (slot c + wm: dead once every wave has passed this
    // barrier; re-issued in phase A of the next K-tile, behind a barrier every wave reaches only after its epilogue).
    P2_STAMP(5);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifndef P2_NOEPI
    {
      int ss = c + wm;
      if (ss >= P2_SLOTS) ss -= P2_SLOTS;
      const unsigned stg = (unsigned)(size_t)(AS3 bf16_t*)smem + ss * (P2_REG * 2) + wn * STG_BYTES;
      int lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      int stores = 0;
      const int mw0 = cm0 + wm * 128, nw0 = cn0 + wn * 64;
      if (vec && plain_rows && cm0 + 256 <= a.M) {
        const bool full_n = cn0 + 128 <= a.N;      // no lane's store is predicated off
        const bool pack16 = a.resid_kind == 0 && a.out_kind == 0 && !((a.N | a.ldo) & 7);
        if (full_n) stores = pack16 && a.act <= 2 ? (a.out2 ? 8 * TM : 4 * TM) : 4 * TM * TN;   // 16 / 32 store instructions per wave
        if (TE && pack16 && a.out2)
          epilogue_pack16<TM, TN, 0>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out2);
        if (pack16 && a.act == 0)
          epilogue_pack16<TM, TN, 0>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);        // QKV
        else if (pack16 && a.act == 1)
          epilogue_pack16<TM, TN, 1>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);        // c_fc + QuickGELU
        else if (pack16 && a.act == 2)
          epilogue_pack16<TM, TN, 2>(acc, a, stg, mw0, nw0, lane_e, bcol, a.out);
        else if (a.resid_kind == 1 && a.act == 0 && a.out_kind == 1)
          epilogue_rows<TM, TN, 1, 0, 1>(acc, a, stg, mw0, nw0, lane_e, bias4);          // out_proj / c_proj into the fp32 stream
        else
          epilogue_rows<TM, TN, -1, -1, -1, TE>(acc, a, stg, mw0, nw0, lane_e, bias4);
      } else {
        epilogue_generic16<TM, TN>(acc, a, vec, mw0, nw0, lane_e);
      }
      epi_extra = stores + 3;
    }
#endif
    P2_STAMP(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing empty pieces must retire before the LDS is released
#ifdef P2_TRACE
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

}  // namespace

bool msclip_gemm_pp2_eligible(const msclip_gemm_desc* d) {
  if (d->mode != 0) return false;
  const long long tiles = (long long)((d->M + 255) / 256) * ((d->N + 127) / 128);
  return (long long)d->ldx * 2 * 256 + (long long)d->K * 2 < (1ll << 31) && (long long)d->ldw * 2 * 128 + (long long)d->K * 2 < (1ll << 31) &&
         tiles * ((d->M + 255) / 256) * P2_CG < (1ll << 32) && tiles < (1ll << 28);
}

#ifdef P2_TRACE
extern "C" int msclip_pp2_trace(void* out, int bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2_trace_buf), bytes) == hipSuccess ? 0 : -2;
}
#endif

void msclip_gemm_pp2_launch(const msclip_gemm_desc* d, hipStream_t st, int ncu) {
  static int delay_mode = -1, delay_unit = 0, per_cu = 2;
  if (delay_mode < 0) {
    const char* m = getenv("MSCLIP_PP2_DELAY");
    const char* u = getenv("MSCLIP_PP2_DELAY_UNIT");
    const char* c = getenv("MSCLIP_PP2_PER_CU");
    delay_mode = m ? atoi(m) : 1;
    delay_unit = u ? atoi(u) : 16;                 // x nk x 64 cycles: ~half a K = 768 tile
    per_cu = c ? atoi(c) : 2;
  }
  const int tiles = ((d->M + 255) / 256) * ((d->N + 127) / 128);
  const int cap = per_cu * ncu;
  hipLaunchKernelGGL(gemm_pp2_kernel, dim3(tiles < cap ? tiles : cap), dim3(256), 0, st, *d, delay_mode, delay_unit);
}
