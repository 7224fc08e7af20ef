// bf16 MFMA GEMM for gfx950 whose epilogue runs UNDER the next tile's main loop.
//
//   out[m, n] = act( sum_k X[m, k] * W[n, k] + bias[n] )        bf16 output, dense row-major X [M, K], W [N, K]
//
// Why: in the 8-wave ping-pong kernel (gemm.hip) every tile ends with all waves storing and the matrix pipe idle; a CU
// drains its 128 KiB of output at ~16 B/clk, 18 % of the whole model step (bench with a store-less build: 15.5 -> 12.7 ms).
// Hiding the stores needs the finished tile to live somewhere while the next one accumulates, and an 8-wave workgroup has
// neither the registers (2 waves per SIMD: 256 each, 242 in use) nor the LDS (160 KiB operand ring) for it.  Here a
// workgroup is 4 waves, ONE per SIMD with the full 512-register budget: a wave owns a 128 x 128 block of the 256 x 256
// tile (256 accumulator registers), and the finished block is packed to bf16 (128 registers, "carry") and stored in
// eight 16-row units during the first K-tiles of the next tile, between its MFMAs.
//
// Structure
//  * LDS: ring of ten 16-KiB regions; a region = one 128-row half of one operand for one 64-deep K-tile (full 128-byte
//    rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA source address and on the fragment read).  K-tile T
//    occupies ring positions 4T .. 4T+3 = X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255.  Filled by
//    buffer-addressed LDS-DMA (4 pieces of 8 rows per wave and region).
//  * W rows are stored PERMUTED inside their region: LDS row ni*16 + i holds W row (i >> 2)*32 + ni*4 + (i & 3) of the
//    half.  With the MFMA operands swapped (A = W, B = X) lane (r16, quad) of an accumulator then owns, for output row
//    r16, the 32 CONSECUTIVE columns quad*32 .. quad*32+31 (4 per 16-column MFMA tile ni) -- 64 contiguous bytes of bf16
//    -- instead of 8 scattered groups of 4, which is what makes a store without an LDS round trip possible.
//  * one s_barrier per K-tile ("sync(T)": K-tile T+1 has landed, K-tile T-1's slots are free).  The MFMA stream is
//    skewed against it by three 4-MFMA groups: an iteration runs the last three groups of K-tile T-1, then the first
//    29 of K-tile T, so the first fragment reads of a K-tile are issued under MFMAs that already have operands.
//    Per iteration (16 groups): 16 LDS-DMA pieces (W of K-tile T+1 first: short distance, L2-resident; then X of
//    K-tile T+2), 32 fragment reads (W fragments of one 32-deep k-step resident in registers, X fragments streamed
//    through a 4-deep register ring two groups ahead), one epilogue unit.
//  * epilogue unit u (rows 16u .. 16u+15 of the wave's block, 16 packed registers): 4 x 4 transpose of 16-byte chunks
//    between the four lanes that share r16 >> 2 (two DPP butterfly stages), so that each of the 4 store instructions of
//    the unit writes 8 full 128-byte lines; unpack, + bias, activation, repack.  The first iteration of a tile packs the
//    previous tile's accumulators in front of its own first MFMAs (which start from C = 0).
//  * VM counter (in order): per iteration W pieces (8), X pieces (8), then the unit's 4 stores; sync waits vmcnt(8 + 4):
//    everything up to this iteration's W pieces has landed, the stores of an iteration get a whole K-tile to retire.
#include <type_traits>
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int WSLOTS = 10, WREG = 128 * 64;   // ring regions, bf16 elements per region
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum { W4_FIRST = 0, W4_UNIT = 1, W4_PLAIN = 2 };

__device__ __forceinline__ unsigned dpp_quad_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned dpp_quad_xor2(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false); }

// ---- accumulator file access by literal register number: accumulator tile A = (ni*8 + mj) lives in a[4A .. 4A+3]
template <int A, bool ZERO_C>
__device__ __forceinline__ void w4_mfma(const bf16x8& w, const bf16x8& x) {
  if (ZERO_C)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(x), "i"(A * 4), "i"(A * 4 + 3));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(x), "i"(A * 4), "i"(A * 4 + 3));
}
// accumulator tile -> two packed bf16 pairs (elements 0, 1 | 2, 3).  The MFMA that wrote it is >= 8 groups back.
template <int A>
__device__ __forceinline__ void w4_pack(unsigned& lo, unsigned& hi) {
  unsigned t0, t1, t2, t3;
  asm volatile("v_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7\n\tv_accvgpr_read_b32 %4, a%c8\n\t"
               "v_accvgpr_read_b32 %5, a%c9\n\tv_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5"
               : "=&v"(lo), "=&v"(hi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
               : "i"(A * 4), "i"(A * 4 + 1), "i"(A * 4 + 2), "i"(A * 4 + 3));
}
template <int A = 0>
__device__ __forceinline__ void w4_pack_all(unsigned (&carry)[8][16]) {
  if constexpr (A < 64) {
    constexpr int ni = A >> 3, mj = A & 7;
    w4_pack<A>(carry[mj][2 * ni], carry[mj][2 * ni + 1]);
    w4_pack_all<A + 1>(carry);
  }
}
// Tell the compiler the whole accumulator file is taken (kernel descriptor and its own allocation).
#define W4_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define W4_RESERVE_AGPRS()                                                                                                    \
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W4_A10(1), W4_A10(2), W4_A10(3), W4_A10(4),  \
               W4_A10(5), W4_A10(6), W4_A10(7), W4_A10(8), W4_A10(9), W4_A10(10), W4_A10(11), W4_A10(12), W4_A10(13),          \
               W4_A10(14), W4_A10(15), W4_A10(16), W4_A10(17), W4_A10(18), W4_A10(19), W4_A10(20), W4_A10(21), W4_A10(22),     \
               W4_A10(23), W4_A10(24), "a250", "a251", "a252", "a253", "a254", "a255")

template <int ACT>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const msclip_gemm_desc a) {
  __shared__ __attribute__((aligned(1024))) bf16_t smem[WSLOTS * WREG];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;           // this wave's 128-row / 128-column half of the tile
  const int nt_n = (a.N + 255) / 256;
  const int nt_m = (a.M + 255) / 256;
  const int ntiles = nt_n * nt_m;
  const int nk = a.K / 64;

  // tile id -> origin: XCD-aware remap, column groups of 4 tiles with rows fastest (same map as gemm_pp_kernel)
  constexpr int CG = 4;
  const unsigned tper = (unsigned)(nt_m * CG);
  const unsigned tper_rcp = 0xffffffffu / tper + 1u;
  const int wg_tail = nt_n - (nt_n - 1) / CG * CG;
  const unsigned wgt_rcp = 0xffffffffu / (unsigned)wg_tail + 1u;
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
    const unsigned id = (unsigned)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3));
    const unsigned g = (unsigned)(((unsigned long long)id * tper_rcp) >> 32);
    const unsigned idg = id - g * tper;
    const bool tail = (int)(g * CG + CG) > nt_n;
    const unsigned wg = tail ? (unsigned)wg_tail : (unsigned)CG;
    const unsigned row = !tail ? idg >> 2 : wg_tail == 1 ? idg : (unsigned)(((unsigned long long)idg * wgt_rcp) >> 32);
    m0 = (int)(row * 256u);
    n0 = (int)((g * CG + (idg - row * wg)) * 256u);
  };

  // ---- issue side.  Piece p (LDS rows 8p .. 8p+7 of a region) belongs to wave p & 3; lane -> LDS row 8p + lane/8,
  // physical chunk lane%8 <- logical chunk (lane%8) ^ ((row >> 1) & 7) of the source row.
  unsigned vx[4], vw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3);                       // LDS row inside the region
    const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) << 4;
    const int wrow = ((row & 15) >> 2) * 32 + (row >> 4) * 4 + (row & 3);   // W row of the half stored in that LDS row
    vx[i] = (unsigned)row * (unsigned)a.ldx * 2u + ch;
    vw[i] = (unsigned)wrow * (unsigned)a.ldw * 2u + ch;
  }
  const unsigned xhalf = 128u * (unsigned)a.ldx * 2u, whalf = 128u * (unsigned)a.ldw * 2u;
  // two cursors: the W stream runs one K-tile ahead of the compute, the X stream two
  int tiw = blockIdx.x, ktw = 0, tix = blockIdx.x, ktx = 0, ipos = 0;      // ipos: ring slot of the next region to issue
  __amdgpu_buffer_rsrc_t rx, rw;
  auto set_x = [&](int t) {
    if (t < ntiles) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const unsigned long long xb = (unsigned long long)(a.M - m0) * (unsigned long long)a.ldx * 2ull;
      rx = make_rsrc((const bf16_t*)a.X + (size_t)m0 * a.ldx, xb > 0xffffffffull ? 0xffffffffu : (unsigned)xb);
    } else {
      rx = make_rsrc(a.X, 0);                        // past the tile list: empty descriptor, the counts stay exact
    }
  };
  auto set_w = [&](int t) {
    if (t < ntiles) {
      int m0, n0;
      tile_origin(t, m0, n0);
      const unsigned long long wb = (unsigned long long)(a.N - n0) * (unsigned long long)a.ldw * 2ull;
      rw = make_rsrc((const bf16_t*)a.W + (size_t)n0 * a.ldw, wb > 0xffffffffull ? 0xffffffffu : (unsigned)wb);
    } else {
      rw = make_rsrc(a.W, 0);
    }
  };
  // piece q (0..15) of the iteration: q 0-7 = W halves 0, 1 of the W cursor's K-tile, q 8-15 = X halves of the X cursor's
  auto issue_piece = [&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    constexpr int reg = Q >> 2, i = Q & 3;           // region of the iteration (0, 1: W; 2, 3: X), piece of the wave
    int slot = ipos + reg;
    if (slot >= WSLOTS) slot -= WSLOTS;
    bf16_t* dst = smem + slot * WREG + (wave + 4 * i) * 512;
    if (reg < 2) blds16(rw, vw[i], (unsigned)ktw * 128u + (reg & 1) * whalf, dst);
    else blds16(rx, vx[i], (unsigned)ktx * 128u + (reg & 1) * xhalf, dst);
  };
  auto advance_cursors = [&]() {                     // end of an iteration: 4 regions issued
    ipos = ipos + 4 >= WSLOTS ? ipos + 4 - WSLOTS : ipos + 4;
    if (++ktw == nk) { ktw = 0; tiw += gridDim.x; set_w(tiw); }
    if (++ktx == nk) { ktx = 0; tix += gridDim.x; set_x(tix); }
  };

  // ---- compute side: fragments of v_mfma_f32_16x16x32_bf16, lane l = (r16 = l % 16, quad = l / 16) holds row r16,
  // 16-byte chunk quad (+4 for k-step 1) of a 16-row tile; la[ks] = that chunk's byte offset inside the tile's LDS rows
  const int r16 = lane & 15, quad = lane >> 4;
  int la[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) la[ks] = r16 * 128 + ((((ks << 2) | quad) ^ ((r16 >> 1) & 7)) << 4);
  const char* lds = (const char*)smem;

  // ---- prologue: X of K-tile 0 (ring 0, 1), W of K-tile 0 (2, 3), X of K-tile 1 (4, 5)
  set_x(tix);
  set_w(tiw);
  {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) blds16(rx, vx[i], h * xhalf, smem + (0 + h) * WREG + (wave + 4 * i) * 512);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) blds16(rw, vw[i], h * whalf, smem + (2 + h) * WREG + (wave + 4 * i) * 512);
    if (++ktw == nk) { ktw = 0; tiw += gridDim.x; set_w(tiw); }
    if (++ktx == nk) { ktx = 0; tix += gridDim.x; set_x(tix); }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) blds16(rx, vx[i], (unsigned)ktx * 128u + h * xhalf, smem + (4 + h) * WREG + (wave + 4 * i) * 512);
    if (++ktx == nk) { ktx = 0; tix += gridDim.x; set_x(tix); }
    ipos = 6;
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // K-tile 0 landed (X of K-tile 1 may still fly)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Accumulators: [16-column tile ni][16-row tile mj] of the wave's 128 x 128 block = a[(ni*8+mj)*4 .. +3], named literally
  // in the MFMA / read statements.  hipcc's allocator cannot hold 256 accumulators + 128 carry registers + fragments across
  // this kernel's control flow (it rotates accumulators through v_accvgpr_mov and scratch: 200-1100 spills in every
  // formulation tried); with the accumulator file out of its hands it only manages the <= 256 VGPRs.  The first K-tile
  // of the kernel restarts every accumulator from C = 0, so the file needs no zeroing.
  W4_RESERVE_AGPRS();
  bf16x8 WA[4], WB[4], XS[4];                        // W fragments of the running / the next phase (4 tiles each), X fragment ring
  unsigned carry[8][16];                             // previous tile, packed bf16: [row tile][2 * ni + (r >> 1)]
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 16; ++k) carry[j][k] = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) WA[i] = WB[i] = XS[i] = bf16x8{};

  int cpos = 0;                                      // ring slot of region 0 of the K-tile this iteration reads
  __amdgpu_buffer_rsrc_t ro = make_rsrc(a.out, 0);   // output rows of the CARRY's tile (empty: nothing to store yet)
  f32x4 biasv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // bias of this lane's 8 epilogue columns
  // epilogue lane constants: after the chunk transpose lane (arow = r16 / 4, b = r16 % 4, quad) stores, for slot s,
  // row 4*arow + s of the unit, columns quad*32 + b*8 .. +7 of the wave's 128
  const int eb = lane & 3, earow = (lane & 15) >> 2;
  const unsigned ovoff = (unsigned)(((wr * 128 + earow * 4) * a.ldo + wc * 128 + quad * 32 + eb * 8) * 2);
  const unsigned ldo2 = (unsigned)a.ldo * 2u;
  const char* xreg = lds;                            // X half wr / W half wc of the K-tile being read
  const char* wreg = lds;

  // The carry now holds tile (m0, n0): output descriptor (rows past M out of range -> dropped) and bias request.
  // The bias loads are inline asm (the compiler must not wait for them) and older than every DMA piece of the
  // iteration, whose sync retires them; first used one iteration later.
  auto retarget_epilogue = [&](int m0, int n0, bool valid) {
    const unsigned long long ob = (unsigned long long)(a.M - m0) * (unsigned long long)a.ldo * 2ull - (unsigned long long)n0 * 2ull;
    ro = valid ? make_rsrc((bf16_t*)a.out + (size_t)m0 * a.ldo + n0, ob > 0xffffffffull ? 0xffffffffu : (unsigned)ob)
               : make_rsrc(a.out, 0);
    const float* bp = a.bias ? a.bias + n0 + wc * 128 + quad * 32 + eb * 8 : (const float*)a.zero;
    const float* bp2 = a.bias ? bp + 4 : bp;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(biasv[0]) : "v"(bp));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(biasv[1]) : "v"(bp2));
  };

  // One epilogue unit (rows 16u .. 16u+15 of the carry -> 4 store instructions of 8 full lines each), cut into 32
  // slices, one per MFMA group: slices 0-7 = first butterfly stage of the 4 x 4 chunk transpose between the lanes
  // b = 0..3 (2 registers each), 8-15 = second stage, 16-31 = one packed register each (unpack, + bias, activation,
  // repack); a slot's store goes out with its 4th register.  All stores are younger than the iteration's W pieces.
  unsigned et[16], ev[16];
  u32x4 eo;
  auto unit_slice = [&](auto uc, auto gc) {
    constexpr int U = decltype(uc)::value;
    constexpr int G = decltype(gc)::value;
    if constexpr (G < 8) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = G * 2 + e, sl = idx >> 2, k = idx & 3;
        const unsigned other = dpp_quad_xor1(carry[U][(sl ^ 1) * 4 + k]);
        et[idx] = ((eb ^ sl) & 1) ? other : carry[U][idx];
      }
    } else if constexpr (G < 16) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = (G - 8) * 2 + e, sl = idx >> 2, k = idx & 3;
        const unsigned other = dpp_quad_xor2(et[(sl ^ 2) * 4 + k]);
        ev[idx] = ((eb ^ sl) & 2) ? other : et[idx];
      }
    } else {
      constexpr int idx = G - 16, sl = idx >> 2, k = idx & 3;
      float v0 = __uint_as_float(ev[idx] << 16) + biasv[k >> 1][(2 * k) & 3];
      float v1 = __uint_as_float(ev[idx] & 0xffff0000u) + biasv[k >> 1][(2 * k + 1) & 3];
      if (ACT == 1) {
        // QuickGELU v * sigmoid(1.702 v) = v / (1 + 2^(-1.702 log2(e) v)).  The two register fences keep both inputs of
        // each transcendental pair live in their own registers until both are issued.  Without them hipcc emitted
        // "v_mul t, c, x0 ; v_exp d0, t ; v_mul t, c, x1 ; v_exp d1, t" (the VALU after a transcendental overwriting its
        // source) between two asm MFMAs, and exactly that register pair came out as 1e30-sized garbage on gfx950 (1 % of
        // the outputs).  The same adjacency is harmless in compiler-only code, so the asm MFMAs next to it (whose hazards
        // hipcc cannot pad) are the likely ingredient; the fence removes the pattern and the fault.
        float a0 = v0 * -2.45546696f, a1 = v1 * -2.45546696f;
        asm volatile("" : "+v"(a0), "+v"(a1));
        float d0 = 1.f + __builtin_amdgcn_exp2f(a0), d1 = 1.f + __builtin_amdgcn_exp2f(a1);
        asm volatile("" : "+v"(d0), "+v"(d1));
        v0 *= __builtin_amdgcn_rcpf(d0);
        v1 *= __builtin_amdgcn_rcpf(d1);
      } else if (ACT == 2) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
      }
      eo[k] = pack_bf16x2(v0, v1);
      if constexpr (k == 3)   // unconditional buffer store (the in-order VM count of the sync depends on it); nt: the tile leaves faster
        __builtin_amdgcn_raw_buffer_store_b128(eo, ro, (int)ovoff, (int)((unsigned)(U * 16 + sl) * ldo2), 2);
    }
  };
  auto store_unit = [&](auto uc) {                   // the whole unit at once (kernel tail)
#define W4_S(N) unit_slice(uc, std::integral_constant<int, N>{});
    W4_S(0) W4_S(1) W4_S(2) W4_S(3) W4_S(4) W4_S(5) W4_S(6) W4_S(7) W4_S(8) W4_S(9) W4_S(10) W4_S(11) W4_S(12) W4_S(13)
    W4_S(14) W4_S(15) W4_S(16) W4_S(17) W4_S(18) W4_S(19) W4_S(20) W4_S(21) W4_S(22) W4_S(23) W4_S(24) W4_S(25) W4_S(26)
    W4_S(27) W4_S(28) W4_S(29) W4_S(30) W4_S(31)
#undef W4_S
  };

  // One group = 4 MFMAs (one X fragment against the 4 resident W fragments) + its share of the iteration's other work.
  // A K-tile is 4 phases of 8 groups: (k-step 0, W tiles 0-3), (k-step 0, W tiles 4-7), (k-step 1, 0-3), (k-step 1, 4-7);
  // the W fragments of a phase sit in WA (phases 0, 2) or WB (1, 3) and are loaded during the phase before.  An
  // iteration is skewed by 3 groups = the X prefetch distance: G 0..2 = row tiles 5..7 of the PREVIOUS K-tile's phase 3
  // (operands already in registers, requested in G 29..31 of the previous iteration), G 3..31 = phases 0, 1, 2 and row
  // tiles 0..4 of phase 3 of this K-tile.  Every read of the iteration targets this K-tile's regions.
  // In program order (pinned by the sched_barriers; a wave issues in order, an MFMA occupies the pipe for 16 cycles after
  // its 4-cycle issue slot): every MFMA is followed by a piece of the group's other work.
  auto group = [&](auto modec, auto uc, auto gc) {
    constexpr int MODE = decltype(modec)::value;
    constexpr int G = decltype(gc)::value;
    constexpr int ph = G < 3 ? 3 : (G - 3) >> 3;                 // phase of this group's MFMAs
    constexpr int mj = G < 3 ? 5 + G : (G - 3) & 7;              // its row tile
    constexpr int h = ph & 1;
    constexpr int ni0 = h * 4;
    constexpr bool restart = MODE == W4_FIRST && ph < 2 && G >= 3;   // first two phases of a new tile: pack, then C = 0
#define W4_STEP(I)                                                                                          \
    if constexpr (restart) w4_pack<(ni0 + I) * 8 + mj>(carry[mj][2 * (ni0 + I)], carry[mj][2 * (ni0 + I) + 1]);   \
    w4_mfma<(ni0 + I) * 8 + mj, restart>(h ? WB[I] : WA[I], XS[G & 3]);
    W4_STEP(0)
    {  // X fragment of the group three ahead (in this K-tile: groups 0..2 of the next iteration are its phase 3 tail)
      constexpr int H = (G + 3) & 31;
      constexpr int phH = H < 3 ? 3 : (H - 3) >> 3;
      constexpr int mjH = H < 3 ? 5 + H : (H - 3) & 7;
      XS[H & 3] = *(const bf16x8*)(xreg + mjH * 2048 + la[phH >> 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    W4_STEP(1)
    // W fragments of the next phase during the first groups of the running one (G 0..2 load phase 0's four)
    if constexpr (G == 0) WA[3] = *(const bf16x8*)(wreg + 3 * 2048 + la[0]);
    if constexpr (G < 3) WA[G] = *(const bf16x8*)(wreg + G * 2048 + la[0]);
    else if constexpr (G < 7) WB[G - 3] = *(const bf16x8*)(wreg + (G + 1) * 2048 + la[0]);                 // phase 1: k-step 0, tiles 4-7
    else if constexpr (G >= 11 && G < 15) WA[G - 11] = *(const bf16x8*)(wreg + (G - 11) * 2048 + la[1]);   // phase 2: k-step 1, tiles 0-3
    else if constexpr (G >= 19 && G < 23) WB[G - 19] = *(const bf16x8*)(wreg + (G - 15) * 2048 + la[1]);   // phase 3: k-step 1, tiles 4-7
    if constexpr ((G & 1) == 0) issue_piece(std::integral_constant<int, (G >> 1)>{});   // 16 LDS-DMA pieces per iteration
    __builtin_amdgcn_sched_barrier(0);
    W4_STEP(2)
    if constexpr (MODE == W4_UNIT) unit_slice(uc, gc);
    __builtin_amdgcn_sched_barrier(0);
    W4_STEP(3)
#undef W4_STEP
    __builtin_amdgcn_sched_barrier(0);
  };

  auto iteration = [&](auto modec, auto uc) {
    constexpr int MODE = decltype(modec)::value;
    {
      int sx = cpos + wr, sw = cpos + 2 + wc;
      if (sx >= WSLOTS) sx -= WSLOTS;
      if (sw >= WSLOTS) sw -= WSLOTS;
      xreg = lds + sx * (WREG * 2);
      wreg = lds + sw * (WREG * 2);
    }
#define W4_G(N) group(modec, uc, std::integral_constant<int, N>{});
    W4_G(0) W4_G(1) W4_G(2) W4_G(3) W4_G(4) W4_G(5) W4_G(6) W4_G(7)
    W4_G(8) W4_G(9) W4_G(10) W4_G(11) W4_G(12) W4_G(13) W4_G(14) W4_G(15)
    W4_G(16) W4_G(17) W4_G(18) W4_G(19) W4_G(20) W4_G(21) W4_G(22) W4_G(23)
    W4_G(24) W4_G(25) W4_G(26) W4_G(27) W4_G(28) W4_G(29) W4_G(30) W4_G(31)
#undef W4_G
    advance_cursors();
    cpos = cpos + 4 >= WSLOTS ? cpos + 4 - WSLOTS : cpos + 4;
    // sync: everything up to this iteration's W pieces has landed (younger: its 8 X pieces and the unit's 4 stores)
    if (MODE == W4_UNIT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  using MF = std::integral_constant<int, W4_FIRST>;
  using MU = std::integral_constant<int, W4_UNIT>;
  using MP = std::integral_constant<int, W4_PLAIN>;
  using I0 = std::integral_constant<int, 0>;
  int lm0 = 0, ln0 = 0;
  bool have_prev = false;
  for (int tc = blockIdx.x; tc < ntiles; tc += gridDim.x) {
    // first K-tile: G 0..3 finish the previous tile, then its accumulators are packed into the carry
    retarget_epilogue(lm0, ln0, have_prev);
    iteration(MF{}, I0{});
    tile_origin(tc, lm0, ln0);
    have_prev = true;
    iteration(MU{}, std::integral_constant<int, 0>{});
    iteration(MU{}, std::integral_constant<int, 1>{});
    iteration(MU{}, std::integral_constant<int, 2>{});
    iteration(MU{}, std::integral_constant<int, 3>{});
    iteration(MU{}, std::integral_constant<int, 4>{});
    iteration(MU{}, std::integral_constant<int, 5>{});
    iteration(MU{}, std::integral_constant<int, 6>{});
    iteration(MU{}, std::integral_constant<int, 7>{});
    for (int kt = 9; kt < nk; ++kt) iteration(MP{}, I0{});
  }
  // ---- tail: row tiles 5..7 of the last K-tile's phase 3, then the last tile leaves un-overlapped
#define W4_TAIL(I, GG) w4_mfma<(4 + I) * 8 + 5 + GG, false>(WB[I], XS[GG]);
  W4_TAIL(0, 0) W4_TAIL(1, 0) W4_TAIL(2, 0) W4_TAIL(3, 0) W4_TAIL(0, 1) W4_TAIL(1, 1) W4_TAIL(2, 1) W4_TAIL(3, 1)
  W4_TAIL(0, 2) W4_TAIL(1, 2) W4_TAIL(2, 2) W4_TAIL(3, 2)
#undef W4_TAIL
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // last MFMA's D -> v_accvgpr_read: hipcc pads nothing around asm
  retarget_epilogue(lm0, ln0, true);
  w4_pack_all(carry);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(biasv[0]), "+v"(biasv[1])::"memory");   // bias arrived; trailing dummy pieces retired
  store_unit(std::integral_constant<int, 0>{});
  store_unit(std::integral_constant<int, 1>{});
  store_unit(std::integral_constant<int, 2>{});
  store_unit(std::integral_constant<int, 3>{});
  store_unit(std::integral_constant<int, 4>{});
  store_unit(std::integral_constant<int, 5>{});
  store_unit(std::integral_constant<int, 6>{});
  store_unit(std::integral_constant<int, 7>{});
}

}  // namespace

static int w4_cus() {
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    ncu = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return ncu;
}

// Shapes this kernel takes: dense X, bf16 output without residual / row scatter, bias + {none, QuickGELU, ReLU}, whole
// 256-column tiles, at least 9 K-tiles (the 8 epilogue units of a tile ride on K-tiles 1..8 of the next one).
bool msclip_gemm_w4_eligible(const msclip_gemm_desc* d) {
  if (d->mode != 0 || d->out_kind != 0 || d->resid_kind != 0) return false;
  if (d->rpg != 0x7fffffff || d->radd || d->roff) return false;
  if (d->act < 0 || d->act > 2 || d->alpha != 1.f) return false;
  if ((d->N % 256) || (d->K % 64) || d->K < 9 * 64) return false;
  if ((d->ldx % 8) || (d->ldw % 8) || (d->ldo % 8) || d->ldw < d->K) return false;
  const long long tiles = (long long)((d->M + 255) / 256) * (d->N / 256);
  if ((long long)d->ldx * 2 * 256 + (long long)d->K * 2 >= (1ll << 31)) return false;
  if ((long long)d->ldw * 2 * 256 + (long long)d->K * 2 >= (1ll << 31)) return false;
  if ((long long)d->ldo * 2 * 256 + 1024 >= (1ll << 31)) return false;
  if (tiles * ((d->M + 255) / 256) * 4 >= (1ll << 32)) return false;
  return true;
}

void msclip_gemm_w4_launch(const msclip_gemm_desc* d, hipStream_t st) {
  const int ncu = w4_cus();
  const int tiles = ((d->M + 255) / 256) * (d->N / 256);
  const int grid = tiles < ncu ? tiles : ncu;
  if (d->act == 1) hipLaunchKernelGGL(gemm_w4_kernel<1>, dim3(grid), dim3(256), 0, st, *d);
  else if (d->act == 2) hipLaunchKernelGGL(gemm_w4_kernel<2>, dim3(grid), dim3(256), 0, st, *d);
  else hipLaunchKernelGGL(gemm_w4_kernel<0>, dim3(grid), dim3(256), 0, st, *d);
}
