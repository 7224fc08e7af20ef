// Fused in_proj (q|k|v projection) + scaled-dot-product attention: one kernel, q|k|v never reach HBM.
// Replaces F.linear(x, in_proj_weight, in_proj_bias) + the bmm / mask / softmax / bmm of Attention_CUST
// (reference lib/models/clip_openai_pe_res_v1.py:612, 707-738; causal mask of :2965-2971 for the captions).
//
// What BASELINE.json's north_star names ("fused QKV-projection + SDPA"), built in round 5 (engine: MSCLIP_FUSED_QKV_ATTN;
// DESIGN.md s0 item 4 has the numbers).  The attention needs one head's q|k|v of whole samples in ONE workgroup:
//   * a tile is 256 token rows (whole samples: 5 images of 50 tokens, or a run of packed captions) x 192 columns (one head's
//     q | k | v: the packed weight is re-ordered head-major), contracted over K = width: 8 waves 4 x 2, 32x32x16 MFMAs, K-slabs
//     of 64 filled by LDS-DMA with counted waits, X two slabs ahead and W one (see the LDS map below);
//   * the tile's accumulators (+ bias, or the LayerNorm fold's rstd / mean / column-sum form) are written to LDS as bf16, three
//     areas of 256 rows x 128 B (q, k, v) in the slabs the tile has consumed -- the ping-pong kernel's LDS ring has no room
//     for that, which is why this kernel has a main loop of its own;
//   * wave w then attends the tile's query rows 32 w .. 32 w + 31 against the key tiles that its rows' samples span, with a
//     block-diagonal (+ causal) mask from a per-row (sample start, sample end) table: S^T = K Q^T and O^T = V^T P^T on
//     32x32x16 MFMAs as in attention.hip, V^T fragments straight from the row-major V rows by ds_read_b64_tr_b16.
// Rows of a tile behind its last whole sample belong to the next tile: computed, never stored.
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int BK = 64, BM = 256, BN = 192, NW = 8, NTH = 512;
constexpr int TM = 2, TN = 3;                      // 32 x 32 accumulator tiles per wave: 64 rows x 96 columns
constexpr int XI = BM * 8 / NTH, WI = BN * 8 / NTH;   // LDS-DMA pieces per lane and K-slab: 4 (X rows) + 3 (W rows)
// LDS map (160 KB): three X slabs of 32 KB (256 rows x 128 B), two W slabs of 24 KB (192 rows x 128 B) with 16 KB between them.
// The X operand runs TWO K-slabs ahead of the MFMAs, W one: 88 KB in flight instead of the two-buffer loop's 56 KB (the loop is
// bound by the latency of the LDS-DMA round trip: bytes in flight / latency).  At a tile's end two X slabs and one W slab + the
// 16 KB beside it are free: three 32 KB areas = the staged q, k and v of the tile (256 rows x 128 B each, same chunk swizzle as
// the operand slabs), while the third X slab and the other W slab already hold the next tile's first K-slab.
constexpr int XB_BYTES = BM * BK * 2, WB_BYTES = BN * BK * 2, SPARE_BYTES = 16384;
constexpr int W0_OFF = 3 * XB_BYTES, W1_OFF = W0_OFF + WB_BYTES + SPARE_BYTES;
static_assert(W1_OFF + WB_BYTES == 163840 && WB_BYTES + SPARE_BYTES >= XB_BYTES, "LDS map");

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;

__device__ __forceinline__ bf16x8 ld_tr8(const char* p0, const char* p1) {   // tokens T0..T0+3 (p0) and T0+8..T0+11 (p1) of this lane's channel
  const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)p0);
  const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// byte offset of (row, 16-byte chunk) in a 128-byte-row area with the operand slabs' swizzle
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// The attention of one staged tile (areas of q, k, v: 128 bytes per row, chunk-swizzled): the workgroup's waves take the tile's
// SAMPLES round-robin (sample = rows cu[s] .. cu[s + 1] of the token matrix, at most 96), one wave per (sample, head) like
// attention.hip's attn_kernel but with every operand in LDS: per 32-query tile S^T = K Q^T over the sample's own key tiles only
// (the causal ones for a caption), softmax in registers, O^T = V^T P^T with the V^T fragments read straight from the row-major
// v rows by ds_read_b64_tr_b16.  (First version: wave w took the tile's rows 32 w .. 32 w + 31 whatever samples they belonged
// to, with a block-diagonal mask -- a 32-row slice of 50-row samples spans two of them, so 60 % of its scores were masked and all
// eight waves shared the SIMDs' VALU: ~4 us per tile and head against ~1 us here.)
__device__ __forceinline__ void attend_samples(const char* sq, const char* sk, const char* sv, const int* __restrict__ cu, int s0,
                                               int s1, int causal_from_row, bf16_t* __restrict__ out, int ldo, int m0, int area_rows,
                                               int chead, int wave, int nwaves, int lane) {
  const int fr = lane & 31, fhi = lane >> 5;
  const int li = lane & 15, dh = (lane >> 4) & 1;
  const int vtok = 4 * fhi + (li >> 2), vch = 2 * dh + ((li & 3) >> 1), vhalf = (li & 1) * 8;
  for (int smp = s0 + wave; smp < s1; smp += nwaves) {
    const int c0 = __builtin_amdgcn_readfirstlane(cu[smp]);
    const int L = min(__builtin_amdgcn_readfirstlane(cu[smp + 1]) - c0, 96);
    const int r0 = c0 - m0;                        // first row of the sample inside the staged tile
    const bool causal = c0 >= causal_from_row;
    const int ntq = (L + 31) >> 5;
    bf16x8 kf[3][4];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
      if (kt < ntq) {
        const int krow = r0 + min(kt * 32 + fr, L - 1);      // clamped: padded keys are masked
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kt][kk] = *(const bf16x8*)(sk + swz(krow, kk * 2 + fhi));
      }
#pragma unroll
    for (int qt = 0; qt < 3; ++qt)
      if (qt < ntq) {
        const int q = qt * 32 + fr;
        const int qrow = r0 + min(q, L - 1);       // clamped: padded queries are never stored
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(sq + swz(qrow, kk * 2 + fhi));
        const int nkt = causal ? qt + 1 : ntq;
        f32x16 s[3];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][kk], qf[kk], s[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
              const bool ok = key < L && (!causal || key <= q);
              s[kt][r] = ok ? s[kt][r] : -INFINITY;
              mx = fmaxf(mx, s[kt][r]);
            }
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float p = __expf(s[kt][r] - mx);
              s[kt][r] = p;
              sum += p;
            }
          }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        // V^T fragment of k-step (key0 .. key0 + 15): lane (d = dt*32 + lane%32, half fhi) needs keys key0 + 4 fhi + {0..3, 8..11};
        // a 16-lane group reads a 4-token x 16-channel block, lane i supplying (token i / 4, channels 4 (i % 4) ..) and receiving
        // channel i: channels dt*32 + 16 dh + 4 (i % 4) = 16-byte chunk dt*4 + 2 dh + (i % 4) / 2, half (i % 4) % 2.  Token rows
        // behind the staged area belong to masked keys (P = 0): clamped to the area's last row.
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
          if (kt < nkt) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              bf16x8 pf;
#pragma unroll
              for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[kt][half * 8 + e];
              const int t0 = min(r0 + kt * 32 + half * 16 + vtok, area_rows - 1);
              const int t1 = min(r0 + kt * 32 + half * 16 + vtok + 8, area_rows - 1);
#pragma unroll
              for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vf = ld_tr8(sv + swz(t0, dt * 4 + vch) + vhalf, sv + swz(t1, dt * 4 + vch) + vhalf);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
              }
            }
          }
        bf16_t* orow = out + (size_t)(c0 + min(q, L - 1)) * ldo + chead * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          unsigned pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            pk[g][0] = pack_bf16x2(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
            pk[g][1] = pack_bf16x2(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {         // 8 consecutive d per lane (see attention.hip)
            const auto x = __builtin_amdgcn_permlane32_swap(pk[2 * s2][0], pk[2 * s2 + 1][0], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(pk[2 * s2][1], pk[2 * s2 + 1][1], false, false);
            if (q < L) *(uint4*)(orow + dt * 32 + s2 * 16 + fhi * 8) = make_uint4(x[0], y[0], x[1], y[1]);
          }
        }
      }
  }
}

__global__ __launch_bounds__(NTH) void qkv_attn_kernel(const msclip_qkvattn_desc a) {
  __shared__ __attribute__((aligned(1024))) char lds[163840];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.heads;
  const int ntiles = a.ntiles_dev ? __builtin_amdgcn_readfirstlane(*a.ntiles_dev) : a.ntiles;
  const int nwork = ntiles * H;
  const int Mtot = a.M;
  const int nk = a.K / BK;                         // >= 3 (host-checked)

  const bf16_t* __restrict__ X = (const bf16_t*)a.X;
  auto xbuf = [&](int i) -> char* { return lds + i * XB_BYTES; };
  auto wbuf = [&](int i) -> char* { return lds + (i ? W1_OFF : W0_OFF); };

  // work item -> (tile, head): XCD-aware bijective remap (consecutive ids -- the heads of one tile, which share its X rows --
  // run on one XCD), as gemm_kernel's tile_origin
  auto work_origin = [&](int t, int& m0, int& smp0, int& smp1, int& head) {
    const int q = nwork >> 3, r = nwork & 7, x = t & 7;
    const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
    const int tile = id / H;
    head = id - tile * H;
    smp0 = a.tile_first[tile];
    smp1 = a.tile_first[tile + 1];
    m0 = a.cu[smp0];
  };

  // loader: lane owns (row = (i * 8 + wave) * 8 + lane / 8, physical 16-byte chunk lane % 8 = logical chunk ^ ((row >> 1) & 7)).
  // Buffer-addressed LDS-DMA (SGPR descriptor of the tile's rows + 32-bit lane offset + scalar K offset), like the ping-pong
  // kernel: rows past the matrix are out of the descriptor's range (read as zero), and -- unlike global_load_lds, behind which
  // hipcc guards every LDS read with s_waitcnt vmcnt(0) -- the counted waits below stay what they are.
  const int pc = lane & 7;
  const int lc = pc ^ ((lane >> 4) | ((wave & 1) << 2));
  const int rsub = lane >> 3;
  unsigned vx[XI], vw[WI];
#pragma unroll
  for (int i = 0; i < XI; ++i) vx[i] = (unsigned)((i * NW + wave) * 8 + rsub) * (unsigned)a.ldx * 2u + (unsigned)lc * 16u;
#pragma unroll
  for (int i = 0; i < WI; ++i) vw[i] = (unsigned)((i * NW + wave) * 8 + rsub) * (unsigned)a.ldw * 2u + (unsigned)lc * 16u;
  auto x_rsrc = [&](int m0, bool live) {
    const long long bytes = live ? (long long)(Mtot - m0) * a.ldx * 2 : 0;
    return make_rsrc((const char*)X + (size_t)m0 * a.ldx * 2, bytes > 0xffffffffll ? 0xffffffffu : (unsigned)bytes);
  };
  auto w_rsrc = [&](int m0, int head, bool live) {
    const char* Wseg = (const char*)((a.W2 && m0 >= a.seg_split) ? a.W2 : a.W);
    return make_rsrc(Wseg + (size_t)head * BN * a.ldw * 2, live ? (unsigned)(BN * a.ldw * 2) : 0u);
  };
  auto issue_x = [&](__amdgpu_buffer_rsrc_t r, int kt, char* dst) {      // 4 pieces: the tile's 256 X rows of K-slab kt
#pragma unroll
    for (int i = 0; i < XI; ++i) blds16(r, vx[i], (unsigned)kt * 128u, dst + (i * NW + wave) * 1024);
  };
  auto issue_w = [&](__amdgpu_buffer_rsrc_t r, int kt, char* dst) {      // 3 pieces: the head's 192 W rows of K-slab kt
#pragma unroll
    for (int i = 0; i < WI; ++i) blds16(r, vw[i], (unsigned)kt * 128u, dst + (i * NW + wave) * 1024);
  };

  const int wm = (wave >> 1) * (TM * 32), wn = (wave & 1) * (TN * 32);
  const int fr = lane & 31, fsw = (lane >> 1) & 7, fhi = lane >> 5;

  int t = blockIdx.x;
  int m0 = 0, smp0 = 0, smp1 = 0, head = 0;
  int xc = 0, wc = 0;                              // buffers of X slab 0 / W slab 0 of the current tile
  __amdgpu_buffer_rsrc_t rx = x_rsrc(0, false), rw = w_rsrc(0, 0, false);
  if (t < nwork) {
    work_origin(t, m0, smp0, smp1, head);
    rx = x_rsrc(m0, true);
    rw = w_rsrc(m0, head, true);
    issue_x(rx, 0, xbuf(0));
    issue_w(rw, 0, wbuf(0));
  }
  for (; t < nwork; t += gridDim.x) {
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int cm0 = m0, csmp0 = smp0, csmp1 = smp1, chead = head;
    const bool has_next = t + (int)gridDim.x < nwork;
    // the next work item's origin is fetched HERE (two dependent loads) and pinned in scalar registers: its use near the end of
    // the K loop must not turn into a wait for every LDS-DMA piece in flight
    int nm0 = 0, nsmp0 = 0, nsmp1 = 0, nhead = 0;
    if (has_next) work_origin(t + gridDim.x, nm0, nsmp0, nsmp1, nhead);
    nm0 = __builtin_amdgcn_readfirstlane(nm0);
    nsmp0 = __builtin_amdgcn_readfirstlane(nsmp0);
    nsmp1 = __builtin_amdgcn_readfirstlane(nsmp1);
    nhead = __builtin_amdgcn_readfirstlane(nhead);
    int xi = xc, wi = wc;                          // buffers of the slab being computed
    for (int kt = 0; kt < nk; ++kt) {
      // X(kt), W(kt) have landed: at a tile's first slab nothing younger is in flight (and the barrier also orders the previous
      // tile's staging reads before the LDS-DMA into those areas); later the 4 pieces of X(kt + 1) are
      if (kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // (raw: __syncthreads()' fence would drain the VM counter, i.e. every piece in flight)
      asm volatile("" ::: "memory");
      const int x1 = xi == 2 ? 0 : xi + 1, x2 = x1 == 2 ? 0 : x1 + 1;
      const bf16_t* xs = (const bf16_t*)xbuf(xi) + (wm + fr) * BK;
      const bf16_t* ws = (const bf16_t*)wbuf(wi) + (wn + fr) * BK;
      bf16x8 wf[2][TN], xf[2][TM];
      {
        const int ph = (fhi ^ fsw) * 8;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[0][i] = *(const bf16x8*)(ws + i * 32 * BK + ph);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[0][j] = *(const bf16x8*)(xs + j * 32 * BK + ph);
      }
      // ---- issued under this slab's MFMAs, in this order (the counted waits depend on it):
      //   kt == 0:        X(1)                        (a tile's second slab: one slab of lead only)
      //   always:         W(kt + 1)   | at the last slab: W(0) of the next tile
      //   kt + 2 < nk:    X(kt + 2)   | kt == nk - 2: X(0) of the next tile | last slab: nothing (both X areas become staging)
      if (kt == 0) issue_x(rx, 1, xbuf(x1));
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
        if (kk == 0) {
          if (kt + 1 < nk) {
            issue_w(rw, kt + 1, wbuf(wi ^ 1));
          } else {                                 // the next tile's W(0) (an empty descriptor behind the last work item)
            m0 = nm0; smp0 = nsmp0; smp1 = nsmp1; head = nhead;
            rw = w_rsrc(m0, head, has_next);
            issue_w(rw, 0, wbuf(wi ^ 1));
          }
        }
        if (kk == 1) {
          if (kt + 2 < nk) {
            issue_x(rx, kt + 2, xbuf(x2));
          } else if (kt + 2 == nk) {               // X(0) of the next tile: the X descriptor moves on one slab before the W one
            rx = x_rsrc(nm0, has_next);
            issue_x(rx, 0, xbuf(x2));
          }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], xf[kk & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      xi = x1;
      wi ^= 1;
    }
    // after the loop: xi = buffer of the next tile's X(0) (= (xc + nk) % 3), wi = buffer of its W(0)
    const int xfree1 = xi == 0 ? 2 : xi - 1;       // X(nk - 1): consumed last
    const int xfree2 = xfree1 == 0 ? 2 : xfree1 - 1;   // X(nk - 2): consumed before; X(nk + 1) was never issued
    xc = xi;
    wc = wi;
    __syncthreads();                               // every wave is done with the last slab: its areas become the staging

    // ---- staging: q | k | v of this tile and head as bf16, three areas of 256 rows x 128 B (swizzled like the operand slabs)
    char* sq = xbuf(xfree1);
    char* sk = xbuf(xfree2);
    char* sv = (wi ^ 1) ? lds + W0_OFF + WB_BYTES : lds + W0_OFF;     // free W slab 1: [spare | W1]; free W slab 0: [W0 | spare]
    {
      const bool seg2 = a.W2 && cm0 >= a.seg_split;
      const float* bias = seg2 ? a.bias2 : a.bias;
      const float* csum = a.rowstat ? (seg2 ? a.csum2 : a.csum) : nullptr;
      float rs[TM], sh[TM];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = cm0 + wm + tm * 32 + fr;
        if (a.rowstat && m < Mtot) {
          const float2 st = *(const float2*)(a.rowstat + 2 * (size_t)m);
          rs[tm] = st.x; sh[tm] = st.y;
        } else {
          rs[tm] = 1.f; sh[tm] = 0.f;
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int cb = wn + tn * 32;               // first column of this 32-column block within the head's 192 (wave-uniform)
        char* area = cb < 64 ? sq : cb < 128 ? sk : sv;
        const int c0 = cb & 63;                    // column within the 64-column area
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb + g * 8 + fhi * 4;
          const float4 b4 = *(const float4*)(bias + chead * BN + c);
          float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (csum) c4 = *(const float4*)(csum + chead * BN + c);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            const int row = wm + tm * 32 + fr;
            const float v0 = acc[tn][tm][g * 4 + 0] * rs[tm] + (b4.x - sh[tm] * c4.x);
            const float v1 = acc[tn][tm][g * 4 + 1] * rs[tm] + (b4.y - sh[tm] * c4.y);
            const float v2 = acc[tn][tm][g * 4 + 2] * rs[tm] + (b4.z - sh[tm] * c4.z);
            const float v3 = acc[tn][tm][g * 4 + 3] * rs[tm] + (b4.w - sh[tm] * c4.w);
            uint2 o;
            o.x = pack_bf16x2(v0, v1);
            o.y = pack_bf16x2(v2, v3);
            *(uint2*)(area + swz(row, (c0 >> 3) + g) + fhi * 8) = o;      // columns c0 + 8 g + 4 fhi .. + 3: chunk c0 / 8 + g, half fhi
          }
        }
      }
    }
    __syncthreads();

    attend_samples(sq, sk, sv, a.cu, csmp0, csmp1, a.causal_from_row, (bf16_t*)a.out, a.ldo, cm0, BM, chead, wave, NW, lane);
    // (the next tile's first barrier orders these staging reads before the LDS-DMA into the areas)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// rowseg[m] = (first row, end row) of the sample that owns row m; tile_first / *ntiles: greedy packing of whole samples into tiles
// of at most max_rows rows, a tile never straddling `split_sample` (the image / text boundary: own LayerNorm parameters per modality).
__global__ __launch_bounds__(256) void qkvattn_tables_kernel(const int* __restrict__ cu, int nsamples, int split_sample,
                                                             int* __restrict__ rowseg, int* __restrict__ tile_first,
                                                             int* __restrict__ ntiles, int max_tiles, int max_rows) {
  const int total = cu[nsamples];
  for (int m = blockIdx.x * 256 + threadIdx.x; m < total; m += gridDim.x * 256) {
    int lo = 0, hi = nsamples;                     // largest s with cu[s] <= m
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cu[mid] <= m) lo = mid; else hi = mid;
    }
    rowseg[2 * m] = cu[lo];
    rowseg[2 * m + 1] = cu[lo + 1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int nt = 0, s = 0;
    while (s < nsamples && nt < max_tiles) {
      tile_first[nt++] = s;
      const int base = cu[s];
      int e = s + 1;
      while (e < nsamples && cu[e + 1] - base <= max_rows && e != split_sample) ++e;
      s = e;
    }
    tile_first[nt] = nsamples;
    *ntiles = s < nsamples ? -1 : nt;              // -1: max_tiles too small (the launch then does nothing)
  }
}

}  // namespace

extern "C" int msclip_qkvattn_tables(const int* cu, int nsamples, int split_sample, int* rowseg, int* tile_first, int* ntiles,
                                     int max_tiles, int max_rows, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_qkvattn_tables, stream, cu, nsamples, split_sample, rowseg, tile_first, ntiles, max_tiles, max_rows);
  if (!cu || !rowseg || !tile_first || !ntiles || nsamples <= 0 || max_tiles <= 0 || (max_rows != 256 && max_rows != 128))
    return MSCLIP_EINVAL;
  hipLaunchKernelGGL(qkvattn_tables_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, cu, nsamples, split_sample, rowseg,
                     tile_first, ntiles, max_tiles, max_rows);
  return msclip_launch_status();
}

extern "C" int msclip_qkv_attention(const msclip_qkvattn_desc* d, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_qkv_attention, stream, d);
  if (!d || !d->X || !d->W || !d->bias || !d->out || !d->cu || !d->tile_first || !d->rowseg || !d->zero) return MSCLIP_EINVAL;
  if (d->heads <= 0 || d->K <= 0 || (d->K % BK) || d->K < 3 * BK || (d->ldx % 8) || (d->ldw % 8) || d->ldw < d->K || (d->ldo % 8) ||
      d->ldo < d->heads * 64 || d->M <= 0 || (!d->ntiles_dev && d->ntiles <= 0))
    return MSCLIP_EINVAL;
  if (d->rowstat && !d->csum) return MSCLIP_EINVAL;
  if (d->W2 && (!d->bias2 || (d->rowstat && !d->csum2) || d->seg_split <= 0)) return MSCLIP_EINVAL;
  if (((size_t)d->bias | (size_t)d->csum | (size_t)d->bias2 | (size_t)d->csum2) & 15) return MSCLIP_EINVAL;
  const int ncu = msclip_device_cus();
  int grid = ncu;
  if (!d->ntiles_dev && d->ntiles * d->heads < grid) grid = d->ntiles * d->heads;
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(NTH), 0, (hipStream_t)stream, *d);
  return msclip_launch_status();
}
