// Backward of the fused softmax(q k^T [+ causal]) v attention (forward: attention.hip; reference M.py:707-738): one
// workgroup of 4 waves per (sample, head), all five contractions on v_mfma_f32_16x16x32_bf16.  Up to 96 tokens (the
// 50-token image grid of ViT-B/32 and the 77-token captions) everything of the head is resident in LDS
// (attn_bwd_kernel, described here); 97-208 tokens (the 197-token grid of ViT-B/16) run the query-blocked form further
// down (attn_bwd_qb_kernel).
//
//   S  = Q K^T (q pre-scaled by the packed in_proj weight),  P = softmax(S),  O = P V            (recomputed / given)
//   dV = P^T dO      dP = dO V^T      dS = P o (dP - delta),  delta_q = sum_d dO[q][d] O[q][d]
//   dQ = dS K        dK = dS^T Q
//
// The MFMA computes D[i][j] = sum_k A[i][k] B[j][k] from two K-contiguous row operands, lane (j = lane % 16,
// quad = lane / 16) holding D[4*quad + r][j].  So S, dP are formed TRANSPOSED (A = K / V rows, B = Q / dO rows): a lane
// owns one query and 4 keys per tile, the softmax statistics of a query are an in-lane reduction plus two quad
// exchanges, and dS^T = P^T o (dP^T - delta) is element-wise in registers.  The three gradients contract over tokens,
// so their operands are the TRANSPOSED tensors: Q^T, K^T, dO^T are built while loading, P^T / dS / dS^T are written to
// LDS from the accumulator layout; the outputs come out as dV^T, dK^T, dQ^T tiles, i.e. 4 consecutive head-dim elements
// per lane and token: 8-byte global stores.
#include <type_traits>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int RS = 72;                      // row stride (elements) of the [token][64] images: 144 B, 16-B aligned

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
__device__ __forceinline__ bf16x8 ld_tr8(const char* p0, const char* p1) {   // tokens T0..T0+3 (p0) and T0+4..T0+7 (p1) of this lane's channel
  const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)p0);
  const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((AS3 bf16x4v*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Workgroup barrier for LDS hand-overs inside the persistent loop: the waves' LDS writes are complete (lgkmcnt), then the raw
// barrier.  __syncthreads() would also fence global memory, i.e. wait (vmcnt(0)) for the NEXT pair's prefetch that is meant to
// stay in flight under this pair's work.
#define LDS_BARRIER()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
  } while (0)

template <int NT16, bool CAUSAL, int BW>    // NT16 = padded length / 16 (4: 64 tokens, 6: 96 tokens); BW = waves per workgroup
__global__ __launch_bounds__(64 * BW) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                       const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv, int Lfix,
                                                       int H, int ldq, int ldo, const int* __restrict__ cu, int nsamples,
                                                       int pad_rows, int pblocks, float* __restrict__ csum_part) {
  constexpr int LP = NT16 * 16, LS = LP + 8;          // padded length, row stride of the [..][token] images
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  bf16_t* Q = sm;                                      // [LP][RS]
  bf16_t* K = Q + LP * RS;
  bf16_t* V = K + LP * RS;
  bf16_t* dO = V + LP * RS;
  // (rounds 3-4 kept transposed copies Q^T, K^T, dO^T [64][LS] beside the row-major images, built with 24 two-byte LDS writes per
  //  thread while loading: 23 of a launch's 118 us, and the 28 KB that made the footprint one workgroup per CU.  Phase 2 now reads
  //  its token-contracting operands from the row-major images with ds_read_b64_tr_b16.)
  bf16_t* P = dO + LP * RS;                            // [query][key]   (LP x LS)  (P^T / dS^T are read through ds_read_b64_tr_b16 too)
  bf16_t* dS = P + LP * LS;                            // [query][key]
  float* delta = (float*)(dS + LP * LS);               // [LP]
  // csum_part: the token sums of every phase-2 output tile (16 head-dim columns each), two buffers (pair parity): the in_proj
  // bias gradient's share of this (sample, head) leaves with the pair instead of a second pass over dqkv [M, 3 D]
  constexpr int TSUM = 3 * 4 * NT16 * 16;
  float* tsum = delta + LP;                            // [2][which][head-dim tile][token tile][16]
  int tbuf = 0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= pblocks) {
    // packed captions (msclip_attention_bwd_varlen): the workgroups behind the persistent ones zero the q | k | v gradient
    // rows of the tile padding [cu[nsamples], + pad_rows): the weight-gradient GEMM contracts over them
    const int r = (blockIdx.x - pblocks) * BW + wave;
    if (cu && r < pad_rows) {
      bf16_t* g = dqkv + (size_t)(cu[nsamples] + r) * ldq;
      for (int c = lane * 8; c < 3 * H * 64; c += 512) *(uint4*)(g + c) = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  // Round 5: PERSISTENT workgroups with a register prefetch.  The head's images fill 92-155 KB of LDS, so a CU holds one
  // workgroup, and with one (sample, head) pair per workgroup the ~3 us of its global loads were exposed in front of ~2.5 us of
  // work, 24-48 times per CU and launch.  Now a workgroup walks pairs blockIdx.x, + pblocks, ...: the five 16-byte pieces per
  // thread of pair i + 1 (q, k, v, dO, O) are requested right after pair i's images are in LDS and land under its two phases.
  constexpr int NI = (LP * 8 + 64 * BW - 1) / (64 * BW);      // load iterations per thread (1: 64 tokens, 2: 96)
  const int npairs = nsamples * H;
  uint4 pq[NI], pk[NI], pv[NI], pd[NI], po[NI];
  auto pair_rows = [&](int pair, int& c0, int& L, int& h) {
    const int b = pair / H;
    h = pair - b * H;
    c0 = cu ? cu[b] : b * Lfix;                       // packed captions: rows cu[b] .. cu[b + 1]
    L = cu ? min(cu[b + 1] - c0, LP) : Lfix;
  };
  auto fetch = [&](int c0, int L, int h) {
    const bf16_t* qb = qkv + (size_t)c0 * ldq + h * 64;
    const bf16_t* ob = o + (size_t)c0 * ldo + h * 64;
    const bf16_t* db = dout + (size_t)c0 * ldo + h * 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = tid + i * 64 * BW, r = idx >> 3, c = idx & 7;
      pq[i] = pk[i] = pv[i] = pd[i] = po[i] = make_uint4(0, 0, 0, 0);
      if (idx < LP * 8 && r < L) {
        pq[i] = *(const uint4*)(qb + (size_t)r * ldq + c * 8);
        pk[i] = *(const uint4*)(qb + (size_t)r * ldq + H * 64 + c * 8);
        pv[i] = *(const uint4*)(qb + (size_t)r * ldq + 2 * H * 64 + c * 8);
        pd[i] = *(const uint4*)(db + (size_t)r * ldo + c * 8);
        po[i] = *(const uint4*)(ob + (size_t)r * ldo + c * 8);
      }
    }
  };
  int c0 = 0, L = 0, h = 0;
  int pair = blockIdx.x;
  if (pair < npairs) {
    pair_rows(pair, c0, L, h);
    fetch(c0, L, h);
  }
  for (; pair < npairs; pair += pblocks) {
  const size_t row0 = (size_t)c0;
  const int hcur = h, bcur = pair / H;
  // ---- the prefetched pieces -> LDS: thread -> (token r, 16-byte chunk c); row-major and transposed images, delta
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + i * 64 * BW;
    if (idx >= LP * 8) continue;
    const int r = idx >> 3, c = idx & 7;
    const uint4 q4 = pq[i], k4 = pk[i], v4 = pv[i], d4 = pd[i], o4 = po[i];
    *(uint4*)(Q + r * RS + c * 8) = q4;
    *(uint4*)(K + r * RS + c * 8) = k4;
    *(uint4*)(V + r * RS + c * 8) = v4;
    *(uint4*)(dO + r * RS + c * 8) = d4;
    float fd[8], fo[8];
    unpack_bf16x8(d4, fd);
    unpack_bf16x8(o4, fo);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += fd[e] * fo[e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (c == 0) delta[r] = s;
  }
  const int Lcur = L;
  {                                                    // the next pair's pieces: in flight under this pair's two phases
    const int nxt = pair + pblocks;
    if (nxt < npairs) {
      pair_rows(nxt, c0, L, h);
      fetch(c0, L, h);
    }
  }
  // zero the padding columns [LP, LS) of the token-contiguous images is unnecessary: k-steps never read past LP.
  LDS_BARRIER();

  const int r16 = lane & 15, quad = lane >> 4;
  // Fragment reads are issued as a batch in front of the MFMAs that use them (and pinned there: the scheduler otherwise
  // sinks every ds_read to its MFMA with an lgkmcnt(0) in between, one exposed LDS round trip per MFMA -- what made the
  // 197-token forward kernel latency-bound).
  auto frag = [&](const bf16_t* p, int stride, int ks) { return *(const bf16x8*)(p + r16 * stride + ks * 32 + quad * 8); };
  // ---- phase 1: per 16-query tile: S^T, softmax over keys, dP^T, dS^T -> P^T, dS, dS^T in LDS
  for (int qt = wave; qt < NT16; qt += BW) {
    const int query = qt * 16 + r16;
    f32x4 st[NT16];
    float mx = -INFINITY;
    {   // S^T tiles of all key tiles: every K fragment (and the two Q fragments) requested before the first MFMA
      bf16x8 kf[NT16][2];
      const bf16x8 q0 = frag(Q + qt * 16 * RS, RS, 0), q1 = frag(Q + qt * 16 * RS, RS, 1);
#pragma unroll
      for (int kt = 0; kt < NT16; ++kt) {
        kf[kt][0] = frag(K + kt * 16 * RS, RS, 0);
        kf[kt][1] = frag(K + kt * 16 * RS, RS, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < NT16; ++kt) {
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], q0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], q1, st[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kt = 0; kt < NT16; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + quad * 4 + r;
        const bool ok = key < Lcur && (!CAUSAL || key <= query);
        st[kt][r] = ok ? st[kt][r] : -INFINITY;
        mx = fmaxf(mx, st[kt][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;                               // padded query row: every key masked
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT16; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[kt][r] = __expf(st[kt][r] - mx);
        sum += st[kt][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    const float dl = delta[query];
    f32x4 dpt[NT16];
    {   // dP^T tiles the same way
      bf16x8 vf[NT16][2];
      const bf16x8 d0 = frag(dO + qt * 16 * RS, RS, 0), d1 = frag(dO + qt * 16 * RS, RS, 1);
#pragma unroll
      for (int kt = 0; kt < NT16; ++kt) {
        vf[kt][0] = frag(V + kt * 16 * RS, RS, 0);
        vf[kt][1] = frag(V + kt * 16 * RS, RS, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < NT16; ++kt) {
        dpt[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt][0], d0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dpt[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt][1], d1, dpt[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kt = 0; kt < NT16; ++kt) {
      const f32x4 dp = dpt[kt];
      float ds[4], pp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pp[r] = st[kt][r] * inv;
        ds[r] = pp[r] * (dp[r] - dl);
      }
      uint2 u, w;
      u.x = pack_bf16x2(ds[0], ds[1]);
      u.y = pack_bf16x2(ds[2], ds[3]);
      w.x = pack_bf16x2(pp[0], pp[1]);
      w.y = pack_bf16x2(pp[2], pp[3]);
      *(uint2*)(dS + query * LS + kt * 16 + quad * 4) = u;
      *(uint2*)(P + query * LS + kt * 16 + quad * 4) = w;
    }
  }
  LDS_BARRIER();

  // ---- phase 2: dV^T = dO^T-rows x P^T-rows (over queries), dK^T = Q^T x dS^T (over queries), dQ^T = K^T x dS (over keys)
  bf16_t* gb = dqkv + row0 * ldq + hcur * 64;
  for (int t = wave; t < 3 * 4 * NT16; t += BW) {
    const int which = t / (4 * NT16), rem = t - which * 4 * NT16;
    const int dt = rem / NT16, tt = rem - dt * NT16;            // head-dim tile, token tile
    const bf16_t* pa = which == 0 ? dO : (which == 1 ? Q : K);          // row-major [token][64]: read transposed
    const bf16_t* pb = which == 0 ? P : dS;            // [query][key]: dV^T and dK^T contract over queries (read transposed), dQ^T over keys
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
      constexpr int KS = LP / 32;
      bf16x8 a[KS], bb[KS];
      // A fragment (head-dim row dt*16 + r16, tokens ks*32 + quad*8 .. + 7): a 16-lane group reads a 4-token x 16-channel block
      // of the row-major image, lane i supplying (token i / 4, channels 4 (i % 4) ..) and receiving channel i's four tokens
      const char* abase = (const char*)(pa + (quad * 8 + (r16 >> 2)) * RS + dt * 16 + 4 * (r16 & 3));
      const char* bbase = (const char*)(pb + (quad * 8 + (r16 >> 2)) * LS + tt * 16 + 4 * (r16 & 3));   // B rows = keys tt*16 + r16, k = queries
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[ks] = ld_tr8(abase + ks * 32 * RS * 2, abase + (ks * 32 + 4) * RS * 2);
        bb[ks] = which == 2 ? frag(pb + tt * 16 * LS, LS, ks) : ld_tr8(bbase + ks * 32 * LS * 2, bbase + (ks * 32 + 4) * LS * 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bb[ks], acc, 0, 0, 0);
    }
    const int tok = tt * 16 + r16;
    if (tok < Lcur) {
      uint2 u;
      u.x = pack_bf16x2(acc[0], acc[1]);
      u.y = pack_bf16x2(acc[2], acc[3]);
      const int col = (which == 0 ? 2 * H * 64 : (which == 1 ? H * 64 : 0)) + dt * 16 + quad * 4;
      *(uint2*)(gb + (size_t)tok * ldq + col) = u;
    }
    if (csum_part) {
      // the tile's sums over its 16 tokens (lane bits 0..3), fixed order: lane exchanges on the DPP path
      f32x4 sv = tok < Lcur ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sv[r];
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
        sv[r] = v;
      }
      if (r16 == 0) *(f32x4*)(tsum + tbuf * TSUM + ((which * 4 + dt) * NT16 + tt) * 16 + quad * 4) = sv;
    }
  }
  LDS_BARRIER();                                       // every wave is done with this pair's images before the next pair's are written
  if (csum_part) {
    // wave 0 folds the pair's token tiles (in tile order) while the others start on the next pair, whose sums go to the other
    // buffer; this buffer is written again two barriers further on, behind this read
    if (wave == 0) {
#pragma unroll
      for (int which = 0; which < 3; ++which) {
        float v = 0.f;
#pragma unroll
        for (int tt = 0; tt < NT16; ++tt) v += tsum[tbuf * TSUM + ((which * 4 + (lane >> 4)) * NT16 + tt) * 16 + (lane & 15)];
        csum_part[(size_t)bcur * 3 * H * 64 + (which == 0 ? 2 * H * 64 : (which == 1 ? H * 64 : 0)) + hcur * 64 + lane] = v;
      }
    }
    tbuf ^= 1;
  }
  }
}

template <int NT16, bool CAUSAL>
int launch_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, int nsamples, int L, int H, int ldq, int ldo,
               hipStream_t st, const int* cu = nullptr, int pad_rows = 0, float* csum_part = nullptr) {
  constexpr int LP = NT16 * 16, LS = LP + 8;
  // 64 tokens: 54 KB of LDS -> TWO workgroups of 4 waves per CU (the kernel's ~250 VGPRs allow 8 waves per CU either way: as one
  // 8-wave workgroup its four query tiles left half the waves idle in phase 1, and nothing ran under its barriers and loads);
  // 96 tokens: 94 KB -> one workgroup of 8 waves
  constexpr int BW = NT16 <= 4 ? 4 : 8;
  const size_t lds = (size_t)(4 * LP * RS + 2 * LP * LS) * 2 + LP * 4 + 2 * (3 * 4 * NT16 * 16) * 4;
  bool attr_ok = true;
  if (lds > 65536) MSCLIP_LDS_ATTR((&attn_bwd_kernel<NT16, CAUSAL, BW>), lds, attr_ok);
  (void)attr_ok;
  const int extra = cu ? (pad_rows + BW - 1) / BW : 0;
  const int ncu = msclip_device_cus();
  const int pairs = nsamples * H;
  const int per_cu = NT16 <= 4 ? 2 : 1;
  const int pblocks = pairs < ncu * per_cu ? pairs : ncu * per_cu;
  hipLaunchKernelGGL((attn_bwd_kernel<NT16, CAUSAL, BW>), dim3(pblocks + extra), dim3(64 * BW), lds, st, (const bf16_t*)qkv,
                     (const bf16_t*)o, (const bf16_t*)dout, (bf16_t*)dqkv, L, H, ldq, ldo, cu, nsamples, pad_rows, pblocks, csum_part);
  return msclip_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Longer sequences (96 < L <= 208: the 197-token grid of ViT-B/16).  P^T, dS and dS^T of a whole head no longer fit the
// LDS next to the operands (3 x 208 x 216 bf16 = 270 KB), so the QUERY axis is processed in blocks of 32:
//   resident for the head:  K, V row-major, K^T                                     (87 KB at L = 197)
//   per query block:        Q, dO rows and their transposes, delta, P^T / dS^T [key][32], dS [32][key]   (66 KB)
// dQ of a block is complete after the block (it contracts over keys); dV^T and dK^T contract over queries and are
// accumulated in registers across the blocks (2 * 4 * NT16 tiles of 16 x 16 over the QB_WAVES waves: NT16 accumulators
// per wave at eight) and stored once at the end.  Phase 1 of a block has only two 16-query tiles, so a wave takes (query
// tile, one of QB_WAVES / 2 parts of the key tiles) and the parts exchange the row maximum and the row sum through LDS.
// The key axis is padded to a multiple of 32 (one MFMA k-step) with zero columns in K^T and dS.
// ------------------------------------------------------------------------------------------------------------
// Waves per workgroup of the query-blocked kernel: its 153 KB of LDS also mean one workgroup per CU.
constexpr int QB_WAVES = 8;

template <int NT16, bool CAUSAL>
__global__ __launch_bounds__(64 * QB_WAVES) void attn_bwd_qb_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv, int L,
                                                          int H, int ldq, int ldo) {
  constexpr int LP = NT16 * 16, LPK = (LP + 31) / 32 * 32, LS = LPK + 8;   // padded keys, k-step padded keys, [..][key] stride
  constexpr int QB = 32, QS = QB + 8;                                       // queries per block, [..][query] stride
  constexpr int NW = QB_WAVES, KH = NW / 2;                                 // waves; key parts of phase 1 (two query tiles x KH)
  constexpr int NA = (2 * 4 * NT16 + NW - 1) / NW;                          // dV^T / dK^T accumulator tiles per wave
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  bf16_t* K = sm;                                      // [LP][RS]
  bf16_t* V = K + LP * RS;
  bf16_t* KT = V + LP * RS;                            // [64][LS]
  bf16_t* Qb = KT + 64 * LS;                           // [QB][RS]
  bf16_t* dOb = Qb + QB * RS;
  bf16_t* QT = dOb + QB * RS;                          // [64][QS]
  bf16_t* dOT = QT + 64 * QS;
  bf16_t* PT = dOT + 64 * QS;                          // [LP][QS]   (key, query in block)
  bf16_t* dST = PT + LP * QS;
  bf16_t* dS = dST + LP * QS;                          // [QB][LS]
  float* delta = (float*)(dS + QB * LS);               // [QB]
  float* red = delta + QB;                             // [2 stats][KH key parts][QB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const size_t row0 = (size_t)b * L;
  const bf16_t* qb = qkv + row0 * ldq + h * 64;
  const bf16_t* ob = o + row0 * ldo + h * 64;
  const bf16_t* db = dout + row0 * ldo + h * 64;
  bf16_t* gb = dqkv + row0 * ldq + h * 64;

  // ---- resident operands: K, V rows, K^T (zero beyond L, and in the k-step padding columns)
  for (int idx = tid; idx < LP * 8; idx += 64 * NW) {
    const int r = idx >> 3, c = idx & 7;
    uint4 k4 = make_uint4(0, 0, 0, 0), v4 = k4;
    if (r < L) {
      k4 = *(const uint4*)(qb + (size_t)r * ldq + H * 64 + c * 8);
      v4 = *(const uint4*)(qb + (size_t)r * ldq + 2 * H * 64 + c * 8);
    }
    *(uint4*)(K + r * RS + c * 8) = k4;
    *(uint4*)(V + r * RS + c * 8) = v4;
    const unsigned kw[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      KT[(c * 8 + 2 * e) * LS + r] = (bf16_t)(kw[e] & 0xffff);
      KT[(c * 8 + 2 * e + 1) * LS + r] = (bf16_t)(kw[e] >> 16);
    }
  }
  if constexpr (LPK > LP) {
    constexpr int PADC = LPK - LP;
    for (int idx = tid; idx < 64 * PADC; idx += 64 * NW) KT[(idx / PADC) * LS + LP + idx % PADC] = 0;
    for (int idx = tid; idx < QB * PADC; idx += 64 * NW) dS[(idx / PADC) * LS + LP + idx % PADC] = 0;
  }

  const int r16 = lane & 15, quad = lane >> 4;
  // fragment reads batched in front of their MFMAs (see attn_bwd_kernel)
  auto mma = [&](f32x4 acc, const bf16_t* pa, int sa, const bf16_t* pb, int sb, auto ksc) {
    constexpr int KS = decltype(ksc)::value;
    bf16x8 a[KS], bb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a[ks] = *(const bf16x8*)(pa + r16 * sa + ks * 32 + quad * 8);
      bb[ks] = *(const bf16x8*)(pb + r16 * sb + ks * 32 + quad * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bb[ks], acc, 0, 0, 0);
    return acc;
  };
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using KL = std::integral_constant<int, LPK / 32>;

  f32x4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NK0 = (NT16 + KH - 1) / KH;            // key tiles per part (the last part may be short)
  const int qt = wave & 1, kh = wave >> 1;             // phase 1: this wave's query tile of the block and key part
  const int kt0 = kh * NK0, kt1 = kt0 + NK0 < NT16 ? kt0 + NK0 : NT16;

  for (int q0 = 0; q0 < L; q0 += QB) {
    __syncthreads();                                   // previous block's phase 2 is done with the block buffers
    if (tid < QB * 8) {
      const int r = tid >> 3, c = tid & 7, q = q0 + r;
      uint4 q4 = make_uint4(0, 0, 0, 0), d4 = q4, o4 = q4;
      if (q < L) {
        q4 = *(const uint4*)(qb + (size_t)q * ldq + c * 8);
        d4 = *(const uint4*)(db + (size_t)q * ldo + c * 8);
        o4 = *(const uint4*)(ob + (size_t)q * ldo + c * 8);
      }
      *(uint4*)(Qb + r * RS + c * 8) = q4;
      *(uint4*)(dOb + r * RS + c * 8) = d4;
      const unsigned qw[4] = {q4.x, q4.y, q4.z, q4.w}, dw[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        QT[(c * 8 + 2 * e) * QS + r] = (bf16_t)(qw[e] & 0xffff);
        QT[(c * 8 + 2 * e + 1) * QS + r] = (bf16_t)(qw[e] >> 16);
        dOT[(c * 8 + 2 * e) * QS + r] = (bf16_t)(dw[e] & 0xffff);
        dOT[(c * 8 + 2 * e + 1) * QS + r] = (bf16_t)(dw[e] >> 16);
      }
      float fd[8], fo[8];
      unpack_bf16x8(d4, fd);
      unpack_bf16x8(o4, fo);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += fd[e] * fo[e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (c == 0) delta[r] = s;
    }
    __syncthreads();

    // ---- phase 1: S^T of (query tile qt, key half kh), softmax statistics shared with the other half, dP^T, dS
    {
      const int qi = qt * 16 + r16, query = q0 + qi;
      f32x4 st[NK0];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NK0; ++i) {
        const int kt = kt0 + i;
        if (kt < kt1) {
          st[i] = mma(f32x4{0.f, 0.f, 0.f, 0.f}, K + kt * 16 * RS, RS, Qb + qt * 16 * RS, RS, K2{});
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + quad * 4 + r;
            const bool ok = key < L && (!CAUSAL || key <= query);
            st[i][r] = ok ? st[i][r] : -INFINITY;
            mx = fmaxf(mx, st[i][r]);
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (quad == 0) red[kh * QB + qi] = mx;
      __syncthreads();
      mx = red[qi];
#pragma unroll
      for (int p = 1; p < KH; ++p) mx = fmaxf(mx, red[p * QB + qi]);
      if (mx == -INFINITY) mx = 0.f;                   // padded query row: every key masked
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NK0; ++i)
        if (kt0 + i < kt1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[i][r] = __expf(st[i][r] - mx);
            sum += st[i][r];
          }
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (quad == 0) red[(KH + kh) * QB + qi] = sum;
      __syncthreads();
      sum = red[KH * QB + qi];
#pragma unroll
      for (int p = 1; p < KH; ++p) sum += red[(KH + p) * QB + qi];
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      const float dl = delta[qi];
#pragma unroll
      for (int i = 0; i < NK0; ++i) {
        const int kt = kt0 + i;
        if (kt < kt1) {
          const f32x4 dp = mma(f32x4{0.f, 0.f, 0.f, 0.f}, V + kt * 16 * RS, RS, dOb + qt * 16 * RS, RS, K2{});
          float ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = st[i][r] * inv;
            ds[r] = p * (dp[r] - dl);
            const int key = kt * 16 + quad * 4 + r;
            PT[key * QS + qi] = f32_to_bf16(p);
            dST[key * QS + qi] = f32_to_bf16(ds[r]);
          }
          uint2 u;
          u.x = pack_bf16x2(ds[0], ds[1]);
          u.y = pack_bf16x2(ds[2], ds[3]);
          *(uint2*)(dS + qi * LS + kt * 16 + quad * 4) = u;
        }
      }
    }
    __syncthreads();

    // ---- phase 2: dV^T += dO^T . P^T, dK^T += Q^T . dS^T over the block's 32 queries (one k-step); dQ^T of the block
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int t = wave + NW * i;
      if (t < 2 * 4 * NT16) {
        const int which = t / (4 * NT16), rem = t - which * 4 * NT16;
        const int dt = rem / NT16, kt = rem - dt * NT16;
        acc[i] = mma(acc[i], (which ? QT : dOT) + dt * 16 * QS, QS, (which ? dST : PT) + kt * 16 * QS, QS, K1{});
      }
    }
    for (int t = wave; t < 4 * (QB / 16); t += NW) {
      const int dt = t >> 1, tq = t & 1;
      const f32x4 a = mma(f32x4{0.f, 0.f, 0.f, 0.f}, KT + dt * 16 * LS, LS, dS + tq * 16 * LS, LS, KL{});
      const int tok = q0 + tq * 16 + r16;
      if (tok < L) {
        uint2 u;
        u.x = pack_bf16x2(a[0], a[1]);
        u.y = pack_bf16x2(a[2], a[3]);
        *(uint2*)(gb + (size_t)tok * ldq + dt * 16 + quad * 4) = u;
      }
    }
  }
  // ---- dV^T, dK^T tiles: 4 consecutive head-dim elements per lane and key
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int t = wave + NW * i;
    const int which = t / (4 * NT16), rem = t - which * 4 * NT16;
    const int dt = rem / NT16, kt = rem - dt * NT16;
    const int tok = kt * 16 + r16;
    if (t < 2 * 4 * NT16 && tok < L) {
      uint2 u;
      u.x = pack_bf16x2(acc[i][0], acc[i][1]);
      u.y = pack_bf16x2(acc[i][2], acc[i][3]);
      *(uint2*)(gb + (size_t)tok * ldq + (which ? H * 64 : 2 * H * 64) + dt * 16 + quad * 4) = u;
    }
  }
}

template <int NT16, bool CAUSAL>
int launch_bwd_qb(const void* qkv, const void* o, const void* dout, void* dqkv, int nsamples, int L, int H, int ldq, int ldo,
                  hipStream_t st) {
  constexpr int LP = NT16 * 16, LPK = (LP + 31) / 32 * 32, LS = LPK + 8, QB = 32, QS = QB + 8;
  const size_t lds = (size_t)(2 * LP * RS + 64 * LS + 2 * QB * RS + 2 * 64 * QS + 2 * LP * QS + QB * LS) * 2 + (1 + QB_WAVES) * QB * 4;
  bool attr_ok = true;
  MSCLIP_LDS_ATTR((&attn_bwd_qb_kernel<NT16, CAUSAL>), lds, attr_ok);
  (void)attr_ok;
  hipLaunchKernelGGL((attn_bwd_qb_kernel<NT16, CAUSAL>), dim3(nsamples * H), dim3(64 * QB_WAVES), lds, st, (const bf16_t*)qkv,
                     (const bf16_t*)o, (const bf16_t*)dout, (bf16_t*)dqkv, L, H, ldq, ldo);
  return msclip_launch_status();
}

}  // namespace

extern "C" int msclip_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, int nsamples, int L,
                                    int heads, int ldq, int ldo, int causal, float* colsum_part, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention_bwd, stream, qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, causal, colsum_part);
  if (!qkv || !o || !dout || !dqkv || nsamples <= 0 || L <= 0 || L > 208 || heads <= 0 || (ldq % 8) || (ldo % 8))
    return MSCLIP_EINVAL;
  if (colsum_part && L > 96) return MSCLIP_EINVAL;      // the query-blocked form does not carry the per-sample column sums
  hipStream_t st = (hipStream_t)stream;
  if (L > 160) return causal ? launch_bwd_qb<13, true>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st)
                             : launch_bwd_qb<13, false>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st);
  if (L > 96) return causal ? launch_bwd_qb<10, true>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st)
                            : launch_bwd_qb<10, false>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st);
  if (L <= 64) return causal ? launch_bwd<4, true>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st, nullptr, 0, colsum_part)
                             : launch_bwd<4, false>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st, nullptr, 0, colsum_part);
  return causal ? launch_bwd<6, true>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st, nullptr, 0, colsum_part)
                : launch_bwd<6, false>(qkv, o, dout, dqkv, nsamples, L, heads, ldq, ldo, st, nullptr, 0, colsum_part);
}

extern "C" int msclip_attention_bwd_varlen(const void* qkv, const void* o, const void* dout, void* dqkv, const int* cu, int nsamples,
                                           int Lmax, int heads, int ldq, int ldo, int causal, int pad_rows, float* colsum_part,
                                           void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention_bwd_varlen, stream, qkv, o, dout, dqkv, cu, nsamples, Lmax, heads, ldq, ldo, causal, pad_rows, colsum_part);
  if (!qkv || !o || !dout || !dqkv || !cu || nsamples <= 0 || Lmax <= 0 || Lmax > 96 || heads <= 0 || (ldq % 8) || (ldo % 8) ||
      pad_rows < 0 || pad_rows > 255)
    return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (Lmax <= 32) return causal ? launch_bwd<2, true>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part)
                                : launch_bwd<2, false>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part);
  if (Lmax <= 64) return causal ? launch_bwd<4, true>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part)
                                : launch_bwd<4, false>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part);
  return causal ? launch_bwd<6, true>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part)
                : launch_bwd<6, false>(qkv, o, dout, dqkv, nsamples, Lmax, heads, ldq, ldo, st, cu, pad_rows, colsum_part);
}
