// Fused scaled-dot-product attention for short sequences (L <= 224): one wave
// per (sample, head); S, P and O never leave registers, V^T is the only LDS
// resident.  Replaces bmm + mask-add + softmax + bmm of Attention_CUST
// (reference lib/models/clip_openai_pe_res_v1.py:707-738).
//
// Layout: qkv rows are tokens (sample-major, L per sample), columns
// [q(H*64) | k(H*64) | v(H*64)], q already scaled by head_dim^-0.5 (folded into
// the packed in_proj weight).  Output o[token][h*64 + d].
//
// MFMA plan (v_mfma_f32_32x32x16_bf16, operands swapped so that the softmax
// axis is register-local):
//   S^T[key][query]  = K . Q^T      A = K rows  (global -> VGPR, 16 B/lane)
//                                   B = Q rows  (global -> VGPR)
//   lane l holds query (l & 31) and 16 of every 32 keys; row max/sum = in-lane
//   reduce + one exchange with lane l^32.
//   O^T[d][query]    = V^T . P^T    A = V^T rows (LDS, ds_read_b128)
//                                   B = P^T     = the S^T accumulators packed to
//                                       bf16 in place (no cross-lane movement):
//   k-step s of PV contracts 16 keys; the order of keys inside a k-step only has
//   to agree between A and B, so V^T is written to LDS in the order the S^T
//   accumulator layout yields: key 16s + w  ->  slot 16s + 8*((w>>2)&1) + (w&3) + 4*(w>>3).
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

template <int NT, bool CAUSAL, int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                   int nsamples, int Lfix, int H, int ldq, int ldo,
                                                   const int* __restrict__ cu, int pad_rows, const int* __restrict__ dims) {
  constexpr int KP = NT * 32;      // padded key count
  constexpr int KPS = KP + 8;      // LDS row stride (elements): 16-B aligned, odd multiple of 16 B
  extern __shared__ __attribute__((aligned(16))) bf16_t vt_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = blockIdx.x * WPB + wave;
  if (pair >= nsamples * H) {
    // packed captions (msclip_attention_varlen): the workgroups behind the last (sample, head) pair zero the output rows of
    // the tile padding [cu[nsamples], cu[nsamples] + pad_rows) -- one wave per row
    const int r = pair - nsamples * H;
    if (dims) pad_rows = min(pad_rows, dims[4]);     // device-side count of the tile-padding rows (msclip_text_lengths)
    if (cu && r < pad_rows) {
      bf16_t* orow = out + (size_t)(cu[nsamples] + r) * ldo;
      for (int c = lane * 8; c < H * 64; c += 512) *(uint4*)(orow + c) = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const int b = pair / H, h = pair - b * H;
  bf16_t* vt = vt_all + wave * (64 * KPS);

  // fixed-length samples: rows b*L .. ; packed captions: rows cu[b] .. cu[b + 1] (wave-uniform either way)
  const int c0 = cu ? __builtin_amdgcn_readfirstlane(cu[b]) : b * Lfix;
  const int L = cu ? min(__builtin_amdgcn_readfirstlane(cu[b + 1]) - c0, KP) : Lfix;
  const size_t row0 = (size_t)c0;
  const bf16_t* qbase = qkv + row0 * ldq + h * 64;
  const bf16_t* kbase = qbase + H * 64;
  const bf16_t* vbase = qbase + 2 * H * 64;

  const int fr = lane & 31, fhi = lane >> 5;

  if constexpr (NT <= 3) {
    // ---- Short sequences: EVERY global load of this (sample, head) is issued up front (V rows, then all Q and K
    // fragments), so the wave pays one memory round trip instead of one per tile; V is transposed into LDS while
    // the Q/K fragments are still in flight, and the rest runs from registers / LDS.
    constexpr int NVI = KP / 8;
    uint4 vreg[NVI];
    {
      const int c = lane & 7;
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int key = i * 8 + (lane >> 3);
        vreg[i] = key < L ? *(const uint4*)(vbase + (size_t)key * ldq + c * 8) : make_uint4(0, 0, 0, 0);
      }
    }
    bf16x8 qf[NT][4], kf[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = min(t * 32 + fr, L - 1);       // clamped: padded queries are never stored, padded keys masked
      if (t * 32 < L) {                              // (wave-uniform: a short packed caption skips its empty tiles' loads)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          qf[t][kk] = *(const bf16x8*)(qbase + (size_t)row * ldq + (kk * 2 + fhi) * 8);
          kf[t][kk] = *(const bf16x8*)(kbase + (size_t)row * ldq + (kk * 2 + fhi) * 8);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[t][kk] = kf[t][kk] = bf16x8{};
      }
    }
    {
      const int c = lane & 7;
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int key = i * 8 + (lane >> 3);
        const int w = key & 15;
        const int slot = (key & ~15) + (((w >> 2) & 1) << 3) + (w & 3) + ((w >> 3) << 2);
        bf16_t* dst = vt + (c * 8) * KPS + slot;
        const uint4 u = vreg[i];
        dst[0 * KPS] = (bf16_t)(u.x & 0xffff); dst[1 * KPS] = (bf16_t)(u.x >> 16);
        dst[2 * KPS] = (bf16_t)(u.y & 0xffff); dst[3 * KPS] = (bf16_t)(u.y >> 16);
        dst[4 * KPS] = (bf16_t)(u.z & 0xffff); dst[5 * KPS] = (bf16_t)(u.z >> 16);
        dst[6 * KPS] = (bf16_t)(u.w & 0xffff); dst[7 * KPS] = (bf16_t)(u.w >> 16);
      }
    }
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      if (qt * 32 < L) {
        const int nkt = CAUSAL ? (qt + 1) : NT;
        f32x16 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
          if (kt < nkt && kt * 32 < L) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][kk], qf[qt][kk], s[kt], 0, 0, 0);
          }
        }
        const int q = qt * 32 + fr;
        float mx = -INFINITY;
        // key tiles above the causal diagonal (kt >= nkt: a compile-time bound here) are skipped in the softmax as well,
        // not only in the MFMAs: a third of the text tower's softmax VALU work at 3 x 3 tiles
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
          if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
              const bool ok = key < L && (!CAUSAL || key <= q);
              s[kt][r] = ok ? s[kt][r] : -INFINITY;
              mx = fmaxf(mx, s[kt][r]);
            }
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
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
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          if (kt < nkt && kt * 32 < L) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              bf16x8 pf;
#pragma unroll
              for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[kt][half * 8 + e];
              const int st = kt * 2 + half;
#pragma unroll
              for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vf = *(const bf16x8*)(vt + (dt * 32 + fr) * KPS + st * 16 + fhi * 8);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
              }
            }
          }
        }
        {
          // lane (q, fhi) holds d = dt*32 + 8g + 4*fhi .. +4; v_permlane32_swap trades group 2s+1 of the low half-wave
          // against group 2s of the high one, so every lane ends up with 8 consecutive d (16-byte stores)
          bf16_t* orow = out + (row0 + min(q, L - 1)) * ldo + h * 64;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              pk[g][0] = pack_bf16x2(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
              pk[g][1] = pack_bf16x2(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const auto x = __builtin_amdgcn_permlane32_swap(pk[2 * s2][0], pk[2 * s2 + 1][0], false, false);
              const auto y = __builtin_amdgcn_permlane32_swap(pk[2 * s2][1], pk[2 * s2 + 1][1], false, false);
              if (q < L) *(uint4*)(orow + dt * 32 + s2 * 16 + fhi * 8) = make_uint4(x[0], y[0], x[1], y[1]);
            }
          }
        }
      }
    }
    return;
  }

  // ---- V -> LDS, transposed and slot-permuted; padded keys are zero
  {
    const int c = lane & 7;  // d chunk (8 values)
#pragma unroll 2
    for (int k0 = 0; k0 < KP; k0 += 8) {
      const int key = k0 + (lane >> 3);
      uint4 u = make_uint4(0, 0, 0, 0);
      if (key < L) u = *(const uint4*)(vbase + (size_t)key * ldq + c * 8);
      const int w = key & 15;
      const int slot = (key & ~15) + (((w >> 2) & 1) << 3) + (w & 3) + ((w >> 3) << 2);
      bf16_t* dst = vt + (c * 8) * KPS + slot;
      dst[0 * KPS] = (bf16_t)(u.x & 0xffff); dst[1 * KPS] = (bf16_t)(u.x >> 16);
      dst[2 * KPS] = (bf16_t)(u.y & 0xffff); dst[3 * KPS] = (bf16_t)(u.y >> 16);
      dst[4 * KPS] = (bf16_t)(u.z & 0xffff); dst[5 * KPS] = (bf16_t)(u.z >> 16);
      dst[6 * KPS] = (bf16_t)(u.w & 0xffff); dst[7 * KPS] = (bf16_t)(u.w >> 16);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes done (wave-private region)
  __builtin_amdgcn_wave_barrier();

  for (int qt = 0; qt < NT; ++qt) {
    if (qt * 32 >= L) break;
    // Q fragments for this query tile (rows clamped: padded queries are never stored)
    const int qrow = min(qt * 32 + fr, L - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qbase + (size_t)qrow * ldq + (kk * 2 + fhi) * 8);

    const int nkt = CAUSAL ? (qt + 1) : NT;  // key tiles that can be unmasked
    f32x16 s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      if (kt < nkt && kt * 32 < L) {
        const int krow = min(kt * 32 + fr, L - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 kf = *(const bf16x8*)(kbase + (size_t)krow * ldq + (kk * 2 + fhi) * 8);
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kt], 0, 0, 0);
        }
      }
    }

    // ---- masked softmax over keys (register-local + one half-wave exchange)
    const int q = qt * 32 + fr;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
        const bool ok = key < L && (!CAUSAL || key <= q) && kt < nkt;
        s[kt][r] = ok ? s[kt][r] : -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __expf(s[kt][r] - mx);  // exp(-inf) = 0 for masked keys; key 0 is never masked
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;

    // ---- O^T = V^T . P^T
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt < nkt && kt * 32 < L) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          bf16x8 pf;
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[kt][half * 8 + e];
          const int st = kt * 2 + half;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8 vf = *(const bf16x8*)(vt + (dt * 32 + fr) * KPS + st * 16 + fhi * 8);
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
          }
        }
      }
    }

    // ---- store: lane owns query q, d = dt*32 + 8g + 4*fhi + 0..3
    if (q < L) {
      bf16_t* orow = out + (row0 + q) * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 v;
          v.x = pack_bf16x2(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
          v.y = pack_bf16x2(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
          *(uint2*)(orow + dt * 32 + g * 8 + fhi * 4) = v;
        }
    }
  }
}

// Longer sequences (L = 197 for the 16-pixel patch grid): one WORKGROUP of 4 waves per (sample, head).  K (row-major,
// rows padded to 144 B: conflict-free fragment reads) and V^T (slot-permuted as above) are loaded into LDS once,
// cooperatively and in full rows; wave w then takes the query tiles w, w+4: its Q fragments come straight from HBM
// (requested one tile ahead), the K fragments of S^T = K.Q^T and the V^T fragments of O^T = V^T.P^T from LDS.
// Two workgroups fit a CU (57 KiB each), i.e. two waves per SIMD overlap one wave's softmax VALU work with the
// other's MFMAs.  (The one-wave-per-pair kernel above re-read K from HBM for every query tile and ran at one wave
// per SIMD: 268 us per layer at B = 256 against 61 GFLOP of work.)
template <int NT, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_wg_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      int nsamples, int L, int H, int ldq, int ldo) {
  constexpr int KP = NT * 32;
  constexpr int KPS = KP + 8;        // V^T row stride (elements)
  constexpr int KROW = 72;           // K row stride (elements): 144 B
  __shared__ __attribute__((aligned(16))) bf16_t kl[KP * KROW];
  __shared__ __attribute__((aligned(16))) bf16_t vt[64 * KPS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = blockIdx.x;
  const int b = pair / H, h = pair - b * H;
  const size_t row0 = (size_t)b * L;
  const bf16_t* qbase = qkv + row0 * ldq + h * 64;
  const bf16_t* kbase = qbase + H * 64;
  const bf16_t* vbase = qbase + 2 * H * 64;
  const int fr = lane & 31, fhi = lane >> 5;

  // ---- this wave's first Q tile is requested before the cooperative K / V pass.  The Q loads are inline asm with a
  // COUNTED wait: as compiler-tracked loads the next tile's prefetch was waited for with vmcnt(0) at the top of the
  // loop, i.e. behind the write acknowledgements of the previous tile's output stores (the VM counter retires in order);
  // the prefetch is issued before those four stores, so vmcnt(4) is its exact wait.
  auto load_q = [&](int qt, bf16x8 (&qf)[4]) {
    const int qrow = min(qt * 32 + fr, L - 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16_t* p = qbase + (size_t)qrow * ldq + (kk * 2 + fhi) * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(qf[kk]) : "v"(p));
    }
  };
  bf16x8 qcur[4], qnext[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qcur[kk] = qnext[kk] = bf16x8{};
  if (wave * 32 < L) load_q(wave, qcur);

  // ---- K rows and V^T into LDS (8 threads per row, 16 B each; padded keys are zero).  Every global load of the pass is
  // issued before the first LDS write: one memory round trip per workgroup instead of one per 32 keys (the loop form
  // -- load, write, load, write -- made this pass 7 dependent round trips at L = 197 and the kernel latency-bound:
  // 121 us per layer at B = 256 against ~40 us of VALU work).
  {
    const int c = tid & 7;
    uint4 ku[NT], vu[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int key = (tid >> 3) + 32 * i;
      const int kc = min(key, L - 1);                  // unconditional (clamped) loads: no exec-mask branch between them
      ku[i] = *(const uint4*)(kbase + (size_t)kc * ldq + c * 8);
      vu[i] = *(const uint4*)(vbase + (size_t)kc * ldq + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if ((tid >> 3) + 32 * i >= L) ku[i] = vu[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int key = (tid >> 3) + 32 * i;
      *(uint4*)(kl + key * KROW + c * 8) = ku[i];
      const int w = key & 15;
      const int slot = (key & ~15) + (((w >> 2) & 1) << 3) + (w & 3) + ((w >> 3) << 2);
      bf16_t* dst = vt + (c * 8) * KPS + slot;
      dst[0 * KPS] = (bf16_t)(vu[i].x & 0xffff); dst[1 * KPS] = (bf16_t)(vu[i].x >> 16);
      dst[2 * KPS] = (bf16_t)(vu[i].y & 0xffff); dst[3 * KPS] = (bf16_t)(vu[i].y >> 16);
      dst[4 * KPS] = (bf16_t)(vu[i].z & 0xffff); dst[5 * KPS] = (bf16_t)(vu[i].z >> 16);
      dst[6 * KPS] = (bf16_t)(vu[i].w & 0xffff); dst[7 * KPS] = (bf16_t)(vu[i].w >> 16);
    }
  }
  // the first Q tile was requested before the K / V loads above, whose (compiler-tracked) wait also covers it
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(qcur[0]), "+v"(qcur[1]), "+v"(qcur[2]), "+v"(qcur[3])::"memory");
  __syncthreads();

  for (int qt = wave; qt < NT && qt * 32 < L; qt += 4) {
    const bool more = (qt + 4) < NT && (qt + 4) * 32 < L;
    if (more) load_q(qt + 4, qnext);
    const int nkt = CAUSAL ? (qt + 1) : NT;
    f32x16 s[NT];
    if constexpr (!CAUSAL) {
      // Straight-line and software-pipelined: the K fragments of key tile kt + 1 are requested before the four MFMAs of
      // tile kt (padded keys are zero in LDS and masked below, so no tile is skipped).  As "read, wait, MFMA" per fragment
      // -- what the guarded per-tile form compiled to -- every one of the 56 MFMAs of a query tile exposed a full LDS
      // round trip: the kernel was LDS-latency-bound (removing the softmax's exp or either MFMA group changed nothing).
      bf16x8 kf[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[0][kk] = *(const bf16x8*)(kl + fr * KROW + (kk * 2 + fhi) * 8);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt + 1 < NT) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            kf[(kt + 1) & 1][kk] = *(const bf16x8*)(kl + ((kt + 1) * 32 + fr) * KROW + (kk * 2 + fhi) * 8);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks every read back in front of its MFMA)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt & 1][kk], qcur[kk], s[kt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        if (kt < nkt && kt * 32 < L) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 kf = *(const bf16x8*)(kl + (kt * 32 + fr) * KROW + (kk * 2 + fhi) * 8);
            s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qcur[kk], s[kt], 0, 0, 0);
          }
        }
      }
    }
    const int q = qt * 32 + fr;
    float mx = -INFINITY;
    // The softmax is this kernel's VALU budget (112 values per lane and query tile).  Key tiles that lie completely
    // inside the sequence (and below the causal diagonal) need no mask: the compare + select pair runs only on the
    // boundary tiles (a wave-uniform choice per tile); p = 2^(s*log2e - max*log2e) is one FMA + v_exp per value.
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const bool inside = kt * 32 + 32 <= L && (!CAUSAL || kt < qt) && kt < nkt;
      if (inside) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
          const bool ok = key < L && (!CAUSAL || key <= q) && kt < nkt;
          s[kt][r] = ok ? s[kt][r] : -INFINITY;
          mx = fmaxf(mx, s[kt][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxl = mx * 1.44269504f;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], 1.44269504f, -mxl));
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    if constexpr (!CAUSAL) {
      // the same for O^T = V^T . P^T: the two V^T fragments of k-step st + 1 are requested before the MFMAs of k-step st
      bf16x8 vf[2][2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) vf[0][dt] = *(const bf16x8*)(vt + (dt * 32 + fr) * KPS + fhi * 8);
#pragma unroll
      for (int st = 0; st < 2 * NT; ++st) {
        if (st + 1 < 2 * NT) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            vf[(st + 1) & 1][dt] = *(const bf16x8*)(vt + (dt * 32 + fr) * KPS + (st + 1) * 16 + fhi * 8);
        }
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[st >> 1][(st & 1) * 8 + e];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st & 1][dt], pf, o[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt && kt * 32 < L) {
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[kt][half * 8 + e];
            const int st = kt * 2 + half;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const bf16x8 vf = *(const bf16x8*)(vt + (dt * 32 + fr) * KPS + st * 16 + fhi * 8);
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
            }
          }
        }
      }
    }
    {
      bf16_t* orow = out + (row0 + min(q, L - 1)) * ldo + h * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        unsigned pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          pk[g][0] = pack_bf16x2(o[dt][g * 4 + 0] * inv, o[dt][g * 4 + 1] * inv);
          pk[g][1] = pack_bf16x2(o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {                           // 8 consecutive d per lane (see attn_kernel)
          const auto x = __builtin_amdgcn_permlane32_swap(pk[2 * s2][0], pk[2 * s2 + 1][0], false, false);
          const auto y = __builtin_amdgcn_permlane32_swap(pk[2 * s2][1], pk[2 * s2 + 1][1], false, false);
          if (q < L) *(uint4*)(orow + dt * 32 + s2 * 16 + fhi * 8) = make_uint4(x[0], y[0], x[1], y[1]);
        }
      }
    }
    if (more) {   // younger than the prefetch: exactly this tile's four output stores (row qt * 32 is a real query: none is skipped)
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(qnext[0]), "+v"(qnext[1]), "+v"(qnext[2]), "+v"(qnext[3])::"memory");
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qcur[kk] = qnext[kk];
    }
  }
}

template <int NT>
int launch_wg(const void* qkv, void* out, int nsamples, int L, int H, int ldq, int ldo, int causal, hipStream_t st) {
  const int pairs = nsamples * H;
  if (causal)
    hipLaunchKernelGGL((attn_wg_kernel<NT, true>), dim3(pairs), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out,
                       nsamples, L, H, ldq, ldo);
  else
    hipLaunchKernelGGL((attn_wg_kernel<NT, false>), dim3(pairs), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out,
                       nsamples, L, H, ldq, ldo);
  return msclip_launch_status();
}

// ---- One query per sample: the last block's attention when only the class row of an image / the EOT row of a caption is
// read afterwards (M.py:2685, 3057-3060).  One wave per (sample, head); lane = (key group g = lane / 8, 16-byte chunk
// c = lane % 8): an iteration reads the 128-byte head slices of 8 keys as full lines (8 lanes per key), a lane forms the partial
// dot product of its 8 dimensions, three shuffles inside the 8-lane group finish the score; the value sum uses the same
// mapping (8 dimensions per lane, the 8 key groups folded at the end).  P is rounded to bf16 like the full kernel's.
template <int MAXI>                                  // iterations of 8 keys: L <= 8 * MAXI
__global__ __launch_bounds__(256) void attn_lastq_kernel(const bf16_t* __restrict__ qc, int ldqc, const bf16_t* __restrict__ kv,
                                                    int ldkv, bf16_t* __restrict__ out, int ldo, int nsamples, int L, int H,
                                                    const int* __restrict__ last_row, int row_base, const int* __restrict__ cu) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= nsamples * H) return;
  const int b = pair / H, h = pair - b * H;
  const int row0 = row_base + (cu ? cu[b] : b * L);           // packed captions: sample b's keys are rows cu[b] .. cu[b + 1]
  int nk = cu ? cu[b + 1] - cu[b] : (last_row ? last_row[b] - row0 + 1 : L);
  nk = min(max(nk, 1), L);
  const int g = lane >> 3, c = lane & 7;
  const bf16_t* kbase = kv + (size_t)row0 * ldkv + H * 64 + h * 64 + c * 8;
  const bf16_t* vbase = kbase + H * 64;
  float q8[8];
  unpack_bf16x8(*(const uint4*)(qc + (size_t)b * ldqc + h * 64 + c * 8), q8);
  float s[MAXI];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int key = i * 8 + g;
    float d = 0.f;
    if (key < nk) {
      float kf[8];
      unpack_bf16x8(*(const uint4*)(kbase + (size_t)key * ldkv), kf);
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(q8[e], kf[e], d);
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    s[i] = key < nk ? d : -INFINITY;
    m = fmaxf(m, s[i]);
  }
  m = fmaxf(m, __shfl_xor(m, 8, 64));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const float p = i * 8 + g < nk ? __expf(s[i] - m) : 0.f;
    sum += p;
    s[i] = bf16_to_f32(f32_to_bf16(p));
  }
  sum += __shfl_xor(sum, 8, 64);
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int key = i * 8 + g;
    if (key < nk) {
      float vf[8];
      unpack_bf16x8(*(const uint4*)(vbase + (size_t)key * ldkv), vf);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(s[i], vf[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 8, 64);
    acc[e] += __shfl_xor(acc[e], 16, 64);
    acc[e] += __shfl_xor(acc[e], 32, 64);
  }
  if (g == 0) {
    const float r = 1.f / sum;
    uint4 o;
    o.x = pack_bf16x2(acc[0] * r, acc[1] * r);
    o.y = pack_bf16x2(acc[2] * r, acc[3] * r);
    o.z = pack_bf16x2(acc[4] * r, acc[5] * r);
    o.w = pack_bf16x2(acc[6] * r, acc[7] * r);
    *(uint4*)(out + (size_t)b * ldo + h * 64 + c * 8) = o;
  }
}

template <int NT>
int launch(const void* qkv, void* out, int nsamples, int L, int H, int ldq, int ldo, int causal, hipStream_t st,
           const int* cu = nullptr, int pad_rows = 0, const int* dims = nullptr) {
  constexpr int WPB = NT > 4 ? 2 : 4;  // keep dynamic LDS under 64 KiB
  const int pairs = nsamples * H + (cu ? pad_rows : 0);         // (one wave per padding row behind the real pairs)
  const int grid = (pairs + WPB - 1) / WPB;
  const size_t lds = WPB * 64 * (NT * 32 + 8) * sizeof(bf16_t);
  if (causal)
    hipLaunchKernelGGL((attn_kernel<NT, true, WPB>), dim3(grid), dim3(WPB * 64), lds, st, (const bf16_t*)qkv,
                       (bf16_t*)out, nsamples, L, H, ldq, ldo, cu, pad_rows, dims);
  else
    hipLaunchKernelGGL((attn_kernel<NT, false, WPB>), dim3(grid), dim3(WPB * 64), lds, st, (const bf16_t*)qkv,
                       (bf16_t*)out, nsamples, L, H, ldq, ldo, cu, pad_rows, dims);
  return msclip_launch_status();
}

}  // namespace

extern "C" int msclip_attention(const void* qkv, void* out, int nsamples, int L, int heads, int ldq, int ldo,
                                int causal, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention, stream, qkv, out, nsamples, L, heads, ldq, ldo, causal);
  if (!qkv || !out || nsamples <= 0 || L <= 0 || heads <= 0 || (ldq % 8) || (ldo % 8)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (L <= 64) return launch<2>(qkv, out, nsamples, L, heads, ldq, ldo, causal, st);
  if (L <= 96) return launch<3>(qkv, out, nsamples, L, heads, ldq, ldo, causal, st);
  if (L <= 224) {
    const char* old = getenv("MSCLIP_ATTN_WAVE");           // the one-wave-per-pair kernel, for cross-checks only
    if (old && old[0] == '1') return launch<7>(qkv, out, nsamples, L, heads, ldq, ldo, causal, st);
    return launch_wg<7>(qkv, out, nsamples, L, heads, ldq, ldo, causal, st);
  }
  if (L <= 288) return launch_wg<9>(qkv, out, nsamples, L, heads, ldq, ldo, causal, st);    // 257 tokens: the 16 x 16 grid of ViT-L/14
  return MSCLIP_EINVAL;
}

static int lastq_launch(const void* q, int ldqc, const void* qkv, int ldq, void* out, int ldo, int nsamples, int L, int heads,
                        const int* last_row, int row_base, const int* cu, void* stream) {
  if (!q || !qkv || !out || nsamples <= 0 || L <= 0 || L > 288 || heads <= 0 || (ldqc % 8) || (ldq % 8) || (ldo % 8) || ldo < heads * 64 || row_base < 0)
    return MSCLIP_EINVAL;
  const int grid = (nsamples * heads + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
#define LASTQ(MAXI)                                                                                                            \
  hipLaunchKernelGGL(attn_lastq_kernel<MAXI>, dim3(grid), dim3(256), 0, st, (const bf16_t*)q, ldqc, (const bf16_t*)qkv, ldq,   \
                     (bf16_t*)out, ldo, nsamples, L, heads, last_row, row_base, cu)
  if (L <= 32) LASTQ(4);
  else if (L <= 64) LASTQ(8);
  else if (L <= 80) LASTQ(10);
  else if (L <= 128) LASTQ(16);
  else if (L <= 256) LASTQ(32);
  else LASTQ(36);
#undef LASTQ
  return msclip_launch_status();
}

extern "C" int msclip_attention_lastq(const void* q, int ldqc, const void* qkv, int ldq, void* out, int ldo, int nsamples, int L,
                                      int heads, const int* last_row, int row_base, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention_lastq, stream, q, ldqc, qkv, ldq, out, ldo, nsamples, L, heads, last_row, row_base);
  return lastq_launch(q, ldqc, qkv, ldq, out, ldo, nsamples, L, heads, last_row, row_base, nullptr, stream);
}

extern "C" int msclip_attention_lastq_varlen(const void* q, int ldqc, const void* qkv, int ldq, void* out, int ldo, int nsamples,
                                             int Lmax, int heads, const int* cu, int row_base, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention_lastq_varlen, stream, q, ldqc, qkv, ldq, out, ldo, nsamples, Lmax, heads, cu, row_base);
  if (!cu) return MSCLIP_EINVAL;
  return lastq_launch(q, ldqc, qkv, ldq, out, ldo, nsamples, Lmax, heads, nullptr, row_base, cu, stream);
}

extern "C" int msclip_attention_varlen(const void* qkv, void* out, const int* cu, int nsamples, int Lmax, int heads, int ldq,
                                       int ldo, int causal, int pad_rows, const int* dims, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_attention_varlen, stream, qkv, out, cu, nsamples, Lmax, heads, ldq, ldo, causal, pad_rows, dims);
  if (!qkv || !out || !cu || nsamples <= 0 || Lmax <= 0 || Lmax > 96 || heads <= 0 || (ldq % 8) || (ldo % 8) || pad_rows < 0 ||
      pad_rows > 255)
    return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (Lmax <= 32) return launch<1>(qkv, out, nsamples, Lmax, heads, ldq, ldo, causal, st, cu, pad_rows, dims);
  if (Lmax <= 64) return launch<2>(qkv, out, nsamples, Lmax, heads, ldq, ldo, causal, st, cu, pad_rows, dims);
  return launch<3>(qkv, out, nsamples, Lmax, heads, ldq, ldo, causal, st, cu, pad_rows, dims);
}
