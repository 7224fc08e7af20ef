// Backward-pass kernels of the transformer blocks and the contrastive head (first slice).
//
// The reference release has no trainer; what these kernels differentiate is the forward the reference defines
// (M.py = lib/models/clip_openai_pe_res_v1.py) plus the symmetric cross-entropy this build adds, and they are pinned
// against autograd of the imported reference (tests/golden/*.grads.npz).  GEMM gradients reuse msclip_gemm:
//   dX = dY @ W        -> msclip_gemm(X = dY, W = W^T)          (W^T packed by the host once per step)
//   dW = dY^T @ X      -> msclip_gemm(X = dY^T, W = X^T)        (both transposed by msclip_transpose_bf16: the
//                          contraction runs over the token rows, which must be the contiguous axis of both operands)
// The shared attention / MLP tensors (M.py:2808-2830) get the SUM of both towers' gradients for free: the towers'
// tokens are rows of one matrix, so one wgrad GEMM contracts over image and text rows together.
// Everything here is HBM-bound elementwise / reduction work: 16-byte accesses, one wave per row where a row reduction
// is needed, fp32 statistics and accumulation, bf16 only for tensors that feed an MFMA GEMM.
#include <mutex>
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

__device__ __forceinline__ float sigmoidf_fast(float z) { return 1.f / (1.f + __expf(-z)); }

// ---- out[c][m] = in[m][c] (bf16), columns m in [M, Mpad) zero-filled: the K-contiguous operand of a wgrad GEMM.
// 64 x 64 tiles through LDS (padded rows: conflict-free both ways).
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int ldi, bf16_t* __restrict__ out,
                                                        int ldo, int M, int C, int Mpad) {
  __shared__ bf16_t tile[64][66];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int m = m0 + r, c = c0 + tx;
    tile[r][tx] = (m < M && c < C) ? in[(size_t)m * ldi + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, m = m0 + tx;
    if (c < C && m < Mpad) out[(size_t)c * ldo + m] = tile[tx][r];
  }
}

// The same with 16-byte global accesses (C % 8 == 0, ldi % 8 == 0, ldo % 8 == 0, Mpad % 64 == 0): 8 lanes read one
// 128-byte row segment and, after the LDS turn, 8 lanes write one 128-byte segment of an output row; the tile's row stride
// of 33 dwords makes the transposed 2-byte reads conflict-free (lanes with consecutive m-chunks sit 8 banks apart).
__global__ __launch_bounds__(256) void transpose_vec_kernel(const bf16_t* __restrict__ in, int ldi, bf16_t* __restrict__ out,
                                                            int ldo, int M, int C) {
  __shared__ bf16_t tile[64][66];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int sub = threadIdx.x >> 3, ch = threadIdx.x & 7;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = sub + 32 * i, m = m0 + r, c = c0 + ch * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < M && c < C) v = *(const uint4*)(in + (size_t)m * ldi + c);
    unsigned* d = (unsigned*)&tile[r][ch * 8];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = sub + 32 * i, c = c0 + r;
    if (c < C) {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = tile[ch * 8 + j][r];
      uint4 v;
      v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
      *(uint4*)(out + (size_t)c * ldo + m0 + ch * 8) = v;
    }
  }
}

// Many matrices in ONE launch (the training step's W^T operands of the dgrad GEMMs: 48 weight matrices per step, issued one
// by one they kept the HOST busy for 0.6 ms at the start of the backward with the main queue idle behind it): a device-resident
// table of items, blk_start[i] = first workgroup of item i (ascending, n_items + 1 entries); every item as transpose_vec_kernel
// (M, C multiples of 64 / 8; 16-byte aligned rows).
__global__ __launch_bounds__(256) void transpose_multi_kernel(const msclip_transpose_item* __restrict__ items,
                                                              const int* __restrict__ blk_start, int n_items) {
  __shared__ bf16_t tile[64][66];
  int lo = 0, hi = n_items;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const msclip_transpose_item it = items[lo];
  const int b = blockIdx.x - blk_start[lo], tx = it.M >> 6;
  const int by = b / tx, bx = b - by * tx;
  const int m0 = bx * 64, c0 = by * 64;
  const bf16_t* in = (const bf16_t*)it.in;
  bf16_t* out = (bf16_t*)it.out;
  const int sub = threadIdx.x >> 3, ch = threadIdx.x & 7;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = sub + 32 * i, m = m0 + r, c = c0 + ch * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < it.C) v = *(const uint4*)(in + (size_t)m * it.ldi + c);
    unsigned* d = (unsigned*)&tile[r][ch * 8];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = sub + 32 * i, c = c0 + r;
    if (c < it.C) {
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = tile[ch * 8 + j][r];
      uint4 v;
      v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
      v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
      *(uint4*)(out + (size_t)c * it.ldo + m0 + ch * 8) = v;
    }
  }
}

// ---- out[n] (+)= sum_m x[m][n]: bias gradients and the second stage of the LayerNorm parameter gradients.
// One block per 64 columns; 4 waves stride the rows, lanes own columns; fixed summation order (deterministic).
// Single-launch form of the chunked sums (round 5: the training step issued 260 second-stage launches per step).  Every
// workgroup of a column block publishes its chunk's partial row in `scratch`, takes a ticket from the column block's counter,
// and the workgroup that draws the LAST ticket folds all `chunks` partial rows -- in chunk order, whichever workgroup that is:
// the result is bitwise independent of the arrival order -- and re-arms the counter.  The XCDs' L2s are not coherent with each
// other: the partial rows are written and read with device-scope (sc1) accesses, which go through to the fabric, and a writer
// waits for its stores' acknowledgements before it draws its ticket.  NO fences: a device-scope release / acquire fence writes
// back and invalidates the XCD's whole L2 (`buffer_wbl2 sc1` / `buffer_inv sc1`) -- with ~3 000 workgroups per launch fencing
// beside the GEMMs' dirty lines the training step went from 58.5 to 68.9 ms (first version, measured).  ncols columns from col0.
__device__ __forceinline__ void colsum_publish(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float colsum_peek(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void colsum_fold_last(const float* __restrict__ scratch, float* __restrict__ out, unsigned* counter,
                                                 int chunks, int N, int col0, int ncols, int accumulate) {
  __shared__ int last_flag;
  __shared__ float fpart[4][64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's published partial sums have been acknowledged ...
  __syncthreads();                                   // ... and so have every other thread's of the workgroup
  if (threadIdx.x == 0)
    last_flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(chunks - 1);
  __syncthreads();
  if (!last_flag) return;
  for (int c0 = 0; c0 < ncols; c0 += 64) {           // 64 columns at a time: 4 waves stride the chunks, lanes own columns
    const int cl = c0 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (cl < ncols && col0 + cl < N) {
      const float* p = scratch + col0 + cl;
      int m = w;
      for (; m + 12 < chunks; m += 16) {
        s0 += colsum_peek(p + (size_t)m * N);
        s1 += colsum_peek(p + (size_t)(m + 4) * N);
        s2 += colsum_peek(p + (size_t)(m + 8) * N);
        s3 += colsum_peek(p + (size_t)(m + 12) * N);
      }
      for (; m < chunks; m += 4) s0 += colsum_peek(p + (size_t)m * N);
    }
    fpart[w][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && cl < ncols && col0 + cl < N) {
      const int l = threadIdx.x;
      const float t = (fpart[0][l] + fpart[1][l]) + (fpart[2][l] + fpart[3][l]);
      float* o = out + col0 + cl;
      *o = accumulate ? *o + t : t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the ring's next round
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, int M, int N,
                                                     int accumulate, int rows_per_chunk, float* __restrict__ final_out = nullptr,
                                                     unsigned* __restrict__ counters = nullptr, int final_accumulate = 0) {
  // blockIdx.y = row chunk: chunk c sums rows [c * rows_per_chunk, ...) into out[c * N + n] (a [chunks, N] partial matrix
  // that a second launch with one chunk folds); one chunk: out[n] directly.
  __shared__ float part[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per_chunk;
  const int m1 = min(M, m0 + rows_per_chunk);
  float s = 0.f;
  if (n < N) {
    auto ld1 = [&](int m) -> float {
      if constexpr (sizeof(T) == 2) return bf16_to_f32(x[(size_t)m * ld + n]);
      else return x[(size_t)m * ld + n];
    };
    // four rows in flight per wave (independent partial sums, folded in a fixed order): the short matrices this kernel
    // mostly sees (1024 partial rows of the LayerNorm parameter gradients, split-K partials) are latency-bound
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m0 + w;
    for (; m + 12 < m1; m += 16) {
      s0 += ld1(m);
      s1 += ld1(m + 4);
      s2 += ld1(m + 8);
      s3 += ld1(m + 12);
    }
    for (; m < m1; m += 4) s0 += ld1(m);
    s = (s0 + s1) + (s2 + s3);
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && n < N) {
    const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    float* o = out + (size_t)blockIdx.y * N + n;
    if (counters) colsum_publish(o, t);
    else *o = accumulate ? *o + t : t;
  }
  if (counters) colsum_fold_last(out, final_out, counters + blockIdx.x, (int)gridDim.y, N, blockIdx.x * 64, 64, final_accumulate);
}

// bf16 matrices with 16-byte row pieces (N % 8 == 0, ld % 8 == 0): a lane owns 8 columns; `cpr` lanes (a power of two
// <= 32) cover up to 256 columns of a row, the block's 256 / cpr row groups are folded through LDS.  Same chunking and
// fixed summation order as the scalar kernel.
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const bf16_t* __restrict__ x, int ld, float* __restrict__ out,
                                                            int M, int N, int accumulate, int rows_per_chunk, int cpr,
                                                            float* __restrict__ final_out = nullptr,
                                                            unsigned* __restrict__ counters = nullptr, int final_accumulate = 0) {
  __shared__ float part[256][9];
  const int tx = threadIdx.x & (cpr - 1), ty = threadIdx.x / cpr, ngrp = 256 / cpr;
  const int n = (blockIdx.x * cpr + tx) * 8;
  const int m0 = blockIdx.y * rows_per_chunk;
  const int m1 = min(M, m0 + rows_per_chunk);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (n < N) {
    for (int m = m0 + ty; m < m1; m += ngrp) {
      float f[8];
      unpack_bf16x8(*(const uint4*)(x + (size_t)m * ld + n), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[threadIdx.x][e] = s[e];
  __syncthreads();
  const int col = threadIdx.x;                         // up to cpr * 8 columns of the block, one per thread
  if (col < cpr * 8) {
    const int cx = col >> 3, ce = col & 7, nn = blockIdx.x * cpr * 8 + col;
    if (nn < N) {
      float t = 0.f;
      for (int r = 0; r < ngrp; ++r) t += part[r * cpr + cx][ce];
      float* o = out + (size_t)blockIdx.y * N + nn;
      if (counters) colsum_publish(o, t);
      else *o = accumulate ? *o + t : t;
    }
  }
  if (counters)
    colsum_fold_last(out, final_out, counters + blockIdx.x, (int)gridDim.y, N, blockIdx.x * cpr * 8, cpr * 8, final_accumulate);
}

// Few rows, very many columns (the S <= 32 fp32 partials of a split-K weight gradient, 0.6-2.4 M columns): one thread per
// 4 columns walks the rows in order -- 16-byte loads, the same fixed summation order.
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ x, size_t ld, float* __restrict__ out, int M,
                                                        size_t N4, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N4; i += (size_t)gridDim.x * 256) {
    float4 s = accumulate ? ((const float4*)out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {
      const float4 v = *(const float4*)(x + (size_t)m * ld + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    ((float4*)out)[i] = s;
  }
}

// ---- Many column sums in ONE launch (msclip_colsum_multi; round 5: the training step folded ~130 per-block partial matrices --
// LayerNorm parameter gradients, bias gradients -- with one msclip_colsum launch each, 1.1 ms of short launches on the main queue).
// The items travel in the kernel arguments (<= 96 per launch, no table upload); workgroup -> (item, 64-column block) by walking
// the items' block counts; thread (row group w, column c) adds rows w, w + 4, ... in order, the four row groups are added in
// order: a fixed summation order.  Optional scale of the first scale_n outputs (the packed in_proj bias: q rows carry 64^-0.5).
struct FoldArgs {
  msclip_fold_item it[96];
  int n;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(const FoldArgs a) {
  __shared__ float red[4][64];
  int b = blockIdx.x, i = 0;
  for (; i < a.n - 1; ++i) {
    const int nb = (a.it[i].N + 63) >> 6;
    if (b < nb) break;
    b -= nb;
  }
  const float* __restrict__ src = a.it[i].src;
  const int M = a.it[i].M, N = a.it[i].N;
  const size_t ld = (size_t)a.it[i].ld;
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6, col = b * 64 + c;
  float s = 0.f;
  if (col < N) {
    int m = w;
#pragma unroll 1
    for (; m + 28 < M; m += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(m + 4 * j) * ld + col];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; m < M; m += 4) s += src[(size_t)m * ld + col];
  }
  red[w][c] = s;
  __syncthreads();
  if (w == 0 && col < N) {
    float t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    if (col < a.it[i].scale_n) t *= a.it[i].scale;
    a.it[i].dst[col] = t;
  }
}

// ---- fp32 -> bf16 copy of a gradient matrix (the operand of its dgrad / wgrad GEMMs).
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int M,
                                                   int C4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)M * C4; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C4;
    const int c = (int)(i - m * C4) * 4;
    const float4 v = *(const float4*)(x + m * ldx + c);
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *(uint2*)(y + m * ldy + c) = o;
  }
}

// ---- the same copy, and the column sums of the fp32 matrix from the same pass (a residual-stream gradient is cast for its
// dgrad / wgrad GEMMs and summed over the tokens for the projection's bias gradient: one read instead of two).  Block b takes
// the rows b, b + gridDim.x, ...; thread t the columns 4t .. 4t+3 of every one of them (blockDim.x >= C/4); part[b][C] receives
// the block's sums (folded by msclip_colsum, fixed order: deterministic).
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                          int M, int C4, float* __restrict__ part, int skip_group) {
  const int t = threadIdx.x;
  if (t >= C4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = blockIdx.x; m < M; m += gridDim.x) {
    // skip_group g > 0: x holds g + 1 rows per sample, the first of which (the class token) is skipped: output row m reads row m + m / g + 1
    const size_t src = skip_group ? (size_t)m + m / skip_group + 1 : (size_t)m;
    const float4 v = *(const float4*)(x + src * ldx + t * 4);
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *(uint2*)(y + (size_t)m * ldy + t * 4) = o;
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
  }
  *(float4*)(part + ((size_t)blockIdx.x * C4 + t) * 4) = acc;
}

// ---- QuickGELU forward on a saved pre-activation (training keeps h for the backward) and its backward:
// y = h * sigma(1.702 h)  (M.py:222-224);  dh = dy * (sigma + 1.702 h sigma (1 - sigma)).
__global__ __launch_bounds__(256) void quickgelu_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 u = ((const uint4*)h)[i];
    float f[8];
    unpack_bf16x8(u, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = f[k] * sigmoidf_fast(1.702f * f[k]);
    uint4 o;
    o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]); o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
    ((uint4*)y)[i] = o;
  }
}
__global__ __launch_bounds__(256) void quickgelu_bwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ dy,
                                                            bf16_t* __restrict__ dh, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 uh = ((const uint4*)h)[i], ud = ((const uint4*)dy)[i];
    float fh[8], fd[8];
    unpack_bf16x8(uh, fh);
    unpack_bf16x8(ud, fd);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float s = sigmoidf_fast(1.702f * fh[k]);
      fd[k] = fd[k] * (s + 1.702f * fh[k] * s * (1.f - s));
    }
    uint4 o;
    o.x = pack_bf16x2(fd[0], fd[1]); o.y = pack_bf16x2(fd[2], fd[3]); o.z = pack_bf16x2(fd[4], fd[5]); o.w = pack_bf16x2(fd[6], fd[7]);
    ((uint4*)dh)[i] = o;
  }
}

// ---- LayerNorm backward (TF-style LN of M.py:204-219: biased variance, eps inside the sqrt, fp32 statistics).
// One wave per row, C = 64 * V4 * 4 (768 -> V4 = 3, 512 -> 2).  x row m comes from x[src(m)], src(m) = row_idx ? row_idx[m]
// : m * row_mul (the cls / EOT gathers of the heads); dy is bf16 or fp32 [M, C]; the result goes to dx[src(m)]
// (overwrite or +=).  dgamma / dbeta: every block accumulates its rows in registers and writes ONE partial row
// part[blockIdx][2][C]; msclip_colsum folds the partials (deterministic).  bf16 copy of dx optional (the operand of the
// next wgrad / dgrad GEMM when the residual gradient stream itself is what feeds it).
template <int V4, typename TDY>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ row_idx,
                                                     int row_mul, const TDY* __restrict__ dy, int lddy,
                                                     const float* __restrict__ gamma, float* __restrict__ dx, int lddx,
                                                     int accumulate, float* __restrict__ part, int M, float eps,
                                                     bf16_t* __restrict__ dxb, int lddxb, float* __restrict__ sum_part,
                                                     int sum_accumulate) {
  constexpr int C = 256 * V4;
  __shared__ float red[4][3][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 g[V4], dg[V4], db[V4], ds[V4];
#pragma unroll
  for (int v = 0; v < V4; ++v) {
    g[v] = *(const float4*)(gamma + v * 256 + lane * 4);
    dg[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    ds[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // Software-pipelined over the wave's rows (round 5): the x, dy and (when accumulating) dx pieces of the NEXT row are requested
  // before the current row's four wave reductions run -- one row at a time the kernel waited out two memory round trips per row
  // (3.5 TB/s of its bytes); the old dx value in particular was only asked for after the last reduction.
  typedef typename std::conditional<sizeof(TDY) == 2, uint2, float4>::type dy_t;
  float4 nx[V4], nold[V4];
  dy_t ndy[V4];
  const int stride = gridDim.x * 4;
  auto request = [&](int m) {
    const size_t src = row_idx ? (size_t)row_idx[m] : (size_t)m * row_mul;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      nx[v] = *(const float4*)(x + src * ldx + v * 256 + lane * 4);
      ndy[v] = *(const dy_t*)((const TDY*)dy + (size_t)m * lddy + v * 256 + lane * 4);
      if (accumulate) nold[v] = *(const float4*)(dx + src * lddx + v * 256 + lane * 4);
    }
  };
  int m = blockIdx.x * 4 + wave;
  if (m < M) request(m);
  for (; m < M; m += stride) {
    const size_t src = row_idx ? (size_t)row_idx[m] : (size_t)m * row_mul;
    float4 xv[V4], dv[V4], old[V4];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      xv[v] = nx[v];
      old[v] = nold[v];
      s += xv[v].x + xv[v].y + xv[v].z + xv[v].w;
      if constexpr (sizeof(TDY) == 2) {
        const uint2 u = ndy[v];
        dv[v] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u));
      } else {
        dv[v] = ndy[v];
      }
    }
    if (m + stride < M) request(m + stride);
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      xv[v].x -= mu; xv[v].y -= mu; xv[v].z -= mu; xv[v].w -= mu;
      q += xv[v].x * xv[v].x + xv[v].y * xv[v].y + xv[v].z * xv[v].z + xv[v].w * xv[v].w;
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / C) + eps);
    float a = 0.f, b = 0.f;                          // sum(dxhat), sum(dxhat * xhat)
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      xv[v].x *= rstd; xv[v].y *= rstd; xv[v].z *= rstd; xv[v].w *= rstd;      // xhat
      dg[v].x += dv[v].x * xv[v].x; dg[v].y += dv[v].y * xv[v].y; dg[v].z += dv[v].z * xv[v].z; dg[v].w += dv[v].w * xv[v].w;
      db[v].x += dv[v].x; db[v].y += dv[v].y; db[v].z += dv[v].z; db[v].w += dv[v].w;
      dv[v].x *= g[v].x; dv[v].y *= g[v].y; dv[v].z *= g[v].z; dv[v].w *= g[v].w;   // dxhat
      a += dv[v].x + dv[v].y + dv[v].z + dv[v].w;
      b += dv[v].x * xv[v].x + dv[v].y * xv[v].y + dv[v].z * xv[v].z + dv[v].w * xv[v].w;
    }
    a = wave_sum(a) * (1.f / C);
    b = wave_sum(b) * (1.f / C);
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      float4 r;
      r.x = rstd * (dv[v].x - a - xv[v].x * b); r.y = rstd * (dv[v].y - a - xv[v].y * b);
      r.z = rstd * (dv[v].z - a - xv[v].z * b); r.w = rstd * (dv[v].w - a - xv[v].w * b);
      if (accumulate) { r.x += old[v].x; r.y += old[v].y; r.z += old[v].z; r.w += old[v].w; }
      *(float4*)(dx + src * lddx + v * 256 + lane * 4) = r;
      if (dxb) {
        // the new residual-stream gradient is the NEXT projection's output gradient: its bf16 copy (the operand of that
        // projection's dgrad / wgrad GEMMs) and its column sums (that projection's bias gradient) leave with this pass
        uint2 o;
        o.x = pack_bf16x2(r.x, r.y);
        o.y = pack_bf16x2(r.z, r.w);
        *(uint2*)(dxb + (size_t)m * lddxb + v * 256 + lane * 4) = o;
        ds[v].x += r.x; ds[v].y += r.y; ds[v].z += r.z; ds[v].w += r.w;
      }
    }
  }
  if (sum_part) {
#pragma unroll
    for (int v = 0; v < V4; ++v) *(float4*)&red[wave][2][v * 256 + lane * 4] = ds[v];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const float t = red[0][2][c] + red[1][2][c] + red[2][2][c] + red[3][2][c];
      float* o = sum_part + (size_t)blockIdx.x * C + c;
      *o = sum_accumulate ? *o + t : t;
    }
  }
  if (part) {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      *(float4*)&red[wave][0][v * 256 + lane * 4] = dg[v];
      *(float4*)&red[wave][1][v * 256 + lane * 4] = db[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      const int which = i / C, c = i - which * C;
      part[((size_t)blockIdx.x * 2 + which) * C + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
  }
}

// ---- y = x / ||x||  backward:  dx = (dy - y (y . dy)) / ||x||   (M.py:2983, :3076).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                         int lddy, float* __restrict__ dx, int lddx, int M, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  float nn = 0.f, dot = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float xv = x[(size_t)m * ldx + e];
    nn += xv * xv;
    dot += xv * dy[(size_t)m * lddy + e];
  }
  nn = wave_sum(nn);
  dot = wave_sum(dot);
  const float inv = rsqrtf(nn);
  for (int e = lane; e < E; e += 64) {
    const float xv = x[(size_t)m * ldx + e];
    dx[(size_t)m * lddx + e] = inv * (dy[(size_t)m * lddy + e] - xv * dot * inv * inv);
  }
}

// ---- contrastive head backward, elementwise part.  S [R, N] = scale * A_loc @ B_all^T (fp32, from msclip_gemm);
// G[r][j] = w * (exp(S - lse_row[r]) + exp(S - lse_col[j]) - 2 [j == label_off + r]),  w = 1 / (2 N_global):
// dL/dS of the symmetric cross-entropy restricted to this rank's rows (rank-major labels, reference
// lib/utils/comm.py:150-153).  G is written as bf16 (operand of the dA = scale * G @ B_all GEMM); dscale_part[r] =
// sum_j G[r][j] * S[r][j] (its sum over r and ranks, divided by scale, is dL/dscale).  One wave per row.
__global__ __launch_bounds__(256) void clip_g_kernel(const float* __restrict__ S, int lds, const float* __restrict__ lse_row,
                                                     const float* __restrict__ lse_col, int label_off, float w,
                                                     bf16_t* __restrict__ G, int ldg, float* __restrict__ dscale_part, int R,
                                                     int N, int Npad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float lr = lse_row[r];
  const int label = label_off + r;
  float acc = 0.f;
  for (int j = lane; j < Npad; j += 64) {
    float g = 0.f;
    if (j < N) {
      const float s = S[(size_t)r * lds + j];
      g = w * (__expf(s - lr) + __expf(s - lse_col[j]) - (j == label ? 2.f : 0.f));
      acc += g * s;
    }
    G[(size_t)r * ldg + j] = f32_to_bf16(g);
  }
  acc = wave_sum(acc);
  if (dscale_part && lane == 0) dscale_part[r] = acc;
}

// ---- token embedding + positional embedding backward (M.py:3047-3048): dX rows of the text tokens scatter-add into
// dEmb[token] (fp32 atomics: captions share ids, padding id 0 above all) and sum over the batch into dPos[l].
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ dx,
                                                        int lddx, float* __restrict__ demb, float* __restrict__ dpos, int B,
                                                        int L, int C, int vocab) {
  const int row = blockIdx.x;                        // b * L + l
  const int l = row % L;
  long long t = tokens[row];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = dx[(size_t)row * lddx + c];
    atomicAdd(demb + (size_t)t * C + c, v);
    if (dpos) atomicAdd(dpos + (size_t)l * C + c, v);           // (NULL: the caller sums over the batch itself, in a fixed order)
  }
}

// Packed captions (msclip_embed_tokens_packed): dX row cu[b] + l belongs to token tokens[b][l], l < n_b.  demb: the same atomic
// scatter-add over the live rows only; dpos[l] = sum over the captions that HAVE a position l of their row, in caption
// order (fixed order: bitwise repeatable), one block per (position, 256-column chunk), lanes = 4 columns each.
__global__ __launch_bounds__(256) void embed_bwd_packed_kernel(const long long* __restrict__ tokens, const float* __restrict__ dx,
                                                               int lddx, const int* __restrict__ cu, float* __restrict__ demb, int L,
                                                               int C, int vocab) {
  const int slot = blockIdx.x;                       // b * L + l
  const int b = slot / L, l = slot - b * L;
  const int base = cu[b];
  if (l >= cu[b + 1] - base) return;
  long long t = tokens[slot];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const float* src = dx + (size_t)(base + l) * lddx;
  for (int c = threadIdx.x; c < C; c += 256) atomicAdd(demb + (size_t)t * C + c, src[c]);
}

// dpos[l] = sum over the captions that reach position l of their row l, in caption order within each quarter of the batch, the four
// quarters added in order (bitwise repeatable).  Workgroup = (position, 256 columns) x 4 caption groups; a caption's first row is
// cu[b] itself, so the rows of 8 captions are requested together (the first version walked all captions in one thread with a
// load per step: 0.4 ms on the lane stream at batch 512).
__global__ __launch_bounds__(256) void pos_bwd_packed_kernel(const float* __restrict__ dx, int lddx, const int* __restrict__ cu,
                                                             float* __restrict__ dpos, int B, int C) {
  __shared__ float4 red[4][64];
  const int l = blockIdx.x, lane = threadIdx.x & 63, grp = threadIdx.x >> 6, col = blockIdx.y * 256 + lane * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int per = (B + 3) / 4, b0 = grp * per, b1 = min(B, b0 + per);
  if (col < C) {
    for (int b = b0; b < b1; b += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b + j < b1) {
          const int base = cu[b + j], next = cu[b + j + 1];
          if (l < next - base) v[j] = *(const float4*)(dx + (size_t)(base + l) * lddx + col);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
  }
  red[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && col < C) {
    float4 t = red[0][lane];
#pragma unroll
    for (int g = 1; g < 4; ++g) { t.x += red[g][lane].x; t.y += red[g][lane].y; t.z += red[g][lane].z; t.w += red[g][lane].w; }
    *(float4*)(dpos + (size_t)l * C + col) = t;
  }
}

// ---- lateral adapter (M.py:1752-1778), the pieces the backward needs.
// adapter_sum: pre-LayerNorm sum  [cls; BN(dw3x3(grid))] + [cls * usecls; t]  (fp32 [B*L, C]), saved by the training
// forward;  adapter_dx: gradient wrt the incoming tokens from the gradient of that sum: cls row (1 + usecls) * d[0],
// grid rows = transposed depthwise 3x3 (BN scale folded in dww, as in the forward).  One block per token, lanes = channels.
__global__ __launch_bounds__(256) void adapter_sum_kernel(const float* __restrict__ xin, int ldx, const float* __restrict__ t,
                                                          int ldt, const float* __restrict__ dww, const float* __restrict__ dwb,
                                                          float* __restrict__ out, int ldo, int B, int L, int g, int C,
                                                          int usecls) {
  const int row = blockIdx.x, b = row / L, p = row - b * L;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v;
    if (p == 0) {
      v = xin[(size_t)row * ldx + c] * (usecls ? 2.f : 1.f);
    } else {
      const int py = (p - 1) / g, px = (p - 1) - py * g;
      float a = dwb[c];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = py + ky - 1, xx = px + kx - 1;
          if (yy >= 0 && yy < g && xx >= 0 && xx < g)
            a += dww[(ky * 3 + kx) * C + c] * xin[((size_t)b * L + 1 + yy * g + xx) * ldx + c];
        }
      v = a + t[((size_t)b * g * g + (p - 1)) * ldt + c];
    }
    out[(size_t)row * ldo + c] = v;
  }
}
__global__ __launch_bounds__(256) void adapter_dx_kernel(const float* __restrict__ dsum, int lds, const float* __restrict__ dww,
                                                         float* __restrict__ dx, int lddx, int B, int L, int g, int C,
                                                         int usecls) {
  const int row = blockIdx.x, b = row / L, p = row - b * L;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v;
    if (p == 0) {
      v = dsum[(size_t)row * lds + c] * (usecls ? 2.f : 1.f);
    } else {
      const int py = (p - 1) / g, px = (p - 1) - py * g;
      v = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = py - (ky - 1), xx = px - (kx - 1);         // output pixel that saw this one through tap (ky, kx)
          if (yy >= 0 && yy < g && xx >= 0 && xx < g)
            v += dww[(ky * 3 + kx) * C + c] * dsum[((size_t)b * L + 1 + yy * g + xx) * lds + c];
        }
    }
    dx[(size_t)row * lddx + c] = v;
  }
}

// 16-byte forms of the two kernels above (C, the leading dimensions % 4 == 0, 16-byte aligned bases): a thread owns four
// channels of a token row, so a 768-channel row is one pass of 192 threads with 9 + 9 float4 loads instead of three
// passes of 4-byte loads (0.7-1.0 TB/s -> the neighbours come from L2 either way, the request count was the limit).
template <bool DX>
__global__ __launch_bounds__(256) void adapter_grid_vec_kernel(const float* __restrict__ src, int lds, const float* __restrict__ t,
                                                               int ldt, const float* __restrict__ dww,
                                                               const float* __restrict__ dwb, float* __restrict__ out, int ldo,
                                                               int B, int L, int g, int C, int usecls) {
  const int row = blockIdx.x, b = row / L, p = row - b * L;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    float4 v;
    if (p == 0) {
      const float4 x = *(const float4*)(src + (size_t)row * lds + c);
      const float m = usecls ? 2.f : 1.f;
      v = make_float4(x.x * m, x.y * m, x.z * m, x.w * m);
    } else {
      const int py = (p - 1) / g, px = (p - 1) - py * g;
      v = DX ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(dwb + c);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          // forward: input pixel under tap (ky, kx); backward: output pixel that saw this one through tap (ky, kx)
          const int yy = DX ? py - (ky - 1) : py + ky - 1, xx = DX ? px - (kx - 1) : px + kx - 1;
          if (yy >= 0 && yy < g && xx >= 0 && xx < g) {
            const float4 w = *(const float4*)(dww + (ky * 3 + kx) * C + c);
            const float4 x = *(const float4*)(src + ((size_t)b * L + 1 + yy * g + xx) * lds + c);
            v.x = fmaf(w.x, x.x, v.x);
            v.y = fmaf(w.y, x.y, v.y);
            v.z = fmaf(w.z, x.z, v.z);
            v.w = fmaf(w.w, x.w, v.w);
          }
        }
      if (!DX) {
        const float4 tt = *(const float4*)(t + ((size_t)b * g * g + (p - 1)) * ldt + c);
        v.x += tt.x;
        v.y += tt.y;
        v.z += tt.z;
        v.w += tt.w;
      }
    }
    *(float4*)(out + (size_t)row * ldo + c) = v;
  }
}

// ---- AdamW (decoupled weight decay), one fused pass per parameter tensor; fp32 states.  One statement of the update with
// the contractions written out, shared by both kernels: their results are bitwise the same.
__device__ __forceinline__ void adamw_update(float gi, float& mi, float& vi, float& pi, float lr, float b1, float b2, float eps,
                                             float wd, float c1, float c2) {
  mi = __fmaf_rn(b1, mi, (1.f - b1) * gi);
  vi = __fmaf_rn(b2, vi, (1.f - b2) * gi * gi);
  const float upd = __fmaf_rn(wd, pi, mi * c1 / (sqrtf(vi * c2) + eps));
  pi = __fmaf_rn(-lr, upd, pi);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float c1, float c2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    adamw_update(g[i], mi, vi, pi, lr, b1, b2, eps, wd, c1, c2);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// ---- the same update over many tensors per launch: block b works on chunk (map[b] >> 8) of tensor (map[b] & 255).
constexpr int AW_TENSORS = 36, AW_BLOCKS = 400, AW_CHUNK = 32768;   // 36 x 64 B + 400 x 4 B of kernel arguments (< 4 KiB)
struct AdamwBatch {
  msclip_adamw_tensor t[AW_TENSORS];
  unsigned map[AW_BLOCKS];
};
static_assert(sizeof(AdamwBatch) <= 4000, "the tensor table travels in the kernel arguments");

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwBatch a, float b1, float b2, float eps, float c1, float c2) {
  const unsigned e = a.map[blockIdx.x];
  const msclip_adamw_tensor& t = a.t[e & 255u];
  const size_t lo = (size_t)(e >> 8) * AW_CHUNK;
  const size_t left = (size_t)t.n - lo;
  const int cnt = left < (size_t)AW_CHUNK ? (int)left : AW_CHUNK;
  float* __restrict__ p = t.p + lo;
  const float* __restrict__ g = t.g + lo;
  float* __restrict__ m = t.m + lo;
  float* __restrict__ v = t.v + lo;
  const float lr = t.lr, wd = t.weight_decay;
  auto upd = [&](float gi, float& mi, float& vi, float& pi) { adamw_update(gi, mi, vi, pi, lr, b1, b2, eps, wd, c1, c2); };
  // packed copy of the new values (the engine's GEMM operand): same rounding as a cast of the updated tensor
  bf16_t* __restrict__ pkb = t.pk && !t.pk_f32 ? (bf16_t*)t.pk + lo : nullptr;
  float* __restrict__ pkf = t.pk && t.pk_f32 ? (float*)t.pk + lo : nullptr;
  const float ps = t.pk_scale;
  int i0 = 0;
  if (!(((size_t)p | (size_t)g | (size_t)m | (size_t)v | (size_t)pkf) & 15) && !((size_t)pkb & 7)) {
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 g4 = ((const float4*)g)[i];
      float4 m4 = ((float4*)m)[i], v4 = ((float4*)v)[i], p4 = ((float4*)p)[i];
      upd(g4.x, m4.x, v4.x, p4.x);
      upd(g4.y, m4.y, v4.y, p4.y);
      upd(g4.z, m4.z, v4.z, p4.z);
      upd(g4.w, m4.w, v4.w, p4.w);
      ((float4*)m)[i] = m4;
      ((float4*)v)[i] = v4;
      ((float4*)p)[i] = p4;
      if (pkb) ((uint2*)pkb)[i] = make_uint2(pack_bf16x2(p4.x * ps, p4.y * ps), pack_bf16x2(p4.z * ps, p4.w * ps));
      if (pkf) ((float4*)pkf)[i] = make_float4(p4.x * ps, p4.y * ps, p4.z * ps, p4.w * ps);
    }
    i0 = n4 << 2;
  }
  for (int i = i0 + threadIdx.x; i < cnt; i += 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    upd(g[i], mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (pkb) pkb[i] = f32_to_bf16(pi * ps);
    if (pkf) pkf[i] = pi * ps;
  }
}

}  // namespace

static int grid_for(size_t n, int per_block, int cap = 4096) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

extern "C" int msclip_transpose_bf16(const void* in, int ldi, void* out, int ldo, int M, int C, int Mpad, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_transpose_bf16, stream, in, ldi, out, ldo, M, C, Mpad);
  if (!in || !out || M <= 0 || C <= 0 || Mpad < M || ldo < Mpad || ldi < C) return MSCLIP_EINVAL;
  if (!(C % 8) && !(ldi % 8) && !(ldo % 8) && !(Mpad % 64) && !((size_t)in % 16) && !((size_t)out % 16))
    hipLaunchKernelGGL(transpose_vec_kernel, dim3(Mpad / 64, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, ldi, (bf16_t*)out, ldo, M, C);
  else
    hipLaunchKernelGGL(transpose_kernel, dim3((Mpad + 63) / 64, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, ldi, (bf16_t*)out, ldo, M, C, Mpad);
  return msclip_launch_status();
}

extern "C" int msclip_colsum_multi(const msclip_fold_item* items, int n_items, void* stream) {
  MSCLIP_PLAN_UNSUPPORTED(msclip_colsum_multi);
  if (!items || n_items <= 0) return MSCLIP_EINVAL;
  for (int i = 0; i < n_items; ++i)
    if (!items[i].src || !items[i].dst || items[i].M <= 0 || items[i].N <= 0 || items[i].ld < items[i].N || items[i].scale_n < 0 ||
        items[i].scale_n > items[i].N)
      return MSCLIP_EINVAL;
  for (int i0 = 0; i0 < n_items; i0 += 96) {
    FoldArgs a;
    a.n = n_items - i0 < 96 ? n_items - i0 : 96;
    long long blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      a.it[i] = items[i0 + i];
      blocks += (items[i0 + i].N + 63) / 64;
    }
    if (blocks > 0x7fffffffll) return MSCLIP_EINVAL;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  return msclip_launch_status();
}

extern "C" int msclip_transpose_bf16_multi(const msclip_transpose_item* items_dev, const int* blk_start_dev, int n_items,
                                           int n_blocks, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_transpose_bf16_multi, stream, items_dev, blk_start_dev, n_items, n_blocks);
  if (!items_dev || !blk_start_dev || n_items <= 0 || n_blocks <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, blk_start_dev, n_items);
  return msclip_launch_status();
}

extern "C" int msclip_cast_bf16(const float* x, int ldx, void* y, int ldy, int M, int C, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_cast_bf16, stream, x, ldx, y, ldy, M, C);
  if (!x || !y || M <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for((size_t)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     (bf16_t*)y, ldy, M, C / 4);
  return msclip_launch_status();
}

extern "C" int msclip_cast_bf16_colsum(const float* x, int ldx, void* y, int ldy, int M, int C, float* part, int part_blocks,
                                       int skip_group, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_cast_bf16_colsum, stream, x, ldx, y, ldy, M, C, part, part_blocks, skip_group);
  if (!x || !y || !part || M <= 0 || C <= 0 || C > 1024 || (C & 3) || (ldx & 3) || (ldy & 3) || part_blocks < 1 ||
      ((size_t)part & 15) || skip_group < 0 || (skip_group && M % skip_group))
    return MSCLIP_EINVAL;
  hipLaunchKernelGGL(cast_colsum_kernel, dim3(part_blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)y, ldy, M, C / 4,
                     part, skip_group);
  return msclip_launch_status();
}

// Ticket counters of the single-launch column sums: one ring of 2^18 zero words per device, allocated on first use.
static bool colsum_two_stage() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MSCLIP_COLSUM_TWO_STAGE");
    v = e && e[0] == '1';
  }
  return v != 0;
}
static unsigned* colsum_counters(int n) {
  constexpr int RING = 1 << 18, MAXDEV = 16;
  static unsigned* ring[MAXDEV] = {};
  static int next[MAXDEV] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV || n > RING) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!ring[dev]) {
    if (hipMalloc((void**)&ring[dev], RING * sizeof(unsigned)) != hipSuccess) { ring[dev] = nullptr; return nullptr; }
    if (hipMemset(ring[dev], 0, RING * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
  }
  if (next[dev] + n > RING) next[dev] = 0;
  unsigned* p = ring[dev] + next[dev];
  next[dev] += n;
  return p;
}

// Everything this library allocates lazily on a device (today: the ticket-counter ring of the single-launch column sums), done
// up front -- a first use inside a stream capture or a launch-plan replay must not hipMalloc / synchronise.
extern "C" int msclip_prepare_device(void) { return colsum_counters(0) ? MSCLIP_OK : MSCLIP_ELAUNCH; }

extern "C" int msclip_colsum(const void* x, int ld, int is_f32, float* out, int M, int N, int accumulate, float* scratch,
                             int chunks, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_colsum, stream, x, ld, is_f32, out, M, N, accumulate, scratch, chunks);
  if (!x || !out || M <= 0 || N <= 0 || ld < N || chunks < 1 || (chunks > 1 && !scratch)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (is_f32 && M <= 32 && N >= 16384 && !(N % 4) && !(ld % 4) && !((size_t)x % 16) && !((size_t)out % 16)) {
    hipLaunchKernelGGL(fold_rows_kernel, dim3(grid_for((size_t)N / 4, 256, 8192)), dim3(256), 0, st, (const float*)x, (size_t)ld,
                       out, M, (size_t)N / 4, accumulate);
    return msclip_launch_status();
  }
  const int rpc = (M + chunks - 1) / chunks;
  float* first = chunks > 1 ? scratch : out;         // [chunks, N] partials, or the result itself
  int cpr = 1;
  while (cpr < 32 && cpr * 8 < N) cpr *= 2;
  const bool wide = !is_f32 && !(N % 8) && !(ld % 8) && !((size_t)x % 16);
  const int colblocks = wide ? (N + cpr * 8 - 1) / (cpr * 8) : (N + 63) / 64;
  // chunks > 1: ONE launch -- the last workgroup of a column block folds the partial rows (colsum_fold_last); its ticket counters
  // come from a per-device ring of zero-initialised words (a launch's counters are re-armed by the launch itself; a slot comes
  // round again 2^18 / colblocks launches later).  MSCLIP_COLSUM_TWO_STAGE=1: the two-launch form of rounds 1-4.
  unsigned* counters = chunks > 1 && !colsum_two_stage() ? colsum_counters(colblocks) : nullptr;
  if (is_f32)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(colblocks, chunks), dim3(256), 0, st, (const float*)x, ld, first, M, N,
                       chunks > 1 ? 0 : accumulate, rpc, out, counters, accumulate);
  else if (wide)
    hipLaunchKernelGGL(colsum_bf16x8_kernel, dim3(colblocks, chunks), dim3(256), 0, st, (const bf16_t*)x,
                       ld, first, M, N, chunks > 1 ? 0 : accumulate, rpc, cpr, out, counters, accumulate);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(colblocks, chunks), dim3(256), 0, st, (const bf16_t*)x, ld, first, M,
                       N, chunks > 1 ? 0 : accumulate, rpc, out, counters, accumulate);
  if (chunks > 1 && !counters)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((N + 63) / 64, 1), dim3(256), 0, st, (const float*)scratch, N, out, chunks, N,
                       accumulate, chunks);
  return msclip_launch_status();
}

extern "C" int msclip_quickgelu(const void* h, void* y, long long n, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_quickgelu, stream, h, y, n);
  if (!h || !y || n <= 0 || (n & 7)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(quickgelu_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                     (bf16_t*)y, (size_t)(n / 8));
  return msclip_launch_status();
}

extern "C" int msclip_quickgelu_bwd(const void* h, const void* dy, void* dh, long long n, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_quickgelu_bwd, stream, h, dy, dh, n);
  if (!h || !dy || !dh || n <= 0 || (n & 7)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(quickgelu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                     (const bf16_t*)dy, (bf16_t*)dh, (size_t)(n / 8));
  return msclip_launch_status();
}

extern "C" int msclip_layernorm_bwd(const float* x, int ldx, const int* row_idx, int row_mul, const void* dy, int lddy,
                                    int dy_is_f32, const float* gamma, float* dx, int lddx, int accumulate, float* part,
                                    int part_blocks, int M, int C, float eps, void* dxb, int lddxb, float* sum_part,
                                    int sum_accumulate, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_layernorm_bwd, stream, x, ldx, row_idx, row_mul, dy, lddy, dy_is_f32, gamma, dx, lddx, accumulate, part, part_blocks, M, C, eps, dxb, lddxb, sum_part, sum_accumulate);
  if (!x || !dy || !gamma || !dx || M <= 0 || (C != 512 && C != 768) || part_blocks < 1) return MSCLIP_EINVAL;
  // bf16 copy + column sums of the written dx rows: the plain row mapping only (dxb row m = dx row m), both or neither
  if ((dxb != nullptr) != (sum_part != nullptr) || (dxb && (row_idx || row_mul != 1 || (lddxb % 4)))) return MSCLIP_EINVAL;
  int blocks = (M + 3) / 4;
  if (blocks > part_blocks) blocks = part_blocks;
  if (sum_part && !sum_accumulate && blocks < part_blocks) {
    if (hipMemsetAsync(sum_part + (size_t)blocks * C, 0, (size_t)(part_blocks - blocks) * C * sizeof(float),
                       (hipStream_t)stream) != hipSuccess) return MSCLIP_ELAUNCH;
  }
  if (part && blocks < part_blocks) {                // unused partial rows must not hold garbage
    if (hipMemsetAsync(part + (size_t)blocks * 2 * C, 0, (size_t)(part_blocks - blocks) * 2 * C * sizeof(float),
                       (hipStream_t)stream) != hipSuccess) return MSCLIP_ELAUNCH;
  }
  hipStream_t st = (hipStream_t)stream;
#define LNB(V4, T) hipLaunchKernelGGL((ln_bwd_kernel<V4, T>), dim3(blocks), dim3(256), 0, st, x, ldx, row_idx, row_mul, \
                                      (const T*)dy, lddy, gamma, dx, lddx, accumulate, part, M, eps, (bf16_t*)dxb, lddxb, \
                                      sum_part, sum_accumulate)
  if (C == 768) { if (dy_is_f32) LNB(3, float); else LNB(3, bf16_t); }
  else { if (dy_is_f32) LNB(2, float); else LNB(2, bf16_t); }
#undef LNB
  return msclip_launch_status();
}

extern "C" int msclip_l2norm_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, int M, int E,
                                 void* stream) {
  MSCLIP_PLAN_HOOK(msclip_l2norm_bwd, stream, x, ldx, dy, lddy, dx, lddx, M, E);
  if (!x || !dy || !dx || M <= 0 || E <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, dx, lddx, M, E);
  return msclip_launch_status();
}

extern "C" int msclip_clip_loss_bwd_g(const float* S, int lds, const float* lse_row, const float* lse_col, int label_off,
                                      float w, void* G, int ldg, float* dscale_part, int R, int N, int Npad, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_clip_loss_bwd_g, stream, S, lds, lse_row, lse_col, label_off, w, G, ldg, dscale_part, R, N, Npad);
  if (!S || !lse_row || !lse_col || !G || R <= 0 || N <= 0 || Npad < N || ldg < Npad || lds < N) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(clip_g_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, lds, lse_row, lse_col, label_off,
                     w, (bf16_t*)G, ldg, dscale_part, R, N, Npad);
  return msclip_launch_status();
}

extern "C" int msclip_embed_tokens_bwd(const long long* tokens, const float* dx, int lddx, float* demb, float* dpos, int B,
                                       int L, int C, int vocab, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_embed_tokens_bwd, stream, tokens, dx, lddx, demb, dpos, B, L, C, vocab);
  if (!tokens || !dx || !demb || B <= 0 || L <= 0 || C <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, tokens, dx, lddx, demb, dpos, B, L, C,
                     vocab);
  return msclip_launch_status();
}

extern "C" int msclip_embed_tokens_bwd_packed(const long long* tokens, const float* dx, int lddx, const int* cu, float* demb,
                                              float* dpos, int B, int L, int C, int vocab, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_embed_tokens_bwd_packed, stream, tokens, dx, lddx, cu, demb, dpos, B, L, C, vocab);
  if (!tokens || !dx || !cu || !demb || B <= 0 || L <= 0 || C <= 0 || (C % 4) || (lddx % 4)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_bwd_packed_kernel, dim3(B * L), dim3(256), 0, st, tokens, dx, lddx, cu, demb, L, C, vocab);
  if (dpos) hipLaunchKernelGGL(pos_bwd_packed_kernel, dim3(L, (C + 255) / 256), dim3(256), 0, st, dx, lddx, cu, dpos, B, C);
  return msclip_launch_status();
}

extern "C" int msclip_adapter_sum(const float* xin, int ldx, const float* t, int ldt, const float* dww, const float* dwb,
                                  float* out, int ldo, int B, int L, int g, int C, int usecls, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_adapter_sum, stream, xin, ldx, t, ldt, dww, dwb, out, ldo, B, L, g, C, usecls);
  if (!xin || !t || !dww || !dwb || !out || B <= 0 || L != g * g + 1) return MSCLIP_EINVAL;
  if (!((C | ldx | ldt | ldo) & 3) && !(((size_t)xin | (size_t)t | (size_t)dww | (size_t)dwb | (size_t)out) & 15))
    hipLaunchKernelGGL(adapter_grid_vec_kernel<false>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, xin, ldx, t, ldt, dww, dwb,
                       out, ldo, B, L, g, C, usecls);
  else
    hipLaunchKernelGGL(adapter_sum_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, xin, ldx, t, ldt, dww, dwb, out, ldo,
                       B, L, g, C, usecls);
  return msclip_launch_status();
}

extern "C" int msclip_adapter_dx(const float* dsum, int lds, const float* dww, float* dx, int lddx, int B, int L, int g, int C,
                                 int usecls, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_adapter_dx, stream, dsum, lds, dww, dx, lddx, B, L, g, C, usecls);
  if (!dsum || !dww || !dx || B <= 0 || L != g * g + 1) return MSCLIP_EINVAL;
  if (!((C | lds | lddx) & 3) && !(((size_t)dsum | (size_t)dww | (size_t)dx) & 15))
    hipLaunchKernelGGL(adapter_grid_vec_kernel<true>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, dsum, lds,
                       (const float*)nullptr, 0, dww, (const float*)nullptr, dx, lddx, B, L, g, C, usecls);
  else
    hipLaunchKernelGGL(adapter_dx_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, dsum, lds, dww, dx, lddx, B, L, g, C,
                       usecls);
  return msclip_launch_status();
}

extern "C" int msclip_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_adamw, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step);
  if (!p || !g || !m || !v || n <= 0 || step < 1) return MSCLIP_EINVAL;
  const float c1 = 1.f / (1.f - powf(beta1, (float)step)), c2 = 1.f / (1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, lr,
                     beta1, beta2, eps, weight_decay, c1, c2);
  return msclip_launch_status();
}

extern "C" int msclip_adamw_multi(const msclip_adamw_tensor* tensors, int count, float beta1, float beta2, float eps, int step,
                                  void* stream) {
  MSCLIP_PLAN_UNSUPPORTED(msclip_adamw_multi);
  if (!tensors || count < 0 || step < 1) return MSCLIP_EINVAL;
  for (int i = 0; i < count; ++i)
    if (!tensors[i].p || !tensors[i].g || !tensors[i].m || !tensors[i].v || tensors[i].n <= 0 ||
        (tensors[i].pk && (tensors[i].pk_f32 < 0 || tensors[i].pk_f32 > 1)))
      return MSCLIP_EINVAL;
  const float c1 = 1.f / (1.f - powf(beta1, (float)step)), c2 = 1.f / (1.f - powf(beta2, (float)step));
  AdamwBatch b;
  int nt = 0, nb = 0;
  auto flush = [&]() {
    if (nb) hipLaunchKernelGGL(adamw_multi_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, b, beta1, beta2, eps, c1, c2);
    nt = nb = 0;
  };
  for (int i = 0; i < count; ++i) {
    const long long chunks = (tensors[i].n + AW_CHUNK - 1) / AW_CHUNK;
    long long c = 0;
    while (c < chunks) {
      if (nt == AW_TENSORS || nb == AW_BLOCKS) flush();
      b.t[nt] = tensors[i];
      // a tensor that continues in the next launch restarts there at chunk c: shift its base instead of carrying an offset
      b.t[nt].p += c * AW_CHUNK;
      b.t[nt].g += c * AW_CHUNK;
      b.t[nt].m += c * AW_CHUNK;
      b.t[nt].v += c * AW_CHUNK;
      b.t[nt].n -= c * AW_CHUNK;
      if (b.t[nt].pk) b.t[nt].pk = (char*)b.t[nt].pk + (size_t)c * AW_CHUNK * (b.t[nt].pk_f32 ? 4 : 2);
      long long local = 0;
      while (c < chunks && nb < AW_BLOCKS) {
        b.map[nb++] = (unsigned)nt | ((unsigned)local << 8);
        ++local;
        ++c;
      }
      ++nt;
    }
  }
  flush();
  return msclip_launch_status();
}
