// Row-wise HBM-bound kernels: one wave64 per token row, 16-byte accesses,
// reductions by lane shuffles (no LDS).  TF-style LayerNorm with fp32
// statistics and eps inside the sqrt (reference
// lib/models/clip_openai_pe_res_v1.py:204-219), token embedding, the fused
// lateral-adapter combine (ibid. 1752-1778) and the L2 normalisation of the
// projected features (ibid. 2983, 3076).
#include <stdlib.h>
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

constexpr int WPB = 4;  // waves (rows) per block

template <int NV>
__device__ __forceinline__ float ln_core(float4 (&v)[NV], const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, int lane) {
  constexpr float invC = 1.f / (NV * 256);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float var = wave_sum(q) * invC;
  const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 g = *(const float4*)(gamma + i * 256 + lane * 4);
    const float4 b = *(const float4*)(beta + i * 256 + lane * 4);
    v[i].x = g.x * (v[i].x * rstd) + b.x;
    v[i].y = g.y * (v[i].y * rstd) + b.y;
    v[i].z = g.z * (v[i].z * rstd) + b.z;
    v[i].w = g.w * (v[i].w * rstd) + b.w;
  }
  return mean;
}

template <int NV>
__device__ __forceinline__ void store_row(const float4 (&v)[NV], void* out, size_t row, int ldo, int out_kind,
                                          int lane) {
  if (out_kind == 1) {
    float* o = (float*)out + row * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) *(float4*)(o + i * 256 + lane * 4) = v[i];
  } else {
    bf16_t* o = (bf16_t*)out + row * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      uint2 p;
      p.x = pack_bf16x2(v[i].x, v[i].y);
      p.y = pack_bf16x2(v[i].z, v[i].w);
      *(uint2*)(o + i * 256 + lane * 4) = p;
    }
  }
}

// LayerNorm of contiguous rows that also leaves the LayerNorm fold's per-row state (msclip_layernorm_stats): center[m] = mean,
// rowstat[m] = (1, 0): the consuming projection reads `out` as it is, the next producing one centres its bf16 copy on the mean.
template <int NV>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, void* out, int ldo, int out_kind,
                                                       float* __restrict__ raw_out, int ld_raw, float* __restrict__ center,
                                                       float* __restrict__ rowstat, int M, float eps,
                                                       const int* __restrict__ m_dev) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m_dev) M = min(M, *m_dev);                      // device-side row count (packed captions): the launch covers the upper bound
  if (m >= M) return;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(x + (size_t)m * ldx + i * 256 + lane * 4);
  if (raw_out) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *(float4*)(raw_out + (size_t)m * ld_raw + i * 256 + lane * 4) = v[i];
  }
  const float mean = ln_core<NV>(v, gamma, beta, eps, lane);
  store_row<NV>(v, out, (size_t)m, ldo, out_kind, lane);
  if (lane == 0) {
    if (center) center[m] = mean;
    if (rowstat) *(float2*)(rowstat + 2 * (size_t)m) = make_float2(1.f, 0.f);
  }
}

// Row statistics from the producing GEMM's per-64-column partial sums of (x - center[m]) (msclip_rowstat_finalize): one
// thread per row, the groups folded in index order (deterministic).
template <int G4>                                      // G4 = groups / 2 float4 pieces per row (groups = C / 64: 12 or 16), or 0 = any
__global__ __launch_bounds__(256) void rowstat_finalize_kernel(const float* __restrict__ part, int groups, float* __restrict__ center,
                                                               float* __restrict__ rowstat, int M, float invC, float eps,
                                                               const int* __restrict__ m_dev) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m_dev) M = min(M, *m_dev);
  if (m >= M) return;
  float s = 0.f, q = 0.f;
  if (G4 > 0) {                                       // all of the row's partials in flight at once
    const float4* p = (const float4*)(part + (size_t)m * (G4 * 4));
    float4 v[G4 > 0 ? G4 : 1];
#pragma unroll
    for (int g = 0; g < G4; ++g) v[g] = p[g];
#pragma unroll
    for (int g = 0; g < G4; ++g) {                    // (group order: same sums as the generic loop)
      s += v[g].x; q += v[g].y;
      s += v[g].z; q += v[g].w;
    }
  } else {
    const float2* p = (const float2*)(part + (size_t)m * groups * 2);
    for (int g = 0; g < groups; ++g) {
      const float2 v = p[g];
      s += v.x;
      q += v.y;
    }
  }
  const float mu = s * invC;
  const float var = fmaxf(q * invC - mu * mu, 0.f);
  const float rstd = 1.f / sqrtf(var + eps);
  *(float2*)(rowstat + 2 * (size_t)m) = make_float2(rstd, mu * rstd);
  center[m] += mu;
}

// y[m] = LN(x[src(m)]) ; src(m) = row_idx ? row_idx[m] : m * row_mul + row_add; rows >= split use (gamma2, beta2)
template <int NV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ row_idx,
                                                 int row_mul, int row_add, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, void* out, int ldo, int out_kind,
                                                 float* __restrict__ raw_out, int ld_raw, int M, float eps,
                                                 const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                 int split) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= M) return;
  if (m >= split) {                                  // second parameter set (the other modality's rows); wave-uniform
    gamma = gamma2;
    beta = beta2;
  }
  const size_t src = row_idx ? (size_t)row_idx[m] : (size_t)m * row_mul + row_add;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(x + src * ldx + i * 256 + lane * 4);
  if (raw_out) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *(float4*)(raw_out + (size_t)m * ld_raw + i * 256 + lane * 4) = v[i];
  }
  ln_core<NV>(v, gamma, beta, eps, lane);
  store_row<NV>(v, out, (size_t)m, ldo, out_kind, lane);
}

// Two rows per wave for the big contiguous launches (the per-layer LayerNorms over all token rows): both rows' loads
// are in flight together and gamma / beta are fetched once for the pair (per row they were two thirds of the wave's
// L1 traffic).  Rows m0 = 2*pair, m0 + 1; a pair that straddles `split` falls back to per-row parameters.
template <int NV>
__global__ __launch_bounds__(256) void ln_pair_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, void* out, int ldo, int out_kind,
                                                      int M, float eps, const float* __restrict__ gamma2,
                                                      const float* __restrict__ beta2, int split) {
  const int lane = threadIdx.x & 63;
  const int m0 = (blockIdx.x * WPB + (threadIdx.x >> 6)) * 2;
  if (m0 >= M) return;
  const bool two = m0 + 1 < M;
  const int m1 = two ? m0 + 1 : m0;
  constexpr float invC = 1.f / (NV * 256);
  float4 a[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a[i] = *(const float4*)(x + (size_t)m0 * ldx + i * 256 + lane * 4);
    b[i] = *(const float4*)(x + (size_t)m1 * ldx + i * 256 + lane * 4);
  }
  const float* g0 = m0 >= split ? gamma2 : gamma;
  const float* b0 = m0 >= split ? beta2 : beta;
  const float* g1 = m1 >= split ? gamma2 : gamma;
  const float* b1 = m1 >= split ? beta2 : beta;
  float4 gv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gv[i] = *(const float4*)(g0 + i * 256 + lane * 4);
    bv[i] = *(const float4*)(b0 + i * 256 + lane * 4);
  }
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    sa += (a[i].x + a[i].y) + (a[i].z + a[i].w);
    sb += (b[i].x + b[i].y) + (b[i].z + b[i].w);
  }
  const float ma = wave_sum(sa) * invC, mb = wave_sum(sb) * invC;
  float qa = 0.f, qb = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a[i].x -= ma; a[i].y -= ma; a[i].z -= ma; a[i].w -= ma;
    b[i].x -= mb; b[i].y -= mb; b[i].z -= mb; b[i].w -= mb;
    qa += (a[i].x * a[i].x + a[i].y * a[i].y) + (a[i].z * a[i].z + a[i].w * a[i].w);
    qb += (b[i].x * b[i].x + b[i].y * b[i].y) + (b[i].z * b[i].z + b[i].w * b[i].w);
  }
  const float ra = 1.f / sqrtf(wave_sum(qa) * invC + eps), rb = 1.f / sqrtf(wave_sum(qb) * invC + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    a[i].x = gv[i].x * (a[i].x * ra) + bv[i].x; a[i].y = gv[i].y * (a[i].y * ra) + bv[i].y;
    a[i].z = gv[i].z * (a[i].z * ra) + bv[i].z; a[i].w = gv[i].w * (a[i].w * ra) + bv[i].w;
  }
  store_row<NV>(a, out, (size_t)m0, ldo, out_kind, lane);
  if (!two) return;
  if (g1 != g0) {                                     // the pair straddles the modality boundary (wave-uniform)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      gv[i] = *(const float4*)(g1 + i * 256 + lane * 4);
      bv[i] = *(const float4*)(b1 + i * 256 + lane * 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    b[i].x = gv[i].x * (b[i].x * rb) + bv[i].x; b[i].y = gv[i].y * (b[i].y * rb) + bv[i].y;
    b[i].z = gv[i].z * (b[i].z * rb) + bv[i].z; b[i].w = gv[i].w * (b[i].w * rb) + bv[i].w;
  }
  store_row<NV>(b, out, (size_t)m1, ldo, out_kind, lane);
}

// x[row_base + b*L + l] = emb[tok[b, l]] + pos[l]      (reference :3047-3048)
template <int NV>
__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                    const float* __restrict__ pos, float* __restrict__ x, int ldx,
                                                    int rows, int L, int vocab) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int l = m % L;
  long long t = tok[m];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const float* e = emb + (size_t)t * (NV * 256);
  const float* p = pos + (size_t)l * (NV * 256);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 a = *(const float4*)(e + i * 256 + lane * 4);
    const float4 b = *(const float4*)(p + i * 256 + lane * 4);
    *(float4*)(x + (size_t)m * ldx + i * 256 + lane * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// eot_row[b] = row_base + b*L + argmax_l tok[b, l]   (first maximum, like torch.argmax; reference :3057-3060)
__global__ __launch_bounds__(64) void eot_kernel(const long long* __restrict__ tok, int* __restrict__ eot_row, int B,
                                                 int L, int row_base) {
  const int b = blockIdx.x, lane = threadIdx.x;
  long long best = -0x7fffffffffffffffLL;
  int bi = 0x7fffffff;
  for (int l = lane; l < L; l += 64) {
    const long long t = tok[(size_t)b * L + l];
    if (t > best) { best = t; bi = l; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const long long ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) eot_row[b] = row_base + b * L + bi;
}

// ---- Packed captions.  Under the causal mask (reference :2965-2971) a token after a caption's EOT position cannot reach the
// EOT row that encode_text returns (:3057-3060) in ANY block, so only the n_b = argmax_l tok[b, l] + 1 leading rows of caption
// b are live.  len_kernel: n_b (first maximum, like torch.argmax); scan_kernel: cu[b] = sum_{j < b} n_j, cu[B] = total,
// cu[B + 1] = max_b n_b, eot_row[b] = row_base + cu[b] + n_b - 1 (one workgroup: B is a batch, not a dataset).
__global__ __launch_bounds__(64) void len_kernel(const long long* __restrict__ tok, int* __restrict__ len, int B, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  long long best = -0x7fffffffffffffffLL;
  int bi = 0x7fffffff;
  for (int l = lane; l < L; l += 64) {
    const long long t = tok[(size_t)b * L + l];
    if (t > best) { best = t; bi = l; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const long long ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) len[b] = bi + 1;
}

__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ len, int* __restrict__ cu, int* __restrict__ eot_row,
                                                    int B, int row_base, int* __restrict__ dims, int pad_to, int cap_rows) {
  __shared__ int wsum[16], wmax[16];
  __shared__ int carry_s, max_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_s = 0; max_s = 0; }
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    const int n = b < B ? len[b] : 0;
    int inc = n;                                       // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    int mx = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane == 63) wsum[wave] = inc;
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    int before = carry_s;
    for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
    if (b < B) {
      const int base = before + inc - n;
      cu[b] = base;
      if (eot_row) eot_row[b] = row_base + base + n - 1;
    }
    __syncthreads();
    if (tid == 0) {
      int s = carry_s, m = max_s;
      for (int w2 = 0; w2 < 16; ++w2) { s += wsum[w2]; m = max(m, wmax[w2]); }
      carry_s = s;
      max_s = m;
    }
    __syncthreads();
  }
  if (tid == 0) {
    cu[B] = carry_s;
    cu[B + 1] = max_s;
    if (dims) {
      // the row counts every launch over the text rows reads on the DEVICE (msclip_text_lengths): no host read sizes them
      int padded = pad_to > 0 ? (carry_s + pad_to - 1) / pad_to * pad_to : carry_s;
      if (padded > cap_rows) padded = carry_s;          // (cannot happen when cap_rows is a multiple of pad_to that holds every caption)
      dims[0] = carry_s;                                // live text rows
      dims[1] = max_s;                                  // longest caption
      dims[2] = padded;                                 // text rows incl. the tile padding
      dims[3] = row_base + padded;                      // rows of the whole token matrix
      dims[4] = padded - carry_s;                       // padding rows
      dims[5] = row_base + carry_s;
      dims[6] = row_base;
      dims[7] = 0;
    }
  }
}

// x[row_base + cu[b] + l] = emb[tok[b, l]] + pos[l] for l < n_b; rows [cu[B], rows_padded) of the text segment are zeroed
// (the tile padding the GEMMs run over: finite, never read back).  Grid: B*L caption slots, then the padding rows.
template <int NV>
__global__ __launch_bounds__(256) void embed_packed_kernel(const long long* __restrict__ tok, const float* __restrict__ emb,
                                                           const float* __restrict__ pos, float* __restrict__ x, int ldx,
                                                           const int* __restrict__ cu, int B, int L, int vocab, int rows_padded,
                                                           const int* __restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (rows_dev) rows_padded = min(rows_padded, *rows_dev);        // device-side padded row count (<= the host's bound)
  if (m >= B * L) {
    const int r = cu[B] + (m - B * L);
    if (r < rows_padded) {
#pragma unroll
      for (int i = 0; i < NV; ++i) *(float4*)(x + (size_t)r * ldx + i * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int b = m / L, l = m - b * L;
  const int base = cu[b];
  if (l >= cu[b + 1] - base) return;
  long long t = tok[m];
  t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
  const float* e = emb + (size_t)t * (NV * 256);
  const float* p = pos + (size_t)l * (NV * 256);
  float* xr = x + (size_t)(base + l) * ldx;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 a = *(const float4*)(e + i * 256 + lane * 4);
    const float4 c = *(const float4*)(p + i * 256 + lane * 4);
    *(float4*)(xr + i * 256 + lane * 4) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}

// x[b*L + 0] = cls + pos[0]   (reference :2421-2425, before ln_pre)
template <int NV>
__global__ __launch_bounds__(256) void cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                  float* __restrict__ x, int ldx, int B, int L) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (b >= B) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 a = *(const float4*)(cls + i * 256 + lane * 4);
    const float4 p = *(const float4*)(pos + i * 256 + lane * 4);
    *(float4*)(x + (size_t)b * L * ldx + i * 256 + lane * 4) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

// Lateral adapter combine: out = ln_adapt( [cls ; BN(dw3x3(grid))] + [cls ; t] )
//   row 0      : (1 + usecls) * x[b, 0]
//   row 1 + p  : dwb + sum_taps dww[tap] * x[b, 1 + nbr(p, tap)] + t[b*g*g + p]
template <int NV>
__global__ __launch_bounds__(256) void adapter_kernel(const float* __restrict__ xin, int ldx,
                                                      const float* __restrict__ t, int ldt,
                                                      const float* __restrict__ dww, const float* __restrict__ dwb,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ xout, int ldo, int B, int L, int g,
                                                      int usecls, float eps) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= B * L) return;
  const int b = m / L, l = m - b * L;
  constexpr int C = NV * 256;
  float4 v[NV];
  if (l == 0) {
    const float f = usecls ? 2.f : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 a = *(const float4*)(xin + (size_t)m * ldx + i * 256 + lane * 4);
      v[i] = make_float4(a.x * f, a.y * f, a.z * f, a.w * f);
    }
  } else {
    const int p = l - 1, gy = p / g, gx = p - gy * g;
    const float* tr = t + ((size_t)b * g * g + p) * ldt;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 a = *(const float4*)(tr + i * 256 + lane * 4);
      const float4 c = *(const float4*)(dwb + i * 256 + lane * 4);
      v[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = gy + ky - 1;
      if (yy < 0 || yy >= g) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = gx + kx - 1;
        if (xx < 0 || xx >= g) continue;
        const float* nb = xin + ((size_t)b * L + 1 + yy * g + xx) * ldx;
        const float* wt = dww + (ky * 3 + kx) * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float4 a = *(const float4*)(nb + i * 256 + lane * 4);
          const float4 w = *(const float4*)(wt + i * 256 + lane * 4);
          v[i].x += a.x * w.x; v[i].y += a.y * w.y; v[i].z += a.z * w.z; v[i].w += a.w * w.w;
        }
      }
    }
  }
  ln_core<NV>(v, gamma, beta, eps, lane);
  store_row<NV>(v, xout, (size_t)m, ldo, 1, lane);
}

// The same pass with a wave owning one ROW of the token grid (g tokens): the nine depthwise filter rows and the bias
// stay in registers across the row instead of being re-read for every token (they were half of the kernel's L1
// traffic: 9 x C weights against 9 x C neighbour values + C of t per token).  The wave of grid row 0 also does the
// class token.
template <int NV>
__global__ __launch_bounds__(256) void adapter_gridrow_kernel(const float* __restrict__ xin, int ldx,
                                                              const float* __restrict__ t, int ldt,
                                                              const float* __restrict__ dww, const float* __restrict__ dwb,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ xout, int ldo, int B, int L, int g,
                                                              int usecls, float eps, const float* __restrict__ gamma1,
                                                              const float* __restrict__ beta1, bf16_t* __restrict__ lno, int ldl,
                                                              float* __restrict__ center, float* __restrict__ rowstat) {
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and neighbouring grid rows -- which read each other's
  // tokens for the 3 x 3 filter -- sit in neighbouring workgroups: consecutive ROWS go to one XCD (bijective remap of the
  // workgroup id), so a token row is fetched from the fabric once instead of by up to three L2s
  const int nb = gridDim.x, q = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
  const int bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
  const int r = bid * WPB + (threadIdx.x >> 6);
  if (r >= B * g) return;
  // msclip_adapter_combine_ln_stats: the block's ln_1 of the row just written, from the registers it is still in (the same
  // fp32 values a LayerNorm pass would read back from xout), plus the LayerNorm fold's per-row state of msclip_layernorm_stats
  auto second_ln = [&](float4 (&v)[NV], size_t row) {
    // (-ffast-math would fold the first LayerNorm's scale / shift into this one's centring: pin the rounded fp32 values, the
    //  ones xout holds, so that the result is bitwise the two-kernel chain's)
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
    const float mean = ln_core<NV>(v, gamma1, beta1, eps, lane);
    store_row<NV>(v, lno, row, ldl, 0, lane);
    if (lane == 0) {
      center[row] = mean;
      *(float2*)(rowstat + 2 * row) = make_float2(1.f, 0.f);
    }
  };
  const int b = r / g, gy = r - b * g;
  constexpr int C = NV * 256;
  float4 w[9][NV], bias[NV];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int i = 0; i < NV; ++i) w[k][i] = *(const float4*)(dww + k * C + i * 256 + lane * 4);
#pragma unroll
  for (int i = 0; i < NV; ++i) bias[i] = *(const float4*)(dwb + i * 256 + lane * 4);
  float4 v[NV];
  if (gy == 0) {
    const float f = usecls ? 2.f : 1.f;
    const size_t m = (size_t)b * L;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 a = *(const float4*)(xin + m * ldx + i * 256 + lane * 4);
      v[i] = make_float4(a.x * f, a.y * f, a.z * f, a.w * f);
    }
    ln_core<NV>(v, gamma, beta, eps, lane);
    store_row<NV>(v, xout, m, ldo, 1, lane);
    if (gamma1) second_ln(v, m);
  }
  for (int gx = 0; gx < g; ++gx) {
    const int p = gy * g + gx;
    const float* tr = t + ((size_t)b * g * g + p) * ldt;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 a = *(const float4*)(tr + i * 256 + lane * 4);
      v[i] = make_float4(a.x + bias[i].x, a.y + bias[i].y, a.z + bias[i].z, a.w + bias[i].w);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = gy + ky - 1;
      if (yy < 0 || yy >= g) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = gx + kx - 1;
        if (xx < 0 || xx >= g) continue;
        const float* nb = xin + ((size_t)b * L + 1 + yy * g + xx) * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float4 a = *(const float4*)(nb + i * 256 + lane * 4);
          const float4 ww = w[ky * 3 + kx][i];
          v[i].x += a.x * ww.x; v[i].y += a.y * ww.y; v[i].z += a.z * ww.z; v[i].w += a.w * ww.w;
        }
      }
    }
    ln_core<NV>(v, gamma, beta, eps, lane);
    store_row<NV>(v, xout, (size_t)b * L + 1 + p, ldo, 1, lane);
    if (gamma1) second_ln(v, (size_t)b * L + 1 + p);
  }
}

// The same pass with a WORKGROUP per sample (round 5; MSCLIP_ADAPTER_SAMPLE=0: the grid-row kernel).  The grid-row kernel is
// latency-bound: a wave walks its 7 tokens one after the other -- 30 loads, two LayerNorms (four wave reductions), two stores per
// token, nothing overlapping -- with 108 VGPRs of filter rows keeping the occupancy at 2-3 waves per SIMD (86 us per launch at
// batch 512 for 272 MB = 3.2 TB/s).  Here the nine filter rows + the bias sit in LDS (30 KB per workgroup, read as conflict-free
// 16-byte pieces), the 8 waves of the workgroup take the sample's tokens round-robin, and all of a sample's neighbour re-reads
// hit the CU's own L1 / the XCD's L2.  Same arithmetic (under -ffast-math the compiler orders the nine-tap sums per kernel:
// last-bit differences against the grid-row kernel).
template <int NV>
__global__ __launch_bounds__(512) void adapter_sample_kernel(const float* __restrict__ xin, int ldx,
                                                             const float* __restrict__ t, int ldt,
                                                             const float* __restrict__ dww, const float* __restrict__ dwb,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ xout, int ldo, int B, int L, int g,
                                                             int usecls, float eps, const float* __restrict__ gamma1,
                                                             const float* __restrict__ beta1, bf16_t* __restrict__ lno, int ldl,
                                                             float* __restrict__ center, float* __restrict__ rowstat) {
  constexpr int C = NV * 256;
  __shared__ float4 wl[10][NV * 64];                 // taps 0..8, then the bias; [k][i * 64 + lane] = columns i*256 + lane*4 .. +3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < 10 * NV * 64; idx += 512) {
    const int k = idx / (NV * 64), j = idx - k * (NV * 64);
    wl[k][j] = *(const float4*)((k < 9 ? dww + (size_t)k * C : dwb) + j * 4);
  }
  __syncthreads();
  const int b = blockIdx.x;
  for (int l = wave; l < L; l += 8) {
    const size_t m = (size_t)b * L + l;
    float4 v[NV];
    if (l == 0) {
      const float f = usecls ? 2.f : 1.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 a = *(const float4*)(xin + m * ldx + i * 256 + lane * 4);
        v[i] = make_float4(a.x * f, a.y * f, a.z * f, a.w * f);
      }
    } else {
      const int p = l - 1, gy = p / g, gx = p - gy * g;
      const float* tr = t + ((size_t)b * g * g + p) * ldt;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 a = *(const float4*)(tr + i * 256 + lane * 4);
        const float4 c = wl[9][i * 64 + lane];
        v[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = gy + ky - 1;
        if (yy < 0 || yy >= g) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = gx + kx - 1;
          if (xx < 0 || xx >= g) continue;
          const float* nb = xin + ((size_t)b * L + 1 + yy * g + xx) * ldx;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const float4 a = *(const float4*)(nb + i * 256 + lane * 4);
            const float4 ww = wl[ky * 3 + kx][i * 64 + lane];
            v[i].x += a.x * ww.x; v[i].y += a.y * ww.y; v[i].z += a.z * ww.z; v[i].w += a.w * ww.w;
          }
        }
      }
    }
    ln_core<NV>(v, gamma, beta, eps, lane);
    store_row<NV>(v, xout, m, ldo, 1, lane);
    if (gamma1) {
      // the block's ln_1 from the registers the row is still in (see adapter_gridrow_kernel: the rounded fp32 values are pinned
      // so that -ffast-math cannot fold the two LayerNorms into each other)
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
      const float mean = ln_core<NV>(v, gamma1, beta1, eps, lane);
      store_row<NV>(v, lno, m, ldl, 0, lane);
      if (lane == 0) {
        center[m] = mean;
        *(float2*)(rowstat + 2 * m) = make_float2(1.f, 0.f);
      }
    }
  }
}

// y = x / ||x||_2 per row, fp32 math; writes fp32 and (optionally) a bf16 copy for the logits GEMM
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int ldx, float* __restrict__ of,
                                                     int ldf, bf16_t* __restrict__ ob, int ldb, int M, int E) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* r = x + (size_t)m * ldx;
  float q = 0.f;
  for (int c = lane * 4; c < E; c += 256) {
    const float4 a = *(const float4*)(r + c);
    q += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  const float inv = 1.f / sqrtf(wave_sum(q));
  for (int c = lane * 4; c < E; c += 256) {
    float4 a = *(const float4*)(r + c);
    a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
    if (of) *(float4*)(of + (size_t)m * ldf + c) = a;
    if (ob) {
      uint2 p;
      p.x = pack_bf16x2(a.x, a.y);
      p.y = pack_bf16x2(a.z, a.w);
      *(uint2*)(ob + (size_t)m * ldb + c) = p;
    }
  }
}

}  // namespace

// row width -> float4 pieces per lane (NV = C / 256): 768 (ViT-B), 1024 (the ViT-L-width stand-in of config C5), 512, 256
#define NV_LAUNCH(C, KERNEL, GRID, BLK, ST, ...)                                                  \
  if ((C) == 768) { hipLaunchKernelGGL(KERNEL<3>, GRID, BLK, 0, ST, __VA_ARGS__); }               \
  else if ((C) == 1024) { hipLaunchKernelGGL(KERNEL<4>, GRID, BLK, 0, ST, __VA_ARGS__); }         \
  else if ((C) == 512) { hipLaunchKernelGGL(KERNEL<2>, GRID, BLK, 0, ST, __VA_ARGS__); }          \
  else if ((C) == 256) { hipLaunchKernelGGL(KERNEL<1>, GRID, BLK, 0, ST, __VA_ARGS__); }          \
  else return MSCLIP_EINVAL;

static int launch_ln(const float* x, int ldx, const int* row_idx, int row_mul, int row_add, const float* gamma,
                     const float* beta, void* out, int ldo, int out_kind, float* raw_out, int ld_raw, int M, int C,
                     float eps, const float* gamma2, const float* beta2, int split, hipStream_t st) {
  const char* single = getenv("MSCLIP_LN_SINGLE_ROW");            // the wave-per-row kernel, for cross-checks only
  if (!row_idx && row_mul == 1 && row_add == 0 && !raw_out && M >= 4096 && !(single && single[0] == '1')) {
    const dim3 grid2(((M + 1) / 2 + WPB - 1) / WPB), blk2(256);
    NV_LAUNCH(C, ln_pair_kernel, grid2, blk2, st, x, ldx, gamma, beta, out, ldo, out_kind, M, eps, gamma2, beta2, split)
    return msclip_launch_status();
  }
  const dim3 grid((M + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, ln_kernel, grid, blk, st, x, ldx, row_idx, row_mul, row_add, gamma, beta, out, ldo, out_kind, raw_out, ld_raw, M, eps, gamma2, beta2, split)
  return msclip_launch_status();
}

extern "C" int msclip_layernorm(const float* x, int ldx, const int* row_idx, int row_mul, int row_add,
                                const float* gamma, const float* beta, void* out, int ldo, int out_kind,
                                float* raw_out, int ld_raw, int M, int C, float eps, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_layernorm, stream, x, ldx, row_idx, row_mul, row_add, gamma, beta, out, ldo, out_kind, raw_out, ld_raw, M, C, eps);
  if (!x || !gamma || !beta || !out || M <= 0 || (ldx % 4) || (ldo % 4)) return MSCLIP_EINVAL;
  return launch_ln(x, ldx, row_idx, row_mul, row_add, gamma, beta, out, ldo, out_kind, raw_out, ld_raw, M, C, eps, gamma,
                   beta, M, (hipStream_t)stream);
}

extern "C" int msclip_layernorm_stats(const float* x, int ldx, const float* gamma, const float* beta, void* out, int ldo,
                                      int out_kind, float* raw_out, int ld_raw, float* center, float* rowstat, int M, int C,
                                      float eps, const int* m_dev, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_layernorm_stats, stream, x, ldx, gamma, beta, out, ldo, out_kind, raw_out, ld_raw, center, rowstat, M, C, eps, m_dev);
  if (!x || !gamma || !beta || !out || M <= 0 || (ldx % 4) || (ldo % 4) || (raw_out && (ld_raw % 4))) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((M + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, ln_stats_kernel, grid, blk, st, x, ldx, gamma, beta, out, ldo, out_kind, raw_out, ld_raw, center, rowstat, M, eps, m_dev)
  return msclip_launch_status();
}

extern "C" int msclip_rowstat_finalize(const float* part, int groups, float* center, float* rowstat, int M, int C, float eps,
                                       const int* m_dev, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_rowstat_finalize, stream, part, groups, center, rowstat, M, C, eps, m_dev);
  if (!part || !center || !rowstat || M <= 0 || groups <= 0 || C != groups * 64) return MSCLIP_EINVAL;
  const dim3 grid((M + 255) / 256), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (groups == 12) hipLaunchKernelGGL(rowstat_finalize_kernel<6>, grid, blk, 0, st, part, groups, center, rowstat, M, 1.f / (float)C, eps, m_dev);
  else if (groups == 16) hipLaunchKernelGGL(rowstat_finalize_kernel<8>, grid, blk, 0, st, part, groups, center, rowstat, M, 1.f / (float)C, eps, m_dev);
  else hipLaunchKernelGGL(rowstat_finalize_kernel<0>, grid, blk, 0, st, part, groups, center, rowstat, M, 1.f / (float)C, eps, m_dev);
  return msclip_launch_status();
}

extern "C" int msclip_layernorm_split(const float* x, int ldx, const float* gamma, const float* beta,
                                      const float* gamma2, const float* beta2, int split, void* out, int ldo,
                                      int out_kind, int M, int C, float eps, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_layernorm_split, stream, x, ldx, gamma, beta, gamma2, beta2, split, out, ldo, out_kind, M, C, eps);
  if (!x || !gamma || !beta || !gamma2 || !beta2 || !out || M <= 0 || split < 0 || split > M || (ldx % 4) || (ldo % 4))
    return MSCLIP_EINVAL;
  return launch_ln(x, ldx, nullptr, 1, 0, gamma, beta, out, ldo, out_kind, nullptr, 0, M, C, eps, gamma2, beta2, split,
                   (hipStream_t)stream);
}

extern "C" int msclip_embed_tokens(const long long* tokens, const float* emb, const float* pos, float* x, int ldx,
                                   int* eot_row, int B, int L, int C, int vocab, int row_base, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_embed_tokens, stream, tokens, emb, pos, x, ldx, eot_row, B, L, C, vocab, row_base);
  if (!tokens || !emb || !pos || !x || B <= 0 || L <= 0 || (ldx % 4)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * L;
  const dim3 grid((rows + WPB - 1) / WPB), blk(256);
  float* xb = x + (size_t)row_base * ldx;
  NV_LAUNCH(C, embed_kernel, grid, blk, st, tokens, emb, pos, xb, ldx, rows, L, vocab)
  if (eot_row) hipLaunchKernelGGL(eot_kernel, dim3(B), dim3(64), 0, st, tokens, eot_row, B, L, row_base);
  return msclip_launch_status();
}

extern "C" int msclip_text_lengths(const long long* tokens, int B, int L, int row_base, int* len, int* cu, int* eot_row,
                                   int* dims, int pad_to, int cap_rows, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_text_lengths, stream, tokens, B, L, row_base, len, cu, eot_row, dims, pad_to, cap_rows);
  if (!tokens || !len || !cu || B <= 0 || L <= 0 || row_base < 0 || pad_to < 0 || (dims && cap_rows < B * L)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(len_kernel, dim3(B), dim3(64), 0, st, tokens, len, B, L);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)len, cu, eot_row, B, row_base, dims, pad_to, cap_rows);
  return msclip_launch_status();
}

extern "C" int msclip_embed_tokens_packed(const long long* tokens, const float* emb, const float* pos, float* x, int ldx,
                                          const int* cu, int B, int L, int C, int vocab, int row_base, int rows_padded,
                                          const int* rows_dev, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_embed_tokens_packed, stream, tokens, emb, pos, x, ldx, cu, B, L, C, vocab, row_base, rows_padded, rows_dev);
  if (!tokens || !emb || !pos || !x || !cu || B <= 0 || L <= 0 || (ldx % 4) || row_base < 0 || rows_padded < 0 ||
      rows_padded > B * L + 255)
    return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int slots = B * L + 256;                     // every caption slot + at most 255 padding rows (+1: rounding)
  const dim3 grid((slots + WPB - 1) / WPB), blk(256);
  float* xb = x + (size_t)row_base * ldx;
  NV_LAUNCH(C, embed_packed_kernel, grid, blk, st, tokens, emb, pos, xb, ldx, cu, B, L, vocab, rows_padded, rows_dev)
  return msclip_launch_status();
}

extern "C" int msclip_fill_cls(const float* cls, const float* pos, float* x, int ldx, int B, int L, int C,
                               void* stream) {
  MSCLIP_PLAN_HOOK(msclip_fill_cls, stream, cls, pos, x, ldx, B, L, C);
  if (!cls || !pos || !x || B <= 0 || (ldx % 4)) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, cls_kernel, grid, blk, st, cls, pos, x, ldx, B, L)
  return msclip_launch_status();
}

// workgroup-per-sample form of the adapter pass (default): 16-byte accesses, C <= 1024 (40 KB of filter rows in LDS)
// (measured, tools/probes/adapter_bench.py: 7 x 7 grids 84 -> 73 us at batch 512, 167 -> 152 us at batch 1024; 14 x 14 grids --
//  197 tokens per workgroup, one workgroup per CU at batch 256 -- 165 -> 183 us: they keep the grid-row kernel)
static bool adapter_sample_form(int C, int L, int ldx, int ldt, int ldo) {
  const char* e = getenv("MSCLIP_ADAPTER_SAMPLE");
  if (e && e[0] == '0') return false;
  if (!(C <= 1024 && !((ldx | ldt | ldo) & 3))) return false;
  return L <= 64 || (e && e[0] == '1');              // "1" forces it (tests cover the large grids with it)
}

extern "C" int msclip_adapter_combine_ln(const float* xin, int ldx, const float* t, int ldt, const float* dww,
                                         const float* dwb, const float* gamma, const float* beta, float* xout,
                                         int ldo, int B, int L, int g, int C, int usecls, float eps, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_adapter_combine_ln, stream, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L, g, C, usecls, eps);
  if (!xin || !t || !dww || !dwb || !gamma || !beta || !xout || xin == xout || L != g * g + 1) return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const char* perrow = getenv("MSCLIP_ADAPTER_PER_TOKEN");       // the wave-per-token kernel, for cross-checks only
  if (!(perrow && perrow[0] == '1') && adapter_sample_form(C, L, ldx, ldt, ldo)) {
    NV_LAUNCH(C, adapter_sample_kernel, dim3(B), dim3(512), st, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L, g, usecls, eps,
              (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, 0, (float*)nullptr, (float*)nullptr)
    return msclip_launch_status();
  }
  if (!(perrow && perrow[0] == '1')) {
    const dim3 grid((B * g + WPB - 1) / WPB), blk(256);
    NV_LAUNCH(C, adapter_gridrow_kernel, grid, blk, st, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L, g, usecls, eps,
              (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, 0, (float*)nullptr, (float*)nullptr)
    return msclip_launch_status();
  }
  const int rows = B * L;
  const dim3 grid((rows + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, adapter_kernel, grid, blk, st, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L, g, usecls, eps)
  return msclip_launch_status();
}

extern "C" int msclip_adapter_combine_ln_stats(const float* xin, int ldx, const float* t, int ldt, const float* dww,
                                               const float* dwb, const float* gamma, const float* beta, float* xout, int ldo,
                                               const float* gamma1, const float* beta1, void* lno, int ldl, float* center,
                                               float* rowstat, int B, int L, int g, int C, int usecls, float eps, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_adapter_combine_ln_stats, stream, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, gamma1, beta1, lno, ldl, center, rowstat, B, L, g, C, usecls, eps);
  if (!xin || !t || !dww || !dwb || !gamma || !beta || !xout || xin == xout || L != g * g + 1 || !gamma1 || !beta1 || !lno ||
      !center || !rowstat || (ldl & 3))
    return MSCLIP_EINVAL;
  if (adapter_sample_form(C, L, ldx, ldt, ldo)) {
    NV_LAUNCH(C, adapter_sample_kernel, dim3(B), dim3(512), (hipStream_t)stream, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L,
              g, usecls, eps, gamma1, beta1, (bf16_t*)lno, ldl, center, rowstat)
    return msclip_launch_status();
  }
  const dim3 grid((B * g + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, adapter_gridrow_kernel, grid, blk, (hipStream_t)stream, xin, ldx, t, ldt, dww, dwb, gamma, beta, xout, ldo, B, L, g,
            usecls, eps, gamma1, beta1, (bf16_t*)lno, ldl, center, rowstat)
  return msclip_launch_status();
}

extern "C" int msclip_l2norm(const float* x, int ldx, float* out_f32, int ldf, void* out_bf16, int ldb, int M, int E,
                             void* stream) {
  MSCLIP_PLAN_HOOK(msclip_l2norm, stream, x, ldx, out_f32, ldf, out_bf16, ldb, M, E);
  if (!x || (!out_f32 && !out_bf16) || M <= 0 || (E % 4) || (ldx % 4)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(l2norm_kernel, dim3((M + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, x, ldx, out_f32, ldf,
                     (bf16_t*)out_bf16, ldb, M, E);
  return msclip_launch_status();
}

namespace {
// out[m] = x[src(m)] as raw bytes (16-byte pieces): the last block's live rows (cls / EOT) moved to a compact matrix.
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ x, long long ldx_bytes, const int* __restrict__ row_idx,
                                                          int row_mul, int row_add, char* __restrict__ out, long long ldo_bytes,
                                                          int M, int pieces) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= M) return;
  const size_t src = row_idx ? (size_t)row_idx[m] : (size_t)m * row_mul + row_add;
  const uint4* s = (const uint4*)(x + src * ldx_bytes);
  uint4* d = (uint4*)(out + (size_t)m * ldo_bytes);
  for (int i = lane; i < pieces; i += 64) d[i] = s[i];
}
}  // namespace

extern "C" int msclip_gather_rows(const void* x, long long ldx_bytes, const int* row_idx, int row_mul, int row_add, void* out,
                                  long long ldo_bytes, int M, int row_bytes, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_gather_rows, stream, x, ldx_bytes, row_idx, row_mul, row_add, out, ldo_bytes, M, row_bytes);
  if (!x || !out || M <= 0 || row_bytes <= 0 || (row_bytes % 16) || (ldx_bytes % 16) || (ldo_bytes % 16)) return MSCLIP_EINVAL;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((M + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx_bytes,
                     row_idx, row_mul, row_add, (char*)out, ldo_bytes, M, row_bytes / 16);
  return msclip_launch_status();
}

namespace {
// LayerNorm -> OCP e4m3 with one scale per row: q[m][c] = fp8(y[m][c] / s[m]), s[m] = max_c |y[m][c]| / 448 (the format's
// largest finite value), y = LN(x[m]) with (gamma, beta) for m < split and (gamma2, beta2) from there on.  The producer
// half of the fp8 projections (msclip_gemm_f8 multiplies s[m] back in its epilogue).  One wave per row.
template <int NV>
__global__ __launch_bounds__(256) void ln_f8_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ gamma2,
                                                    const float* __restrict__ beta2, int split, unsigned char* __restrict__ q,
                                                    int ldq, float* __restrict__ row_scale, int M, float eps,
                                                    const int* __restrict__ m_dev) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m_dev) M = min(M, *m_dev);
  if (m >= M) return;
  if (m >= split) {
    gamma = gamma2;
    beta = beta2;
  }
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(x + (size_t)m * ldx + i * 256 + lane * 4);
  ln_core<NV>(v, gamma, beta, eps, lane);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / s;
  if (lane == 0) row_scale[m] = s;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, p, true);
    *(int*)(q + (size_t)m * ldq + i * 256 + lane * 4) = p;
  }
}

// bf16 rows -> e4m3 + per-row scale (same convention); C % 8 == 0, one wave per row, 16-byte loads.
__global__ __launch_bounds__(256) void quant_f8_rows_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ q,
                                                            int ldq, float* __restrict__ row_scale, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (m >= M) return;
  const bf16_t* xr = x + (size_t)m * ldx;
  float amax = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack_bf16x8(*(const uint4*)(xr + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / s;
  if (lane == 0) row_scale[m] = s;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack_bf16x8(*(const uint4*)(xr + c), f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    *(int2*)(q + (size_t)m * ldq + c) = make_int2(lo, hi);
  }
}
}  // namespace

extern "C" int msclip_layernorm_f8(const float* x, int ldx, const float* gamma, const float* beta, const float* gamma2,
                                   const float* beta2, int split, void* q, int ldq, float* row_scale, int M, int C, float eps,
                                   const int* m_dev, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_layernorm_f8, stream, x, ldx, gamma, beta, gamma2, beta2, split, q, ldq, row_scale, M, C, eps, m_dev);
  if (!x || !gamma || !beta || !gamma2 || !beta2 || !q || !row_scale || M <= 0 || split < 0 || split > M || (ldx % 4) || (ldq % 4))
    return MSCLIP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((M + WPB - 1) / WPB), blk(256);
  NV_LAUNCH(C, ln_f8_kernel, grid, blk, st, x, ldx, gamma, beta, gamma2, beta2, split, (unsigned char*)q, ldq, row_scale, M, eps, m_dev)
  return msclip_launch_status();
}

extern "C" int msclip_quant_f8_rows(const void* x, int ldx, void* q, int ldq, float* row_scale, int M, int C, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_quant_f8_rows, stream, x, ldx, q, ldq, row_scale, M, C);
  if (!x || !q || !row_scale || M <= 0 || C <= 0 || (C % 8) || (ldx % 8) || (ldq % 8)) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(quant_f8_rows_kernel, dim3((M + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (unsigned char*)q, ldq, row_scale, M, C);
  return msclip_launch_status();
}
