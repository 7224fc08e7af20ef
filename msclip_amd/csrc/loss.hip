// Contrastive head.  The symmetric cross-entropy is not in the reference; labels follow
// gather_tensors' rank-major order (reference lib/utils/comm.py:150-153): label of local row i = label_off + i.
//
// msclip_clip_lse_fused: scaled-cosine logits block  s * A_loc @ B_all^T  (A_loc [R, E], B_all [N, E], bf16 unit
// rows) reduced on the fly to per-row (max, sum-exp) partials -- the [R, N] logits are never written.  One wave owns
// 32 rows (its A fragments stay in registers for the whole sweep) and one of `nsplit` column ranges; the operands are
// swapped in the MFMA (D[col][row]) so each lane owns ONE row and 16 of every 32 columns: the online max/sum is a
// register-local reduction plus one exchange with lane^32.  B rows are read straight from L2 (N*E*2 bytes, a few MB)
// in fragment shape.  The same launch also emits the label logit (the diagonal) from the split that contains it.
// msclip_clip_loss_from_partials merges the splits of both directions into the scalar partial loss of this rank.
#include "common.h"
#include "plan.h"
#include "../../include/msclip_hip.h"

namespace {

// feature width handled in registers (E % 16 == 0): EMAX = 512 (the released configs' EMBED_DIM) or 768 (the ViT-L-width
// stand-in of config C5: 192 fragment registers)
template <int EMAX>
__global__ __launch_bounds__(256) void lse_fused_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ Bm, int ldb, int R, int N, int E,
                                                        float scale, int label_off, int nsplit,
                                                        float* __restrict__ part_max, float* __restrict__ part_sum,
                                                        float* __restrict__ diag) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rb = blockIdx.x;                         // 32-row block
  const int split = blockIdx.y * 4 + wave;           // column range
  if (split >= nsplit) return;
  const int fr = lane & 31, fhi = lane >> 5;
  const int row = rb * 32 + fr;
  const int rowc = min(row, R - 1);
  const int nks = E >> 4;                            // k-steps of 16

  bf16x8 af[EMAX / 16];                              // this lane's row of A: k-chunks (2*ks + fhi)
#pragma unroll
  for (int ks = 0; ks < EMAX / 16; ++ks)
    if (ks < nks) af[ks] = *(const bf16x8*)(A + (size_t)rowc * lda + (ks * 2 + fhi) * 8);

  const int ntile = (N + 31) >> 5;
  const int per = (ntile + nsplit - 1) / nsplit;
  const int t0 = split * per, t1 = min(ntile, t0 + per);
  const int label = label_off + row;                 // column holding this row's positive pair

  float mx = -INFINITY, sum = 0.f, dval = 0.f;
  bool have_d = false;
  for (int t = t0; t < t1; ++t) {
    const int col = t * 32 + fr;                     // B row fetched by this lane (as the MFMA A operand)
    const int colc = min(col, N - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < EMAX / 16; ++ks)
      if (ks < nks) {
        const bf16x8 bf = *(const bf16x8*)(Bm + (size_t)colc * ldb + (ks * 2 + fhi) * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af[ks], acc, 0, 0, 0);
      }
    // lane owns logits[row][t*32 + (r&3) + 8*(r>>2) + 4*fhi]
    float tmx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
      const float v = c < N ? acc[r] * scale : -INFINITY;
      acc[r] = v;
      tmx = fmaxf(tmx, v);
      if (c == label) { dval = v; have_d = true; }
    }
    const float nm = fmaxf(mx, tmx);
    if (nm > -INFINITY) {
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ts += __expf(acc[r] - nm);
      sum = sum * __expf(mx - nm) + ts;
      mx = nm;
    }
  }
  // merge the two half-waves (they own disjoint columns of the same rows)
  const float omx = __shfl_xor(mx, 32, 64), osum = __shfl_xor(sum, 32, 64);
  const float od = __shfl_xor(dval, 32, 64);
  const int ohd = __shfl_xor((int)have_d, 32, 64);
  const float nm = fmaxf(mx, omx);
  float tot = 0.f;
  if (nm > -INFINITY) tot = sum * __expf(mx - nm) + osum * __expf(omx - nm);
  if (row < R && fhi == 0) {
    part_max[(size_t)row * nsplit + split] = nm;
    part_sum[(size_t)row * nsplit + split] = tot;
    if (have_d) diag[row] = dval;
    else if (ohd) diag[row] = od;
  }
}

__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

__device__ __forceinline__ float merged_lse(const float* pm, const float* ps, int nsplit) {
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pm[s]);
  float t = 0.f;
  for (int s = 0; s < nsplit; ++s) t += ps[s] * __expf(pm[s] - m);
  return m + __logf(t);
}

// out[0] = scale * sum_i (lse_img[i] - d_i) + (lse_txt[i] - d_i)
__global__ __launch_bounds__(256) void loss_from_partials_kernel(const float* __restrict__ pm_i,
                                                                 const float* __restrict__ ps_i,
                                                                 const float* __restrict__ pm_t,
                                                                 const float* __restrict__ ps_t,
                                                                 const float* __restrict__ diag, int R, int nsplit,
                                                                 float scale, float* __restrict__ out,
                                                                 float* __restrict__ lse_out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) {
    const float li = merged_lse(pm_i + (size_t)i * nsplit, ps_i + (size_t)i * nsplit, nsplit);
    const float lt = merged_lse(pm_t + (size_t)i * nsplit, ps_t + (size_t)i * nsplit, nsplit);
    if (lse_out) { lse_out[i] = li; lse_out[R + i] = lt; }
    s += (li - diag[i]) + (lt - diag[i]);
  }
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// ---- unfused helpers kept for tests / forward() users that already hold a logits block
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ lg, int ld, float* __restrict__ lse,
                                                       int N) {
  __shared__ float sh[4];
  const float* r = lg + (size_t)blockIdx.x * ld;
  const bool vec = !((N | ld) & 3);
  float mx = -INFINITY;
  if (vec) {
    for (int c = threadIdx.x * 4; c < N; c += 1024) {
      const float4 a = *(const float4*)(r + c);
      mx = fmaxf(fmaxf(mx, fmaxf(a.x, a.y)), fmaxf(a.z, a.w));
    }
  } else {
    for (int c = threadIdx.x; c < N; c += 256) mx = fmaxf(mx, r[c]);
  }
  mx = block_reduce(mx, true, sh);
  float s = 0.f;
  if (vec) {
    for (int c = threadIdx.x * 4; c < N; c += 1024) {
      const float4 a = *(const float4*)(r + c);
      s += (__expf(a.x - mx) + __expf(a.y - mx)) + (__expf(a.z - mx) + __expf(a.w - mx));
    }
  } else {
    for (int c = threadIdx.x; c < N; c += 256) s += __expf(r[c] - mx);
  }
  s = block_reduce(s, false, sh);
  if (threadIdx.x == 0) lse[blockIdx.x] = mx + __logf(s);
}

__global__ __launch_bounds__(256) void clip_loss_kernel(const float* __restrict__ lse_i, const float* __restrict__ lse_t,
                                                        const float* __restrict__ rows, int ld, int label_off, int R,
                                                        float scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) {
    const float d = rows[(size_t)i * ld + label_off + i];
    s += (lse_i[i] - d) + (lse_t[i] - d);
  }
  s = block_reduce(s, false, sh);
  if (threadIdx.x == 0) out[0] = s * scale;
}

}  // namespace

extern "C" int msclip_clip_lse_fused(const void* A, int lda, const void* Bm, int ldb, int R, int N, int E, float scale,
                                     int label_off, int nsplit, float* part_max, float* part_sum, float* diag,
                                     void* stream) {
  MSCLIP_PLAN_HOOK(msclip_clip_lse_fused, stream, A, lda, Bm, ldb, R, N, E, scale, label_off, nsplit, part_max, part_sum, diag);
  if (!A || !Bm || !part_max || !part_sum || !diag || R <= 0 || N <= 0 || nsplit <= 0) return MSCLIP_EINVAL;
  if (E <= 0 || E > 768 || (E % 16) || (lda % 8) || (ldb % 8)) return MSCLIP_EINVAL;
  if (label_off < 0 || label_off + R > N) return MSCLIP_EINVAL;
  const dim3 grid((R + 31) / 32, (nsplit + 3) / 4);
  if (E <= 512)
    hipLaunchKernelGGL(lse_fused_kernel<512>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A, lda,
                       (const bf16_t*)Bm, ldb, R, N, E, scale, label_off, nsplit, part_max, part_sum, diag);
  else
    hipLaunchKernelGGL(lse_fused_kernel<768>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A, lda,
                       (const bf16_t*)Bm, ldb, R, N, E, scale, label_off, nsplit, part_max, part_sum, diag);
  return msclip_launch_status();
}

extern "C" int msclip_clip_loss_from_partials(const float* pmax_img, const float* psum_img, const float* pmax_txt,
                                              const float* psum_txt, const float* diag, int R, int nsplit, float scale,
                                              float* out, float* lse_out, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_clip_loss_from_partials, stream, pmax_img, psum_img, pmax_txt, psum_txt, diag, R, nsplit, scale, out, lse_out);
  if (!pmax_img || !psum_img || !pmax_txt || !psum_txt || !diag || !out || R <= 0 || nsplit <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(loss_from_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pmax_img, psum_img,
                     pmax_txt, psum_txt, diag, R, nsplit, scale, out, lse_out);
  return msclip_launch_status();
}

extern "C" int msclip_lse_rows(const float* logits, int ld, float* lse, int R, int N, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_lse_rows, stream, logits, ld, lse, R, N);
  if (!logits || !lse || R <= 0 || N <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(lse_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, lse, N);
  return msclip_launch_status();
}

extern "C" int msclip_clip_loss_partial(const float* lse_img, const float* lse_txt, const float* img_rows, int ld,
                                        int label_off, int R, float scale, float* out, void* stream) {
  MSCLIP_PLAN_HOOK(msclip_clip_loss_partial, stream, lse_img, lse_txt, img_rows, ld, label_off, R, scale, out);
  if (!lse_img || !lse_txt || !img_rows || !out || R <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lse_img, lse_txt, img_rows, ld,
                     label_off, R, scale, out);
  return msclip_launch_status();
}
