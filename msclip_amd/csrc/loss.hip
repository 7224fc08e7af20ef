// Contrastive head reductions: row log-sum-exp over a logits block and the
// symmetric cross-entropy partial sum.  The loss itself is not in the reference
// (SURVEY.md s8 a14); labels follow gather_tensors' rank-major order
// (reference lib/utils/comm.py:150-153): label of local row i = label_off + i.
#include "common.h"
#include "../../include/msclip_hip.h"

namespace {

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

extern "C" int msclip_lse_rows(const float* logits, int ld, float* lse, int R, int N, void* stream) {
  if (!logits || !lse || R <= 0 || N <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(lse_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, lse, N);
  return msclip_launch_status();
}

extern "C" int msclip_clip_loss_partial(const float* lse_img, const float* lse_txt, const float* img_rows, int ld,
                                        int label_off, int R, float scale, float* out, void* stream) {
  if (!lse_img || !lse_txt || !img_rows || !out || R <= 0) return MSCLIP_EINVAL;
  hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lse_img, lse_txt, img_rows, ld,
                     label_off, R, scale, out);
  return msclip_launch_status();
}
