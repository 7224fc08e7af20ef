// Shared device helpers for the gfx950 kernels (wave64, MFMA, bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

#define MSCLIP_OK 0
#define MSCLIP_EINVAL (-1)
#define MSCLIP_ELAUNCH (-2)

// ---- per-DEVICE host-side caches (a process may drive more than one GPU: function attributes and CU counts belong to the device
// that is current at the call).  Devices beyond the table fall back to slot 0 semantics that re-query every time.
constexpr int MSCLIP_MAXDEV = 16;
static inline int msclip_dev_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MSCLIP_MAXDEV) return -1;
  return dev;
}
static inline int msclip_device_cus() {
  static int ncus[MSCLIP_MAXDEV] = {};
  const int di = msclip_dev_index();
  if (di >= 0 && ncus[di]) return ncus[di];
  int dev = 0, n = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (di >= 0) ncus[di] = n;
  return n;
}
// hipFuncAttributeMaxDynamicSharedMemorySize of `kernel`, set once per device (static table per expansion site = per instantiation)
#define MSCLIP_LDS_ATTR(kernel, bytes, ok)                                                                         \
  do {                                                                                                             \
    static bool done_[MSCLIP_MAXDEV] = {};                                                                         \
    const int di_ = msclip_dev_index();                                                                            \
    if (di_ < 0 || !done_[di_]) {                                                                                  \
      (ok) = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)(bytes)) == hipSuccess;                                                      \
      if (di_ >= 0) done_[di_] = (ok);                                                                             \
    }                                                                                                              \
  } while (0)

static inline int msclip_launch_status() {
  return hipGetLastError() == hipSuccess ? MSCLIP_OK : MSCLIP_ELAUNCH;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32)
  __bf16 h = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return *reinterpret_cast<uint32_t*>(&v);
}
__device__ __forceinline__ void unpack_bf16x8(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Buffer-addressed LDS-DMA piece (buffer_load_dwordx4 ... lds): SGPR descriptor + 32-bit per-lane byte offset +
// scalar byte offset.  Measured on MI355X (tools/probes/overlap_probe): beside a saturated MFMA stream this form
// costs the MFMA waves nothing (1738 TF vs 1650 alone) while global_load_lds with 64-bit lane addresses drops them
// to 1295 TF -- the address VALU work and operands compete for the SIMD's issue slots.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (AS3 void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// One 1-KiB LDS-DMA piece: every lane supplies its own 16-byte global source, the
// destination is the wave-uniform LDS base + lane*16 (cdna_hip_programming.md s5).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const AS1 void*)gsrc, (AS3 void*)lds_wave_base, 16, 0, 0);
}
