// Library introspection entry points (no GPU work) and HIP stream plumbing for the training step.
#include <hip/hip_runtime.h>

#include "../../include/msclip_hip.h"

extern "C" int msclip_abi_version(void) { return MSCLIP_ABI_VERSION; }
extern "C" const char* msclip_build_arch(void) { return "gfx950"; }

extern "C" int msclip_stream_priority_range(int* least, int* greatest) {
  return (int)hipDeviceGetStreamPriorityRange(least, greatest);
}

extern "C" int msclip_stream_create(int priority, void** stream) {
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
  *stream = (void*)s;
  return (int)e;
}

extern "C" int msclip_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }
