// ddp_internal.h - shared helpers of libddp_hip.so (not part of the C ABI).
#ifndef DDP_INTERNAL_H
#define DDP_INTERNAL_H
#include <hip/hip_runtime.h>

int ddp_fail(int code, const char* msg);
int ddp_fail_hip(hipError_t err, const char* where);

// Dynamic-LDS limit of a kernel, raised only when a launch needs more than any launch before it: hipFuncSetAttribute is not
// permitted while a stream is being captured (hipErrorStreamCaptureUnsupported), so a captured step must find the limit already
// set by the ordinary steps that ran before the capture (sampler.Sampler captures after two such steps).
static inline hipError_t ddp_need_lds(const void* kernel, int bytes, int* have) {
  if (bytes <= *have) return hipSuccess;
  const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (err == hipSuccess) *have = bytes;
  return err;
}

#endif
