// ddp_internal.h - shared helpers of libddp_hip.so (not part of the C ABI).
#ifndef DDP_INTERNAL_H
#define DDP_INTERNAL_H
#include <hip/hip_runtime.h>

int ddp_fail(int code, const char* msg);
int ddp_fail_hip(hipError_t err, const char* where);

#endif
