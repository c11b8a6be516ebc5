// ddp_internal.h - shared helpers of libddp_hip.so (not part of the C ABI).
#ifndef DDP_INTERNAL_H
#define DDP_INTERNAL_H
#include <hip/hip_runtime.h>

#include "ddp_hip.h"

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

// csrc/ddp_conv_rows16.hip: the launch of ddp_conv_rows for tasks with rows_form = 1 (arguments validated by ddp_conv_rows; sc = size class)
int ddp_conv_rows16_launch(const ddp_conv_shape_t* shape, const ddp_conv_task_t* tasks, int ntasks, int sc, void* stream);

// Occupancy shaping (ddp_set_occupancy_shaping, include/ddp_hip.h): launch-time LDS floors that decide how many workgroups of a kernel
// share a CU.  Plain ints read when a launch is enqueued.
extern int ddp_shape_rows_min_lds;      // ddp_conv_rows: dynamic LDS of a launch is at least this many bytes
extern int ddp_shape_stage_a_pad;       // ddp_stage_a*: dynamic LDS added to the kernels' static LDS

#endif
