// ddp_capi.hip - error bookkeeping and ABI version of libddp_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "ddp_hip.h"
#include "ddp_internal.h"

static thread_local char g_err[256] = "ok";

int ddp_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int ddp_fail_hip(hipError_t err, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(err));
  return (int)err;
}

int ddp_shape_rows_min_lds = 0;
int ddp_shape_stage_a_pad = 0;
// Launches enqueued after this call: ddp_conv_rows with at least `rows_min_lds_bytes` of dynamic LDS (> 80 KiB: ONE 4-wave workgroup per
// CU, i.e. one 256-register wave per SIMD, instead of two), the MFMA forms of ddp_stage_a* with `stage_a_lds_pad_bytes` on top of their
// static LDS (> 25 KiB: one workgroup per CU).  0 / 0 = the kernels' own occupancy.  Results do not depend on it.
extern "C" int ddp_set_occupancy_shaping(int rows_min_lds_bytes, int stage_a_lds_pad_bytes) {
  if (rows_min_lds_bytes < 0 || rows_min_lds_bytes > 160 * 1024 || stage_a_lds_pad_bytes < 0 || stage_a_lds_pad_bytes > 100 * 1024)
    return ddp_fail(DDP_EINVAL, "ddp_set_occupancy_shaping: bytes out of range");
  ddp_shape_rows_min_lds = rows_min_lds_bytes;
  ddp_shape_stage_a_pad = stage_a_lds_pad_bytes;
  return 0;
}

extern "C" int ddp_abi_version(void) { return DDP_ABI_VERSION; }
extern "C" const char* ddp_last_error(void) { return g_err; }

// 16 hex digits of the SHA-256 over the kernel sources this library was compiled from (diffdock_pocket_amd/build.py passes
// -DDDP_SRC_SHA16); _lib.load() compares it with the sources in the tree, so a stale binary cannot be loaded silently.
#ifndef DDP_SRC_SHA16
#define DDP_SRC_SHA16 "unknown"
#endif
// (the tag in front lets build.built_hash() find the hash in the file bytes without loading the library)
static const char ddp_src_tag[] = "DDP_SRC_SHA16=" DDP_SRC_SHA16;
extern "C" const char* ddp_source_hash(void) { return ddp_src_tag + 14; }
