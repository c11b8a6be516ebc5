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
