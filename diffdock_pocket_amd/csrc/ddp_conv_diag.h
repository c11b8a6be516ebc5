// ddp_conv_diag.h - diagnostics of the conv kernels (ddp_conv.hip), compiled in only by the diagnostic builds of
// diffdock_pocket_amd/build.py (-DDDP_STAMPS: in-kernel phase stamps read by tools/stamp_conv.py; -DDDP_ABLATE=n: timing-only
// ablations for tools/ablate_conv.py whose results are wrong by construction).  In the product build every macro below is a
// no-op / the identity.  Included by ddp_conv.hip after its vector typedefs (f32x4).
#ifndef DDP_CONV_DIAG_H
#define DDP_CONV_DIAG_H

#ifdef DDP_STAMPS
// Diagnostic build only (python -m diffdock_pocket_amd.build --stamps): thread 0 of every workgroup records
// s_memtime at the phase boundaries into a device buffer that no kernel code reads (tools/stamp_conv.py).
#define DDP_STAMP_SLOTS 40
#define DDP_STAMP_WGS 32768
__device__ unsigned long long ddp_stamp_buf[DDP_STAMP_WGS * DDP_STAMP_SLOTS];
__device__ __forceinline__ unsigned long long ddp_stamp_now(bool realtime) {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  if (realtime)
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  else
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(k)                                                                                          \
  do {                                                                                                    \
    const unsigned long long t_ = ddp_stamp_now((k) >= 22);                                               \
    if (threadIdx.x == 0 && blockIdx.x < DDP_STAMP_WGS) ddp_stamp_buf[blockIdx.x * DDP_STAMP_SLOTS + (k)] = t_; \
  } while (0)
extern "C" int ddp_debug_read_stamps(unsigned long long* host_dst, int n_wgs) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ddp_stamp_buf),
                                  sizeof(unsigned long long) * DDP_STAMP_SLOTS * (size_t)n_wgs);
}
#define STAMP_SYNC() __syncthreads()
// running stamps of wave 0 inside g_stage (slot 0 pass only): slots 24..31 (s_memtime)
#define GSTAMP()                                                                                          \
  do {                                                                                                    \
    const unsigned long long t_ = ddp_stamp_now(false);                                                   \
    if (slot == 0 && threadIdx.x == 0 && blockIdx.x < DDP_STAMP_WGS && gstamp_i < 8)                      \
      ddp_stamp_buf[blockIdx.x * DDP_STAMP_SLOTS + 24 + gstamp_i] = t_;                                   \
    ++gstamp_i;                                                                                           \
  } while (0)
#else
#define GSTAMP() do {} while (0)
#define STAMP(k) do {} while (0)
#define STAMP_SYNC() do {} while (0)
#endif

// Timing-only ablations for tools/ablate_conv.py (never defined in the product build): DDP_ABLATE=1 drops the weight
// loads of the scalar-block main loop, =2 drops its LDS A-operand reads; results are then wrong by construction.
#if defined(DDP_ABLATE) && (DDP_ABLATE == 1 || DDP_ABLATE == 7)   // 7 = 1 + 3: neither weight nor G loads
#define DDP_ABL_B(x) (f32x4{1e-9f, 2e-9f, 3e-9f, 4e-9f} * (float)(lane + 1))
#else
#define DDP_ABL_B(x) (x)
#endif
#if defined(DDP_ABLATE) && (DDP_ABLATE == 3 || DDP_ABLATE == 7)   // G pass without its global loads
#define DDP_ABL_G(x, q) (f32x4{1e-12f, 2e-12f, 3e-12f, 4e-12f} * (float)((q) + lane))
#elif defined(DDP_G_NT)   // experiment: the G rows as non-temporal (streaming) loads - they should not push the weights out of L2
#define DDP_ABL_G(x, q) __builtin_nontemporal_load(&(x))
#else
#define DDP_ABL_G(x, q) (x)
#endif
#if defined(DDP_ABLATE) && DDP_ABLATE == 4   // G pass with 1/9 of its FMAs and LDS reads
#define DDP_ABL_NQ(n) 1
#else
#define DDP_ABL_NQ(n) (n)
#endif
#if defined(DDP_ABLATE) && DDP_ABLATE == 5   // G pass without its per-unit epilogue (no read-modify-write of the message tile)
#define DDP_ABL_EPI (S.hid < 0)
#else
#define DDP_ABL_EPI true
#endif
#if defined(DDP_ABLATE) && DDP_ABLATE == 6   // G pass without the K-sliced extra columns
#define DDP_ABL_KSLICE false
#else
#define DDP_ABL_KSLICE true
#endif
#if defined(DDP_ABLATE) && DDP_ABLATE == 2
#define DDP_ABL_A(x, old) ((old) * 1.0001f)
#else
#define DDP_ABL_A(x, old) (x)
#endif

#endif /* DDP_CONV_DIAG_H */
