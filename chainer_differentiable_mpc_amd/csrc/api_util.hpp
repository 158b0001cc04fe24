// api_util.hpp - small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include <mutex>
#include <unordered_map>

namespace dmpc {

// Vector loads (float2/float4) assume 16-byte aligned array bases; NULL is "absent", hence fine.
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static constexpr size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, set once per kernel (and again only for a larger request), not per
// launch: the call costs a few microseconds of host time, as much as the launch it precedes
inline void set_max_lds(const void *kernel, int bytes) {
  static std::mutex m;
  static std::unordered_map<const void *, int> done;
  std::lock_guard<std::mutex> g(m);
  auto it = done.find(kernel);
  if (it != done.end() && it->second >= bytes) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done[kernel] = bytes;
}

// The kernel this thread launched last, for dmpc_last_kernel_name() (diagnostics and benchmark labelling: the name a
// profiler lists is asked of the runtime, not kept by hand).  Thread-local: entry points stay thread-compatible.
inline thread_local const void *t_last_kernel = nullptr;
static inline void note_kernel(const void *host_function) { t_last_kernel = host_function; }
#define DMPC_LAUNCH_GGL(kernel, ...)                                 \
  do {                                                               \
    ::dmpc::note_kernel(reinterpret_cast<const void *>(kernel));     \
    hipLaunchKernelGGL(kernel, __VA_ARGS__);                         \
  } while (0)

// lqr_api.hip: DiffLqr.backward in one launch from the saving solve's gains and value functions (reads neither C nor c)
int lqr_adjoint(int T, int B, int nx, int nu, const float *F, const float *grad_x, const float *grad_u, const float *Ks,
                const float *Quu, const float *Qxu, const float *Vv, const float *x, const float *u, int strict_math,
                float *d_x_init, float *dC, float *dc, float *dF, float *df, int32_t *info, hipStream_t stream);
// lqr_api.hip: DiffLqr.backward's second solve on [grad_x; grad_u] as two arrays (kkt_api.hip)
int lqr_second_solve(int T, int B, int nx, int nu, const float *C, const float *cx, const float *cu, const float *F,
                     const float *Ks, const float *Quu, const float *Qxu, float *x_out, float *u_out, int32_t *info,
                     hipStream_t stream);

}  // namespace dmpc
