// api_util.hpp - small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace dmpc {

// Vector loads (float2/float4) assume 16-byte aligned array bases; NULL is "absent", hence fine.
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static constexpr size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// What this library launched last (a hint, not a synchronisation: plain reads and writes).  The headline solve exists
// in two forms, a three-step loop and a sweep unrolled over the horizon (75 KB of straight-line code, ~3 % faster when
// its code is still on chip, 4-5 us slower when other kernels ran in between - scripts/unroll_ab.sh): the unrolled
// form is taken when the previous launch was the same solve.
enum { kLaunchOther = 0, kLaunchPlainSolve = 1 };
inline int g_last_launch = kLaunchOther;
// lqr_api.hip: DiffLqr.backward's second solve on [grad_x; grad_u] as two arrays (kkt_api.hip)
int lqr_second_solve(int T, int B, int nx, int nu, const float *C, const float *cx, const float *cu, const float *F,
                     const float *Ks, const float *Quu, const float *Qxu, float *x_out, float *u_out, int32_t *info,
                     hipStream_t stream);

static inline void note_other_launch() { g_last_launch = kLaunchOther; }

}  // namespace dmpc
