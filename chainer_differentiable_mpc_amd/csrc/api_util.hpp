// api_util.hpp - small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <cstddef>
#include <cstdint>

namespace dmpc {

// Vector loads (float2/float4) assume 16-byte aligned array bases; NULL is "absent", hence fine.
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static constexpr size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace dmpc
