// mpc_fwd_asm_kernel.hpp - MPCstep.forward_rec (mpc/mpc_step.py:175-286) for a LinDx and a QuadCost as ONE generated
// gfx950 instruction stream (MpcFwdAsm<nx, nu> of mpc_fwd_asm_gen.hpp, emitted by gen_mpc_fwd_asm.py).  This file is
// the C++ side: LDS layout (one ring of input slots per wavefront), the per-lane operands, and the per-trajectory
// outputs the stream leaves in registers.  Same results as mpc_forward_rec_kernel up to the order of a few sums.
// Needs: B % 4 == 0, 16-byte aligned arrays (checked by the launcher).
#pragma once
#include "lqr_asm_kernel.hpp"       // dmpc_zero_chunks, lds_byte_address
#include "mpc_fwd_asm_gen.hpp"
#include "mpc_kernels.hpp"

namespace dmpc {

template <int NX, int NU>
constexpr size_t mpc_fwd_asm_lds_bytes() {
  return (size_t)4 * MpcFwdAsm<NX, NU>::RING_BYTES;
}

template <int NX, int NU>
__device__ __forceinline__ void mpc_forward_asm_body(const MpcFwdArgs &a, const int block) {
  using G = MpcFwdAsm<NX, NU>;
  static_assert(G::kAvailable, "no generated instruction stream for this shape");
  constexpr int NS = NX + NU;
  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  const int b0 = __builtin_amdgcn_readfirstlane((block * 4 + wave) * 4);
  if (b0 >= a.B) return;      // whole wavefront (B % 4 == 0); the stream has no workgroup barrier
  const int b = b0 + r;

  extern __shared__ float fwd_asm_lds[];
  const unsigned ring = lds_byte_address(fwd_asm_lds) + (unsigned)wave * G::RING_BYTES;

  MpcFwdAsmIn in{};
  in.ring = __builtin_amdgcn_readfirstlane(ring);
  in.T = __builtin_amdgcn_readfirstlane(T);
  in.cap = a.ls_cap;
  in.decay = a.ls_decay;
  in.want_objs = a.objs != nullptr ? 1 : 0;
  in.uf_mask = a.u_first != nullptr ? ~0u : 0u;
  // ---- DMA: chunk g = q*64 + lane64 of the slot [C | c | F | f | K | k | u | lower | upper | x | zeros]
  const bool has_f = a.f != nullptr && T > 1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= G::KD) continue;
    const int g = q * 64 + lane64;
    const char *base = reinterpret_cast<const char *>(dmpc_zero_chunks);   // padding and absent arrays: zeros
    size_t per = 0;
    int g0 = g;
    bool isF = false;
    if (g < G::CH_c) { base = (const char *)a.C; per = (size_t)NS * NS * 4; g0 = G::CH_C; }
    else if (g < G::CH_F) { base = (const char *)a.c; per = (size_t)NS * 4; g0 = G::CH_c; }
    else if (g < G::CH_f) { if (T > 1) { base = (const char *)a.F; per = (size_t)NX * NS * 4; g0 = G::CH_F; isF = true; } }
    else if (g < G::CH_K) { if (has_f) { base = (const char *)a.f; per = (size_t)NX * 4; g0 = G::CH_f; isF = true; } }
    else if (g < G::CH_k) { base = (const char *)a.Ks; per = (size_t)NU * NX * 4; g0 = G::CH_K; }
    else if (g < G::CH_u) { base = (const char *)a.ks; per = (size_t)NU * 4; g0 = G::CH_k; }
    else if (g < G::CH_lo) { base = (const char *)a.controls; per = (size_t)NU * 4; g0 = G::CH_u; }
    else if (g < G::CH_hi) { base = (const char *)a.lower; per = (size_t)NU * 4; g0 = G::CH_lo; }
    else if (g < G::CH_x) { base = (const char *)a.upper; per = (size_t)NU * 4; g0 = G::CH_hi; }
    else if (g < G::CH_END) { base = (const char *)a.states; per = (size_t)NX * 4; g0 = G::CH_x; }
    {  // the trajectories (of the wave's four) whose data this chunk holds: a later line-search pass fetches it only while
       // one of them still searches.  Padding / absent arrays and lane 0 (the instruction is never empty): always.
      unsigned rows = 0xfu;
      if (per != 0 && lane64 != 0) {
        const unsigned r0 = (unsigned)((size_t)(g - g0) * 16 / per), r1 = (unsigned)(((size_t)(g - g0) * 16 + 15) / per);
        rows = 0;
        for (unsigned rr = (r0 < 4 ? r0 : 3); rr <= (r1 < 4 ? r1 : 3); ++rr) rows |= 1u << rr;   // (a chunk of 4-byte rows spans all four)
      }
      in.dmrow |= rows << (4 * q);
    }
    in.ptr0[q] = reinterpret_cast<uint64_t>(base) + (size_t)b0 * per + (size_t)(g - g0) * 16 - (uint64_t)q * 1024u;
    in.str[q] = (uint64_t)(B * per);
    in.strl[q] = isF ? 0 : in.str[q];   // there is no F_{T-1}: the step t = T-1 fetches slice T-2 again (never consumed)
  }
  // ---- LDS read addresses (ring slot 0).  Lanes past the controls read the slot's zero padding and stay zero.
  const bool is_x = lane < NX, is_u = lane >= NX && lane < NS;
  const unsigned zero = ring + (unsigned)G::ZERO;
  const int m = lane - NX;
  in.a_row = is_x ? ring + (unsigned)((G::CH_F * 4 + (r * NX + lane) * NS) * 4)
                  : (is_u ? ring + (unsigned)((G::CH_K * 4 + (r * NU + m) * NX) * 4) : zero);
  in.a_row2 = is_x ? in.a_row + (unsigned)(NX * 4) : zero;   // control lanes: their rows have no u columns
  in.a_aff = is_x ? ring + (unsigned)((G::CH_f * 4 + r * NX + lane) * 4)
                  : (is_u ? ring + (unsigned)((G::CH_k * 4 + r * NU + m) * 4) : zero);
  in.a_crow = lane < NS ? ring + (unsigned)((G::CH_C * 4 + (r * NS + lane) * NS) * 4) : zero;
  in.a_caff = lane < NS ? ring + (unsigned)((G::CH_c * 4 + r * NS + lane) * 4) : zero;
  in.a_hat = is_x ? ring + (unsigned)((G::CH_x * 4 + r * NX + lane) * 4)
                  : (is_u ? ring + (unsigned)((G::CH_u * 4 + r * NU + m) * 4) : zero);
  in.a_lb = is_u ? ring + (unsigned)((G::CH_lo * 4 + r * NU + m) * 4) : zero;
  // ---- stores
  in.pst0 = is_x ? reinterpret_cast<uint64_t>(a.x + (size_t)b * NX + lane)
                 : (is_u ? reinterpret_cast<uint64_t>(a.u + (size_t)b * NU + m) : 0);
  in.dst = is_x ? (uint64_t)(B * NX * 4) : (uint64_t)(B * NU * 4);
  in.puf0 = (is_u && a.u_first != nullptr) ? reinterpret_cast<uint64_t>(a.u_first + (size_t)b * NU + m) : 0;
  in.pobj0 = a.objs != nullptr ? reinterpret_cast<uint64_t>(a.objs + b) : 0;
  in.dobj = (uint64_t)(B * 4);

  float cost, oldc, alpha;
  int nls, worse;
  G::run(in, cost, oldc, alpha, nls, worse);

  if (lane == 0) {
    int info_bits = 0;
    if (worse) {                           // cap hit: the reference would still be looping; :274
      alpha /= a.ls_decay;
      info_bits |= 8;
    }
    if (!is_finite(cost)) info_bits |= 2;
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = oldc;
    a.alphas[b] = alpha;
    a.n_ls[b] = nls;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

template <int NX, int NU>
__global__ __launch_bounds__(256) void mpc_forward_asm_kernel(const MpcFwdArgs a) {
  mpc_forward_asm_body<NX, NU>(a, blockIdx.x);
}

}  // namespace dmpc
