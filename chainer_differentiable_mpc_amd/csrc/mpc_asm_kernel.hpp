// mpc_asm_kernel.hpp - MPCstep.backward_rec (mpc/mpc_step.py:70-173) as ONE generated gfx950 instruction stream
// (MpcAsm<nx, nu> of lqr_asm_gen.hpp, emitted by gen_lqr_asm.py with mpc=True): the Riccati sweep of the fused LQR
// stream with the box QP of every timestep solved inside it (projected Newton, mpc/pnqp.py:37-201, per-trajectory
// termination).  This file is the C++ side, as lqr_asm_kernel.hpp is for the LQR solve: LDS layout (one ring per
// wavefront, nothing else - the gains go straight to HBM) and the per-lane operands.
//
// Against the HIP kernels (mpc_kernels.hpp / mpc_dma_kernels.hpp) the arithmetic is the same up to the last bit of a
// reciprocal (the gain solve uses v_rcp_f32 as the LQR stream does; the QP's own quotients carry a Newton step as
// pnqp_device.hpp's do), and nothing is left to the compiler between the first DMA and the last store.
// Needs: B % 4 == 0, per-trajectory termination, 16-byte aligned C, c, F, f (and 12 nu + 4 nx <= 64 for the
// re-centring variant: its x_t share the slot padding's one dword DMA with u_t and the bounds).
#pragma once
#include "box_ddp_kernels.hpp"
#include "lqr_asm_kernel.hpp"
#include "mpc_kernels.hpp"

namespace dmpc {

template <int NX, int NU>
constexpr size_t mpc_asm_lds_bytes() {
  return (size_t)4 * MpcAsm<NX, NU, false>::RING_BYTES;
}

// EXPAND: a.states given - c is the ORIGINAL linear term and the sweep re-centres it (need_expand, mpc_step.py:305-317)
// One wavefront: the sweep of the four trajectories b0 .. b0 + 3 with its ring in slot `slot` of the workgroup's LDS
template <int NX, int NU, bool HAS_F, bool EXPAND>
__device__ __forceinline__ void mpc_backward_asm_wave(const MpcBackArgs &a, const int b0, const int slot) {
  using G = MpcAsm<NX, NU, EXPAND>;
  static_assert(G::kAvailable, "no generated instruction stream for this shape");
  constexpr int NS = NX + NU;
  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  if (b0 >= a.B) return;      // whole wavefront (B % 4 == 0); the stream has no workgroup barrier
  const int b = b0 + r;

  extern __shared__ float lds[];
  const unsigned ring = lds_byte_address(lds) + (unsigned)slot * G::RING_BYTES;

  LqrAsmIn<NX, NU> in{};      // the forward sweep's operands stay zero: the stream stops after the backward sweep
  in.ring = __builtin_amdgcn_readfirstlane(ring);
  in.T = T;
  in.bwd_only = 1;
  in.n_qp_iter = a.n_qp_iter;
  lqr_asm_backward_sources<NX, NU, G, HAS_F>(in, a.C, a.c, a.F, a.f, T, B, b0, lane64);
  {  // the slot padding takes [u | lower | upper (| x)] of the wave's four trajectories: dword l comes from lane l
    constexpr int n_u = 12 * NU, n_all = n_u + (EXPAND ? 4 * NX : 0);
    const int l = lane64 < n_all ? lane64 : 0;
    const bool is_x = EXPAND && l >= n_u;
    const int arr = l / (4 * NU), j = is_x ? l - n_u : l % (4 * NU);
    const float *base = is_x ? a.states : (arr == 0 ? a.controls : (arr == 1 ? a.lower : a.upper));
    const size_t per = is_x ? NX : NU;   // floats per trajectory and timestep
    in.pm = reinterpret_cast<uint64_t>(base + ((size_t)(T - 1) * B + (size_t)b0) * per + j) - (uint64_t)G::PADM;
    in.dm = (uint64_t)0 - (uint64_t)(B * per * 4);
    in.am = ring + (unsigned)(r * NU * 4);
    if constexpr (EXPAND) {  // lane j < nx: x_t[j], lane nx + m: u_t[m] (the stream broadcasts from lanes < ns only)
      in.atau = ring + (unsigned)((lane < NX ? n_u + r * NX + lane : (lane < NS ? r * NU + (lane - NX) : 0)) * 4);
      in.act = ring + (unsigned)(G::OFF_C + (r * NS + (lane < NS ? lane : NS - 1)) * NS * 4);   // row `lane` of C_t
    }
  }
  G::issue_first(in);  // the first DMA groups leave now; the rest of the set-up overlaps their flight
  lqr_asm_row_addresses<NX, NU, G>(in, ring, r, lane);
  const bool col_aff = lane == NS;
  in.eaff = col_aff ? 1.f : 0.f;
  {
    const size_t tb = (size_t)(T - 1) * B + (size_t)b;
#pragma unroll
    for (int m = 0; m < NU; ++m)
      in.pk[m] = col_aff ? reinterpret_cast<uint64_t>(a.ks + tb * NU + m)
                         : reinterpret_cast<uint64_t>(a.Ks + (tb * NU + m) * NX + (lane < NX ? lane : 0));
    in.dk = (uint64_t)0 - (uint64_t)(col_aff ? B * NU * 4 : B * NU * NX * 4);
  }
  in.pxi = reinterpret_cast<uint64_t>(a.C);  // the stream loads x_init unconditionally: any readable word

  float xvout, minpiv;
  int nqp, qpinfo;
  G::run(in, xvout, minpiv, nqp, qpinfo);

  if (lane == 0) {
    a.n_qp_total[b] = nqp;
    if (a.info != nullptr) {
      if (a.info_store) a.info[b] = qpinfo;
      else if (qpinfo != 0) atomicOr(&a.info[b], qpinfo);
    }
  }
}

template <int NX, int NU, bool HAS_F, bool EXPAND>
__device__ __forceinline__ void mpc_backward_asm_body(const MpcBackArgs &a, const int block) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  mpc_backward_asm_wave<NX, NU, HAS_F, EXPAND>(a, __builtin_amdgcn_readfirstlane((block * 4 + wave) * 4), wave);
}

template <int NX, int NU, bool HAS_F, bool EXPAND>
__global__ __launch_bounds__(256) void mpc_backward_asm_kernel(const MpcBackArgs a) {
  mpc_backward_asm_body<NX, NU, HAS_F, EXPAND>(a, blockIdx.x);
}

// the sweep with the previous box-DDP iteration's bookkeeping in the launch's last n_sel workgroups (see
// mpc_backward_rec_dma_select_kernel of mpc_dma_kernels.hpp)
template <int NX, int NU, bool HAS_F, bool EXPAND>
__global__ __launch_bounds__(256) void mpc_backward_asm_select_kernel(const MpcBackArgs a, const DdpSelectArgs s,
                                                                      const int n_sel, unsigned *sel_sync) {
  const int n_back = (int)gridDim.x - n_sel;
  if ((int)blockIdx.x >= n_back) {
    box_ddp_select_body<256, NX, NU>(s, (int)blockIdx.x - n_back, n_sel, sel_sync);
    return;
  }
  mpc_backward_asm_body<NX, NU, HAS_F, EXPAND>(a, blockIdx.x);
}

// ONE launch per box-DDP iteration of the built-in pendulum (configs 2 and 4; round 5): a workgroup owns four trajectories -
// its first wavefront runs their backward sweep (the generated stream with the QP inside), a workgroup barrier, then each of
// its four wavefronts runs one trajectory's speculative line search (mpc_forward_rec_pendulum_spec4_wave); the previous
// iteration's batch-mixing bookkeeping (box_ddp.py:200-230) rides in the launch's last n_sel workgroups as before.  Gains,
// flags and the QP counts pass from sweep to search through global memory inside the workgroup (workgroup-scope ordering: the
// wavefront drains its stores and LDS-DMA before the barrier); the LDS ring of the sweep and the searches' slots share the
// dynamic LDS one after the other.  Saves the launch boundary between sweep and search: ten of a solve's 22 launches.
// pa.T > 0 (the solve's first iteration): the nominal trajectory, its linearisation and the re-centred cost first - the lanes
// 0..15 of the first wavefront, four per trajectory (pendulum_rollout_linearize4_lane) - and the chain's words cleared; that
// launch must not be given the stop flag (it clears it) nor accumulate into `info` (it clears that too).
__global__ __launch_bounds__(256) void box_ddp_pendulum_iter_kernel(const MpcBackArgs ba, const MpcFwdArgs fa, const DdpSelectArgs s,
                                                                    const int n_sel, unsigned *sel_sync, const PendulumArgs pa) {
  const int n_main = (int)gridDim.x - n_sel;
  if ((int)blockIdx.x >= n_main) {
    box_ddp_select_body<256, 3, 1>(s, (int)blockIdx.x - n_main, n_sel, sel_sync);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b0 = (int)blockIdx.x * 4;
  if (pa.T > 0) {
    pa.clear.run((int)(blockIdx.x * blockDim.x + threadIdx.x));
    if (threadIdx.x < 16) pendulum_rollout_linearize4_lane(pa, b0 + ((int)threadIdx.x >> 2), (int)threadIdx.x & 3);
  }
  if (wave == 0) {
    if (pa.T > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the sweep's LDS-DMA reads what these lanes just stored
    mpc_backward_asm_wave<3, 1, false, false>(ba, b0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  mpc_forward_rec_pendulum_spec4_wave(fa, b0 + wave, wave);
}

}  // namespace dmpc
