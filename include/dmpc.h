/*
 * dmpc.h - C-ABI of the MI355X-native differentiable-MPC inner solver (libdmpc_hip.so).
 *
 * Drop-in boundary for the hot path of pfnet-research/chainer-differentiable-mpc
 * (the reference is pure Python; these entry points are what a ctypes binding placed in
 * the reference's own modules would call - see INTEGRATION.md).
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to a C-contiguous, time-major float32 array
 *     (shapes as in the reference: C [T,B,ns,ns], c [T,B,ns], F [T-1 or T,B,nx,ns]
 *     - only F[t], t < T-1 is read -, f [T-1,B,nx] or NULL, x_init [B,nx],
 *     x [T,B,nx], u [T,B,nu], Ks [T,B,nu,nx], ks [T,B,nu]); ns = nx + nu;
 *   - the caller owns every buffer; the library allocates nothing, inputs are read-only;
 *   - work is enqueued on `stream` (a hipStream_t, may be NULL) and NOT synchronised;
 *   - return value: 0 ok, >0 a hipError_t, <0 an argument error (DMPC_E_*);
 *   - `info` (optional, int32 [B]) receives a per-trajectory bit mask (DMPC_INFO_*).
 *   - entry points are thread-compatible (no global mutable state).
 */
#ifndef DMPC_H_
#define DMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a signature, the meaning of an argument or a workspace size changes; the Python binding refuses a library
 * whose number differs from the one its ctypes signatures were written for (_lib.py: ABI_VERSION).
 * 400 (round 4): covers the round-3 changes that were made under 202 (dmpc_lqr_solve_saving + Vv_out, dmpc_lqr_kkt_grad_saved
 * + Vv, dmpc_pendulum_rollout_linearize + clamp_grad_closed, dmpc_mpc_step_backward + 5 arguments, dmpc_box_ddp dyn_params[6],
 * larger dmpc_lqr_workspace_bytes) and this round's additions. */
#define DMPC_VERSION 411

#define DMPC_E_BADARG (-1)      /* NULL / non-positive size */
#define DMPC_E_UNSUPPORTED (-2) /* dimensions outside what the kernels cover */
#define DMPC_E_WORKSPACE (-3)   /* workspace too small */

#define DMPC_INFO_SINGULAR 1   /* exact zero pivot met in a Quu / H factorisation */
#define DMPC_INFO_NONFINITE 2  /* NaN/Inf in the outputs of this trajectory */
#define DMPC_INFO_QP_ITERCAP 4 /* projected-Newton QP hit n_iter (reference: warnings.warn, pnqp.py:192) */
#define DMPC_INFO_LS_ITERCAP 8 /* MPC-step line search hit the safety cap (reference loop is unbounded, mpc_step.py:196) */

typedef void *dmpc_stream_t; /* hipStream_t */

int dmpc_version(void);

/* Hash of the source set the library was built from (csrc/build.py); lets a loader refuse a stale binary. */
const char *dmpc_source_hash(void);

/* Demangled name - as a profiler (rocprofv3 --kernel-trace) lists it - of the kernel the calling thread's last dmpc_*
 * call launched last, asked of the HIP runtime (hipKernelNameRefByPtr), not kept by hand.  Writes a NUL-terminated string
 * into buf (truncated to buf_bytes) and returns its length; 0 = nothing launched yet.  Diagnostics / benchmark labels. */
int dmpc_last_kernel_name(char *buf, size_t buf_bytes);

/* Which kernel family a shape dispatches to: 1 = DPP row kernel (16 lanes / trajectory),
 * 2 = wave kernel (64 lanes / trajectory), 3 = generic LDS kernel (runtime dimensions),
 * 4 = a container: the shape has no specialisation of its own and runs padded by the loads inside a larger kernel of
 *     family 1 (nu <= 4, nx + nu <= 15) or 2 (up to 32 states, 8 controls),  <0 unsupported. */
int dmpc_lqr_kernel_family(int nx, int nu);

/* Which kernel a plain (unmasked) dmpc_lqr_solve of this size runs (diagnostics, benchmark labelling):
 *   0 lqr_generic_kernel (runtime dims, LDS)      1 lqr_kernel (HIP, register prefetch)
 *   2 lqr_dma_kernel (HIP, LDS-DMA ring)          3 lqr_asm_kernel, ring (generated stream, F fetched twice)
 *   4 lqr_asm_kernel, stash (generated stream, F kept in accumulation registers)
 *   5 lqr_wave_mfma_backward (one wavefront per trajectory, e.g. (32,8): MFMA backward sweep, then the same wavefront
 *     rolls its trajectory out)
 *   6 lqr_asm_kernel, ring, gain rows through the workspace (horizons whose gain rows do not fit in LDS: T > 74 at
 *     (8,2); needs `ws`)
 *   7 a container: lqr_kernel<..., PAD> of a larger shape, or lqr_wave_mfma_backward<..., PAD> + the forward-only
 *     container kernel (needs `ws` unless the caller takes the gains)
 *   8 lqr_tiled_kernel (any size: a workgroup per trajectory, matrices in `ws`)
 *   9 lqr_wide_kernel (17 to 32 augmented columns, at most 16 states - (16,4), (16,8), (12,4), (12,8), and padded inside
 *     them every shape with nx <= 16, nu <= 8, nx + nu >= 16 when B % 4 == 0: four trajectories per wavefront, two
 *     registers per matrix row, outer products on the matrix cores; plain and masked (LQR_active) solves alike; needs `ws` -
 *     without it, with B < 4, T < 2 or B (nx+nu)^2 floats beyond 2^31 bytes the solve takes path 7 / 5)   <0 unsupported
 *   (round 5: the plain sweep of path 5 at (32,8), (24,8), (32,4), (24,4) is lqr_tile16_kernel - 16x16x4 tiles) */
int dmpc_lqr_solve_path(int T, int B, int nx, int nu);

/* ---- A. LqrRecursion (lqr/lqr_recursion.py:69-209) and LQR_active
 *         (mpc/active_constrained_lqr.py:67-202 when `u_zero_mask` != NULL) ------------ */

/* Bytes of device workspace needed by dmpc_lqr_solve (gains that do not fit in LDS). */
size_t dmpc_lqr_workspace_bytes(int T, int B, int nx, int nu);

/* solve_recursion(): backward Riccati sweep + forward rollout in ONE launch.
 *   f            NULL = no affine dynamics term (lqr_recursion.py:90-92)
 *   u_zero_mask  NULL, or uint8 [T,B,nu]: clamped controls (LQR_active semantics)
 *   Ks_out/ks_out NULL, or gains in forward time order (lqr_recursion.py:156-158)  */
int dmpc_lqr_solve(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                   const float *f, const float *x_init, const uint8_t *u_zero_mask, float *Ks_out,
                   float *ks_out, float *x_out, float *u_out, void *ws, size_t ws_bytes, int32_t *info,
                   dmpc_stream_t stream);

/* solve_recursion() in its training form: as dmpc_lqr_solve with gains, and the control blocks of every step's
 * Q-function left in HBM as well - Quu_out [T,B,nu,nu] and Qxu_out [T,B,nx,nu] (lqr_recursion.py:112-120) - together
 * with every step's value function Vv_out [T,B,nx,nx+1] (row i = V_t[i][:], v_t[i]; lqr_recursion.py:151-152).  The
 * gradient's second Riccati solve (DiffLqr.backward, lqr/differentiable_lqr.py:83-106) shares C and F with the
 * forward solve, so it can reuse K_t, Quu_t, Qxu_t and only redo the affine terms (dmpc_lqr_saved_solve below), and
 * the co-states are the value function's gradients, lambda_t = V_t x_t + v_t (dmpc_lqr_kkt_grad_saved).
 * Served by the generated instruction stream that keeps F on chip only (dmpc_lqr_saving_available: dmpc_lqr_solve_path
 * == 4, B % 4 == 0, room in LDS for the staging area of the saved blocks; all pointers 16-byte aligned);
 * DMPC_E_UNSUPPORTED otherwise - the caller then uses dmpc_lqr_solve and the full second solve.
 * `info` [B], if given, is WRITTEN (0 = clean) rather than or-ed into: the caller need not clear it first. */
int dmpc_lqr_saving_available(int T, int B, int nx, int nu);   /* 1: dmpc_lqr_solve_saving (and the saved-gains gradient) serves this size */
int dmpc_lqr_solve_saving(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                          const float *f, const float *x_init, float *Ks_out, float *ks_out, float *Quu_out,
                          float *Qxu_out, float *Vv_out, float *x_out, float *u_out, int32_t *info,
                          dmpc_stream_t stream);

/* The re-solve: the LQR problem of an earlier dmpc_lqr_solve_saving (same C, F) with another affine cost term c
 * [T,B,ns], f = 0 and another x_init.  K_t does not depend on c; k_t = -Quu_t^-1 (c_u + F_u^T v_{t+1}),
 * v_t = q_x + Qxu_t k_t (lqr_recursion.py:92,119-120,152), then the rollout (:160-200).  C is not read.
 * DMPC_E_UNSUPPORTED unless dmpc_lqr_solve_path == 4 (generated stream, F kept on chip) and B % 4 == 0. */
int dmpc_lqr_saved_solve(int T, int B, int nx, int nu, const float *c, const float *F, const float *Ks,
                         const float *Quu, const float *Qxu, const float *x_init, float *x_out, float *u_out,
                         int32_t *info, dmpc_stream_t stream);

/* backward(): gains only (lqr_recursion.py:69-158).  The `_ws` form takes the workspace of dmpc_lqr_workspace_bytes: the
 * shapes of kernel family 5 (nx + nu + 1 > 64: a workgroup per trajectory, matrices in the workspace) need it, every other
 * shape ignores it; the plain form is the `_ws` form with ws = NULL (DMPC_E_WORKSPACE for family 5). */
int dmpc_lqr_backward_sweep(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                            const float *f, const uint8_t *u_zero_mask, float *Ks_out, float *ks_out,
                            int32_t *info, dmpc_stream_t stream);
int dmpc_lqr_backward_sweep_ws(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                               const float *f, const uint8_t *u_zero_mask, float *Ks_out, float *ks_out, void *ws,
                               size_t ws_bytes, int32_t *info, dmpc_stream_t stream);

/* forward(Ks, ks): closed-loop rollout (lqr_recursion.py:160-200). */
int dmpc_lqr_forward_sweep(int T, int B, int nx, int nu, const float *Ks, const float *ks, const float *F,
                           const float *f, const float *x_init, const uint8_t *u_zero_mask, float *x_out,
                           float *u_out, int32_t *info, dmpc_stream_t stream);

/* ---- B. DiffLqr.backward (lqr/differentiable_lqr.py:78-142): analytic KKT gradient ----
 *   inputs: retained (C, c, F), solution (x, u), upstream (grad_x [T,B,nx], grad_u [T,B,nu]).
 *   outputs: d_x_init [B,nx], dC [T,B,ns,ns], dc [T,B,ns], dF [T-1,B,nx,ns], df [T-1,B,nx]
 *            (any of dC/dF/df may be NULL to skip it).
 *   strict_math = 0 reproduces the reference bit-faithfully in structure: dC = 0.5*dtau(x)tau +
 *   tau(x)dtau (:128) and df = d_lambda[0:T-1] (:133); 1 gives the symmetric dC and d_lambda[1:T]. */
size_t dmpc_lqr_kkt_workspace_bytes(int T, int B, int nx, int nu);
int dmpc_lqr_kkt_grad(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                      const float *x, const float *u, const float *grad_x, const float *grad_u,
                      int strict_math, float *d_x_init, float *dC, float *dc, float *dF, float *df,
                      void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream);

/* The same gradient when the forward solve was dmpc_lqr_solve_saving.
 * With Vv (its value functions) the whole gradient is ONE launch that reads neither C nor c: the reference's solve is exact
 * block elimination of its KKT system, so its co-state recursions (:85-104, :115-126) equal lambda_t = V_t x_t + v_t and
 * d_lambda_t = V_t dx_t + v'_t, v' the affine value term of the second solve (:108-114) - the affine re-solve from Ks, Quu,
 * Qxu on [grad_x; grad_u] forms both while it rolls d_tau out and writes dC, dc, dF, df, d_x_init itself (all five
 * required then; ws is not touched).  With Vv == NULL, or where that stream does not serve the size: the second solve
 * reuses the gains (dmpc_lqr_saved_solve) and the co-state sweep reads C once.  DMPC_E_UNSUPPORTED where
 * dmpc_lqr_saved_solve is (nothing has been launched then: call dmpc_lqr_kkt_grad instead). */
int dmpc_lqr_kkt_grad_saved(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                            const float *x, const float *u, const float *Ks, const float *Quu, const float *Qxu,
                            const float *Vv, const float *grad_x, const float *grad_u, int strict_math,
                            float *d_x_init, float *dC, float *dc, float *dF, float *df, void *ws, size_t ws_bytes,
                            int32_t *info, dmpc_stream_t stream);

/* ---- A / B in float64 (optional `_f64` variants of SURVEY.md 8b): the reference computes in float64
 *      (lqr/differentiable_lqr.py:169-172, numpy's default), these return outputs at its precision.  Two families:
 *      register-resident kernels in double for the shapes with an instantiation ((1,1) ... (12,3) at 16 lanes per trajectory,
 *      (16,4), (16,8), (32,8) at a wavefront per trajectory; plain and clamped) - the fast path - and, for every other shape
 *      (any size), one lane per trajectory with runtime dimensions and every matrix of a trajectory in `ws`
 *      (dmpc_lqr_f64_workspace_bytes, required in both cases).  dmpc_lqr_f64_path: 1 = 16-lane kernel, 2 = wavefront kernel,
 *      0 = one lane per trajectory.  Arrays are the float64 twins of dmpc_lqr_solve's / dmpc_lqr_kkt_grad's, same shapes;
 *      Ks_out / ks_out may be NULL (both). */
int dmpc_lqr_f64_path(int nx, int nu);
size_t dmpc_lqr_f64_workspace_bytes(int T, int B, int nx, int nu);
int dmpc_lqr_solve_f64(int T, int B, int nx, int nu, const double *C, const double *c, const double *F, const double *f,
                       const double *x_init, const uint8_t *u_zero_mask, double *Ks_out, double *ks_out, double *x_out,
                       double *u_out, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream);
int dmpc_lqr_kkt_grad_f64(int T, int B, int nx, int nu, const double *C, const double *c, const double *F, const double *x,
                          const double *u, const double *grad_x, const double *grad_u, int strict_math, double *d_x_init,
                          double *dC, double *dc, double *dF, double *df, void *ws, size_t ws_bytes, int32_t *info,
                          dmpc_stream_t stream);

/* ---- D. batched LU (util.py:462-482 torch.lu, util.py:505-528 torch.lu_solve in float32) */
/* A [B,n,n] -> LU [B,n,n], piv [B,n] int32 1-based (LAPACK getrf). */
int dmpc_batch_lu_factor(int B, int n, const float *A, float *LU, int32_t *piv, int32_t *info,
                         dmpc_stream_t stream);
/* b [B,n,k] (k = 1 for the reference's 2-D right-hand sides) -> x [B,n,k]. */
int dmpc_batch_lu_solve(int B, int n, int k, const float *LU, const int32_t *piv, const float *b, float *x,
                        dmpc_stream_t stream);

/* ---- C. PNQP (mpc/pnqp.py:37-201): min 1/2 x'Hx + q'x, lower <= x <= upper ------------
 *   x_init NULL = cold start (-H^-1 q clamped).
 *   batch_coupled = 0: per-row termination (= the reference called with a batch of one per row; what shards
 *   across GPUs).  batch_coupled = 1: the reference's own batch semantics - the convergence test and the Armijo
 *   loop are reduced over the WHOLE batch (pnqp.py:139-144,172,187), so a row's answer depends on its batch-mates;
 *   needs `ws` of dmpc_pnqp_workspace_bytes(B, n, n_iter, 1) bytes.  Any n, any B (round 5): n <= 8 with the whole batch
 *   resident runs in registers (one lane per row, grid barriers at the decisions); a batch that is not, and every n > 8, runs
 *   on a fixed resident grid whose workgroups walk their rows piece by piece, the rows' vectors in `ws` (mpc_coupled.hpp;
 *   DMPC_NO_COOP_REGISTER=1 forces that form).  (With only dmpc_coupled_workspace_bytes(1, n_iter) bytes - the decision
 *   slots, what rounds 1-4 asked for - the register form still runs; DMPC_E_WORKSPACE when it is the other form's turn.)
 *   outputs: x [B,n]; fac [B,n,n] = LU of the last free-set Hessian (n == 1: H_f [B,1,1]);
 *   piv [B,n] int32 1-based; index_f [B,n] float {0,1}; n_iter_out [B] int32 (the reference's `i`; one value
 *   for the whole batch when coupled). */
size_t dmpc_coupled_workspace_bytes(int T, int n_qp_iter_max);
size_t dmpc_pnqp_workspace_bytes(int B, int n, int n_iter, int batch_coupled);   /* 0 when not coupled */
int dmpc_pnqp(int B, int n, const float *H, const float *q, const float *lower, const float *upper,
              const float *x_init, int n_iter, int batch_coupled, float *x, float *fac, int32_t *piv,
              float *index_f, int32_t *n_iter_out, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream);

/* ---- E. MPCstep (mpc/mpc_step.py:70-460), LinDx true dynamics + QuadCost true cost --------
 * forward(): Taylor re-centre (need_expand), backward_rec with one PNQP per timestep, forward_rec
 * (clamped rollout + per-trajectory line search on the true cost).
 *   controls/u_lower/u_upper [T,B,nu], states [T,B,nx] (current iterate);
 *   C_hat/c_hat/F_hat/f_hat the quadratic/linear model; C_true/c_true/F_true/f_true the true
 *   QuadCost / LinDx (may alias the model);  f_hat, f_true may be NULL.
 *   outputs: x_out [T,B,nx], u_out [T,B,nu], Ks_out/ks_out (NULL ok), costs [B], old_costs [B]
 *   (NULL ok), alphas [B], objs [T,B] (NULL ok), u_first [T,B,nu] (NULL ok) = the controls of the first
 *   (alpha = 1) line-search pass, from which the reference derives full_du_norm (mpc_step.py:260-263),
 *   n_qp_iter [B] int32 = sum_t (1 + i_t), n_ls_iter [B] int32 = line-search passes run.
 *   Where both halves have a generated instruction stream ((8,2), (3,1), (4,2), (2,2), (1,1), (2,1), (3,2); B % 4 == 0,
 *   per-trajectory termination) they run in ONE launch - a wavefront goes from its sweep straight into its line search
 *   (DMPC_NO_MPC_FUSED=1: two launches, bit-identical results).                                    */
size_t dmpc_mpc_step_workspace_bytes(int T, int B, int nx, int nu);
int dmpc_mpc_step_forward(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                          const float *F_hat, const float *f_hat, const float *controls, const float *states,
                          const float *u_lower, const float *u_upper, const float *C_true, const float *c_true,
                          const float *F_true, const float *f_true, int need_expand, float ls_decay,
                          int max_ls_iter, int n_qp_iter_max, int batch_coupled, float *x_out, float *u_out,
                          float *Ks_out, float *ks_out, float *costs, float *old_costs, float *alphas, float *objs,
                          float *u_first, int32_t *n_qp_iter, int32_t *n_ls_iter, void *ws, size_t ws_bytes,
                          int32_t *info, dmpc_stream_t stream);

/* One read-back per step for the host layer (round 5): the per-trajectory flags and counters of a step, reduced on the device
 * to eight words, so that MPCstep.forward (mpc_step.py:288-328: its NaN asserts, LqrBackOut.n_total_qp_iter,
 * LqrForOut.mean_alphas) synchronises once instead of once per scalar.  info / n_qp_iter / alphas [B] (each may be NULL):
 *   status[0] = OR of info, status[1] = max of n_qp_iter, status[2] = trajectories with DMPC_INFO_NONFINITE,
 *   status[3] = 0, status[4..5] = the bits of the DOUBLE sum of alphas (low word first), status[6..7] = 0.
 * One workgroup; `status` [8] int32 is written, not accumulated into. */
int dmpc_mpc_step_status(int B, const int32_t *info, const int32_t *n_qp_iter, const float *alphas, int32_t *status,
                         dmpc_stream_t stream);

/* The two halves of forward(), separately callable like the reference's methods:
 * backward_rec (mpc_step.py:70-173): c_hat must already be re-centred when need_expand applies;
 *   batch_coupled as in dmpc_pnqp (one PNQP per timestep; `ws` of dmpc_mpc_backward_rec_workspace_bytes(..., 1) bytes;
 *   dmpc_mpc_step_forward / dmpc_box_ddp carve it out of their own workspace).  Any shape, any batch: the register kernels
 *   when the whole batch is resident in one launch of theirs, else mpc_coupled.hpp's fixed grid (slow, same decisions);
 * forward_rec (mpc_step.py:175-286): gains in, line-searched trajectory out.  Its loop condition is batch-global in
 *   the reference too (:196), but there a trajectory's result does not depend on its batch-mates (a trajectory that
 *   is no longer worse keeps its step size and re-computes the same rollout), so it has no coupled mode. */
int dmpc_mpc_backward_rec(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                          const float *F_hat, const float *f_hat, const float *controls, const float *u_lower,
                          const float *u_upper, int n_qp_iter_max, int batch_coupled, float *Ks_out, float *ks_out,
                          int32_t *n_qp_iter, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream);
/* what dmpc_mpc_backward_rec wants in `ws`: batch-coupled, the decision slots and every trajectory's matrices and QP vectors
 * (the fixed-grid form's state); per-trajectory termination, shapes with more than 8 controls or more than 64 columns (the
 * tiled kernels: a workgroup per trajectory, matrices in the workspace): every trajectory's matrices; 0 otherwise (ws may
 * then be NULL). */
size_t dmpc_mpc_backward_rec_workspace_bytes(int T, int B, int nx, int nu, int n_qp_iter_max, int batch_coupled);
int dmpc_mpc_forward_rec(int T, int B, int nx, int nu, const float *Ks, const float *ks, const float *controls,
                         const float *states, const float *u_lower, const float *u_upper, const float *C_true,
                         const float *c_true, const float *F_true, const float *f_true, float ls_decay,
                         int max_ls_iter, float *x_out, float *u_out, float *costs, float *old_costs,
                         float *alphas, float *objs, float *u_first, int32_t *n_ls_iter, int32_t *info,
                         dmpc_stream_t stream);

/* forward_rec with the TRUE dynamics of the pendulum (env_dx/pendulum.py:65-102, simple model: parameters g, m, l;
 * time step dt; torque clamp max_torque) evaluated inside the kernel's line search - the reference calls the
 * Python callable once per timestep and pass (mpc_step.py:237-240).  nx = 3 (cos th, sin th, dth), nu = 1. */
int dmpc_mpc_forward_rec_pendulum(int T, int B, const float *Ks, const float *ks, const float *controls,
                                  const float *states, const float *u_lower, const float *u_upper,
                                  const float *C_true, const float *c_true, float g, float m, float l, float dt,
                                  float max_torque, float ls_decay, int max_ls_iter, float *x_out, float *u_out,
                                  float *costs, float *old_costs, float *alphas, float *objs, float *u_first,
                                  int32_t *n_ls_iter, int32_t *info, dmpc_stream_t stream);

/* Caller-side helper of the pendulum experiments: rollout x_{t+1} = pendulum(x_t, u_t)
 * (util.py:201-277 get_traj with env_dx/pendulum.py:65-102) and its analytic linearisation
 * F_t [x_t;u_t] + f_t = pendulum(x_t, u_t) (mpc/approximate.py:77-119, there by chainer.grad) in one launch.
 *   x_out [T,B,3];  F_out [T-1,B,3,4] or NULL;  f_out [T-1,B,3] or NULL
 *   clamp_grad_closed: derivative of the torque clamp (env_dx/pendulum.py:86, F.clip) AT u = +-max_torque: != 0 -> 1 (the
 *   closed interval, what Chainer's ClipGrad is taken to compute), 0 -> 0.  Box-DDP's bounds equal the torque limit, so
 *   saturated controls sit exactly there. */
int dmpc_pendulum_rollout_linearize(int T, int B, const float *x_init, const float *u, float g, float m, float l,
                                    float dt, float max_torque, int clamp_grad_closed, float *x_out, float *F_out,
                                    float *f_out, dmpc_stream_t stream);

/* Nominal rollout under a LinDx (util.py:239-277 get_traj): x_0 = x_init, x_{t+1} = F_t [x_t; u_t] + f_t (f may be
 * NULL), with the summation order of the MPC step's own line-search rollout - so that a trajectory re-rolled from
 * unchanged controls reproduces the nominal one bit for bit and the box-DDP stop test sees du = 0 at a fixed point. */
int dmpc_lin_rollout(int T, int B, int nx, int nu, const float *x_init, const float *u, const float *F,
                     const float *f, float *x_out, dmpc_stream_t stream);

/* The outer box-DDP loop (BoxDDP.forward, mpc/box_ddp.py:93-230) for a QuadCost and either a LinDx (dyn_kind 0:
 * F [T-1|T,B,nx,ns], f [T-1,B,nx] or NULL) or the built-in pendulum (dyn_kind 1: F = f = NULL, dyn_params = HOST
 * array {g, m, l, dt, max_torque, clamp_grad_closed (0 or 1, as dmpc_pendulum_rollout_linearize)}, nx = 3, nu = 1), as
 * ONE chain of launches: per iteration the nominal rollout
 * (util.py:239-277) and, for the pendulum, its linearisation (mpc/approximate.py:77-119), the MPC step with
 * need_expand (mpc_step.py:288-328), the per-sample "best so far" update (box_ddp.py:200-209) and the stop tests
 * (:223-230), all decided on the device: max_iter iterations are enqueued, those after the stop are no-ops.
 *   u_init [T,B,nu];  outputs x_best [T,B,nx], u_best [T,B,nu], costs_best [B], du_norm_best [B] (full_du_norm of
 *   the best iterate), du_norm_last [B] (of the last executed step, box_ddp.py:263-289 reads it);
 *   state [8] int32: [0] stopped early, [1] iterations run, [2] 1 Converged / 2 Not improved lim / 3 Not Converged,
 *   [3] iterations without improvement, and - reduced by the last launch of the chain, so that a caller needs ONE
 *   read-back per solve - [4] NaN in u_init, [5] some u_lower > u_upper (the input asserts of mpc_step.py:133-138),
 *   [6] trajectories whose info carries DMPC_INFO_NONFINITE (0 without info), [7] du_norm_best > eps somewhere;
 *   scrambled_norm != 0 reproduces the reference's reshape in full_du_norm (mpc_step.py:261-263);
 *   batch_coupled != 0: batch-global PNQP termination inside every step (as dmpc_mpc_backward_rec);
 *   info [B] (optional) is cleared by the call and receives the OR of the MPC step flags of every iteration.
 *   Any shape since round 5 (shapes beyond 8 controls / 64 columns run their steps on the tiled kernels, matrices in `ws`);
 *   DMPC_E_UNSUPPORTED only at T*B*(nx+nu) >= 2^31.  The pendulum at T <= 32, B % 4 == 0: ONE launch per iteration (a workgroup
 *   sweeps its four trajectories, then searches them; the first launch also rolls out and linearises) - 11-12 launches a solve;
 *   DMPC_NO_DDP_ITER_FUSED=1: 22-23, bit-identical.  The chain contains no host synchronisation and no allocation: it can be
 *   recorded in a hipGraph (batch_coupled = 0) - BoxDDP (box_ddp.py) does so by itself for a solve called again on the same buffers. */
size_t dmpc_box_ddp_workspace_bytes(int T, int B, int nx, int nu);
int dmpc_box_ddp(int T, int B, int nx, int nu, const float *x_init, const float *C, const float *c, const float *F,
                 const float *f, int dyn_kind, const float *dyn_params, const float *u_init, const float *u_lower,
                 const float *u_upper, float eps, int not_improved_lim, float ls_decay, int max_ls_iter,
                 float best_cost_eps, int max_iter, int n_qp_iter_max, int scrambled_norm, int batch_coupled,
                 float *x_best, float *u_best, float *costs_best, float *du_norm_best, float *du_norm_last,
                 int32_t *state, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream);

/* backward(): active-set LQR on (-d_tau) + co-state sweeps + outer products (mpc_step.py:330-460).
 *   outputs carry the reference's signs: dC = -1/2(dtau'(x)tau + tau(x)dtau'), dc = -dtau',
 *   dF = -(dlam(x)tau + lam(x)dtau'), df = -dlam[1:] (NULL to skip), dx_init = -dlam[0]; dC, dF may be NULL.
 *   dC_sum [ns,ns], dc_sum [ns] (both or neither): sum over t and b of dC, dc - the gradient of a cost that is ONE (C, c)
 *   tiled over time and batch (env_dx/il_env.py:119-129, the imitation loop's learnable cost), formed in the kernel
 *   instead of by a reduction of [T,B,ns,ns] afterwards; dc may then be NULL too.  DMPC_E_UNSUPPORTED (nothing launched)
 *   unless B % 4 == 0 and the shape has a 16-lane specialisation.
 *   detach_norm [B] (or NULL), detach_flag (device int, or NULL = set), detach_eps: BoxDDP's detach mask for samples that
 *   did not reach a fixed point (mpc/box_ddp.py:263-289), applied to the incoming gradient on the device: if *detach_flag
 *   != 0, trajectories with detach_norm[b] >= detach_eps get grad_x = grad_u = 0 (dmpc_box_ddp's du_norm_last and
 *   state[7] are these two arrays: no read-back between the solve and its gradient). */
int dmpc_mpc_step_backward(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                           const float *F_hat, const float *x, const float *u, const float *u_lower,
                           const float *u_upper, const float *grad_x, const float *grad_u, float *d_x_init,
                           float *dC, float *dc, float *dF, float *df, float *dC_sum, float *dc_sum,
                           const float *detach_norm, const int32_t *detach_flag, float detach_eps, void *ws,
                           size_t ws_bytes, int32_t *info, dmpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMPC_H_ */
