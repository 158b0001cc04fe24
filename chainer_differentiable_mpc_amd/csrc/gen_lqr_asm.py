#!/usr/bin/env python
"""Emit lqr_asm_gen.hpp: the fused LQR solve (Riccati backward sweep + rollout) of the 16-lane row layout as ONE
hand-scheduled gfx950 instruction stream per (nx, nu, write_k).

Why a generator and why whole-kernel asm.  B = 4096 trajectories at four per wavefront is exactly one wavefront
per SIMD, and a lone wavefront issues one instruction every ~4.5 cycles whatever its type
(profiles/r01/microbench_valu_issue.txt: v_fmac 4.45, DPP 5.4, s_nop 4.45).  The kernel's time is therefore its
instruction count.  The HIP version (lqr_dma_kernel.hpp) spends 554 instructions per timestep of which only ~230
are arithmetic; the rest is exec-mask branching around partial DMA chunks, scalar address arithmetic, copies and
s_nops that hipcc puts around asm statements.  Here every instruction is chosen:

  * inputs of one timestep (C, c, F, f of the wave's four trajectories, 3168 B at (8,2)) arrive by FOUR full-width
    LDS-DMA instructions: each lane carries its own 64-bit source pointer (one 16-byte chunk of whichever array
    its position in the slot belongs to) which advances by that array's time stride - no partial chunks, no
    exec masks, no scalar pointer bookkeeping; M0 is written once per group and the instruction offset moves
    both the global and the LDS address;
  * three rotating register sets: the value function V of step t is accumulated IN PLACE in the x-rows of Q_t
    (no copies), while set t-1 is being filled by ds_read_b32 from the ring slot that the DMA of three steps ago
    has completed (counted vmcnt);
  * W = V [F|f] + [0|v] starts with v_mul_f32_dpp (no zero-init) and takes v from lane `aff` with one more DPP
    FMA against a constant unit vector; the 2x2 / 1x1 pivoted solve is spelled out (LAPACK getf2/getrs order,
    reciprocal pivots, one Newton step on v_rcp_f32); gain rows are written to LDS as [K_m | 0 | k_m | pad]
    under an exec mask that also keeps lanes nx..ns-1 of K~ at exactly 0 - which is what makes the in-place value
    update and the forward sweep's in-place control FMAs legal;
  * forward sweep: lane i < nx owns row i of [F_t | f_t] (ring slot), lane nx+m owns gain row m (same shape), so
    one stream of ds_read2_b64 + nx + nu DPP FMAs yields [x_{t+1} | u_t] in one register, stored by ONE
    global_store_dword through per-lane pointers.

Hazards (hipcc pads nothing inside asm) are tracked by the emitter: VALU write -> DPP read of the same VGPR needs
two wait states; a transcendental's result needs one before a non-trans VALU reads it (gfx940 forwarding hazard;
two are kept); s_mov m0 -> LDS-DMA needs one.  Labels reset the tracker pessimistically.

    python chainer_differentiable_mpc_amd/csrc/gen_lqr_asm.py     # rewrites lqr_asm_gen.hpp
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "lqr_asm_gen.hpp")

SHAPES = [(8, 2), (3, 1), (4, 2), (6, 2), (2, 2), (1, 1), (2, 1), (3, 2), (4, 4)]   # (4,4): round 4, plain / gains-out / any-horizon forms
DB = 3        # backward ring depth == number of rotating register sets
DF = int(os.environ.get("GEN_FWD_DEPTH", "6"))   # forward ring depth == unroll (a multiple of 6: lcm of 2 row sets and 3 accumulators)
KROW = 12     # floats per gain row in LDS: [K_m (nx) | 0 (nu) | k_m | pad], 8-byte aligned rows
VBASE = 128   # first VGPR owned by the asm block (operands chosen by hipcc live below)

# timing experiments only (scripts/asm_variants.sh): the results of such builds are wrong on purpose
X_NO_VMWAIT = os.environ.get("GEN_NO_VMWAIT") == "1"    # drop the counted vmcnt waits inside the loops
X_NO_DMA = os.environ.get("GEN_NO_DMA") == "1"          # issue no DMA inside the loops
X_NO_LDSREAD = os.environ.get("GEN_NO_LDSREAD") == "1"  # no ds_read inside the loops
X_SKIP_FWD = os.environ.get("GEN_SKIP_FWD") == "1"
X_SKIP_BWD = os.environ.get("GEN_SKIP_BWD") == "1"
X_FWD = set(os.environ.get("GEN_FWD_SKIP", "").split(","))   # forward-sweep parts to drop: store,pst,stage,read,xpart,upart
X_TIMING = os.environ.get("GEN_TIMING") == "1"
X_NEWTON = os.environ.get("GEN_NEWTON") == "1"          # one Newton step on each v_rcp_f32 (1 ulp -> ~0.5 ulp)
X_WARM_STUBS = os.environ.get("GEN_WARM_STUBS") == "1"  # experiment: run every stash stub once during the prologue wait
X_RET_DIRECT = os.environ.get("GEN_RET_SETPC") != "1"   # stash stubs return by a direct s_branch (else s_setpc_b64)
MFMA_USE = int(os.environ.get("GEN_MFMA_USE", "5"))     # wait states kept before a non-accumulating use of an MFMA result
MFMA_DEP = int(os.environ.get("GEN_MFMA_DEP", "2"))     # wait states kept before an accumulation into the same tile
# box QP: the clamped-set test on the VALU alone (6 VALU per control instead of 4 VALU + 3 SALU that wait for them): MPC step
# at config-3 size 101.7 -> 99.0 us.  (Replacing the scalar OR behind the off-diagonal mask by two selects: no change.)
X_QP_VALU = os.environ.get("GEN_QP_VALU", "1") == "1"
X_G10_MIX = os.environ.get("GEN_G10_MIX") == "1"        # g1 DPP FMAs between (not before) the G MFMAs
# Ring depth of the plain / saving / affine streams (the masked and MPC streams keep DB slots of whole 1 KB pieces: their
# flags and bounds ride in the slot padding).  With more than DB slots the slot stride is the slot's own size (rounded up
# to 64 B) and the last DMA of a group runs under an exec mask - whole pieces would not fit the 160 KB of a CU.
RING_DEPTH = int(os.environ.get("GEN_RING_DEPTH", "3"))
# nt (streaming) cache policy on the backward groups' LDS-DMA, per kind of stream (plain, save, affine, masked, mpc):
# the inputs of a solve are read once; with the default policy they displace each other from L2 / the Infinity Cache on
# their way through (HBM-streamed headline solve 38-40 -> 33.5-35 us, B = 8192: 78 -> 64 us; profiles/r03/ring_ab.txt).
# With steady clocks (profiles/r03/nt_steady_state_variants.txt): save / affine / adj (DiffLqr's two launches) 110 -> 103.7 us
# per training step; masked: no change; mpc: 99 -> 104 us - so the DiffLqr streams carry it too, the other two do not.
X_NT = set(os.environ.get("GEN_NT", "plain,save,affine,adj").split(","))
USE_MFMA = os.environ.get("GEN_NO_MFMA") != "1"         # F^T V F on v_mfma_f32_4x4x1_16b_f32 (else DPP FMAs)          # s_memtime at the phase boundaries -> info[] (no flags then)


class Layout:
    def __init__(self, nx, nu, depth=DB, ghbm=False):
        self.nx, self.nu, self.ns = nx, nu, nx + nu
        self.depth = depth
        ns = self.ns
        self.nC, self.nc, self.nF, self.nf = ns * ns, ns, nx * ns, nx          # 16-byte chunks per wave-step
        self.OFF_C = 0
        self.OFF_c = 16 * self.nC
        self.OFF_F = self.OFF_c + 16 * self.nc
        self.OFF_f = self.OFF_F + 16 * self.nF
        self.nchunk_b = self.nC + self.nc + self.nF + self.nf
        self.ndma_b = (self.nchunk_b + 63) // 64
        self.compact = depth > DB
        self.SLOT_B = self.ndma_b * 1024 if not self.compact else (16 * self.nchunk_b + 63) // 64 * 64
        # lanes of a group's last DMA that stay inside the slot (compact: the others would land in the next slot)
        self.last_lanes = (self.SLOT_B - (self.ndma_b - 1) * 1024) // 16
        self.FOFF_f = 16 * self.nF
        self.nG = nu * KROW // 4 * 1 if ghbm else 0      # ghbm: the wave's 4 trajectories x nu gain rows of KROW floats
        assert (nu * KROW) % 4 == 0
        self.nG = nu * KROW if ghbm else 0               # 16-byte chunks: 4 trajectories x nu x KROW floats / 4
        self.FOFF_G = 16 * (self.nF + self.nf)
        self.nchunk_f = self.nF + self.nf + self.nG
        self.ndma_f = (self.nchunk_f + 63) // 64
        self.SLOT_F = self.ndma_f * 1024
        # one ring size per shape, whatever the stream: RING_DEPTH compact slots or DB slots of whole pieces
        deep = RING_DEPTH * ((16 * self.nchunk_b + 63) // 64 * 64) if RING_DEPTH > DB else 0
        self.RING = max(deep, DB * self.ndma_b * 1024, DF * self.SLOT_F)
        assert depth * self.SLOT_B <= self.RING
        assert self.ndma_b * 1024 - 1024 <= 4095 and ns + 1 <= 12 and nu in (1, 2, 3, 4)
        assert (depth - 1) * self.ndma_b <= 63 and (DF - 1) * self.ndma_f <= 63 and 1 <= self.last_lanes <= 64
        # ---- F stash (registers instead of a second HBM read of F), in the layout the forward sweep consumes:
        # lane i < 8 of a 16-lane row keeps columns [0, H) of row i of F_t, lane 8 + i columns [H, ns) - H
        # accumulation registers per timestep, read from the ring slot as pairs (ds_read2_b32) + a single
        self.stash_ok = (64 % nx == 0) and nx <= 8
        self.H = (ns + 1) // 2
        self.stash_pieces = []         # (registers, dword offset inside the half row)
        o = 0
        while self.H - o >= 2:
            self.stash_pieces.append((2, o))
            o += 2
        if self.H - o == 1:
            self.stash_pieces.append((1, o))
        self.stash_regs = self.H
        self.NSTASH = min(256 // self.stash_regs - (1 if 256 % self.stash_regs == 0 else 0), 64)
        self.SPD = 64 // nx                              # timesteps of f per DMA instruction
        self.NFD = (self.NSTASH + 1 + self.SPD - 1) // self.SPD
        self.FAREA = self.NFD * 1024                     # bytes of f per wave: f[tt] at tt * nx * 16


class Prog:
    """instruction emitter with a small hazard tracker (ages are in wait states since the VALU write)"""

    def __init__(self):
        self.lines = []
        self.age = {}       # vgpr -> wait states since a VALU wrote it
        self.trans = {}     # vgpr -> wait states since a transcendental wrote it
        self.mf = {}        # vgpr -> wait states since an MFMA wrote it
        self.n_instr = 0

    def _tick(self, n=1):
        for d in (self.age, self.trans, self.mf):
            for r in list(d):
                d[r] += n
                if d[r] > 12:
                    del d[r]

    def raw(self, text, ticks=1):
        self.lines.append(text)
        self.n_instr += 1
        self._tick(ticks)

    def nop(self, n):      # n wait states
        if n > 0:
            self.raw("s_nop %d" % (n - 1), ticks=n)

    def comment(self, text):
        self.lines.append("; " + text)

    def label(self, name, reset=True):
        self.lines.append(name + ":")
        if reset:  # anything may have been written right before a jump here
            self.age = {"*": 0}
            self.trans = {"*": 0}
            self.mf = {"*": 0}

    def exec_written(self):
        # SALU write of EXEC -> DPP: not a documented hazard (the documented one is a VALU write, 5 wait states);
        # two wait states are kept anyway
        self.age = {"*": 0}

    def _need(self, table, reg, states):
        have = table.get(reg, table.get("*", 99))
        if have < states:
            self.nop(states - have)

    def mfma(self, dst, a, b, c, abid):
        """v_mfma_f32_4x4x1_16b_f32 dst[4], a, b, c[4] | 0, A broadcast from block `abid` of each 16-lane row.
        Wait states kept: MFMA_USE (5; the ISA asks passes + 2 = 4 for this 2-pass, non-XDL op) before anything
        but an accumulating MFMA reads an MFMA result, MFMA_DEP (2 = passes) before a dependent accumulation, 2 after
        a VALU write of any source.  GEN_MFMA_USE=4 / 8 give bit-identical results (scripts/asm_variant_check.sh)."""
        for r in (a, b):
            self._need(self.age, r, 2)
            self._need(self.trans, r, 2)
            self._need(self.mf, r, MFMA_USE)
        if c:
            for r in c:
                self._need(self.age, r, 2)
                self._need(self.mf, r, MFMA_DEP)
        self.raw("v_mfma_f32_4x4x1_16b_f32 %s, %s, %s, %s cbsz:2 abid:%d" % (vrange(dst), a, b, vrange(c) if c else "0", abid))
        for r in dst:
            self.mf[r] = 0

    def uses(self, regs):
        """a non-VALU instruction (LDS / memory) is about to read or overwrite these VGPRs"""
        for r in regs:
            self._need(self.mf, r, MFMA_USE)

    def valu(self, text, writes=(), reads=(), dpp=None, trans=False):
        for r in tuple(reads) + tuple(writes) + ((dpp,) if dpp else ()):
            self._need(self.mf, r, MFMA_USE)
        if dpp is not None:
            self._need(self.age, dpp, 2)
        for r in tuple(reads) + ((dpp,) if dpp else ()):
            if not trans:
                self._need(self.trans, r, 2)
        self.raw(text)
        for w in writes:
            self.age[w] = 0
            if trans:
                self.trans[w] = 0
            else:
                self.trans.pop(w, None)

    # ---- instruction helpers ------------------------------------------------------------------------------
    def fmac_dpp(self, acc, a, b, lane):
        self.valu("v_fmac_f32_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (acc, a, b, lane),
                  writes=(acc,), reads=(b, acc), dpp=a)

    def mul_dpp(self, dst, a, b, lane):
        self.valu("v_mul_f32_dpp %s, %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (dst, a, b, lane),
                  writes=(dst,), reads=(b,), dpp=a)

    def mov_dpp(self, dst, a, lane):
        self.valu("v_mov_b32_dpp %s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (dst, a, lane),
                  writes=(dst,), dpp=a)

    def v(self, text, writes=(), reads=(), trans=False):
        self.valu(text, writes=writes, reads=reads, trans=trans)

    def text(self):
        return self.lines


class Regs:
    def __init__(self, base):
        self.next = base

    def take(self, n=1, align=1):
        while self.next % align:
            self.next += 1
        r = list(range(self.next, self.next + n))
        self.next += n
        return ["v%d" % i for i in r]


def vrange(regs):
    a, b = int(regs[0][1:]), int(regs[-1][1:])
    assert b - a + 1 == len(regs)
    return "v[%d:%d]" % (a, b)


def gen_kernel(nx, nu, write_k, stash, masked=False, mpc=False, expand=False, ghbm=False, save=False, affine=False, adj=False):
    """masked: LQR_active (mpc/active_constrained_lqr.py:110-137) - clamped controls get a zero right-hand side, Quu
    is zeroed outside free x free with 1e-8 on the clamped diagonal, so their gain rows come out exactly 0 (and the
    rollout needs no change); the value update keeps the unmasked blocks (:143-145).
    mpc: MPCstep.backward_rec (mpc/mpc_step.py:70-173), backward sweep only - the clamped set of every step is not an
    input but the result of the box QP min 1/2 k'Quu k + qu'k, lower - u <= k <= upper - u, solved IN the stream by the
    projected-Newton iteration of mpc/pnqp.py:37-201 (per-trajectory termination, warm-started from the later step,
    wave-uniform loops as pnqp_solve_rows of pnqp_device.hpp); k_t is its solution, K_t the masked solve with 1e-11 on
    the diagonal (the QP's own last factorisation, :147-157), the value update keeps the unmasked blocks (:165-166).
    u, lower, upper of the wave's four trajectories arrive as 12 nu dwords in the slot padding (the flag DMA of the
    masked variant, one float per lane).
    save (write_k only): the training form of the solve - besides K_t, k_t it leaves Quu_t and Qxu_t of every step in HBM,
    which is all DiffLqr.backward's second solve needs next to F (differentiable_lqr.py:95-134: same C, F, hence the
    same gains; only the affine terms change - the `affine` form below).
    affine (stash, no gains out): the re-solve of an LQR problem whose C and F have been solved before (the saving form
    above) with another c, f = 0 - DiffLqr.backward's second solve (differentiable_lqr.py:108-114: c = [grad_x; grad_u],
    x_init = 0).  The gains K_t and the blocks Quu_t, Qxu_t do not depend on c, so only the affine recursion is redone:
    q = c_t + F_t^T v_{t+1}, k_t = -Quu_t^-1 qu, v_t = qx + Qxu_t k_t (lqr_recursion.py:92,119-120,152 with
    qu + Quu k = 0), then the same rollout.  The slot's C region carries [K_t | Qxu_t | Quu_t] instead of C_t (36
    chunks instead of 100 at (8,2); the other lanes of those groups fetch a resident zero chunk): 504 B per
    timestep-solve instead of 832 B, and ~60 instructions per step instead of ~165.
    ghbm (ring form): the gain rows [K_m | 0 | k_m | pad] of every step go to a caller workspace in HBM ([T,B,nu,KROW] floats)
    instead of LDS and come back to the rollout with F and f, through the forward ring - the horizon is then bounded by
    nothing but the workspace (the LDS form holds T <= 74 at (8,2)).
    adj (affine only): DiffLqr.backward in ONE launch that never reads C (differentiable_lqr.py:78-142).  The reference's
    solve is exact block elimination of the KKT system, so its co-states are the value function's gradients:
    lambda_t = V_t x_t + v_t and d_lambda_t = V_t dx_t + v'_t (v' the affine value term of the second solve) for any, also
    non-symmetric, C (tests/test_saved_gains_identity_cpu.py).  With [V_t | v_t] left in HBM by the saving solve the backward
    sweep is the affine recursion above (v'_t kept in LDS, in the area f would take), and the rollout of d_tau computes
    lambda_{t+1}, d_lambda_{t+1} on the way and writes dC_t, dc_t, dF_t, df_t (:128-134) itself - rows through an LDS
    staging area, whole 16-byte chunks to HBM.  [V_t | v_t | x_t | u_t] come through a four-slot LDS-DMA ring in the idle
    backward ring.
    expand (mpc only): need_expand of MPCstep.forward (mpc_step.py:305-317) inside the sweep - the slot padding also
    takes x_t (4 nx more dwords) and every step starts with c_hat = C [x_t; u_t] + c in the affine column."""
    D = DB if (masked or mpc) else RING_DEPTH      # ring slots (the register sets stay three)
    kind = "mpc" if mpc else "masked" if masked else "save" if save else "adj" if adj else "affine" if affine else "plain"
    L = Layout(nx, nu, D, ghbm)
    ns, aff = L.ns, L.ns
    assert not stash or L.stash_ok
    assert not ghbm or not (stash or write_k or masked or mpc or save or affine)
    assert not mpc or (masked and write_k and not stash)
    assert not expand or (mpc and 12 * nu + 4 * nx <= 64)
    assert not save or (write_k and not masked and not mpc)
    assert not affine or (stash and not write_k and not masked and not mpc and not save and ns >= 4)
    assert not adj or affine
    P = Prog()
    R = Regs(VBASE)
    # ---- operand names (C++ side: struct LqrAsmIn of lqr_asm_gen.hpp, filled by lqr_asm_kernel.hpp)
    ptr = ["%%[ptr%d]" % q for q in range(L.ndma_b)]
    str1 = ["%%[str1_%d]" % q for q in range(L.ndma_b)]
    strd = ["%%[str%d]" % q for q in range(L.ndma_b)]
    aq = ["%%[aq%d]" % i for i in range(ns)]
    af = ["%%[af%d]" % k for k in range(nx)]
    fptr = ["%%[fptr%d]" % q for q in range(2 if adj else L.ndma_f)]
    fstr = ["%%[fstr%d]" % q for q in range(2 if adj else L.ndma_f)]
    fp = ["%%[fp%d]" % q for q in range(L.NFD)]
    pk = ["%%[pk%d]" % m for m in range(nu)]

    # ---- fixed registers
    mfma = USE_MFMA and ns <= 12 and nx % 4 == 0 and not affine
    NQ = 4 * ((ns + 3) // 4)            # MFMA accumulator tiles are 4 consecutive rows
    # affine: a set is [c | - | Qxu (nu) | Quu (nu*nu) | K rows (nu)] at (nu = 2: 0, 2, 4, 8; nu = 1: 0, 1, 2, 3)
    AQX, AQU, AKR = (2, 4, 8) if nu == 2 else (1, 2, 3)
    Q = [R.take(max(ns, AKR + nu) if affine else (NQ if mfma else ns), align=4) for _ in range(3)]
    F = [R.take(nx) for _ in range(3)]
    W = R.take(max(nx, 8 if mpc else 4), align=4)     # DPP path: W = V F~ ; MFMA path: G = V^T F~ (rows = x columns of V); mpc: the QP's temporaries
    G10 = R.take(1)[0]                  # MFMA path: row "1" of G^ = F^T v
    A = [R.take(nu, align=4 if save and nu == 2 and m_ == 0 else 1) for m_ in range(nu)]   # save: Quu is staged as one b128
    # save: staging area [Vv | Qxu | Quu] of the wave's four trajectories (bytes), its chunks and store instructions
    SV_Q, SV_U = 16 * nx * (nx + 1), 16 * (nx * (nx + 1) + nx * nu)
    SV_N = nx * (nx + 1) + nx * nu + nu * nu
    SV_NST = (SV_N + 63) // 64
    Kt = R.take(nu)
    Rr = R.take(nu)
    tP, tPQ, tL0, tM1, tRA, tRB, tRP, tT, tLL, tD2, tRD, tY1, tT2 = TMP13 = R.take(13, align=2 if expand else 1)
    MINPIV = R.take(1)[0]
    QB = R.take(1)[0] if affine else None   # affine: second accumulator of q = c + F^T v
    XV0 = R.take(1)[0]                  # x_init (lanes < nx), loaded by the stream itself
    SVD = [R.take(4, align=4) for _ in range(SV_NST)] if save else None   # save: the staged chunks on their way out
    ACT = [R.take(nu) for _ in range(3)] if masked and not mpc else None   # clamped-control flags of the slot in each register set
    EPS = R.take(1)[0] if masked and not mpc else None
    UC = [R.take(nu) for _ in range(3)] if mpc else None      # mpc: u_t, lower_t, upper_t of the slot in each register set
    LB = [R.take(nu) for _ in range(3)] if mpc else None
    UB = [R.take(nu) for _ in range(3)] if mpc else None
    XK = R.take(nu) if mpc else None                          # QP iterate / k_t (the next step's warm start)
    QU = R.take(nu) if mpc else None                          # qu in every lane
    NQP = R.take(1)[0] if mpc else None                       # sum over t of the QP passes run          (mpc_step.py:145)
    QINFO = R.take(1)[0] if mpc else None                     # 4 once a QP ran into the iteration cap
    TAU = R.take(3) if expand else None                       # expand: lane j < ns of set s holds [x_t; u_t][j]
    Am = [R.take(nu) for _ in range(nu)] if masked else None   # masked Quu
    Rm = R.take(nu) if masked else None                        # masked right-hand side rows
    # forward sweep registers reuse the Q / F sets (the backward sweep is over by then)
    RF = Regs(VBASE)
    M = [RF.take(ns, align=4), RF.take(ns, align=4), RF.take(ns, align=4)]
    ACC = RF.take(4)
    assert RF.next <= int(MINPIV[1:])
    last_vgpr = R.next - 1
    assert last_vgpr <= 255

    # stash registers (AGPRs): piece p of stash slot j
    def stash_regs_of(p, j):
        """AGPR numbers of piece p of stash slot j (pairs are 64-bit aligned: all pair regions come first)"""
        base = 0
        for pp in range(p):
            base += L.stash_pieces[pp][0] * L.NSTASH
        w = L.stash_pieces[p][0]
        return [base + j * w + k for k in range(w)]

    n_agpr = L.stash_regs * L.NSTASH if stash else 0
    assert n_agpr <= 256

    S_N, S_TF = "s70", "%[tf]"   # tf: time strides the DMA pointers may still take (an operand: it crosses the two asm blocks)
    S_KM, S_SM, S_UM, S_XM, S_HI = "s[72:73]", "s[74:75]", "s[76:77]", "s[88:89]", "s[90:91]"
    S_RET, S_STUB, S_JMP, S_TMP = "s[78:79]", "s[80:81]", "s[82:83]", "s84"
    S_ROW0 = "s[98:99]"         # save: lane 0 of each 16-lane row
    S_GM = "s[98:99]"           # ghbm: lanes 0..ns of each row (a whole gain row)
    S_AFF = "s[100:101]"        # mpc: the lanes `aff` (the stash pairs above are the QP's masks there - no stash in that mode)

    def mask64(lanes):
        m16 = sum(1 << l for l in lanes)
        return m16 | (m16 << 16)

    km = mask64(list(range(nx)) + [aff])
    sm = mask64(range(ns))
    um = mask64(range(nx, ns))
    xm = mask64(range(nx))
    hi = mask64(range(nx, 16))   # the lanes whose row registers take gain rows (lanes < nx take F rows)
    in_loop = [False]

    PADM = 16 * L.nchunk_b              # masked: the flags of the wave's 4 trajectories land in the slot's padding
    assert not masked or L.SLOT_B - PADM >= 256
    NDB_ALL = L.ndma_b + (1 if masked else 0)   # VMEM operations per backward group

    def issue_group(ptrs, slot, slot_bytes, gap=None):
        """gap: an instruction to emit in the wait state between the M0 write and the first LDS-DMA (else s_nop)"""
        if X_NO_DMA and in_loop[0]:
            if gap:
                P.raw(gap)
            return
        if slot == 0:
            P.raw("s_mov_b32 m0, %[ring]")
        else:
            P.raw("s_add_u32 m0, %%[ring], %d" % (slot * slot_bytes))
        if gap:
            P.raw(gap)
        else:
            P.nop(1)
        for q, p in enumerate(ptrs):
            off = (" offset:%d" % (q * 1024)) if q else ""
            part = ptrs is ptr and L.compact and q == len(ptrs) - 1 and L.last_lanes < 64
            if part:     # the rest of this piece would land in the next slot
                P.raw("s_mov_b64 exec, 0x%x" % ((1 << L.last_lanes) - 1))
            P.raw("global_load_lds_dwordx4 %s, off%s%s" % (p, off, " nt" if kind in X_NT and ptrs is ptr else ""))
            if part:
                P.raw("s_mov_b64 exec, -1")
                P.exec_written()
        if masked and ptrs is ptr:   # 4 * nu flag bytes = nu dwords, one per lane (the other lanes repeat dword 0)
            P.raw("global_load_lds_dword %%[pm], off offset:%d" % PADM)

    uniq = [0]

    def advance(ptrs, strides, by_steps_left=None):
        """move the DMA pointers one timestep back unless they already sit on the first timestep.  In the prologue
        that is counted in tf.  Inside the backward sweep the strides left are a fixed distance from the loop counter
        S_N (step n of the sweep, n = 0 peeled: tf = T-1-D-n, S_N = T-2 for n = 0 and T-1-n after), so the sweep
        tests S_N against by_steps_left (D for the peeled step, D+1 in the loop) and tf is not maintained there."""
        uniq[0] += 1
        lab = "Ladv%d_%%=" % uniq[0]
        if by_steps_left is not None:
            P.raw("s_cmp_ge_i32 %s, %d" % (S_N, by_steps_left))
            P.raw("s_cbranch_scc0 " + lab)
            for p, s in zip(ptrs, strides):
                P.v("v_lshl_add_u64 %s, %s, 0, %s" % (p, p, s))
            if masked and ptrs is ptr:
                P.v("v_lshl_add_u64 %[pm], %[pm], 0, %[dm]")
            P.label(lab, reset=False)
            return
        P.raw("s_cmp_gt_i32 %s, 0" % S_TF)
        P.raw("s_cbranch_scc0 " + lab)
        for p, s in zip(ptrs, strides):
            P.v("v_lshl_add_u64 %s, %s, 0, %s" % (p, p, s))
        if masked and ptrs is ptr:
            P.v("v_lshl_add_u64 %[pm], %[pm], 0, %[dm]")
        P.raw("s_sub_i32 %s, %s, 1" % (S_TF, S_TF))
        P.label(lab, reset=False)   # only pointer registers are written on the fall-through path

    def vmwait(n):
        if not ((X_NO_VMWAIT or X_NO_DMA) and in_loop[0]):
            P.raw("s_waitcnt vmcnt(%d)" % n)

    def read_set(s, slot):
        if X_NO_LDSREAD and in_loop[0]:
            return
        off = slot * L.SLOT_B
        P.uses(Q[s][:ns] + F[s])
        if affine:
            P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][0], aq[0], off))                       # c_t, lane j = c[j]
            if nu == 2:
                P.raw("ds_read_b64 %s, %s offset:%d" % (vrange(Q[s][AQX:AQX + 2]), aq[1], off))  # Qxu row of lane i < nx
                P.raw("ds_read_b128 %s, %s offset:%d" % (vrange(Q[s][AQU:AQU + 4]), aq[2], off)) # Quu, the same in every lane
            else:
                P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][AQX], aq[1], off))
                P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][AQU], aq[2], off))
            for m in range(nu):                                                                  # K rows, lane j < nx = K[m][j]
                P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][AKR + m], aq[3], off + m * nx * 4))
        for i in range(0 if affine else ns):
            P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][i], aq[i], off))
        for k in range(nx):
            P.raw("ds_read_b32 %s, %s offset:%d" % (F[s][k], af[k], off))
        if mpc:
            P.uses(UC[s] + LB[s] + UB[s])
            for a_, regs in enumerate((UC[s], LB[s], UB[s])):     # [u | lower | upper], 4 nu floats each
                for m in range(nu):
                    P.raw("ds_read_b32 %s, %%[am] offset:%d" % (regs[m], off + PADM + (a_ * 4 * nu + m) * 4))
            if expand:
                P.uses([TAU[s]])
                P.raw("ds_read_b32 %s, %%[atau] offset:%d" % (TAU[s], off + PADM))
        elif masked:
            for m in range(nu):
                P.raw("ds_read_u8 %s, %%[am] offset:%d" % (ACT[s][m], off + PADM + m))

    def read_ct(slot):
        """expand: row min(lane, ns-1) of C_t of the slot into the temporaries (lane i: C[i][0..ns-1]) - the operand of the
        re-centring C tau at the top of the step that consumes the slot; issued where the temporaries are idle (end of
        the step before), waited for by that step's first s_waitcnt lgkmcnt(0)"""
        off = slot * L.SLOT_B
        if ns % 2 == 0:
            for k in range(0, ns, 2):
                P.raw("ds_read_b64 %s, %%[act] offset:%d" % (vrange(TMP13[k:k + 2]), off + k * 4))
        else:
            for k in range(ns):
                P.raw("ds_read_b32 %s, %%[act] offset:%d" % (TMP13[k], off + k * 4))

    S_ACT = ["s[92:93]", "s[94:95]"]
    QP_REG = "0x2d2febff"       # 1e-11f  (pnqp.py:73)
    QP_TOLSQ = "0x322bcc76"     # smallest float32 whose root reaches 1e-4f (pnqp.py:140; pnqp_device.hpp kPnqpDxTolSqBits)
    QP_GAMMA = "0x3dcccccd"     # 0.1f: GAMMA (pnqp.py:23) and the step decay (:163)
    QP_MAXLS = 10               # pnqp.py:172

    def rcp_newton(dst, src, tmp):
        """dst = 1/src, v_rcp_f32 + one Newton step (fast_rcp of colwise.hpp: the QP's tests sit on these quotients)"""
        P.v("v_rcp_f32_e32 %s, %s" % (dst, src), writes=(dst,), reads=(src,), trans=True)
        P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tmp, src, dst), writes=(tmp,), reads=(src, dst))
        P.v("v_fmac_f32_e32 %s, %s, %s" % (dst, tmp, dst), writes=(dst,), reads=(tmp, dst))

    def neg_solve(Au, rhs, out):
        """out = -Au^-1 rhs, nu x nu with partial pivoting in the operation order of lu_factor_rinv / lu_solve_rinv
        (colwise.hpp; LAPACK getf2 / getrs), reciprocal pivots with a Newton step"""
        if nu == 1:
            rcp_newton(tRP, Au[0][0], tT)
            P.v("v_mul_f32_e64 %s, %s, -%s" % (out[0], rhs[0], tRP), writes=(out[0],), reads=(rhs[0], tRP))
            return
        a00, a01, a10, a11 = Au[0][0], Au[0][1], Au[1][0], Au[1][1]
        P.v("v_cmp_gt_f32_e64 vcc, |%s|, |%s|" % (a10, a00), reads=(a10, a00))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tP, a00, a10), writes=(tP,), reads=(a00, a10))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tL0, a10, a00), writes=(tL0,), reads=(a00, a10))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tPQ, a01, a11), writes=(tPQ,), reads=(a01, a11))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tM1, a11, a01), writes=(tM1,), reads=(a01, a11))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRA, rhs[0], rhs[1]), writes=(tRA,), reads=(rhs[0], rhs[1]))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRB, rhs[1], rhs[0]), writes=(tRB,), reads=(rhs[0], rhs[1]))
        rcp_newton(tRP, tP, tT)
        P.v("v_mul_f32_e32 %s, %s, %s" % (tLL, tL0, tRP), writes=(tLL,), reads=(tL0, tRP))
        P.v("v_fma_f32 %s, -%s, %s, %s" % (tD2, tLL, tPQ, tM1), writes=(tD2,), reads=(tLL, tPQ, tM1))
        P.v("v_fma_f32 %s, -%s, %s, %s" % (tY1, tLL, tRA, tRB), writes=(tY1,), reads=(tLL, tRA, tRB))
        rcp_newton(tRD, tD2, tT)
        P.v("v_mul_f32_e64 %s, %s, -%s" % (out[1], tY1, tRD), writes=(out[1],), reads=(tY1, tRD))
        P.v("v_fma_f32 %s, %s, %s, %s" % (tT2, tPQ, out[1], tRA), writes=(tT2,), reads=(tPQ, out[1], tRA))
        P.v("v_mul_f32_e64 %s, %s, -%s" % (out[0], tT2, tRP), writes=(out[0],), reads=(tT2, tRP))

    def masked_hessian(reg):
        """Am = Quu on free x free, 0 elsewhere, + `reg` on the diagonal (a register or a literal), from S_ACT"""
        if nu == 2:
            P.raw("s_or_b64 s[96:97], s[92:93], s[94:95]")
        for m in range(nu):
            for l in range(nu):
                if m == l:
                    P.v("v_cndmask_b32_e64 %s, %s, 0, %s" % (Am[m][l], A[m][l], S_ACT[m]), writes=(Am[m][l],), reads=(A[m][l],))
                    P.v("v_add_f32_e32 %s, %s, %s" % (Am[m][l], reg, Am[m][l]), writes=(Am[m][l],), reads=(Am[m][l],))
                else:
                    P.v("v_cndmask_b32_e64 %s, %s, 0, s[96:97]" % (Am[m][l], A[m][l]), writes=(Am[m][l],), reads=(A[m][l],))

    def box_qp(s, first):
        """PNQP (mpc/pnqp.py:37-201) on H = A (Quu, in every lane), q = QU, bounds LB/UB - UC of register set s; warm
        start XK (first: cold start -H^-1 q, :75-83).  Leaves the solution in XK and the clamped set of the last pass in
        S_ACT.  Every lane of a 16-lane row runs it on its trajectory's data; both loops are wave-uniform (they run
        while any lane needs them) and finished lanes keep their state by construction - see pnqp_solve_rows."""
        uniq[0] += 1
        u_ = uniq[0]
        LO, HI = LB[s], UB[s]                       # turned into the QP's bounds in place
        Gv, DXv, XH, GD = W[0:nu], W[2:2 + nu], W[4:4 + nu], W[6:6 + nu]   # W is idle between the Q update and the next step
        AL = G10
        S_DONE, S_SRCH, S_T0, S_T1, S_T2 = "s[78:79]", "s[80:81]", "s[82:83]", "s[86:87]", "s[98:99]"
        S_I, S_CNT = S_TMP, "s85"
        for m in range(nu):
            P.v("v_sub_f32_e32 %s, %s, %s" % (LO[m], LO[m], UC[s][m]), writes=(LO[m],), reads=(LO[m], UC[s][m]))   # mpc_step.py:136-138
            P.v("v_sub_f32_e32 %s, %s, %s" % (HI[m], HI[m], UC[s][m]), writes=(HI[m],), reads=(HI[m], UC[s][m]))
        if first:
            neg_solve(A, QU, XK)
        for m in range(nu):                                                                   # pnqp.py:93
            P.v("v_max_f32_e32 %s, %s, %s" % (XK[m], XK[m], LO[m]), writes=(XK[m],), reads=(XK[m], LO[m]))
            P.v("v_min_f32_e32 %s, %s, %s" % (XK[m], XK[m], HI[m]), writes=(XK[m],), reads=(XK[m], HI[m]))
        P.raw("s_mov_b64 %s, 0" % S_DONE)
        P.raw("s_mov_b32 %s, 0" % S_I)
        P.label("Lqp%d_%%=" % u_, reset=False)   # (the back edge leaves no DPP / transcendental hazard open)
        # grad = Hx + q (:98); clamped = at a bound with the gradient pushing outwards (:110, exact equality)
        for m in range(nu):
            P.v("v_fma_f32 %s, %s, %s, %s" % (Gv[m], A[m][0], XK[0], QU[m]), writes=(Gv[m],), reads=(A[m][0], XK[0], QU[m]))
            for l in range(1, nu):
                P.v("v_fmac_f32_e32 %s, %s, %s" % (Gv[m], A[m][l], XK[l]), writes=(Gv[m],), reads=(A[m][l], XK[l], Gv[m]))
        for m in range(nu):
            if X_QP_VALU:
                # clamped = (x == lo & g > 0) | (x == hi & g < 0)  <=>  max(x == lo ? g : -1, x == hi ? -g : -1) > 0
                # (a NaN gradient gives "not clamped" either way: v_max returns the other operand, the compare is false)
                P.v("v_cmp_eq_f32_e32 vcc, %s, %s" % (XK[m], LO[m]), reads=(XK[m], LO[m]))
                P.v("v_cndmask_b32_e32 %s, -1.0, %s, vcc" % (tRA, Gv[m]), writes=(tRA,), reads=(Gv[m],))
                P.v("v_cmp_eq_f32_e32 vcc, %s, %s" % (XK[m], HI[m]), reads=(XK[m], HI[m]))
                P.v("v_cndmask_b32_e64 %s, -1.0, -%s, vcc" % (tRB, Gv[m]), writes=(tRB,), reads=(Gv[m],))
                P.v("v_max_f32_e32 %s, %s, %s" % (tRA, tRA, tRB), writes=(tRA,), reads=(tRA, tRB))
                P.v("v_cmp_lt_f32_e64 %s, 0, %s" % (S_ACT[m], tRA), reads=(tRA,))
                continue
            P.v("v_cmp_eq_f32_e64 %s, %s, %s" % (S_T0, XK[m], LO[m]), reads=(XK[m], LO[m]))
            P.v("v_cmp_lt_f32_e64 %s, 0, %s" % (S_T1, Gv[m]), reads=(Gv[m],))
            P.v("v_cmp_eq_f32_e64 %s, %s, %s" % (S_T2, XK[m], HI[m]), reads=(XK[m], HI[m]))
            P.v("v_cmp_gt_f32_e64 vcc, 0, %s" % Gv[m], reads=(Gv[m],))
            P.raw("s_and_b64 %s, %s, %s" % (S_T0, S_T0, S_T1))
            P.raw("s_and_b64 %s, %s, vcc" % (S_T2, S_T2))
            P.raw("s_or_b64 %s, %s, %s" % (S_ACT[m], S_T0, S_T2))
        for m in range(nu):
            P.v("v_cndmask_b32_e64 %s, %s, 0, %s" % (GD[m], Gv[m], S_ACT[m]), writes=(GD[m],), reads=(Gv[m],))
        masked_hessian(QP_REG)                                                                # :124-129
        neg_solve(Am, GD, DXv)                                                                # :134-136
        P.v("v_mul_f32_e32 %s, %s, %s" % (tT, DXv[0], DXv[0]), writes=(tT,), reads=(DXv[0],))
        for m in range(1, nu):
            P.v("v_fmac_f32_e32 %s, %s, %s" % (tT, DXv[m], DXv[m]), writes=(tT,), reads=(DXv[m], tT))
        P.v("v_cmp_le_f32_e32 vcc, %s, %s" % (QP_TOLSQ, tT), reads=(tT,))                     # large = |dx| >= 1e-4 (:139-140)
        # one more pass counted for every lane that had not converged before this one (sum = 1 + it, mpc_step.py:145)
        P.raw("s_not_b64 %s, %s" % (S_T0, S_DONE))
        P.v("v_addc_co_u32_e64 %s, %s, %s, 0, %s" % (NQP, S_T1, NQP, S_T0), writes=(NQP,), reads=(NQP,))
        P.raw("s_orn2_b64 %s, %s, vcc" % (S_DONE, S_DONE))                                    # done |= !large  (:141-144)
        P.raw("s_cmp_eq_u64 %s, -1" % S_DONE)
        P.raw("s_cbranch_scc1 Lqpx%d_%%=" % u_)
        # backtracking line search (:162-190); lhs = 1 + 0.5 d'Hd / g'd as in pnqp_device.hpp
        P.v("v_mov_b32_e32 %s, 1.0" % AL, writes=(AL,))
        P.raw("s_not_b64 %s, %s" % (S_SRCH, S_DONE))
        P.raw("s_mov_b32 %s, 0" % S_CNT)
        P.label("Lls%d_%%=" % u_, reset=False)
        for m in range(nu):
            P.v("v_fma_f32 %s, %s, %s, %s" % (XH[m], AL, DXv[m], XK[m]), writes=(XH[m],), reads=(AL, DXv[m], XK[m]))   # :173
            P.v("v_max_f32_e32 %s, %s, %s" % (XH[m], XH[m], LO[m]), writes=(XH[m],), reads=(XH[m], LO[m]))
            P.v("v_min_f32_e32 %s, %s, %s" % (XH[m], XH[m], HI[m]), writes=(XH[m],), reads=(XH[m], HI[m]))
        for m in range(nu):
            P.v("v_sub_f32_e32 %s, %s, %s" % (GD[m], XH[m], XK[m]), writes=(GD[m],), reads=(XH[m], XK[m]))
        P.v("v_mul_f32_e32 %s, %s, %s" % (tRA, Gv[0], GD[0]), writes=(tRA,), reads=(Gv[0], GD[0]))       # g'd
        for m in range(1, nu):
            P.v("v_fmac_f32_e32 %s, %s, %s" % (tRA, Gv[m], GD[m]), writes=(tRA,), reads=(Gv[m], GD[m], tRA))
        for m in range(nu):                                                                    # d'Hd
            P.v("v_mul_f32_e32 %s, %s, %s" % (tRB, A[m][0], GD[0]), writes=(tRB,), reads=(A[m][0], GD[0]))
            for l in range(1, nu):
                P.v("v_fmac_f32_e32 %s, %s, %s" % (tRB, A[m][l], GD[l]), writes=(tRB,), reads=(A[m][l], GD[l], tRB))
            if m == 0:
                P.v("v_mul_f32_e32 %s, %s, %s" % (tLL, GD[0], tRB), writes=(tLL,), reads=(GD[0], tRB))
            else:
                P.v("v_fmac_f32_e32 %s, %s, %s" % (tLL, GD[m], tRB), writes=(tLL,), reads=(GD[m], tRB, tLL))
        rcp_newton(tRD, tRA, tT)
        P.v("v_mul_f32_e32 %s, 0.5, %s" % (tLL, tLL), writes=(tLL,), reads=(tLL,))
        P.v("v_fma_f32 %s, %s, %s, 1.0" % (tLL, tLL, tRD), writes=(tLL,), reads=(tLL, tRD))              # :175-176
        P.v("v_cmp_ge_f32_e32 vcc, %s, %s" % (QP_GAMMA, tLL), reads=(tLL,))                               # lhs <= GAMMA
        P.raw("s_and_b64 vcc, vcc, %s" % S_SRCH)                                                          # fails
        P.v("v_mul_f32_e32 %s, %s, %s" % (tT, QP_GAMMA, AL), writes=(tT,), reads=(AL,))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (AL, AL, tT), writes=(AL,), reads=(AL, tT))            # :185-186
        P.raw("s_add_u32 %s, %s, 1" % (S_CNT, S_CNT))
        P.raw("s_cmp_lt_u32 %s, %d" % (S_CNT, QP_MAXLS))
        P.raw("s_cselect_b64 %s, vcc, 0" % S_SRCH)                                                        # :172
        P.raw("s_cmp_lg_u64 %s, 0" % S_SRCH)
        P.raw("s_cbranch_scc1 Lls%d_%%=" % u_)
        for m in range(nu):                                                                                # :190
            P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (XK[m], XH[m], XK[m], S_DONE), writes=(XK[m],), reads=(XH[m], XK[m]))
        P.raw("s_add_u32 %s, %s, 1" % (S_I, S_I))
        P.raw("s_cmp_lt_u32 %s, %%[nqp_iter]" % S_I)
        P.raw("s_cbranch_scc1 Lqp%d_%%=" % u_)
        # iteration cap (:192): lanes that never converged
        P.v("v_cndmask_b32_e64 %s, 4, 0, %s" % (tT, S_DONE), writes=(tT,))
        P.v("v_or_b32_e32 %s, %s, %s" % (QINFO, tT, QINFO), writes=(QINFO,), reads=(tT, QINFO))
        P.label("Lqpx%d_%%=" % u_, reset=False)

    swap_id = [0]

    def gains(s, first=False):
        """K~ = -Quu^-1 [Qux | Quu | qu] per lane (lqr_recursion.py:112-120); leaves A (Quu), Kt, and MINPIV"""
        Qs = Q[s]
        if nu <= 2:      # (three and four controls: nobody needs Quu in every lane - the rows are eliminated where they lie and
            for m in range(nu):      # the value update broadcasts Quu inside its FMAs)
                for l in range(nu):
                    P.mov_dpp(A[m][l], Qs[nx + m], nx + l)
        rhs = [Qs[nx + m] for m in range(nu)]
        Au = A
        if mpc:
            for m in range(nu):
                P.mov_dpp(QU[m], Qs[nx + m], aff)
            box_qp(s, first)
            for m in range(nu):      # K_t: rows of clamped controls zero, the QP's last H_f (mpc_step.py:147-157)
                P.v("v_cndmask_b32_e64 %s, %s, 0, %s" % (Rm[m], Qs[nx + m], S_ACT[m]), writes=(Rm[m],), reads=(Qs[nx + m],))
            # Am still holds that H_f: the QP's last pass built it from the same clamped set (a lane that converged
            # earlier rebuilt the same one from its frozen iterate)
            rhs = Rm
            Au = Am
        elif masked:
            for m in range(nu):
                P.v("v_cmp_ne_u32_e64 %s, 0, %s" % (S_ACT[m], ACT[s][m]), reads=(ACT[s][m],))
            if nu == 2:
                P.raw("s_or_b64 s[96:97], s[92:93], s[94:95]")
            for m in range(nu):
                P.v("v_cndmask_b32_e64 %s, %s, 0, %s" % (Rm[m], Qs[nx + m], S_ACT[m]), writes=(Rm[m],), reads=(Qs[nx + m],))
                for l in range(nu):
                    if m == l:
                        P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (Am[m][l], A[m][l], EPS, S_ACT[m]), writes=(Am[m][l],),
                            reads=(A[m][l], EPS))
                    else:
                        P.v("v_cndmask_b32_e64 %s, %s, 0, s[96:97]" % (Am[m][l], A[m][l]), writes=(Am[m][l],), reads=(A[m][l],))
            rhs = Rm
            Au = Am
        if nu >= 3:
            # Three and four controls (round 4): Gauss-Jordan on the rows of [Qux | Quu | qu] where they lie (the HIP kernels'
            # gauss_jordan_rows, riccati_blocks.hpp).  Row m is ONE register across the lanes and the multiplier of row i at
            # pivot k one lane of it (lane nx + k): a DPP broadcast per multiplier, an FMA per row.  Partial pivoting in
            # LAPACK's order: the candidates below the pivot bubble the first largest entry into row k (strict >: the first
            # maximum wins; the remaining rows end up permuted differently from getf2's single interchange, which changes
            # neither which values the later pivots see nor which register holds which unknown).  Working rows: W (the G
            # tiles are dead here); the K registers themselves are written once, under the K mask, as before.
            assert not (masked or mpc or save) and len(W) >= nu and nu <= 4
            Wk = W[:nu]
            LI = [tL0, tM1, tRA, tRB][:nu]
            for m in range(nu):
                P.v("v_mov_b32_e32 %s, %s" % (Wk[m], rhs[m]), writes=(Wk[m],), reads=(rhs[m],))
            for k in range(nu):
                P.mov_dpp(tP, Wk[k], nx + k)
                for i in range(k + 1, nu):
                    P.mov_dpp(LI[i], Wk[i], nx + k)
                if k + 1 < nu:
                    # the interchange behind a wave-uniform branch: taken only when some trajectory of the wavefront has a
                    # larger entry below its pivot (a Quu that is close to diagonally dominant never does)
                    cand = [LI[i] for i in range(k + 1, nu)]
                    if len(cand) == 3:
                        P.v("v_max3_f32 %s, |%s|, |%s|, |%s|" % (tT, cand[0], cand[1], cand[2]), writes=(tT,), reads=tuple(cand))
                        mx = "%s" % tT
                    elif len(cand) == 2:
                        P.v("v_max_f32_e64 %s, |%s|, |%s|" % (tT, cand[0], cand[1]), writes=(tT,), reads=tuple(cand))
                        mx = "%s" % tT
                    else:
                        mx = "|%s|" % cand[0]
                    P.v("v_cmp_gt_f32_e64 vcc, %s, |%s|" % (mx, tP), reads=(tT, cand[0], tP))
                    swap_id[0] += 1
                    P.raw("s_cbranch_vccz Lnoswap%d_%%=" % swap_id[0])
                for i in range(k + 1, nu):
                    P.v("v_cmp_gt_f32_e64 vcc, |%s|, |%s|" % (LI[i], tP), reads=(LI[i], tP))
                    P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tT, Wk[k], Wk[i]), writes=(tT,), reads=(Wk[k], Wk[i]))
                    P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (Wk[i], Wk[i], Wk[k]), writes=(Wk[i],), reads=(Wk[k], Wk[i]))
                    P.v("v_mov_b32_e32 %s, %s" % (Wk[k], tT), writes=(Wk[k],), reads=(tT,))
                    P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tT2, tP, LI[i]), writes=(tT2,), reads=(tP, LI[i]))
                    P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (LI[i], LI[i], tP), writes=(LI[i],), reads=(tP, LI[i]))
                    P.v("v_mov_b32_e32 %s, %s" % (tP, tT2), writes=(tP,), reads=(tT2,))
                if k + 1 < nu:
                    P.label("Lnoswap%d_%%=" % swap_id[0])
                P.v("v_min_f32_e64 %s, |%s|, %s" % (MINPIV, tP, MINPIV), writes=(MINPIV,), reads=(tP, MINPIV))
                rcp_newton(tRP, tP, tT)
                P.v("v_mul_f32_e32 %s, %s, %s" % (Wk[k], Wk[k], tRP), writes=(Wk[k],), reads=(Wk[k], tRP))
                for i in range(nu):
                    if i == k:
                        continue
                    l_ = LI[i]
                    if i < k:
                        l_ = tLL
                        P.mov_dpp(tLL, Wk[i], nx + k)
                    P.v("v_fma_f32 %s, -%s, %s, %s" % (Wk[i], l_, Wk[k], Wk[i]), writes=(Wk[i],), reads=(l_, Wk[k], Wk[i]))
            P.raw("s_mov_b64 exec, " + S_KM)
            for m in range(nu):
                P.v("v_mul_f32_e32 %s, -1.0, %s" % (Kt[m], Wk[m]), writes=(Kt[m],), reads=(Wk[m],))
        elif nu == 1:
            P.v("v_rcp_f32_e32 %s, %s" % (tRP, Au[0][0]), writes=(tRP,), reads=(Au[0][0],), trans=True)
            P.v("v_min_f32_e64 %s, |%s|, %s" % (MINPIV, Au[0][0], MINPIV), writes=(MINPIV,), reads=(Au[0][0], MINPIV))
            P.nop(1)
            if X_NEWTON:
                P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, Au[0][0], tRP), writes=(tT,), reads=(Au[0][0], tRP))
                P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP))
            P.raw("s_mov_b64 exec, " + S_KM)
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[0], rhs[0], tRP), writes=(Kt[0],), reads=(rhs[0], tRP))
        else:
            a00, a01, a10, a11 = Au[0][0], Au[0][1], Au[1][0], Au[1][1]
            P.v("v_cmp_gt_f32_e64 vcc, |%s|, |%s|" % (a10, a00), reads=(a10, a00))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tP, a00, a10), writes=(tP,), reads=(a00, a10))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tL0, a10, a00), writes=(tL0,), reads=(a00, a10))
            P.v("v_rcp_f32_e32 %s, %s" % (tRP, tP), writes=(tRP,), reads=(tP,), trans=True)
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tPQ, a01, a11), writes=(tPQ,), reads=(a01, a11))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tM1, a11, a01), writes=(tM1,), reads=(a01, a11))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRA, rhs[0], rhs[1]), writes=(tRA,), reads=(rhs[0], rhs[1]))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRB, rhs[1], rhs[0]), writes=(tRB,), reads=(rhs[0], rhs[1]))
            if X_NEWTON:
                P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, tP, tRP), writes=(tT,), reads=(tP, tRP))
                P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP))
            P.v("v_mul_f32_e32 %s, %s, %s" % (tLL, tL0, tRP), writes=(tLL,), reads=(tL0, tRP))
            P.v("v_fma_f32 %s, -%s, %s, %s" % (tD2, tLL, tPQ, tM1), writes=(tD2,), reads=(tLL, tPQ, tM1))
            P.v("v_rcp_f32_e32 %s, %s" % (tRD, tD2), writes=(tRD,), reads=(tD2,), trans=True)
            P.v("v_fma_f32 %s, -%s, %s, %s" % (tY1, tLL, tRA, tRB), writes=(tY1,), reads=(tLL, tRA, tRB))
            P.v("v_min3_f32 %s, |%s|, |%s|, %s" % (MINPIV, tP, tD2, MINPIV), writes=(MINPIV,), reads=(tP, tD2, MINPIV))
            if X_NEWTON:
                P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, tD2, tRD), writes=(tT,), reads=(tD2, tRD))
                P.v("v_fmac_f32_e32 %s, %s, %s" % (tRD, tT, tRD), writes=(tRD,), reads=(tT, tRD))
            P.raw("s_mov_b64 exec, " + S_KM)
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[1], tY1, tRD), writes=(Kt[1],), reads=(tY1, tRD))
            P.v("v_fma_f32 %s, %s, %s, %s" % (tT2, tPQ, Kt[1], tRA), writes=(tT2,), reads=(tPQ, Kt[1], tRA))
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[0], tT2, tRP), writes=(Kt[0],), reads=(tT2, tRP))
        if mpc:   # k_t is the QP's solution (lane aff), not the masked solve's          mpc_step.py:141-146
            for m in range(nu):
                P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (Kt[m], Kt[m], XK[m], S_AFF), writes=(Kt[m],), reads=(Kt[m], XK[m]))
        # gain rows -> LDS (and HBM when the caller wants Ks/ks), still under the K mask
        if ghbm:     # ... -> the workspace, whole rows [K_m | 0 | k_m]: lanes nx..ns-1 of K~ are exactly 0, lane aff is column ns
            P.uses(Kt)
            P.raw("s_mov_b64 exec, " + S_GM)
            for m in range(nu):
                P.raw("global_store_dword %%[pgw], %s, off offset:%d" % (Kt[m], m * KROW * 4))
        elif not mpc:
            for m in range(nu):
                off = (" offset:%d" % (m * KROW * 4)) if m else ""
                P.raw("ds_write_b32 %%[ak], %s%s" % (Kt[m], off))
        if write_k:
            P.uses(Kt)
            for m in range(nu):
                P.raw("global_store_dword %s, %s, off" % (pk[m], Kt[m]))
        if save:
            # The saved blocks leave through an LDS staging area as the contiguous 16-byte chunks they form in HBM (the
            # 8-byte row pieces of Qxu, 36-byte rows of [V | v] cost more as stores than the solve's arithmetic: the saving
            # solve took 68 us with them against 40 us plain).  Qxu_t: rows i < nx of Q~ hold it in lanes nx..ns-1 (the value
            # update below writes over them); Quu_t: every lane has it in A - lane 0 of each row writes the nu * nu floats
            P.uses(Qs[:nx] + [r_ for row_ in A for r_ in row_])
            P.raw("s_mov_b64 exec, " + S_UM)
            for i in range(nx):
                P.raw("ds_write_b32 %%[asq], %s offset:%d" % (Qs[i], SV_Q + i * nu * 4))
            P.raw("s_mov_b64 exec, " + S_ROW0)
            quu = [r_ for row_ in A for r_ in row_]
            if nu == 2 and int(quu[0][1:]) % 4 == 0:
                P.raw("ds_write_b128 %%[asu], %s offset:%d" % (vrange(quu), SV_U))
            else:
                for j_, r_ in enumerate(quu):
                    P.raw("ds_write_b32 %%[asu], %s offset:%d" % (r_, SV_U + 4 * j_))
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
        if write_k:
            for m in range(nu):
                P.v("v_lshl_add_u64 %s, %s, 0, %%[dk]" % (pk[m], pk[m]))
        if ghbm:
            P.v("v_lshl_add_u64 %[pgw], %[pgw], 0, %[dgw]")
        elif not mpc:
            P.v("v_add_u32_e32 %%[ak], %d, %%[ak]" % ((-nu * KROW * 4) & 0xffffffff))

    def gains_affine(s, first):
        """One step of the affine recursion.  The step is a chain of dependent instructions on a wavefront that has its
        SIMD to itself, so everything that does not depend on v_{t+1} - the factorisation of Quu_t (pivot choice, both
        reciprocals) - is interleaved with the chain that does, q = c_t + F_t^T v_{t+1} (two accumulators):
            q = c_t + F_t^T v_{t+1}   k_t = -Quu_t^-1 q_u (every lane)   v_t = q_x + Qxu_t k_t (lanes < nx of G10)
        and the gain rows [K_t | 0 | k_t] go to LDS for the rollout."""
        Qs = Q[s]
        QUa, KK = W[0:nu], W[2:2 + nu]
        Aa = [[Qs[AQU + m * nu + l] for l in range(nu)] for m in range(nu)]
        S_PIV = "s[86:87]"
        fact, qch = [], []
        if nu == 1:
            fact.append(lambda: P.v("v_rcp_f32_e32 %s, %s" % (tRP, Aa[0][0]), writes=(tRP,), reads=(Aa[0][0],), trans=True))
            fact.append(lambda: P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, Aa[0][0], tRP), writes=(tT,), reads=(Aa[0][0], tRP)))
            fact.append(lambda: P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP)))
        else:
            a00, a01, a10, a11 = Aa[0][0], Aa[0][1], Aa[1][0], Aa[1][1]
            fact.append(lambda: P.v("v_cmp_gt_f32_e64 %s, |%s|, |%s|" % (S_PIV, a10, a00), reads=(a10, a00)))
            fact.append(lambda: P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tP, a00, a10, S_PIV), writes=(tP,), reads=(a00, a10)))
            fact.append(lambda: P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tL0, a10, a00, S_PIV), writes=(tL0,), reads=(a00, a10)))
            fact.append(lambda: P.v("v_rcp_f32_e32 %s, %s" % (tRP, tP), writes=(tRP,), reads=(tP,), trans=True))
            fact.append(lambda: P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tPQ, a01, a11, S_PIV), writes=(tPQ,), reads=(a01, a11)))
            fact.append(lambda: P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tM1, a11, a01, S_PIV), writes=(tM1,), reads=(a01, a11)))
            fact.append(lambda: P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, tP, tRP), writes=(tT,), reads=(tP, tRP)))
            fact.append(lambda: P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP)))
            fact.append(lambda: P.v("v_mul_f32_e32 %s, %s, %s" % (tLL, tL0, tRP), writes=(tLL,), reads=(tL0, tRP)))
            fact.append(lambda: P.v("v_fma_f32 %s, -%s, %s, %s" % (tD2, tLL, tPQ, tM1), writes=(tD2,), reads=(tLL, tPQ, tM1)))
            fact.append(lambda: P.v("v_rcp_f32_e32 %s, %s" % (tRD, tD2), writes=(tRD,), reads=(tD2,), trans=True))
            fact.append(lambda: P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT2, tD2, tRD), writes=(tT2,), reads=(tD2, tRD)))
            fact.append(lambda: P.v("v_fmac_f32_e32 %s, %s, %s" % (tRD, tT2, tRD), writes=(tRD,), reads=(tT2, tRD)))
        if not first:       # q = c_t + F_t^T v_{t+1}: lane j takes column j of F_t, v_{t+1}[k] broadcast from lane k  (:92)
            for k in range(nx):
                if k == 1:
                    qch.append(lambda k=k: P.mul_dpp(QB, G10, F[s][k], k))
                elif k % 2 == 1:
                    qch.append(lambda k=k: P.fmac_dpp(QB, G10, F[s][k], k))
                else:
                    qch.append(lambda k=k: P.fmac_dpp(Qs[0], G10, F[s][k], k))
        while fact or qch:
            if qch:
                qch.pop(0)()
            if fact:
                fact.pop(0)()
        if not first and nx > 1:
            P.v("v_add_f32_e32 %s, %s, %s" % (Qs[0], QB, Qs[0]), writes=(Qs[0],), reads=(QB, Qs[0]))
        for m in range(nu):
            P.mov_dpp(QUa[m], Qs[0], nx + m)
        if nu == 1:
            P.v("v_mul_f32_e64 %s, %s, -%s" % (KK[0], QUa[0], tRP), writes=(KK[0],), reads=(QUa[0], tRP))
        else:                # the back half of neg_solve (LAPACK getrs order)
            P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tRA, QUa[0], QUa[1], S_PIV), writes=(tRA,), reads=(QUa[0], QUa[1]))
            P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (tRB, QUa[1], QUa[0], S_PIV), writes=(tRB,), reads=(QUa[0], QUa[1]))
            P.v("v_fma_f32 %s, -%s, %s, %s" % (tY1, tLL, tRA, tRB), writes=(tY1,), reads=(tLL, tRA, tRB))
            P.v("v_mul_f32_e64 %s, %s, -%s" % (KK[1], tY1, tRD), writes=(KK[1],), reads=(tY1, tRD))
            P.v("v_fma_f32 %s, %s, %s, %s" % (tT2, tPQ, KK[1], tRA), writes=(tT2,), reads=(tPQ, KK[1], tRA))
            P.v("v_mul_f32_e64 %s, %s, -%s" % (KK[0], tT2, tRP), writes=(KK[0],), reads=(tT2, tRP))
        P.v("v_fma_f32 %s, %s, %s, %s" % (G10, Qs[AQX], KK[0], Qs[0]), writes=(G10,), reads=(Qs[AQX], KK[0], Qs[0]))
        for m in range(1, nu):
            P.v("v_fmac_f32_e32 %s, %s, %s" % (G10, Qs[AQX + m], KK[m]), writes=(G10,), reads=(Qs[AQX + m], KK[m], G10))
        for m in range(nu):
            P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (Kt[m], Qs[AKR + m], KK[m], S_AFF), writes=(Kt[m],),
                reads=(Qs[AKR + m], KK[m]))
        P.raw("s_mov_b64 exec, " + S_KM)
        for m in range(nu):
            off = (" offset:%d" % (m * KROW * 4)) if m else ""
            P.raw("ds_write_b32 %%[ak], %s%s" % (Kt[m], off))
        if adj:      # v'_t (lanes < nx of G10) goes where f_t would be: the rollout's d_lambda_t = V_t dx_t + v'_t reads it
            P.raw("s_mov_b64 exec, " + S_XM)
            P.raw("ds_write_b32 %%[avp], %s" % G10)
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
        P.v("v_add_u32_e32 %%[ak], %d, %%[ak]" % ((-nu * KROW * 4) & 0xffffffff))
        if adj:
            P.v("v_add_u32_e32 %%[avp], %d, %%[avp]" % ((-nx * 16) & 0xffffffff))

    def vupdate(s):
        """V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~) in place in Q[s][0..nx-1]   (lqr_recursion.py:151-152)"""
        Qs = Q[s]
        if nu >= 3:      # R = Q~u. + Quu K~ with Quu[m][l] broadcast from lane nx + l of row m inside the FMA
            for m in range(nu):
                P.v("v_mov_b32_e32 %s, %s" % (Rr[m], Qs[nx + m]), writes=(Rr[m],), reads=(Qs[nx + m],))
            for l in range(nu):
                for m in range(nu):
                    P.fmac_dpp(Rr[m], Qs[nx + m], Kt[l], nx + l)
        for m in range(nu if nu <= 2 else 0):
            P.v("v_fma_f32 %s, %s, %s, %s" % (Rr[m], A[m][0], Kt[0], Qs[nx + m]), writes=(Rr[m],),
                reads=(A[m][0], Kt[0], Qs[nx + m]))
            for l in range(1, nu):
                P.v("v_fmac_f32_e32 %s, %s, %s" % (Rr[m], A[m][l], Kt[l]), writes=(Rr[m],), reads=(A[m][l], Kt[l]))
        for m in range(nu):
            for i in range(nx):
                P.fmac_dpp(Qs[i], Qs[i], Kt[m], nx + m)
        if mfma:      # V~ += K^T R: the A^T B shape again (A = K~ row m, block i/4)
            for m in range(nu):
                for I in range(nx // 4):
                    P.mfma(Qs[4 * I:4 * I + 4], Kt[m], Rr[m], Qs[4 * I:4 * I + 4], I)
        else:
            for m in range(nu):
                for i in range(nx):
                    P.fmac_dpp(Qs[i], Kt[m], Rr[m], i)

    def bstep(step, first, extra_outstanding=0):
        """step `step` of the sweep (0 is the peeled first step; the loop form passes its body position, the step modulo
        the loop length): register set step % 3, ring slot step % D"""
        s, slot, nslot = step % 3, step % D, (step + 1) % D
        p, n = (s + 2) % 3, (s + 1) % 3
        V = Q[p]
        issue_group(ptr, slot, L.SLOT_B, gap="s_waitcnt lgkmcnt(0)")   # the slot's last reads are in before it is refilled
        advance(ptr, strd, by_steps_left=D if first else D + 1)
        if expand:
            # c_hat = c + C tau: lane i holds row i of C_t (read_ct), tau_k comes from lane k of TAU by a DPP broadcast -
            # two accumulators; then entry i of the sum goes to the affine lane of row i of Q~ (one DPP FMA against e_aff).
            # (Until round 2's last day the rows of Q~ - columns per lane - were multiplied by tau and summed across the
            # lanes by four rotations each: 61 instructions instead of 2 ns + 2 and ns / 2 LDS reads.)
            assert ns + 2 <= len(TMP13)
            ACCA, ACCB = TMP13[ns], TMP13[ns + 1]
            P.mul_dpp(ACCA, TAU[s], TMP13[0], 0)
            if ns > 1:
                P.mul_dpp(ACCB, TAU[s], TMP13[1], 1)
            for k in range(2, ns):
                P.fmac_dpp(ACCA if k % 2 == 0 else ACCB, TAU[s], TMP13[k], k)
            if ns > 1:
                P.v("v_add_f32_e32 %s, %s, %s" % (ACCA, ACCB, ACCA), writes=(ACCA,), reads=(ACCA, ACCB))
            for i in range(ns):
                P.fmac_dpp(Q[s][i], ACCA, "%[eaff]", i)
        if affine:
            pass    # (all of the step's arithmetic comes after the reads of the next step have been issued: gains_affine)
        elif not first and mfma:
            # Q~ += F~^T V^ F^ as (V^^T F^)^T F^ - both products have the A^T B shape that an outer-product MFMA
            # computes from column-per-lane registers (A operand = 4 lanes of a register = 4 rows of A^T):
            #   G[b][j]  = sum_a V[a][b] F~[a][j]      rows b = x columns: tiles of 4, A = V[a] block b/4, B = F~[a]
            #   g1[j]    = sum_a v[a] F~[a][j]         the row of the homogeneous coordinate (8 DPP FMAs)
            # This is the reference's own association, (F^T V) F  (lqr_recursion.py:89,96).
            if not X_G10_MIX:
                P.mul_dpp(G10, V[0], F[s][0], aff)
                for a_ in range(1, nx):
                    P.fmac_dpp(G10, V[a_], F[s][a_], aff)
            for a_ in range(nx):
                for I in range(nx // 4):
                    Gt = W[4 * I:4 * I + 4]
                    P.mfma(Gt, V[a_], F[s][a_], Gt if a_ else None, I)
                if X_G10_MIX:     # the g1 FMAs sit between dependent accumulations of the same tile
                    if a_ == 0:
                        P.mul_dpp(G10, V[0], F[s][0], aff)
                    else:
                        P.fmac_dpp(G10, V[a_], F[s][a_], aff)
        elif not first:
            # W~ = V~ F~  (+ v in column aff):  W[i] = sum_k bcast<k>(V[i]) F[k] + bcast<aff>(V[i]) e_aff
            for i in range(nx):
                P.mul_dpp(W[i], V[i], F[s][0], 0)
            for k in range(1, nx):
                for i in range(nx):
                    P.fmac_dpp(W[i], V[i], F[s][k], k)
            for i in range(nx):
                P.fmac_dpp(W[i], V[i], "%[eaff]", aff)
        # The stores of a step (gains, the saved blocks) are younger than its DMA group and retire in order with it
        # (one counter, gfx9): between the group this step needs and now lie D-1 younger groups AND the stores of the
        # D-1 steps since - allowing for them keeps D-1 groups in flight (without, eleven stores per step would leave
        # one).  Body positions j < D-1 also run step j of the sweep, where only j steps' stores and the prologue's
        # extra loads lie in between: the smaller of the two counts.
        st_allow = 0
        if not first and not mpc and n_step_stores:
            st_allow = (D - 1) * n_step_stores
            if step <= D - 2:
                st_allow = step * n_step_stores + min(n_extra, (D - 1 - step) * n_step_stores)
        vmwait((D - 1) * NDB_ALL + extra_outstanding + st_allow)
        read_set(n, nslot)
        if stash:
            # F of the NEXT step goes from its ring slot into this step's stash registers: the register numbers
            # differ per step, so the two reads live in a table of stubs (one per step) that is called here
            lo = int(S_STUB[2:S_STUB.index(":")])
            if X_RET_DIRECT:
                callsite[0] += 1
                P.raw("s_setpc_b64 " + S_STUB)
                P.lines.append(".p2align 6")
                P.label("Lret%d_%%=" % callsite[0], reset=False)
                bret["first" if first else step] = callsite[0]
            else:
                P.raw("s_swappc_b64 %s, %s" % (S_RET, S_STUB))
            P.raw("s_add_u32 s%d, s%d, %d" % (lo, lo, BSTUB))
            P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
        if affine:
            gains_affine(s, first)
            return
        if not first and mfma:
            #   Q~[i][j] += sum_b G[b][i] F~[b][j] + g1[i] e_aff[j]    tiles of 4 rows i: A = G[b] block i/4, B = F~[b]
            for b_ in range(nx):
                for I in range(NQ // 4):
                    Qt = Q[s][4 * I:4 * I + 4]
                    P.mfma(Qt, W[b_], F[s][b_], Qt, I)
            for I in range(NQ // 4):
                Qt = Q[s][4 * I:4 * I + 4]
                P.mfma(Qt, G10, "%[eaff]", Qt, I)
        elif not first:
            # Q~ += F~^T W~ ; the u-rows first in the last pass so that the Quu broadcasts need no wait states
            for k in range(nx):
                order = list(range(ns)) if k < nx - 1 else list(range(nx, ns)) + list(range(nx))
                for i in order:
                    P.fmac_dpp(Q[s][i], F[s][k], W[k], i)
        gains(s, first)
        if expand:
            read_ct(nslot)   # (the last step reads a slot nobody consumes)
        vupdate(s)
        if save:
            # the value function of the step, [V_t | v_t] (row i of Q~ now: V_t[i][:] in lanes < nx, v_t[i] in lane aff), as
            # rows of nx + 1 floats: what the one-pass gradient reads instead of C (lambda_t = V_t x_t + v_t, the `adj` form);
            # then [Vv | Qxu | Quu] of the wave's four trajectories go out as whole chunks
            P.uses(Q[s][:nx])
            P.raw("s_mov_b64 exec, " + S_KM)
            for i in range(nx):
                P.raw("ds_write_b32 %%[asv], %s offset:%d" % (Q[s][i], i * (nx + 1) * 4))
            P.raw("s_mov_b64 exec, -1")
            P.exec_written()
            for q in range(SV_NST):
                P.raw("ds_read_b128 %s, %%[ach] offset:%d" % (vrange(SVD[q]), q * 1024))
            P.raw("s_waitcnt lgkmcnt(0)")
            for q in range(SV_NST):
                part = SV_N - 64 * q
                if part <= 32:
                    P.raw("s_mov_b64 exec, 0x%x" % ((1 << part) - 1))
                elif part < 64:
                    P.raw("s_mov_b32 exec_hi, 0x%x" % ((1 << (part - 32)) - 1))
                P.raw("global_store_dwordx4 %%[pso%d], %s, off" % (q, vrange(SVD[q])))
                if part < 64:
                    P.raw("s_mov_b64 exec, -1")
                    P.exec_written()
            for q in range(SV_NST):
                P.v("v_lshl_add_u64 %%[pso%d], %%[pso%d], 0, %%[sso%d]" % (q, q, q))

    n_step_stores = 0      # global stores per backward step (not mpc)
    if write_k or ghbm:
        n_step_stores += nu
    if save:
        n_step_stores += SV_NST
    assert (D - 1) * (NDB_ALL + n_step_stores) <= 63
    LC = 3 * D // (3 if D % 3 == 0 else 1)     # steps per trip of the loop form: lcm(register sets, ring slots)
    callsite = [0]
    bret = {}      # body position of the loop (or "first") -> return label number of its call site
    BSTUB = 32     # bytes per backward stub (two LDS reads + s_setpc_b64 = 20)
    assert 8 * len(L.stash_pieces) + 4 <= BSTUB

    TS = R.take(4) if X_TIMING and not mpc else []     # (the mpc streams have no registers left for the stamps)
    last_vgpr = R.next - 1

    def stamp(i):
        if TS:
            P.raw("s_memtime s[86:87]")
            P.raw("s_waitcnt lgkmcnt(0)")
            P.v("v_mov_b32_e32 %s, s86" % TS[i], writes=(TS[i],))

    # =============================================================== backward sweep
    P.comment("---- prologue (the first DMA groups are already in flight)")
    P.raw("s_waitcnt lgkmcnt(0)")
    stamp(0)
    for name, val in ((S_KM, km), (S_SM, sm), (S_UM, um), (S_XM, xm), (S_HI, hi)):
        lo = int(name[2:name.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo, val & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo + 1, val & 0xffffffff))
    for m in range(nu):
        P.v("v_mov_b32_e32 %s, 0" % Kt[m], writes=(Kt[m],))
    P.v("v_mov_b32_e32 %s, 0x7f7fffff" % MINPIV, writes=(MINPIV,))
    if ghbm:
        lo_g = int(S_GM[2:S_GM.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_g, mask64(range(ns + 1)) & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_g + 1, mask64(range(ns + 1)) & 0xffffffff))
    if save:
        lo_r0 = int(S_ROW0[2:S_ROW0.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_r0, mask64([0]) & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_r0 + 1, mask64([0]) & 0xffffffff))
    if affine:
        lo_aff = int(S_AFF[2:S_AFF.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_aff, mask64([aff]) & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_aff + 1, mask64([aff]) & 0xffffffff))
    if mpc:
        for r_ in XK + [NQP, QINFO]:
            P.v("v_mov_b32_e32 %s, 0" % r_, writes=(r_,))
        lo_aff = int(S_AFF[2:S_AFF.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_aff, mask64([aff]) & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo_aff + 1, mask64([aff]) & 0xffffffff))
    elif masked:
        P.v("v_mov_b32_e32 %s, 0x322bcc77" % EPS, writes=(EPS,))   # 1e-8f (active_constrained_lqr.py:121-122)
    if stash:
        lo = int(S_STUB[2:S_STUB.index(":")])
        P.raw("s_getpc_b64 " + S_STUB)
        P.label("Lpcb_%=", reset=False)
        P.raw("s_add_u32 s%d, s%d, Lbstub_%%=-Lpcb_%%=" % (lo, lo))
        P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
    # The first DB groups are issued by a SEPARATE asm block (issue_first) that the C++ side runs as soon as the
    # DMA pointers exist - the rest of the operand set-up then overlaps their flight.
    P_main = P
    P = Prog()
    P.raw("s_sub_i32 %s, %%[T], 1" % S_TF)
    issue_group(ptr, 0, L.SLOT_B)
    advance(ptr, str1)
    for j in range(1, D):
        issue_group(ptr, j, L.SLOT_B)
        advance(ptr, strd)
    P_first = P
    P = P_main
    n_extra = 0
    if stash and not adj:
        # all of f (the forward sweep's only input from memory) goes to LDS now, BEHIND the first groups: the
        # backward sweep's first two waits allow for these NFD younger operations, later ones are merely conservative
        n_extra = L.NFD
        for q in range(L.NFD):
            if q == 0:
                P.raw("s_mov_b32 m0, %[farea]")
            else:
                P.raw("s_add_u32 m0, %%[farea], %d" % (q * 1024))
            P.nop(1)
            P.raw("global_load_lds_dwordx4 %s, off" % fp[q])
    P.raw("global_load_dword %s, %%[pxi], off" % XV0)
    n_extra += 1
    # zero this wave's gain rows while the first slots are in flight (columns nx..ns-1 and the pad of every row
    # are never written afterwards; the region is padded to whole 1 KB pieces by lqr_asm_kernel.hpp)
    if not mpc and not ghbm:
        for i in range(4):
            P.v("v_mov_b32_e32 %s, 0" % W[i], writes=(W[i],))
        P.raw("s_mov_b32 %s, %%[nz]" % S_TMP)
        P.label("Lzero_%=", reset=False)
        P.raw("ds_write_b128 %%[gz], %s" % vrange(W[0:4]))
        P.v("v_add_u32_e32 %[gz], 0x400, %[gz]")
        P.raw("s_sub_u32 %s, %s, 1" % (S_TMP, S_TMP))
        P.raw("s_cmp_lg_u32 %s, 0" % S_TMP)
        P.raw("s_cbranch_scc1 Lzero_%=")
    if stash and X_WARM_STUBS and not X_RET_DIRECT:
        lo = int(S_STUB[2:S_STUB.index(":")])
        P.raw("s_mov_b64 s[82:83], %s" % S_STUB)
        P.raw("s_mov_b32 %s, %d" % (S_TMP, L.NSTASH))
        P.label("Lwarm_%=", reset=False)
        P.raw("s_swappc_b64 %s, s[82:83]" % S_RET)
        P.raw("s_add_u32 s82, s82, %d" % BSTUB)
        P.raw("s_addc_u32 s83, s83, 0")
        P.raw("s_sub_u32 %s, %s, 1" % (S_TMP, S_TMP))
        P.raw("s_cmp_lg_u32 %s, 0" % S_TMP)
        P.raw("s_cbranch_scc1 Lwarm_%=")
    P.raw("s_waitcnt vmcnt(%d)" % ((D - 1) * NDB_ALL + n_extra))
    read_set(0, 0)
    if expand:
        read_ct(0)
    P.raw("s_sub_i32 %s, %%[T], 2" % S_N)      # steps left after the first, minus one
    P.comment("---- t = T-1")
    stamp(1)
    bstep(0, True, n_extra)
    if X_SKIP_BWD:
        P.raw("s_branch Lbwd_done_%=")
    in_loop[0] = True
    if True:       # (one form of the sweep: the loop over lcm(register sets, ring slots) steps)
        P.label("Lbwd_%=")
        for k in range(1, LC + 1):       # steps k, k + LC, k + 2 LC, ... of the sweep
            P.comment("---- backward step, register set %d, ring slot %d" % (k % 3, k % D))
            bstep(k, False)
            P.raw("s_sub_u32 %s, %s, 1" % (S_N, S_N))       # SCC = borrow: that was the last step
            if k != LC:
                P.raw("s_cbranch_scc1 Lbwd_done_%=")
            else:
                P.raw("s_cbranch_scc0 Lbwd_%=")
    P.label("Lbwd_done_%=")
    in_loop[0] = False
    n_bwd = P.n_instr
    if mpc:      # backward sweep only
        P.v("v_mov_b32_e32 %[xvout], 0")
        P.v("v_mov_b32_e32 %%[nqp], %s" % NQP)
        P.v("v_mov_b32_e32 %%[qpinfo], %s" % QINFO)
        P.raw("s_branch Ldone_%=")

    # =============================================================== forward rollout
    def read_rows(c, a, aff_too=True):
        if X_NO_LDSREAD and in_loop[0]:
            return
        Mc = M[c]
        if ns % 2 == 0:
            i = 0
            while i + 4 <= ns:
                P.raw("ds_read2_b64 %s, %%[arow] offset0:%d offset1:%d" % (vrange(Mc[i:i + 4]), i // 2, i // 2 + 1))
                i += 4
            if i < ns:
                P.raw("ds_read_b64 %s, %%[arow] offset:%d" % (vrange(Mc[i:i + 2]), i * 4))
        else:
            i = 0
            while i + 2 <= ns:
                P.raw("ds_read2_b32 %s, %%[arow] offset0:%d offset1:%d" % (vrange(Mc[i:i + 2]), i, i + 1))
                i += 2
            if i < ns:
                P.raw("ds_read_b32 %s, %%[arow] offset:%d" % (Mc[i], i * 4))
        if aff_too:
            P.raw("ds_read_b32 %s, %%[aaff]" % ACC[a])

    def fcompute(c, a, ap, mask, last, fillers=()):
        """fillers: independent work emitted in the wait states between the dependent control FMAs"""
        fillers = list(fillers)
        skip = X_FWD if in_loop[0] else set()
        for jj in range(nx):                       # u_t = K_t x_t + k_t (lanes nx+m) ; f_t + Fx x_t (lanes < nx)
            if "xpart" not in skip:
                P.fmac_dpp(ACC[a], ACC[ap], M[c][jj], jj)
        if not last:
            for m in range(nu):                    # x_{t+1} += Fu u_t : the gain rows hold 0 in these columns
                if fillers:
                    fillers.pop(0)()
                if "upart" not in skip:
                    P.fmac_dpp(ACC[a], ACC[a], M[c][nx + m], nx + m)
        for f in fillers:
            f()
        if adj:        # d_tau leaves as dc, through the staging area
            return
        if "store" not in skip:
            P.raw("s_mov_b64 exec, " + mask)
            P.raw("global_store_dword %%[pst], %s, off" % ACC[a])
            P.raw("s_mov_b64 exec, -1")
            P.exec_written()
        if not last and "pst" not in skip:
            P.v("v_lshl_add_u64 %[pst], %[pst], 0, %[dst]")

    P.comment("---- forward rollout")
    stamp(2)
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    P.v("v_mov_b32_e32 %[xvout], 0")
    P.raw("v_readfirstlane_b32 %s, %%[bwd_only]" % S_TMP)   # (a VGPR operand: hipcc ran out of SGPRs for an "s" one at (1,1))
    P.raw("s_cmp_lg_u32 %s, 0" % S_TMP)                    # LqrRecursion.backward(): gains only
    P.raw("s_cbranch_scc1 Ldone_%=")
    if not adj:
        P.raw("s_mov_b64 exec, " + S_XM)                   # x_0 = x_init
        P.raw("global_store_dword %%[px0], %s, off" % XV0)
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
    if not stash:
        rare = []      # ghbm: out-of-line tails of the pointer steps

        def advance_f():
            """the forward DMA pointers go one timestep on.  ghbm: the gain rows have a slice T-1, F and f have not - the
            last step moves the gain lanes alone (`fstrl`: their stride, 0 in the F / f lanes); it lives out of line, the
            common path is the same four instructions"""
            if not ghbm:
                advance(fptr, fstr)
                return
            uniq[0] += 1
            lab_r, lab_b = "Lfrare%d_%%=" % uniq[0], "Lfback%d_%%=" % uniq[0]
            P.raw("s_cmp_gt_i32 %s, 1" % S_TF)
            P.raw("s_cbranch_scc0 " + lab_r)
            for p_, s_ in zip(fptr, fstr):
                P.v("v_lshl_add_u64 %s, %s, 0, %s" % (p_, p_, s_))
            P.raw("s_sub_i32 %s, %s, 1" % (S_TF, S_TF))
            P.label(lab_b, reset=False)
            rare.append((lab_r, lab_b))

        P.raw("s_sub_i32 %s, %%[T], %d" % (S_TF, 1 if ghbm else 2))
        for j in range(DF):
            issue_group(fptr, j, L.SLOT_F)
            advance_f()
        P.raw("s_waitcnt vmcnt(%d)" % ((DF - 1) * L.ndma_f))
        read_rows(0, 0)
        P.v("v_mov_b32_e32 %s, %s" % (ACC[2], XV0), writes=(ACC[2],))
        P.raw("s_sub_i32 %s, %%[T], 1" % S_N)      # full steps t = 0 .. T-2
        P.raw("s_cmp_lg_u32 %s, 0" % S_N)
        P.raw("s_cbranch_scc0 Lfin0_%=")
        if X_SKIP_FWD:
            P.raw("s_branch Lfin0_%=")
        in_loop[0] = True
        P.label("Lfwd_%=")
        n_fwd0 = P.n_instr
        for j in range(DF):
            c, a = j % 2, j % 3
            o, an, ap = 1 - c, (a + 1) % 3, (a + 2) % 3
            P.comment("---- forward step, slot %d" % j)
            P.raw("s_waitcnt lgkmcnt(0)")
            issue_group(fptr, j, L.SLOT_F)
            advance_f()
            vmwait((DF - 1) * L.ndma_f)
            d = "2" if j == DF - 1 else ""
            P.v("v_add_u32_e32 %%[arow], %%[drow%s], %%[arow]" % d)
            P.v("v_add_u32_e32 %%[aaff], %%[daff%s], %%[aaff]" % d)
            read_rows(o, an)
            fcompute(c, a, ap, S_SM, False)
            P.raw("s_sub_i32 %s, %s, 1" % (S_N, S_N))
            P.raw("s_cmp_lg_u32 %s, 0" % S_N)
            if j < DF - 1:
                P.raw("s_cbranch_scc0 Lfin%d_%%=" % (j + 1))
            else:
                P.raw("s_cbranch_scc1 Lfwd_%=")
                P.raw("s_branch Lfin0_%=")
        n_fwd = P.n_instr - n_fwd0
        n_fwd_steps = DF
        in_loop[0] = False
        for j in range(DF):                            # t = T-1: only u_{T-1}
            c, a = j % 2, j % 3
            ap = (a + 2) % 3
            P.label("Lfin%d_%%=" % j)
            P.raw("s_waitcnt lgkmcnt(0)")
            fcompute(c, a, ap, S_UM, True)
            P.v("v_mov_b32_e32 %%[xvout], %s" % ACC[a])
            P.raw("s_branch Ldone_%=")
        for lab_r, lab_b in rare:       # (reached by their branches only)
            P.label(lab_r, reset=False)
            P.raw("s_cmp_eq_i32 %s, 1" % S_TF)
            P.raw("s_cbranch_scc0 " + lab_b)
            for q_, p_ in enumerate(fptr):
                P.v("v_lshl_add_u64 %s, %s, 0, %%[fstrl%d]" % (p_, p_, q_))
            P.raw("s_sub_i32 %s, %s, 1" % (S_TF, S_TF))
            P.raw("s_branch " + lab_b)
    else:
        # F comes back from the stash registers through two staging buffers in the (now idle) ring: the code is
        # unrolled over n = T-1-t (the stash slot is a register NUMBER), entered at n = T-1 through a table of
        # entry stubs, and runs down to n = 1 without any loop control; n = 0 is the u-only step t = T-1.
        # Rows are prepared TWO steps ahead (three row sets, four accumulators): a step is only ~25 instructions,
        # shorter than the LDS round trip stash -> staging -> rows.
        def sets(n):
            return n % 3, n % 4, (n + 1) % 4     # row set, accumulator, x_t register of step n

        def prefetch_a(m, move=True):
            """first half of the preparation of step m (>= 0): F_t rows out of the stash registers (lanes < 8 get
            columns [0, H) directly and columns [H, ns) from lane + 8 by a row rotation), row pointers moved"""
            Mc = M[sets(m)[0]]
            if m >= 1 and not ("stage" in X_FWD and in_loop[0]):
                regs = [r for p in range(len(L.stash_pieces)) for r in stash_regs_of(p, m - 1)]
                for jj in range(L.H):
                    P.v("v_accvgpr_read_b32 %s, a%d" % (Mc[jj], regs[jj]), writes=(Mc[jj],))
                for jj in range(L.H):
                    if L.H + jj < ns:
                        P.valu("v_mov_b32_dpp %s, %s row_ror:8 row_mask:0xf bank_mask:0xf" % (Mc[L.H + jj], Mc[jj]),
                               writes=(Mc[L.H + jj],), dpp=Mc[jj])
            if move:   # the gain-row / f pointers go from step m+1 to step m
                P.v("v_add_u32_e32 %[arow], %[drowo], %[arow]")
                P.v("v_add_u32_e32 %[aaff], %[daff], %[aaff]")

        def prefetch_b(m):
            """second half: the gain rows of step m into lanes nx..15 of the same registers (AFTER the VALU writes
            above - a later VALU write would clobber them), then f_t / k_t into the accumulator of step m"""
            c, a, _ = sets(m)
            if "read" in X_FWD and in_loop[0]:
                return
            P.raw("s_mov_b64 exec, " + S_HI)
            read_rows(c, a, aff_too=False)
            P.raw("s_mov_b64 exec, -1")
            P.exec_written()
            P.raw("ds_read_b32 %s, %%[aaff]" % ACC[a])

        def prefetch(m, move=True):
            """stage + read the rows of step m (>= 0); returns the number of LDS operations issued"""
            before = len(P.lines)
            prefetch_a(m, move)
            prefetch_b(m)
            return sum(1 for ln in P.lines[before:] if ln.startswith("ds_"))

        def group_size(m):      # LDS operations of prefetch(m), without emitting anything
            if m < 0:
                return 0
            save = (list(P.lines), P.n_instr, dict(P.age), dict(P.trans))
            k = prefetch(m)
            P.lines, P.n_instr, P.age, P.trans = save
            return k

        if adj:
            # =========================================================== the one-pass gradient's forward sweep
            # Step n (time t = T-1-n) rolls d_tau out as above and, with [V | v | x | u] of time t+1 from the ring,
            #   lambda_{t+1} = V_{t+1} x_{t+1} + v_{t+1}        d_lambda_{t+1} = V_{t+1} dx_{t+1} + v'_{t+1}
            #   dC_t = wa dtau (x) tau + wb tau (x) dtau   dc_t = dtau   dF_t = d_lambda_{t+1} (x) tau + lambda_{t+1} (x) dtau
            #   df_t = d_lambda_t (the reference's index, differentiable_lqr.py:133) or d_lambda_{t+1} (strict)
            # (differentiable_lqr.py:128-134; wa = 0.5, wb = 1 reproduce :128's precedence).  Rows are columns-per-lane
            # registers (lane j = column j), written to the staging area under the mask of the tau lanes and sent to
            # HBM as the contiguous 16-byte chunks they form there: [dC | dc | dF | df] of the wave's four trajectories.
            DFA = 4                                     # ring slots of [Vv | x | u]
            nVv, nX, nU = nx * (nx + 1), nx, nu         # 16-byte chunks per wave-step
            nda = (nVv + nX + nU + 63) // 64
            SLOTA = nda * 1024
            STG = DFA * SLOTA                           # staging area, byte offset in the ring
            nout = L.nchunk_b                           # [dC | dc | dF | df]: ns^2 + ns + nx ns + nx chunks
            nst = (nout + 63) // 64
            assert STG + 16 * nout <= L.RING and (DFA - 1) * nda + 4 * nst <= 63
            O_dc, O_dF, O_df = 16 * ns * ns, 16 * (ns * ns + ns), 16 * (ns * ns + ns + nx * ns)
            fixed = set(r_ for grp in M for r_ in grp) | set(ACC) | {MINPIV, XV0} | set(TS)
            pool = ["v%d" % i for i in range(VBASE, 256) if "v%d" % i not in fixed]

            def take(k, align=1):
                for st in range(len(pool)):
                    blk = pool[st:st + k]
                    if len(blk) == k and int(blk[0][1:]) % align == 0 and int(blk[-1][1:]) - int(blk[0][1:]) == k - 1:
                        del pool[st:st + k]
                        return blk
                raise AssertionError("out of forward registers")

            VR = [take(nx, align=2) for _ in range(2)]  # row min(lane, nx-1) of V of the step two ahead / one ahead
            LN = take(2)                                # v -> lambda_{t+1}
            DLN = take(3)                               # v' -> d_lambda: accumulating, current (df), being loaded
            TAUR = take(3)
            DTAU, HDT, WT, DFV = take(4)
            RC = take(ns)
            RFm = take(nx)
            SD = [take(4, align=4) for _ in range(nst)]
            S_DFSEL = "s[86:87]"

            def load_data(m):
                """LDS reads of [V | v | tau] of step m from its ring slot (v' comes with prefetch_b) - issued two steps
                ahead of their use, behind the counted wait for the slot's DMA group"""
                off = (m % DFA) * SLOTA
                Vs = VR[m % 2]
                av = "%%[avr%d]" % (m % DFA)           # (ds_read2_b32 reaches 1 KB: one row address per slot)
                for j in range(0, nx - 1, 2):
                    P.raw("ds_read2_b32 %s, %s offset0:%d offset1:%d" % (vrange(Vs[j:j + 2]), av, j, j + 1))
                if nx % 2:
                    P.raw("ds_read_b32 %s, %s offset:%d" % (Vs[nx - 1], av, (nx - 1) * 4))
                P.raw("ds_read_b32 %s, %s offset:%d" % (LN[m % 2], av, nx * 4))
                P.raw("ds_read_b32 %s, %%[atx] offset:%d" % (TAUR[m % 3], off))

            def issue_data(m):
                """LDS-DMA group of step m (time T-1-m; the pointers walk forward in time) into slot m % DFA"""
                if m < 0:
                    return
                issue_group(fptr[:nda], m % DFA, SLOTA)
                for q in range(nda):
                    P.v("v_lshl_add_u64 %s, %s, 0, %s" % (fptr[q], fptr[q], fstr[q]))

            def adj_prefetch_b(m):
                """gain rows of step m into lanes nx..15, k_t into the accumulator (f = 0 in its state lanes), v'_t"""
                c, a, _ = sets(m)
                P.v("v_mov_b32_e32 %s, 0" % ACC[a], writes=(ACC[a],))
                P.raw("ds_read_b32 %s, %%[aaff]" % DLN[m % 3])
                P.raw("s_mov_b64 exec, " + S_HI)
                read_rows(c, a, aff_too=False)
                P.raw("ds_read_b32 %s, %%[aaff]" % ACC[a])
                P.raw("s_mov_b64 exec, -1")
                P.exec_written()

            def store_chunks(n_valid):
                """the first n_valid chunks of the staging area -> HBM, then the store pointers move on one timestep"""
                nq = (n_valid + 63) // 64
                for q in range(nq):
                    P.raw("ds_read_b128 %s, %%[ach] offset:%d" % (vrange(SD[q]), STG + q * 1024))
                P.raw("s_waitcnt lgkmcnt(0)")
                return nq

            def outputs(n, last):
                """everything of step n after its rollout"""
                c, a, ap = sets(n)
                TAU = TAUR[n % 3]
                P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (DTAU, ACC[a], ACC[ap], S_XM), writes=(DTAU,), reads=(ACC[a], ACC[ap]))
                P.v("v_mul_f32_e32 %s, %%[wa], %s" % (HDT, DTAU), writes=(HDT,), reads=(DTAU,))
                P.v("v_mul_f32_e32 %s, %%[wb], %s" % (WT, TAU), writes=(WT,), reads=(TAU,))
                if not last:      # d_lambda_{t+1} += V_{t+1} dx_{t+1}
                    Vs, dl = VR[(n - 1) % 2], DLN[(n - 1) % 3]
                    for j in range(nx):
                        P.fmac_dpp(dl, ACC[a], Vs[j], j)
                for i in range(ns):
                    P.mul_dpp(RC[i], HDT, TAU, i)
                for i in range(ns):
                    P.fmac_dpp(RC[i], WT, DTAU, i)
                if not last:
                    ln, dl = LN[(n - 1) % 2], DLN[(n - 1) % 3]
                    for i in range(nx):
                        P.mul_dpp(RFm[i], dl, TAU, i)
                    for i in range(nx):
                        P.fmac_dpp(RFm[i], ln, DTAU, i)
                    P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (DFV, DLN[n % 3], dl, S_DFSEL), writes=(DFV,), reads=(DLN[n % 3], dl))
                P.raw("s_mov_b64 exec, " + S_SM)
                for i in range(ns):
                    P.raw("ds_write_b32 %%[awc], %s offset:%d" % (RC[i], STG + i * ns * 4))
                P.raw("ds_write_b32 %%[awe], %s offset:%d" % (DTAU, STG + O_dc))
                if not last:
                    for i in range(nx):
                        P.raw("ds_write_b32 %%[awf], %s offset:%d" % (RFm[i], STG + O_dF + i * ns * 4))
                    P.raw("s_mov_b64 exec, " + S_XM)
                    P.raw("ds_write_b32 %%[awd], %s offset:%d" % (DFV, STG + O_df))
                P.raw("s_mov_b64 exec, -1")
                P.exec_written()
                n_valid = nout if not last else ns * ns + ns
                nq = store_chunks(n_valid)
                if not last and n - 2 - DFA >= 0:
                    issue_data(n - 2 - DFA)       # the slot read at the top of this step is free (the wait above)
                P.uses([r_ for q in range(nq) for r_ in SD[q]])
                for q in range(nq):
                    part = n_valid - 64 * q
                    if part <= 32:
                        P.raw("s_mov_b64 exec, 0x%x" % ((1 << part) - 1))       # (a 32-bit literal, zero-extended)
                    elif part < 64:
                        P.raw("s_mov_b32 exec_hi, 0x%x" % ((1 << (part - 32)) - 1))
                    P.raw("global_store_dwordx4 %%[pso%d], %s, off" % (q, vrange(SD[q])))
                    if part < 64:
                        P.raw("s_mov_b64 exec, -1")
                        P.exec_written()
                if not last:
                    for q in range(nst):
                        P.v("v_lshl_add_u64 %%[pso%d], %%[pso%d], 0, %%[sso%d]" % (q, q, q))

            def data_wait(n):
                """counted wait for the DMA group of step n-2 at the top of step n: younger are the groups of the steps in
                between that exist, and the stores of the DFA steps since it was issued"""
                younger = len([m for m in range(n - 1 - DFA, n - 2) if m >= 0])
                P.raw("s_waitcnt vmcnt(%d)" % (younger * nda + DFA * nst))

            def emit_stub(n):
                """entry at step n = T-1 (time 0): the first DFA groups, everything landed once (the only exposed round
                trip), tau_0, d_lambda_0 = v'_0 = dx_init, the data of time 1, the rows of the first two steps; then the
                next two groups - and, so that the counted waits of the first steps see as many younger stores as later
                ones do, dx_init stored 2 nst + 1 times instead of once"""
                c, a, ap = sets(n)
                for m in range(n, n - DFA, -1):
                    issue_data(m)
                P.raw("s_waitcnt vmcnt(0)")
                prefetch_a(n, move=False)
                adj_prefetch_b(n)
                P.raw("ds_read_b32 %s, %%[atx] offset:%d" % (TAUR[n % 3], (n % DFA) * SLOTA))
                prefetch_a(n - 1)
                adj_prefetch_b(n - 1)
                load_data(n - 1)
                P.v("v_mov_b32_e32 %s, 0" % ACC[ap], writes=(ACC[ap],))       # dx_0 = 0
                P.raw("s_waitcnt lgkmcnt(0)")
                issue_data(n - DFA)
                issue_data(n - DFA - 1)
                P.raw("s_mov_b64 exec, " + S_XM)
                for _ in range(2 * nst + 1):
                    P.raw("global_store_dword %%[px0], %s, off" % DLN[n % 3])
                P.raw("s_mov_b64 exec, -1")
                P.exec_written()
                P.raw("s_branch Lfs%d_%%=" % n)

            P_keep = P
            stubs = []
            for n in range(1, L.NSTASH + 1):
                P = Prog()
                emit_stub(n)
                stubs.append(P)
            P = P_keep
            FSTUB = 64
            while FSTUB < 8 * max(q.n_instr for q in stubs):
                FSTUB *= 2
            lo = int(S_JMP[2:S_JMP.index(":")])
            P.v("v_cmp_ne_u32_e64 %s, 0, %%[dfshift]" % S_DFSEL)
            P.raw("s_getpc_b64 " + S_JMP)
            P.label("Lpcf_%=", reset=False)
            P.raw("s_sub_i32 %s, %%[T], 2" % S_TMP)                 # entry stub index: (T-1) - 1
            P.raw("s_mul_i32 %s, %s, %d" % (S_TMP, S_TMP, FSTUB))
            P.raw("s_add_u32 s%d, s%d, Lfstub_%%=-Lpcf_%%=" % (lo, lo))
            P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
            P.raw("s_add_u32 s%d, s%d, %s" % (lo, lo, S_TMP))
            P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
            P.raw("s_setpc_b64 " + S_JMP)
            n_fwd0 = P.n_instr
            for n in range(L.NSTASH, 0, -1):
                c, a, ap = sets(n)
                P.label("Lfs%d_%%=" % n)
                P.raw("s_waitcnt lgkmcnt(0)")
                Vn, ln = VR[(n - 1) % 2], LN[(n - 1) % 2]
                lam = [lambda j=j: P.fmac_dpp(ln, TAUR[(n - 1) % 3], Vn[j], j) for j in range(nx)]   # lambda_{t+1} += V x_{t+1}
                if n >= 2:
                    data_wait(n)
                    fill = [lambda m=n - 2: prefetch_a(m), lambda m=n - 2: (adj_prefetch_b(m), load_data(m))]
                else:
                    fill = []
                fcompute(c, a, ap, S_SM, False, fill + lam)
                outputs(n, False)
            n_fwd = P.n_instr - n_fwd0
            n_fwd_steps = L.NSTASH
            c, a, ap = sets(0)
            P.raw("s_waitcnt lgkmcnt(0)")
            fcompute(c, a, ap, S_UM, True)
            outputs(0, True)
            P.v("v_mov_b32_e32 %%[xvout], %s" % ACC[a])
            P.raw("s_branch Ldone_%=")
            P.lines.append(".p2align %d" % (FSTUB.bit_length() - 1))
            P.label("Lfstub_%=")
            for q in stubs:
                P.lines.append(".p2align %d" % (FSTUB.bit_length() - 1))
                P.lines += q.lines
                P.n_instr += q.n_instr
        else:
            FSTUB = 64
            while FSTUB < 8 * (2 * (2 * L.H + ns // 2 + 6) + 4):
                FSTUB *= 2
            lo = int(S_JMP[2:S_JMP.index(":")])
            P.raw("s_getpc_b64 " + S_JMP)
            P.label("Lpcf_%=", reset=False)
            P.raw("s_sub_i32 %s, %%[T], 2" % S_TMP)                 # entry stub index: (T-1) - 1
            P.raw("s_mul_i32 %s, %s, %d" % (S_TMP, S_TMP, FSTUB))
            P.raw("s_add_u32 s%d, s%d, Lfstub_%%=-Lpcf_%%=" % (lo, lo))
            P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
            P.raw("s_add_u32 s%d, s%d, %s" % (lo, lo, S_TMP))
            P.raw("s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1))
            P.raw("s_setpc_b64 " + S_JMP)
            n_fwd0 = P.n_instr
            in_loop[0] = True
            for n in range(L.NSTASH, 0, -1):
                c, a, ap = sets(n)
                P.label("Lfs%d_%%=" % n)
                P.raw("s_waitcnt lgkmcnt(%d)" % group_size(n - 1))   # rows of step n are in; those of n-1 may be in flight
                fill = []
                if n >= 2:
                    fill = [lambda m=n - 2: prefetch_a(m), lambda m=n - 2: prefetch_b(m)]
                    if nu == 1:
                        fill = [lambda m=n - 2: prefetch(m)]
                fcompute(c, a, ap, S_SM, False, fill)
            n_fwd = P.n_instr - n_fwd0
            n_fwd_steps = L.NSTASH
            in_loop[0] = False
            c, a, ap = sets(0)
            P.raw("s_waitcnt lgkmcnt(0)")
            fcompute(c, a, ap, S_UM, True)
            P.v("v_mov_b32_e32 %%[xvout], %s" % ACC[a])
            P.raw("s_branch Ldone_%=")
            # ---- entry stubs: stage F of the first two steps, read their rows, place x_init
            P.lines.append(".p2align %d" % (FSTUB.bit_length() - 1))
            P.label("Lfstub_%=")
            for n in range(1, L.NSTASH + 1):
                c, a, ap = sets(n)
                P.lines.append(".p2align %d" % (FSTUB.bit_length() - 1))
                prefetch(n, move=False)
                prefetch(n - 1)
                P.v("v_mov_b32_e32 %s, %s" % (ACC[ap], XV0), writes=(ACC[ap],))
                P.raw("s_branch Lfs%d_%%=" % n)
        # ---- backward stubs: F block of ring slot (n % 3) -> stash slot n-1
        P.lines.append(".p2align 5")
        P.label("Lbstub_%=")
        for n in range(1, L.NSTASH + 1):
            P.lines.append(".p2align 5")
            for p, (w, off) in enumerate(L.stash_pieces):
                regs = stash_regs_of(p, n - 1)
                if w == 2:
                    P.raw("ds_read2_b32 a[%d:%d], %%[sr%d] offset0:%d offset1:%d" % (regs[0], regs[1], n % D, off, off + 1))
                else:
                    P.raw("ds_read_b32 a%d, %%[sr%d] offset:%d" % (regs[0], n % D, off * 4))
            if X_RET_DIRECT:   # stub n is called from step n-1: the peeled first step, then body positions 1 .. LC, 1, ...
                P.raw("s_branch Lret%d_%%=" % (bret["first"] if n == 1 else bret[(n - 2) % LC + 1]))
            else:
                P.raw("s_setpc_b64 " + S_RET)
    P.label("Ldone_%=")
    P.v("v_mov_b32_e32 %%[minpiv], %s" % MINPIV)
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    stamp(3)
    for i in range(len(TS)):
        P.v("v_mov_b32_e32 %%[ts%d], %s" % (i, TS[i]))

    # ---- operand lists
    outs = [("xvout", '"=&v"(xvout)'), ("minpiv", '"=&v"(minpiv)')]
    if mpc:
        outs += [("nqp", '"=&v"(nqp)'), ("qpinfo", '"=&v"(qpinfo)')]
    if TS:
        outs += [("ts%d" % i, '"=&v"(in.ts[%d])' % i) for i in range(4)]
    rw = []
    for q in range(L.ndma_b):
        rw.append(("ptr%d" % q, '"+v"(in.ptr[%d])' % q))
    NDA = (nx * (nx + 1) + ns + 63) // 64          # adj: DMAs per group of [Vv | x | u]
    NST = (L.nchunk_b + 63) // 64                  # adj: store instructions per step
    assert NDA <= 2
    if not stash or adj:
        for q in range(NDA if adj else L.ndma_f):
            rw.append(("fptr%d" % q, '"+v"(in.fptr[%d])' % q))
    if adj:
        rw.append(("avp", '"+v"(in.avp)'))
        for q in range(NST):
            rw.append(("pso%d" % q, '"+v"(in.pso[%d])' % q))
    rw += [("tf", '"+s"(in.tf)'), ("gz", '"+v"(in.gz)'), ("ak", '"+v"(in.ak)'), ("arow", '"+v"(in.arow)'), ("aaff", '"+v"(in.aaff)'), ("pst", '"+v"(in.pst)')]
    if write_k:
        for m in range(nu):
            rw.append(("pk%d" % m, '"+v"(in.pk[%d])' % m))
    if ghbm:
        rw.append(("pgw", '"+v"(in.pgw)'))
    if masked:
        rw.append(("pm", '"+v"(in.pm)'))
    if save:
        for q in range(SV_NST):
            rw.append(("pso%d" % q, '"+v"(in.pso[%d])' % q))
    ins = []
    for q in range(L.ndma_b):
        ins.append(("str1_%d" % q, '"v"(in.str1[%d])' % q))
        ins.append(("str%d" % q, '"v"(in.str[%d])' % q))
    for i in range(4 if affine else ns):
        ins.append(("aq%d" % i, '"v"(in.aq[%d])' % i))
    for k in range(nx):
        ins.append(("af%d" % k, '"v"(in.af[%d])' % k))
    ins += [("eaff", '"v"(in.eaff)'), ("dst", '"v"(in.dst)'), ("pxi", '"v"(in.pxi)'), ("px0", '"v"(in.px0)')]
    if stash:
        for q in range(L.NFD):
            ins.append(("fp%d" % q, '"v"(in.fp[%d])' % q))
        for q in range(D):
            ins.append(("sr%d" % q, '"v"(in.sr[%d])' % q))
        ins += [("drowo", '"v"(in.drow)'), ("daff", '"v"(in.daff)'),
                ("farea", '"s"(in.farea)')]
    else:
        for q in range(L.ndma_f):
            ins.append(("fstr%d" % q, '"v"(in.fstr[%d])' % q))
        ins += [("drow", '"v"(in.drow)'), ("drow2", '"v"(in.drow2)'), ("daff", '"v"(in.daff)'), ("daff2", '"v"(in.daff2)')]
    if adj:
        for q in range(NDA):
            ins.append(("fstr%d" % q, '"v"(in.fstr[%d])' % q))
        for q in range(NST):
            ins.append(("sso%d" % q, '"v"(in.sso[%d])' % q))
        for q in range(4):
            ins.append(("avr%d" % q, '"v"(in.avr[%d])' % q))
        ins += [("atx", '"v"(in.atx)'), ("awc", '"v"(in.awc)'), ("awe", '"v"(in.awe)'), ("awf", '"v"(in.awf)'),
                ("awd", '"v"(in.awd)'), ("ach", '"v"(in.ach)'), ("wa", '"v"(in.wa)'), ("wb", '"v"(in.wb)'),
                ("dfshift", '"v"(in.dfshift)')]
    if write_k:
        ins.append(("dk", '"v"(in.dk)'))
    if ghbm:
        ins.append(("dgw", '"v"(in.dgw)'))
        for q in range(L.ndma_f):
            ins.append(("fstrl%d" % q, '"v"(in.fstrl[%d])' % q))
    if save:
        for q in range(SV_NST):
            ins.append(("sso%d" % q, '"v"(in.sso[%d])' % q))
        ins += [("asv", '"v"(in.asv)'), ("asq", '"v"(in.asq)'), ("asu", '"v"(in.asu)'), ("ach", '"v"(in.ach)')]
    if masked:
        ins += [("dm", '"v"(in.dm)'), ("am", '"v"(in.am)')]
    ins += [("ring", '"s"(in.ring)'), ("T", '"s"(in.T)'), ("nz", '"s"(in.nz)'), ("bwd_only", '"v"(in.bwd_only)')]
    if mpc:
        ins.append(("nqp_iter", '"s"(in.n_qp_iter)'))
    if expand:
        ins += [("atau", '"v"(in.atau)'), ("act", '"v"(in.act)')]
    clob = ['"v%d"' % i for i in range(VBASE, (255 if adj else last_vgpr) + 1)] + ['"a%d"' % i for i in range(n_agpr)] + \
        ['"s%d"' % i for i in ([70] + list(range(72, 102 if (mpc or affine) else (100 if (save or ghbm) else 98))))] + ['"vcc"', '"scc"', '"memory"']

    tf = lambda b: "true" if b else "false"
    name = "LqrAsm<%d, %d, %s, %s, %s, %s, %s, %s, %s>" % (nx, nu, tf(write_k), tf(stash), tf(masked), tf(ghbm), tf(save), tf(affine), tf(adj))
    if mpc:
        name = "MpcAsm<%d, %d, %s>" % (nx, nu, tf(expand))
    o = []
    o.append("// (%d,%d) write_k=%d stash=%d masked=%d: %d instructions in prologue + 4 backward steps, %d in %d unrolled forward steps\n"
             % (nx, nu, write_k, stash, masked, n_bwd, n_fwd, n_fwd_steps))
    o.append("template <>\nstruct %s {\n" % name)
    o.append("  static constexpr bool kAvailable = true;\n")
    o.append("  static constexpr int NDB = %d, NDF = %d, SLOT_B = %d, SLOT_F = %d, RING_BYTES = %d, KROW = %d, DEPTH_F = %d, DEPTH_B = %d;\n"
             % (L.ndma_b, L.ndma_f, L.SLOT_B, L.SLOT_F, L.RING, KROW, DF, D))
    o.append("  static constexpr int OFF_C = %d, OFF_c = %d, OFF_F = %d, OFF_f = %d, FOFF_f = %d, FOFF_G = %d;\n"
             % (L.OFF_C, L.OFF_c, L.OFF_F, L.OFF_f, L.FOFF_f, L.FOFF_G))
    o.append("  static constexpr int NSTASH = %d, NFD = %d, FAREA_BYTES = %d, HROW = %d, SPD = %d, PADM = %d;\n"
             % (L.NSTASH, L.NFD, L.FAREA, L.H, L.SPD, 16 * L.nchunk_b))
    if adj:
        o.append("  static constexpr int ADJ_NDA = %d, ADJ_SLOT = %d, ADJ_NST = %d;\n" % (NDA, NDA * 1024, NST))
    if save:
        o.append("  static constexpr int SAVE_STAGE_BYTES = %d, SAVE_NST = %d;\n" % (16 * SV_N, SV_NST))
    # block 1: the first DB groups
    rw1 = [("ptr%d" % q, '"+v"(in.ptr[%d])' % q) for q in range(L.ndma_b)] + [("tf", '"+s"(in.tf)')]
    if masked:
        rw1.append(("pm", '"+v"(in.pm)'))
    ins1 = []
    for q in range(L.ndma_b):
        ins1.append(("str1_%d" % q, '"v"(in.str1[%d])' % q))
        ins1.append(("str%d" % q, '"v"(in.str[%d])' % q))
    if masked:
        ins1.append(("dm", '"v"(in.dm)'))
    ins1 += [("ring", '"s"(in.ring)'), ("T", '"s"(in.T)')]
    o.append("  static __device__ __forceinline__ void issue_first(LqrAsmIn<%d, %d> &in) {\n" % (nx, nu))
    o.append("    asm volatile(\n")
    for ln in P_first.text():
        o.append('        "%s\\n\\t"\n' % ln)
    o.append("        : " + ", ".join("[%s] %s" % x for x in rw1) + "\n")
    o.append("        : " + ", ".join("[%s] %s" % x for x in ins1) + "\n")
    o.append('        : "scc", "memory");\n')
    o.append("  }\n")
    o.append("  static __device__ __forceinline__ void run(LqrAsmIn<%d, %d> &in, float &xvout, float &minpiv%s) {\n"
             % (nx, nu, ", int &nqp, int &qpinfo" if mpc else ""))
    o.append("    asm volatile(\n")
    for ln in P.text():
        o.append('        "%s\\n\\t"\n' % ln)
    o.append("        : " + ", ".join("[%s] %s" % x for x in outs + rw) + "\n")
    o.append("        : " + ", ".join("[%s] %s" % x for x in ins) + "\n")
    o.append("        : " + ", ".join(clob) + ");\n")
    o.append("  }\n};\n\n")
    return "".join(o)


HEADER = """// lqr_asm_gen.hpp - GENERATED by gen_lqr_asm.py; do not edit.
// Whole-kernel gfx950 instruction streams of the fused LQR solve (lqr/lqr_recursion.py:69-209 of the reference),
// 16-lane row layout, one per (nx, nu, write_k, stash).  The C++ side that prepares the per-lane operands is
// lqr_asm_kernel.hpp.
#pragma once
#include <cstdint>

namespace dmpc {

// per-lane operands of the instruction stream (see lqr_asm_kernel.hpp for how they are filled)
template <int NX, int NU>
struct LqrAsmIn {
  static constexpr int NS = NX + NU;
  // backward sweep
  uint64_t ptr[4], str1[4], str[4];  // LDS-DMA source of this lane's chunk (t = T-1), first / later time strides
  unsigned aq[NS], af[NX];           // LDS byte addresses (ring slot 0) of this lane's [C|c] rows and [F|f] rows
  unsigned ak;                       // LDS byte address of gain row (T-1, 0), this lane's column
  unsigned gz;                       // LDS byte address of this wave's gain rows + lane64 * 16 (zero fill)
  int nz;                            // wave-uniform: 1 KB pieces of the gain rows of one wave
  int tf;                            // wave-uniform: time strides the DMA pointers may still take (set by issue_first)
  int bwd_only;                      // 1 = stop after the backward sweep (LqrRecursion.backward()); same in every lane
  float eaff;                        // 1 in lane `aff`, else 0
  uint64_t pk[NU], dk;               // Ks/ks store pointers (t = T-1) and their time stride (write_k)
  uint64_t pgw, dgw;                 // ghbm: store pointer of column min(lane, ns) of gain row (T-1, 0) in the workspace, time stride
  // save: [Vv | Qxu | Quu] leave through a staging area (LDS byte addresses of: column min(lane, nx) of row 0 of this
  // trajectory's [V | v] - lane ns is column nx -, column lane - nx of row 0 of Qxu, Quu) as 16-byte chunks: pso / sso / ach
  unsigned asv, asq, asu;
  uint64_t pm, dm;                   // masked: DMA source of this lane's dword of clamped-control flags, time stride
  unsigned am;                       // masked: LDS byte address (ring slot 0, without the padding offset) of this row's flags
  int n_qp_iter;                     // mpc (wave-uniform): iteration cap of the box QP
  unsigned act;                      // mpc, expand: LDS byte address (ring slot 0) of row min(lane, ns-1) of this trajectory's C_t
  unsigned atau;                     // mpc, expand: LDS byte address (ring slot 0, without the padding offset) of [x_t; u_t][lane]
  // forward sweep
  uint64_t fptr[2], fstr[2];         // ring variant: DMA source of this lane's [F|f] chunk (t = 0) and time stride
  uint64_t fstrl[2];                 // ghbm: stride of the LAST step (T-2 -> T-1): that of the gain lanes, 0 for F / f
  uint64_t fp[8];                    // stash variant: DMA sources of all of f (issued in the prologue)
  unsigned sr[6];                    // stash variant: LDS byte address of this lane's half row of F in ring slot q < DEPTH_B
  unsigned farea;                    // stash variant (wave-uniform): LDS byte address of this wave's f area
  unsigned arow, aaff, drow, drow2, daff, daff2;
  uint64_t pst, dst;                 // [x_{t+1} | u_t] store pointer and time stride
  uint64_t pxi, px0;                 // &x_init[b][lane] (every lane valid), &x[0][b][lane] (lanes < nx store)
  // wave-uniform
  unsigned ring;                     // LDS byte address of this wave's ring
  int T;
  unsigned ts[4];                    // GEN_TIMING builds only: s_memtime at the phase boundaries
  // adj (the one-pass gradient; fptr / fstr carry the DMA sources of [Vv | x | u], walking forward in time)
  unsigned avp;                      // LDS byte address of v'_{T-1}[lane] in the f area (lanes < nx), moves back per step
  unsigned avr[4];                   // LDS byte address of row min(lane, nx-1) of this trajectory's [V | v] in ring slot q
  unsigned atx;                      // ... of [x; u][min(lane, ns-1)] in ring slot 0
  unsigned awc, awe, awf, awd;       // staging area (without its ring offset): column `lane` of row 0 of dC / dc / dF, df[lane]
  unsigned ach;                      // staging area + lane64 * 16: this lane's chunk (adj: without the area's ring offset)
  uint64_t pso[4], sso[4];           // store pointer of this lane's chunk (adj: of [dC | dc | dF | df] at t = 0), time stride
  float wa, wb;                      // dC = wa dtau (x) tau + wb tau (x) dtau
  int dfshift;                       // 0: df[t] = d_lambda[t] (the reference), 1: d_lambda[t+1]
};

// GHBM (ring form, no gains out): the gain rows pass through a caller workspace in HBM instead of LDS - any horizon
// SAVE (with WRITE_K): Quu_t and Qxu_t of every step go to HBM as well (DiffLqr's training form)
// AFFINE (STASH, no gains out): the re-solve with saved K_t, Quu_t, Qxu_t and another c (DiffLqr.backward's second solve)
// ADJ (with AFFINE): DiffLqr.backward in one launch - the affine re-solve whose rollout writes dC, dc, dF, df, dx_init from
// [V_t | v_t] of the saving solve instead of C (see gen_kernel)
template <int NX, int NU, bool WRITE_K, bool STASH, bool MASKED = false, bool GHBM = false, bool SAVE = false, bool AFFINE = false,
          bool ADJ = false>
struct LqrAsm {
  static constexpr bool kAvailable = false;
};

// MPCstep.backward_rec: backward sweep with the box QP in the stream (gains to HBM, no rollout); EXPAND: with the
// Taylor re-centring of c (need_expand) in the sweep
template <int NX, int NU, bool EXPAND>
struct MpcAsm {
  static constexpr bool kAvailable = false;
};

"""


def main():
    import sys
    global OUT
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        OUT = sys.argv[2]
    out = [HEADER.replace("#pragma once\n", "#pragma once\n#define DMPC_ASM_TIMING_GEN 1\n") if X_TIMING else HEADER]
    for nx, nu in SHAPES:
        for write_k in (False, True):
            for stash in (False, True):
                L0 = Layout(nx, nu)
                if stash and not L0.stash_ok:
                    continue
                out.append(gen_kernel(nx, nu, write_k, stash))
                if write_k and nu in (1, 2):
                    out.append(gen_kernel(nx, nu, write_k, stash, save=True))
                if stash and not write_k and nx + nu >= 4 and nu in (1, 2):
                    out.append(gen_kernel(nx, nu, write_k, stash, affine=True))
                    out.append(gen_kernel(nx, nu, write_k, stash, affine=True, adj=True))
                if not stash and not write_k:
                    out.append(gen_kernel(nx, nu, write_k, stash, ghbm=True))
                if nu not in (1, 2):     # three and four controls: the plain, gains-out and any-horizon forms (the masked, MPC
                    continue             # and training forms spell their pivoted solves out for 1 x 1 and 2 x 2)
                if not write_k and L0.SLOT_B - 16 * L0.nchunk_b >= 256:   # room for the flag dwords in the slot padding
                    out.append(gen_kernel(nx, nu, write_k, stash, masked=True))
                if write_k and not stash and L0.SLOT_B - 16 * L0.nchunk_b >= 256:
                    out.append(gen_kernel(nx, nu, True, False, masked=True, mpc=True))
                    if 12 * nu + 4 * nx <= 64:
                        out.append(gen_kernel(nx, nu, True, False, masked=True, mpc=True, expand=True))
    out.append("}  // namespace dmpc\n")
    with open(OUT, "w") as fh:
        fh.write("".join(out))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
