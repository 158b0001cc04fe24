#!/usr/bin/env python
"""Emit mpc_fwd_asm_gen.hpp: MPCstep.forward_rec (mpc/mpc_step.py:175-286) for a LinDx and a QuadCost as ONE gfx950
instruction stream per (nx, nu) - the clamped closed-loop rollout under the TRUE dynamics with the per-trajectory
backtracking line search on the TRUE cost, in the 16-lane row layout of the LQR stream's forward sweep.

Layout.  A wavefront owns four consecutive trajectories, one 16-lane DPP row each.  Lane i < nx is state row i, lane
nx + m control row m (lanes >= ns read zeros and stay zero).  Per timestep a lane holds ONE row of [F_t | f_t] (state
lanes) or [K_t | k_t] (control lanes) and one row of [C_t | c_t]; [x_t; u_t] lives element-per-lane in one register, so

    x_{t+1}[i] = f_i + sum_j F[i][j] tau_j          u_t[m] = clamp(u^_m + alpha k_m + sum_j K[m][j] (x_j - x^_j))
    (C tau)[i],  (C d)[i]   with d = tau - tau^

are chains of DPP broadcast-FMAs (row_newbcast:j supplies element j to every lane of the row), 3 nx + 2 ns + nu of them.
The inputs of a step - C, c, F, f, K, k, u^, lower, upper, x^ of the four trajectories, one contiguous run per array -
arrive by per-lane gather LDS-DMA (kDma instructions) in a ring of DB slots, DB steps ahead; two register sets
alternate, the next step's rows are read from LDS while this step computes.

The line search (mpc_step.py:196-268) is a loop over passes that runs while ANY of the wave's four trajectories still
searches; a trajectory that is done neither stores nor commits (exec / select masks per row).  The decision
`cost > OLD_COST` is taken on the difference summed per timestep, obj(tau) - obj(tau^) = 1/2 d'(C tau) + 1/2 tau^'(C d) + c'd
(see mpc_kernels.hpp), accumulated per lane and reduced over the row once per pass.

    python chainer_differentiable_mpc_amd/csrc/gen_mpc_fwd_asm.py     # rewrites mpc_fwd_asm_gen.hpp
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_lqr_asm import Prog, Regs, VBASE   # noqa: E402  (emitter with the hazard tracker)

OUT = os.path.join(HERE, "mpc_fwd_asm_gen.hpp")
SHAPES = [(8, 2), (3, 1), (4, 2), (6, 2), (2, 2), (1, 1), (2, 1), (3, 2)]
DB = 6        # ring depth == loop unroll (two alternating register sets)
# timing experiments only (the results of such builds are wrong on purpose)
X_NO_STORE = os.environ.get("GEN_FWD_NO_STORE") == "1"
X_NO_DMA = os.environ.get("GEN_FWD_NO_DMA") == "1"
X_NO_LDS = os.environ.get("GEN_FWD_NO_LDS") == "1"
X_NO_COST = os.environ.get("GEN_FWD_NO_COST") == "1"
X_NO_ADV = os.environ.get("GEN_FWD_NO_ADV") == "1"
# later passes fetch only the chunks of the trajectories that still search (round 4; GEN_FWD_ROW_MASK=0: every pass fetches all)
X_ROW_MASK = os.environ.get("GEN_FWD_ROW_MASK", "1") == "1"
X_NT = os.environ.get("GEN_FWD_NT") == "1"      # nt cache policy on the input DMAs (measured, not a default: the search re-reads C and F)


class FwdLayout:
    def __init__(self, nx, nu):
        ns = nx + nu
        self.nx, self.nu, self.ns = nx, nu, ns
        ch = 0
        self.CH = {}
        for name, n in (("C", ns * ns), ("c", ns), ("F", nx * ns), ("f", nx), ("K", nu * nx), ("k", nu), ("u", nu),
                        ("lo", nu), ("hi", nu), ("x", nx)):
            self.CH[name] = ch
            ch += n
        self.END = ch
        self.kDma = (ch + 63) // 64
        self.SLOT_B = self.kDma * 1024
        self.ZERO = ch * 16                      # byte offset of the slot's padding: the padding lanes fetch zeros
        assert self.kDma <= 4, "one M0 per group"
        assert self.SLOT_B - self.ZERO >= 4 * ns + 16, "padding too small for the zero reads"
        assert (DB - 1) * self.kDma <= 63
        self.wide = nx % 2 == 0 and ns % 2 == 0  # rows are 8-byte aligned: ds_read_b64


def v2(pair):
    a = int(pair[0][1:])
    return "v[%d:%d]" % (a, a + 1)


def gen_fwd(nx, nu):
    L = FwdLayout(nx, nu)
    ns, KD = L.ns, L.kDma
    P = Prog()
    R = Regs(VBASE)
    ptr0 = ["%%[ptr0_%d]" % q for q in range(KD)]
    strd = ["%%[str%d]" % q for q in range(KD)]
    strl = ["%%[strl%d]" % q for q in range(KD)]

    PTR = [R.take(2, align=2) for _ in range(KD)]
    PST, PUF, POBJ = R.take(2, align=2), R.take(2, align=2), R.take(2, align=2)
    sets = []
    for _ in range(2):
        sets.append(dict(ROW=R.take(ns, align=2), AFF=R.take(1)[0], CROW=R.take(ns, align=2), CAFF=R.take(1)[0],
                         HAT=R.take(1)[0], LB=R.take(1)[0], UB=R.take(1)[0]))
    XA, XB, DX, A2, U, TAU, D, Q1, QD, LIN, T1, T2, OBJ = R.take(13)
    COSTP, DELTA, OLDP, COST, OLD, ALPHA, SC, NLS, WF, CE8 = R.take(10)
    last_vgpr = R.next - 1
    assert last_vgpr <= 255

    S_N, S_TI, S_PASS, S_TMP = "s70", "s88", "s89", "s90"
    S_CAP, S_WOBJ, S_UFW = "s91", "s92", "s93"
    S_XM, S_UM, S_SM, S_SRCH, S_ST, S_W, S_T0, S_ROW0 = ("s[72:73]", "s[74:75]", "s[76:77]", "s[78:79]", "s[80:81]",
                                                         "s[82:83]", "s[84:85]", "s[86:87]")
    # exec masks of the DMA instructions during a pass (S_W and S_T0 are only used between passes)
    S_DM = ["s[82:83]", "s[84:85]", "s[96:97]", "s[94:95]"]
    assert KD <= len(S_DM)

    def mask64(lanes):
        m16 = sum(1 << l for l in lanes)
        return m16 | (m16 << 16)

    uniq = [0]
    in_body = [False]

    def issue_group(slot):
        if slot == 0:
            P.raw("s_mov_b32 m0, %[ring]")
        else:
            P.raw("s_add_u32 m0, %%[ring], %d" % (slot * L.SLOT_B))
        P.raw("s_waitcnt lgkmcnt(0)")          # the wait state an LDS-DMA needs after the M0 write; the slot's reads are in
        for q in range(KD):
            off = (" offset:%d" % (q * 1024)) if q else ""
            if not (X_NO_DMA and in_body[0]):
                if X_ROW_MASK:   # only the chunks a searching trajectory needs (lane 0 always: the instruction is never empty,
                    P.raw("s_mov_b64 exec, %s" % S_DM[q])      # the counted waits stay exact)
                    P.raw("s_nop 0")
                P.raw("global_load_lds_dwordx4 %s, off%s%s" % (v2(PTR[q]), off, " nt" if X_NT else ""))
        if X_ROW_MASK and not (X_NO_DMA and in_body[0]):
            P.raw("s_mov_b64 exec, -1")
            P.exec_written()

    slow = []   # (label, return label): the one advance of a pass that must leave the F / f lanes where they are

    def advance():
        """move the DMA pointers one timestep on while there is one: S_TI counts the advances left; the LAST one (to
        t = T-1) uses the strides that keep the F / f lanes on slice T-2 (there is no F_{T-1}) - out of line"""
        if X_NO_ADV and in_body[0]:
            return
        uniq[0] += 1
        n_ = uniq[0]
        P.raw("s_cmp_lt_i32 %s, 2" % S_TI)
        P.raw("s_cbranch_scc1 Lslow%d_%%=" % n_)
        for q in range(KD):
            P.v("v_lshl_add_u64 %s, %s, 0, %s" % (v2(PTR[q]), v2(PTR[q]), strd[q]))
        P.raw("s_sub_i32 %s, %s, 1" % (S_TI, S_TI))
        P.label("Ladv%d_%%=" % n_, reset=False)   # only pointer registers and S_TI differ between the paths
        slow.append(n_)

    def emit_slow_paths():
        for n_ in slow:
            P.label("Lslow%d_%%=" % n_, reset=False)
            P.raw("s_cmp_lt_i32 %s, 1" % S_TI)
            P.raw("s_cbranch_scc1 Ladv%d_%%=" % n_)
            for q in range(KD):
                P.v("v_lshl_add_u64 %s, %s, 0, %s" % (v2(PTR[q]), v2(PTR[q]), strl[q]))
            P.raw("s_mov_b32 %s, 0" % S_TI)
            P.raw("s_branch Ladv%d_%%=" % n_)

    def read_set(c, slot):
        if X_NO_LDS and in_body[0]:
            return
        S = sets[c]
        off = slot * L.SLOT_B
        if L.wide:
            for j in range(0, nx, 2):
                P.raw("ds_read_b64 %s, %%[a_row] offset:%d" % (v2(S["ROW"][j:j + 2]), off + j * 4))
            for j in range(nx, ns, 2):
                P.raw("ds_read_b64 %s, %%[a_row2] offset:%d" % (v2(S["ROW"][j:j + 2]), off + (j - nx) * 4))
            for j in range(0, ns, 2):
                P.raw("ds_read_b64 %s, %%[a_crow] offset:%d" % (v2(S["CROW"][j:j + 2]), off + j * 4))
        else:
            for j in range(nx):
                P.raw("ds_read_b32 %s, %%[a_row] offset:%d" % (S["ROW"][j], off + j * 4))
            for j in range(nx, ns):
                P.raw("ds_read_b32 %s, %%[a_row2] offset:%d" % (S["ROW"][j], off + (j - nx) * 4))
            for j in range(ns):
                P.raw("ds_read_b32 %s, %%[a_crow] offset:%d" % (S["CROW"][j], off + j * 4))
        P.raw("ds_read_b32 %s, %%[a_aff] offset:%d" % (S["AFF"], off))
        P.raw("ds_read_b32 %s, %%[a_caff] offset:%d" % (S["CAFF"], off))
        P.raw("ds_read_b32 %s, %%[a_hat] offset:%d" % (S["HAT"], off))
        P.raw("ds_read_b32 %s, %%[a_lb] offset:%d" % (S["LB"], off))
        P.raw("ds_read_b32 %s, %%[a_lb] offset:%d" % (S["UB"], off + 16 * nu))   # upper sits 4 nu floats behind lower

    def snap(bound, toward_upper):
        """U = bound where U is within bound_tol(bound) of it (mpc_kernels.hpp: 1e-8 + 4 eps max(1, |bound|))"""
        if toward_upper:
            P.v("v_sub_f32_e32 %s, %s, %s" % (T1, bound, U), writes=(T1,), reads=(bound, U))
        else:
            P.v("v_sub_f32_e32 %s, %s, %s" % (T1, U, bound), writes=(T1,), reads=(bound, U))
        P.v("v_max_f32_e64 %s, 1.0, |%s|" % (T2, bound), writes=(T2,), reads=(bound,))
        P.v("v_fmamk_f32 %s, %s, 0x35000000, %s" % (T2, T2, CE8), writes=(T2,), reads=(T2, CE8))   # 4 * 2^-23
        P.v("v_cmp_le_f32_e32 vcc, %s, %s" % (T1, T2), reads=(T1, T2))
        P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (U, U, bound), writes=(U,), reads=(U, bound))

    def fstep(c, X, XN, j, objs, uf):
        """one timestep from register set c: X holds [x_t | -] element per lane, XN receives [x_{t+1} | -].
        objs / uf: this pass stores the per-step objective / the first pass's controls (known per pass: the body exists
        once per combination, so that the counted wait below is exact - vmcnt counts the stores too, and a wait that
        ignores them leaves the DMAs ~3 instead of DB - 1 steps of flight: the kernel then runs at the memory latency)"""
        S = sets[c]
        ROW, CROW, HAT = S["ROW"], S["CROW"], S["HAT"]
        o = 1 - c
        issue_group(j)                       # slot j went to registers one step ago: refill it, DB steps ahead
        advance()
        n_st = 1 + (1 if objs else 0) + (1 if uf else 0)
        if not X_NO_DMA:
            P.raw("s_waitcnt vmcnt(%d)" % min(63, (DB - 1) * (KD + n_st)))
        read_set(o, (j + 1) % DB)
        # controls: u = clamp(u^ + alpha k + K (x - x^))                                           mpc_step.py:209-221
        P.v("v_sub_f32_e32 %s, %s, %s" % (DX, X, HAT), writes=(DX,), reads=(X, HAT))
        P.v("v_mul_f32_e32 %s, %s, %s" % (A2, S["AFF"], SC), writes=(A2,), reads=(S["AFF"], SC))   # f_i | alpha k_m
        P.v("v_mov_b32_e32 %s, %s" % (XN, A2), writes=(XN,), reads=(A2,))
        for jj in range(nx):
            P.fmac_dpp(A2, DX, ROW[jj], jj)
            P.fmac_dpp(XN, X, ROW[jj], jj)                                                          # :229-236, x part
        P.v("v_add_f32_e32 %s, %s, %s" % (U, A2, HAT), writes=(U,), reads=(A2, HAT))
        P.v("v_max_f32_e32 %s, %s, %s" % (U, U, S["LB"]), writes=(U,), reads=(U, S["LB"]))
        P.v("v_min_f32_e32 %s, %s, %s" % (U, U, S["UB"]), writes=(U,), reads=(U, S["UB"]))
        snap(S["LB"], False)
        snap(S["UB"], True)
        P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (TAU, X, U, S_UM), writes=(TAU,), reads=(X, U))   # [x_t ; u_t]
        for m in range(nu):
            P.fmac_dpp(XN, TAU, ROW[nx + m], nx + m)                                                # u part
        # cost of the step and its difference to the iterate's                                      :246-251, util.py:162-198
        if X_NO_COST:
            ns_cost = 0
        else:
            ns_cost = ns
        P.v("v_sub_f32_e32 %s, %s, %s" % (D, TAU, HAT), writes=(D,), reads=(TAU, HAT))
        P.mul_dpp(Q1, TAU, CROW[0], 0)
        P.mul_dpp(QD, D, CROW[0], 0)
        for jj in range(1, ns_cost):     # two independent chains, interleaved
            P.fmac_dpp(Q1, TAU, CROW[jj], jj)
            P.fmac_dpp(QD, D, CROW[jj], jj)
        P.v("v_fma_f32 %s, 0.5, %s, %s" % (LIN, Q1, S["CAFF"]), writes=(LIN,), reads=(Q1, S["CAFF"]))
        P.v("v_mul_f32_e32 %s, %s, %s" % (OBJ, TAU, LIN), writes=(OBJ,), reads=(TAU, LIN))
        P.v("v_add_f32_e32 %s, %s, %s" % (COSTP, COSTP, OBJ), writes=(COSTP,), reads=(COSTP, OBJ))
        P.v("v_mul_f32_e32 %s, 0.5, %s" % (T1, HAT), writes=(T1,), reads=(HAT,))
        P.v("v_mul_f32_e32 %s, %s, %s" % (T1, T1, QD), writes=(T1,), reads=(T1, QD))
        P.v("v_fmac_f32_e32 %s, %s, %s" % (T1, D, LIN), writes=(T1,), reads=(D, LIN, T1))
        P.v("v_add_f32_e32 %s, %s, %s" % (DELTA, DELTA, T1), writes=(DELTA,), reads=(DELTA, T1))
        P.v("v_sub_f32_e32 %s, %s, %s" % (T2, Q1, QD), writes=(T2,), reads=(Q1, QD))                # C tau^ = C tau - C d   :191
        P.v("v_fma_f32 %s, 0.5, %s, %s" % (T2, T2, S["CAFF"]), writes=(T2,), reads=(T2, S["CAFF"]))
        P.v("v_fmac_f32_e32 %s, %s, %s" % (OLDP, HAT, T2), writes=(OLDP,), reads=(HAT, T2, OLDP))
        # outputs of the trajectories that still search (the last pass that writes is the accepted one)
        if objs:
            for rot in (8, 4, 2, 1):
                P.valu("v_add_f32_dpp %s, %s, %s row_ror:%d row_mask:0xf bank_mask:0xf" % (OBJ, OBJ, OBJ, rot),
                       writes=(OBJ,), reads=(OBJ,), dpp=OBJ)
            P.raw("s_and_b64 exec, %s, %s" % (S_ROW0, S_SRCH))
            if not X_NO_STORE:
                P.raw("global_store_dword %s, %s, off" % (v2(POBJ), OBJ))
        P.raw("s_mov_b64 exec, %s" % S_ST)
        if not X_NO_STORE:
            P.raw("global_store_dword %s, %s, off" % (v2(PST), TAU))
        if uf:
            P.raw("s_mov_b64 exec, %s" % S_UM)                                                     # :260-263 (first pass)
            if not X_NO_STORE:
                P.raw("global_store_dword %s, %s, off" % (v2(PUF), TAU))
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
        if objs:
            P.v("v_lshl_add_u64 %s, %s, 0, %%[dobj]" % (v2(POBJ), v2(POBJ)))
        P.v("v_lshl_add_u64 %s, %s, 0, %%[dst]" % (v2(PST), v2(PST)))
        if uf:
            P.v("v_lshl_add_u64 %s, %s, 0, %%[dst]" % (v2(PUF), v2(PUF)))

    def row_sum(reg):
        for rot in (8, 4, 2, 1):
            P.valu("v_add_f32_dpp %s, %s, %s row_ror:%d row_mask:0xf bank_mask:0xf" % (reg, reg, reg, rot),
                   writes=(reg,), reads=(reg,), dpp=reg)

    # =============================================================== set-up
    for name, val in ((S_XM, mask64(range(nx))), (S_UM, mask64(range(nx, ns))), (S_SM, mask64(range(ns))),
                      (S_ROW0, mask64([0]))):
        lo = int(name[2:name.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo, val & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo + 1, val & 0xffffffff))
    # (hipcc gives an inline-asm block only a handful of "s" operands: the rest come as VGPRs, same in every lane)
    P.raw("v_readfirstlane_b32 %s, %%[cap]" % S_CAP)
    P.raw("v_readfirstlane_b32 %s, %%[want_objs]" % S_WOBJ)
    P.raw("v_readfirstlane_b32 %s, %%[uf_mask]" % S_UFW)
    P.raw("s_and_b32 %s, %s, 1" % (S_UFW, S_UFW))
    P.v("v_mov_b32_e32 %s, 1.0" % ALPHA, writes=(ALPHA,))
    for r_ in (COST, OLD, NLS, WF):
        P.v("v_mov_b32_e32 %s, 0" % r_, writes=(r_,))
    P.v("v_mov_b32_e32 %s, 0x322bcc77" % CE8, writes=(CE8,))       # 1e-8f
    P.raw("s_mov_b32 %s, 0" % S_PASS)
    P.raw("s_mov_b64 %s, -1" % S_SRCH)
    # =============================================================== one pass of the line search
    P.label("Lpass_%=")
    if X_ROW_MASK:
        # A later pass is run for the trajectories that still search; the others' chunks of the slot are not fetched again
        # (their lanes compute on whatever the ring holds - finite leftovers - and commit nothing): rb = the searching
        # rows as four bits, S_DM[q] = the lanes whose chunk of DMA q holds data of a searching row (a chunk can straddle rows)
        P.raw("s_and_b32 s94, s78, 1")
        P.raw("s_bfe_u32 s95, s78, 0x10010")
        P.raw("s_lshl_b32 s95, s95, 1")
        P.raw("s_or_b32 s94, s94, s95")
        P.raw("s_and_b32 s95, s79, 1")
        P.raw("s_lshl_b32 s95, s95, 2")
        P.raw("s_or_b32 s94, s94, s95")
        P.raw("s_bfe_u32 s95, s79, 0x10010")
        P.raw("s_lshl_b32 s95, s95, 3")
        P.raw("s_or_b32 s94, s94, s95")
        for q in range(KD):      # (the last one overwrites s[94:95]: rb has been consumed by then)
            P.v("v_bfe_u32 %s, %%[dmrow], %d, 4" % (T1, 4 * q), writes=(T1,))
            P.v("v_and_b32_e32 %s, s94, %s" % (T1, T1), writes=(T1,), reads=(T1,))
            P.v("v_cmp_ne_u32_e64 %s, 0, %s" % (S_DM[q], T1), reads=(T1,))
    for q in range(KD):
        a = int(PTR[q][0][1:])
        P.v("v_mov_b32_e32 v%d, %s" % (a, ptr0[q].replace("]", "_lo]")), writes=("v%d" % a,))
        P.v("v_mov_b32_e32 v%d, %s" % (a + 1, ptr0[q].replace("]", "_hi]")), writes=("v%d" % (a + 1),))
    for pair, name in ((PST, "pst0"), (PUF, "puf0"), (POBJ, "pobj0")):
        a = int(pair[0][1:])
        P.v("v_mov_b32_e32 v%d, %%[%s_lo]" % (a, name), writes=("v%d" % a,))
        P.v("v_mov_b32_e32 v%d, %%[%s_hi]" % (a + 1, name), writes=("v%d" % (a + 1),))
    for r_ in (COSTP, DELTA, OLDP):
        P.v("v_mov_b32_e32 %s, 0" % r_, writes=(r_,))
    P.v("v_cndmask_b32_e64 %s, 1.0, %s, %s" % (SC, ALPHA, S_UM), writes=(SC,), reads=(ALPHA,))    # 1 | alpha per lane
    P.raw("s_and_b64 %s, %s, %s" % (S_ST, S_SM, S_SRCH))
    P.raw("s_sub_i32 %s, %%[T], 1" % S_TI)
    P.raw("s_mov_b32 %s, %%[T]" % S_N)
    for j in range(DB):
        issue_group(j)
        advance()
    P.raw("s_waitcnt vmcnt(%d)" % ((DB - 1) * KD))
    read_set(0, 0)
    P.raw("s_waitcnt lgkmcnt(0)")
    P.v("v_mov_b32_e32 %s, %s" % (XA, sets[0]["HAT"]), writes=(XA,), reads=(sets[0]["HAT"],))       # new_x[0] = states[0]   :198
    # the body of the pass, once per combination of stores (see fstep)
    P.raw("s_cmp_lg_u32 %s, 0" % S_PASS)
    P.raw("s_cselect_b32 %s, 0, %s" % (S_TMP, S_UFW))          # u_first: first pass only, when wanted
    P.raw("s_lshl_b32 vcc_lo, %s, 1" % S_WOBJ)                # (vcc is free here)
    P.raw("s_or_b32 %s, %s, vcc_lo" % (S_TMP, S_TMP))
    for idx in (1, 2, 3):
        P.raw("s_cmp_eq_u32 %s, %d" % (S_TMP, idx))
        P.raw("s_cbranch_scc1 Lbody%d_%%=" % idx)
    for idx in (0, 1, 2, 3):
        objs, uf = bool(idx & 2), bool(idx & 1)
        P.label("Lbody%d_%%=" % idx)
        # The counted wait of fstep assumes (DB - 1) steps' worth of stores behind the DMA group it waits for; the
        # first steps of a pass have issued none yet, so that many stores are put into the queue here (x^_0 / u^_0 to
        # the slots the first step overwrites, searching trajectories only): the count is then never too large.
        n_st = 1 + (1 if objs else 0) + (1 if uf else 0)
        P.raw("s_mov_b64 exec, %s" % S_ST)
        for _ in range((DB - 1) * n_st):
            P.raw("global_store_dword %s, %s, off" % (v2(PST), XA))
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
        P.label("Lloop%d_%%=" % idx, reset=False)
        in_body[0] = True
        for j in range(DB):
            c = j % 2
            X, XN = (XA, XB) if c == 0 else (XB, XA)
            P.comment("---- step, slot %d (objs %d, u_first %d)" % (j, objs, uf))
            fstep(c, X, XN, j, objs, uf)
            P.raw("s_sub_i32 %s, %s, 1" % (S_N, S_N))
            P.raw("s_cmp_lg_u32 %s, 0" % S_N)
            if j < DB - 1:
                P.raw("s_cbranch_scc0 Lpassend_%=")
            else:
                P.raw("s_cbranch_scc1 Lloop%d_%%=" % idx)
        in_body[0] = False
        if idx < 3:
            P.raw("s_branch Lpassend_%=")
    P.label("Lpassend_%=")
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")       # the ring is refilled from t = 0 by the next pass
    row_sum(COSTP)
    row_sum(DELTA)
    row_sum(OLDP)
    P.raw("s_cmp_lg_u32 %s, 0" % S_PASS)
    P.raw("s_cbranch_scc1 Lnotfirst_%=")
    P.v("v_mov_b32_e32 %s, %s" % (OLD, OLDP), writes=(OLD,), reads=(OLDP,))                          # :191
    P.label("Lnotfirst_%=", reset=False)
    P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (COST, COST, COSTP, S_SRCH), writes=(COST,), reads=(COST, COSTP))
    P.v("v_addc_co_u32_e64 %s, %s, %s, 0, %s" % (NLS, S_T0, NLS, S_SRCH), writes=(NLS,), reads=(NLS,))  # passes run
    P.v("v_cmp_lt_f32_e64 %s, 0, %s" % (S_W, DELTA), reads=(DELTA,))                                 # :266 cost > OLD_COST
    P.raw("s_and_b64 %s, %s, %s" % (S_W, S_W, S_SRCH))
    P.v("v_mul_f32_e32 %s, %%[decay], %s" % (T1, ALPHA), writes=(T1,), reads=(ALPHA,))
    P.v("v_cndmask_b32_e64 %s, %s, %s, %s" % (ALPHA, ALPHA, T1, S_W), writes=(ALPHA,), reads=(ALPHA, T1))   # :268
    P.raw("s_add_u32 %s, %s, 1" % (S_PASS, S_PASS))
    P.raw("s_cmp_lt_u32 %s, %s" % (S_PASS, S_CAP))
    P.raw("s_cselect_b64 %s, %s, 0" % (S_SRCH, S_W))                                                 # :196
    P.raw("s_cmp_lg_u64 %s, 0" % S_SRCH)
    P.raw("s_cbranch_scc1 Lpass_%=")
    P.v("v_cndmask_b32_e64 %s, 0, 1, %s" % (WF, S_W), writes=(WF,))        # still worse at the cap             :274
    P.raw("s_branch Lend_%=")
    emit_slow_paths()
    P.label("Lend_%=")
    for name, reg in (("cost", COST), ("oldc", OLD), ("alpha", ALPHA), ("nls", NLS), ("worse", WF)):
        P.v("v_mov_b32_e32 %%[%s], %s" % (name, reg))

    outs = [("cost", '"=&v"(cost)'), ("oldc", '"=&v"(oldc)'), ("alpha", '"=&v"(alpha)'), ("nls", '"=&v"(nls)'),
            ("worse", '"=&v"(worse)')]
    ins = []
    for q in range(KD):
        ins.append(("ptr0_%d_lo" % q, '"v"((unsigned)in.ptr0[%d])' % q))
        ins.append(("ptr0_%d_hi" % q, '"v"((unsigned)(in.ptr0[%d] >> 32))' % q))
        ins.append(("str%d" % q, '"v"(in.str[%d])' % q))
        ins.append(("strl%d" % q, '"v"(in.strl[%d])' % q))
    for name in ("pst0", "puf0", "pobj0"):
        ins.append((name + "_lo", '"v"((unsigned)in.%s)' % name))
        ins.append((name + "_hi", '"v"((unsigned)(in.%s >> 32))' % name))
    for name in ("a_row", "a_row2", "a_aff", "a_crow", "a_caff", "a_hat", "a_lb", "dst", "dobj", "decay", "cap", "want_objs",
                 "uf_mask", "dmrow"):
        ins.append((name, '"v"(in.%s)' % name))
    for name in ("ring", "T"):
        ins.append((name, '"s"(in.%s)' % name))
    clob = ['"v%d"' % i for i in range(VBASE, last_vgpr + 1)] + ['"s%d"' % i for i in ([70] + list(range(72, 98)))] + \
        ['"vcc"', '"scc"', '"memory"']
    o = []
    o.append("// (%d,%d): %d instructions: set-up, pass prologue and epilogue, four pass bodies of %d unrolled steps\n"
             % (nx, nu, P.n_instr, DB))
    o.append("template <>\nstruct MpcFwdAsm<%d, %d> {\n" % (nx, nu))
    o.append("  static constexpr bool kAvailable = true;\n")
    o.append("  static constexpr int KD = %d, SLOT_B = %d, DEPTH = %d, RING_BYTES = %d, ZERO = %d;\n"
             % (KD, L.SLOT_B, DB, DB * L.SLOT_B, L.ZERO))
    o.append("  static constexpr int CH_C = %d, CH_c = %d, CH_F = %d, CH_f = %d, CH_K = %d, CH_k = %d, CH_u = %d, CH_lo = %d, CH_hi = %d, CH_x = %d, CH_END = %d;\n"
             % (L.CH["C"], L.CH["c"], L.CH["F"], L.CH["f"], L.CH["K"], L.CH["k"], L.CH["u"], L.CH["lo"], L.CH["hi"],
                L.CH["x"], L.END))
    o.append("  static __device__ __forceinline__ void run(const MpcFwdAsmIn &in, float &cost, float &oldc, float &alpha, int &nls, int &worse) {\n")
    o.append("    asm volatile(\n")
    for ln in P.text():
        o.append('        "%s\\n\\t"\n' % ln)
    o.append("        : " + ", ".join("[%s] %s" % x for x in outs) + "\n")
    o.append("        : " + ", ".join("[%s] %s" % x for x in ins) + "\n")
    o.append("        : " + ", ".join(clob) + ");\n")
    o.append("  }\n};\n\n")
    return "".join(o)


HEADER = """// mpc_fwd_asm_gen.hpp - GENERATED by gen_mpc_fwd_asm.py; do not edit.
// Whole-kernel gfx950 instruction streams of MPCstep.forward_rec (mpc/mpc_step.py:175-286) for a LinDx and a QuadCost,
// one per (nx, nu).  The C++ side that prepares the per-lane operands is mpc_fwd_asm_kernel.hpp.
#pragma once
#include <cstdint>

namespace dmpc {

// per-lane operands of the instruction stream (see mpc_fwd_asm_kernel.hpp for how they are filled)
struct MpcFwdAsmIn {
  uint64_t ptr0[4], str[4], strl[4];  // LDS-DMA source of this lane's chunk at t = 0, time stride, stride of the LAST advance
  unsigned a_row, a_row2;             // LDS byte addresses (ring slot 0): this lane's row of [F | K] columns 0..nx-1 / nx..ns-1
  unsigned a_aff, a_crow, a_caff;     //   f_i | k_m ; row of C ; c
  unsigned a_hat, a_lb;               //   [x^ ; u^][lane] ; lower (upper sits 4 nu floats behind)
  uint64_t pst0, puf0, pobj0;         // store pointers at t = 0: [x_t | u_t], u_first (control lanes), objs (lane 0 of a row)
  uint64_t dst, dobj;                 // their time strides (per lane)
  float decay;                        // line-search decay (same in every lane)
  unsigned ring;                      // wave-uniform: LDS byte address of this wave's ring
  int T;
  int cap, want_objs;                 // (VGPR operands, same in every lane)
  unsigned uf_mask;                   // all ones: u_first wanted, 0: not
  unsigned dmrow;                     // per lane, 4 bits per DMA instruction q (bits 4q..4q+3): the trajectories of the wave whose
                                      // data this lane's chunk of instruction q holds (all four: always fetched)
};

template <int NX, int NU>
struct MpcFwdAsm {
  static constexpr bool kAvailable = false;
};

"""


def main():
    out = [HEADER]
    for nx, nu in SHAPES:
        try:
            out.append(gen_fwd(nx, nu))
        except AssertionError as e:     # a shape whose slot has no room for the zero reads: no stream
            print("skip (%d,%d): %s" % (nx, nu, e))
    out.append("}  // namespace dmpc\n")
    with open(OUT, "w") as fh:
        fh.write("".join(out))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
