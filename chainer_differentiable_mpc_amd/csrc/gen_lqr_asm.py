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

SHAPES = [(8, 2), (3, 1), (4, 2), (6, 2), (2, 2), (1, 1), (2, 1), (3, 2)]
DB = 3        # backward ring depth == number of rotating register sets
DF = 6        # forward ring depth == unroll (lcm of 2 row sets and 3 accumulators)
KROW = 12     # floats per gain row in LDS: [K_m (nx) | 0 (nu) | k_m | pad], 8-byte aligned rows
VBASE = 128   # first VGPR owned by the asm block (operands chosen by hipcc live below)


class Layout:
    def __init__(self, nx, nu):
        self.nx, self.nu, self.ns = nx, nu, nx + nu
        ns = self.ns
        self.nC, self.nc, self.nF, self.nf = ns * ns, ns, nx * ns, nx          # 16-byte chunks per wave-step
        self.OFF_C = 0
        self.OFF_c = 16 * self.nC
        self.OFF_F = self.OFF_c + 16 * self.nc
        self.OFF_f = self.OFF_F + 16 * self.nF
        self.nchunk_b = self.nC + self.nc + self.nF + self.nf
        self.ndma_b = (self.nchunk_b + 63) // 64
        self.SLOT_B = self.ndma_b * 1024
        self.FOFF_f = 16 * self.nF
        self.nchunk_f = self.nF + self.nf
        self.ndma_f = (self.nchunk_f + 63) // 64
        self.SLOT_F = self.ndma_f * 1024
        self.RING = max(DB * self.SLOT_B, DF * self.SLOT_F)
        assert self.ndma_b * 1024 - 1024 <= 4095 and ns + 1 <= 12 and nu in (1, 2)
        assert (DB - 1) * self.ndma_b <= 63 and (DF - 1) * self.ndma_f <= 63


class Prog:
    """instruction emitter with a small hazard tracker (ages are in wait states since the VALU write)"""

    def __init__(self):
        self.lines = []
        self.age = {}       # vgpr -> wait states since a VALU wrote it
        self.trans = {}     # vgpr -> wait states since a transcendental wrote it
        self.n_instr = 0

    def _tick(self, n=1):
        for d in (self.age, self.trans):
            for r in list(d):
                d[r] += n
                if d[r] > 8:
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

    def exec_written(self):
        # SALU write of EXEC -> DPP: not a documented hazard (the documented one is a VALU write, 5 wait states);
        # four wait states are kept anyway
        self.age = {"*": -2}

    def _need(self, table, reg, states):
        have = table.get(reg, table.get("*", 99))
        if have < states:
            self.nop(states - have)

    def valu(self, text, writes=(), reads=(), dpp=None, trans=False):
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


def gen_kernel(nx, nu, write_k):
    L = Layout(nx, nu)
    ns, aff = L.ns, L.ns
    P = Prog()
    R = Regs(VBASE)
    # ---- operand names (C++ side: struct LqrAsmIn of lqr_asm_kernel.hpp)
    ptr = ["%%[ptr%d]" % q for q in range(L.ndma_b)]
    str1 = ["%%[str1_%d]" % q for q in range(L.ndma_b)]
    strd = ["%%[str%d]" % q for q in range(L.ndma_b)]
    aq = ["%%[aq%d]" % i for i in range(ns)]
    af = ["%%[af%d]" % k for k in range(nx)]
    fptr = ["%%[fptr%d]" % q for q in range(L.ndma_f)]
    fstr = ["%%[fstr%d]" % q for q in range(L.ndma_f)]
    pk = ["%%[pk%d]" % m for m in range(nu)]

    # ---- fixed registers
    Q = [R.take(ns) for _ in range(3)]
    F = [R.take(nx) for _ in range(3)]
    W = R.take(nx)
    A = [R.take(nu) for _ in range(nu)]
    Kt = R.take(nu)
    Rr = R.take(nu)
    tP, tPQ, tL0, tM1, tRA, tRB, tRP, tT, tLL, tD2, tRD, tY1, tT2 = R.take(13)
    MINPIV = R.take(1)[0]
    # forward sweep registers reuse the Q / F sets (the backward sweep is over by then)
    RF = Regs(VBASE)
    M = [RF.take(ns, align=4), RF.take(ns, align=4)]
    ACC = RF.take(3)
    assert RF.next <= int(MINPIV[1:])
    last_vgpr = R.next - 1
    assert last_vgpr <= 255

    S_N, S_TF = "s70", "s71"
    S_KM, S_SM, S_UM = "s[72:73]", "s[74:75]", "s[76:77]"

    def mask64(lanes):
        m16 = sum(1 << l for l in lanes)
        return m16 | (m16 << 16)

    km = mask64(list(range(nx)) + [aff])
    sm = mask64(range(ns))
    um = mask64(range(nx, ns))

    def issue_group(ptrs, slot, slot_bytes):
        if slot == 0:
            P.raw("s_mov_b32 m0, %[ring]")
        else:
            P.raw("s_add_u32 m0, %%[ring], %d" % (slot * slot_bytes))
        P.nop(1)
        for q, p in enumerate(ptrs):
            off = (" offset:%d" % (q * 1024)) if q else ""
            P.raw("global_load_lds_dwordx4 %s, off%s" % (p, off))

    uniq = [0]

    def advance(ptrs, strides):
        uniq[0] += 1
        lab = "Ladv%d_%%=" % uniq[0]
        P.raw("s_cmp_gt_i32 %s, 0" % S_TF)
        P.raw("s_cbranch_scc0 " + lab)
        for p, s in zip(ptrs, strides):
            P.v("v_lshl_add_u64 %s, %s, 0, %s" % (p, p, s))
        P.raw("s_sub_i32 %s, %s, 1" % (S_TF, S_TF))
        P.label(lab, reset=False)   # only pointer registers are written on the fall-through path

    def read_set(s, slot):
        off = slot * L.SLOT_B
        for i in range(ns):
            P.raw("ds_read_b32 %s, %s offset:%d" % (Q[s][i], aq[i], off))
        for k in range(nx):
            P.raw("ds_read_b32 %s, %s offset:%d" % (F[s][k], af[k], off))

    def gains(s):
        """K~ = -Quu^-1 [Qux | Quu | qu] per lane (lqr_recursion.py:112-120); leaves A (Quu), Kt, and MINPIV"""
        Qs = Q[s]
        for m in range(nu):
            for l in range(nu):
                P.mov_dpp(A[m][l], Qs[nx + m], nx + l)
        if nu == 1:
            P.v("v_rcp_f32_e32 %s, %s" % (tRP, A[0][0]), writes=(tRP,), reads=(A[0][0],), trans=True)
            P.v("v_min_f32_e64 %s, |%s|, %s" % (MINPIV, A[0][0], MINPIV), writes=(MINPIV,), reads=(A[0][0], MINPIV))
            P.nop(1)
            P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, A[0][0], tRP), writes=(tT,), reads=(A[0][0], tRP))
            P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP))
            P.raw("s_mov_b64 exec, " + S_KM)
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[0], Qs[nx], tRP), writes=(Kt[0],), reads=(Qs[nx], tRP))
        else:
            a00, a01, a10, a11 = A[0][0], A[0][1], A[1][0], A[1][1]
            P.v("v_cmp_gt_f32_e64 vcc, |%s|, |%s|" % (a10, a00), reads=(a10, a00))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tP, a00, a10), writes=(tP,), reads=(a00, a10))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tL0, a10, a00), writes=(tL0,), reads=(a00, a10))
            P.v("v_rcp_f32_e32 %s, %s" % (tRP, tP), writes=(tRP,), reads=(tP,), trans=True)
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tPQ, a01, a11), writes=(tPQ,), reads=(a01, a11))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tM1, a11, a01), writes=(tM1,), reads=(a01, a11))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRA, Qs[nx], Qs[nx + 1]), writes=(tRA,), reads=(Qs[nx], Qs[nx + 1]))
            P.v("v_cndmask_b32_e32 %s, %s, %s, vcc" % (tRB, Qs[nx + 1], Qs[nx]), writes=(tRB,), reads=(Qs[nx], Qs[nx + 1]))
            P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, tP, tRP), writes=(tT,), reads=(tP, tRP))
            P.v("v_fmac_f32_e32 %s, %s, %s" % (tRP, tT, tRP), writes=(tRP,), reads=(tT, tRP))
            P.v("v_mul_f32_e32 %s, %s, %s" % (tLL, tL0, tRP), writes=(tLL,), reads=(tL0, tRP))
            P.v("v_fma_f32 %s, -%s, %s, %s" % (tD2, tLL, tPQ, tM1), writes=(tD2,), reads=(tLL, tPQ, tM1))
            P.v("v_rcp_f32_e32 %s, %s" % (tRD, tD2), writes=(tRD,), reads=(tD2,), trans=True)
            P.v("v_fma_f32 %s, -%s, %s, %s" % (tY1, tLL, tRA, tRB), writes=(tY1,), reads=(tLL, tRA, tRB))
            P.v("v_min3_f32 %s, |%s|, |%s|, %s" % (MINPIV, tP, tD2, MINPIV), writes=(MINPIV,), reads=(tP, tD2, MINPIV))
            P.v("v_fma_f32 %s, -%s, %s, 1.0" % (tT, tD2, tRD), writes=(tT,), reads=(tD2, tRD))
            P.v("v_fmac_f32_e32 %s, %s, %s" % (tRD, tT, tRD), writes=(tRD,), reads=(tT, tRD))
            P.raw("s_mov_b64 exec, " + S_KM)
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[1], tY1, tRD), writes=(Kt[1],), reads=(tY1, tRD))
            P.v("v_fma_f32 %s, %s, %s, %s" % (tT2, tPQ, Kt[1], tRA), writes=(tT2,), reads=(tPQ, Kt[1], tRA))
            P.v("v_mul_f32_e64 %s, %s, -%s" % (Kt[0], tT2, tRP), writes=(Kt[0],), reads=(tT2, tRP))
        # gain rows -> LDS (and HBM when the caller wants Ks/ks), still under the K mask
        for m in range(nu):
            off = (" offset:%d" % (m * KROW * 4)) if m else ""
            P.raw("ds_write_b32 %%[ak], %s%s" % (Kt[m], off))
        if write_k:
            for m in range(nu):
                P.raw("global_store_dword %s, %s, off" % (pk[m], Kt[m]))
        P.raw("s_mov_b64 exec, -1")
        P.exec_written()
        if write_k:
            for m in range(nu):
                P.v("v_lshl_add_u64 %s, %s, 0, %%[dk]" % (pk[m], pk[m]))
        P.v("v_add_u32_e32 %%[ak], %d, %%[ak]" % ((-nu * KROW * 4) & 0xffffffff))

    def vupdate(s):
        """V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~) in place in Q[s][0..nx-1]   (lqr_recursion.py:151-152)"""
        Qs = Q[s]
        for m in range(nu):
            P.v("v_fma_f32 %s, %s, %s, %s" % (Rr[m], A[m][0], Kt[0], Qs[nx + m]), writes=(Rr[m],),
                reads=(A[m][0], Kt[0], Qs[nx + m]))
            for l in range(1, nu):
                P.v("v_fmac_f32_e32 %s, %s, %s" % (Rr[m], A[m][l], Kt[l]), writes=(Rr[m],), reads=(A[m][l], Kt[l]))
        for m in range(nu):
            for i in range(nx):
                P.fmac_dpp(Qs[i], Qs[i], Kt[m], nx + m)
        for m in range(nu):
            for i in range(nx):
                P.fmac_dpp(Qs[i], Kt[m], Rr[m], i)

    def bstep(s, first):
        p, n = (s + 2) % 3, (s + 1) % 3
        V = Q[p]
        P.raw("s_waitcnt lgkmcnt(0)")
        issue_group(ptr, s, L.SLOT_B)
        advance(ptr, strd)
        if not first:
            # W~ = V~ F~  (+ v in column aff):  W[i] = sum_k bcast<k>(V[i]) F[k] + bcast<aff>(V[i]) e_aff
            for i in range(nx):
                P.mul_dpp(W[i], V[i], F[s][0], 0)
            for k in range(1, nx):
                for i in range(nx):
                    P.fmac_dpp(W[i], V[i], F[s][k], k)
            for i in range(nx):
                P.fmac_dpp(W[i], V[i], "%[eaff]", aff)
        P.raw("s_waitcnt vmcnt(%d)" % ((DB - 1) * L.ndma_b))
        read_set(n, n)
        if not first:
            # Q~ += F~^T W~ ; the u-rows first in the last pass so that the Quu broadcasts need no wait states
            for k in range(nx):
                order = list(range(ns)) if k < nx - 1 else list(range(nx, ns)) + list(range(nx))
                for i in order:
                    P.fmac_dpp(Q[s][i], F[s][k], W[k], i)
        gains(s)
        vupdate(s)

    # =============================================================== backward sweep
    P.comment("---- prologue")
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for name, val in ((S_KM, km), (S_SM, sm), (S_UM, um)):
        lo = int(name[2:name.index(":")])
        P.raw("s_mov_b32 s%d, 0x%x" % (lo, val & 0xffffffff))
        P.raw("s_mov_b32 s%d, 0x%x" % (lo + 1, val & 0xffffffff))
    for m in range(nu):
        P.v("v_mov_b32_e32 %s, 0" % Kt[m], writes=(Kt[m],))
    P.v("v_mov_b32_e32 %s, 0x7f7fffff" % MINPIV, writes=(MINPIV,))
    P.raw("s_sub_i32 %s, %%[T], 1" % S_TF)
    issue_group(ptr, 0, L.SLOT_B)
    advance(ptr, str1)
    for j in range(1, DB):
        issue_group(ptr, j, L.SLOT_B)
        advance(ptr, strd)
    P.raw("s_waitcnt vmcnt(%d)" % ((DB - 1) * L.ndma_b))
    read_set(0, 0)
    P.raw("s_sub_i32 %s, %%[T], 1" % S_N)      # steps left after the first
    P.comment("---- t = T-1")
    bstep(0, True)
    P.label("Lbwd_%=")
    for s in (1, 2, 0):
        P.comment("---- backward step, register set %d" % s)
        bstep(s, False)
        P.raw("s_sub_i32 %s, %s, 1" % (S_N, S_N))
        P.raw("s_cmp_lg_u32 %s, 0" % S_N)
        if s != 0:
            P.raw("s_cbranch_scc0 Lbwd_done_%=")
        else:
            P.raw("s_cbranch_scc1 Lbwd_%=")
    P.label("Lbwd_done_%=")
    n_bwd = P.n_instr

    # =============================================================== forward rollout
    P.comment("---- forward rollout")
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    P.raw("s_sub_i32 %s, %%[T], 2" % S_TF)
    for j in range(DF):
        issue_group(fptr, j, L.SLOT_F)
        advance(fptr, fstr)
    P.raw("s_waitcnt vmcnt(%d)" % ((DF - 1) * L.ndma_f))

    def read_rows(c, a):
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
        P.raw("ds_read_b32 %s, %%[aaff]" % ACC[a])

    read_rows(0, 0)
    P.v("v_mov_b32_e32 %s, %%[xv]" % ACC[2], writes=(ACC[2],))
    P.raw("s_sub_i32 %s, %%[T], 1" % S_N)      # full steps t = 0 .. T-2
    P.raw("s_cmp_lg_u32 %s, 0" % S_N)
    P.raw("s_cbranch_scc0 Lfin0_%=")
    P.label("Lfwd_%=")
    n_fwd0 = P.n_instr
    for j in range(DF):
        c, a = j % 2, j % 3
        o, an, ap = 1 - c, (a + 1) % 3, (a + 2) % 3
        P.comment("---- forward step, slot %d" % j)
        P.raw("s_waitcnt lgkmcnt(0)")
        issue_group(fptr, j, L.SLOT_F)
        advance(fptr, fstr)
        P.raw("s_waitcnt vmcnt(%d)" % ((DF - 1) * L.ndma_f))
        d = "2" if j == DF - 1 else ""
        P.v("v_add_u32_e32 %%[arow], %%[drow%s], %%[arow]" % d)
        P.v("v_add_u32_e32 %%[aaff], %%[daff%s], %%[aaff]" % d)
        read_rows(o, an)
        for jj in range(nx):                       # u_t = K_t x_t + k_t (lanes nx+m) ; f_t + Fx x_t (lanes < nx)
            P.fmac_dpp(ACC[a], ACC[ap], M[c][jj], jj)
        for m in range(nu):                        # x_{t+1} += Fu u_t : the gain rows hold 0 in these columns
            P.fmac_dpp(ACC[a], ACC[a], M[c][nx + m], nx + m)
        P.raw("s_mov_b64 exec, " + S_SM)
        P.raw("global_store_dword %%[pst], %s, off" % ACC[a])
        P.raw("s_mov_b64 exec, -1")
        P.v("v_lshl_add_u64 %[pst], %[pst], 0, %[dst]")
        P.raw("s_sub_i32 %s, %s, 1" % (S_N, S_N))
        P.raw("s_cmp_lg_u32 %s, 0" % S_N)
        if j < DF - 1:
            P.raw("s_cbranch_scc0 Lfin%d_%%=" % (j + 1))
        else:
            P.raw("s_cbranch_scc1 Lfwd_%=")
            P.raw("s_branch Lfin0_%=")
    n_fwd = P.n_instr - n_fwd0
    for j in range(DF):                            # t = T-1: only u_{T-1}
        c, a = j % 2, j % 3
        ap = (a + 2) % 3
        P.label("Lfin%d_%%=" % j)
        P.raw("s_waitcnt lgkmcnt(0)")
        for jj in range(nx):
            P.fmac_dpp(ACC[a], ACC[ap], M[c][jj], jj)
        P.raw("s_mov_b64 exec, " + S_UM)
        P.raw("global_store_dword %%[pst], %s, off" % ACC[a])
        P.raw("s_mov_b64 exec, -1")
        P.v("v_mov_b32_e32 %%[xvout], %s" % ACC[a])
        P.raw("s_branch Ldone_%=")
    P.label("Ldone_%=")
    P.v("v_mov_b32_e32 %%[minpiv], %s" % MINPIV)
    P.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")

    # ---- operand lists
    outs = [("xvout", '"=&v"(xvout)'), ("minpiv", '"=&v"(minpiv)')]
    rw = []
    for q in range(L.ndma_b):
        rw.append(("ptr%d" % q, '"+v"(in.ptr[%d])' % q))
    for q in range(L.ndma_f):
        rw.append(("fptr%d" % q, '"+v"(in.fptr[%d])' % q))
    rw += [("ak", '"+v"(in.ak)'), ("arow", '"+v"(in.arow)'), ("aaff", '"+v"(in.aaff)'), ("pst", '"+v"(in.pst)')]
    if write_k:
        for m in range(nu):
            rw.append(("pk%d" % m, '"+v"(in.pk[%d])' % m))
    ins = []
    for q in range(L.ndma_b):
        ins.append(("str1_%d" % q, '"v"(in.str1[%d])' % q))
        ins.append(("str%d" % q, '"v"(in.str[%d])' % q))
    for q in range(L.ndma_f):
        ins.append(("fstr%d" % q, '"v"(in.fstr[%d])' % q))
    for i in range(ns):
        ins.append(("aq%d" % i, '"v"(in.aq[%d])' % i))
    for k in range(nx):
        ins.append(("af%d" % k, '"v"(in.af[%d])' % k))
    ins += [("eaff", '"v"(in.eaff)'), ("drow", '"v"(in.drow)'), ("drow2", '"v"(in.drow2)'),
            ("daff", '"v"(in.daff)'), ("daff2", '"v"(in.daff2)'), ("dst", '"v"(in.dst)'), ("xv", '"v"(in.xv)')]
    if write_k:
        ins.append(("dk", '"v"(in.dk)'))
    ins += [("ring", '"s"(in.ring)'), ("T", '"s"(in.T)')]
    clob = ['"v%d"' % i for i in range(VBASE, last_vgpr + 1)] + ['"s70"', '"s71"', '"s72"', '"s73"', '"s74"', '"s75"',
                                                                 '"s76"', '"s77"', '"vcc"', '"scc"', '"memory"']

    name = "LqrAsm<%d, %d, %s>" % (nx, nu, "true" if write_k else "false")
    o = []
    o.append("// (%d,%d) write_k=%d: %d instructions in the 3 unrolled backward steps + prologue, %d in the %d unrolled\n"
             "// forward steps\n" % (nx, nu, write_k, n_bwd, n_fwd, DF))
    o.append("template <>\nstruct %s {\n" % name)
    o.append("  static constexpr bool kAvailable = true;\n")
    o.append("  static constexpr int NDB = %d, NDF = %d, SLOT_B = %d, SLOT_F = %d, RING_BYTES = %d, KROW = %d, DEPTH_F = %d;\n"
             % (L.ndma_b, L.ndma_f, L.SLOT_B, L.SLOT_F, L.RING, KROW, DF))
    o.append("  static constexpr int OFF_C = %d, OFF_c = %d, OFF_F = %d, OFF_f = %d, FOFF_f = %d;\n"
             % (L.OFF_C, L.OFF_c, L.OFF_F, L.OFF_f, L.FOFF_f))
    o.append("  static __device__ __forceinline__ void run(LqrAsmIn<%d, %d> &in, float &xvout, float &minpiv) {\n" % (nx, nu))
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
// 16-lane row layout, one per (nx, nu, write_k).  The C++ side that prepares the per-lane operands is
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
  float eaff;                        // 1 in lane `aff`, else 0
  uint64_t pk[NU], dk;               // Ks/ks store pointers (t = T-1) and their time stride (write_k)
  // forward sweep
  uint64_t fptr[2], fstr[2];
  unsigned arow, aaff, drow, drow2, daff, daff2;
  uint64_t pst, dst;                 // [x_{t+1} | u_t] store pointer and time stride
  float xv;                          // x_init in lanes < nx
  // wave-uniform
  unsigned ring;                     // LDS byte address of this wave's ring
  int T;
};

template <int NX, int NU, bool WRITE_K>
struct LqrAsm {
  static constexpr bool kAvailable = false;
};

"""


def main():
    out = [HEADER]
    for nx, nu in SHAPES:
        for write_k in (False, True):
            out.append(gen_kernel(nx, nu, write_k))
    out.append("}  // namespace dmpc\n")
    with open(OUT, "w") as fh:
        fh.write("".join(out))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
