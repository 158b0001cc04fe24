#!/usr/bin/env python
"""Emit dpp_blocks_gen.hpp: hand-scheduled gfx950 instruction blocks for the 16-lane row kernels.

Why a generator: the inner products of the Riccati sweep are  acc[i] += bcast<k>(a[i]) * b[k]
with bcast = DPP row_newbcast.  gfx950 can fold the broadcast into the FMA
(`v_fmac_f32_dpp acc, a, b row_newbcast:k` - ONE issue slot), but hipcc (ROCm 7.2) never forms
that instruction from the builtin: it emits v_mov_b32_dpp + s_nop (DPP read-after-VALU-write
hazard) + v_fmac_f32, i.e. ~2.8 issue slots per FMA (measured on the (8,2) kernel: 150 fmac,
148 mov_dpp, 123 s_nop per timestep).  With one wavefront per SIMD the kernel is VALU-issue
bound, so the blocks are written as inline asm.  The asm text needs literal lane numbers and
a fixed operand count, hence one specialisation per (nx, nu) - generated here, committed, and
selected at compile time; every other shape uses the builtin-based template in riccati_blocks.hpp.

Hazards handled inside each statement (hipcc pads nothing inside asm):
  * VALU write -> DPP read of the same VGPR needs 2 wait states: every DPP source of a block is an
    INPUT of the statement (never written inside it), and the statement opens with `s_nop 1`, which
    covers a producer that hipcc scheduled right in front of it;
  * the statement ends with `s_nop 1` so that a compiler-generated v_mov_b32_dpp reading one of its
    outputs right behind it is safe as well;
  * accumulators are early-clobber ("+&v"): they are written before all inputs have been read.
Accumulators are interleaved (k outer, i inner) so that no two consecutive FMAs depend on each other.

    python chainer_differentiable_mpc_amd/csrc/gen_dpp_blocks.py   # rewrites dpp_blocks_gen.hpp
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "dpp_blocks_gen.hpp")

# (nx, nu) with nx + nu + 1 <= 16 that get asm blocks (DMPC_LQR_SHAPES with L = 16 and DMPC_LQR_CONTAINERS of lqr_api.hip)
SHAPES = [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (4, 2), (6, 2), (8, 2), (4, 4), (8, 4), (12, 3),
          (14, 1), (13, 2), (11, 4), (5, 5), (10, 5), (9, 6), (8, 7), (7, 8)]      # from (14,1) on: containers only (lqr_api.hip DMPC_LQR_CONTAINERS)
MAX_OPERANDS = 30
# float64 variant (gen_dpp_blocks_f64.py -> dpp_blocks_f64_gen.hpp, `RiccatiBlocks64<NX, NU, 16>`): gfx90a+ has the DPP form
# of the double-precision FMA as well - `v_fmac_f64_dpp ... row_newbcast:k` (the only dpp_ctrl a 64-bit DPP takes), operands
# are 64-bit register pairs ("v" with a double); there is no v_mul_f64_dpp, so outer2 starts from a zeroed row
PREC = {"fmac": "v_fmac_f32_dpp", "type": "float", "struct": "RiccatiBlocks", "mul": "v_mul_f32_dpp"}


def fmac(acc, a, b, lane):
    return '"%s %%[%s], %%[%s], %%[%s] row_newbcast:%d row_mask:0xf bank_mask:0xf\\n\\t"' % (
        PREC["fmac"], acc, a, b, lane)


def statement(lines, outs, ins, tail_nop=True):
    """one asm statement; outs/ins: list of (name, c_expr).  tail_nop: the outputs may be read through DPP
    by compiler-generated code right behind the statement (true only for ftw: Q feeds the Quu broadcast);
    the other blocks' outputs are next consumed by another block, which opens with its own s_nop 1."""
    assert len(outs) + len(ins) <= MAX_OPERANDS, (len(outs), len(ins))
    lines = list(lines)
    if not tail_nop:
        lines[-1] = lines[-1].replace('\\n\\t"', '"')
    body = ['"s_nop 1\\n\\t"'] + lines + (['"s_nop 1"'] if tail_nop else [])
    # "+&v": early-clobber.  A block writes its accumulators before it has read every input, so an
    # accumulator must never share a register with an input - which hipcc would otherwise do whenever it
    # can prove both hold the same value on entry (e.g. V[i] = Q[i] right before vupd).
    o = ", ".join('[%s] "+&v"(%s)' % (n, e) for n, e in outs)
    i = ", ".join('[%s] "v"(%s)' % (n, e) for n, e in ins)
    return "    asm(" + "\n        ".join(body) + "\n        : " + o + "\n        : " + i + ");\n"


def chunks(seq, n):
    return [seq[i:i + n] for i in range(0, len(seq), n)]


def gen_vf(nx):
    """W[i] += sum_k bcast<k>(V[i]) * Fc[k]   (i < nx, k < nx): operands W rows + V rows + all Fc"""
    rows_per = max(1, (MAX_OPERANDS - nx) // 2)
    out = ""
    for rows in chunks(list(range(nx)), rows_per):
        lines = [fmac("w%d" % i, "v%d" % i, "f%d" % k, k) for k in range(nx) for i in rows]
        outs = [("w%d" % i, "W[%d]" % i) for i in rows]
        ins = [("v%d" % i, "V[%d]" % i) for i in rows] + [("f%d" % k, "Fc[%d]" % k) for k in range(nx)]
        out += statement(lines, outs, ins, tail_nop=False)
    return out


def gen_ftw(nx, ns):
    """Q[i] += sum_k bcast<i>(Fc[k]) * W[k]   (i < ns, k < nx): operands Q rows + all Fc + all W"""
    rows_per = max(1, MAX_OPERANDS - 2 * nx)
    out = ""
    for rows in chunks(list(range(ns)), rows_per):
        lines = [fmac("q%d" % i, "f%d" % k, "w%d" % k, i) for k in range(nx) for i in rows]
        outs = [("q%d" % i, "Q[%d]" % i) for i in rows]
        ins = [("f%d" % k, "Fc[%d]" % k) for k in range(nx)] + [("w%d" % k, "W[%d]" % k) for k in range(nx)]
        out += statement(lines, outs, ins)
    return out


def gen_vupd(nx, nu):
    """V[i] += sum_m bcast<nx+m>(Q[i]) * Kt[m] + sum_m bcast<i>(Kt[m]) * R[m]   (i < nx)"""
    rows_per = max(1, (MAX_OPERANDS - 2 * nu) // 2)
    out = ""
    for rows in chunks(list(range(nx)), rows_per):
        lines = [fmac("v%d" % i, "q%d" % i, "k%d" % m, nx + m) for m in range(nu) for i in rows]
        lines += [fmac("v%d" % i, "k%d" % m, "r%d" % m, i) for m in range(nu) for i in rows]
        outs = [("v%d" % i, "V[%d]" % i) for i in rows]
        ins = [("q%d" % i, "Q[%d]" % i) for i in rows] + [("k%d" % m, "Kt[%d]" % m) for m in range(nu)] + \
              [("r%d" % m, "R[%d]" % m) for m in range(nu)]
        out += statement(lines, outs, ins, tail_nop=False)
    return out


def gen_rowdot(n, name, lane0):
    """acc += sum_j bcast<lane0+j>(x) * M[j]  (single accumulator chain, forward sweep)"""
    outs = [("acc", "acc")]
    out = ""
    for js in chunks(list(range(n)), MAX_OPERANDS - 2):
        lines = [fmac("acc", "x", "m%d" % j, lane0 + j) for j in js]
        ins = [("x", "xu")] + [("m%d" % j, "M[%d]" % j) for j in js]
        out += statement(lines, outs, ins)
    return out


def mul_dpp(dst, a, b, lane):
    return '"%s %%[%s], %%[%s], %%[%s] row_newbcast:%d row_mask:0xf bank_mask:0xf\\n\\t"' % (
        PREC["mul"], dst, a, b, lane)


def gen_outer2(n):
    """row[j] = bcast<j>(x) * a + bcast<j>(y) * b   (j < n): the rows of dC / dF in the co-state kernel"""
    out = ""
    for js in chunks(list(range(n)), MAX_OPERANDS - 4):
        outs = [("r%d" % j, "row[%d]" % j) for j in js]
        ins = [("x", "x"), ("y", "y"), ("a", "a"), ("b", "b")]
        if PREC["mul"] is None:      # no DPP multiply in this precision: two FMAs into a zeroed row
            out += "".join("    row[%d] = 0.0;\n" % j for j in js)
            lines = [fmac("r%d" % j, "x", "a", j) for j in js] + [fmac("r%d" % j, "y", "b", j) for j in js]
            out += statement(lines, outs, ins, tail_nop=False)
            continue
        lines = [mul_dpp("r%d" % j, "x", "a", j) for j in js] + [fmac("r%d" % j, "y", "b", j) for j in js]
        body = statement(lines, outs, ins, tail_nop=False)
        out += body.replace('"+&v"', '"=&v"')
    return out


def gen_dots2(n):
    """p += sum_j bcast<j>(x) * M[j] ; q += sum_j bcast<j>(y) * M[j]   (two interleaved accumulator chains)"""
    out = ""
    for js in chunks(list(range(n)), MAX_OPERANDS - 4):
        lines = []
        for j in js:
            lines += [fmac("p", "x", "m%d" % j, j), fmac("q", "y", "m%d" % j, j)]
        outs = [("p", "p"), ("q", "q")]
        ins = [("x", "x"), ("y", "y")] + [("m%d" % j, "M[%d]" % j) for j in js]
        out += statement(lines, outs, ins, tail_nop=False)
    return out


def main(prec="f32"):
    global OUT
    if prec == "f64":
        PREC.update(fmac="v_fmac_f64_dpp", type="double", struct="RiccatiBlocks64", mul=None)
        OUT = os.path.join(HERE, "dpp_blocks_f64_gen.hpp")
    T_, S_ = PREC["type"], PREC["struct"]
    s = ["// GENERATED by gen_dpp_blocks.py - do not edit; regenerate and commit.",
         "// Fused broadcast-FMA (%s row_newbcast) blocks for the 16-lane row kernels." % PREC["fmac"],
         "#pragma once", '#include "%s"' % ("riccati_blocks.hpp" if prec == "f32" else "f64_row_blocks.hpp"), "",
         "namespace dmpc {", ""]
    for nx, nu in SHAPES:
        ns = nx + nu
        assert ns + 1 <= 16
        s.append("template <>\nstruct %s<%d, %d, 16> {" % (S_, nx, nu))
        s.append("  static constexpr bool kAsm = true;")
        s.append("  static __device__ __forceinline__ void vf(T_ (&W)[%d], const T_ (&V)[%d], "
                 "const T_ (&Fc)[%d]) {\n%s  }" % (nx, nx, nx, gen_vf(nx)))
        s.append("  static __device__ __forceinline__ void ftw(T_ (&Q)[%d], const T_ (&Fc)[%d], "
                 "const T_ (&W)[%d]) {\n%s  }" % (ns, nx, nx, gen_ftw(nx, ns)))
        s.append("  static __device__ __forceinline__ void vupd(T_ (&V)[%d], const T_ (&Q)[%d], "
                 "const T_ (&Kt)[%d], const T_ (&R)[%d]) {\n%s  }" % (nx, ns, nu, nu, gen_vupd(nx, nu)))
        s.append("  // acc += sum_{j<nx} bcast<j>(xu) * M[j]")
        s.append("  static __device__ __forceinline__ void dot_x(T_ &acc, const T_ xu, const T_ (&M)[%d]) "
                 "{\n%s  }" % (ns + 1, gen_rowdot(nx, "dot_x", 0)))
        s.append("  // acc += sum_{m<nu} bcast<nx+m>(xu) * M[nx+m]")
        body = gen_rowdot(nu, "dot_u", nx).replace("M[", "M[%d + " % nx)
        s.append("  static __device__ __forceinline__ void dot_u(T_ &acc, const T_ xu, const T_ (&M)[%d]) "
                 "{\n%s  }" % (ns + 1, body))
        s.append("  // row[j] = bcast<j>(x) * a + bcast<j>(y) * b, j < ns")
        s.append("  static __device__ __forceinline__ void outer2(T_ (&row)[%d], const T_ x, const T_ y, "
                 "const T_ a, const T_ b) {\n%s  }" % (ns, gen_outer2(ns)))
        s.append("  // p += sum_j bcast<j>(x) M[j], q += sum_j bcast<j>(y) M[j]")
        s.append("  static __device__ __forceinline__ void dots2_ns(T_ &p, T_ &q, const T_ (&M)[%d], const T_ x, "
                 "const T_ y) {\n%s  }" % (ns, gen_dots2(ns)))
        s.append("  static __device__ __forceinline__ void dots2_nx(T_ &p, T_ &q, const T_ (&M)[%d], const T_ x, "
                 "const T_ y) {\n%s  }" % (nx, gen_dots2(nx)))
        s.append("};\n")
    s.append("}  // namespace dmpc")
    open(OUT, "w").write("\n".join(s).replace("T_", T_) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
