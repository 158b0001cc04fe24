"""Instruction counts of a kernel's loops, from the code object that is in the library: the gfx950 code object of one
translation unit (csrc/build/<unit>.o) is unbundled and disassembled, the kernel whose demangled name contains <substring> is cut
out, and every loop (a backward branch) is listed with its instructions by class.  At one wavefront per SIMD a kernel's time is
its instruction issue, so these counts x the measured cycles per issue are its time.
    python scripts/isa_loop_count.py lqr_api "lqr_kernel<8, 4, 16, false, 0, true, false>" """
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(unit):
    obj = os.path.join(ROOT, "chainer_differentiable_mpc_amd", "csrc", "build", unit + ".o")
    tmp = "/tmp/isa_%s.o" % unit
    subprocess.run(["cp", obj, tmp], check=True)
    subprocess.run([LLVM + "llvm-objdump", "--offloading", tmp], capture_output=True, check=True)   # writes <tmp>.0.<target> beside it
    co = tmp + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    return subprocess.run([LLVM + "llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True, check=True).stdout


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    unit, want = sys.argv[1], sys.argv[2]
    text = disassemble(unit)
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    hits = [b for b in blocks if want in b.split("\n", 1)[0]]
    if not hits:
        raise SystemExit("no kernel with %r in %s" % (want, unit))
    for blk in hits:
        head, body = blk.split("\n", 1)
        name = head[head.index("<") + 1:head.rindex(">")]
        ins = []      # (address, opcode, text)
        for ln in body.split("\n"):
            m = re.match(r"\s+(\S+)\s(.*?)\s*//\s*([0-9A-Fa-f]{12}):(.*)$", ln)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), (m.group(1) + " " + m.group(2)).strip(), m.group(4)))
        addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
        total = collections.Counter(classify(op) for _, op, _, _ in ins)
        print("%s\n  whole kernel: %d instructions  %s" % (name, len(ins), dict(total)))
        loops = []
        for i, (a, op, txt, tail) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", tail)
                tgt = None
                if m:
                    base = ins[0][0]
                    tgt = base + int(m.group(1), 16)
                if tgt is not None and tgt <= a and tgt in addr_index:
                    loops.append((addr_index[tgt], i))
        for lo, hi in sorted(loops):
            cnt = collections.Counter(classify(op) for _, op, _, _ in ins[lo:hi + 1])
            dpp = sum(1 for _, _, t, _ in ins[lo:hi + 1] if "dpp" in t or "row_" in t)
            inner = [1 for l2, h2 in loops if lo < l2 and h2 < hi]
            print("  loop @%d..%d: %4d instructions%s  %s  (dpp-modified: %d)" % (
                lo, hi, hi - lo + 1, "  [contains %d inner loops]" % len(inner) if inner else "", dict(cnt), dpp))


if __name__ == "__main__":
    main()
