"""Assemble ONE generated stream of lqr_asm_gen.hpp by itself (llvm-mc, dummy operand registers): hipcc reports errors
inside a 7,000-line asm statement at positions that do not name the instruction.
    python scripts/asm_check.py "LqrAsm<8, 2, false, true, false, false, false, true, true>" [run|issue_first]"""
import os, re, subprocess, sys
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer_differentiable_mpc_amd", "csrc")
s = open(os.path.join(HERE, sys.argv[3] if len(sys.argv) > 3 else "lqr_asm_gen.hpp")).read()
i = s.index("struct " + sys.argv[1])
blk = s[i:s.index("};", i)]
body = blk[blk.index("void %s(" % (sys.argv[2] if len(sys.argv) > 2 else "run")):]
body = body[:body.index("  }\n")]
lines = [m.group(1) for m in re.finditer(r'^\s*"(.*?)\\n\\t"\s*$', body, flags=re.M)]
decl = {a: b for a, b, _ in re.findall(r'\[(\w+)\] "[=+&]*([vs])"\((?:in\.)?(\w+)', body)}
wide = set(re.findall(r"uint64_t ([^;]+);", s[:s.index("template <int NX, int NU, bool WRITE_K")]))
wide = {w.split("[")[0].strip() for grp in wide for w in grp.split(",")}
member = {m.group(1): m.group(3) for m in re.finditer(r'\[(\w+)\] "[=+&]*([vs])"\((?:in\.)?(\w+)', body)}
regs, nv, nsr = {}, 0, 20
for name in sorted(set(re.findall(r"%\[(\w+)\]", "\n".join(lines)))):
    if decl.get(name) == "s":
        regs[name] = "s%d" % nsr; nsr += 1
    elif member.get(name) in wide:
        nv += nv % 2
        regs[name] = "v[%d:%d]" % (nv, nv + 1); nv += 2
    else:
        regs[name] = "v%d" % nv; nv += 1
out = [re.sub(r"%\[(\w+)\]", lambda m: regs[m.group(1)], l).replace("%=", "0").replace("%%", "%") for l in lines]
open("/tmp/asm_check.s", "w").write(".text\n" + "\n".join(out) + "\n")
r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-mc", "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", "/tmp/asm_check.o",
                    "/tmp/asm_check.s"], capture_output=True, text=True)
print("%d instructions, %d operand VGPRs" % (len(out), nv))
print(r.stderr[:4000] or "assembles")
