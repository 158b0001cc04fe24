"""Verdict r04 item 7: what a step of the (8,4) and (4,4) solves costs in executed instructions and wavefront cycles (PMC passes
`s84_mix`, `s44_mix`, `s82_mix` of scripts/gpu_profile_r05.sh B, read from pmc_summary.txt), beside the headline stream's (8,2),
and what the asked fractions of the HBM roof would need.  One wavefront per SIMD (B = 4096 at 4 trajectories per wavefront = 1,024
wavefronts): a kernel's time is its wavefronts' cycles.
    python scripts/shape_floor.py gpurun_out/r05/pmc_summary.txt"""
import re
import sys

T, B = 50, 4096
SHAPES = {"s82_mix": (8, 2, None), "s84_mix": (8, 4, 0.45), "s44_mix": (4, 4, 0.40)}


def sections(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r"== (\w+) ", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"(\w+)\s+n=\s*(\d+)\s+mean=([0-9.e+]+)\s*$", ln)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
    return out


def main():
    sec = sections(sys.argv[1])
    print("# scripts/shape_floor.py: B = %d, T = %d, fused solve (sweep + rollout); counters per LAUNCH / wavefronts / T" % (B, T))
    print("# SQ_WAVE_CYCLES counts in units of 4 cycles (checked against s_memtime stamps and kernel durations)")
    ref = None
    for name, (nx, nu, target) in SHAPES.items():
        c = sec.get(name)
        if not c or "SQ_WAVES" not in c:
            print("%s: no counters" % name)
            continue
        ns = nx + nu
        waves = c["SQ_WAVES"]
        kinds = ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
        per = {k: c.get(k, 0.0) / waves / T for k in kinds}
        per["SQ_INSTS_VALU"] -= per["SQ_INSTS_MFMA"]          # (the VALU count includes the matrix instructions)
        total = sum(per.values())
        cyc = 4.0 * c["SQ_WAVE_CYCLES"] / waves / T
        bytes_step = 4 * (ns * ns + ns + nx * ns + nx + ns)
        print("(%d,%d): %.0f wavefronts; per step of a wavefront (4 trajectories): %.0f instructions = %.0f vector + %.0f matrix + %.0f "
              "scalar + %.0f LDS + %.0f loads + %.0f stores; %.0f cycles = %.2f cycles per instruction; %d algorithmic bytes per "
              "timestep-solve" % (nx, nu, waves, total, per["SQ_INSTS_VALU"], per["SQ_INSTS_MFMA"], per["SQ_INSTS_SALU"],
                                  per["SQ_INSTS_LDS"], per["SQ_INSTS_VMEM_RD"], per["SQ_INSTS_VMEM_WR"], cyc, cyc / total, bytes_step))
        if ref is None:
            ref = (total, cyc, bytes_step)
        if target is not None:
            # time at the target fraction -> cycles per step a wavefront may take at the clock the counters imply
            us_now = None
            t_target = bytes_step * B * T / (target * 8e12)
            # the clock: cycles of a wavefront's whole life / kernel time is not in the counters; use 2.1 GHz (s_memtime stamps, HISTORY 5.3)
            budget = t_target * 2.1e9 / T
            print("        %.2f of the HBM roof = %.1f us per solve = %.0f cycles per step at 2.1 GHz: at this stream's %.2f cycles per "
                  "instruction %.0f instructions a step (now %.0f: x%.2f); at the headline stream's %.2f, %.0f" % (
                      target, t_target * 1e6, budget, cyc / total, budget / (cyc / total), total, total / (budget / (cyc / total)),
                      ref[1] / ref[0], budget / (ref[1] / ref[0])))


if __name__ == "__main__":
    main()
