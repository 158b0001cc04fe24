"""Host-side profile of the imitation step (config 4): cProfile of 20 steps after warm-up + rocprof-free GPU split."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imitation_step import imitation_step
for _ in range(3):
    imitation_step()
ts = [imitation_step() for _ in range(10)]
print("fwd ms: %s" % " ".join("%.2f" % r["fwd_ms"] for r in ts))
print("bwd ms: %s" % " ".join("%.2f" % r["bwd_ms"] for r in ts))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    imitation_step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
