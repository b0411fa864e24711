"""Experiment: cost of the neighbour gathers. Runs real steps, then ONE step in which every gather reads the lane's own
(coalesced) record instead of the neighbour's (SPH_EXP_SELF, experimental build only), with stage timing."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scenes
sc = scenes.liquid_box((50.0, 50.0, 50.0), (100, 100, 100), mask=0xffff)
h = scenes.hip_for(sc)
for it in range(10): h.step(it)
h.synchronize(); h.set_stage_timing(True)
for tag in ("normal", "self"):
    if tag == "self": os.environ["SPH_EXP_SELF"] = "1"
    h.reset_stage_times()
    h.step(10 if tag == "normal" else 11)
    h.synchronize()
    print(tag, {k: round(ms, 4) for k, (ms, n) in h.stage_times().items() if n})
