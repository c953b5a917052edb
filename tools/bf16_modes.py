"""Development aid: worst relative deviation of the bf16 engine from the fp32 reference golden at 7B depth, per VTG compensation mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as P
from oracle.gen_golden import problem_of

case = sys.argv[1] if len(sys.argv) > 1 else "full7b"
g = np.load(os.path.join(P.GOLD, f"{case}.npz"))
for dtype in ("bf16", "f16"):
    t = P._build(case, device_synth=True, dtype=dtype)
    for mode in ((None, "full") if dtype == "bf16" else (None,)):
        t.model.vtg_precise = mode                  # read by PairScorer at construction (inside _six_passes)
        t0 = time.time()
        w = P._worst_rel(P._six_passes(t, False), g)
        if "syn" in t.spec:
            sprob = problem_of(t.spec, t.dims, t.spec["syn"])
            ws = P._worst_rel(P._six_passes(t, False, prob=sprob, spec=t.spec["syn"], names=t.spec["syn"]["passes"], max_tokens=1 << 16), g, "SYN_")
            w.update({"SYN_" + k: v for k, v in ws.items()})
        print(f"[{case} {dtype} vtg_precise={mode}] " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()) + f"  ({time.time() - t0:.1f}s)", flush=True)
    t.model.engine.close()
