"""Development aid: the `sink` / `heavy` fixtures' VTG passes on an fp16 engine with vtg_precise = none / full."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as P
from blim_amd import synth
from blim_amd.modeling import BlimModel
from oracle.gen_golden_heavy import CASES, heavy_weights
for case in sys.argv[1:] or ["sink"]:
    SPEC = CASES[case]
    g = np.load(os.path.join(P.GOLD, f"{case}.npz"))
    dims = synth.ModelDims(**SPEC["dims"])
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    model.engine.load_weights(heavy_weights(dims, SPEC["wseed"], sink=SPEC.get("sink", False)))
    prob = synth.make_problem(SPEC["pseed"], SPEC["n"], dims, tok_per_clip=SPEC["tok_per_clip"], text_len=SPEC["text_len"])
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    t = types.SimpleNamespace(spec=SPEC, dims=dims, model=model, prob=prob, dtype="f16", case=case)
    for mode in (None, "full"):
        model.vtg_precise = mode
        for lit in (False, True):
            w = P._worst_rel(P._six_passes(t, lit, names=("v2t_vtg", "v2t_vtg_cpn", "t2v_vtg")), g)
            print(case, "vtg_precise", mode, "literal" if lit else "fused", {k: float(f"{v:.2e}") for k, v in w.items()}, flush=True)
    model.engine.close()
