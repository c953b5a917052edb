"""Development aid: the heavy7b fixture's TVG-type passes on an fp16 / bf16 engine under option toggles (precise_act, plain TVG), fused and literal."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as P
from blim_amd import synth, engine as E
from blim_amd.modeling import BlimModel
from oracle.gen_golden_heavy import SPEC7B, heavy_items
g = np.load(os.path.join(P.GOLD, "heavy7b.npz"))
dims = synth.ModelDims(**SPEC7B["dims"])
names = ("v2t_tvg", "t2v_tvg", "t2v_tvg_cpn", "v2t_vtg_cpn")
orig = E.Engine.set_precise
for dt in sys.argv[1:] or ["f16"]:
    model = BlimModel(dims, max_positions=1024, dtype=dt)
    model.engine.init_synthetic_weights(SPEC7B["wseed"])
    for name, arr in heavy_items(dims, SPEC7B["wseed"], only_changed=True):
        model.engine.load_weight(name, arr)
    prob = synth.make_problem(SPEC7B["pseed"], SPEC7B["n"], dims, tok_per_clip=SPEC7B["tok_per_clip"], text_len=SPEC7B["text_len"])
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    t = types.SimpleNamespace(spec=SPEC7B, dims=dims, model=model, prob=prob, dtype=dt, case="heavy7b")
    for label, act in (("default", None), ("precise_act=1", True), ("precise_act=0", False)):
        E.Engine.set_precise = (lambda self, on, embeds=False, mlp=True, act=None, _a=act: orig(self, on, embeds, mlp, _a if _a is not None else act))
        for lit in (False, True):
            w = P._worst_rel(P._six_passes(t, lit, names=names), g)
            print(dt, label, "literal" if lit else "fused", {k: float(f"{v:.2e}") for k, v in w.items()}, flush=True)
    E.Engine.set_precise = orig
    got = P._six_passes(t, False, names=("t2v_tvg_cpn",))["t2v_tvg_cpn"]; G = g["S_t2v_tvg_cpn"]; m = G != -100
    print("fused t2v_tvg_cpn entries:", np.round(got[m], 5).tolist()); print("reference:               ", np.round(G[m], 5).tolist())
    model.engine.close()
