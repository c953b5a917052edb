"""Which MX format can the compensated modes' second pass run in?  (round 5)

The second walk over K carries x_lo = x - f32(x_hi) against a low-precision copy of W.  Today both are e4m3 (one E8M0 scale per 128 columns of x_lo, one per W
row).  gfx950 issues e2m3 / e2m1 operands at twice the e4m3 rate; whether their 32-block grids are accurate enough is measured here BEFORE a kernel is written:
BLIM_LO_EMULATE_A / BLIM_LO_EMULATE_W make the quantisers round every 32-block onto the e2m3 (1) or e2m1 (2) grid times the block's own power-of-two scale and
store the result as e4m3 -- which represents those values exactly -- so the existing e4m3 kernels compute what an fp6 / fp4 pass would.

Every call fully compensated (vtg_precise = tvg_precise = "full"), fused path, all six passes against the fp32 reference goldens at the real 7B configuration:
Gaussian weights (full7b) and the two trained-like sets (heavy7b, sink7b).  Prints one JSON line per (case, format).
"""
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_parity as T   # noqa: E402  (helpers only: _build, _six_passes, _worst_rel)
from blim_amd import synth   # noqa: E402
from blim_amd.modeling import BlimModel   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FORMATS = {"e4m3": (0, 0), "e2m3": (1, 1), "e2m3_a_only": (1, 0), "e2m3_w_only": (0, 1), "e2m1": (2, 2), "e2m1_a_e2m3_w": (2, 1)}


def build(case):
    if case in ("full7b", "deep"):
        return T._build(case, device_synth=True, dtype="f16")
    from oracle.gen_golden_heavy import CASES as HEAVY_CASES, heavy_items, heavy_weights
    spec = HEAVY_CASES[case]
    sink = bool(spec.get("sink", False))
    dims = synth.ModelDims(**spec["dims"])
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    if case.endswith("7b"):
        model.engine.init_synthetic_weights(spec["wseed"])
        for name, arr in heavy_items(dims, spec["wseed"], only_changed=True, sink=sink):
            model.engine.load_weight(name, arr)
    else:
        model.engine.load_weights(heavy_weights(dims, spec["wseed"], sink=sink))
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    return types.SimpleNamespace(spec=spec, dims=dims, model=model, prob=prob, dtype="f16", case=case)


def main():
    cases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["full7b", "heavy7b", "sink7b"]
    fmts = sys.argv[2].split(",") if len(sys.argv) > 2 else list(FORMATS)
    for case in cases:
        g = np.load(os.path.join(GOLD, f"{case}.npz"))
        base = None
        for fmt in fmts:
            a, w = FORMATS[fmt]
            os.environ["BLIM_LO_EMULATE_A"], os.environ["BLIM_LO_EMULATE_W"] = str(a), str(w)
            t0 = time.time()
            t = build(case)             # a fresh engine per format: the W copies are built on the first compensated call
            try:
                t.model.vtg_precise = "full"
                t.model.tvg_precise = "full"
                got = T._six_passes(t, False)
                worst = T._worst_rel(got, g)
                rms = {}
                for k, S in got.items():
                    G = g[f"S_{k}"]; m = G != -100.0
                    rms[k] = float(np.sqrt(np.mean(((S[m].astype(np.float64) - G[m]) / G[m]) ** 2)))
                if fmt == "e4m3":
                    base = got
                vs = {}
                if base is not None and fmt != "e4m3":
                    for k, S in got.items():
                        m = base[k] != -100.0
                        vs[k] = float(np.max(np.abs(S[m].astype(np.float64) - base[k][m]) / np.abs(base[k][m])))
            finally:
                t.model.engine.close()
            print(json.dumps({"case": case, "format": fmt, "worst_vs_fp32": {k: f"{v:.1e}" for k, v in worst.items()}, "rms_vs_fp32": {k: f"{v:.1e}" for k, v in rms.items()},
                              "worst_vs_e4m3": {k: f"{v:.1e}" for k, v in vs.items()}, "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
