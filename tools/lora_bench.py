"""What keeping the LoRA adapters APART costs on the headline step (bench.py's 880-pair v2t VTG plan, 7B, seeded weights) and on reference-shaped
TVG plans: the same plans with no adapters (= the cost of merged adapters: merging is free per call) and with seeded non-zero adapters on every
q/k/v/o_proj, lm_head and projector Linear (blim_load_adapter), per kernel class.   python tools/lora_bench.py [--dtype f16|bf16] [--steps 6]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--vtg-precise", default="none")
    a = ap.parse_args()
    import torch
    import bench
    from blim_amd import checkpoint as CK
    from blim_amd import lora, synth
    from blim_amd.modeling import BlimModel
    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype=a.dtype)
    model.engine.init_synthetic_weights(0)
    model.vtg_precise = None if a.vtg_precise == "none" else a.vtg_precise
    plans = [(sc, pl) for sc, pl, _, _ in bench.build_step_plans(model, 0, 2, 55, 16)]
    tr = lora.synthetic_trainable(dims, 8, 47)

    def run(tag):
        for i in range(2):
            plans[i % 2][0].run(plans[i % 2][1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(a.steps):
            out = plans[i % 2][0].run(plans[i % 2][1])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        model.engine.timing_enable(True)
        plans[0][0].run(plans[0][1])
        rep = model.engine.timing_report()
        model.engine.timing_enable(False)
        cls = {k: round(v["ms"], 2) for k, v in rep.items() if v["calls"]}
        print(json.dumps({"case": tag, "dtype": a.dtype, "vtg_precise": a.vtg_precise, "ms_per_step": round(dt * 1e3, 2), "pairs_per_s": round(plans[0][1].n_pairs / dt, 1),
                          "classes_ms": cls, "finite": bool(torch.isfinite(out).all())}), flush=True)
        return dt

    for rnd in range(2):
        model.engine.clear_adapters()
        t_none = run("no adapters (= merged)")
        for n in CK.expected_adapters(dims):
            model.engine.load_adapter(n, tr[n + ":A"], tr[n + ":B"], 8, 32.0)
        t_ap = run("adapters apart")
        print(f"round {rnd}: adapters apart cost {100 * (t_ap / t_none - 1):+.2f} % on the step", flush=True)
    model.engine.close()


if __name__ == "__main__":
    main()
