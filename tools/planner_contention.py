"""What does the HOST side of an 8-rank evaluation cost when eight ranks share one node's CPUs?  (/root/reference/retrieval_utils.py:207-262)

No 8-GPU node is available to the builder, and the strong-scaling prediction of bench.py plays the ranks one after another -- each with the whole host to itself.
This probe measures the part that emulation cannot see.  A rank's host work is: draining the loader, top-k, pooling the pairs of both directions, planning and
packing every super-batch (numpy + torch CPU tensors), uploading.  It is run here WITHOUT a GPU: `evaluation()` of the fixed-size job (N x N items, top-k, six
passes, pair ownership, `--shard W r`) over a PairScorer whose engine calls are replaced by zeros on the CPU -- every plan is built and packed exactly as in a real
run, nothing is computed.  Measured: each rank's share ALONE (the other CPUs idle), then all W shares CONCURRENTLY (what one node's CPUs see in a real job); with
--gpu-rank the concurrent run has rank 0 as a REAL evaluation on the GPU (blim_amd.main --shard W 0) next to W - 1 CPU-only planners, and that rank alone.

    python tools/planner_contention.py [--n 1000] [--topk 16] [--world 8] [--gpu-rank]
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n, topk, world, rank, seconds):
    import numpy as np
    import torch
    from blim_amd import distributed as D
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    D.limit_host_threads(world)                                   # what the drivers do: intra-op threads = usable CPUs / ranks of this node
    dims = synth.ModelDims()
    t0 = time.perf_counter()
    prob = synth.make_problem(1, n, dims, tok_per_clip=64, fast_video=True)
    loader = synth.ProblemLoader(prob, 64, video_dtype=torch.float16)
    t_data = time.perf_counter() - t0
    H = dims.hidden_size

    class Engine:                                                 # nothing computes: the planner only asks for shapes and modes
        can_precise, dtype, lo6, _vocab_key = True, "f16", True, None
        def set_video_vocab(self, v): self._vocab_key = id(v)
        def set_precise(self, *a, **k): pass
        def set_option(self, *a, **k): pass

    class Model:
        def __init__(self):
            self.engine, self.device, self.dims, self.dtype = Engine(), torch.device("cpu"), dims, torch.float16
            self.vtg_precise, self.tvg_precise, self.tokenizer_model_max_length = None, "attn", None
        def vtg_mode(self): return None
        def tvg_mode(self): return "attn"
        def tvg_resolved(self): return True
        def eval(self): return self
        def set_tvg_prefix_length(self, k): self.tvg_prefix_length = k
        def set_video_vocab(self, v): pass
        def clear_cache(self): pass
        def project_many(self, feats, tvg):                       # [clips * T, H] rows per video (VTG) or [clips, 2 H] (TVG rows are hi | lo)
            clips, T = feats[0].shape[-3], feats[0].shape[-2]
            return [torch.zeros((clips, 2 * H) if tvg else (clips * T, H), dtype=torch.float16) for _ in feats]

    class PlanOnly(RU.PairScorer):
        def run(self, plan):                                      # the engine call: every array of the plan has been built, packed and "uploaded" by now
            self.exec_tokens += plan.n_tokens
            return torch.zeros(plan.n_pairs, dtype=torch.float32)

    m = Model()
    ddp = types.SimpleNamespace(module=m, eval=lambda: None)
    nz = lambda a: np.where(a == 0, np.float32(1e-6), a)
    args = types.SimpleNamespace(topk=topk, num_clips=dims.num_clips, cpn=True, resume="x", eval=True, dataset="MSRVTT", batch_size_eval=16,
                                 iv2_scores={"v2t": torch.from_numpy(nz(prob.v2t_sims)), "t2v": torch.from_numpy(nz(prob.t2v_sims))}, max_tokens=32768, dedup=True,
                                 shard=(world, rank))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    args._scorer = PlanOnly(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v).half() for v in prob.video], torch.from_numpy(prob.video_vocab),
                            torch.from_numpy(prob.tvg_video_labels), dims.num_clips, max_tokens=32768)
    # the evaluation is repeated -- at least three times, and for `seconds` of wall time, so that concurrently started processes (whose start-up and data building
    # take different times) overlap for most of their repetitions -- and the MEDIAN repetition is reported
    times, t_start = [], time.perf_counter()
    while len(times) < 3 or time.perf_counter() - t_start < seconds:
        args._scorer._vfeat.clear(); args._scorer.exec_tokens = 0
        t0 = time.perf_counter()
        RU.evaluation(ddp, loader, torch.device("cpu"), tok, args)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    st = args._eval_stats
    print(json.dumps({"rank": rank, "seconds": round(dt, 3), "repetitions": len(times), "fastest": round(min(times), 3), "data_seconds": round(t_data, 3), "pairs_scored": st.get("pairs_scored"), "tokens_planned": args._scorer.exec_tokens,
                      "torch_threads": torch.get_num_threads(), "host_marks": st.get("host_marks")}), flush=True)


def spawn(n, topk, world, rank, seconds=0.0):
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(rank), "--n", str(n), "--topk", str(topk), "--world", str(world), "--seconds", str(seconds)],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))


def collect(p):
    out, err = p.communicate(timeout=1800)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f"worker failed: {err[-1500:]}")
    return json.loads(lines[-1])


def gpu_rank(n, topk, world):
    import re
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "blim_amd.main", "--eval", "--synthetic", str(n), "--synthetic_7b", "--cpn", "--resume", "x", "--topk", str(topk), "--shard", str(world), "0",
                        "--vtg_precise", "none", "--tvg_precise", "attn", "--output_dir", "/tmp/planner_contention_out"], cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"in ([0-9.]+)s = (\d+) pairs/s per process", r.stdout)
    return {"evaluation_seconds": float(m.group(1)), "process_seconds": round(time.perf_counter() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000); ap.add_argument("--topk", type=int, default=16); ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--worker", type=int, default=None); ap.add_argument("--gpu-rank", action="store_true"); ap.add_argument("--seconds", type=float, default=0.0)
    a = ap.parse_args()
    if a.worker is not None:
        return worker(a.n, a.topk, a.world, a.worker, a.seconds)
    from blim_amd import distributed as D
    cpus = D.usable_cpus() if hasattr(D, "usable_cpus") else os.cpu_count()
    alone = [collect(spawn(a.n, a.topk, a.world, r)) for r in range(a.world)]
    first = 1 if a.gpu_rank else 0
    gpu_alone = gpu_rank(a.n, a.topk, a.world) if a.gpu_rank else None
    procs = [spawn(a.n, a.topk, a.world, r, seconds=60.0 if a.gpu_rank else 20.0) for r in range(first, a.world)]      # (the GPU rank's evaluation falls inside the planners' window)
    gpu_busy = gpu_rank(a.n, a.topk, a.world) if a.gpu_rank else None
    together = [collect(p) for p in procs]
    print(f"# planner contention: N = {a.n}, top-{a.topk}, six passes, pair ownership, world {a.world}; host CPUs usable by this job: {cpus}; torch intra-op threads per rank: {alone[0]['torch_threads']}\n")
    print("| rank | host side alone (s) | ... with the other ranks' host sides running (s) | ratio | packed tokens planned |\n|---|---|---|---|---|")
    for r in range(first, a.world):
        t = together[r - first]
        print(f"| {r} | {alone[r]['seconds']:.2f} | {t['seconds']:.2f} | {t['seconds'] / alone[r]['seconds']:.2f} | {t['tokens_planned']} |")
    print(f"\nslowest host side alone {max(x['seconds'] for x in alone):.2f} s, under contention {max(x['seconds'] for x in together):.2f} s "
          f"(building the synthetic data set, outside the evaluation: {alone[0]['data_seconds']:.1f} s per rank)")
    if a.gpu_rank:
        print(f"\nrank 0 as a REAL evaluation on the GPU (blim_amd.main --shard {a.world} 0, plain VTG / attn TVG): {gpu_alone['evaluation_seconds']:.2f} s alone, "
              f"{gpu_busy['evaluation_seconds']:.2f} s with {a.world - 1} CPU-only planners beside it ({gpu_busy['evaluation_seconds'] / gpu_alone['evaluation_seconds']:.2f}x)")
    print("\n" + json.dumps({"alone": alone, "together": together, "gpu_alone": gpu_alone, "gpu_with_planners": gpu_busy}))


if __name__ == "__main__":
    main()
