"""Benchmark of the likelihood-scoring hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one engine pass over one device-resident super-batch of synthetic (query, candidate) pairs of
BASELINE.json's headline shape: Qwen2-7B dims, 96 video tokens + 32 text tokens per pair, top-16 candidates per
video query (v2t VTG pass).  The 96-token video prefix of a query is computed once and its K/V reused by the 16
candidates (identical scores, fewer executed FLOPs); `roofline` is priced on the FLOPs actually executed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, I, V, LAYERS = 3584, 18944, 152064, 28
FLOP_TOKEN_LAYER = 2 * H * (H + 2 * 512) + 2 * H * H + 6 * H * I      # 466,092,032
FLOP_HEAD_ROW = 2 * H * V                                           # 1,089,994,752
PEAK_BF16_TFLOPS = 2500.0                                           # MI355X dense bf16/f16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0                                            # dense fp8 (block-scaled MFMA), same table
PEAK_FP6_TFLOPS = 10000.0                                           # dense fp6 / fp4 (block-scaled MFMA; gfx950 issues fp6 at the fp4 rate), same table


def f_pair(L, t_lab):
    """Algorithmic FLOPs of one pair scored on its own."""
    return LAYERS * (FLOP_TOKEN_LAYER * L + 2 * H * L * L) + FLOP_HEAD_ROW * t_lab


def measured_traffic(kernel_class, dtype, compensated=False):
    """HBM-side bytes per launch of a kernel class from the committed PMC summary that profiles/CURRENT.json names for this kind of run ("plain" / "full": written by
    tools/collect_profiles.sh next to the bench command it profiled; made by tools/parse_profiles.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes); None if absent."""
    epi = {"gemm_qkv_rope": 3, "gemm_o_resid": 2, "gemm_down_resid": 2, "gemm_gateup_swiglu": 4, "lm_head_lse": 5}.get(kernel_class)
    try:
        cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
        f = cur.get("traffic", {}).get(("full" if compensated else "plain") + "_" + dtype)
        if epi is None or not f:
            return None
        d = json.load(open(os.path.join(ROOT, "profiles", f)))
        dt = dict(bf16=0, f16=1, f8=2)[dtype]
        split = "true" if (compensated and epi in (3, 4)) else "false"
        for name in (f"void gemm_kernel<{epi}, {dt}, {split}, {'true' if compensated and dtype == 'f16' else 'false'}>(GemmParams)",):
            if name in d:
                return d[name]["hbm_bytes_per_launch"], os.path.join("profiles", f)
        return None
    except Exception:
        return None


def cpu_baseline(seconds_budget=30.0):
    """The CPU restatement (oracle/torch_port.py: fp32, torch intra-op threads = the host's physical cores) on a bounded sample of
    the same workload, as the reference runs it (no prefix sharing): pairs x 128 tokens through ALL 28 decoder layers at 7B width
    -- 28 layer executions on one seeded weight set (the timing does not depend on the values; 28 distinct sets are 30 GB of
    fp32) -- then lm_head + log-softmax on the 32 label rows of every pair.  The pair count is chosen from the first layer's
    time so that the leg stays within `seconds_budget`."""
    import torch
    from oracle import torch_port as TP
    n_max = TP.usable_cores()
    prev_thr = torch.get_num_threads()
    try:
        g = torch.Generator().manual_seed(0)
        nh, nkv, hd = 28, 4, 128
        rnd = lambda *shape: torch.randn(shape, generator=g) * 0.02
        P = "layers.0."
        w = {P + "input_norm": torch.ones(H), P + "post_norm": torch.ones(H), P + "q_proj.w": rnd(nh * hd, H), P + "q_proj.b": rnd(nh * hd),
             P + "k_proj.w": rnd(nkv * hd, H), P + "k_proj.b": rnd(nkv * hd), P + "v_proj.w": rnd(nkv * hd, H), P + "v_proj.b": rnd(nkv * hd),
             P + "o_proj.w": rnd(H, H), P + "gate_proj.w": rnd(I, H), P + "up_proj.w": rnd(I, H), P + "down_proj.w": rnd(H, I)}
        lm = rnd(V, H)
        L, T_LAB = 128, 32
        cos, sin = TP.rope_tables(hd, 1e6, L)
        layer = lambda x, am: TP.decoder_layer(x, w, P, am, cos, sin, nh, nkv, 1e-6)
        with torch.no_grad():
            x2 = rnd(2, L, H); am2 = TP.additive_mask(torch.ones(2, L), L)
            # thread count: the fastest of {usable cores, /2, /4, ...} on a 2-pair probe layer (more threads than the memory system
            # or the cgroup quota feeds run slower, and the count reported in `cores` must be the one that was used)
            cands, probes = sorted({max(1, n_max >> k) for k in range(0, 5)}, reverse=True), {}
            for n_try in cands:
                torch.set_num_threads(n_try)
                layer(x2, am2)                               # thread pool / allocator warm-up
                t0 = time.time(); layer(x2, am2); probes[n_try] = (time.time() - t0) / 2     # seconds per pair-layer
            n_thr = min(probes, key=probes.get)
            torch.set_num_threads(n_thr)
            t_probe = probes[n_thr]
            B = int(max(2, min(16, seconds_budget * 0.8 / (LAYERS * t_probe + 1e-9))))
            x = rnd(B, L, H); am = TP.additive_mask(torch.ones(B, L), L)
            t0 = time.time()
            for _ in range(LAYERS):
                x = layer(x, am)
            t_layers = time.time() - t0
            rows = TP.rms_norm(x, torch.ones(H), 1e-6)[:, L - T_LAB:].reshape(-1, H)
            labels = torch.randint(0, V, (rows.shape[0],), generator=g)
            t0 = time.time(); lp = TP.label_logprobs(rows, lm, labels); t_head = time.time() - t0
            assert torch.isfinite(lp).all()
        t_batch = t_layers + t_head
    finally:
        torch.set_num_threads(prev_thr)
    tiny = tiny_config_evaluation()
    return {"value": round(B / t_batch, 4), "unit": "pairs/s", "cores": n_thr, "kind": "port", "tiny_config_evaluation": tiny,
            "sample": f"torch-CPU fp32 port of the oracle, {n_thr} intra-op threads (fastest of {cands} on a probe layer; {TP.physical_cores()} physical / {os.cpu_count()} logical cores, "
                      f"{n_max} usable by this process): {B} pairs x 128 tokens through "
                      f"all 28 decoder layers at 7B width (28 executions of one weight set, no extrapolation) = {t_layers:.2f}s, lm_head + log-softmax on "
                      f"{B * T_LAB} label rows = {t_head:.2f}s; no prefix sharing (the reference's own batching)"}


def tiny_config_evaluation():
    """One complete v2t + t2v VTG/TVG evaluation of a tiny 2-layer configuration (6 videos x 6 texts, top-4), run by the numpy
    oracle on the host cores and by the engine on the GPU in this same process; the two score sets are compared (1e-3)."""
    import types
    import torch
    from oracle import blim_oracle as O
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    from blim_amd.modeling import BlimModel, DDPLike
    d = dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
    dims = synth.ModelDims(**d)
    w = synth.synthetic_weights(dims, 3)
    n, topk = 6, 4
    prob = synth.make_problem(4, n, dims, tok_per_clip=8, text_len=(3, 9))
    om = O.OracleModel(O.OracleConfig(**d), w); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    ot = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    passes = [(True, "vtg"), (True, "tvg"), (False, "vtg"), (False, "tvg")]
    t0 = time.time()
    want = []
    for qv, ft in passes:
        ids, lab, msk = ov if ft == "vtg" else ot
        fn = O.compute_v2t_scores_x if qv else O.compute_t2v_scores_x
        want.append(fn(np.full((n, n), -100.0, np.float32), prob.v2t_sims if qv else prob.t2v_sims, 0, ids, msk, lab, prob.video, prob.video_vocab,
                       prob.tvg_video_labels, om, topk, 3, dims.num_clips, ft, False))
    t_cpu = time.time() - t0
    model = BlimModel(dims, max_positions=512, dtype="f16")
    model.engine.load_weights(w)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    scorer = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                           torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
    worst, t_gpu = 0.0, 0.0
    for rep in range(2):                                   # second repetition is timed (first one pays allocation / planning warm-up)
        torch.cuda.synchronize(); t0 = time.time()
        got = []
        for qv, ft in passes:
            pairs = RU._topk_pairs(torch.from_numpy(prob.v2t_sims if qv else prob.t2v_sims), 0, topk, qv)
            got.append((pairs, scorer.vtg(pairs) if ft == "vtg" else scorer.tvg(pairs)))
        torch.cuda.synchronize(); t_gpu = time.time() - t0
    for (qv, ft), W, (pairs, sc) in zip(passes, want, got):
        r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
        worst = max(worst, float(np.max(np.abs(sc - W[r, c]) / np.abs(W[r, c]))))
    model.engine.close()
    return {"pairs": 4 * n * topk, "oracle_s": round(t_cpu, 3), "engine_s": round(t_gpu, 4), "max_rel_diff": float(f"{worst:.2e}"), "agree_1e-3": worst < 1e-3}


def strong_scaling(model, world, rank, dev, n=1000, topk=16, emulate=8, pg=False, tvg_precise="auto"):
    """The fixed-size job north_star's scaling clause names: ONE complete evaluation of an MSRVTT-1kA-shaped test set (N = 1000 videos and
    texts, top-16 re-rank, all six passes of the fine-tuned + CPN flow = 96,000 (query, candidate) pairs, reference-shaped rows with
    4 x 64 = 256 video tokens, the full 7B model) through blim_amd.retrieval_utils.evaluation -- data handling, planning, pair ownership
    and the RCCL all-gather of the score blocks all INSIDE the timed region (/root/reference/retrieval_utils.py:169-281).  Total work is
    fixed, so seconds(W = 1) / seconds(W) is the strong-scaling speed-up.  At W = 1 the same process also plays each rank of an
    `emulate`-process job in turn (`shard`: the rank's own blocks, no merge; the merge is one all-gather of < 1 MB) and the slowest rank's
    time is the predicted `emulate`-GPU time."""
    import types
    import torch
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    from blim_amd.modeling import DDPLike
    dims = model.dims
    pg = pg or world > 1                                                     # a process group is up (world 1 + BLIM_FORCE_COLLECTIVE: the RCCL smoke run)
    peak = (PEAK_FP8_TFLOPS if model.engine.dtype == "f8" else PEAK_BF16_TFLOPS) * 1e12
    prob = synth.make_problem(1, n, dims, tok_per_clip=64, fast_video=True)
    loader = synth.ProblemLoader(prob, 64, video_dtype=torch.float16)        # fp16 per-video feature tensors, as the reference's files hold
    tokenizer = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    nz = lambda a: np.where(a == 0, np.float32(1e-6), a)
    ddp = DDPLike(model)

    peers = {}                                                               # TVG clip features of the W = 1 run: what the all-gather of a real job delivers to each rank
    agreed = {}                                                              # ... and the numeric modes the W = 1 run's calibration chose: what a real job's ranks agree on

    def run(shard):
        args = types.SimpleNamespace(topk=topk, num_clips=dims.num_clips, cpn=True, resume="x", eval=True, dataset="MSRVTT", batch_size_eval=16,
                                     iv2_scores={"v2t": torch.from_numpy(nz(prob.v2t_sims)), "t2v": torch.from_numpy(nz(prob.t2v_sims))},
                                     max_tokens=32768, dedup=True, shard=shard, keep_tvg_feats=shard is None, peer_tvg_feats=peers if shard is not None else None,
                                     agreed_modes=agreed if (shard is not None and agreed) else None)
        model.clear_cache()
        model.tvg_precise = tvg_precise if model.engine.can_precise else "full"   # "auto" (the driver's default): calibrated by every run, INSIDE its timed region
        if pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t2v, v2t = RU.evaluation(ddp, loader, dev, tokenizer, args)          # ends with the matrices on the host: synchronises
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = args._eval_stats
        if shard is None:
            peers.update(getattr(args, "_tvg_feats", {}))
            for kind in ("vtg", "tvg"):          # (mode, did the decision need the confirmation sample): an emulated rank measures its block of the same stages and adopts the mode
                if f"{kind}_precise_table" in st:
                    agreed[kind] = (st[f"{kind}_precise"], any("confirm" in v for v in st[f"{kind}_precise_table"].values()))
        st["executed_flops_job"] = st.get("executed_flops", 0.0)
        st["executed_flops_lo6_job"] = st.get("executed_flops_lo6", 0.0)
        if pg:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
            f = torch.tensor([st.get("executed_flops", 0.0), st.get("executed_flops_lo6", 0.0)], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.SUM)      # the whole job's executed FLOPs (every rank's own calls)
            st["executed_flops_job"], st["executed_flops_lo6_job"] = float(f[0].item()), float(f[1].item())
        return dt, st, (t2v, v2t)

    # seconds the executed FLOPs would take at the dense peaks: the e2m3 second pass of compensated calls (engine option "precise_lo6") at the fp6 peak, the rest
    # at this engine's own (16-bit or fp8) peak
    at_peak = lambda total, f6: ((total - f6) / peak + f6 / (PEAK_FP6_TFLOPS * 1e12))
    run((4 * max(world, emulate), rank))                                      # warm-up: workspaces, allocator, first-call costs (a small share)
    dt, st, (t2v, v2t) = run(None)
    ok = all(np.isfinite(m).all() for d in (t2v, v2t) for m in d.values())
    pairs = 6 * n * min(topk, n)
    out = {"workload": f"one evaluation, N = {n} videos x {n} texts, top-{topk}, six passes (VTG/TVG likelihoods both directions + both CPN priors) = {pairs} "
                       "(query, candidate) pairs, reference-shaped rows (256 video tokens), full 7B; host planning, pair ownership and the RCCL "
                       "all-gather of score blocks inside the timed region",
           "scaling": "strong", "world": world, "seconds": round(dt, 3), "pairs_per_s": round(pairs / dt, 1), "pairs": pairs,
           # the fixed job's roofline fraction: GEMM FLOPs executed by ALL ranks' engine calls (RU.executed_flops: the per-token constants on the token
           # counts launched, compensated TVG calls counted twice, last-layer pruning subtracted) / wall time of the job / (world x dense peak)
           "executed_tflop_job": round(st["executed_flops_job"] / 1e12, 1),
           "executed_tflops_per_gpu": round(st["executed_flops_job"] / dt / 1e12 / world, 1),
           "executed_tflop_job_e2m3_pass": round(st["executed_flops_lo6_job"] / 1e12, 1),
           "frac_mfma_peak": round(at_peak(st["executed_flops_job"], st["executed_flops_lo6_job"]) / (dt * world), 4),
           "tvg_precise": f"{tvg_precise} -> {st['tvg_precise']}" if "tvg_precise" in st else getattr(model, "tvg_precise", "full"),
           "pairs_scored_rank0": st["pairs_scored"], "finite": bool(ok), "host_marks_rank0": st["host_marks"]}
    if world == 1 and emulate > 1:
        per_rank, per_rank_frac, n_have, n_same = [], [], 0, 0
        for r in range(emulate):
            d_r, st_r, (t2v_r, v2t_r) = run((emulate, r))
            per_rank.append(round(d_r, 3))
            for whole, part in ((t2v, t2v_r), (v2t, v2t_r)):                     # what the rank scored is what the one-process job scored there (entries it does not own: -100)
                for k_, m_ in part.items():
                    if k_ != "internvideo2" and k_ in whole and m_.shape == whole[k_].shape:
                        have = m_ != -100.0
                        n_have += int(have.sum()); n_same += int((m_[have] == whole[k_][have]).sum())
            per_rank_frac.append(round(at_peak(st_r.get("executed_flops", 0.0), st_r.get("executed_flops_lo6", 0.0)) / d_r, 4))
            if r == 0:
                out["emulated_rank0_host_marks"] = st_r["host_marks"]
        out.update({"emulated_world": emulate, "emulated_rank_seconds": per_rank, "predicted_seconds": max(per_rank),
                    "predicted_speedup": round(dt / max(per_rank), 2), "emulated_rank_frac_mfma_peak": per_rank_frac,
                    "emulated_entries": n_have, "emulated_entries_bit_equal_to_world_1": n_same,
                    "predicted_note": f"slowest of the {emulate} ranks' own shares run one after another on this GPU; each projects the TVG clip features of its own video "
                                      "block and takes the other blocks' from the W = 1 run (PairScorer.adopt_tvg_feats), as a real job's all-gather delivers them; that "
                                      "all-gather (57 MB) and the merge (one all-gather of < 1 MB) are not on an emulated rank's clock"})
    return out


def build_step_plans(model, rank, n_plans, Q, K):
    """The device-resident super-batches the timed steps run: plan `pi` of rank `rank` = the v2t VTG pass over synthetic problem
    1000 + 17 * rank + pi (Q videos and texts of the headline shape: 96 video + 32 text tokens; top-K texts per video query).
    Returns [(scorer, plan, problem, pairs)].  tests/test_gpu_parity.py scores plan 0 of rank 0 and compares the query rows that
    tests/golden/full7b_bench.npz holds (the reference's own loops on the same problem) -- the benched batch itself meets the oracle."""
    import torch
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    from blim_amd.modeling import DDPLike
    dims = model.dims
    tok = type("T", (), {"pad_token_id": synth.PAD_ID})()
    out = []
    for pi in range(n_plans):
        prob = synth.make_problem(1000 + 17 * rank + pi, Q, dims, tok_per_clip=24, text_len=(32, 32), reference_layout=False)
        Tt = lambda rows: [torch.from_numpy(r) for r in rows]
        vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
        tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
        scorer = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                               torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips, max_tokens=1 << 20)
        pairs = RU._topk_pairs(torch.from_numpy(prob.v2t_sims), 0, K, True)
        (plan,) = scorer.plan_vtg(pairs)
        out.append((scorer, plan, prob, pairs))
    return out


def launcher_command(gpus: int, argv, port: int = 0):
    """`python bench.py --gpus N` without a launcher around it: the command of the child that runs the ranks
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 -- the reference's env:// init, util/misc.py:199-229)."""
    if not port:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=55, help="video queries per step per GPU (55 x 592 packed tokens = 32,560 -> 128 row tiles of 256)")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f8"],
                    help="compute format: f16 (default: the engine's default and the reference's own autocast dtype; the one mode in which every pass "
                         "kind holds 1e-3 against the fp32 reference at all 28 layers of the 7B configuration, tests/test_gpu_parity.py::test_depth_*), "
                         "bf16 (same MFMA rate, ~3 %% faster under the power limit, but 1 - 2e-3 on the VTG scores at depth: a non-parity mode), "
                         "f8 (separate mode, deviations reported)")
    ap.add_argument("--topk", type=int, default=16)
    ap.add_argument("--vtg-precise", default="none", choices=["none", "full"],
                    help="compensated (hi + lo) activations on the benched VTG calls: none (default; fp16 holds 1e-3 without on these weights), full = what weights with a "
                         "trained checkpoint's massive activations need (fp16: second pass on the e2m3 MFMA) and the bf16 PARITY mode (2x GEMM flops; what "
                         "`--dtype bf16` needs to hold 1e-3 at 7B depth: tests/test_gpu_parity.py::test_depth_*)")
    ap.add_argument("--tvg-precise", default="auto", choices=["auto", "attn", "full"],
                    help="strong-scaling leg only (the headline step is a VTG pass): how much of the TVG calls' MLP branch runs compensated; auto = measured on the "
                         "job's own pairs inside the timed region, as main.py's default does (blim_amd/retrieval_utils.py: PairScorer.calibrate_tvg)")
    ap.add_argument("--second-pass", default=None, choices=["e2m3", "16bit"], help="second walk over K of the compensated calls (default: e2m3 on fp16 engines, 16bit on bf16 engines; "
                                                                                   "bf16 + e2m3 is the fast form of the bf16 engine's compensated mode)")
    ap.add_argument("--no-compensated", action="store_true", help="skip the extra timing of the same step with fully compensated VTG calls (reported as `compensated_mode`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the fixed-size N = 1000 evaluation (strong-scaling leg)")
    ap.add_argument("--strong-only", action="store_true", help="only the strong-scaling leg (development aid; prints that object alone)")
    ap.add_argument("--strong-n", type=int, default=1000)
    ap.add_argument("--leg-timeout", type=int, default=420, help="seconds after which the legs behind the timed steps (strong scaling, compensated mode, CPU baseline) are given up: "
                                                                 "rank 0 prints the headline line with the timeout recorded in it (0 = no watchdog)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # Invoked as plain `python bench.py --gpus N`: start the N ranks as a CHILD process group and exit with its code.  This
        # parent never touches the GPU (no torch import, no HIP call), so nothing is exec'ed from a GPU-initialised process.
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(launcher_command(a.gpus, sys.argv[1:]), env=env))

    import torch
    from blim_amd import distributed as D
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    from blim_amd.modeling import BlimModel, DDPLike

    # BLIM_FORCE_COLLECTIVE=1 with RANK / WORLD_SIZE = 0 / 1 in the environment: a process group (backend nccl = RCCL) at world size 1, and every
    # collective below runs -- the one-GPU smoke run of the multi-GPU path (tests/test_main_driver.py)
    forced = os.environ.get("BLIM_FORCE_COLLECTIVE", "0") == "1" and "RANK" in os.environ
    rank, world, local = D.init_distributed_mode() if (a.gpus > 1 or forced) else (0, 1, 0)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    pg = world > 1 or (forced and D.is_dist_avail_and_initialized())
    D.limit_host_threads(world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype=a.dtype)
    model.engine.init_synthetic_weights(0)                       # torch seed 0 of BASELINE.md -> engine seed 0
    model.vtg_precise = None if (a.vtg_precise == "none" or a.dtype == "f8") else a.vtg_precise
    if a.second_pass and model.engine.can_precise:
        model.second_pass = a.second_pass
    if a.strong_only:
        ss = strong_scaling(model, world, rank, dev, n=a.strong_n, topk=a.topk, pg=pg, tvg_precise=a.tvg_precise)
        if rank == 0:
            print(json.dumps({"strong_scaling": ss}), flush=True)
        if pg:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    Q, K = a.queries, a.topk
    n_plans = max(1, min(3, a.steps))
    plans = [(sc, pl) for sc, pl, _, _ in build_step_plans(model, rank, n_plans, Q, K)]
    plan0 = plans[0][1]
    n_pairs, n_tok, n_rows = plan0.n_pairs, plan0.n_tokens, plan0.n_rows
    model.engine.reserve(n_tok, n_rows)

    def step(i):
        sc, pl = plans[i % n_plans]
        return sc.run(pl)

    for i in range(a.warmup):
        step(i)
    if pg:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [step(i) for i in range(a.steps)]
    mine = torch.stack(outs)                                      # [steps, pairs] score rows of this rank
    if pg:
        gathered = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(gathered, mine)              # RCCL all-gather of the score rows (north_star)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if pg:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        assert torch.equal(gathered[rank], mine), "all_gather returned other rows than this rank sent"
    assert torch.isfinite(mine).all(), "non-finite scores"

    # ---- per-kernel timing (HIP events on the launch stream) of one more step, for the roofline object
    model.engine.timing_enable(True)
    step(0)
    rep = model.engine.timing_report()
    model.engine.timing_enable(False)

    vtg_headline = model.vtg_precise                            # (the compensated leg below switches the model's mode for its own timing)

    def emit(ss, comp, cpu=True):
        """Rank 0's ONE JSON line (everything in it but `strong_scaling`, `compensated_mode` and `cpu_baseline` was measured above)."""
        if rank == 0:
            total_pairs = n_pairs * a.steps * world
            value = total_pairs / dt
            # executed GEMM FLOPs of one step (attention excluded, < 1 %): the per-token constants on the packed tokens, the compensated modes' doubled GEMMs,
            # MINUS the last layer's o_proj / MLP on the rows nobody reads (engine option prune_last: 4,400 of the 32,560 tokens of a step are video-prefix rows)
            exec_flops_step = RU.executed_flops(dims, n_tok, n_rows, "vtg", vtg_headline, prune=model.engine.dtype != "f8")
            assert vtg_headline is not None or model.engine.dtype == "f8" or \
                abs(exec_flops_step - (LAYERS * FLOP_TOKEN_LAYER * n_tok + FLOP_HEAD_ROW * n_rows - (FLOP_TOKEN_LAYER - 2 * H * (H + 2 * 512)) * (n_tok - n_rows))) < 1e6
            dom = max((k for k in rep if rep[k]["flops"] > 0), key=lambda k: rep[k]["ms"])
            d = rep[dom]
            ach = d["flops"] / d["calls"] / (d["ms"] / d["calls"] * 1e-3) / 1e12
            peak = PEAK_FP8_TFLOPS if model.engine.dtype == "f8" else PEAK_BF16_TFLOPS
            # compensated mode on an fp16 engine (option "precise_lo6", default): the second walk over K runs on the e2m3 MFMA -- those flops are priced at the fp6
            # peak, the rest at the 16-bit one: peak_mixed = flops / (flops16 / P16 + flops6 / P6).  Plain mode (the headline): share 0, nothing changes.
            lo6 = bool(getattr(model.engine, "lo6", False))
            lo6_step = RU.lo6_pass_flops(dims, n_tok, n_rows, "vtg", vtg_headline, prune=True) if lo6 else 0.0
            mixed = lambda total, f6: total / ((total - f6) / PEAK_BF16_TFLOPS + f6 / PEAK_FP6_TFLOPS) if total > 0 else PEAK_BF16_TFLOPS
            peak_step = mixed(exec_flops_step, lo6_step) if model.engine.dtype != "f8" else peak
            dom_lo6 = 0.5 if (lo6_step > 0 and rep[dom]["flops"] > 0 and dom != "attention") else 0.0       # fully compensated: half of every GEMM's flops are its second pass
            if model.engine.dtype != "f8":
                peak = mixed(1.0, dom_lo6)
            tr = measured_traffic(dom, model.engine.dtype, compensated=vtg_headline == "full")
            es = 1 if model.engine.dtype == "f8" else 2
            alg_bytes = {"gemm_gateup_swiglu": n_tok * H * es + 2 * I * H * es + n_tok * I * 2, "gemm_down_resid": n_tok * I * es + H * I * es + 2 * n_tok * H * 4,
                         "gemm_qkv_rope": n_tok * H * es + 4608 * H * es + n_tok * 4608 * 2, "gemm_o_resid": n_tok * H * es + H * H * es + 2 * n_tok * H * 4,
                         "lm_head_lse": n_rows * H * es + V * H * es}.get(dom)
            out = {
                "metric": "candidate-pairs/sec (7B, 96+32 tok)", "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": model.engine.dtype, "data": "synthetic",
                "config": {"workload": "SYN v2t-VTG re-rank: Qwen2-7B dims (28 layers), seeded synthetic weights, 96 video + 32 text tokens per pair, "
                                       f"top-{K} text candidates per video query, {Q} queries ({n_pairs} pairs, {n_tok} packed tokens, {n_rows} label rows) per step per GPU",
                           "prefix_reuse": True, "vtg_compensated": vtg_headline or "none", "pairs_per_step_per_gpu": n_pairs, "tokens_per_step_per_gpu": n_tok,
                           "parallelism": f"query rows sharded over {world} GPU(s); RCCL all-gather of score rows at the end"},
                "algorithmic_gflop_per_pair": round(f_pair(128, 32) / 1e9, 1),
                "executed_gflop_per_pair": round(exec_flops_step / n_pairs / 1e9, 1),
                "executed_tflops_per_gpu": round(exec_flops_step * a.steps / dt / 1e12, 1),
                "executed_gflop_per_pair_e2m3": round(lo6_step / n_pairs / 1e9, 1),
                "frac_mfma_peak_whole_step": round(exec_flops_step * a.steps / dt / 1e12 / peak_step, 4),
                "roofline": {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                             "peak_note": None if dom_lo6 == 0.0 else "half of this kernel's flops are the e2m3 second pass of the compensated mode: peak = 2 / (1 / 2500 + 1 / 10000)",
                             "frac": round(ach / peak, 4), "traffic": tr[0] if tr else None,
                             "traffic_source": (f"{tr[1]}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (2 x FETCH_SIZE + WRITE_SIZE, "
                                                "fabric side of L2, Infinity-Cache hits included); looked up, not re-measured in this run") if tr else None,
                             "algorithmic_bytes_per_launch": alg_bytes,
                             "traffic_over_algorithmic": round(tr[0] / alg_bytes, 2) if (tr and alg_bytes) else None,
                             "avg_launch_ms": round(d["ms"] / d["calls"], 4), "flop_per_launch": d["flops"] / d["calls"]},
                "kernel_classes_ms": {k: round(v["ms"], 3) for k, v in rep.items() if v["calls"]},
            }
            if comp is not None:
                out["compensated_mode"] = comp
            if ss is not None:
                out["strong_scaling"] = ss
            if cpu and world == 1 and not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
            print(json.dumps(out), flush=True)

    # ---- watchdog: the legs below run AFTER the headline was measured, and the strong-scaling leg is the only part of this file whose collectives have never met more than
    # one GPU (no multi-GPU box was available to the builder).  Should it hang, every rank leaves after --leg-timeout seconds and rank 0 still prints the headline line
    # (with the timeout recorded inside it) instead of the launcher's time limit ending the run with nothing.
    import threading
    done = threading.Event()

    def on_timeout():
        if done.is_set():
            return
        msg = f"watchdog: the legs after the timed steps did not finish within {a.leg_timeout} s on rank {rank}"
        print("[bench] " + msg, file=sys.stderr, flush=True)
        if rank == 0:
            emit({"error": msg}, None, cpu=False)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)          # (every rank: a non-zero exit of any rank makes the launcher tear the job down, possibly before rank 0 has flushed its line)
    dog = threading.Timer(float(a.leg_timeout), on_timeout)
    dog.daemon = True
    if a.leg_timeout > 0:
        dog.start()

    # ---- strong-scaling leg (all ranks): one fixed-size N = 1000 evaluation, after and outside the timed steps above
    ss, ss_failed = None, False
    if not a.no_strong:
        try:
            ss = strong_scaling(model, world, rank, dev, n=a.strong_n, topk=a.topk, pg=pg, tvg_precise=a.tvg_precise)
        except Exception as e:                 # the headline line above is already measured: report the failure inside it instead of losing both
            import traceback
            ss = {"error": f"{type(e).__name__}: {e}", "traceback_tail": traceback.format_exc()[-1500:]}
            ss_failed = True
            print(f"[bench rank {rank}] strong-scaling leg failed: {ss['error']}", file=sys.stderr, flush=True)

    # ---- the same step with every VTG call FULLY COMPENSATED (what `--vtg_precise auto` picks on weights with a trained checkpoint's massive activations, DESIGN.md
    # section 4): reported beside the headline, never as `value`.  One GPU, fp16 engines, after everything above.
    comp = None
    if world == 1 and not a.no_compensated and model.engine.dtype == "f16" and model.vtg_precise is None:
        try:
            model.vtg_precise = "full"
            (sc_c, pl_c, _, _), = build_step_plans(model, rank, 1, Q, K)
            model.engine.reserve(pl_c.n_tokens, pl_c.n_rows, compensated=True)   # the weights' e2m3 images and the tile workspaces: a memory shortage is reported here
            sc_c.run(pl_c)                                        # warm-up: feature rows in the [hi | lo] layout, the e2m3 weight images
            torch.cuda.synchronize(); tc = time.perf_counter()
            for _ in range(3):
                out_c = sc_c.run(pl_c)
            torch.cuda.synchronize(); dtc = (time.perf_counter() - tc) / 3
            fl = RU.executed_flops(dims, pl_c.n_tokens, pl_c.n_rows, "vtg", "full")
            f6 = RU.lo6_pass_flops(dims, pl_c.n_tokens, pl_c.n_rows, "vtg", "full") if getattr(model.engine, "lo6", False) else 0.0
            comp = {"vtg_compensated": "full", "value": round(pl_c.n_pairs / dtc, 2), "unit": "pairs/s", "ms_per_step": round(dtc * 1e3, 3), "finite": bool(torch.isfinite(out_c).all()),
                    "second_pass": "e2m3 (engine option precise_lo6)" if f6 else "16-bit",
                    "frac_mfma_peak_whole_step": round(((fl - f6) / (PEAK_BF16_TFLOPS * 1e12) + f6 / (PEAK_FP6_TFLOPS * 1e12)) / dtc, 4),
                    "frac_mfma_peak_note": "the second pass's half of the executed flops priced at the fp6 peak (10 PFLOP/s), the rest at the 16-bit one: the pass is bound by what it "
                                           "moves (25 KiB per operand tile and 128-deep K-step through LDS-DMA and LDS), not by the e2m3 MFMAs -- DESIGN.md section 3",
                    "note": "same batch, every activation hi + lo; <= 4e-5 from the fp32 reference at 7B depth (tests/test_gpu_parity.py::test_e2m3_second_pass_of_the_compensated_gemms)"}
        except Exception as e:
            comp = {"error": f"{type(e).__name__}: {e}"}
        finally:
            model.vtg_precise = None

    done.set(); dog.cancel()
    emit(ss, comp)
    if ss_failed and pg:
        # a rank that failed inside the leg may have left the others inside one of its collectives: no barrier here (it could never complete).  The failed rank
        # exits non-zero without tearing the group down, so the launcher ends the job instead of hanging; rank 0's line above (if it is the one that failed) says why
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)
    if pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
