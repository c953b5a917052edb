"""GPU bring-up: every kernel class against the numpy oracle, stage by stage, with numbers printed even on mismatch.
Run on the GPU box:  python tools/bringup.py [--skip-wide] [--perf]   (writes gpurun_out/bringup.log)"""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from blim_amd import engine as eng, synth  # noqa: E402
from blim_amd.modeling import BlimModel, DDPLike  # noqa: E402
from blim_amd import retrieval_utils as RU  # noqa: E402
from oracle import blim_oracle as O  # noqa: E402
from oracle.gen_golden import CASES  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "bringup.log"), "a")


def say(*a):
    msg = " ".join(str(x) for x in a)
    print(msg, flush=True)
    LOG.write(msg + "\n"); LOG.flush()


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max()), float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)), float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


DT = os.environ.get("BLIM_DTYPE", "f16")
TDT = torch.float16 if DT == "f16" else torch.bfloat16


def bf(x):
    if TDT == torch.bfloat16:
        return torch.from_numpy(synth.bf16_bits(np.asarray(x, np.float32)).view(np.int16)).view(torch.bfloat16)
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.float16)


def stage(name):
    def deco(fn):
        def run(*a, **k):
            say(f"\n===== {name}")
            t0 = time.time()
            try:
                r = fn(*a, **k)
                say(f"----- {name}: done in {time.time() - t0:.1f}s")
                return r
            except Exception:
                say(f"!!!!! {name}: EXCEPTION\n" + traceback.format_exc())
                return None
        return run
    return deco


@stage("A synth fill bit-exactness")
def test_fill():
    for n, std, mean in ((1000, 0.02, 0.0), (4099, 0.1, 1.0)):
        out = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        eng.fill_bell_bf16(out, 7, "layers.3.q_proj.w", std, mean)
        got = out.view(torch.int16).cpu().numpy().view(np.uint16)
        want = synth.bf16_bits(synth.bell_f32(7, "layers.3.q_proj.w", n, std, mean))
        say(f"fill n={n}: mismatches {int((got != want).sum())}")


@stage("B plain GEMM")
def test_gemm(perf):
    rs = np.random.RandomState(0)
    for (M, N, K) in ((256, 256, 64), (300, 500, 128), (1000, 260, 256), (513, 1028, 3584)):
        a = synth.bf16_round(rs.randn(M, K).astype(np.float32)); w = synth.bf16_round(rs.randn(N, K).astype(np.float32) * 0.05)
        got = eng.gemm_bf16(bf(a).cuda(), bf(w).cuda()).float().cpu().numpy()
        want = a @ w.T
        say(f"gemm {M}x{N}x{K}: maxabs/rel/rms", relerr(got, want))
    if perf:
        for (M, N, K) in ((8192, 3584, 3584), (8192, 37888, 3584), (8192, 3584, 18944), (8192, 4608, 3584), (16384, 37888, 3584), (4096, 4096, 4096), (8192, 8192, 8192)):
            a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda"); w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
            eng.fill_bell_bf16(a, 1, "a", 1.0); eng.fill_bell_bf16(w, 1, "w", 0.02)
            a = a.to(TDT); w = w.to(TDT)
            for _ in range(2):
                eng.gemm_bf16(a, w)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eng.gemm_bf16(a, w)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            say(f"gemm perf {M}x{N}x{K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")


def build(case, layers=None, device_synth=False):
    spec = CASES[case]
    d = dict(spec["dims"])
    if layers is not None:
        d["num_layers"] = layers
    dims = synth.ModelDims(**d)
    model = BlimModel(dims, max_positions=1024)
    if device_synth:
        model.engine.init_synthetic_weights(spec["wseed"])
        w = None
    else:
        w = synth.synthetic_weights(dims, spec["wseed"])
        model.engine.load_weights(w)
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    return spec, dims, model, w, prob


@stage("C tiny, ONE layer: per-stage intermediates")
def test_layer_parts():
    spec, dims, model, w, prob = build("tiny", layers=1)
    ocfg = O.OracleConfig(**{**spec["dims"], "num_layers": 1})
    om = O.OracleModel(ocfg, w); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    vtg = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    sel = [0, 1, 2]
    mask, cpn, emb, lab = om.prepare_inputs_labels_for_multimodal(vtg[0][sel], vtg[2][sel], vtg[1][sel], [prob.video[i] for i in sel])
    # projector
    f = model.project(torch.from_numpy(prob.video[0]).cuda(), False).float().cpu().numpy()
    say("projector mlp:", relerr(f, om.project_video(prob.video[0], False).reshape(-1, dims.hidden_size)))
    f = model.project(torch.from_numpy(prob.video[0]).cuda(), True).float().cpu().numpy()
    say("projector tvg_mlp+mean:", relerr(f, om.project_video(prob.video[0], True).mean(axis=1)))
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    r = model.prepare_inputs_labels_for_multimodal(T(vtg[0][sel]), None, T(vtg[2][sel]), None, T(vtg[1][sel]),
                                                   [T(prob.video[i]) for i in sel], ["video"] * 3, video_feature=True, cpn=True)
    (_, _, (m_t, c_t), _, e_t, l_t) = r
    say("prepare: mask eq", np.array_equal(m_t.cpu().numpy(), mask), "cpn eq", np.array_equal(c_t.cpu().numpy(), cpn),
        "labels eq", np.array_equal(l_t.cpu().numpy(), lab), "embeds", relerr(e_t.float().cpu().numpy(), emb))
    B, L, H = emb.shape
    for tag, mm in (("mask", mask), ("cpn", cpn)):
        parts = {}
        cos, sin = O.rope_tables(ocfg.head_dim, ocfg.rope_theta, L)
        x1 = om.decoder_layer(0, bf(emb).float().numpy(), O.additive_mask(mm, L), cos, sin, parts)
        lg, hd = model.engine.forward(e_t, T(mm.astype(np.uint8)), want_logits=False, want_hidden=True)
        nq, nk = dims.num_heads * 128, dims.num_kv_heads * 128
        qkv = model.engine.debug_read("qkv", (B * L, nq + 2 * nk), TDT).float().cpu().numpy().reshape(B, L, -1)
        say(f"[{tag}] q:", relerr(qkv[..., :nq], parts["q"]), " k:", relerr(qkv[..., nq:nq + nk], parts["k"]), " v:", relerr(qkv[..., nq + nk:], parts["v"]))
        at = model.engine.debug_read("attn", (B * L, H), TDT).float().cpu().numpy().reshape(B, L, H)
        valid = mask.astype(bool)
        say(f"[{tag}] attn (valid rows):", relerr(at[valid], parts["attn"][valid]))
        if tag == "cpn":
            rows_any = np.array([[(mm[b, :t + 1] != 0).any() for t in range(L)] for b in range(B)]) & valid
            say(f"[{tag}] attn (rows with >=1 visible key):", relerr(at[rows_any], parts["attn"][rows_any]))
        ac = model.engine.debug_read("act", (B * L, dims.intermediate_size), TDT).float().cpu().numpy().reshape(B, L, -1)
        say(f"[{tag}] act:", relerr(ac[valid], parts["act"][valid]))
        rs = model.engine.debug_read("resid", (B * L, H), torch.float32).cpu().numpy().reshape(B, L, H)
        say(f"[{tag}] resid out:", relerr(rs[valid], x1[valid]))
        hid = O.rms_norm(x1, om.w["final_norm"], ocfg.rms_eps)
        say(f"[{tag}] final hidden:", relerr(hd.cpu().numpy()[valid], hid[valid]))
        for tr in (0, 1):
            model.engine.set_option("attn_tr_read", tr)
            model.engine.forward(e_t, T(mm.astype(np.uint8)), want_logits=False, want_hidden=True)
            a2 = model.engine.debug_read("attn", (B * L, H), TDT).float().cpu().numpy().reshape(B, L, H)
            say(f"[{tag}] attn tr_read={tr}:", relerr(a2[valid], parts["attn"][valid]))
    model.engine.close()


def six_passes(model, prob, spec, dims, literal):
    ddp = DDPLike(model)
    dev = model.device
    tok = type("T", (), {"pad_token_id": synth.PAD_ID})()
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    video = [torch.from_numpy(v) for v in prob.video]
    vocab = torch.from_numpy(prob.video_vocab); vlab = torch.from_numpy(prob.tvg_video_labels)
    n = spec["n"]
    args = type("A", (), {"topk": spec["topk"], "batch_size_eval": spec["bs"], "num_clips": dims.num_clips})()
    out = {}
    passes = [("v2t_vtg", True, "vtg", False), ("v2t_vtg_cpn", True, "vtg", True), ("v2t_tvg", True, "tvg", False),
              ("t2v_vtg", False, "vtg", False), ("t2v_tvg", False, "tvg", False), ("t2v_tvg_cpn", False, "tvg", True)]
    if not literal:
        scorer = RU.PairScorer(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, vocab, vlab, dims.num_clips, max_tokens=4096)
    for name, qv, ft, cpn in passes:
        sims = torch.from_numpy(prob.v2t_sims if qv else prob.t2v_sims)
        S = torch.full((n, n), -100.0, device=dev)
        if literal:
            fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
            ids, lab, msk = vtg if ft == "vtg" else tvg
            S = fn(S, sims, 0, ids, msk, lab, video, vocab.to(dev), vlab, ddp, dev, args, forward_type=ft, cpn=cpn)
        else:
            pairs = RU._topk_pairs(sims, 0, args.topk, qv)
            sc = scorer.vtg(pairs, cpn) if ft == "vtg" else scorer.tvg(pairs, cpn)
            r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
            S[torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev)] = torch.from_numpy(sc).to(dev)
        out[name] = S.cpu().numpy()
    return out


def compare_passes(tag, got, g):
    worst = 0.0
    for name, S in got.items():
        G = g[f"S_{name}"]
        same_bg = np.array_equal(S == -100.0, G == -100.0)
        m = G != -100.0
        rel = np.abs(S[m] - G[m]) / np.abs(G[m])
        worst = max(worst, float(rel.max()))
        say(f"{tag} {name}: background eq {same_bg}; max rel err {rel.max():.3e}; sample got {S[m][:3]} want {G[m][:3]}")
    say(f"{tag} worst rel err {worst:.3e}  ({'PASS' if worst <= 1e-3 else 'FAIL'} at 1e-3)")


@stage("D tiny, full model: hidden + six passes (literal and fused) vs the reference's golden vectors")
def test_tiny_full():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tiny.npz"))
    spec, dims, model, w, prob = build("tiny")
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    for kind in ("vtg", "tvg"):
        emb = g[f"prep_{kind}_embeds"]
        for tag, mk in (("", f"prep_{kind}_mask"), ("_cpn", f"prep_{kind}_cpn_mask")):
            mm = g[mk]
            lg, hd = model.engine.forward(bf(emb).cuda(), T(mm.astype(np.uint8)), want_logits=(kind == "vtg" and tag == ""), want_hidden=True)
            valid = g[f"prep_{kind}_mask"].astype(bool)
            say(f"forward {kind}{tag} hidden (valid rows):", relerr(hd.cpu().numpy()[valid], g[f"fwd_{kind}{tag}_hidden"][valid]))
            if lg is not None:
                pos = g["fwd_vtg_logits_row0_pos"]
                say("logits row0 subsample:", relerr(lg[0, torch.from_numpy(pos).cuda()][:, ::997].cpu().numpy(), g["fwd_vtg_logits_row0_sub"]))
                sc = RU.vtg_criterion(lg, T(g["prep_vtg_labels"]))
                say("vtg_criterion on engine logits:", sc.cpu().numpy(), "want", g["fwd_vtg_score"])
    compare_passes("literal", six_passes(model, prob, spec, dims, True), g)
    compare_passes("fused", six_passes(model, prob, spec, dims, False), g)
    model.engine.close()


@stage("E wide (7B width, 1 layer, weights generated ON DEVICE): six passes vs golden")
def test_wide():
    g = np.load(os.path.join(ROOT, "tests", "golden", "wide.npz"))
    spec, dims, model, w, prob = build("wide", device_synth=True)
    compare_passes("fused", six_passes(model, prob, spec, dims, False), g)
    compare_passes("literal", six_passes(model, prob, spec, dims, True), g)
    model.engine.close()


@stage("F perf snapshot: 7B, 28 layers, SYN 96+32 shape")
def test_perf(n_query=24):
    dims = synth.ModelDims()
    t0 = time.time()
    model = BlimModel(dims, max_positions=1024)
    model.engine.init_synthetic_weights(0)
    torch.cuda.synchronize()
    say(f"7B synthetic weights on device in {time.time() - t0:.1f}s; mem {torch.cuda.mem_get_info()}")
    E = model.engine
    H = dims.hidden_size
    # packed batch: n_query prefixes of 96 tokens + 16 suffixes of 31 tokens each
    pos, vis, ss, sl, ps, pl, rows, rstart = [], [], [], [], [], [], [], [0]
    t = 0
    for q in range(n_query):
        p0 = t; pos += list(range(96)); ss.append(t); sl.append(96); ps.append(0); pl.append(0); t += 96
        for c in range(16):
            ss.append(t); sl.append(31); ps.append(p0); pl.append(96); pos += list(range(96, 127))
            rows += [p0 + 95] + list(range(t, t + 31)); rstart.append(len(rows)); t += 31
    batch = eng.PackedBatch(np.array(pos), np.ones(t, np.uint8), np.array(ss), np.array(sl), np.array(ps), np.array(pl))
    emb = torch.empty((t, H), dtype=torch.bfloat16, device="cuda"); eng.fill_bell_bf16(emb, 1, "emb", 0.02); emb = emb.to(TDT)
    rows_t = torch.tensor(rows, dtype=torch.int32, device="cuda"); rs_t = torch.tensor(rstart, dtype=torch.int32, device="cuda")
    labels = torch.from_numpy(synth.uniform_ids(1, "lab", len(rows), 1000, 150000).astype(np.int32)).cuda()
    say(f"tokens {t}, rows {len(rows)}, pairs {len(rstart) - 1}")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        sc = E.score_vtg(batch, emb, rows_t, labels, rs_t)
        torch.cuda.synchronize(); dt = time.time() - t0
        say(f"iter {it}: {dt * 1e3:.1f} ms -> {(len(rstart) - 1) / dt:.1f} pairs/s; scores {sc[:3].cpu().numpy()}")
    E.timing_enable(True)
    E.score_vtg(batch, emb, rows_t, labels, rs_t)
    rep = E.timing_report(); E.timing_enable(False)
    tot = sum(v["ms"] for v in rep.values())
    for k, v in rep.items():
        tf = v["flops"] / v["ms"] / 1e9 if v["ms"] > 0 and v["flops"] > 0 else 0
        say(f"  {k:22s} {v['ms']:9.3f} ms  {100 * v['ms'] / tot:5.1f}%  calls {v['calls']:4d}  {tf:8.1f} TFLOP/s")
    model.engine.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-wide", action="store_true")
    ap.add_argument("--perf", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    say(f"device: {torch.cuda.get_device_name(0)}; lib {eng.LIB_PATH}")
    want = set(a.only.split(",")) if a.only else None
    def on(k): return want is None or k in want
    if on("A"): test_fill()
    if on("B"): test_gemm(a.perf)
    if on("C"): test_layer_parts()
    if on("D"): test_tiny_full()
    if on("E") and not a.skip_wide: test_wide()
    if on("F") and a.perf: test_perf()
