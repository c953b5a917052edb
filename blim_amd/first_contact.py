"""First contact with REAL artefacts: one command that answers "can this engine evaluate this checkpoint on this dataset, and in which mode?"

    python -m blim_amd.first_contact --model_path ./pretrained/VideoChat-Flash-Qwen2-7B_res448 --resume ./checkpoint/msrvtt.pth --dataset MSRVTT

The build environment has no checkpoint, tokenizer or dataset (no network), so everything that depends on them was developed against synthetic
stand-ins.  This runs, in order, every check DESIGN.md lists as "the first thing to do with a real one" and prints GO / NO-GO:

  1. config.json: inside the scoring path's splice (main.py:96 loads it through from_pretrained; modeling_videochat_flash.py:209-243, 340-353, 452-485)?
  2. tokenizer-dependent constants: <|im_end|> = 151645 (videochat_flash/conversation.py:13, read by retrieval_utils.py:99 as a LABEL id), "\\n" id, pad id,
     the TVG prefix length the dataset computes (base_dataset.py:20-24; the reference's comment at modeling_videochat_flash.py:408 spells out 21 ids),
     row shapes (one <image> per row, response spans) and row lengths against config.tokenizer_model_max_length;
  3. checkpoint key naming: every engine tensor found in the base checkpoint; the resume file checked as main.py:125-128 checks it (every key maps, every
     expected adapter + visual_head present, parameter total = trainable total) with the provenance table;
  4. weights into the engine (adapters apart), then the numeric-mode table: `--vtg_precise auto` measured on this checkpoint's own pairs;
  5. up to 64 pairs scored twice -- the fused PairScorer and the literal reference-shaped API (prepare_inputs_labels_for_multimodal -> forward ->
     criterion) -- VTG and TVG: they must agree within 1e-3 (they are the same arithmetic batched differently).
Exit code 0 = GO, 1 = NO-GO (the failed checks are listed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np


class Report:
    def __init__(self):
        self.rows = []

    def add(self, ok, what, detail=""):
        self.rows.append((bool(ok), what, detail))
        print(f"[{'ok' if ok else 'FAIL'}] {what}" + (f": {detail}" if detail else ""), flush=True)
        return bool(ok)

    @property
    def go(self):
        return all(r[0] for r in self.rows)


def check_config(rep: Report, model_path: str, num_clips: int):
    from .main import dims_from_config
    cfg = json.load(open(os.path.join(model_path, "config.json")))
    try:
        dims = dims_from_config(model_path, num_clips)
        rep.add(True, "config.json inside the scoring path's splice", f"H {dims.hidden_size}, {dims.num_layers} layers, {dims.num_heads}/{dims.num_kv_heads} heads, "
                f"I {dims.intermediate_size}, V {dims.vocab_size}, mm_hidden {dims.mm_hidden_size}")
    except NotImplementedError as e:
        rep.add(False, "config.json inside the scoring path's splice", str(e))
        return None, cfg
    rep.add(dims.hidden_size // dims.num_heads == 128, "head_dim 128 (the attention / RoPE kernels' size)", str(dims.hidden_size // dims.num_heads))
    # Which attention class the reference would build from this config (modeling_qwen2_flash.py:736: QWEN2_ATTENTION_CLASSES[config._attn_implementation]; main.py:96
    # passes no attn_implementation, so transformers takes the config's value, else flash_attention_2 when flash-attn is importable -- setup.sh:7 installs it -- else
    # sdpa).  Parity is pinned to the eager / SDPA semantics; under flash_attention_2 masked query rows yield a zero attention output, which moves the TVG-CPN prior.
    impl = cfg.get("_attn_implementation", cfg.get("attn_implementation"))
    impl_note = (f"config.json names {impl!r}" if impl else "config.json names none: the reference's run took flash_attention_2 if flash-attn was installed (setup.sh:7), sdpa otherwise")
    if impl == "flash_attention_2" or impl is None:
        impl_note += ("  WARN: under flash_attention_2 the reference drops masked positions before the attention kernel and pads zeros back (modeling_qwen2_flash.py:526-563); "
                      "this engine's default computes such rows (eager / SDPA semantics, what its goldens pin).  To compare with numbers from a flash-attn run: --masked_query_zero "
                      "(parity-unpinned; only the t2v candidate_prior changes)")
    rep.add(True, "attention implementation the checkpoint's config selects", impl_note)
    rep.add(cfg.get("tokenizer_padding_side", "right") == "right", "tokenizer_padding_side = right (modeling_videochat_flash.py:472-485)", str(cfg.get("tokenizer_padding_side", "right")))
    return dims, cfg


def check_tokenizer_and_rows(rep: Report, tokenizer, loader, cfg):
    from .retrieval_utils import IMAGE_TOKEN_ID
    from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX
    enc = lambda s: list(tokenizer(s).input_ids)
    rep.add(enc("<|im_end|>") == [IMAGE_TOKEN_ID], "<|im_end|> tokenises to the single id 151645 (conversation.py:13; the TVG glue finds it among the LABELS)", str(enc("<|im_end|>")))
    rep.add(len(enc("\n")) == 1, "newline is one token", str(enc("\n")))
    rep.add(getattr(tokenizer, "pad_token_id", None) is not None, "tokenizer has a pad_token_id (retrieval_utils.py:155-167 pads with it)", str(getattr(tokenizer, "pad_token_id", None)))
    ds = loader.dataset
    rep.add(ds.tvg_prefix_length > 0, "dataset.tvg_prefix_length (base_dataset.py:20-24; 21 with the released tokenizer per modeling_videochat_flash.py:408)", str(ds.tvg_prefix_length))
    limit = cfg.get("tokenizer_model_max_length")
    n_img_ok, span_ok, end_ok, lens_v, lens_t = True, True, True, [], []
    n_video_tokens = None
    for idx in range(min(len(ds), 256)):
        it = ds[idx]
        for kind in ("vtg", "tvg"):
            ids, lab = it[f"{kind}_ids"].numpy(), it[f"{kind}_labels"].numpy()
            n_img_ok &= int((ids == IMAGE_TOKEN_INDEX).sum()) == 1
            resp = lab != IGNORE_INDEX
            first = int(np.argmax(resp)) if resp.any() else len(ids)
            span_ok &= bool(resp[first:].all()) and resp.any()
            if kind == "tvg":
                end_ok &= int((lab == IMAGE_TOKEN_ID).sum()) == 1 and int(ids[first]) == IMAGE_TOKEN_INDEX
        v = it["video"]
        n_video_tokens = int(v.shape[0] * v.shape[1])
        lens_v.append(len(it["vtg_ids"]) - 1 + n_video_tokens)
        lens_t.append(len(it["tvg_ids"]) - 1 + int(v.shape[0]))
    rep.add(n_img_ok, "every row holds exactly one <image> placeholder")
    rep.add(span_ok, "labels are -100 on the prompt and one trailing response span")
    rep.add(end_ok, "TVG rows: the response starts with <image> and carries exactly one <|im_end|> label (retrieval_utils.py:99)")
    detail = f"VTG rows {min(lens_v)}..{max(lens_v)} tokens ({n_video_tokens} video tokens), TVG rows {min(lens_t)}..{max(lens_t)}; tokenizer_model_max_length = {limit}"
    rep.add(limit is None or max(lens_t) <= limit, "no TVG row is cut by tokenizer_model_max_length (a cut row has no <|im_end|> label left)", detail)
    if limit is not None and max(lens_v) > limit:
        print(f"[note] {sum(l > limit for l in lens_v)} of {len(lens_v)} sampled VTG rows exceed the limit and lose their tail, as in the reference (modeling_videochat_flash.py:452-457)")


def check_checkpoint_keys(rep: Report, dims, model_path: str, resume: str, lora_r: int, lora_alpha: float):
    from . import checkpoint as CK
    from .synth import weight_shapes
    shapes = weight_shapes(dims)
    keys, get = CK.open_base_checkpoint(model_path)
    have = {CK.hf_to_canonical(k) for k in keys} - {None}
    need = [n for n in shapes if not n.startswith("tvg_mlp.") and n != "visual_head"]
    missing = [n for n in need if n not in have]
    rep.add(not missing, f"base checkpoint holds every engine tensor ({len(need)} names)", f"missing e.g. {[CK.canonical_to_hf(n) for n in missing[:3]]}" if missing else f"{len(keys)} keys in the files")
    other = sorted({k.split(".")[0] + "." + k.split(".")[1] for k in keys if CK.hf_to_canonical(k) is None})
    if other:
        print(f"[note] key groups outside the scoring path (not loaded): {other[:8]}")
    if resume:
        try:
            adapters, full = CK.read_resume(resume, dims, lora_r, strict_resume=True)
            rep.add(True, "resume file: every key maps onto an engine tensor, all expected adapters + visual_head present, parameter total = trainable total (main.py:127)",
                    f"{len(adapters)} adapters, full tensors {sorted(full)}")
            ratios = []
            for n in list(adapters)[:: max(1, len(adapters) // 12)]:
                if n in have or n.startswith("tvg_mlp."):
                    src = n if n in have else "mlp." + n[len("tvg_mlp."):]
                    w = get([k for k in keys if CK.hf_to_canonical(k) == src][0])
                    d = CK.lora_delta(adapters[n]["A"], adapters[n]["B"], lora_r, lora_alpha)
                    ratios.append(float(np.linalg.norm(d) / (np.linalg.norm(w) + 1e-30)))
            if ratios:
                print(f"[note] size of the update, ||(alpha / r) B A|| / ||W||, over a sample of adapters: median {np.median(ratios):.2e}, max {np.max(ratios):.2e}")
        except (ValueError, KeyError) as e:
            rep.add(False, "resume file checked as main.py:125-128 checks it", str(e)[:600])


def numeric_checks(rep: Report, model, loader, tokenizer, args, n_pairs: int = 64):
    import torch
    from . import retrieval_utils as RU
    from .modeling import DDPLike
    ddp = DDPLike(model)
    video, vlab = [], []
    rows = {k: [] for k in ("vtg_ids", "vtg_labels", "vtg_masks", "tvg_ids", "tvg_labels", "tvg_masks")}
    for data in loader:
        video += list(data["video"]); vlab.append(torch.as_tensor(data["tvg_video_labels"]))
        for k in rows:
            rows[k] += data[k]
    vtg = RU.padding_ids(rows["vtg_ids"], rows["vtg_labels"], rows["vtg_masks"], tokenizer)
    tvg = RU.padding_ids(rows["tvg_ids"], rows["tvg_labels"], rows["tvg_masks"], tokenizer)
    vlab = torch.cat(vlab)
    vocab = loader.dataset.video_vocab
    model.set_tvg_prefix_length(loader.dataset.tvg_prefix_length)
    N = len(video)
    scores_file = f"./scores/{args.dataset.lower()}{'' if args.resume else '_zeroshot'}.pth"
    if os.path.exists(scores_file):
        sims = torch.as_tensor(torch.load(scores_file, weights_only=True)["v2t"])
    else:
        print(f"[note] {scores_file} not found: calibration pairs taken around the diagonal")
        sims = -(torch.arange(N)[:, None] - torch.arange(N)[None, :]).abs().float()
    k = min(N, args.topk)
    scorer = RU.PairScorer(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, vocab, vlab, args.num_clips, max_tokens=args.max_tokens)
    # ---- 4. the numeric-mode table
    model.vtg_precise = "auto"
    n_eval = 3 * N * k                                     # entries of each kind in a full evaluation of this set (the tail extrapolation's horizon)
    chosen, table = scorer.calibrate_vtg(RU.calibration_pairs(sims, k, n_queries=32, per_query=8), n_eval=n_eval)
    rate = {"none": 1.0, "full": 0.67}          # VTG throughput relative to plain fp16 (fp16 engines, e2m3 second pass)
    rep.add(True, "vtg_precise auto (PairScorer.calibrate_vtg) on this checkpoint",
            ", ".join(f"{m} max {v['max']:.1e} rms {v['rms']:.1e} predicted max {v['pred']:.1e}" for m, v in table.items()) + f" -> {chosen}"
            + (f" (VTG calls at about {rate[chosen]:.2f} of the plain-fp16 rate)" if model.engine.dtype == "f16" and chosen in rate else ""))
    if args.resume and scorer.split_tvg:
        tp = RU.calibration_pairs(sims.T, k, n_queries=64, per_query=4)
        tchosen, ttable = scorer.calibrate_tvg(np.stack([tp[:, 1], tp[:, 0]], axis=1), n_eval=n_eval)
        rep.add(True, "tvg_precise auto (PairScorer.calibrate_tvg: how much of the TVG calls' MLP branch runs compensated)",
                ", ".join(f"{m} max {v['max']:.1e} rms {v['rms']:.1e} predicted max {v['pred']:.1e}" for m, v in ttable.items()) + f" -> {tchosen}")
    # ---- 5. fused vs literal on the same pairs (the TVG pass in the mode just chosen, both paths)
    q = max(1, min(N, n_pairs // k))
    a = types.SimpleNamespace(topk=k, batch_size_eval=min(16, k), num_clips=args.num_clips)
    dev = model.device
    worst = {}
    finetuned = bool(args.resume)
    for name, qv, ft in (("v2t VTG", True, "vtg"),) + ((("t2v TVG", False, "tvg"),) if finetuned else ()):
        s_rows = (sims if qv else sims.T)[:q]
        pairs = RU._topk_pairs(s_rows, 0, k, qv)
        fused = scorer.vtg(pairs) if ft == "vtg" else scorer.tvg(pairs)
        S = torch.full((N, N), -100.0, device=dev)
        fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
        ids, lab, msk = vtg if ft == "vtg" else tvg
        S = fn(S, s_rows, 0, ids, msk, lab, video, vocab.to(dev), vlab, ddp, dev, a, forward_type=ft, cpn=False).cpu().numpy()
        r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
        lit = S[r, c]
        worst[name] = float(np.max(np.abs(fused - lit) / np.abs(lit)))
        rep.add(np.isfinite(fused).all() and np.isfinite(lit).all() and worst[name] < 1e-3, f"{name}: fused PairScorer == literal reference-shaped API on {len(pairs)} pairs",
                f"worst relative difference {worst[name]:.2e}; scores {float(np.min(fused)):.3f} .. {float(np.max(fused)):.3f}")
    return chosen, table, worst


def run(args, tokenizer=None) -> Report:
    import torch
    from . import checkpoint as CK
    from .dataloader import load_data
    from .main import load_tokenizer
    from .modeling import BlimModel
    rep = Report()
    dims, cfg = check_config(rep, args.model_path, args.num_clips)
    if dims is None:
        return rep
    tokenizer = tokenizer or load_tokenizer(args.model_path)
    loader = load_data(args, tokenizer=tokenizer, split="test")
    rep.add(len(loader.dataset) > 0, f"dataset {args.dataset}: test split readable", f"{len(loader.dataset)} items, {len(loader.dataset.vids)} videos, "
            f"{len(loader.dataset.features)} feature files")
    check_tokenizer_and_rows(rep, tokenizer, loader, cfg)
    check_checkpoint_keys(rep, dims, args.model_path, args.resume, args.lora_r, float(args.lora_alpha))
    if not rep.go:
        print("NO-GO before any weight was loaded (fix the above first)")
        return rep
    model = BlimModel(dims, dtype=args.dtype, tokenizer_model_max_length=cfg.get("tokenizer_model_max_length"))
    report = CK.load_checkpoint(model.engine, dims, args.model_path, args.resume or None, lora_r=args.lora_r, lora_alpha=args.lora_alpha, lora_mode=args.lora_mode)
    rep.add(model.engine.weights_ready(), "weights streamed into the engine", CK.summarize_report(report) + f"; {model.engine.num_adapters()} adapters kept apart")
    try:
        numeric_checks(rep, model, loader, tokenizer, args)
    finally:
        model.engine.close()
    return rep


def get_args_parser():
    p = argparse.ArgumentParser("first contact with a real checkpoint / resume file / dataset")
    p.add_argument("--model_path", required=True)
    p.add_argument("--resume", default="")
    p.add_argument("--dataset", default="MSRVTT", choices=["DiDeMo", "ActivityNet", "LSMDC", "MSRVTT"])
    p.add_argument("--topk", default=16, type=int)
    p.add_argument("--num_clips", default=4, type=int)
    p.add_argument("--batch_size_eval", default=64, type=int)
    p.add_argument("--num_workers", default=0, type=int)
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--lora_alpha", default=32, type=int)
    p.add_argument("--lora_mode", default="apart", choices=["apart", "merge"])
    p.add_argument("--dtype", default=None, choices=["f16", "bf16"])
    p.add_argument("--max_tokens", default=32768, type=int)
    return p


def main(argv=None, tokenizer=None) -> int:
    args = get_args_parser().parse_args(argv)
    rep = run(args, tokenizer=tokenizer)
    failed = [w for ok, w, _ in rep.rows if not ok]
    print(("GO" if rep.go else "NO-GO") + f": {sum(ok for ok, _, _ in rep.rows)} of {len(rep.rows)} checks passed" + ("" if rep.go else "; failed: " + "; ".join(failed)))
    return 0 if rep.go else 1


if __name__ == "__main__":
    sys.exit(main())
