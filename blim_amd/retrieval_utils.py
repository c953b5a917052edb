"""Scoring API of the reference (retrieval_utils.py) on the MI355X engine.

Two layers:

* The literal surface -- `compute_v2t_scores_x`, `compute_t2v_scores_x`, `padding_ids`,
  `vtg_criterion`, `tvg_criterion`, `evaluation` -- keeps the reference's names, arguments and
  per-query / per-batch control flow (retrieval_utils.py:18-281) so its eval loop is a drop-in.
  All tensor work goes through the HIP engine (no torch math on the hot path).

* `PairScorer` -- the fused fast path `evaluation` uses by default.  A likelihood is a function of the
  (video, text) pair only, so pairs from many queries are packed into large token batches; tokens that
  are identical for every candidate of a query (the video+prompt prefix for VTG, the caption prompt
  for TVG) are computed once and their K/V reused (SURVEY.md section 7 "prefix-KV reuse"); hidden
  states that no score reads (tail tokens) are not computed; priors that do not depend on the query
  (v2t VTG-CPN) are computed once per candidate.
"""
from __future__ import annotations

import sys
import time

import numpy as np

from . import distributed as dist_utils
from . import engine as eng
from .calibration import TVG_MODES, VTG_MODES, VTG_SPLIT_MODES, calibration_pairs, predicted_max_deviation      # noqa: F401  (re-exported: the names callers import from here)
from .pair_scorer import PairScorer, Plan, _PackState, _clip_major_vocab, _split_prompt_response, executed_flops, lo6_pass_flops      # noqa: F401
from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX

IMAGE_TOKEN_ID = 151645  # <|im_end|>, videochat_flash/conversation.py:13


# ----------------------------------------------------------------------------- criteria (literal)

class VTGCriterion:
    """retrieval_utils.py:18-33 on materialised logits [B, L, V] (device f32) and labels [B, L]."""

    def __call__(self, logits, labels):
        import torch
        B, L, V = logits.shape
        shift = torch.full((B, L), IGNORE_INDEX, dtype=torch.int32, device=logits.device)
        shift[:, :-1] = labels[:, 1:].to(torch.int32)                       # :24-25
        lp = eng.ce_rows(logits.reshape(B * L, V), shift.reshape(-1).contiguous())
        row_start = (torch.arange(B + 1, dtype=torch.int32, device=logits.device) * L).contiguous()
        return eng.segment_mean(lp, row_start, mode=0)                       # -loss.sum / loss.bool().sum, :32-33

    forward = __call__


class TVGCriterion:
    """retrieval_utils.py:35-43 on logits [B, clips, N] and labels [B, clips]."""

    def __call__(self, logits, labels):
        import torch
        B, Cn, N = logits.shape
        lp = eng.ce_rows(logits.reshape(B * Cn, N).contiguous(), labels.reshape(-1).to(torch.int32).contiguous())
        row_start = (torch.arange(B + 1, dtype=torch.int32, device=logits.device) * Cn).contiguous()
        return eng.segment_mean(lp, row_start, mode=1)

    forward = __call__


vtg_criterion = VTGCriterion()
tvg_criterion = TVGCriterion()


def _tvg_logits_literal(model, hidden_states, tvg_labels, video_vocab, num_clips, device):
    """retrieval_utils.py:99, 104-106: gather 4 hidden rows, visual_head, scaled dot with the video vocabulary."""
    import torch
    idx = (tvg_labels == IMAGE_TOKEN_ID).nonzero()[:, 1][:, None].repeat(1, num_clips) + (torch.arange(num_clips) - (num_clips + 1)).to(device)
    emb = torch.gather(hidden_states, 1, idx[..., None].repeat(1, 1, hidden_states.shape[-1]))
    emb = model.module.forward_visual(emb)                                    # [B, clips, M] f32
    eng_ = model.module.engine
    if getattr(eng_, "can_precise", False):
        # 16-bit engines: float32 visual-head outputs against the vocabulary registered as hi + lo operands (three-term compensated product: a 16-bit cast of
        # either side alone left 2 - 8e-4 on the bf16 engine's literal TVG scores)
        from .engine import vocab_key_of
        if getattr(eng_, "_vocab_key", None) != vocab_key_of(video_vocab):      # (address + shape + version + content fingerprint: a recycled address is not the same vocabulary)
            eng_.set_video_vocab(video_vocab)
        return eng_.tvg_logits_f32(emb.reshape(-1, emb.shape[-1]).float().contiguous(), emb.shape[0])
    vh = emb.to(model.module.dtype).reshape(-1, emb.shape[-1]).contiguous()
    return eng_.tvg_logits(vh, _clip_major_vocab(video_vocab, device, model.module.dtype), emb.shape[0])


def compute_v2t_scores_x(v2t_scores_x, iterator, start, input_ids, attention_masks, labels, video, video_vocab, tvg_video_labels,
                         model, device, args, forward_type=None, cpn=False):
    """retrieval_utils.py:48-111 (same arguments; mutates and returns v2t_scores_x)."""
    import torch
    for i, sims in enumerate(iterator):
        k = min(len(sims), args.topk)
        bs = args.batch_size_eval
        _, topk_idx = sims.topk(k=k, dim=0)
        topk_idx = topk_idx.cpu()
        enc = video[start + i].to(device, non_blocking=True)
        out = []
        for j in range(0, len(topk_idx), bs):
            sel = topk_idx[j:j + bs]
            n = len(sel)
            tvg = forward_type == "tvg"
            (_, _, (masks, cpn_masks), _, embeds, lab) = model.module.prepare_inputs_labels_for_multimodal(
                input_ids[sel].to(device), None, attention_masks[sel].to(device), None, labels[sel].to(device), [enc for _ in range(n)],
                ["video" for _ in range(n)], image_sizes=[(448, 448) for _ in range(n)], video_feature=True, tvg=tvg, cpn=True)
            outputs = model(inputs_embeds=embeds, attention_mask=cpn_masks if cpn else masks)
            if forward_type == "vtg":
                out.append(vtg_criterion(outputs.logits, lab))
            else:
                lg = _tvg_logits_literal(model, outputs.hidden_states, lab, video_vocab, args.num_clips, device)
                out.append(tvg_criterion(lg, tvg_video_labels[start + i].repeat(n, args.num_clips).to(device)))
        v2t_scores_x[start + i, topk_idx] = torch.cat(out, dim=0).to(v2t_scores_x.dtype)
    return v2t_scores_x


def compute_t2v_scores_x(t2v_scores_x, iterator, start, input_ids, attention_masks, labels, video, video_vocab, tvg_video_labels,
                         model, device, args, forward_type=None, cpn=False):
    """retrieval_utils.py:113-153 (text row repeated, candidate videos vary)."""
    import torch
    for i, sims in enumerate(iterator):
        k = min(len(sims), args.topk)
        bs = args.batch_size_eval
        _, topk_idx = sims.topk(k=k, dim=0)
        topk_idx = topk_idx.cpu()
        out = []
        for j in range(0, len(topk_idx), bs):
            sel = topk_idx[j:j + bs]
            enc = [video[int(v)].to(device) for v in sel]
            n = len(enc)
            tvg = forward_type == "tvg"
            (_, _, (masks, cpn_masks), _, embeds, lab) = model.module.prepare_inputs_labels_for_multimodal(
                input_ids[start + i].repeat(n, 1).to(device), None, attention_masks[start + i].repeat(n, 1).to(device), None,
                labels[start + i].repeat(n, 1).to(device), enc, ["video" for _ in range(n)], image_sizes=[(448, 448) for _ in range(n)],
                video_feature=True, tvg=tvg, cpn=True)
            outputs = model(inputs_embeds=embeds, attention_mask=cpn_masks if cpn else masks)
            if forward_type == "vtg":
                out.append(vtg_criterion(outputs.logits, lab))
            else:
                lg = _tvg_logits_literal(model, outputs.hidden_states, lab, video_vocab, args.num_clips, device)
                out.append(tvg_criterion(lg, tvg_video_labels[sel][:, None].repeat(1, args.num_clips).to(device)))
        t2v_scores_x[start + i, topk_idx] = torch.cat(out, dim=0).to(t2v_scores_x.dtype)
    return t2v_scores_x


def padding_ids(input_ids, labels, masks, tokenizer=None):
    """retrieval_utils.py:155-167: LEFT-pad ids (pad id), labels (-100), masks (0) to the longest row."""
    import torch
    n = len(input_ids)
    L = max(len(x) for x in input_ids)
    ids_p = torch.full((n, L), tokenizer.pad_token_id, dtype=torch.long)
    lab_p = torch.full((n, L), IGNORE_INDEX, dtype=torch.long)
    msk_p = torch.zeros((n, L), dtype=torch.long)
    for i in range(n):
        c = len(input_ids[i])
        ids_p[i, L - c:] = torch.as_tensor(input_ids[i])
        lab_p[i, L - c:] = torch.as_tensor(labels[i])
        msk_p[i, L - c:] = torch.as_tensor(masks[i])
    return ids_p, lab_p, msk_p


# ----------------------------------------------------------------------------- evaluation

def _topk_pairs(sims_rows, start: int, topk: int, query_is_video: bool) -> np.ndarray:
    """(video, text) pairs of the top-k candidates of each local query row (sims.topk, retrieval_utils.py:52, 117)."""
    import torch
    sims = torch.as_tensor(sims_rows)
    k = min(sims.shape[1], topk)
    idx = sims.topk(k=k, dim=1).indices.cpu().numpy()
    q = np.repeat(np.arange(start, start + sims.shape[0]), k)
    c = idx.reshape(-1)
    return np.stack([q, c], axis=1) if query_is_video else np.stack([c, q], axis=1)


def evaluation(model, data_loader, device, tokenizer, args):
    """retrieval_utils.py:169-281.  Returns (t2v_dict, v2t_dict) of numpy [N, N] matrices (W=1-equivalent for any
    world size: row blocks are merged with an all-gather, not the reference's all_reduce(SUM) of -100-filled
    matrices -- SURVEY.md section 5 'the -100 offset quirk'; args.compat_allreduce_offset reproduces the offset)."""
    import torch
    model.eval()
    t_start = time.time()
    marks = []                                                          # (stage, host seconds since the start): device work is asynchronous,
    mark = lambda name: marks.append((name, round(time.time() - t_start, 4)))   # so a stage's host time is planning + launching, not its device time
    video, tvg_video_labels = [], []
    vtg_ids, vtg_labels, vtg_masks, tvg_ids, tvg_labels, tvg_masks = [], [], [], [], [], []
    for data in data_loader:                                                # :182-193
        video += [v for v in data["video"]]
        vtg_ids += data["vtg_ids"]; vtg_labels += data["vtg_labels"]; vtg_masks += data["vtg_masks"]
        tvg_ids += data["tvg_ids"]; tvg_labels += data["tvg_labels"]; tvg_masks += data["tvg_masks"]
        tvg_video_labels.append(torch.as_tensor(data["tvg_video_labels"]))
    vtg_ids, vtg_labels, vtg_masks = padding_ids(vtg_ids, vtg_labels, vtg_masks, tokenizer)      # :195-196
    tvg_ids, tvg_labels, tvg_masks = padding_ids(tvg_ids, tvg_labels, tvg_masks, tokenizer)
    tvg_video_labels = torch.cat(tvg_video_labels, dim=0)

    finetuned = (getattr(args, "resume", "") != "") or not getattr(args, "eval", True)          # :199, 227, 242
    scores = getattr(args, "iv2_scores", None)
    if scores is None:
        path = f"./scores/{args.dataset.lower()}{'' if finetuned else '_zeroshot'}.pth"          # :199-203
        scores = torch.load(path, weights_only=True)
    v2t_iv2, t2v_iv2 = torch.as_tensor(scores["v2t"]), torch.as_tensor(scores["t2v"])
    num_texts, num_videos = t2v_iv2.shape
    W, rank = dist_utils.get_world_size(), dist_utils.get_rank()
    video_vocab = data_loader.dataset.video_vocab
    model.module.set_tvg_prefix_length(data_loader.dataset.tvg_prefix_length)                     # :210

    literal = bool(getattr(args, "literal", False))
    # shard emulation (one process plays rank r of W without a process group: the rank's own share of the work, no merge;
    # used to time configurations that are quoted on 8 GPUs on a 1-GPU box)
    emulate = getattr(args, "shard", None)
    if emulate is not None:
        W, rank = int(emulate[0]), int(emulate[1])
    collective = (W > 1 or dist_utils.force_collective()) and emulate is None
    # pair pooling / ownership (fused path; below): log P(text i | video j) is v2t.candidate_likelihood[j, i] AND t2v.query_likelihood[i, j];
    # log P(video j | text i) is v2t.query_likelihood[j, i] AND t2v.candidate_likelihood[i, j].  args.dedup = False
    # (--no_dedup), compat_allreduce_offset and the literal path keep the reference's six row-sharded passes.
    dedup = (not literal) and bool(getattr(args, "dedup", True)) and not bool(getattr(args, "compat_allreduce_offset", False))
    full = lambda n, m: torch.full((n, m), -100.0, dtype=torch.float32, device=device)
    scorer = getattr(args, "_scorer", None)              # test hook: any object with .vtg(pairs, cpn) / .tvg(pairs, cpn)
    if scorer is None and not literal:
        scorer = PairScorer(model, vtg_ids, vtg_masks, vtg_labels, tvg_ids, tvg_masks, tvg_labels, video, video_vocab,
                            tvg_video_labels, args.num_clips, max_tokens=getattr(args, "max_tokens", 24576))
    elif isinstance(scorer, PairScorer):                 # a caller's scorer: follow what the model asks for / has resolved NOW
        m0 = model.module
        if hasattr(m0, "vtg_mode") and m0.vtg_mode() != "auto":
            scorer.set_vtg_mode(m0.vtg_mode())
        if hasattr(m0, "tvg_mode"):
            scorer.set_tvg_mode(m0.tvg_mode())
    stats = {"pairs_requested": 0, "pairs_scored": 0}

    def agree(chosen, modes, setter):
        """One mode for the whole job: every rank measures the same pairs with deterministic kernels, so the choices agree -- this makes it a guarantee (ranks on
        different devices, a future non-deterministic kernel): the most compensated choice of any rank, by one all-reduce(MAX) of the mode's index."""
        if collective and dist_utils.is_dist_avail_and_initialized():
            t = torch.tensor([modes.index(chosen)], dtype=torch.int32, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            if modes[int(t.item())] != chosen:
                chosen = modes[int(t.item())]
                setter(chosen)
        return chosen

    def new_scorer():
        return PairScorer(model, vtg_ids, vtg_masks, vtg_labels, tvg_ids, tvg_masks, tvg_labels, video, video_vocab, tvg_video_labels, args.num_clips,
                          max_tokens=getattr(args, "max_tokens", 24576))

    def adopt_of(kind):
        """Shard emulation only: the decision of the W-process job this process plays a rank of -- args.agreed_modes = {"vtg": (mode, confirmed), "tvg": (...)}, e.g. from
        the one-process run's _eval_stats (bench.py) -- so that the emulated rank does not decide on its 1 / W of the calibration sample."""
        am = getattr(args, "agreed_modes", None)
        return tuple(am[kind]) if (emulate is not None and am and am.get(kind)) else None

    def share_of(kind):
        """(world, rank) when the calibration sample is split: a real multi-rank job (every rank scores its block, the deviations are all-gathered) and an emulated rank that
        was handed the job's decision (its block is the COST a rank's share stands for).  An emulated rank WITHOUT one (`--shard W r` on its own) has nobody to gather from
        and must not decide on 1 / W of the sample -- 32 pairs extrapolated to 472,000 entries sent BASELINE config 4's rank 0 to the compensated mode in round 5 where the
        job's 256-pair sample says plain (profiles/r06_calibrator_false_rejects.md): it measures the whole sample, as a one-process job does."""
        if collective and dist_utils.is_dist_avail_and_initialized():
            return (W, rank)
        return (W, rank) if (emulate is not None and adopt_of(kind) is not None) else None

    def fmt_table(table):
        return ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e} (predicted max {v['pred']:.1e})"
                         + (f"; confirmation sample of {v['confirm']['n']}: {v['confirm']['max']:.1e} / {v['confirm']['rms']:.1e} (predicted max {v['confirm']['pred']:.1e})" if "confirm" in v else "")
                         for k, v in table.items())

    mod = model.module
    t_cal = time.time()
    if getattr(mod, "vtg_precise", None) == "auto":
        # `--vtg_precise auto` (the driver's default): which compensation the VTG calls need is MEASURED on this checkpoint (PairScorer.calibrate_vtg) -- once per set
        # of weights: what an earlier evaluation() resolved stands while the engine's weights and adapters are unchanged (BlimModel.vtg_mode) and is measured again
        # after every change (the training loop's validation loads new adapters every epoch: main.py:166)
        if hasattr(mod, "vtg_mode") and mod.vtg_mode() != "auto":
            stats["vtg_precise"] = mod.vtg_mode() or "none"
            if isinstance(scorer, PairScorer):
                scorer.set_vtg_mode(mod.vtg_mode())
        else:
            cal = scorer if isinstance(scorer, PairScorer) else new_scorer()
            kt_, kv_ = min(args.topk, num_texts), min(args.topk, num_videos)
            n_eval_vtg = num_videos * kt_ * (2 if args.cpn else 1) + num_texts * kv_                 # VTG-type entries of the whole evaluation (every rank's)
            cal_share = share_of("vtg")
            # 256 pairs, 32 distinct prefixes; and -- only when that sample is inside the bar but its extrapolation to n_eval entries is not -- a confirmation sample of
            # up to 2,048 pairs over 256 prefixes (calibration.CalibrationMixin._decide)
            chosen, table = cal.calibrate_vtg(calibration_pairs(v2t_iv2, args.topk, n_queries=32, per_query=8), n_eval=n_eval_vtg, share=cal_share,
                                              confirm_pairs=calibration_pairs(v2t_iv2, args.topk, n_queries=256, per_query=8), adopt=adopt_of("vtg"))
            chosen = agree(chosen, VTG_MODES, lambda m_: (cal.set_vtg_mode(m_), getattr(mod, "resolve_vtg", lambda x: None)(m_)))
            stats["vtg_precise"] = chosen; stats["vtg_precise_table"] = table
            if rank == 0:
                print("vtg_precise auto: deviation from the fully compensated mode on the calibration pairs (max / rms): "
                      + fmt_table(table) + f" -> {chosen}", file=sys.stderr, flush=True)
    if getattr(mod, "second_pass", None) == "auto" and isinstance(scorer, PairScorer) and (mod.vtg_mode() if hasattr(mod, "vtg_mode") else None) == "full":
        # `--second_pass auto` (bf16 engines, round 6): with the VTG calls compensated, may their second walk over K run on the e2m3 MFMA?  Measured like the modes above
        if mod.second_pass_resolved():
            stats["second_pass"] = "e2m3" if bool(getattr(mod.engine, "lo6", False)) else "16bit"
        else:
            cal = scorer if isinstance(scorer, PairScorer) else new_scorer()
            kt_, kv_ = min(args.topk, num_texts), min(args.topk, num_videos)
            chosen, table = cal.calibrate_second_pass(calibration_pairs(v2t_iv2, args.topk, n_queries=32, per_query=8),
                                                      n_eval=num_videos * kt_ * (2 if args.cpn else 1) + num_texts * kv_, share=share_of("second"),
                                                      confirm_pairs=calibration_pairs(v2t_iv2, args.topk, n_queries=256, per_query=8), adopt=adopt_of("second"))
            chosen = agree(chosen, ("e2m3", "16bit"), lambda m_: mod.resolve_second_pass(m_))
            stats["second_pass"] = chosen; stats["second_pass_table"] = table
            if rank == 0:
                print("second_pass auto: deviation of the e2m3 second pass from the bf16 one on the calibration pairs (max / rms): " + fmt_table(table) + f" -> {chosen}", file=sys.stderr, flush=True)
    if getattr(mod, "tvg_precise", None) == "auto" and finetuned:
        # likewise for the TVG calls' MLP branch (PairScorer.calibrate_tvg); zero-shot evaluations run no TVG pass
        if hasattr(mod, "tvg_resolved") and mod.tvg_resolved():
            stats["tvg_precise"] = mod.tvg_mode()
            if isinstance(scorer, PairScorer):
                scorer.set_tvg_mode(mod.tvg_mode())
        else:
            cal = scorer if isinstance(scorer, PairScorer) else new_scorer()
            # a few TEXT queries and their top videos, as the t2v TVG passes score them: the text prefix is shared by a query's videos -- but MANY queries with few
            # videos each: a TVG score's deviation depends mostly on its text prefix, so 8 queries x 16 videos were 8 effective samples (heavy7b weights, N = 1,000:
            # sample rms 2.7e-5 against 5.0e-5 over the whole evaluation, and `attn` was let through with 5 of 48,000 entries above the bar)
            tp = calibration_pairs(t2v_iv2, args.topk, n_queries=64, per_query=4)      # 256 pairs x (likelihood, prior) = 512 entries, 64 distinct text prefixes
            kt_, kv_ = min(args.topk, num_texts), min(args.topk, num_videos)
            cal_share = share_of("tvg")
            tc = calibration_pairs(t2v_iv2, args.topk, n_queries=256, per_query=4)     # the confirmation sample, should the first one's extrapolation alone miss the bar
            chosen, table = cal.calibrate_tvg(np.stack([tp[:, 1], tp[:, 0]], axis=1), n_eval=num_videos * kt_ + num_texts * kv_ * (2 if args.cpn else 1), share=cal_share,
                                              confirm_pairs=np.stack([tc[:, 1], tc[:, 0]], axis=1), adopt=adopt_of("tvg"))
            chosen = agree(chosen, TVG_MODES, lambda m_: (cal.set_tvg_mode(m_), getattr(mod, "resolve_tvg", lambda x: None)(m_)))
            stats["tvg_precise"] = chosen; stats["tvg_precise_table"] = table
            if rank == 0:
                print("tvg_precise auto: deviation from the fully compensated mode on the calibration pairs, likelihood + prior (max / rms): "
                      + fmt_table(table) + f" -> {chosen}", file=sys.stderr, flush=True)
    if "vtg_precise_table" in stats or "tvg_precise_table" in stats or "second_pass_table" in stats:
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        stats["calibration_seconds"] = round(time.time() - t_cal, 4)
        # an emulated rank with no decision handed in measured the job's WHOLE sample (share_of): W times a real rank's share of this time
        stats["calibration_whole_sample"] = bool(emulate is not None and getattr(args, "agreed_modes", None) is None and W > 1)
    mark("setup")

    def run_pass(S, sims_rows, start, query_is_video, ftype, cpn):
        """One of the reference's six passes over this rank's query rows (literal: its own loops; fused: the PairScorer)."""
        if literal:
            fn = compute_v2t_scores_x if query_is_video else compute_t2v_scores_x
            ids, msk, lab = (vtg_ids, vtg_masks, vtg_labels) if ftype == "vtg" else (tvg_ids, tvg_masks, tvg_labels)
            return fn(S, sims_rows, start, ids, msk, lab, video, video_vocab.to(device), tvg_video_labels, model, device, args,
                      forward_type=ftype, cpn=cpn)
        if sims_rows.shape[0] == 0:
            return S
        pairs = _topk_pairs(sims_rows, start, args.topk, query_is_video)
        stats["pairs_requested"] += len(pairs)
        stats["pairs_scored"] += len(pairs)
        r, c = (pairs[:, 0], pairs[:, 1]) if query_is_video else (pairs[:, 1], pairs[:, 0])
        if hasattr(scorer, "vtg_device"):                    # no host round trip: the pass's scores go device -> device
            sc = scorer.vtg_device(pairs, cpn) if ftype == "vtg" else scorer.tvg_device(pairs, cpn)
            S[torch.from_numpy(r).to(device), torch.from_numpy(c).to(device)] = sc
        else:
            sc = scorer.vtg(pairs, cpn) if ftype == "vtg" else scorer.tvg(pairs, cpn)
            S[torch.from_numpy(r).to(device), torch.from_numpy(c).to(device)] = torch.from_numpy(sc).to(device)
        return S

    def merge(dicts_blocks):
        compat = bool(getattr(args, "compat_allreduce_offset", False))
        keys = [(d, k, blk) for d, blk in dicts_blocks for k in list(d)]
        merged = dist_utils.merge_row_blocks_many([d[k] for d, k, _ in keys], [blk for _, _, blk in keys], W, compat_offset=compat)
        for (d, k, _), m in zip(keys, merged):                                                   # one RCCL all-gather for all matrices
            d[k] = m

    v2t, t2v = {}, {}
    if dedup:
        # ---- pair ownership (fused path).  Every likelihood is a function of the (video, text) pair, and what is expensive is shared per
        # VIDEO for VTG (the header + 256 video tokens + instruction prefix) and per TEXT for TVG (the caption prompt).  So the pairs of both
        # directions are pooled -- P = {(j, i): i in top-k of video j} U {(j, i): j in top-k of text i} -- each scored ONCE, and VTG pairs
        # are owned by the rank that owns video j, TVG pairs by the rank that owns text i (the reference's row blocks, :213-215 / :233-235):
        # a prefix is computed once per video / text over both directions.  (Sharding the t2v pass by text rows, as the reference's loop
        # order suggests, leaves ~2 candidates per video prefix at W = 8: 3x the tokens of the v2t pass.)  The owner fills a row block of
        # the (video, text) matrix for VTG and a column block for TVG; two all-gathers (+ the priors) assemble all matrices on every rank.
        Nv, Nt = num_videos, num_texts
        kt, kv = min(Nt, args.topk), min(Nv, args.topk)
        m_v2t = np.zeros((Nv, Nt), dtype=bool)
        m_v2t[np.repeat(np.arange(Nv), kt), v2t_iv2.topk(k=kt, dim=1).indices.cpu().numpy().reshape(-1)] = True
        m_t2v = np.zeros((Nv, Nt), dtype=bool)                                                 # (video, text) layout of the t2v requests
        m_t2v[t2v_iv2.topk(k=kv, dim=1).indices.cpu().numpy().reshape(-1), np.repeat(np.arange(Nt), kv)] = True
        need = m_v2t | m_t2v
        vs, ve = dist_utils.row_block(Nv, W, rank)
        ts, te = dist_utils.row_block(Nt, W, rank)
        n_v2t = 1 + (1 if args.cpn else 0) + (1 if finetuned else 0)
        n_t2v = 1 + ((1 + (1 if args.cpn else 0)) if finetuned else 0)
        stats["pairs_requested"] = int(m_v2t[vs:ve].sum()) * n_v2t + int(m_t2v[:, ts:te].sum()) * n_t2v
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)

        def score_owned(ftype, own):
            jj, ii = np.nonzero(own)
            M = full(Nv, Nt)
            if len(jj):
                pairs = np.stack([jj, ii], axis=1)
                stats["pairs_scored"] += len(pairs)
                fn = getattr(scorer, f"{ftype}_device", None)
                sc = fn(pairs, False) if fn is not None else torch.from_numpy(np.asarray(getattr(scorer, ftype)(pairs, False), dtype=np.float32)).to(device)
                M[to_dev(jj), to_dev(ii)] = sc
            return M

        own_v = need.copy(); own_v[:vs] = False; own_v[ve:] = False                            # VTG: rows of my videos
        prior_mine = None
        jv_, iv_ = np.nonzero(own_v)
        if args.cpn and te > ts and len(jv_) and hasattr(scorer, "vtg_jobs_device"):
            # ... and the v2t prior of my block of texts (below) in the SAME engine calls: at 8 ranks it is one small, latency-bound call of its own otherwise
            pl_ = np.stack([jv_, iv_], axis=1)
            tp_ = np.stack([np.zeros(te - ts, dtype=np.int64), np.arange(ts, te, dtype=np.int64)], axis=1)
            sc = scorer.vtg_jobs_device([(pl_, False), (tp_, True)])
            stats["pairs_scored"] += len(pl_) + (te - ts)
            M_vtg = full(Nv, Nt); M_vtg[to_dev(jv_), to_dev(iv_)] = sc[: len(pl_)]
            prior_mine = sc[len(pl_):]
        else:
            M_vtg = score_owned("vtg", own_v)
        mark("vtg")
        M_tvg_T = S_t2v_prior = None
        if finetuned:
            if collective and hasattr(scorer, "share_tvg_feats"):
                scorer.share_tvg_feats(W, rank)                                                # each rank projects its block of videos; one all-gather of the clip features
            elif emulate is not None and getattr(args, "peer_tvg_feats", None) and hasattr(scorer, "adopt_tvg_feats"):
                scorer.adopt_tvg_feats(W, rank, args.peer_tvg_feats)                           # shard emulation: own block projected, the peers' blocks as the all-gather delivers them
            own_t = need.copy(); own_t[:, :ts] = False; own_t[:, te:] = False                  # TVG: columns of my texts
            jj_, ii_ = np.nonzero(own_t)
            if args.cpn and te > ts and len(jj_) and hasattr(scorer, "tvg_jobs_device"):
                # the likelihoods of my texts' pairs and the t2v prior of my texts (keyed on (prompt, video)) are planned into the SAME engine calls: a rank's
                # share of either is a fraction of one call at 8 ranks
                pl_, pp_ = np.stack([jj_, ii_], axis=1), _topk_pairs(t2v_iv2[ts:te], ts, args.topk, False)
                sc = scorer.tvg_jobs_device([(pl_, False), (pp_, True)])
                stats["pairs_scored"] += len(pl_) + len(pp_)
                M_ = full(Nv, Nt); M_[to_dev(jj_), to_dev(ii_)] = sc[: len(pl_)]
                M_tvg_T = M_.T.contiguous()
                S_t2v_prior = full(Nt, Nv); S_t2v_prior[to_dev(pp_[:, 1]), to_dev(pp_[:, 0])] = sc[len(pl_):]
            else:
                M_tvg_T = score_owned("tvg", own_t).T.contiguous()                            # text-major: a row block
            mark("tvg")
        prior_t = None
        if args.cpn:
            # the v2t prior log P(text | masked video) does not depend on the query video: every rank scores its block of TEXTS once
            mine = torch.full((Nt // W + 1,), -100.0, dtype=torch.float32, device=device)
            if prior_mine is not None:
                mine[: te - ts] = prior_mine
            elif te > ts:
                tp = np.stack([np.zeros(te - ts, dtype=np.int64), np.arange(ts, te, dtype=np.int64)], axis=1)
                mine[: te - ts] = scorer.vtg_device(tp, True) if hasattr(scorer, "vtg_device") else \
                    torch.from_numpy(np.asarray(scorer.vtg(tp, True), dtype=np.float32)).to(device)
                stats["pairs_scored"] += te - ts
            if collective:
                parts = [torch.empty_like(mine) for _ in range(W)]
                torch.distributed.all_gather(parts, mine)
                prior_t = torch.cat([parts[r_][: dist_utils.row_block(Nt, W, r_)[1] - dist_utils.row_block(Nt, W, r_)[0]] for r_ in range(W)])
            else:
                prior_t = torch.full((Nt,), -100.0, dtype=torch.float32, device=device)
                prior_t[ts:te] = mine[: te - ts]
        mark("v2t_prior")
        if finetuned and args.cpn and S_t2v_prior is None:                                     # t2v TVG prior: keyed on (prompt, video); rows of my texts
            S_t2v_prior = full(Nt, Nv)
            if te > ts:
                pairs = _topk_pairs(t2v_iv2[ts:te], ts, args.topk, False)
                stats["pairs_scored"] += len(pairs)
                sc = scorer.tvg_device(pairs, True) if hasattr(scorer, "tvg_device") else torch.from_numpy(np.asarray(scorer.tvg(pairs, True), dtype=np.float32)).to(device)
                S_t2v_prior[to_dev(pairs[:, 1]), to_dev(pairs[:, 0])] = sc
        mark("t2v_prior")
        if collective:                                                                         # one all-gather for the row blocks of all three
            mats, blocks = [M_vtg], [(vs, ve)]
            if M_tvg_T is not None:
                mats.append(M_tvg_T); blocks.append((ts, te))
            if S_t2v_prior is not None:
                mats.append(S_t2v_prior); blocks.append((ts, te))
            merged = dist_utils.merge_row_blocks_many(mats, blocks, W)
            M_vtg = merged[0]
            if M_tvg_T is not None:
                M_tvg_T = merged[1]
            if S_t2v_prior is not None:
                S_t2v_prior = merged[-1]
        mv, mt = to_dev(m_v2t), to_dev(m_t2v)
        neg = torch.tensor(-100.0, dtype=torch.float32, device=device)
        v2t["candidate_likelihood"] = torch.where(mv, M_vtg, neg)
        if args.cpn:
            v2t["candidate_prior"] = torch.where(mv, prior_t[None, :].expand(Nv, Nt), neg)
        if finetuned:
            v2t["query_likelihood"] = torch.where(mv, M_tvg_T.T, neg)
        t2v["query_likelihood"] = torch.where(mt, M_vtg, neg).T.contiguous()
        if finetuned:
            t2v["candidate_likelihood"] = torch.where(mt.T, M_tvg_T, neg).contiguous()
            if args.cpn:
                t2v["candidate_prior"] = S_t2v_prior
    else:
        start, end = dist_utils.row_block(num_videos, W, rank)                                       # :213-215
        v2t["candidate_likelihood"] = run_pass(full(num_videos, num_texts), v2t_iv2[start:end], start, True, "vtg", False)
        if args.cpn:
            if W > 1 and not literal:
                # the v2t prior log P(text | masked video) does not depend on the query video: every rank scores its block of
                # TEXTS once (N/W forwards instead of rows*k/W), the [N] vector is all-gathered and scattered into the rank's top-k
                # entries
                t0, t1 = dist_utils.row_block(num_texts, W, rank)
                mine = torch.full((num_texts // W + 1,), -100.0, dtype=torch.float32, device=device)
                if t1 > t0:
                    tp = np.stack([np.zeros(t1 - t0, dtype=np.int64), np.arange(t0, t1, dtype=np.int64)], axis=1)
                    mine[: t1 - t0] = scorer.vtg_device(tp, True) if hasattr(scorer, "vtg_device") else \
                        torch.from_numpy(np.asarray(scorer.vtg(tp, True), dtype=np.float32)).to(device)
                    stats["pairs_scored"] += t1 - t0
                if collective:
                    parts = [torch.empty_like(mine) for _ in range(W)]
                    torch.distributed.all_gather(parts, mine)
                    prior = torch.cat(parts)[:num_texts]
                else:                                        # shard emulation: only this rank's texts are known
                    prior = torch.full((num_texts,), -100.0, dtype=torch.float32, device=device)
                    prior[t0:t1] = mine[: t1 - t0]
                S = full(num_videos, num_texts)
                if end > start:
                    pairs = _topk_pairs(v2t_iv2[start:end], start, args.topk, True)
                    stats["pairs_requested"] += len(pairs)
                    r_, c_ = torch.from_numpy(pairs[:, 0]).to(device), torch.from_numpy(pairs[:, 1]).to(device)
                    S[r_, c_] = prior[c_]
                v2t["candidate_prior"] = S
            else:
                v2t["candidate_prior"] = run_pass(full(num_videos, num_texts), v2t_iv2[start:end], start, True, "vtg", True)
        if finetuned:
            v2t["query_likelihood"] = run_pass(full(num_videos, num_texts), v2t_iv2[start:end], start, True, "tvg", False)
        v_block = (start, end)
        start, end = dist_utils.row_block(num_texts, W, rank)                                        # :233-235
        t2v["query_likelihood"] = run_pass(full(num_texts, num_videos), t2v_iv2[start:end], start, False, "vtg", False)
        if finetuned:
            t2v["candidate_likelihood"] = run_pass(full(num_texts, num_videos), t2v_iv2[start:end], start, False, "tvg", False)
            if args.cpn:
                t2v["candidate_prior"] = run_pass(full(num_texts, num_videos), t2v_iv2[start:end], start, False, "tvg", True)
        t_block = (start, end)

        if collective:                                                                               # :252-262
            merge([(v2t, v_block), (t2v, t_block)])
    mark("queued")
    t2v_dict = {k: v.cpu().numpy() for k, v in t2v.items()}                                      # :264-276
    v2t_dict = {k: v.cpu().numpy() for k, v in v2t.items()}
    mark("done")
    if isinstance(scorer, PairScorer):
        stats["executed_flops"] = scorer.exec_flops; stats["executed_tokens"] = scorer.exec_tokens
        stats["executed_flops_lo6"] = getattr(scorer, "exec_flops_lo6", 0.0)
    args._eval_stats = dict(stats, seconds=time.time() - t_start, world=W, rank=rank, host_marks=marks)
    if getattr(args, "keep_tvg_feats", False) and isinstance(scorer, PairScorer):                # bench.py: what a later shard emulation adopts as its peers' blocks
        args._tvg_feats = {j: f for (j, tvg_), f in scorer._vfeat.items() if tvg_}
    t2v_dict["internvideo2"] = t2v_iv2.cpu().numpy()
    v2t_dict["internvideo2"] = v2t_iv2.cpu().numpy()
    if getattr(args, "verbose", False) and rank == 0:
        print(f"Evaluation time {time.time() - t_start:.1f}s")
    return t2v_dict, v2t_dict
