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

import math
import sys
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import distributed as dist_utils
from . import engine as eng
from .engine import PackedBatch
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


def _clip_major_vocab(video_vocab, device, dtype):
    """[N, clips, M] -> 16-bit [clips, N, M] on device (layout blim_tvg_* expects)."""
    return video_vocab.to(device=device, dtype=dtype).permute(1, 0, 2).contiguous()


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


# ----------------------------------------------------------------------------- fused path

@dataclass
class Plan:
    """One engine call: packed batch + row bookkeeping, all device-resident."""
    kind: str                  # "vtg" | "tvg"
    batch: PackedBatch
    src_index: object          # int32 [n_tokens]  (assemble input)
    feats: object              # bf16 [n_feat_rows, H]
    rows: object               # int32
    labels: object             # int32 (vtg: [n_rows] token ids; tvg: [n_pairs] video labels)
    row_start: Optional[object]
    n_pairs: int
    out_index: np.ndarray      # host: which requested pair each scored pair answers (many-to-one allowed)
    n_tokens: int
    n_rows: int


def _split_prompt_response(ids: np.ndarray, labels: np.ndarray):
    """ids/labels of one row (left pad stripped) -> (prompt ids, response ids) with labels == -100 on the prompt."""
    resp = labels != IGNORE_INDEX
    n_prompt = int(np.argmax(resp)) if resp.any() else len(ids)
    assert resp[n_prompt:].all(), "response must be one trailing span"
    return ids[:n_prompt], ids[n_prompt:]


def executed_flops(dims, n_tokens: int, n_rows: int, kind: str, mode=None, n_vocab: int = 0, prune: bool = True) -> float:
    """GEMM FLOPs one engine call EXECUTES (the per-token constants applied to the token counts actually launched; attention,
    < 1 - 3 %, excluded): decoder layers over n_tokens packed tokens + the head over n_rows scored rows.  `mode`: compensation of the call (None: plain;
    "attn": QKV, o_proj and the head take hi + lo inputs; "full": every GEMM -- a TVG call of a 16-bit engine runs in one of the last two, TVG_MODES; a VTG
    call plain or "full", VTG_MODES).  prune: the last layer's o_proj / MLP run on the scored rows only (engine option prune_last) when they are < 15/16 of the tokens."""
    H, I = dims.hidden_size, dims.intermediate_size
    q = 2.0 * H * (dims.num_heads + 2 * dims.num_kv_heads) * dims.head_dim
    o, gu, d = 2.0 * H * H, 4.0 * H * I, 2.0 * H * I
    fq = fo = 2.0 if mode in ("attn", "full") else 1.0      # "attn" (TVG calls only): the attention branch and the scored rows compensated, the MLP branch plain
    fg = fd = 2.0 if mode == "full" else 1.0
    per_tok = fq * q + fo * o + fg * gu + fd * d
    total = dims.num_layers * per_tok * n_tokens
    if prune and n_rows <= n_tokens - n_tokens // 16:
        total -= (fo * o + fg * gu + fd * d) * (n_tokens - n_rows)
    if kind == "vtg":
        total += fo * 2.0 * H * dims.vocab_size * n_rows
    else:
        # the visual head and the product with the video vocabulary: three-term compensated products (one GEMM of depth 3 K each) on a compensated call
        total += (3.0 if fo == 2.0 else 1.0) * (2.0 * H * dims.mm_hidden_size + 2.0 * dims.mm_hidden_size * n_vocab) * n_rows
    return total


def lo6_pass_flops(dims, n_tokens: int, n_rows: int, kind: str, mode=None, prune: bool = True) -> float:
    """The part of executed_flops() that runs on the e2m3 MFMA when the engine's option "precise_lo6" is on (fp16 engines, default): the second walk over K of the
    decoder GEMMs and of lm_head in the compensated modes (the TVG head's three-term products are 16-bit GEMMs of depth 3 K).  A roofline for such a call prices
    these flops at the fp6 peak (4x the 16-bit one) and the rest at the 16-bit one."""
    if mode not in ("attn", "full"):
        return 0.0
    H, I = dims.hidden_size, dims.intermediate_size
    q = 2.0 * H * (dims.num_heads + 2 * dims.num_kv_heads) * dims.head_dim
    o, gu, d = 2.0 * H * H, 4.0 * H * I, 2.0 * H * I
    g2 = gu if mode == "full" else 0.0
    d2 = d if mode == "full" else 0.0
    total = dims.num_layers * (q + o + g2 + d2) * n_tokens
    if prune and n_rows <= n_tokens - n_tokens // 16:
        total -= (o + g2 + d2) * (n_tokens - n_rows)
    if kind == "vtg":
        total += 2.0 * H * dims.vocab_size * n_rows
    return total


TVG_MODES = ("attn", "full")     # compensation of the TVG calls (always hi + lo embeddings, QKV, attention, o_proj, head), cheapest first: "attn" leaves the MLP branch plain
                                 # (1.6x faster than full), "full" compensates everything
VTG_MODES = ("none", "full")     # compensation of the VTG calls: plain 16-bit, or every activation as hi + lo (0.67x the plain rate on fp16 engines with the e2m3 second
                                 # pass).  Round 4 had four modes between the two (qk, qkx, attn, act0: 0.975 ... 0.70x); on the weight sets where plain fails they
                                 # either failed too or sat at the edge of the bar, and each was a kernel variant, an engine option and a calibration pass: removed.
VTG_SPLIT_MODES = ("full",)      # modes whose VTG rows (embeddings, features) travel as [hi | lo]


def predicted_max_deviation(dev, n_eval: Optional[int]) -> float:
    """The largest relative deviation to expect among the `n_eval` entries of a whole evaluation, from a SAMPLE of deviations (`--vtg_precise` / `--tvg_precise
    auto`).  The deviations of a cheap numeric mode from the fully compensated one are not Gaussian on weights with massive activations: over the 16,000 v2t VTG
    entries of an N = 1,000 evaluation on the heavy7b weights they follow a log-normal law to within a few percent from the median to the maximum (median 9.7e-5,
    99 % 7.9e-4, 99.9 % 1.7e-3, max 2.6e-3: sigma_log = 0.90; profiles/r04_auto_tail_validation.md), so the largest of 48,000 entries is ~ 20 x the rms where a
    Gaussian would give 4.3 x -- a 256-pair sample cannot SEE that tail (its own maximum read 7.3e-4), but it pins the law: least-squares line through the upper
    half of the sample's order statistics in (normal quantile, log deviation) coordinates, read off at the quantile 1 - 1 / n_eval.  For genuinely Gaussian
    deviations the same fit overshoots by ~ 2 x (8 x rms at n_eval = 48,000): conservative, never optimistic.  n_eval <= the sample size (the tests' small
    fixtures, where the sample IS the evaluation): the sample maximum itself."""
    x = np.asarray(dev, dtype=np.float64).reshape(-1)
    x = np.sort(x[np.isfinite(x) & (x > 0)])
    n = len(x)
    if n == 0:
        return 0.0
    if n_eval is None or n_eval <= n or n < 32:
        return float(x[-1])
    from statistics import NormalDist
    inv = NormalDist().inv_cdf
    k = np.arange(n // 2, n)
    zq = np.array([inv((i + 0.5) / n) for i in k])
    slope, icpt = np.polyfit(zq, np.log(x[k]), 1)
    return float(max(x[-1], math.exp(icpt + slope * inv(1.0 - 1.0 / float(n_eval)))))


def calibration_pairs(v2t_sims, topk: int, n_queries: int = 16, per_query: int = 16) -> np.ndarray:
    """(video, text) pairs `--vtg_precise auto` measures on: the top candidates of a few query videos spread over the test set (up to 256 pairs:
    under a second in all five modes at 7B size) -- the same pairs on every rank (the choice must not depend on the rank)."""
    import torch
    sims = torch.as_tensor(v2t_sims)
    Nv, Nt = sims.shape
    q = np.unique(np.linspace(0, Nv - 1, num=min(n_queries, Nv)).round().astype(np.int64))
    k = min(Nt, topk, per_query)
    idx = sims[torch.from_numpy(q)].topk(k=k, dim=1).indices.cpu().numpy()
    return np.stack([np.repeat(q, k), idx.reshape(-1)], axis=1)


class PairScorer:
    """Fused scoring of arbitrary (video, text) pairs.  See the module docstring for what is shared."""

    def __init__(self, model, vtg_ids, vtg_masks, vtg_labels, tvg_ids, tvg_masks, tvg_labels, video: Sequence, video_vocab,
                 tvg_video_labels, num_clips: int, max_tokens: int = 24576, precise_tvg: bool = True, feat_chunk: int = 64):
        import torch
        self.precise_tvg = bool(precise_tvg)
        eng_ = getattr(model.module if hasattr(model, "module") else model, "engine", None)
        self.split_tvg = self.precise_tvg and eng_ is not None and bool(getattr(eng_, "can_precise", False))   # TVG rows as [hi | lo]
        # VTG calls: plain 16-bit on fp16 engines; bf16 engines run them compensated too (modeling.py: vtg_precise), feature rows included --
        # with plain bf16 features the projector's 8-bit rounding alone left 1e-3 on the scores at 7B depth
        m_ = model.module if hasattr(model, "module") else model
        can = eng_ is not None and bool(getattr(eng_, "can_precise", False))
        vm = (m_.vtg_mode() if hasattr(m_, "vtg_mode") else getattr(m_, "vtg_precise", None)) if can else None
        self.vtg_mode = None if vm in ("auto", "none") else vm           # an unresolved "auto": plain until calibrate_vtg decides (evaluation() does it before the first pass)
        tm = m_.tvg_mode() if hasattr(m_, "tvg_mode") else getattr(m_, "tvg_precise", None)
        self.tvg_mode = tm if tm in TVG_MODES else "full"                # (an unresolved "auto": full until calibrate_tvg says otherwise)
        self.split_vtg = self.vtg_mode in VTG_SPLIT_MODES
        self.m = model.module if hasattr(model, "module") else model
        self.engine = self.m.engine
        self.device = self.m.device
        self.max_tokens = int(max_tokens)
        self.num_clips = int(num_clips)
        if self.num_clips != int(self.m.dims.num_clips):         # blim_score_tvg reads n_pairs * blim_config.num_clips rows
            raise ValueError(f"num_clips = {num_clips} but the engine was created with num_clips = {self.m.dims.num_clips}")
        strip = lambda ids, msk, lab: [(np.asarray(ids[i])[np.asarray(msk[i]) != 0], np.asarray(lab[i])[np.asarray(msk[i]) != 0])
                                       for i in range(len(ids))]
        self.vtg_rows = strip(vtg_ids, vtg_masks, vtg_labels)
        self.tvg_rows = strip(tvg_ids, tvg_masks, tvg_labels)
        self.video = video
        self.tvg_video_labels = np.asarray(tvg_video_labels).astype(np.int32)
        # 16-bit engines: the vocabulary is registered with the engine as hi + lo operands (blim_set_video_vocab) and the TVG calls name none; fp8 engines take the
        # plain 16-bit clip-major copy
        self.vocab_cm, self.n_vocab, self._vocab_src, self._vocab_key = None, 0, None, None
        if video_vocab is not None:
            self.n_vocab = int(video_vocab.shape[0])
            if self.split_tvg and hasattr(self.engine, "set_video_vocab"):
                self._vocab_src = video_vocab
                self.engine.set_video_vocab(video_vocab)
                self._vocab_key = self.engine._vocab_key
            else:
                self.vocab_cm = _clip_major_vocab(video_vocab, self.device, self.m.dtype)
        self.exec_flops = 0.0            # GEMM FLOPs of the engine calls run so far (executed_flops; bench.py's roofline fractions)
        self.exec_flops_lo6 = 0.0        # ... of which on the e2m3 MFMA (lo6_pass_flops: the compensated modes' second pass under the engine's "precise_lo6")
        self.exec_tokens = 0
        self._vfeat: Dict[Tuple[int, bool], object] = {}
        self._upcoming: Dict[bool, List[int]] = {}; self._upcoming_pos: Dict[bool, int] = {}
        self.feat_chunk = int(feat_chunk)
        # per-text splits
        self.vtg_split = []
        for ids, lab in self.vtg_rows:
            prompt, resp = _split_prompt_response(ids, lab)
            w = np.nonzero(prompt == IMAGE_TOKEN_INDEX)[0]
            assert len(w) == 1, "VTG prompt must hold exactly one <image> placeholder"
            self.vtg_split.append((prompt[: w[0]].astype(np.int64), prompt[w[0] + 1:].astype(np.int64), resp.astype(np.int64)))
        # rows longer than config.tokenizer_model_max_length lose their tail after the splice (modeling_videochat_flash.py:452-457; None = no limit)
        self.max_row_len = getattr(self.m, "tokenizer_model_max_length", None)
        self.tvg_split = []
        for ids, lab in self.tvg_rows:
            prompt, resp = _split_prompt_response(ids, lab)
            assert len(resp) >= 1 and resp[0] == IMAGE_TOKEN_INDEX, "TVG response must start with the <image> placeholder"
            if self.max_row_len is not None and len(ids) - 1 + self.num_clips > self.max_row_len:
                # the reference reads the clip positions relative to the <|im_end|> label of the row's tail (retrieval_utils.py:99-107); a row cut
                # inside its clip tokens or tail has no such label any more
                raise ValueError(f"TVG row of {len(ids) - 1 + self.num_clips} tokens exceeds tokenizer_model_max_length = {self.max_row_len}")
            self.tvg_split.append(prompt.astype(np.int64))

    # ---- projected video features, cached on device (K1 once per video instead of once per pair)
    def video_feat(self, j: int, tvg: bool):
        """Projected feature rows of video j, cached on device.  TVG rows (clip means) are produced in the compensated mode when the TVG
        calls run in it: [clips, 2H] rows of hi | lo -- at 7B depth the 16-bit rounding of the projector output was the largest remaining
        error of the TVG scores.

        A miss projects a CHUNK: video j together with the next videos the running pass will ask for (`expect`), one upload and one
        projector call per `feat_chunk` videos.  A projected row depends on its own input row only, so the values are those of a
        per-video call; what changes is the fixed cost -- one upload + six launches per video was 0.4 ms x N on EVERY rank of a sharded
        evaluation (every rank needs the clip features of nearly all videos), the part of the job that did not shrink with the world size."""
        key = (int(j), bool(tvg))
        f = self._vfeat.get(key)
        if f is None:
            self._project_chunk(int(j), bool(tvg))
            f = self._vfeat[key]
        return f

    def expect(self, video_ids, tvg: bool) -> None:
        """Order in which the pass being planned will first ask for its videos (chunked projection looks ahead along it)."""
        ids = np.asarray(video_ids, dtype=np.int64)
        _, first = np.unique(ids, return_index=True)
        self._upcoming[bool(tvg)] = [int(v) for v in ids[np.sort(first)]]
        self._upcoming_pos[bool(tvg)] = 0

    def share_tvg_feats(self, world: int, rank: int) -> bool:
        """Multi-GPU evaluations: every rank needs the TVG clip features (tvg_mlp projection + clip means, a few KB per video) of nearly ALL videos --
        its texts' candidates -- which left an upload + projection of N videos on every rank whatever the world size.  Instead each rank projects the
        videos of its own row block and ONE all-gather ([N / W + 1, clips, width] per rank; 57 MB in total at N = 1000) hands everyone the rest.  The
        values are those of a local projection (a projected row depends on its own input row only).  Returns False (nothing done) when the videos differ
        in shape or no process group is up; then video_feat() projects on demand as before."""
        import torch
        if (world <= 1 and not dist_utils.force_collective()) or not dist_utils.is_dist_avail_and_initialized() or not hasattr(self.m, "project_many"):
            return False
        N = len(self.video)
        if len({tuple(v.shape) for v in self.video}) != 1:
            return False
        s_, e_ = dist_utils.row_block(N, world, rank)
        step = N // world + 1
        self.expect(np.arange(s_, e_), True)
        mine = [self.video_feat(j, True) for j in range(s_, e_)]
        per, width = (int(mine[0].shape[0]), int(mine[0].shape[1])) if mine else (self.num_clips, self.m.dims.hidden_size * (2 if self.split_tvg else 1))
        buf = torch.zeros((step * per, width), dtype=self.m.dtype, device=self.device)
        if mine:
            buf[: (e_ - s_) * per] = torch.cat(mine, dim=0)
        parts = [torch.empty_like(buf) for _ in range(world)]
        torch.distributed.all_gather(parts, buf)
        for r in range(world):
            rs, re = dist_utils.row_block(N, world, r)
            for j in range(rs, re):
                self._vfeat[(j, True)] = parts[r][(j - rs) * per:(j - rs + 1) * per]
        return True

    def adopt_tvg_feats(self, world: int, rank: int, peers) -> bool:
        """Shard emulation's stand-in for share_tvg_feats (one process plays rank `rank` of `world`; there is nobody to gather from): the rank projects the
        videos of its OWN row block here, as share_tvg_feats would, and takes the other blocks' clip features from `peers` ({video index: [clips, width] device
        rows}, e.g. an earlier evaluation's, handed in by the caller) -- what the all-gather would have delivered; the all-gather's own time is NOT part of
        an emulated rank's clock.  Rows of another width (another TVG mode) are not adopted; such videos are projected on demand as before."""
        N = len(self.video)
        if world <= 1 or not peers or len({tuple(v.shape) for v in self.video}) != 1:
            return False
        s_, e_ = dist_utils.row_block(N, world, rank)
        self.expect(np.arange(s_, e_), True)
        mine = [self.video_feat(j, True) for j in range(s_, e_)]
        width = int(mine[0].shape[1]) if mine else self.m.dims.hidden_size * (2 if self.split_tvg else 1)
        for j, f in peers.items():
            if not (s_ <= j < e_) and (int(j), True) not in self._vfeat and int(f.shape[1]) == width and f.dtype == self.m.dtype:
                self._vfeat[(int(j), True)] = f
        return True

    def _project_chunk(self, j: int, tvg: bool) -> None:
        shape = tuple(self.video[j].shape)
        chunk = [j]
        up, pos = self._upcoming.get(tvg, []), self._upcoming_pos.get(tvg, 0)
        while pos < len(up) and len(chunk) < self.feat_chunk:
            v = up[pos]; pos += 1
            if v != j and (v, tvg) not in self._vfeat and tuple(self.video[v].shape) == shape:
                chunk.append(v)
        self._upcoming_pos[tvg] = pos
        split = self.split_tvg if tvg else self.split_vtg
        many = getattr(self.m, "project_many", None)
        if split:
            self.engine.set_precise(True, embeds=True)
        try:
            if many is not None:
                outs = many([self.video[v] for v in chunk], tvg)
            else:                                                            # a model surface with the per-video projector only
                outs = [self.m.project(self.video[v].to(self.device), tvg, cache=False) for v in chunk]
        finally:
            if split:
                self.engine.set_precise(False)
        for v, y in zip(chunk, outs):
            self._vfeat[(v, tvg)] = y

    # ---- planning (host) ------------------------------------------------------------------------
    def plan_vtg(self, pairs: np.ndarray, cpn: bool = False) -> List[Plan]:
        return list(self.iter_vtg(pairs, cpn))

    def plan_tvg(self, pairs: np.ndarray, cpn: bool = False) -> List[Plan]:
        return list(self.iter_tvg(pairs, cpn))

    def iter_vtg(self, pairs: np.ndarray, cpn: bool = False):
        """pairs: [P, 2] (video j, text i).  cpn=True: video keys masked -> the score depends on the text only.
        Yields one Plan per engine call, so that packing call k+1 (host) overlaps call k (device)."""
        return self.iter_vtg_jobs([(pairs, cpn)])

    def iter_vtg_jobs(self, jobs):
        """Several VTG passes -- [(pairs, cpn), ...] -- planned into the SAME engine calls (outputs concatenated in job order): a plan does not know which pass a
        sequence belongs to (a prior's prompt is its own sequence with the video's positions left out), so a rank's text-block prior rides in the last, partly
        filled call of its likelihood pass instead of being a latency-bound call of its own (iter_tvg_jobs: the TVG counterpart)."""
        items, base = [], 0
        for pairs, cpn in jobs:
            pairs = np.asarray(pairs, dtype=np.int64)
            items += self._vtg_items(pairs, bool(cpn), base)
            base += len(pairs)
        yield from self._pack_vtg(items)

    def _vtg_items(self, pairs: np.ndarray, cpn: bool, base: int):
        """Groups of one VTG pass: (video j or None for a prior, its token count, texts, output slots per text), output slot of pair p = base + p."""
        if cpn:
            texts, inv = np.unique(pairs[:, 1], return_inverse=True)
            # the prior masks the video keys but keeps their positions: every video must contribute the same number of tokens,
            # else the reference's per-pair cpn forward would differ between queries too
            nvs = {int(np.prod(self.video[int(j)].shape[-3:-1])) for j in np.unique(pairs[:, 0])}
            if len(nvs) != 1:
                raise ValueError(f"VTG candidate prior: the videos of this pass have different token counts {sorted(nvs)}; score them per count")
            nv = nvs.pop()
            groups: Dict[Tuple, List[int]] = {}
            for ti, i in enumerate(texts):
                pre, post, _ = self.vtg_split[int(i)]
                groups.setdefault((pre.tobytes(), post.tobytes()), []).append(ti)
            return [(None, nv, [int(texts[t]) for t in g], [base + np.nonzero(inv == t)[0] for t in g]) for g in groups.values()]
        order = np.lexsort((pairs[:, 1], pairs[:, 0]))
        self.expect(pairs[order, 0], False)
        items = []
        j_prev, cur = None, None
        for idx in order:
            j, i = int(pairs[idx, 0]), int(pairs[idx, 1])
            pre, post, _ = self.vtg_split[i]
            key = (j, pre.tobytes(), post.tobytes())
            if key != j_prev:
                cur = (j, None, [], [])
                items.append(cur); j_prev = key
            cur[2].append(i); cur[3].append(np.array([base + idx]))
        return items

    def _pack_vtg(self, items):
        # pack groups into super-batches
        st = _PackState(self, "vtg")
        for (j, nv, texts_g, outs_g) in items:
            pre, post, _ = self.vtg_split[texts_g[0]]
            n_vid = nv if j is None else int(self.video_feat(j, False).shape[0])
            need = len(pre) + (0 if j is None else n_vid) + len(post) + sum(max(len(self.vtg_split[i][2]) - 1, 0) for i in texts_g)
            if st.n_tok and st.n_tok + need > self.max_tokens:
                yield st.finish(); st = _PackState(self, "vtg")
            # prefix sequence
            if j is None:
                if len(pre) + len(post) == 0:
                    # the reference's rows always open with the ChatML header; with no visible token in front of the response the
                    # prior's first factor would be read from a fully masked video position (undefined attention row)
                    raise ValueError("VTG candidate prior (cpn=True) needs at least one prompt token besides the <image> placeholder")
                ptoks = np.concatenate([pre, post]); ppos = np.concatenate([np.arange(len(pre)), len(pre) + n_vid + np.arange(len(post))])
                p0 = st.add_seq(ptoks, ppos, np.ones(len(ptoks), np.uint8), None)
            else:
                fo = st.add_feat(self.video_feat(j, False))
                ptoks = np.concatenate([pre, -(1 + fo + np.arange(n_vid)), post])
                p0 = st.add_seq(ptoks, np.arange(len(ptoks)), np.ones(len(ptoks), np.uint8), None)
            plen = len(ptoks); ppos_end = len(pre) + n_vid + len(post)
            for i, outs in zip(texts_g, outs_g):
                resp = self.vtg_split[i][2]
                if self.max_row_len is not None and ppos_end + len(resp) > self.max_row_len:
                    if ppos_end >= self.max_row_len:                       # no label left: the reference's criterion divides 0 by 0 there
                        raise ValueError(f"tokenizer_model_max_length = {self.max_row_len} leaves no response token of text {i} ({ppos_end} prompt + video tokens)")
                    resp = resp[: self.max_row_len - ppos_end]             # :452-457: the row's tail is cut, the score averages the tokens that remain
                body = resp[:-1]                                           # the last response token predicts nothing
                rows = [p0 + plen - 1]
                if len(body):
                    s0 = st.add_seq(body, ppos_end + np.arange(len(body)), np.ones(len(body), np.uint8), (p0, plen))
                    rows += list(range(s0, s0 + len(body)))
                st.add_pair(rows, resp.astype(np.int32), outs)
        if st.n_pairs:
            yield st.finish()

    def iter_tvg(self, pairs: np.ndarray, cpn: bool = False):
        """pairs: [P, 2] (video j, text i); score = log P(video j | text i) (mean over clips)."""
        return self.iter_tvg_jobs([(pairs, cpn)])

    def iter_tvg_jobs(self, jobs):
        """Several TVG passes -- [(pairs, cpn), ...] -- planned into the SAME engine calls: outputs are concatenated in job order.  A plan does not know which pass
        a sequence belongs to (visibility is per token, a prior's prefix is its own sequence), so a small likelihood pass and its prior fill one call instead of
        leaving two partly filled ones -- what a rank's share of a sharded evaluation and the calibration sample consist of."""
        box, base = [_PackState(self, "tvg")], 0
        for pairs, cpn in jobs:
            pairs = np.asarray(pairs, dtype=np.int64)
            yield from self._plan_tvg(pairs, bool(cpn), box, base)
            base += len(pairs)
        if box[0].n_pairs:
            yield box[0].finish()

    def _plan_tvg(self, pairs: np.ndarray, cpn: bool, box, base: int):
        """Plans one TVG pass into the pack state box[0] (replaced whenever a call is full and yielded); output slot of pair p = base + p."""
        C = self.num_clips
        st = box[0]
        # The continuations of one prefix -- the C - 1 clip tokens of every candidate video of a text (prior: last prompt token + clip tokens) -- are
        # packed into ONE sequence whose segments do not see each other (blim_batch.own_start): the 32-query attention blocks are dense instead of
        # holding 3 - 4 queries each (2,919 -> ~500 blocks per 13,700-token call at the reference's shapes) and the planner adds one sequence per
        # group instead of one per pair.  SEG_MAX bounds a merged sequence (own-segment tiles below a query's segment are computed and masked).
        SEG_MAX = 256
        if cpn:
            # prior depends on (prompt length, last prompt token, first tvg_prefix_length tokens, video) only
            tp = self.m.tvg_prefix_length
            keyed: Dict[Tuple, List[int]] = {}
            for idx, (j, i) in enumerate(pairs):
                pr = self.tvg_split[int(i)]
                keyed.setdefault((pr[:tp].tobytes(), len(pr), int(pr[-1]), int(j)), []).append(idx)
            by_prefix: Dict[bytes, List[Tuple]] = {}
            for k, v in keyed.items():
                by_prefix.setdefault(k[0], []).append((k, v))
            self.expect([k[3] for lst in by_prefix.values() for (k, _) in lst], True)
            for pbytes, lst in by_prefix.items():
                ptoks = np.frombuffer(pbytes, dtype=np.int64)
                pos_in, p0 = 0, None
                while pos_in < len(lst):
                    room = (self.max_tokens - st.n_tok - (len(ptoks) if p0 is None else 0)) // C
                    if st.n_tok and room < 1:
                        yield st.finish(); st = box[0] = _PackState(self, "tvg"); p0 = None
                        room = (self.max_tokens - len(ptoks)) // C
                    n = max(1, min(len(lst) - pos_in, room, SEG_MAX // C))
                    if p0 is None:                       # the prefix is packed once per engine call; every merged sequence of the group names it
                        p0 = st.add_seq(ptoks, np.arange(len(ptoks)), np.ones(len(ptoks), np.uint8), None)
                    toks, posn, vis, own = [], [], [], []
                    for m_, (k, outs) in enumerate(lst[pos_in:pos_in + n]):
                        _, plen_full, last_tok, j = k
                        fo = st.add_feat(self.video_feat(j, True))
                        toks.append(np.concatenate([[last_tok], -(1 + fo + np.arange(C - 1))]))
                        posn.append(plen_full - 1 + np.arange(C))
                        vis.append(np.concatenate([[1 if plen_full - 1 < tp else 0], np.ones(C - 1)]).astype(np.uint8))
                        own.append(np.full(C, m_ * C, np.int32))
                    s0 = st.add_seq(np.concatenate(toks), np.concatenate(posn), np.concatenate(vis), (p0, len(ptoks)), own_start=np.concatenate(own))
                    for m_, (k, outs) in enumerate(lst[pos_in:pos_in + n]):
                        st.add_pair(list(range(s0 + m_ * C, s0 + (m_ + 1) * C)), np.array([self.tvg_video_labels[k[3]]], np.int32), base + np.array(outs))
                    pos_in += n
        else:
            order = np.lexsort((pairs[:, 0], pairs[:, 1]))
            self.expect(pairs[order, 0], True)
            # candidates of each text, in order
            groups: List[Tuple[int, List[int]]] = []
            for idx in order:
                i = int(pairs[idx, 1])
                if not groups or groups[-1][0] != i:
                    groups.append((i, []))
                groups[-1][1].append(int(idx))
            for i, idxs in groups:
                pr = self.tvg_split[i]
                plen = len(pr)
                pos_in, p0 = 0, None
                while pos_in < len(idxs):
                    per = max(C - 1, 1)
                    room = (self.max_tokens - st.n_tok - (plen if p0 is None else 0)) // per
                    if st.n_tok and room < 1:
                        yield st.finish(); st = box[0] = _PackState(self, "tvg"); p0 = None
                        room = (self.max_tokens - plen) // per
                    n = max(1, min(len(idxs) - pos_in, room, SEG_MAX // per))
                    if p0 is None:                       # the prompt is packed once per engine call; every merged sequence of the text names it
                        p0 = st.add_seq(pr, np.arange(plen), np.ones(plen, np.uint8), None)
                    chunk = idxs[pos_in:pos_in + n]
                    s0 = None
                    if C > 1:
                        toks, own = [], []
                        for m_, idx in enumerate(chunk):
                            fo = st.add_feat(self.video_feat(int(pairs[idx, 0]), True))
                            toks.append(-(1 + fo + np.arange(C - 1)))
                            own.append(np.full(C - 1, m_ * (C - 1), np.int32))
                        s0 = st.add_seq(np.concatenate(toks), np.tile(plen + np.arange(C - 1), n), np.ones(n * (C - 1), np.uint8), (p0, plen),
                                        own_start=np.concatenate(own))
                    for m_, idx in enumerate(chunk):
                        rows = [p0 + plen - 1]
                        if C > 1:
                            rows += list(range(s0 + m_ * (C - 1), s0 + (m_ + 1) * (C - 1)))
                        else:
                            self.video_feat(int(pairs[idx, 0]), True)
                        st.add_pair(rows, np.array([self.tvg_video_labels[int(pairs[idx, 0])]], np.int32), np.array([base + idx]))
                    pos_in += n

    # ---- execution (device) ---------------------------------------------------------------------
    def run(self, plan: Plan):
        """One engine call; returns a device f32 tensor [plan.n_pairs]."""
        self.exec_tokens += plan.n_tokens
        f8 = getattr(self.engine, "dtype", "") == "f8"
        if plan.kind == "vtg":
            mode = self.vtg_mode                                             # None | "full"
            self.exec_flops += executed_flops(self.m.dims, plan.n_tokens, plan.n_rows, "vtg", mode, prune=not f8)
            if getattr(self.engine, "lo6", False):
                self.exec_flops_lo6 += lo6_pass_flops(self.m.dims, plan.n_tokens, plan.n_rows, "vtg", mode, prune=not f8)
            comp = mode in VTG_SPLIT_MODES
            self.engine.set_precise(comp, embeds=comp, mlp=True)
            try:
                embeds = self.engine.assemble(plan.src_index, plan.feats)
                return self.engine.score_vtg(plan.batch, embeds, plan.rows, plan.labels, plan.row_start)
            finally:
                self.engine.set_precise(False)
        self.exec_flops += executed_flops(self.m.dims, plan.n_tokens, plan.n_rows, "tvg", self.tvg_mode if self.split_tvg else None,
                                          n_vocab=self.n_vocab, prune=not f8)
        if getattr(self.engine, "lo6", False) and self.split_tvg:
            self.exec_flops_lo6 += lo6_pass_flops(self.m.dims, plan.n_tokens, plan.n_rows, "tvg", self.tvg_mode, prune=not f8)
        if self.vocab_cm is None and getattr(self.engine, "_vocab_key", None) != self._vocab_key:
            self.engine.set_video_vocab(self._vocab_src)                     # another scorer / the literal path registered its own vocabulary since
        # TVG calls: compensated (3-5 new tokens per pair: cheap); how much of the MLP branch is compensated follows tvg_mode (calibrate_tvg)
        self.engine.set_precise(self.split_tvg, embeds=self.split_tvg, mlp=self.tvg_mode != "attn")
        try:
            embeds = self.engine.assemble(plan.src_index, plan.feats)
            return self.engine.score_tvg(plan.batch, embeds, plan.rows, self.vocab_cm, plan.labels)
        finally:
            self.engine.set_precise(False)

    def score(self, plans, n_requested: int) -> np.ndarray:
        """plans: list or generator of Plan.  Engine calls are asynchronous, so with a generator the host packs plan k+1 while
        the device runs plan k; the scores are copied back once, at the end."""
        out = np.full(n_requested, np.nan, dtype=np.float32)
        done = [(p.out_index, self.run(p)) for p in plans]
        for out_index, r in done:
            sc = r.float().cpu().numpy()
            for k, outs in enumerate(out_index):
                out[outs] = sc[k]
        return out

    def score_device(self, plans, n_requested: int):
        """As score(), but the result stays on the device (f32 [n_requested], NaN where nothing was scored) and nothing
        synchronises: evaluation() scatters it into the score matrix on the device, so the host goes straight on to plan the next
        pass while the device is still running this one."""
        import torch
        out = torch.full((n_requested,), float("nan"), dtype=torch.float32, device=self.device)
        res, src, dst, base = [], [], [], 0
        for p in plans:                                           # every engine call of the pass is queued first ...
            res.append(self.run(p))
            src.append(base + np.repeat(np.arange(len(p.out_index)), [len(o) for o in p.out_index]))
            dst.append(np.concatenate(p.out_index) if len(p.out_index) else np.zeros(0, np.int64))
            base += p.n_pairs
        if res:                                                   # ... then one index upload and one device-side scatter
            idx = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)).to(self.device)
            out[idx[1]] = torch.cat(res).float()[idx[0]]
        return out

    # ---- which compensation the VTG calls need (`--vtg_precise auto`) ---------------------------------------------------------------
    def set_vtg_mode(self, mode) -> None:
        """Compensation of this scorer's following VTG calls: None | "full".  The cached VTG feature rows are dropped when their layout changes ([hi | lo] rows in
        the compensated mode).  (The model's own record of what `auto` resolved to is BlimModel.resolve_vtg: calibrate_vtg and evaluation() write it.)"""
        mode = None if mode in (None, "none") else mode
        if mode not in (None,) + VTG_MODES[1:]:
            raise ValueError(f"vtg mode {mode!r}: one of {VTG_MODES}")
        if not bool(getattr(self.engine, "can_precise", False)):
            mode = None
        split = mode in VTG_SPLIT_MODES
        if split != self.split_vtg:
            self._vfeat = {k: v for k, v in self._vfeat.items() if k[1]}
        self.vtg_mode, self.split_vtg = mode, split

    def _gather_dev(self, dev: np.ndarray, share) -> np.ndarray:
        """Multi-rank calibration: every rank scored its own block of the sample; all ranks get all deviations (one all-gather of <= 256 floats)."""
        import torch
        W = int(share[0]) if share is not None else 0
        if share is None or not dist_utils.is_dist_avail_and_initialized() or W != torch.distributed.get_world_size():
            # one process playing rank r of W (`--shard`, bench.py's emulated ranks and its warm-up, also inside a real job): nobody to gather from -- it decides on its
            # own block; the COST of a rank's share of the calibration is what such a run stands for
            return dev
        n = int(share[2])                                                   # the largest block
        buf = torch.full((n,), -1.0, dtype=torch.float64, device=self.device)            # padding: -1 (a deviation is >= 0; a non-finite one travels as +inf and rejects the mode)
        dev = np.where(np.isfinite(dev), dev, np.inf)
        buf[: len(dev)] = torch.from_numpy(np.ascontiguousarray(dev, dtype=np.float64)).to(self.device)
        parts = [torch.empty_like(buf) for _ in range(W)]
        torch.distributed.all_gather(parts, buf)
        out = torch.cat(parts).cpu().numpy()
        return out[out >= 0.0]

    @staticmethod
    def _my_block(pairs: np.ndarray, share):
        """share = (world, rank): this rank's contiguous block of the sample (whole queries stay together: their prefix is computed once) -> (block, share + largest block)."""
        if share is None or (share[0] <= 1 and not dist_utils.force_collective()):       # (world size 1 with BLIM_FORCE_COLLECTIVE=1: the gather runs, through RCCL, on one block)
            return pairs, None
        W, r = int(share[0]), int(share[1])
        blocks = np.array_split(np.arange(len(pairs)), W)
        return pairs[blocks[r]], (W, r, max(len(b) for b in blocks))

    def calibrate_vtg(self, pairs, bar: float = 1e-3, z: float = 4.5, n_eval: Optional[int] = None, tail_margin: float = 0.8, share=None):
        """The reference has ONE numeric mode (training_utils.py:142: autocast fp16) and no decision to make; this engine's plain 16-bit VTG
        calls are the fastest of five modes, and whether they hold the 1e-3 bar depends on the checkpoint's statistics (attention sinks,
        massive activations: tests/golden/sink.npz).  So the decision is MEASURED on the loaded weights: `pairs` (up to 256 (video, text)
        pairs of the evaluation itself) are scored in every mode, cheapest first, against the fully compensated mode -- which sits at
        2e-6 .. 1e-5 of the fp32 reference on every fixture, i.e. is a yardstick that needs no oracle on the box -- and the cheapest mode
        that passes is kept.  The bar is per ENTRY of the whole evaluation while the calibration sees a sample, and the sample's maximum is a
        noisy statistic (on sink.npz the same mode reads 7e-4 or 1.2e-3 depending on last-bit differences upstream), so a mode passes when
        (a) the sample's largest relative deviation is inside the bar AND (b) z x the sample's RMS deviation is: for near-Gaussian deviations
        the largest of the ~10^4 .. 10^5 entries of an evaluation is 4 - 4.8 sigma; z = 4.5 -- AND (c) the largest deviation PREDICTED for the
        n_eval entries of the whole evaluation is (predicted_max_deviation: a log-normal tail fitted to the sample; on weights with massive activations
        the tail is that heavy, and (a) + (b) alone let modes through that left 0.1 - 0.6 % of an N = 1,000 evaluation's entries above the bar); the
        prediction has to stay inside tail_margin x bar: from 256 samples it lands at 0.74 - 1.8 x the true largest entry (tools/vtg_modes_population.py: 15
        mode x weight-set populations of 16,000 entries), and the one underestimate that would have let a mode through with an entry at 1.1e-3 read 0.82e-3.
        Returns (mode name, {mode: {max, rms, pred}} for the modes tried)."""
        pairs = np.asarray(pairs, dtype=np.int64)
        resolve = getattr(self.m, "resolve_vtg", lambda mode: None)
        if not bool(getattr(self.engine, "can_precise", False)):              # fp8 engines have no compensated modes: plain it is (and resolved: ADVICE r4)
            self.set_vtg_mode(None)
            resolve("none")
            return "none", {}
        n_all = len(pairs)
        pairs, share = self._my_block(pairs, share)                           # share = (world, rank): each rank scores its block, the deviations are all-gathered
        self.set_vtg_mode("full")
        ref = self.vtg(pairs).astype(np.float64) if len(pairs) else np.zeros(0)
        table = {}
        chosen = "full"
        limit = tail_margin * bar if (n_eval or 0) > n_all else bar
        for mode in VTG_MODES[:-1]:
            self.set_vtg_mode(mode)
            dev = np.abs(self.vtg(pairs).astype(np.float64) - ref) / np.abs(ref) if len(pairs) else np.zeros(0)
            dev = self._gather_dev(dev, share)
            table[mode] = {"max": float(np.max(dev)), "rms": float(np.sqrt(np.mean(dev * dev))), "pred": predicted_max_deviation(dev, n_eval)}
            if np.all(np.isfinite(dev)) and table[mode]["max"] <= bar and z * table[mode]["rms"] <= bar and table[mode]["pred"] <= limit:
                chosen = mode
                break                                                          # the dearer mode is not needed
        self.set_vtg_mode(chosen)
        resolve(chosen)
        return chosen, table

    def set_tvg_mode(self, mode) -> None:
        if mode not in TVG_MODES:
            raise ValueError(f"tvg mode {mode!r}: one of {TVG_MODES}")
        self.tvg_mode = mode

    def calibrate_tvg(self, pairs, bar: float = 1e-3, z: float = 4.5, n_eval: Optional[int] = None, tail_margin: float = 0.8, share=None):
        """The TVG calls' counterpart of calibrate_vtg (same criterion, same yardstick = the fully compensated mode).  Every TVG call of a 16-bit engine carries its
        embeddings, QKV, attention, o_proj and head as hi + lo; what is decided here is the MLP branch (87 % of the flops): `attn` leaves it plain (1.6x faster than
        `full`).  Gaussian-like weights need neither more than `attn` since the TVG head is exact
        (round 4); weights with massive residual channels need `full` (tests/golden/heavy7b.npz: the prior moved by 2.5e-3 with a plain SwiGLU output) -- measured per
        checkpoint on the likelihood AND the prior (the prior's queries see one prefix token and their own segment: the most sensitive pass)."""
        pairs = np.asarray(pairs, dtype=np.int64)
        resolve = getattr(self.m, "resolve_tvg", lambda mode: None)
        self.set_tvg_mode("full")
        if not self.split_tvg:                                             # fp8 / fp32-less engines: nothing to choose
            resolve("full")
            return "full", {}
        n_all = len(pairs)
        pairs, share = self._my_block(pairs, share)
        both = lambda: (self.score(self.iter_tvg_jobs([(pairs, False), (pairs, True)]), 2 * len(pairs)).astype(np.float64) if len(pairs) else np.zeros(0))   # likelihood and prior in the same engine calls
        ref = both()
        table, chosen = {}, "full"
        for mode in TVG_MODES[:-1]:
            self.set_tvg_mode(mode)
            got = both()
            d_ = np.abs(got - ref) / np.abs(ref) if len(pairs) else np.zeros(0)
            h_ = len(d_) // 2
            dl, dp = self._gather_dev(d_[:h_], share), self._gather_dev(d_[h_:], share)      # likelihood entries, prior entries: two laws, each extrapolated on its own
            dev, half = np.concatenate([dl, dp]), n_all
            pred = max(predicted_max_deviation(dl, n_eval), predicted_max_deviation(dp, n_eval))
            table[mode] = {"max": float(np.max(dev)), "rms": float(np.sqrt(np.mean(dev * dev))), "pred": pred}
            if np.all(np.isfinite(dev)) and table[mode]["max"] <= bar and z * table[mode]["rms"] <= bar and pred <= (tail_margin * bar if (n_eval or 0) > half else bar):
                chosen = mode
                break
        self.set_tvg_mode(chosen)
        resolve(chosen)
        return chosen, table

    def vtg(self, pairs, cpn=False) -> np.ndarray:
        return self.score(self.iter_vtg(pairs, cpn), len(pairs))

    def tvg(self, pairs, cpn=False) -> np.ndarray:
        return self.score(self.iter_tvg(pairs, cpn), len(pairs))

    def vtg_device(self, pairs, cpn=False):
        return self.score_device(self.iter_vtg(pairs, cpn), len(pairs))

    def vtg_jobs_device(self, jobs):
        """Several VTG passes through shared engine calls (iter_vtg_jobs); scores concatenated in job order."""
        return self.score_device(self.iter_vtg_jobs(jobs), sum(len(p) for p, _ in jobs))

    def tvg_device(self, pairs, cpn=False):
        return self.score_device(self.iter_tvg(pairs, cpn), len(pairs))

    def tvg_jobs_device(self, jobs):
        """Several TVG passes through shared engine calls (iter_tvg_jobs); scores concatenated in job order."""
        return self.score_device(self.iter_tvg_jobs(jobs), sum(len(p) for p, _ in jobs))


class _PackState:
    """Accumulates sequences / rows of one super-batch on the host, then uploads once."""

    def __init__(self, scorer: PairScorer, kind: str):
        self.s, self.kind = scorer, kind
        self.tok: List[np.ndarray] = []; self.pos: List[np.ndarray] = []; self.vis: List[np.ndarray] = []
        self.seq_start: List[int] = []; self.seq_len: List[int] = []; self.pfx_start: List[int] = []; self.pfx_len: List[int] = []
        self.own: List[np.ndarray] = []; self.any_own = False             # per token: first own-segment index it attends to (segmented sequences)
        self.feats: List[object] = []; self.feat_key: Dict[int, int] = {}; self.n_feat = 0
        self.rows: List[int] = []; self.labels: List[np.ndarray] = []; self.row_start: List[int] = [0]
        self.out_index: List[np.ndarray] = []
        self.n_tok = 0; self.n_pairs = 0

    def add_feat(self, f) -> int:
        k = f.data_ptr()
        if k in self.feat_key:
            return self.feat_key[k]
        off = self.n_feat
        self.feats.append(f); self.feat_key[k] = off; self.n_feat += int(f.shape[0])
        return off

    def add_seq(self, toks, pos, vis, prefix, own_start=None) -> int:
        start = self.n_tok
        self.tok.append(np.asarray(toks, np.int64)); self.pos.append(np.asarray(pos, np.int64)); self.vis.append(np.asarray(vis, np.uint8))
        if own_start is None:
            self.own.append(np.zeros(len(toks), np.int32))
        else:
            self.own.append(np.asarray(own_start, np.int32)); self.any_own = True
        self.seq_start.append(start); self.seq_len.append(len(toks))
        self.pfx_start.append(prefix[0] if prefix else 0); self.pfx_len.append(prefix[1] if prefix else 0)
        self.n_tok += len(toks)
        return start

    def add_pair(self, rows, labels, outs):
        self.rows += rows
        self.labels.append(np.asarray(labels, np.int32))
        self.row_start.append(len(self.rows))
        self.out_index.append(np.asarray(outs))
        self.n_pairs += 1

    def finish(self) -> Plan:
        import torch
        dev = self.s.device
        src = np.concatenate(self.tok).astype(np.int32)
        batch = PackedBatch(np.concatenate(self.pos), np.concatenate(self.vis), np.array(self.seq_start), np.array(self.seq_len),
                            np.array(self.pfx_start), np.array(self.pfx_len), device=dev, own_start=np.concatenate(self.own) if self.any_own else None)
        H = self.s.m.dims.hidden_size
        wide = self.s.split_tvg if self.kind == "tvg" else self.s.split_vtg                   # feature rows are [hi | lo]
        feats = torch.cat(self.feats, dim=0) if self.feats else torch.zeros((1, H * (2 if wide else 1)), dtype=self.s.m.dtype, device=dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        labels = np.concatenate(self.labels)
        return Plan(kind=self.kind, batch=batch, src_index=t(src), feats=feats, rows=t(np.array(self.rows)), labels=t(labels),
                    row_start=t(np.array(self.row_start)) if self.kind == "vtg" else None, n_pairs=self.n_pairs,
                    out_index=self.out_index, n_tokens=self.n_tok, n_rows=len(self.rows))


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

    mod = model.module
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
            cal_share = (W, rank) if ((collective and dist_utils.is_dist_avail_and_initialized()) or emulate is not None) else None   # every rank scores its block of the sample;
                                                                                                   # deviations all-gathered (shard emulation: the rank's own block, nobody to gather from)
            chosen, table = cal.calibrate_vtg(calibration_pairs(v2t_iv2, args.topk, n_queries=32, per_query=8), n_eval=n_eval_vtg, share=cal_share)      # 256 pairs, 32 distinct prefixes
            chosen = agree(chosen, VTG_MODES, lambda m_: (cal.set_vtg_mode(m_), getattr(mod, "resolve_vtg", lambda x: None)(m_)))
            stats["vtg_precise"] = chosen; stats["vtg_precise_table"] = table
            if rank == 0:
                print("vtg_precise auto: deviation from the fully compensated mode on the calibration pairs (max / rms): "
                      + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e} (predicted max {v['pred']:.1e})" for k, v in table.items()) + f" -> {chosen}", file=sys.stderr, flush=True)
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
            cal_share = (W, rank) if ((collective and dist_utils.is_dist_avail_and_initialized()) or emulate is not None) else None
            chosen, table = cal.calibrate_tvg(np.stack([tp[:, 1], tp[:, 0]], axis=1), n_eval=num_videos * kt_ + num_texts * kv_ * (2 if args.cpn else 1), share=cal_share)
            chosen = agree(chosen, TVG_MODES, lambda m_: (cal.set_tvg_mode(m_), getattr(mod, "resolve_tvg", lambda x: None)(m_)))
            stats["tvg_precise"] = chosen; stats["tvg_precise_table"] = table
            if rank == 0:
                print("tvg_precise auto: deviation from the fully compensated mode on the calibration pairs, likelihood + prior (max / rms): "
                      + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e} (predicted max {v['pred']:.1e})" for k, v in table.items()) + f" -> {chosen}", file=sys.stderr, flush=True)
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
