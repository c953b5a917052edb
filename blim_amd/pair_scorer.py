"""`PairScorer` -- the fused scoring path `retrieval_utils.evaluation` runs by default -- and its planner (split out of retrieval_utils.py in round 6, VERDICT r5 item 7;
retrieval_utils re-exports every name).

A likelihood is a function of the (video, text) pair only, so pairs from many queries are packed into large token batches; tokens that are identical for every
candidate of a query (the video + prompt prefix for VTG, the caption prompt for TVG) are computed once and their K/V reused (SURVEY.md section 7 "prefix-KV reuse");
hidden states that no score reads (tail tokens) are not computed; priors that do not depend on the query (v2t VTG-CPN) are computed once per
candidate.  Replaces the per-query / per-batch loops of /root/reference/retrieval_utils.py:48-153 (which `retrieval_utils.compute_*_scores_x` keep literally).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import distributed as dist_utils
from .calibration import TVG_MODES, VTG_MODES, VTG_SPLIT_MODES, CalibrationMixin
from .engine import PackedBatch
from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX


def _clip_major_vocab(video_vocab, device, dtype):
    """[N, clips, M] -> 16-bit [clips, N, M] on device (layout blim_tvg_* expects)."""
    return video_vocab.to(device=device, dtype=dtype).permute(1, 0, 2).contiguous()


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


class PairScorer(CalibrationMixin):
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
        if getattr(self.engine, "lo6", False) and self.split_tvg and getattr(self.engine, "dtype", "") != "bf16":      # (bf16 engines: TVG calls keep the bf16 second pass, Engine.set_precise)
            self.exec_flops_lo6 += lo6_pass_flops(self.m.dims, plan.n_tokens, plan.n_rows, "tvg", self.tvg_mode, prune=not f8)
        if self.vocab_cm is None and getattr(self.engine, "_vocab_key", None) != self._vocab_key:
            self.engine.set_video_vocab(self._vocab_src)                     # another scorer / the literal path registered its own vocabulary since
        # TVG calls: compensated (3-5 new tokens per pair: cheap); how much of the MLP branch is compensated follows tvg_mode (calibrate_tvg)
        self.engine.set_precise(self.split_tvg, embeds=self.split_tvg, mlp=self.tvg_mode != "attn", tvg=True)
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

    def set_tvg_mode(self, mode) -> None:
        if mode not in TVG_MODES:
            raise ValueError(f"tvg mode {mode!r}: one of {TVG_MODES}")
        self.tvg_mode = mode

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
