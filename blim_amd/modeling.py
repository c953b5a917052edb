"""Host-side mirror of the reference's model surface for the scoring path.

`BlimModel` plays the role of VideoChatFlashQwenForCausalLM on the eval path
(videochat_flash/modeling_videochat_flash.py:572-629): same method names, argument meaning and return
shapes for the calls retrieval_utils.py makes, with every tensor op executed by the HIP engine
(blim_amd/engine.py -> libblim_hip.so).  `DDPLike` provides the `.module` attribute the reference's
loops expect from DistributedDataParallel (retrieval_utils.py:66, 105, 210).

Only the `inputs_embeds` + `attention_mask` form of forward() is functional;
other argument combinations raise NotImplementedError.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import numpy as np

from .engine import Engine
from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX, ModelDims


class BlimModel:
    def __init__(self, dims: ModelDims, max_positions: int = 4096, tokenizer_model_max_length: Optional[int] = None,
                 dtype: Optional[str] = None):
        self.dims = dims
        self.engine = Engine(dims, max_positions=max_positions, dtype=dtype)
        self.dtype = self.engine.torch_dtype                        # torch dtype of activations (fp16 default, like the reference)
        self.device = self.engine.device
        self.tvg_prefix_length = 0
        self.video_vocab = None
        self.tokenizer_model_max_length = tokenizer_model_max_length
        self.training = False
        self._proj_cache = {}
        self._tvg_rows = False          # set by prepare_inputs_labels_for_multimodal(tvg=...): the next forward() is a TVG forward
        # Numeric modes.  What the USER asks for:
        #   vtg_precise: None = plain 16-bit VTG calls (what the reference's autocast computes; holds 1e-3 at 28 layers of the 7B configuration on N(0, 0.02^2) weights),
        #                "full" = every activation as hi + lo (what weights with a trained checkpoint's massive activations need; bf16 engines' default: 8-bit
        #                mantissas miss the bar when plain), "auto" = measured on the loaded checkpoint by evaluation() (PairScorer.calibrate_vtg);
        #   tvg_precise: the TVG calls always carry hi + lo embeddings, QKV, attention, o_proj and head; "attn" leaves their MLP branch plain (1.6x faster),
        #                "full" (library default) compensates it too, "auto" = measured (PairScorer.calibrate_tvg).
        # What "auto" RESOLVED to is kept apart (resolve_vtg / resolve_tvg) together with the engine's weights_version it was measured on: new weights or adapters --
        # another epoch of the training loop's validation, a reload -- make it unresolved again and the next evaluation() measures again.
        self._vtg_request = None
        self._tvg_request = "full"
        self._vtg_resolved = None                                   # (mode, engine.weights_version)
        self._tvg_resolved = None
        self.vtg_precise = os.environ.get("BLIM_VTG_PRECISE") or ("full" if self.engine.dtype == "bf16" else None)
        self.tvg_precise = os.environ.get("BLIM_TVG_PRECISE") or "full"
        if not self.engine.can_precise:                             # fp8 engines have no compensated modes: nothing to measure, nothing to ask for
            self._vtg_request, self._tvg_request = None, "full"

    @property
    def vtg_precise(self):
        return self._vtg_request

    @vtg_precise.setter
    def vtg_precise(self, mode):
        mode = None if mode in (None, "none", "0", "") else mode
        if mode not in (None, "full", "auto"):
            raise ValueError(f"vtg_precise = {mode!r}: one of none, full, auto (round 5 removed the intermediate modes qk / qkx / attn / act0)")
        self._vtg_request, self._vtg_resolved = mode, None

    @property
    def tvg_precise(self):
        return self._tvg_request

    @tvg_precise.setter
    def tvg_precise(self, mode):
        if mode not in ("attn", "full", "auto"):
            raise ValueError(f"tvg_precise = {mode!r}: one of attn, full, auto (round 5 removed act0)")
        self._tvg_request, self._tvg_resolved = mode, None

    @property
    def second_pass(self) -> str:
        """What the compensated calls' second walk over K (the activations' lo parts) runs in -- the REQUEST: "e2m3" (the block-scaled MFMA, gemm.hip phase 2: the default
        of fp16 engines; an opt-in on bf16 engines since round 6: 0.69x the plain rate, about one fp16 rounding's accuracy), "16bit" (a second walk in the engine's own
        format: the default and the parity mode of bf16 engines, 0.5x) or "auto" (bf16 engines: measured on the loaded checkpoint by evaluation(), like vtg_precise "auto":
        the e2m3 form is kept when its scores stay inside the bar of the 16-bit form's; PairScorer.calibrate_second_pass)."""
        return getattr(self, "_second_request", None) or ("e2m3" if bool(getattr(self.engine, "lo6", False)) else "16bit")

    @second_pass.setter
    def second_pass(self, mode) -> None:
        if mode not in ("e2m3", "16bit", "auto"):
            raise ValueError(f"second_pass = {mode!r}: one of e2m3, 16bit, auto")
        if not self.engine.can_precise:
            raise ValueError("second_pass: fp8 engines have no compensated modes")
        if mode == "auto" and self.engine.dtype != "bf16":
            mode = "e2m3"                                           # fp16 engines: nothing to measure -- the e2m3 pass is their compensated mode (<= 4e-5 from fp32 on every fixture)
        self._second_request, self._second_resolved = mode, None
        if mode != "auto":
            self.engine.set_option("precise_lo6", 1 if mode == "e2m3" else 0)
        else:
            self.engine.set_option("precise_lo6", 0)                # unresolved: the parity form
        self._vtg_resolved = self._tvg_resolved = None              # a measured `auto` was measured with the other second pass

    def resolve_second_pass(self, mode) -> None:
        """Records what second_pass = "auto" was measured to allow on the weights now loaded, and switches the engine to it."""
        self._second_resolved = (mode, self.engine.weights_version)
        self.engine.set_option("precise_lo6", 1 if mode == "e2m3" else 0)

    def second_pass_resolved(self) -> bool:
        r = getattr(self, "_second_resolved", None)
        return getattr(self, "_second_request", None) != "auto" or (r is not None and r[1] == self.engine.weights_version)

    @property
    def masked_query_zero(self) -> bool:
        return bool(getattr(self, "_masked_query_zero", False))

    @masked_query_zero.setter
    def masked_query_zero(self, on) -> None:
        """PARITY-UNPINNED switch (engine option of the same name, include/blim.h): masked query positions write a zero attention output, as the reference's
        flash-attention-2 class does (modeling_qwen2_flash.py:526-563); default off = its eager / SDPA classes, the semantics every golden vector was recorded with."""
        self.engine.set_option("masked_query_zero", int(bool(on)))
        self._masked_query_zero = bool(on)

    def resolve_vtg(self, mode) -> None:
        """Records what `vtg_precise = "auto"` was measured to need on the weights now loaded (evaluation() -> PairScorer.calibrate_vtg)."""
        self._vtg_resolved = (None if mode in (None, "none") else mode, self.engine.weights_version)

    def resolve_tvg(self, mode) -> None:
        self._tvg_resolved = (mode, self.engine.weights_version)

    def vtg_mode(self):
        """The mode VTG calls run in: the request, or -- for "auto" -- what it resolved to on the CURRENT weights; "auto" itself while unresolved."""
        if self._vtg_request != "auto":
            return self._vtg_request
        r = self._vtg_resolved
        return r[0] if (r is not None and r[1] == self.engine.weights_version) else "auto"

    def tvg_mode(self):
        """Likewise for the TVG calls; an unresolved "auto" runs fully compensated."""
        if self._tvg_request != "auto":
            return self._tvg_request
        r = self._tvg_resolved
        return r[0] if (r is not None and r[1] == self.engine.weights_version) else "full"

    def tvg_resolved(self) -> bool:
        return self._tvg_request != "auto" or (self._tvg_resolved is not None and self._tvg_resolved[1] == self.engine.weights_version)

    # ---- nn.Module-ish surface used by the eval loop
    def eval(self):
        self.training = False
        return self

    def set_video_vocab(self, video_vocab):                      # modeling_videochat_flash.py:589-590
        self.video_vocab = video_vocab

    def set_tvg_prefix_length(self, n: int):                     # modeling_videochat_flash.py:592-593
        self.tvg_prefix_length = int(n)

    def clear_cache(self):
        self._proj_cache.clear()

    # ---- K1 with a per-video cache (the reference re-projects identical copies, retrieval_utils.py:60)
    def project(self, feat, tvg: bool, cache: bool = True):
        """feat: [clips, T, mm_hidden] device tensor -> 16-bit [clips*T, H] (vtg) or [clips, H] (tvg: mean over T).

        The cache is keyed on the tensor's storage address AND keeps a reference to the tensor, so the address
        cannot be recycled for different data while the entry lives."""
        import torch
        key = (feat.data_ptr(), tuple(feat.shape), feat.dtype, bool(tvg), feat._version, bool(getattr(self.engine, "_precise_embeds", False)))
        if cache:
            hit = self._proj_cache.get(key)
            if hit is not None:
                return hit[1]
        x = feat.squeeze(0) if feat.ndim == 4 else feat          # :195 unsqueeze(0) / :157 squeeze(0)
        clips, T, M = x.shape
        x = x.to(device=self.device, dtype=self.dtype).reshape(clips * T, M).contiguous()
        y = self.engine.project_video(x, 1 if tvg else 0)
        if tvg:
            y = self.engine.group_mean(y, T)                     # :243 frame_feature.mean(1)
        if cache:
            if len(self._proj_cache) >= 256:
                self._proj_cache.pop(next(iter(self._proj_cache)))
            self._proj_cache[key] = (feat, y)
        return y

    def project_many(self, feats, tvg: bool):
        """K1 for a list of same-shaped per-video feature tensors in ONE projector call (a projected row depends on its own input row
        only: same values as project() per video).  The tensors are uploaded one by one -- a host-side stack of 64 x 1 MB costs 170 ms of
        torch CPU time on the GPU boxes against 4.5 ms for the 64 copies -- and gathered / converted on the device.  Returns one
        [clips*T, H] (vtg) or [clips, H] (tvg) view per video; in the engine's compensated mode the rows are [hi | lo] of width 2H."""
        import torch
        x = torch.stack([torch.as_tensor(f).to(self.device) for f in feats])
        x = x.reshape((len(feats),) + tuple(x.shape[-3:]))                   # [n, clips, T, M] (a leading 1 of the reference's unsqueeze dropped)
        n, clips, T, M = x.shape
        y = self.engine.project_video(x.to(self.dtype).reshape(n * clips * T, M), 1 if tvg else 0)
        if tvg:
            y = self.engine.group_mean(y, T)                                 # :243 frame_feature.mean(1)
        per = y.shape[0] // n
        return [y[k * per:(k + 1) * per] for k in range(n)]

    def forward_visual(self, visual_token_embeds):               # modeling_videochat_flash.py:598-599
        import torch
        shp = visual_token_embeds.shape
        x = visual_token_embeds.reshape(-1, shp[-1])
        if x.dtype == torch.float32 and self.engine.can_precise:
            # the literal API hands the final hidden states over as float32: keep their bits through the head as hi + lo 16-bit operands (two passes of a
            # [B * clips, H] x [H, mm_hidden] product) -- a plain 16-bit cast here was the largest error of the literal TVG scores on weights with a trained
            # checkpoint's dynamic ranges (heavy7b: 1.0e-3 fp16 / 1.2e-3 bf16 on the TVG prior)
            # (and float32 out: the head's 16-bit output rounding was the next largest term -- 3e-4 of the bf16 engine's 4e-4 on the literal TVG scores)
            out = self.engine.visual_head_f32(x.contiguous())
        else:
            out = self.engine.visual_head(x.to(self.dtype).contiguous()).float()
        return out.reshape(*shp[:-1], self.dims.mm_hidden_size)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                             modalities=["image"], image_sizes=None, video_feature=False, tvg=False, cpn=False):
        """modeling_videochat_flash.py:185-515, eval branch (video_feature=True, one <image> per row).

        input_ids / attention_mask / labels: LEFT-padded [B, Lt] device tensors; images: list of B feature tensors.
        Returns (None, position_ids, mask | (mask, cpn_mask), past_key_values, embeds [B,L,H], labels [B,L]).  embeds: the engine's 16-bit dtype for VTG rows on
        fp16 engines (the reference's contract); float32 (hi + lo formed in the compensated mode, one tensor of the same shape) for every row of a bf16 engine and for
        TVG rows of an fp16 engine -- forward() takes either."""
        import torch
        if not video_feature:
            raise NotImplementedError("only pre-extracted video features (video_feature=True) are supported")
        if images is None or input_ids.shape[1] == 1:
            raise NotImplementedError("text-only / single-token inputs are outside the scoring path")
        self._tvg_rows = bool(tvg)
        # bf16 engines (8-bit mantissas): the spliced embeddings are formed as hi + lo in the compensated mode and handed out as ONE float32 [B, L, H]
        # tensor (the reference hands its model dtype; a bf16 tensor here would by itself put ~1e-3 on the scores at 7B depth); forward() splits it again.
        # fp16 engines keep the reference's 16-bit [B, L, H] contract.
        # TVG rows on fp16 engines likewise: their forward runs compensated, and the clip tokens (means of projected features) rounded to fp16 alone left 1.0e-3 on
        # the literal TVG scores of heavy7b.npz; VTG rows on fp16 engines keep the 16-bit tensor (plain forward).
        wide = self.engine.can_precise and (self.engine.dtype == "bf16" or bool(tvg))
        if wide:
            self.engine.set_precise(True, embeds=True)
        try:
            return self._prepare(input_ids, position_ids, attention_mask, past_key_values, labels, images, tvg, cpn, wide)
        finally:
            if wide:
                self.engine.set_precise(False)

    def _prepare(self, input_ids, position_ids, attention_mask, past_key_values, labels, images, tvg, cpn, wide):
        import torch
        ids_h = input_ids.detach().cpu().numpy()
        msk_h = (attention_mask.detach().cpu().numpy() != 0) if attention_mask is not None else np.ones_like(ids_h, dtype=bool)
        lab_h = labels.detach().cpu().numpy() if labels is not None else np.full_like(ids_h, IGNORE_INDEX)
        B = ids_h.shape[0]
        feats, feat_off = [], []
        n_feat_rows = 0
        for b in range(B):
            f = self.project(images[b], tvg)
            feats.append(f); feat_off.append(n_feat_rows); n_feat_rows += f.shape[0]
        rows_src, rows_lab, rows_cpn = [], [], []
        for b in range(B):
            ids = ids_h[b][msk_h[b]]; lab = lab_h[b][msk_h[b]]              # :333-334 strip the left pad
            where = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0]
            if len(where) != 1:
                raise NotImplementedError(f"row {b}: expected exactly one <image> placeholder, found {len(where)}")
            w = int(where[0]); nf = feats[b].shape[0]
            src = np.concatenate([ids[:w], -(1 + feat_off[b] + np.arange(nf)), ids[w + 1:]])
            lb = np.concatenate([lab[:w], np.full(nf, IGNORE_INDEX), lab[w + 1:]])
            first = np.zeros(w, dtype=np.int64)
            if tvg:
                first[: self.tvg_prefix_length] = 1                      # :414-417
            else:
                first[:] = 1                                             # :419
            cm = np.concatenate([first, np.full(nf, 1 if tvg else 0), np.ones(len(ids) - w - 1, dtype=np.int64)])  # :431-433
            if self.tokenizer_model_max_length is not None:              # :452-457
                n = self.tokenizer_model_max_length
                src, lb, cm = src[:n], lb[:n], cm[:n]
            rows_src.append(src); rows_lab.append(lb); rows_cpn.append(cm)
        L = max(len(r) for r in rows_src)
        zero_row = n_feat_rows
        src = np.full((B, L), -(1 + zero_row), dtype=np.int32)          # right padding with zero rows, :472-485
        out_lab = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
        mask = np.zeros((B, L), dtype=np.int64); cpn_mask = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            n = len(rows_src[b])
            src[b, :n] = rows_src[b]; out_lab[b, :n] = rows_lab[b]; mask[b, :n] = 1; cpn_mask[b, :n] = rows_cpn[b]
        Hh = self.dims.hidden_size
        feat_all = torch.cat(feats + [torch.zeros((1, Hh * (2 if wide else 1)), dtype=self.dtype, device=self.device)], dim=0)
        embeds = self.engine.assemble(torch.from_numpy(src.reshape(-1)).to(self.device), feat_all).reshape(B, L, -1)
        if wide:                                                          # [hi | lo] -> one float32 tensor (a dtype repack of the literal, non-hot path)
            embeds = embeds[..., :Hh].float() + embeds[..., Hh:].float()
        mdt = attention_mask.dtype if attention_mask is not None else torch.long
        new_labels = torch.from_numpy(out_lab).to(self.device) if labels is not None else None
        if attention_mask is None:
            m_t, c_t = None, torch.from_numpy(cpn_mask).to(self.device)
        else:
            m_t = torch.from_numpy(mask).to(self.device).to(mdt); c_t = torch.from_numpy(cpn_mask).to(self.device).to(mdt)
        # the row kind travels WITH the tensor (forward() reads it): prepare(tvg) / prepare(vtg) / forward(tvg embeds) may be interleaved freely
        embeds._blim_rows = "tvg" if tvg else "vtg"
        if cpn:
            return None, position_ids, (m_t, c_t), past_key_values, embeds, new_labels
        return None, position_ids, m_t, past_key_values, embeds, new_labels

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, image_sizes=None, return_dict=None,
                modalities=["image"], dpo_forward=False, cache_position=None, want_logits: bool = True):
        """modeling_videochat_flash.py:601-629 -> modeling_qwen2_flash.py:1392-1478.
        Returns an object with .logits [B,L,V] f32 and .hidden_states [B,L,H] (final-norm last hidden state)."""
        import torch
        if inputs_embeds is None or input_ids is not None:
            raise NotImplementedError("forward(): only inputs_embeds=... is supported on the scoring path")
        if position_ids is not None or past_key_values is not None or labels is not None or use_cache or output_attentions or dpo_forward:
            raise NotImplementedError("forward(): position_ids / cache / labels / attentions are outside the scoring path")
        B, L, _ = inputs_embeds.shape
        # row kind: the tag prepare_inputs_labels_for_multimodal put on the tensor; an untagged tensor (the caller derived a new one) falls back to the
        # last prepare call -- except that on an fp16 engine only TVG rows are ever handed out as float32, so a float32 tensor there IS a TVG batch
        # (treating it as VTG rows would silently round it to fp16 and run it plain: ~1e-3 on the scores)
        kind = getattr(inputs_embeds, "_blim_rows", None)
        tvg_rows = (kind == "tvg") if kind is not None else (self._tvg_rows or (inputs_embeds.dtype == torch.float32 and self.engine.dtype == "f16" and self.engine.can_precise))
        vmode = self.vtg_mode()
        if vmode == "auto" and not tvg_rows:
            raise RuntimeError("vtg_precise = 'auto' has not been resolved on the weights now loaded: evaluation() measures it before its first pass "
                               "(PairScorer.calibrate_vtg); set BlimModel.vtg_precise to none / full to call forward() directly")
        wide = inputs_embeds.dtype == torch.float32 and self.engine.can_precise and (self.engine.dtype == "bf16" or tvg_rows)
        if wide:                                                          # float32 embeddings of prepare_inputs_labels_for_multimodal (bf16 engines; TVG rows on fp16 ones): back to [hi | lo]
            hi = inputs_embeds.to(self.dtype)
            emb = torch.cat([hi, (inputs_embeds - hi.float()).to(self.dtype)], dim=-1).contiguous()
        else:
            emb = inputs_embeds.to(self.dtype).contiguous()
        if attention_mask is None:
            m8 = torch.ones((B, L), dtype=torch.uint8, device=self.device)
        else:
            m8 = (attention_mask != 0).to(torch.uint8).contiguous()
        # a forward over rows prepared with tvg=True runs in the compensated mode, like the fused TVG calls (engine.set_precise); VTG rows
        # follow self.vtg_precise
        if tvg_rows:
            self.engine.set_precise(True, embeds=wide, mlp=self.tvg_mode() != "attn", tvg=True)       # (an unresolved "auto" runs fully compensated)
        else:
            on = vmode == "full"
            self.engine.set_precise(on, embeds=wide and on, mlp=True)
            if wide and not on:                                           # plain bf16 VTG forward asked for (vtg_precise none): plain embeddings
                emb = inputs_embeds.to(self.dtype).contiguous()
        try:
            logits, hidden = self.engine.forward(emb, m8, want_logits=want_logits, want_hidden=True)
        finally:
            self.engine.set_precise(False)
        return SimpleNamespace(loss=None, logits=logits, past_key_values=None, hidden_states=hidden, attentions=None)

    __call__ = forward


class DDPLike:
    """Stand-in for DistributedDataParallel: the eval loop calls model.module.* and model(...)."""

    def __init__(self, module: BlimModel):
        self.module = module

    def eval(self):
        self.module.eval()
        return self

    def __call__(self, *a, **k):
        return self.module(*a, **k)
