"""`--vtg_precise auto` / `--tvg_precise auto`: which compensation a checkpoint's scoring calls need is MEASURED on the evaluation's own pairs.

The reference has ONE numeric mode (training_utils.py:142: autocast fp16) and no decision to make; this engine's plain 16-bit calls are the fastest of its modes and
whether they hold the 1e-3 bar depends on the checkpoint's statistics (attention sinks, massive activations).  Split out of retrieval_utils.py in round 6 (VERDICT r5
item 7): `predicted_max_deviation` (the tail extrapolation), `calibration_pairs` (which pairs are measured), and `CalibrationMixin` -- the `calibrate_vtg` /
`calibrate_tvg` methods of `pair_scorer.PairScorer`.  retrieval_utils re-exports all of it under the old names.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import distributed as dist_utils

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
    fixtures, where the sample IS the evaluation): the sample maximum itself.

    Round 6: the line goes through the top max(n / 8, min(n / 2, 128)) order statistics -- the upper half of a 256-entry sample as before, the top EIGHTH of the
    2,048-entry confirmation sample (CalibrationMixin._decide).  A log-normal law has the same slope everywhere, so nothing changes for it (simulated: predicted / true
    largest of 472,000 = 0.99 median, 0.75 at the 5 % quantile from 2,048 samples); a Gaussian-like law bends downwards in these coordinates, and a window nearer the
    tail overshoots it by 1.35 x instead of 2.2 x at n_eval = 472,000 (profiles/r06_calibrator_false_rejects.md)"""
    x = np.asarray(dev, dtype=np.float64).reshape(-1)
    x = np.sort(x[np.isfinite(x) & (x > 0)])
    n = len(x)
    if n == 0:
        return 0.0
    if n_eval is None or n_eval <= n or n < 32:
        return float(x[-1])
    from statistics import NormalDist
    inv = NormalDist().inv_cdf
    k = np.arange(n - max(n // 8, min(n // 2, 128)), n)
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


class CalibrationMixin:
    """calibrate_vtg / calibrate_tvg of PairScorer (which provides vtg(), score(), iter_tvg_jobs(), set_vtg_mode(), set_tvg_mode(), engine, m, device, split_tvg)."""

    def _gather_dev(self, dev: np.ndarray, share) -> np.ndarray:
        """Multi-rank calibration: every rank scored its own block of the sample; all ranks get all deviations (one all-gather of <= 256 floats)."""
        import torch
        W = int(share[0]) if share is not None else 0
        if share is None or not dist_utils.is_dist_avail_and_initialized() or W != torch.distributed.get_world_size():
            # one process playing rank r of W (bench.py's emulated ranks and its warm-up, also inside a real job): nobody to gather from -- it measures its own block,
            # the COST a rank's share of the calibration stands for, and takes the job's decision (`adopt`); evaluation() hands a share to such a process only together
            # with that decision (retrieval_utils.share_of)
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

    @staticmethod
    def _stats(dev, n_eval):
        dev = np.asarray(dev, dtype=np.float64)
        if len(dev) == 0:
            return {"max": 0.0, "rms": 0.0, "pred": 0.0, "n": 0}
        return {"max": float(np.max(dev)), "rms": float(np.sqrt(np.mean(dev * dev))), "pred": predicted_max_deviation(dev, n_eval), "n": int(len(dev))}

    @staticmethod
    def _passes(devs, st, n_eval, n_sampled, bar, z, tail_margin):
        """(sample inside the bar, extrapolated tail inside the bar) for one cheap mode: (a) the sample's largest relative deviation and (b) z x its RMS are inside the
        bar; (c) the largest deviation PREDICTED for the n_eval entries of the whole evaluation (predicted_max_deviation of every law in `devs`, each extrapolated on
        its own) is inside tail_margin x bar -- or inside the bar itself when the sample IS the evaluation."""
        finite = all(np.all(np.isfinite(d)) for d in devs)
        limit = tail_margin * bar if (n_eval or 0) > n_sampled else bar
        return bool(finite and st["max"] <= bar and z * st["rms"] <= bar), bool(st["pred"] <= limit)

    def _decide(self, measure, pairs, confirm_pairs, share, n_eval, bar, z, tail_margin, cheap, full, adopt=None):
        """The decision both calibrations share.  measure(block) -> list of deviation arrays (one per law: VTG one, TVG likelihood and prior) of the CHEAP mode from the
        fully compensated one on this rank's block, already gathered over the ranks.

        Stage 1 -- the 256-pair sample: the cheap mode is REJECTED when the sample itself is outside the bar ((a) or (b): what weights with massive activations do,
        e.g. sink.npz 2.3e-3) and ACCEPTED when the extrapolated tail (c) is inside it as well.
        Stage 2 (round 6) -- only when the sample is inside the bar but its extrapolation is not: a 256-entry sample pins a log-normal tail but
        overshoots a Gaussian-like one by 2.2 x at the 472,000 entries of an ActivityNet-sized evaluation, which sent BASELINE config 4 to the compensated mode (0.69 x
        the rate) on weights whose largest deviation over the whole evaluation is 4e-4.  Before paying for `full`, up to 2,048 pairs (`confirm_pairs`, the stage-1 pairs
        included: 1.7 s of a one-GPU evaluation, 1 / W of it per rank) are measured and the law is read nearer the tail (predicted_max_deviation: top eighth); the cheap
        mode is kept iff (a), (b) and (c) hold on the larger sample.  A heavy tail shows itself there directly: at sigma_log = 0.9 (heavy7b) the largest of 2,048
        entries is already ~ 2e-3.  Every rank sees the same gathered deviations, so every rank takes the same branch (the stage-2 gathers are collective).
        adopt = (mode, confirmed): one process PLAYING a rank of a W-process job (`--shard`, bench.py's emulated ranks) has nobody to gather from and would decide on
        1 / W of the sample; handed the job-wide outcome, it measures its block of the same stages the job ran -- the cost a rank's share stands for -- and takes the
        job's decision."""
        n_all = len(pairs)
        block, sh = self._my_block(pairs, share)
        devs = measure(block, sh)
        dev = np.concatenate(devs) if devs else np.zeros(0)
        st = self._stats(dev, n_eval)
        st["pred"] = max([predicted_max_deviation(d, n_eval) for d in devs] or [0.0])
        ok_sample, ok_tail = self._passes(devs, st, n_eval, n_all, bar, z, tail_margin)
        entry = dict(st)
        chosen = cheap if (ok_sample and ok_tail) else full
        go_on = (ok_sample and not ok_tail) if adopt is None else bool(adopt[1])
        if go_on and confirm_pairs is not None and len(confirm_pairs):
            seen = {(int(a), int(b)) for a, b in np.asarray(pairs)}
            extra = np.array([p for p in np.asarray(confirm_pairs, dtype=np.int64) if (int(p[0]), int(p[1])) not in seen], dtype=np.int64).reshape(-1, 2)
            if len(extra):
                block2, sh2 = self._my_block(extra, share)
                devs2 = measure(block2, sh2)
                devs = [np.concatenate([a, b]) for a, b in zip(devs, devs2)]
                dev = np.concatenate(devs)
                st2 = self._stats(dev, n_eval)
                st2["pred"] = max(predicted_max_deviation(d, n_eval) for d in devs)
                ok2, tail2 = self._passes(devs, st2, n_eval, n_all + len(extra), bar, z, tail_margin)
                entry["confirm"] = dict(st2, accepted=bool(ok2 and tail2))
                chosen = cheap if (ok2 and tail2) else full
        if adopt is not None:
            entry["adopted"] = str(adopt[0]); chosen = adopt[0]
        return chosen, entry

    def calibrate_vtg(self, pairs, bar: float = 1e-3, z: float = 4.5, n_eval: Optional[int] = None, tail_margin: float = 0.8, share=None, confirm_pairs=None, adopt=None):
        """The decision is MEASURED on the loaded weights: `pairs` (up to 256 (video, text) pairs of the evaluation itself) are scored plain and fully compensated -- the
        compensated mode sits at 2e-6 .. 1e-4 of the fp32 reference on every fixture, i.e. is a yardstick that needs no oracle on the box -- and plain is kept when it
        passes `_decide` (sample inside the bar; tail extrapolated to the evaluation's n_eval entries inside tail_margin x bar: from 256 samples the extrapolation lands at
        0.74 - 1.8 x the true largest entry, tools/vtg_modes_population.py; a larger confirmation sample `confirm_pairs` before a merely EXTRAPOLATED miss costs the
        compensated rate).  The bar is per ENTRY of the whole evaluation while the calibration sees a sample, and a sample's maximum is a noisy statistic (on sink.npz the
        same mode reads 7e-4 or 1.2e-3 depending on last-bit differences upstream): hence (b) z x rms, for near-Gaussian deviations the largest of 10^4 .. 10^5 entries
        is 4 - 4.8 sigma.  share = (world, rank): each rank scores its block of the sample, the deviations are all-gathered.
        Returns (mode name, {mode: {max, rms, pred, n[, confirm: {...}]}} for the modes tried)."""
        pairs = np.asarray(pairs, dtype=np.int64)
        resolve = getattr(self.m, "resolve_vtg", lambda mode: None)
        if not bool(getattr(self.engine, "can_precise", False)):              # fp8 engines have no compensated modes: plain it is (and resolved: ADVICE r4)
            self.set_vtg_mode(None)
            resolve("none")
            return "none", {}

        def measure(block, sh):
            if not len(block):
                return [self._gather_dev(np.zeros(0), sh)]
            self.set_vtg_mode("full")
            ref = self.vtg(block).astype(np.float64)
            self.set_vtg_mode("none")
            dev = np.abs(self.vtg(block).astype(np.float64) - ref) / np.abs(ref)
            return [self._gather_dev(dev, sh)]

        chosen, entry = self._decide(measure, pairs, confirm_pairs, share, n_eval, bar, z, tail_margin, "none", "full", adopt=adopt)
        self.set_vtg_mode(chosen)
        resolve(chosen)
        return chosen, {"none": entry}

    def calibrate_tvg(self, pairs, bar: float = 1e-3, z: float = 4.5, n_eval: Optional[int] = None, tail_margin: float = 0.8, share=None, confirm_pairs=None, adopt=None):
        """The TVG calls' counterpart of calibrate_vtg (same criterion, same yardstick = the fully compensated mode).  Every TVG call of a 16-bit engine carries its
        embeddings, QKV, attention, o_proj and head as hi + lo; what is decided here is the MLP branch (87 % of the flops): `attn` leaves it plain (1.6x faster than
        `full`).  Gaussian-like weights need no more than `attn` since the TVG head is exact (round 4); weights with massive residual channels need `full`
        (tests/golden/heavy7b.npz: the prior moved by 2.5e-3 with a plain SwiGLU output) -- measured per checkpoint on the likelihood AND the prior (the prior's queries see
        one prefix token and their own segment: the most sensitive pass), two laws, each extrapolated on its own."""
        pairs = np.asarray(pairs, dtype=np.int64)
        resolve = getattr(self.m, "resolve_tvg", lambda mode: None)
        self.set_tvg_mode("full")
        if not self.split_tvg:                                             # fp8 / fp32-less engines: nothing to choose
            resolve("full")
            return "full", {}

        def measure(block, sh):
            if not len(block):
                return [self._gather_dev(np.zeros(0), sh), self._gather_dev(np.zeros(0), sh)]
            both = lambda: self.score(self.iter_tvg_jobs([(block, False), (block, True)]), 2 * len(block)).astype(np.float64)      # likelihood and prior in the same engine calls
            self.set_tvg_mode("full")
            ref = both()
            self.set_tvg_mode("attn")
            d_ = np.abs(both() - ref) / np.abs(ref)
            h_ = len(d_) // 2
            return [self._gather_dev(d_[:h_], sh), self._gather_dev(d_[h_:], sh)]

        chosen, entry = self._decide(measure, pairs, confirm_pairs, share, n_eval, bar, z, tail_margin, "attn", "full", adopt=adopt)
        self.set_tvg_mode(chosen)
        resolve(chosen)
        return chosen, {"attn": entry}

    def calibrate_second_pass(self, pairs, bar: float = 1e-3, z: float = 4.5, n_eval: Optional[int] = None, tail_margin: float = 0.8, share=None, confirm_pairs=None, adopt=None):
        """bf16 engines, `--second_pass auto` (round 6): may the VTG calls' second walk over K run on the e2m3 MFMA (0.705x the plain fp16 rate) instead of in bf16 (0.50x)?
        Same criterion and stages as calibrate_vtg; the yardstick is the fully compensated mode with the bf16 second pass (1 - 3e-6 from the fp32 reference at 7B depth), the
        cheap mode the same calls with engine option "precise_lo6" = 1 (3 - 7e-5 on N(0, 0.02^2) weights; on weights with massive activations it inherits about one fp16
        rounding's noise and is rejected like plain fp16 is there).  The TVG calls of a bf16 engine keep the bf16 pass either way (Engine.set_precise).
        Returns ("e2m3" | "16bit", table)."""
        pairs = np.asarray(pairs, dtype=np.int64)
        resolve = getattr(self.m, "resolve_second_pass", lambda mode: None)
        if getattr(self.engine, "dtype", "") != "bf16":
            return ("e2m3" if bool(getattr(self.engine, "lo6", False)) else "16bit"), {}
        self.set_vtg_mode("full")

        def measure(block, sh):
            if not len(block):
                return [self._gather_dev(np.zeros(0), sh)]
            self.engine.set_option("precise_lo6", 0)
            ref = self.vtg(block).astype(np.float64)
            self.engine.set_option("precise_lo6", 1)
            dev = np.abs(self.vtg(block).astype(np.float64) - ref) / np.abs(ref)
            return [self._gather_dev(dev, sh)]

        try:
            chosen, entry = self._decide(measure, pairs, confirm_pairs, share, n_eval, bar, z, tail_margin, "e2m3", "16bit", adopt=adopt)
        finally:
            self.engine.set_option("precise_lo6", 0)
        resolve(chosen)
        if not hasattr(self.m, "resolve_second_pass"):
            self.engine.set_option("precise_lo6", 1 if chosen == "e2m3" else 0)
        return chosen, {"e2m3": entry}
