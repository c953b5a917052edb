"""Rank / metric math of the reference's eval epoch (training_utils.py:106-221): CPN subtraction,
the two linear ensembles and R@1/5/10.  Host-side numpy on the [N, N] score matrices."""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import distributed as dist_utils
from .retrieval_utils import evaluation


def _recall_one(m: np.ndarray, ids) -> tuple:
    if np.count_nonzero(m == 0) != 0:                               # zero sentinel, training_utils.py:174-175, 195-196
        return 0.0, 0.0, 0.0
    ranks = np.zeros(m.shape[0])
    for index, score in enumerate(m):
        inds = np.argsort(score)[::-1]
        gt = ids[index]
        if isinstance(gt, (int, np.integer)):
            ranks[index] = np.where(inds == gt)[0][0]
        else:
            ranks[index] = min(np.where(inds == g)[0][0] for g in gt)
    n = len(ranks)
    return tuple(100.0 * len(np.where(ranks < t)[0]) / n for t in (1, 5, 10))


def get_recall(t2v, v2t, t2v_ids, v2t_ids) -> Dict[str, float]:
    """training_utils.py:173-221."""
    v1, v5, v10 = _recall_one(v2t, v2t_ids)
    t1, t5, t10 = _recall_one(t2v, t2v_ids)
    vm, tm = (v1 + v5 + v10) / 3, (t1 + t5 + t10) / 3
    res = {"t2v_r1": t1, "t2v_r5": t5, "t2v_r10": t10, "t2v_r_mean": tm, "v2t_r1": v1, "v2t_r5": v5, "v2t_r10": v10,
           "v2t_r_mean": vm, "r_mean": (vm + tm) / 2}
    return {k: round(v, 2) for k, v in res.items()}


def _grid_search(combine, t2v_ids, v2t_ids):
    """The reference's coefficient sweep: c in linspace(0, 1, 11); per direction the FIRST c with a strictly better R@1 wins, rounded to 0.1."""
    best_v2t = best_t2v = 0
    v2t_c = t2v_c = 0
    for c in np.linspace(0, 1, 11):
        t2v, v2t = combine(c, c)
        res = get_recall(t2v, v2t, t2v_ids, v2t_ids)
        if best_v2t < res["v2t_r1"]:
            best_v2t, v2t_c = res["v2t_r1"], round(float(c), 1)
        if best_t2v < res["t2v_r1"]:
            best_t2v, t2v_c = res["t2v_r1"], round(float(c), 1)
    t2v, v2t = combine(t2v_c, v2t_c)
    return t2v, v2t, t2v_c, v2t_c


def calculate_score(t2v_1, v2t_1, t2v_2, v2t_2, t2v_ids, v2t_ids):
    """training_utils.py:106-122: best linear ensemble c * S_1 + (1 - c) * S_2 per direction (the sweep the paper's `--c` values come from)."""
    return _grid_search(lambda ct, cv: (ct * t2v_1 + (1 - ct) * t2v_2, cv * v2t_1 + (1 - cv) * v2t_2), t2v_ids, v2t_ids)


def calculate_cpn_score(t2v, v2t, t2v_prior, v2t_prior, t2v_ids, v2t_ids):
    """training_utils.py:124-140: best CPN strength S - c * prior per direction (the sweep behind `--alpha`)."""
    return _grid_search(lambda ct, cv: (t2v - ct * t2v_prior, v2t - cv * v2t_prior), t2v_ids, v2t_ids)


def combine_and_rank(t2v_dict, v2t_dict, args, n: int) -> Dict[str, Dict[str, float]]:
    """The rank-0 part of val_one_epoch (training_utils.py:145-169): recall of the first-stage scores and of the two likelihoods, then -- with the candidate prior
    subtracted at strength alpha (CPN, :153-155) -- of the debiased candidate likelihood, and of the ensemble: c0 / c1 mix query and (debiased) candidate likelihood
    per direction, c2 / c3 mix that with the first-stage scores (:161-164).  A zero-shot run (--eval without --resume) has no t2v candidate pass and no v2t query
    pass: the reference substitutes zero matrices, which get_recall's sentinel reports as 0."""
    ids = {i: i for i in range(n)}
    finetuned = (getattr(args, "resume", "") != "") or not getattr(args, "eval", True)
    zeros = np.zeros((n, n))
    results = {name: get_recall(t2v_dict.get(name, zeros), v2t_dict.get(name, zeros), ids, ids) for name in ("internvideo2", "candidate_likelihood", "query_likelihood")}
    t2v_cand = t2v_dict["candidate_likelihood"] if finetuned else zeros
    v2t_cand = v2t_dict["candidate_likelihood"]
    if args.cpn:
        if finetuned:
            t2v_cand = t2v_cand - args.alpha[0] * t2v_dict["candidate_prior"]
        v2t_cand = v2t_cand - args.alpha[1] * v2t_dict["candidate_prior"]
        results["cpn_candidate_likelihood"] = get_recall(t2v_cand, v2t_cand, ids, ids)
    c0, c1, c2, c3 = args.c
    t2v_lm = c0 * t2v_dict["query_likelihood"] + (1 - c0) * t2v_cand
    v2t_lm = c1 * v2t_dict["query_likelihood"] + (1 - c1) * v2t_cand if finetuned else v2t_cand
    results["blim"] = get_recall(c2 * t2v_lm + (1 - c2) * t2v_dict["internvideo2"], c3 * v2t_lm + (1 - c3) * v2t_dict["internvideo2"], ids, ids)
    return results


def val_one_epoch(model, data_loader, optimizer, device, epoch, loss_scaler, tokenizer=None, args=None):
    """training_utils.py:140-169 (no autocast: the engine computes in its 16-bit compute dtype -- fp16 by default -- with f32 accumulation)."""
    t2v_dict, v2t_dict = evaluation(model, data_loader, device, tokenizer, args)
    dump = getattr(args, "dump_scores", None)
    if dump and dist_utils.is_main_process():                          # engine-side option: the score matrices themselves (mode comparisons)
        import time
        t0 = time.time()
        np.savez_compressed(dump, **{f"t2v_{k}": v for k, v in t2v_dict.items()}, **{f"v2t_{k}": v for k, v in v2t_dict.items()})
        args._dump_seconds = time.time() - t0                          # (a debugging aid's file I/O: main.py keeps it out of the evaluation time it reports)
    if getattr(args, "shard", None) is not None:                       # one rank's share played by a single process: the matrices are partial, there is no table to rank
        return None
    if dist_utils.is_main_process():
        return combine_and_rank(t2v_dict, v2t_dict, args, len(data_loader.dataset))
    return None
