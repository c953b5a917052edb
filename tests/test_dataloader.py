"""CPU: the dataset / tokenisation front end (blim_amd/dataloader.py) against golden vectors recorded from the REFERENCE's
dataloader (oracle/gen_golden_dataset.py) on the same synthetic on-disk trees and the same stand-in tokenizer."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import dataset_fixture as F  # noqa: E402

from blim_amd import dataloader as DL  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataset.npz")


@pytest.mark.parametrize("ds", ["MSRVTT", "DiDeMo", "ActivityNet", "LSMDC"])
def test_eval_loader_matches_reference(tmp_path, ds):
    g = np.load(GOLD)
    F.build_tree(str(tmp_path), ds)
    args = types.SimpleNamespace(dataset=ds, batch_size_eval=4, num_workers=0, pin_mem=False)
    loader = DL.load_data(args, tokenizer=F.StubTokenizer(), split="test", root=str(tmp_path))
    dset = loader.dataset
    assert len(dset) == int(g[f"{ds}_n"])
    assert dset.tvg_prefix_length == int(g[f"{ds}_tvg_prefix_length"])
    assert list(dset.vids) == list(g[f"{ds}_vids"])
    np.testing.assert_allclose(dset.video_vocab.float().numpy()[:, :, ::64], g[f"{ds}_video_vocab_sub"], rtol=1e-6, atol=1e-6)
    nb = 0
    for bi, batch in enumerate(loader):
        nb += 1
        assert list(batch["vid"]) == list(g[f"{ds}_b{bi}_vid"])
        assert np.array_equal(batch["tvg_video_labels"].numpy(), g[f"{ds}_b{bi}_tvg_video_labels"])
        np.testing.assert_allclose([float(v.float().sum()) for v in batch["video"]], g[f"{ds}_b{bi}_video_sum"], rtol=1e-5, atol=1e-3)
        for k in ("vtg_ids", "vtg_labels", "vtg_masks", "tvg_ids", "tvg_labels", "tvg_masks"):
            assert isinstance(batch[k], list)                                   # eval collate returns lists (base_dataset.py:148-155)
            for j, t in enumerate(batch[k]):
                assert np.array_equal(t.numpy(), g[f"{ds}_b{bi}_{k}_{j}"]), (ds, bi, k, j)
    assert nb == int(g[f"{ds}_n_batches"])


def test_row_structure_and_missing_feature(tmp_path):
    F.build_tree(str(tmp_path), "MSRVTT", missing=(2,))
    args = types.SimpleNamespace(dataset="MSRVTT", batch_size_eval=3)
    d = DL.RetrievalDataset(args, tokenizer=F.StubTokenizer(), split="test", root=str(tmp_path))
    it = d[2]
    assert torch.count_nonzero(it["video"]) == 0 and it["video"].shape == (4, 64, 1024)      # missing feature file -> zeros
    assert (it["vtg_ids"] == -200).sum() == 1 and (it["tvg_ids"] == -200).sum() == 1
    # VTG: labels = -100 on the prompt, response = text + <|im_end|> + \n ; TVG: response = <image>, <|im_end|>, \n
    resp = it["vtg_labels"][it["vtg_labels"] != -100]
    assert resp[-2:].tolist() == [151645, 198]
    assert it["tvg_labels"][-3:].tolist() == [-200, 151645, 198] and (it["tvg_labels"][:-3] == -100).all()
    # system block (8 ids) + <|im_end|> \n + user header (3) + the 6-word instruction = 19 with the stand-in tokenizer
    # (21 with Qwen2's: modeling_videochat_flash.py:408); the trailing <|im_end|> \n are the '- 2' of base_dataset.py:23
    assert d.tvg_prefix_length == 8 + 2 + 3 + len(DL.TVG_PROMPT.split()) == int(np.load(GOLD)["MSRVTT_tvg_prefix_length"])


def test_train_collate_left_pads(tmp_path):
    F.build_tree(str(tmp_path), "DiDeMo", missing=())
    os.rename(os.path.join(tmp_path, "data", "DiDeMo", "didemo_ret_test.json"), os.path.join(tmp_path, "data", "DiDeMo", "didemo_ret_train.json"))
    args = types.SimpleNamespace(dataset="DiDeMo", batch_size_eval=3)
    d = DL.RetrievalDataset(args, tokenizer=F.StubTokenizer(), split="train", root=str(tmp_path))
    b = d.collate_fn([d[0], d[4]])
    assert b["vtg_ids"].shape == b["vtg_labels"].shape == b["vtg_masks"].shape and b["vtg_ids"].dim() == 2
    short = 1 if len(d[4]["vtg_ids"]) < len(d[0]["vtg_ids"]) else 0
    pad = b["vtg_ids"].shape[1] - len(d[[0, 4][short]]["vtg_ids"])
    assert pad > 0 and (b["vtg_ids"][short, :pad] == F.StubTokenizer.pad_token_id).all() and (b["vtg_masks"][short, :pad] == 0).all()
    assert (b["vtg_labels"][short, :pad] == -100).all()
