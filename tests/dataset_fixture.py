"""Synthetic on-disk dataset trees (annotations + feature files) and a deterministic stand-in tokenizer, shared by the
golden generator (oracle/gen_golden_dataset.py, which runs the REFERENCE's dataloader on them) and tests/test_dataloader.py."""
import json
import os
import re
import zlib
from types import SimpleNamespace

import numpy as np
import torch

SPECIALS = {"<|im_start|>": 151644, "<|im_end|>": 151645, "\n": 198}


class StubTokenizer:
    """Deterministic tokenizer: specials and newlines map to Qwen2's ids, every other whitespace-separated piece to a
    crc32-derived id in [1000, 150000).  Same interface the reference uses: tokenizer(text).input_ids, pad_token_id, bos_token_id."""
    pad_token_id = 151643
    bos_token_id = None

    def __call__(self, text):
        ids = []
        for piece in re.split(r"(<\|im_start\|>|<\|im_end\|>|\n)", text):
            if piece in SPECIALS:
                ids.append(SPECIALS[piece])
            else:
                ids += [1000 + zlib.crc32(w.encode()) % 149000 for w in piece.split()]
        return SimpleNamespace(input_ids=ids)


CAPTIONS = ["a man is cooking pasta in a kitchen", "two dogs run across a field", "a woman explains how to fold a paper plane",
            "people dance at a wedding party while a band plays", "a cat", "someone assembles a wooden chair step by step"]


def annotations(dataset):
    if dataset == "MSRVTT":
        return "msrvtt_ret_test.json", [{"video": f"video{i}.mp4", "caption": f" {c} "} for i, c in enumerate(CAPTIONS)]
    if dataset == "DiDeMo":
        return "didemo_ret_test.json", [{"video": f"v{i}.avi", "caption": [c, "then it ends"]} for i, c in enumerate(CAPTIONS)]
    if dataset == "ActivityNet":
        return "anet_ret_val_1.json", [{"video": f"v_{i}.mp4", "caption": [c + ". ", "Later they stop."]} for i, c in enumerate(CAPTIONS)]
    if dataset == "LSMDC":
        return "lsmdc_ret_test_1000.json", [{"video": f"movie{i % 2}/clip_{i}.avi", "caption": c} for i, c in enumerate(CAPTIONS)]
    raise ValueError(dataset)


def vid_of(dataset, anno):
    return anno["video"][:-4].split("/")[1] if dataset == "LSMDC" else anno["video"].split(".")[0]


def build_tree(root, dataset, missing=(2,)):
    """Writes ./data/<dataset>/{annotation json, features/*.pth}; feature `missing` indices are left out (-> zeros)."""
    d = os.path.join(root, "data", dataset, "features")
    os.makedirs(d, exist_ok=True)
    fname, annos = annotations(dataset)
    json.dump(annos, open(os.path.join(root, "data", dataset, fname), "w"))
    for i, a in enumerate(annos):
        if i in missing:
            continue
        rs = np.random.RandomState(100 + i)
        torch.save(torch.from_numpy(rs.randn(4, 64, 1024).astype(np.float16)), os.path.join(d, f"{vid_of(dataset, a)}.pth"))
    return annos
