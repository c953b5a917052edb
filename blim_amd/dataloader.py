"""Dataset / tokenisation front end of the scoring path.

Produces exactly what `evaluation()` drains from the reference's loader (dataloader/base_dataset.py, dataloader/__init__.py
and the four 15-20-line dataset subclasses): per item the pre-extracted video feature `[clips, 64, 1024]`, the VTG row
(`[system][user: <image>\\n<instruction>][assistant: text]`) and the TVG row (`[system][user: instruction\\nCaption: text]
[assistant: <image>]`) as token ids with the `<image>` placeholder = -200, labels = -100 on the prompt, masks, plus the
dataset-level `video_vocab` (clip means), `vids` and `tvg_prefix_length`.

On-disk formats (unchanged from the reference):
  ./data/<DS>/features/<vid>.pth          torch tensor [4, 64, 1024] (extract.py:107-110); missing file -> zeros (base_dataset.py:27-28)
  ./data/<DS>/<annotation>.json           list of {"video": ..., "caption": ...}; per-dataset file names / caption joins below
  ./scores/<ds>[_zeroshot].pth            {"v2t": [Nv,Nt], "t2v": [Nt,Nv]} first-stage scores (read by evaluation())
"""
from __future__ import annotations

import copy
import glob
import json
import os
from typing import List

import torch

from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX

DEFAULT_IMAGE_TOKEN = "<image>"

# conv_templates["qwen_2"] (videochat_flash/conversation.py:440-449): ChatML
CHATML_SYSTEM = "<|im_start|>system\nYou are a helpful assistant."
CHATML_ROLES = ("<|im_start|>user", "<|im_start|>assistant")
CHATML_SEP = "<|im_end|>"

VTG_PROMPTS = {                                    # base_dataset.py:61-66
    "DiDeMo": "Describe this video in detail.", "ActivityNet": "Describe this video in detail.",
    "LSMDC": "Describe this video in one sentence.", "MSRVTT": "Describe this video briefly.",
}
TVG_PROMPT = "Generate a video given the caption."  # base_dataset.py:87


def _annotation_file(dataset: str, split: str) -> str:
    if dataset == "MSRVTT":
        return f"msrvtt_ret_{split}.json"                                       # msrvtt.py:7
    if dataset == "DiDeMo":
        return f"didemo_ret_{split}.json"                                       # didemo.py:8
    if dataset == "ActivityNet":
        return "anet_ret_train.json" if split == "train" else "anet_ret_val_1.json"      # activitynet.py:7-10
    if dataset == "LSMDC":
        return "lsmdc_ret_train.json" if split == "train" else "lsmdc_ret_test_1000.json"  # lsmdc.py:7-10
    raise ValueError(f"unknown dataset {dataset}")


def _vid_and_text(dataset: str, anno: dict):
    """Per-dataset video id and caption rules (msrvtt.py:10-13, didemo.py:11-14, activitynet.py:13-16, lsmdc.py:13-16)."""
    if dataset == "LSMDC":
        return anno["video"][:-4].split("/")[1], anno["caption"].strip()
    vid = anno["video"].split(".")[0]
    if dataset == "DiDeMo":
        return vid, (" ".join(anno["caption"])).strip()
    if dataset == "ActivityNet":
        return vid, ("".join(anno["caption"]).strip())
    return vid, anno["caption"].strip()


def chatml_prompt(messages) -> str:
    """Conversation.get_prompt() for SeparatorStyle.CHATML (conversation.py:90-100); message None = open assistant turn."""
    ret = CHATML_SYSTEM + CHATML_SEP + "\n"
    for role, message in messages:
        ret += role + "\n" + message + CHATML_SEP + "\n" if message else role + "\n"
    return ret


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> torch.Tensor:
    """base_dataset.py:39-58: tokenise the chunks between <image> markers and splice the placeholder id in."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids: List[int] = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == getattr(tokenizer, "bos_token_id", None):
        offset = 1
        ids.append(chunks[0][0])
    for k, c in enumerate(chunks):
        if k > 0:
            ids.append(image_token_index)          # the separator [image]*(offset+1) sliced by [offset:] is one placeholder
        ids.extend(c[offset:])
    return torch.tensor(ids, dtype=torch.long)


class RetrievalDataset(torch.utils.data.Dataset):
    """The reference's BaseDataset + {MSRVTT, DiDeMo, ActivityNet, LSMDC} in one class (eval and train splits)."""

    def __init__(self, args, tokenizer=None, image_processor=None, split: str = "test", root: str = "."):
        self.args, self.tokenizer, self.split, self.root = args, tokenizer, split, root
        self.dataset = args.dataset
        self.feature_dir = os.path.join(root, "data", self.dataset, "features")
        self.features = set(glob.glob(os.path.join(self.feature_dir, "*.pth")))                        # base_dataset.py:16
        self.tvg_prefix_length = len(tokenizer_image_token(chatml_prompt([(CHATML_ROLES[0], TVG_PROMPT)]), tokenizer)) - 2   # :20-24
        annotations = json.load(open(os.path.join(root, "data", self.dataset, _annotation_file(self.dataset, split))))
        self.data = []
        for anno in annotations:
            vid, text = _vid_and_text(self.dataset, anno)
            if split == "test" or (split == "train" and self._feature_path(vid) in self.features):
                self.data.append({"vid": vid, "text": text})
        self.num_annotations = len(annotations)
        self.vids = sorted(set(d["vid"] for d in self.data))                                           # :33-37
        self.video_vocab = torch.stack([self.load_video_feature(v).mean(1) for v in self.vids], dim=0)
        self._vid_index = {v: i for i, v in enumerate(self.vids)}

    def _feature_path(self, vid: str) -> str:
        return os.path.join(self.feature_dir, f"{vid}.pth")

    def load_video_feature(self, vid: str) -> torch.Tensor:
        p = self._feature_path(vid)
        if p not in self.features:
            return torch.zeros(4, 64, 1024)                                                           # :27-28
        return torch.load(p, weights_only=True)

    def _ids_labels(self, user_msg: str, response: str):
        prompt_ids = tokenizer_image_token(chatml_prompt([(CHATML_ROLES[0], user_msg), (CHATML_ROLES[1], None)]), self.tokenizer)
        input_ids = tokenizer_image_token(chatml_prompt([(CHATML_ROLES[0], user_msg), (CHATML_ROLES[1], response)]), self.tokenizer)
        assert (prompt_ids != input_ids[: len(prompt_ids)]).sum() == 0                                 # :78, :99
        labels = copy.deepcopy(input_ids)
        labels[: len(prompt_ids)] = IGNORE_INDEX
        masks = input_ids.ne(self.tokenizer.pad_token_id).long()
        return input_ids, labels, masks

    def get_vtg_id(self, item):                                                                        # :60-84
        return self._ids_labels(f"{DEFAULT_IMAGE_TOKEN}\n{VTG_PROMPTS[self.dataset]}", item["text"])

    def get_tvg_id(self, item):                                                                        # :86-105
        return self._ids_labels(f"{TVG_PROMPT}\nCaption: {item['text']}", DEFAULT_IMAGE_TOKEN)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):                                                                        # :107-114
        item = self.data[idx]
        vtg_ids, vtg_labels, vtg_masks = self.get_vtg_id(item)
        tvg_ids, tvg_labels, tvg_masks = self.get_tvg_id(item)
        return {"vid": item["vid"], "video": self.load_video_feature(item["vid"]), "vtg_ids": vtg_ids, "vtg_labels": vtg_labels,
                "vtg_masks": vtg_masks, "tvg_ids": tvg_ids, "tvg_labels": tvg_labels, "tvg_masks": tvg_masks,
                "tvg_video_labels": self._vid_index[item["vid"]]}

    def collate_fn(self, batch):                                                                       # :119-163
        keys = ("vtg_ids", "vtg_labels", "vtg_masks", "tvg_ids", "tvg_labels", "tvg_masks")
        out = {"vid": [b["vid"] for b in batch], "video": [b["video"] for b in batch]}
        if self.split == "train":
            fill = {"ids": self.tokenizer.pad_token_id, "labels": IGNORE_INDEX, "masks": 0}
            for k in keys:
                L = max(len(b[k]) for b in batch)
                t = torch.full((len(batch), L), fill[k.split("_")[1]], dtype=torch.long)
                for i, b in enumerate(batch):
                    t[i, L - len(b[k]):] = b[k]                                                       # left padding
                out[k] = t
        else:
            for k in keys:
                out[k] = [b[k] for b in batch]
        out["tvg_video_labels"] = torch.tensor([b["tvg_video_labels"] for b in batch])
        return out


def load_data(args, tokenizer=None, image_processor=None, split: str = "train", root: str = "."):
    """dataloader/__init__.py:8-20.  The eval loader has no sampler: every rank iterates the whole test set."""
    from . import distributed as D
    dataset = RetrievalDataset(args, tokenizer=tokenizer, image_processor=image_processor, split=split, root=root)
    if split == "train":
        sampler = torch.utils.data.DistributedSampler(dataset, num_replicas=D.get_world_size(), rank=D.get_rank(), shuffle=True)
        return torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch_size, num_workers=getattr(args, "num_workers", 0),
                                           collate_fn=dataset.collate_fn, pin_memory=getattr(args, "pin_mem", False), drop_last=False)
    return torch.utils.data.DataLoader(dataset, batch_size=args.batch_size_eval, num_workers=getattr(args, "num_workers", 0),
                                       collate_fn=dataset.collate_fn, shuffle=False, pin_memory=getattr(args, "pin_mem", False), drop_last=False)
