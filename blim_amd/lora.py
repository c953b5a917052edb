"""Trainable-parameter inventory of the reference's fine-tuning set-up (main.py:96-111), shared by the trainer, the checkpoint
writer and the tests.

The reference trains, with everything else frozen:
  * LoRA adapters (peft; r = --lora_r 8, alpha = --lora_alpha 32, dropout = --lora_drop 0.05) on the projector `mlp` Linear "0" and "2"
    (main.py:96-97), on `tvg_mlp = deepcopy(mlp)` (main.py:98), and on every q/k/v/o_proj and lm_head (main.py:100-101):
        y = W x + b + (alpha / r) * B (A dropout(x)),      A [r, in] (kaiming-uniform init), B [out, r] (zero init);
  * `visual_head.weight` [mm_hidden, H] in fp32 (main.py:104-107).

Canonical names: "<weight name>:A" / "<weight name>:B" with <weight name> from blim_amd/checkpoint.py:expected_adapters, and
"visual_head".  The order of `trainable_names` is the order of the flat parameter / gradient / moment buffers the engine's trainer
uses (include/blim.h: blim_train_*), so it is part of the C ABI's contract.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Tuple

import numpy as np

from .checkpoint import canonical_to_hf, expected_adapters
from .synth import ModelDims, weight_shapes


def trainable_shapes(dims: ModelDims, r: int) -> Dict[str, Tuple[int, ...]]:
    shapes = weight_shapes(dims)
    out: Dict[str, Tuple[int, ...]] = {}
    for w in expected_adapters(dims):
        n_out, n_in = shapes[w]
        out[w + ":A"] = (r, n_in)
        out[w + ":B"] = (n_out, r)
    out["visual_head"] = tuple(shapes["visual_head"])
    return out


def trainable_names(dims: ModelDims) -> List[str]:
    return list(trainable_shapes(dims, 1).keys())


def flat_layout(dims: ModelDims, r: int) -> Tuple[Dict[str, Tuple[int, Tuple[int, ...]]], int]:
    """name -> (offset in elements, shape) inside the flat f32 buffers; every tensor starts on a 64-element boundary."""
    off, lay = 0, {}
    for n, s in trainable_shapes(dims, r).items():
        lay[n] = (off, s)
        off += (int(np.prod(s)) + 63) // 64 * 64
    return lay, off


def resume_key(name: str) -> str:
    """Canonical trainable name -> key of the reference's checkpoint file (util/misc.py:282-285 saves named_parameters() with
    requires_grad of the peft-wrapped model; layout documented in blim_amd/checkpoint.py)."""
    if name == "visual_head":
        return "base_model.model.visual_head.weight"
    w, kind = name.split(":")
    hf = canonical_to_hf(w)                               # ...<module>.weight
    mod = hf[: -len(".weight")]
    m = re.fullmatch(r"model\.mm_projector\.(mlp|tvg_mlp)\.(\d)", mod)
    if m:
        mod = f"model.mm_projector.{m.group(1)}.base_model.model.{m.group(2)}"
    return f"base_model.model.{mod}.lora_{kind}.default.weight"


def init_trainable(dims: ModelDims, r: int, seed: int, visual_head: np.ndarray = None) -> Dict[str, np.ndarray]:
    """peft's LoRA init: A ~ kaiming_uniform(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)), B = 0; tvg_mlp's adapters are copies of mlp's
    (deepcopy, main.py:98).  visual_head keeps the checkpoint's values; when the checkpoint has none (the public VideoChat-Flash checkpoint
    predates the head, modeling_videochat_flash.py:584) it is drawn as from_pretrained() initialises a missing nn.Linear: N(0, initializer_range = 0.02)."""
    rng = np.random.default_rng(seed)
    out = {}
    for n, s in trainable_shapes(dims, r).items():
        if n.endswith(":A"):
            b = 1.0 / math.sqrt(s[1])
            out[n] = rng.uniform(-b, b, size=s).astype(np.float32)
        elif n.endswith(":B"):
            out[n] = np.zeros(s, np.float32)
        else:
            out[n] = (rng.standard_normal(s) * 0.02).astype(np.float32) if visual_head is None else np.asarray(visual_head, np.float32).copy()
    for i in (0, 2):
        for k in ("A", "B"):
            out[f"tvg_mlp.{i}.w:{k}"] = out[f"mlp.{i}.w:{k}"].copy()
    return out


def synthetic_trainable(dims: ModelDims, r: int, seed: int, rel: float = 5e-2, alpha: float = 32.0, w_std: float = 0.02) -> Dict[str, np.ndarray]:
    """Seeded NON-ZERO adapters shaped like the outcome of a fine-tuning run (tests, golden fixtures, dry runs): A keeps peft's init scale
    (std 1 / sqrt(3 in) = the std of U(-1/sqrt(in), 1/sqrt(in))), B is scaled so that || (alpha / r) B A ||_F = rel * || W ||_F for a base
    weight of std w_std -- rel = 5e-2 is what this repo's own fine-tuning runs produce.  Values are plain float32
    (not 16-bit representable), as trained adapters are.  tvg_mlp's adapters are their own tensors (trained apart after main.py:98's deepcopy);
    visual_head is a full fp32 tensor (main.py:104-107)."""
    from .synth import bell_f32
    s = alpha / r
    out = {}
    shapes = trainable_shapes(dims, r)
    for n, shape in shapes.items():
        if n == "visual_head":
            out[n] = bell_f32(seed, "adapter/" + n, int(np.prod(shape)), w_std).reshape(shape)
            continue
        n_in = shapes[n[:-1] + "A"][1]
        a_std = 1.0 / math.sqrt(3.0 * n_in)
        std = a_std if n.endswith(":A") else rel * w_std / (s * a_std * math.sqrt(r))
        out[n] = bell_f32(seed, "adapter/" + n, int(np.prod(shape)), std).reshape(shape)
    return out


def resume_state(trainable: Dict[str, np.ndarray]) -> Dict[str, object]:
    """The reference's resume-file layout (util/misc.py:276-297: {'model': {peft-named trainable tensors}, ...}) for a set of trainables."""
    import torch
    return {"model": {resume_key(n): torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for n, v in trainable.items()}}
