"""Checkpoint loading for the scoring engine.

What the reference does at start-up (main.py:96-111, 125-128; util/misc.py:276-311):
  1. `VideoChatFlashQwenForCausalLM.from_pretrained(model_path).half()` -- HF safetensors, keys `model.embed_tokens.weight`,
     `model.layers.N.self_attn.q_proj.{weight,bias}`, ..., `model.mm_projector.mlp.{0,2}.{weight,bias}`, `lm_head.weight`;
  2. LoRA (r = --lora_r 8, alpha = --lora_alpha 32) on the projector `mlp` Linear "0" and "2"; `tvg_mlp = deepcopy(mlp)`;
     LoRA on every q/k/v/o_proj and lm_head; `visual_head` trainable in fp32;
  3. `--resume`: a torch file whose `['model']` holds ONLY the trainable tensors (LoRA A/B, visual_head), loaded strict=False.
The engine keeps the adapters APART by default, as the reference does (`blim_load_adapter`: y = W x + (alpha / r) B (A x), the rank-r term in
the base product's accumulation); lora_mode = "merge" instead folds W' = W + (alpha / r) * B @ A on the host and hands the merged matrix to
`blim_load_weight` (which rounds it to the engine's 16-bit format and lays it out for the kernels).

Key naming of the resume file follows peft's convention (`base_model.model.<path>.lora_{A,B}.default.weight`, wrapped
Linear at `<path>.base_layer`); the inner projector adapters sit under `mm_projector.{mlp,tvg_mlp}.base_model.model.{0,2}`.
peft is not installed in the build image, so this naming is implemented from peft's documented layout and exercised with
synthetic adapters (tests/test_checkpoint.py), not against a file written by the reference.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np

from .synth import ModelDims, weight_shapes


# ----------------------------------------------------------------------------- key mapping

def canonical_to_hf(name: str) -> str:
    """Canonical tensor name (blim_amd/synth.py:weight_shapes) -> key in the HF checkpoint / reference state_dict."""
    if name == "embed_tokens":
        return "model.embed_tokens.weight"
    if name == "final_norm":
        return "model.norm.weight"
    if name == "lm_head":
        return "lm_head.weight"
    if name == "visual_head":
        return "visual_head.weight"
    m = re.fullmatch(r"(mlp|tvg_mlp)\.(\d)\.(w|b)", name)
    if m:
        return f"model.mm_projector.{m.group(1)}.{m.group(2)}.{'weight' if m.group(3) == 'w' else 'bias'}"
    m = re.fullmatch(r"layers\.(\d+)\.(input_norm|post_norm)", name)
    if m:
        return f"model.layers.{m.group(1)}.{'input_layernorm' if m.group(2) == 'input_norm' else 'post_attention_layernorm'}.weight"
    m = re.fullmatch(r"layers\.(\d+)\.(q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj)\.(w|b)", name)
    if m:
        grp = "self_attn" if m.group(2) in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
        return f"model.layers.{m.group(1)}.{grp}.{m.group(2)}.{'weight' if m.group(3) == 'w' else 'bias'}"
    raise KeyError(name)


def hf_to_canonical(key: str) -> Optional[str]:
    """Inverse of canonical_to_hf; None for tensors the scoring path does not use (vision tower, rotary buffers ...)."""
    key = key.replace(".base_layer.", ".")
    table = {"model.embed_tokens.weight": "embed_tokens", "model.norm.weight": "final_norm", "lm_head.weight": "lm_head",
             "visual_head.weight": "visual_head"}
    if key in table:
        return table[key]
    m = re.fullmatch(r"model\.mm_projector\.(mlp|tvg_mlp)\.(?:base_model\.model\.)?(\d)\.(weight|bias)", key)
    if m:
        return f"{m.group(1)}.{m.group(2)}.{'w' if m.group(3) == 'weight' else 'b'}"
    m = re.fullmatch(r"model\.layers\.(\d+)\.(input_layernorm|post_attention_layernorm)\.weight", key)
    if m:
        return f"layers.{m.group(1)}.{'input_norm' if m.group(2) == 'input_layernorm' else 'post_norm'}"
    m = re.fullmatch(r"model\.layers\.(\d+)\.(?:self_attn|mlp)\.(q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj)\.(weight|bias)", key)
    if m:
        return f"layers.{m.group(1)}.{m.group(2)}.{'w' if m.group(3) == 'weight' else 'b'}"
    return None


_PEFT_PREFIX = "base_model.model."


def parse_resume_key(key: str) -> Optional[Tuple[str, str]]:
    """Key of the reference's resume file -> (canonical weight name, kind) with kind in {'A', 'B', 'full'}."""
    k = key
    while k.startswith(_PEFT_PREFIX):
        k = k[len(_PEFT_PREFIX):]
    m = re.fullmatch(r"(.*)\.lora_(A|B)\.[^.]+\.weight", k)
    if m:
        base = m.group(1).replace(".base_model.model.", ".")       # inner projector adapter
        name = hf_to_canonical(base + ".weight")
        return (name, m.group(2)) if name else None
    name = hf_to_canonical(k.replace(".base_model.model.", "."))
    return (name, "full") if name else None


# ----------------------------------------------------------------------------- tensor sources

def _to_f32(t) -> np.ndarray:
    import torch
    if isinstance(t, np.ndarray):
        return np.ascontiguousarray(t, dtype=np.float32)
    return t.detach().to(torch.float32).cpu().numpy()


def open_base_checkpoint(path: str) -> Tuple[Iterable[str], Callable[[str], np.ndarray]]:
    """(keys, get(key) -> float32 ndarray) over a HF checkpoint directory (single or sharded *.safetensors, optionally with
    model.safetensors.index.json) or a torch state-dict file."""
    if os.path.isdir(path):
        from safetensors import safe_open
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {path}")
        where: Dict[str, str] = {}
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(idx):
            where = {k: os.path.join(path, v) for k, v in json.load(open(idx))["weight_map"].items()}
        else:
            for f in files:
                with safe_open(f, framework="pt") as h:
                    for k in h.keys():
                        where[k] = f

        def get(key: str) -> np.ndarray:
            with safe_open(where[key], framework="pt") as h:
                return _to_f32(h.get_tensor(key))
        return list(where.keys()), get
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("model", sd) if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict) else sd
    return list(sd.keys()), lambda k: _to_f32(sd[k])


def lora_delta(A: np.ndarray, B: np.ndarray, r: int, alpha: float) -> np.ndarray:
    """(alpha / r) * B @ A with A [r, in], B [out, r] (peft LoRA Linear.get_delta_weight)."""
    assert A.shape[0] == r and B.shape[1] == r, (A.shape, B.shape, r)
    return (np.float32(alpha / r) * (B.astype(np.float32) @ A.astype(np.float32))).astype(np.float32)


# ----------------------------------------------------------------------------- loader

def expected_adapters(dims: ModelDims):
    """Tensors the reference fine-tunes with LoRA (main.py:98-111): both projector MLPs' Linear 0 / 2, every q/k/v/o_proj, lm_head."""
    names = [f"{p}.{i}.w" for p in ("mlp", "tvg_mlp") for i in (0, 2)] + ["lm_head"]
    names += [f"layers.{l}.{q}.w" for l in range(dims.num_layers) for q in ("q_proj", "k_proj", "v_proj", "o_proj")]
    return names


def default_lora_mode(engine) -> str:
    """'apart' (the reference's own arithmetic: y = W x + s B (A x), blim.h: blim_load_adapter) wherever the engine offers it; 'merge' is the
    explicit opt-in that folds W + s B A into the engine's 16-bit weight on the host (no per-call cost; rounds the sum -- fine in fp16 for a bf16
    base checkpoint, 8 % of the update in bf16, most of it in e4m3: DESIGN.md section 8, f-2)."""
    return "apart" if hasattr(engine, "load_adapter") else "merge"


def read_resume(resume_path: str, dims: ModelDims, lora_r: int = 8, strict_resume: bool = True):
    """The reference's resume file (util/misc.py:276-297: {'model': {peft-named trainable tensors}}) -> (adapters {weight name: {'A', 'B'}},
    full tensors {name: array}).  Checked the way the reference checks it (main.py:127 asserts that the number of checkpoint parameters equals
    the number of trainable parameters): every key must map onto an engine tensor, every expected adapter (`expected_adapters`) and
    `visual_head` must be present, and the parameter total must equal the trainable total.  A naming drift therefore raises instead of
    silently evaluating the base model; strict_resume=False downgrades the checks to warnings (partial adapter files)."""
    import torch
    shapes = weight_shapes(dims)
    adapters: Dict[str, Dict[str, np.ndarray]] = {}
    full: Dict[str, np.ndarray] = {}
    ck = torch.load(resume_path, map_location="cpu", weights_only=False)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    unparsed, n_params = [], 0
    for k, v in sd.items():
        parsed = parse_resume_key(k)
        if parsed is None or parsed[0] not in shapes:
            unparsed.append(k)
            continue
        name, kind = parsed
        n_params += int(np.prod(v.shape))
        if kind == "full":
            full[name] = _to_f32(v)
        else:
            adapters.setdefault(name, {})[kind] = _to_f32(v)
    problems = []
    if unparsed:
        problems.append(f"{len(unparsed)} key(s) map onto no engine tensor, e.g. {unparsed[:3]}")
    missing = [n for n in expected_adapters(dims) if n not in adapters]
    if missing:
        problems.append(f"{len(missing)} expected LoRA adapter(s) absent, e.g. {missing[:3]}")
    if "visual_head" not in full:
        problems.append("visual_head absent")
    want = sum(lora_r * (shapes[n][0] + shapes[n][1]) for n in expected_adapters(dims)) + int(np.prod(shapes["visual_head"]))
    if not missing and not unparsed and "visual_head" in full and n_params != want:
        problems.append(f"{n_params} parameters in the file, {want} trainable parameters expected (main.py:127)")
    if problems:
        msg = f"resume file {resume_path}: " + "; ".join(problems)
        if strict_resume:
            raise ValueError(msg + " -- refusing to evaluate a partially adapted model (strict_resume=False to override)")
        import warnings
        warnings.warn(msg)
    for name, ad in adapters.items():
        if "A" not in ad or "B" not in ad:
            raise KeyError(f"incomplete LoRA adapter for {name}")
    stray = sorted(n for n in adapters if n not in expected_adapters(dims))
    if stray:
        raise KeyError(f"LoRA adapters for tensors the engine does not adapt: {stray[:3]}")
    return adapters, full


def apply_resume(engine, dims: ModelDims, resume_path: str, lora_r: int = 8, lora_alpha: float = 32.0, strict_resume: bool = True) -> Dict[str, str]:
    """Adapters + visual_head of a resume file onto an engine whose BASE weights are already loaded, adapters kept apart (blim_load_adapter)."""
    adapters, full = read_resume(resume_path, dims, lora_r, strict_resume)
    report = {}
    for name, w in full.items():
        engine.load_weight(name, w)
        report[name] = "resume"
    for name, ad in adapters.items():
        engine.load_adapter(name, ad["A"], ad["B"], lora_r, lora_alpha)
        report[name] = "base + LoRA (apart)"
    return report


def load_checkpoint(engine, dims: ModelDims, base_path: str, resume_path: Optional[str] = None, lora_r: int = 8, lora_alpha: float = 32.0,
                    allow_missing_visual_head: bool = True, verbose: bool = False, strict_resume: bool = True, lora_mode: Optional[str] = None) -> Dict[str, str]:
    """Streams the base checkpoint (+ the LoRA adapters / visual_head of `resume_path`) into `engine`, one tensor at a time.
    Returns {canonical name: provenance} for every tensor loaded.  lora_mode: 'apart' (default: adapters stay separate matrices, as in the
    reference) or 'merge' (W + (alpha / r) B A in fp32 on the host, rounded once by blim_load_weight); see `default_lora_mode`.
    The resume file is checked as `read_resume` describes."""
    shapes = weight_shapes(dims)
    keys, get = open_base_checkpoint(base_path)
    have = {}
    for k in keys:
        n = hf_to_canonical(k)
        if n is not None and n in shapes:
            have[n] = k
    mode = lora_mode or default_lora_mode(engine)
    if mode not in ("apart", "merge"):
        raise ValueError(f"lora_mode {mode!r}: 'apart' or 'merge'")
    adapters: Dict[str, Dict[str, np.ndarray]] = {}
    full: Dict[str, np.ndarray] = {}
    if resume_path:
        adapters, full = read_resume(resume_path, dims, lora_r, strict_resume)
    if hasattr(engine, "clear_adapters"):
        engine.clear_adapters()
    report: Dict[str, str] = {}
    for name, shape in shapes.items():
        src = name
        prov = "base"
        if name not in have and name.startswith("tvg_mlp."):
            src = "mlp." + name[len("tvg_mlp."):]                     # tvg_mlp = deepcopy(mlp), main.py:102
            prov = "base (copy of mlp)"
        if name in full:
            w = full[name]; prov = "resume"
        elif src in have:
            w = get(have[src])
        elif name == "visual_head" and allow_missing_visual_head:
            # the reference initialises visual_head randomly and only uses it after fine-tuning; zero-shot eval never reads it
            w = np.zeros(shape, dtype=np.float32); prov = "absent (zeros; TVG passes need a fine-tuned checkpoint)"
        else:
            raise KeyError(f"tensor '{name}' ({canonical_to_hf(name)}) not found in {base_path}")
        assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
        ad = adapters.get(name)
        if ad is not None and mode == "merge":
            w = w + lora_delta(ad["A"], ad["B"], lora_r, lora_alpha)
            prov += " + LoRA"
        engine.load_weight(name, w)
        if ad is not None and mode == "apart":
            engine.load_adapter(name, ad["A"], ad["B"], lora_r, lora_alpha)
            prov += " + LoRA (apart)"
        report[name] = prov
        if verbose:
            print(f"{name:32s} {str(tuple(shape)):20s} {prov}")
    return report


def summarize_report(report: Dict[str, str]) -> str:
    """One line for the driver's log: how many tensors came from where."""
    from collections import Counter
    c = Counter("base + LoRA" if "+ LoRA" in p else p.split(" (")[0] for p in report.values())
    return ", ".join(f"{v} {k}" for k, v in sorted(c.items()))


def save_hf_checkpoint(weights: Dict[str, np.ndarray], out_dir: str, shards: int = 2, dtype: str = "bf16") -> None:
    """Writes canonical-name weights as a sharded HF-style safetensors checkpoint (tests / synthetic checkpoints)."""
    import torch
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
    names = [n for n in weights if not n.startswith("tvg_mlp.")]     # the base checkpoint has no tvg_mlp
    weight_map = {}
    for s in range(shards):
        part = {canonical_to_hf(n): torch.from_numpy(np.ascontiguousarray(weights[n])).to(td).contiguous() for n in names[s::shards]}
        fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
        save_file(part, os.path.join(out_dir, fn))
        weight_map.update({k: fn for k in part})
    json.dump({"metadata": {}, "weight_map": weight_map}, open(os.path.join(out_dir, "model.safetensors.index.json"), "w"))
