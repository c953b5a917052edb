"""TEST INFRASTRUCTURE ONLY -- imports the reference's own modules from /root/reference, in place.

Used by oracle/gen_golden.py (and by tests marked `needs_reference`, which skip when
/root/reference is absent, i.e. on the GPU box).  Nothing from the reference is copied into this
repository: this file only sets up sys.path / stub modules so the reference's files import under
the transformers version of this image (procedure of SURVEY.md Appendix A), and maps this repo's
canonical weight names onto the reference's state_dict keys.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("BLIM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "videochat_flash"))


_loaded = {}


def load():
    """Returns a namespace with the reference modules (cached)."""
    if _loaded:
        return _loaded["ns"]
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before stub modules exist)
    from transformers import AutoConfig  # noqa: F401

    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    def stub(name, **kw):
        if name in sys.modules:
            return
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m

    stub("timm")
    stub("timm.layers", drop_path=lambda x, p, t: x,
         to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x), trunc_normal_=lambda t, std=.02: t)
    for n in ("av", "cv2", "imageio"):
        stub(n)
    stub("decord", VideoReader=object)

    from videochat_flash.modeling_videochat_flash import VideoChatFlashQwenForCausalLM, VideoChatFlashQwenConfig
    import retrieval_utils as RU
    import training_utils as TU

    ns = types.SimpleNamespace(Model=VideoChatFlashQwenForCausalLM, Config=VideoChatFlashQwenConfig, RU=RU, TU=TU)
    _loaded["ns"] = ns
    return ns


def build_model(ocfg, weights):
    """Reference model (fp32, eval, eager attention) carrying `weights` (canonical names, numpy)."""
    import torch
    ns = load()
    cfg = ns.Config(vocab_size=ocfg.vocab_size, hidden_size=ocfg.hidden_size, intermediate_size=ocfg.intermediate_size,
                    num_hidden_layers=ocfg.num_layers, num_attention_heads=ocfg.num_heads,
                    num_key_value_heads=ocfg.num_kv_heads, max_position_embeddings=2048, rms_norm_eps=ocfg.rms_eps,
                    use_sliding_window=False, attn_implementation="eager")
    cfg.rope_theta = ocfg.rope_theta
    cfg.attention_dropout = 0.0
    cfg.use_cache = False
    cfg.mm_vision_tower = "umt-hd-fake"
    cfg.delay_load = True
    cfg.mm_projector_type = "tome16_mlp_hd64"
    cfg.mm_hidden_size = ocfg.mm_hidden_size
    cfg.mm_local_num_frames = 4
    cfg.mm_vision_select_layer = -2
    cfg.mm_pos_num_frames = 8
    cfg.vision_encode_type = "video_image"
    cfg.mm_patch_merge_type = "spatial_nopad"
    cfg.mm_newline_position = "nothing"
    cfg.mm_llm_compress = False
    with torch.no_grad():
        model = ns.Model(cfg).eval().float()
        sd = model.state_dict()
        for name, arr in weights.items():
            key = ref_key(name)
            assert key in sd, (name, key)
            assert tuple(sd[key].shape) == tuple(arr.shape), (name, sd[key].shape, arr.shape)
            sd[key].copy_(torch.from_numpy(arr))
    return model


def ref_key(name: str) -> str:
    """canonical tensor name -> reference state_dict key."""
    if name == "embed_tokens":
        return "model.embed_tokens.weight"
    if name == "final_norm":
        return "model.norm.weight"
    if name == "lm_head":
        return "lm_head.weight"
    if name == "visual_head":
        return "visual_head.weight"
    if name.startswith("mlp.") or name.startswith("tvg_mlp."):
        p, idx, kind = name.split(".")
        return f"model.mm_projector.{p}.{idx}.{'weight' if kind == 'w' else 'bias'}"
    assert name.startswith("layers."), name
    _, i, rest = name.split(".", 2)
    if rest == "input_norm":
        return f"model.layers.{i}.input_layernorm.weight"
    if rest == "post_norm":
        return f"model.layers.{i}.post_attention_layernorm.weight"
    proj, kind = rest.split(".")
    grp = "self_attn" if proj in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
    return f"model.layers.{i}.{grp}.{proj}.{'weight' if kind == 'w' else 'bias'}"


class DDPish:
    """Stands in for DistributedDataParallel: the reference calls model.module.* (retrieval_utils.py:66, 105)."""

    def __init__(self, m):
        self.module = m

    def __call__(self, *a, **k):
        return self.module(*a, **k)

    def eval(self):
        self.module.eval()
        return self
