"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/truncate.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_truncate

The `tiny` model and problem of oracle/gen_golden.py with `config.tokenizer_model_max_length` set, so that
prepare_inputs_labels_for_multimodal cuts the spliced rows (modeling_videochat_flash.py:452-457): VTG rows are 63 - 69
tokens long after the splice and the limit is 64 -- rows 0, 1, 4 and 5 lose 1 - 5 response tokens (their scores average
fewer terms), rows 2 and 3 are untouched; TVG rows (38 - 44 tokens) are below the limit.  Stored: the prepared masks /
labels / embeddings, the forward's VTG scores, and all six pass kinds through the reference's scoring loops.
The fixture is data: no reference source text is stored."""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from blim_amd import synth  # noqa: E402
from oracle import gen_golden as G  # noqa: E402
from oracle import ref_harness  # noqa: E402
from oracle.blim_oracle import OracleConfig  # noqa: E402

LIMIT = 64


def main(out_dir: str) -> None:
    import torch
    torch.set_num_threads(8)
    spec = G.CASES["tiny"]
    dims = synth.ModelDims(**spec["dims"])
    weights = synth.synthetic_weights(dims, spec["wseed"])
    prob = G.problem_of(spec, dims)
    ns = ref_harness.load()
    model = ref_harness.build_model(OracleConfig(**spec["dims"]), weights)
    model.config.tokenizer_model_max_length = LIMIT
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    ddp = ref_harness.DDPish(model)
    T = lambda a: torch.from_numpy(np.asarray(a))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    out = {"limit": np.array(LIMIT)}
    vtg = ns.RU.padding_ids([T(x) for x in prob.vtg_ids], [T(x) for x in prob.vtg_labels], [T(x) for x in prob.vtg_masks], tok)
    video = [T(v) for v in prob.video]
    n = spec["n"]
    with torch.no_grad():
        ids, lab, msk = vtg
        r = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, lab, video, ["video"] * n, image_sizes=None, video_feature=True,
                                                       tvg=False, cpn=True)
        (_, _, (m, cm), _, emb, lab2) = r
        assert emb.shape[1] == LIMIT and int((lab2 != -100).sum()) < int((lab != -100).sum()), "the limit must cut response tokens"
        out["prep_vtg_mask"] = m.numpy(); out["prep_vtg_cpn_mask"] = cm.numpy()
        out["prep_vtg_embeds"] = emb.numpy(); out["prep_vtg_labels"] = lab2.numpy()
        for tag, mm in (("", m), ("_cpn", cm)):
            o = model(inputs_embeds=emb, attention_mask=mm)
            out[f"fwd_vtg{tag}_score"] = ns.RU.vtg_criterion(o.logits, lab2).numpy()
    G.run_passes(out, "S_", ns.RU, ddp, torch.device("cpu"), prob, spec, dims, list(G.PASS_KINDS), "truncate")
    out["meta_case"] = np.array("truncate")
    path = os.path.join(out_dir, "truncate.npz")
    np.savez_compressed(path, **out)
    print(f"[truncate] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    main(a.out)
