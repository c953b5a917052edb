"""TEST INFRASTRUCTURE ONLY -- golden vectors for the dataset front end, recorded from the REFERENCE's dataloader
(dataloader/base_dataset.py + msrvtt/didemo/activitynet/lsmdc.py) run on the synthetic trees of tests/dataset_fixture.py.
Build container only:  python -m oracle.gen_golden_dataset"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness  # noqa: E402
import dataset_fixture as F  # noqa: E402


def main():
    ref_harness.load()                      # sys.path + stub modules for the reference's imports
    import dataloader as refdl              # the reference's package (from /root/reference)
    out = {}
    cwd = os.getcwd()
    for ds in ("MSRVTT", "DiDeMo", "ActivityNet", "LSMDC"):
        with tempfile.TemporaryDirectory() as tmp:
            F.build_tree(tmp, ds)
            os.chdir(tmp)                   # the reference reads ./data/... relative to the working directory
            try:
                args = types.SimpleNamespace(dataset=ds, batch_size_eval=4, num_workers=0, pin_mem=False)
                loader = refdl.load_data(args, tokenizer=F.StubTokenizer(), image_processor=None, split="test")
                dset = loader.dataset
                out[f"{ds}_n"] = np.array(len(dset))
                out[f"{ds}_tvg_prefix_length"] = np.array(dset.tvg_prefix_length)
                out[f"{ds}_vids"] = np.array(dset.vids)
                out[f"{ds}_video_vocab_sub"] = dset.video_vocab.float().numpy()[:, :, ::64]
                nb = 0
                for bi, batch in enumerate(loader):
                    nb += 1
                    out[f"{ds}_b{bi}_vid"] = np.array(batch["vid"])
                    out[f"{ds}_b{bi}_tvg_video_labels"] = batch["tvg_video_labels"].numpy()
                    out[f"{ds}_b{bi}_video_sum"] = np.array([float(v.float().sum()) for v in batch["video"]])
                    for k in ("vtg_ids", "vtg_labels", "vtg_masks", "tvg_ids", "tvg_labels", "tvg_masks"):
                        for j, t in enumerate(batch[k]):
                            out[f"{ds}_b{bi}_{k}_{j}"] = t.numpy()
                out[f"{ds}_n_batches"] = np.array(nb)
            finally:
                os.chdir(cwd)
    path = os.path.join(ROOT, "tests", "golden", "dataset.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
