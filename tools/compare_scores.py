"""Per-pass worst / median relative deviation of score dumps (blim_amd.main --dump_scores) from the first one."""
import sys
import numpy as np
base = np.load(sys.argv[1])
for path in sys.argv[2:]:
    d = np.load(path)
    out = []
    for k in base.files:
        if k.endswith("internvideo2"):
            continue
        a, b = base[k], d[k]
        m = (a != -100.0) & (a != 0)
        assert np.array_equal(a != -100.0, b != -100.0), k
        rel = np.abs(b[m] - a[m]) / np.abs(a[m])
        same_order = np.mean([np.array_equal(np.argsort(-a[r], kind="stable"), np.argsort(-b[r], kind="stable")) for r in range(a.shape[0])])
        top1 = np.mean(np.argmax(a, axis=1) == np.argmax(b, axis=1))
        out.append(f"{k}: rel max {rel.max():.2e} median {np.median(rel):.1e}, abs max {np.abs(b[m] - a[m]).max():.3f} on |score| {np.abs(a[m]).min():.2f}..{np.abs(a[m]).max():.1f}, "
                   f"same top-1 candidate {100 * top1:.1f} % of rows, same full order {100 * same_order:.0f} %")
    print(f"{path.split('/')[-1]} vs {sys.argv[1].split('/')[-1]}:\n  " + "\n  ".join(out))
