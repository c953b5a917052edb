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
        out.append(f"{k} {rel.max():.2e} (median {np.median(rel):.1e})")
    print(f"{path.split('/')[-1]} vs {sys.argv[1].split('/')[-1]}: " + ", ".join(out))
