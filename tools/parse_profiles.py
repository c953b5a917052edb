"""Summarise gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/: kernel stats CSV + HBM traffic JSON.
FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM);
WRITE_SIZE is taken as is.  Both come from the L2's fabric-side counters, so Infinity-Cache hits are included."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(f"{src}/stats/runc/*_kernel_stats.csv")[0]
shutil.copy(ks, f"profiles/{tag}_bench_kernel_stats.csv")
def per_kernel(d):
    rows = list(csv.DictReader(open(glob.glob(f"{src}/{d}/runc/*_counter_collection.csv")[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}
fetch, nf = per_kernel("fetch"); write, nw = per_kernel("write")
stats = {r["Name"]: r for r in csv.DictReader(open(ks))}
out = {}
for k in fetch:
    if "gemm_kernel" in k or "attn_kernel" in k or "rmsnorm" in k:
        out[k] = {"launches": nf[k], "fetch_size_kb_raw": fetch[k], "write_size_kb": write.get(k, 0.0),
                  "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024,
                  "avg_ns": float(stats[k]["AverageNs"]) if k in stats else None}
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k[:50], f"{v['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch", v["avg_ns"])
