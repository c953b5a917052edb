"""Summarise gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/: kernel stats CSV + HBM traffic JSON.
FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM);
WRITE_SIZE is taken as is.  Both come from the L2's fabric-side counters, so Infinity-Cache hits are included."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(f"{src}/stats/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"profiles/{tag}_bench_kernel_stats.csv")
def per_kernel(d):
    rows = list(csv.DictReader(open(glob.glob(f"{src}/{d}/*/*_counter_collection.csv")[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}
fetch, nf = per_kernel("fetch"); write, nw = per_kernel("write")
stats = {r["Name"]: r for r in csv.DictReader(open(ks))}
out = {}
for k in fetch:
    if "gemm_kernel" in k or "attn_kernel" in k or "rmsnorm" in k or "quant_rows" in k:
        out[k] = {"launches": nf[k], "fetch_size_kb_raw": fetch[k], "write_size_kb": write.get(k, 0.0),
                  "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024,
                  "avg_ns": float(stats[k]["AverageNs"]) if k in stats else None}
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
# matrix-pipe utilisation and effective clock per kernel (separate PMC passes; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
if glob.glob(f"{src}/mfma/*/*_counter_collection.csv") and glob.glob(f"{src}/clk/*/*_counter_collection.csv"):
    mf, _ = per_kernel("mfma"); ck, _ = per_kernel("clk")
    lines = [f"# {tag}: per-kernel summary of `python bench.py --steps 2 --warmup 1` under rocprofv3 (one PMC counter per pass)", "",
             "| kernel | launches | avg us | HBM-side GB / launch (2 x FETCH_SIZE + WRITE_SIZE) | MFMA busy cycles / launch | GRBM_GUI_ACTIVE / 8 | matrix-pipe busy | effective clock (GHz) |",
             "|---|---|---|---|---|---|---|---|"]
    for k in sorted(mf, key=lambda k: -(float(stats[k]["TotalDurationNs"]) if k in stats else 0)):
        if k not in stats or k not in ck or float(stats[k]["Percentage"]) < 0.5:
            continue
        cyc = ck[k] / 8.0
        avg_ns = float(stats[k]["AverageNs"])
        traffic = out.get(k, {}).get("hbm_bytes_per_launch")
        lines.append(f"| `{k[:60]}` | {stats[k]['Calls']} | {avg_ns / 1e3:.1f} | {traffic / 1e9:.2f} | {mf[k]:.3e} | {cyc:.3e} | "
                     f"{100 * mf[k] / (1024 * cyc):.1f} % | {cyc / avg_ns:.2f} |" if traffic is not None else
                     f"| `{k[:60]}` | {stats[k]['Calls']} | {avg_ns / 1e3:.1f} | - | {mf[k]:.3e} | {cyc:.3e} | {100 * mf[k] / (1024 * cyc):.1f} % | {cyc / avg_ns:.2f} |")
    open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
for k, v in out.items():
    print(k[:50], f"{v['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch", v["avg_ns"])
# the tag travels with the summary: profiles/CURRENT.json names, per kind of run ("plain_<dtype>" / "full_<dtype>"), the traffic file bench.py's roofline object looks up
# (usage: parse_profiles.py <tag> [plain|full] [dtype]) -- no more "newest file name wins"
if len(sys.argv) > 2:
    kind, dtype = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "f16")
    cur_path = "profiles/CURRENT.json"
    cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
    cur.setdefault("traffic", {})[f"{kind}_{dtype}"] = f"{tag}_traffic.json"
    cur.setdefault("command", {})[f"{kind}_{dtype}"] = open(f"{src}/command.txt").read().strip() if os.path.exists(f"{src}/command.txt") else None
    json.dump(cur, open(cur_path, "w"), indent=1)
