#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the attention kernel on the SYN step and reference-shaped plans (tools/attn_classes.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/attn_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/tools/attn_classes.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- python3 $R/tools/attn_classes.py > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $R/tools/attn_classes.py > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/[abc]/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_kernel" not in r["Kernel_Name"]: continue
            agg[(r["Kernel_Name"][:44], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, c in agg.items():
        print(key, {k: f"{sum(v)/len(v):.4g}" for k, v in c.items()}, "n=", len(next(iter(c.values()))))
PY
