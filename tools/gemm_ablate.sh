#!/bin/bash
# Run ON THE GPU BOX: the plain-epilogue GEMM on the decoder's shapes with the product library and with the main-loop ablation builds
# (make -C blim_amd/csrc ablate), each in its own process, two rounds; then the LDS counters of the product and the wread1 build.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ablate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for lib in "" tools/bin/libblim_hip_ablate_wread1.so tools/bin/libblim_hip_ablate_wread2.so tools/bin/libblim_hip_ablate_dma1.so; do
    echo "== round $round lib ${lib:-product}"; BLIM_LIB=$lib python3 $R/tools/gemm_prof.py 5
  done
done > $OUT/times.txt 2>&1
for lib in "" tools/bin/libblim_hip_ablate_wread1.so; do
  n=$(basename ${lib:-product} .so)
  BLIM_LIB=$lib rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/lds_$n -- python3 $R/tools/gemm_prof.py 2 > $OUT/lds_$n.log 2>&1
  BLIM_LIB=$lib rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_$n -- python3 $R/tools/gemm_prof.py 2 > $OUT/mfma_$n.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -i -E "dram|mall|hbm|EA0_RD|EA0_WR|LDS" | head -80 > $OUT/counters.txt
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/lds_*/") + glob.glob("$OUT/mfma_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_kernel" not in r["Kernel_Name"]: continue
            key = (r["Kernel_Name"][:60], r["Grid_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
    print(d)
    for key, c in agg.items():
        print("  ", key, {k: round(v / cnt[(key, k)]) for k, v in c.items()})
PY
