#!/bin/bash
# Run ON THE GPU BOX: SQ / TA / TCP counters of the compensated GEMM kernel (16-bit pass + e2m3 pass), one counter group per pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/lo6_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/gemm_lo6_target.py > $OUT/$1.log 2>&1 || tail -3 $OUT/$1.log; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
run b "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
run c "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
run d "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/[abcd]/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_kernel" not in r["Kernel_Name"]: continue
            agg[(r["Kernel_Name"][:48], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, c in agg.items():
        print(key, {k: f"{sum(v)/len(v):.4g}" for k, v in c.items()}, "n=", len(next(iter(c.values()))))
PY
