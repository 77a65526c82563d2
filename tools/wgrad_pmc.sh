#!/bin/bash
# LDS / MFMA counters of the weight-gradient kernels on one ResNet-50 layer shape (GPU box): bash tools/wgrad_pmc.sh [shape index ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ONLY=${1:-12,14}
rm -rf $O/pmc_wg1 $O/pmc_wg2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/pmc_wg1 -- python3 $R/tools/wgrad_micro.py --only $ONLY --iters 3 > $O/pmc_wg1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_wg2 -- python3 $R/tools/wgrad_micro.py --only $ONLY --iters 3 > $O/pmc_wg2.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_wg1", "gpurun_out/pmc_wg2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "wgrad" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            print(k)
            for c, x in sorted(v.items()): print(f"    {c:34s} {x:.4g}")
PY
