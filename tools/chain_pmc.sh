#!/bin/bash
# where a chain_gemm wave's cycles go: SQ counters of the micro bench (two passes), summed per kernel
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/pmc_chain1 $O/pmc_chain2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_chain1 -- python3 $R/tools/chain_micro.py > $O/pmc_chain1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_chain2 -- python3 $R/tools/chain_micro.py > $O/pmc_chain2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for d in ("gpurun_out/pmc_chain1", "gpurun_out/pmc_chain2"):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        k = "chain_gemm_kernel" + nm.split("chain_gemm_kernel")[1].split(">")[0] + ">" if "chain_gemm_kernel" in nm else nm.split("(")[0][-70:]
        if "chain_gemm" in nm or "igemm_conv_kernel<unsigned short, 256, 256, 4, 4" in nm:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x:.4g}")
PY
