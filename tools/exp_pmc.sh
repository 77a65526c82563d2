cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MT4_DBG=${1:-2}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_a -o a -- python3 tools/layer_bench.py 256 256 3 1 14 --tile 15 --iters 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc_b -o b -- python3 tools/layer_bench.py 256 256 3 1 14 --tile 15 --iters 6 > /dev/null 2>&1
ls gpurun_out/pmc_a gpurun_out/pmc_b
