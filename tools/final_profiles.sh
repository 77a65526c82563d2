#!/bin/bash
# round-final measurement set on the GPU box: bench line + per-layer table, rocprofv3 kernel stats, PMC traffic passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_final
timeout -k 10 500 python bench.py --per-layer gpurun_out/layers_final.json > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o final -- python3 bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-temporal > gpurun_out/prof_final.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > gpurun_out/pmc_fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > gpurun_out/pmc_write.log 2>&1
echo write done
python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 54 resnet50_bf16_b1336_224x224 gpurun_out/traffic.json
rm -rf gpurun_out/pmc_mfma
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > gpurun_out/pmc_mfma.log 2>&1
echo mfma done
python tools/collect_mfma_util.py gpurun_out/pmc_mfma 54 resnet50_bf16_b1336_224x224 gpurun_out/mfma_util.json
