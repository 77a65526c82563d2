#!/bin/bash
# round-final measurement set on the GPU box: bench line + per-layer table, rocprofv3 kernel stats, PMC traffic / MFMA passes for the
# spatial extractor (the headline kernel) and for the temporal head (per-video latency), kernel stats of the MS-TCT training step.
# Every rocprofv3 command has the program directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write $O/prof_final $O/pmc_mfma $O/prof_tenco_f32 $O/prof_tenco_bf16 $O/pmc_tenco_FETCH_SIZE $O/pmc_tenco_WRITE_SIZE $O/prof_mstct_train $O/prof_q2l_train $O/prof_tenco2000_f32 $O/prof_tenco2000_bf16 $O/prof_swin
cd $R
timeout -k 10 900 python bench.py --per-layer gpurun_out/layers_final.json > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo bench done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o final -- python3 $R/bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-temporal > $O/prof_final.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > $O/pmc_fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > $O/pmc_write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-temporal > $O/pmc_mfma.log 2>&1
echo mfma done
for dt in f32 bf16; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tenco_$dt -o tenco -- python3 $R/tools/tenco_infer_prof.py --dtype $dt > $O/prof_tenco_$dt.log 2>&1
done
for dt in f32 bf16; do      # a whole video (T = 2000): the unit the reference runs the head on
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tenco2000_$dt -o tenco -- python3 $R/tools/tenco_infer_prof.py --T 2000 --replays 20 --dtype $dt > $O/prof_tenco2000_$dt.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_tenco_$c -- python3 $R/tools/tenco_infer_prof.py --replays 3 > $O/pmc_tenco_$c.log 2>&1
done
echo tenco done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_swin -o swin -- python3 $R/tools/swin_all_prof.py > $O/prof_swin.log 2>&1
echo swin done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mstct_train -o mt -- python3 $R/tools/mstct_train_prof.py > $O/prof_mstct_train.log 2>&1
echo mstct done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q2l_train -o q2l -- python3 $R/tools/q2l_train_prof.py > $O/prof_q2l_train.log 2>&1
rm -f $O/prof_q2l_train/*kernel_trace.csv
echo q2l done
rm -rf $O/prof_cnn_train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cnn_train -o ct -- python3 $R/tools/bf16_train_prof.py > $O/prof_cnn_train.log 2>&1
rm -f $O/prof_cnn_train/*kernel_trace.csv
echo cnn train done
cd $R
cp profiles/traffic.json gpurun_out/traffic.json; cp profiles/mfma_util.json gpurun_out/mfma_util.json
python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 36 resnet50_bf16_b2672_224x224 gpurun_out/traffic.json
python tools/collect_mfma_util.py gpurun_out/pmc_mfma 36 resnet50_bf16_b2672_224x224 gpurun_out/mfma_util.json
python tools/collect_tcn_traffic.py gpurun_out/pmc_tenco_FETCH_SIZE gpurun_out/pmc_tenco_WRITE_SIZE tenco4_f32_T256 gpurun_out/traffic.json
echo all done
find gpurun_out/prof_final gpurun_out/prof_tenco_f32 gpurun_out/prof_tenco_bf16 gpurun_out/prof_tenco2000_f32 gpurun_out/prof_tenco2000_bf16 gpurun_out/prof_mstct_train gpurun_out/prof_swin -name "*kernel_trace.csv" -delete 2>/dev/null || true
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_tenco_FETCH_SIZE gpurun_out/pmc_tenco_WRITE_SIZE
