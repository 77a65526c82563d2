#!/bin/bash
# same-box A/B of a compile-time variant of igemm_conv.hip: tools/ab_build.sh "<-D flag>" (rebuilds on the GPU box between runs)
cd computervision_codes_amd/csrc
for v in 0 1 0 1; do
  if [ $v = 1 ]; then F="$1"; else F=""; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $F -c igemm_conv.hip -o igemm_conv.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 igemm_conv.o misc_kernels.o transformer_kernels.o train_kernels.o train2d_kernels.o -o ../libmt4hip.so
  echo -n "variant=$v: "; (cd ../.. && timeout -k 10 300 python bench.py --no-cpu-baseline --no-temporal 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
done
