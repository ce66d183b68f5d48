#!/bin/bash
# ablation builds of the split-bf16 GEMM (timing only; wrong numbers): the library with one ingredient of the
# K loop removed, selected at run time with PLNLP_HIP_LIB
set -e
cd "$(dirname "$0")/.."
B=plnlp_amd/build
python -m plnlp_amd.build > /dev/null
mkdir -p $B/abl
VARIANTS=${VARIANTS:-"NOSPLIT NOSTORE NOGLOAD"}
OBJS=$(ls $B/*.o | grep -v gemm_f32_x3.o)
for v in $VARIANTS; do
  flag="-DABL_X3_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -x hip -c plnlp_amd/csrc/gemm_f32.hip \
      -DPLNLP_GEMM_BK=16 -DPLNLP_GEMM_X3=1 -DPLNLP_ABLATION $flag -o $B/abl/x3_$v.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libplnlp_hip_x3_$v.so $OBJS $B/abl/x3_$v.o
done
ls -la $B/abl/*.so
