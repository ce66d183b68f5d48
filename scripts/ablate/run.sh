#!/bin/bash
# build + run the GEMM ablations / tuning variants on the GPU box
cd "$(dirname "$0")"
for v in "-DPLNLP_GEMM_PF=2" "-DPLNLP_GEMM_PF=1" "-DABL_NOSTORE" "-DABL_NOGLOAD" "-DABL_NOSTAGE" "-DABL_NOBARRIER" "-DABL_NOSTAGE -DABL_NOGLOAD" "-DABL_NOSTORE -DABL_NOGLOAD" "-DABL_NOSTORE -DABL_NOGLOAD -DABL_NOSTAGE -DABL_NOBARRIER"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPLNLP_ABLATION $v gemm_ablate.cpp -o /tmp/gemm_ablate 2>/dev/null || { echo "build failed $v"; continue; }
  echo "== [$v]"; /tmp/gemm_ablate 235868 256 512; /tmp/gemm_ablate 130000 512 256; /tmp/gemm_ablate 4096 4096 4096
done
