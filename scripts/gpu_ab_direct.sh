#!/bin/bash
# A/B of the direct conv kernel: parity subset with the in-tree library + micro-benchmark of it against build_ab/*.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -rf -p no:cacheprovider -x -k "conv or d2s or forward_bf16x3 or forward_f16f8 or batch" > gpurun_out/pytest_direct.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_direct.log | cut -c1-400
export CONV_SHAPES="12,544,992,64,64,3,1;12,544,992,64,64,3,0;12,272,496,128,128,3,1;12,136,248,256,256,3,1;12,544,992,64,256,7,0;12,136,248,512,256,2,0"
for so in build_ab/*.so fisr_amd/libfisr_hip.so; do
  [ -f "$so" ] || continue
  FISR_HIP_SO=$PWD/$so TAG=$(basename $so) timeout 900 python scripts/conv_bench.py ${PRECS:-bf16x3 f16f8 fp32}
done 2>&1 | tee gpurun_out/ab_direct.log
