#!/bin/bash
# quick Winograd iteration: parity subset + micro-benchmark + per-workgroup trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -rf -p no:cacheprovider -x -k "winograd" > gpurun_out/pytest_wino.log 2>&1; echo "pytest rc=$?"; tail -${PYTEST_TAIL:-6} gpurun_out/pytest_wino.log | cut -c1-800
TAG=${TAG:-w8} timeout 600 python scripts/conv_bench.py ${PRECS:-fp32w} 2>&1 | tee gpurun_out/convbench_quick.log
timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 544 992 64 64 3 1 fp32w
timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 136 248 256 256 3 1 fp32w
