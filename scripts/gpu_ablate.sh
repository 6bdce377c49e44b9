#!/bin/bash
# A/B of Winograd kernel builds: per-workgroup traces of one launch with each FISR_DIAG build in build_ab/
# (python -c "from fisr_amd import lib; lib.build(diag=True, defines=['FISR_WABL=2'], out='build_ab/abl2.so')")
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$PARITY" ]; then
  for so in fisr_amd/libfisr_hip.so ${PARITY_ALL:+build_ab/*.so}; do
    [ -f "$so" ] || continue
    FISR_HIP_SO=$PWD/$so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -rf -p no:cacheprovider -x -k "${PARITY_K:-winograd}" > gpurun_out/pytest_wino.log 2>&1; echo "pytest $so rc=$?"; tail -3 gpurun_out/pytest_wino.log | cut -c1-600
  done
fi
for so in build_ab/*.so; do
  [ -f "$so" ] || continue
  for res in 1 0; do
    echo "== $so res=$res 64->64"
    FISR_HIP_SO=$PWD/$so timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 544 992 64 64 3 $res fp32w 2>&1 | grep -v "^shader"
  done
  echo "== $so 64->256 d2s"
  FISR_HIP_SO=$PWD/$so timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 544 992 64 256 7 0 fp32w 2>&1 | grep -v "^shader"
  echo "== $so 256->256 res"
  FISR_HIP_SO=$PWD/$so timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 136 248 256 256 3 1 fp32w 2>&1 | grep -v "^shader"
done 2>&1 | tee gpurun_out/ablate.log
