#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for a in ${ABLS:-0 1 2 4 5 8 16 64}; do
  echo "== WABL=$a"
  FISR_HIP_SO=$PWD/fisr_amd/abl_$a.so timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 544 992 64 64 3 1 fp32w 2>&1 | tail -3
  FISR_HIP_SO=$PWD/fisr_amd/abl_$a.so timeout 120 python scripts/trace_conv.py /tmp/t.bin 12 136 248 256 256 3 1 fp32w 2>&1 | tail -1
done
