#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for tag in ${CB_TAGS:-main a1 a2 a3 a4}; do
  so=$PWD/fisr_amd/libfisr_hip_$tag.so; [ "$tag" = main ] && so=$PWD/fisr_amd/libfisr_hip.so
  TAG=$tag FISR_HIP_SO=$so timeout 600 python scripts/conv_bench.py ${CB_PRECS:-bf16x3} 2>&1 | grep -v amdgpu.ids
done
