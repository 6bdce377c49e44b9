cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for so in build_ab/*.so; do
for shp in "12 544 992 64 64 3 1" "12 136 248 256 256 3 1" "12 68 124 512 512 3 1" "12 544 992 64 256 7 0"; do
  echo "== $(basename $so) $shp"; FISR_HIP_SO=$PWD/$so timeout 120 python scripts/trace_conv.py /tmp/t.bin $shp ${TPREC:-fp16} 2>&1 | grep -v "amdgpu.ids" | tail -3
done; done
