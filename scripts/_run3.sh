cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dma.py -m gpu -q --no-header -rf -p no:cacheprovider -x 2>&1 | tail -2
export CONV_SHAPES="12,544,992,64,64,3,1;12,544,992,64,64,0,0;12,272,496,128,128,3,1;12,136,248,256,256,3,1;12,68,124,512,512,3,1;12,544,992,64,256,7,0;12,136,248,512,256,2,0"
for so in fisr_amd/libfisr_hip.so build_ab/*.so; do [ -f "$so" ] || continue; FISR_HIP_SO=$PWD/$so TAG=$(basename $so .so) timeout 600 python scripts/conv_bench.py ${PRECS:-fp16} 2>&1 | grep -v amdgpu.ids; done
