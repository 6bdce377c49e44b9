cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pwc_ops.py tests/test_gpu_pwc.py -m gpu -q --no-header -rf -p no:cacheprovider -s > gpurun_out/r3_pwc.log 2>&1; echo "pwc rc=$?"; grep -E "fp16|lds-dma|passed|failed|Error|error" gpurun_out/r3_pwc.log | tail -50
timeout 300 python scripts/pwc_prof.py 3 fp32 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python scripts/pwc_prof.py 3 fp16 2>&1 | grep -v amdgpu.ids | tail -3
