cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dma.py -m gpu -q --no-header -rf -p no:cacheprovider -s -x > gpurun_out/r3_dma.log 2>&1; echo "dma rc=$?"; grep -E "dma fp16|pred_l|passed|failed|Error|error|assert" gpurun_out/r3_dma.log | tail -40
