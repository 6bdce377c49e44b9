#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests (MR=1)"; timeout 900 python -m pytest tests -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -5
echo "== tests (MR=2)"; FISR_CONV_MR=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | tail -3
for mr in 1 2; do
  echo "== convbench MR=$mr"; TAG=mr$mr FISR_CONV_MR=$mr timeout 600 python scripts/conv_bench.py bf16x3 fp32 2>&1 | grep -v amdgpu.ids
done
for mr in 1 2; do for pr in bf16x3 fp32; do
  FISR_CONV_MR=$mr timeout 600 python bench.py --steps 3 --warmup 1 --precision $pr --no-cpu-baseline --no-fp32-ref > gpurun_out/mr${mr}_$pr.log 2>gpurun_out/mr${mr}_$pr.err
  python - <<PY
import json
l=json.loads(open("gpurun_out/mr${mr}_$pr.log").read().strip().splitlines()[-1]); r=l["roofline"]
print("MR=$mr $pr: fps %.2f ms/step %.1f conv TF %.1f (all %.1f)" % (l["value"], l["ms_per_step"], r["achieved"], r["all_conv_tflops"]))
PY
done; done
