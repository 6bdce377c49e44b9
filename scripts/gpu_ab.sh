#!/bin/bash
# A/B of kernel builds: for each fisr_amd/libfisr_hip_<tag>.so run the bench (interleaved rounds).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for round in 1 2; do
for tag in ${AB_TAGS:-v0 v1 v2}; do
for pr in ${AB_PRECS:-bf16x3 fp32}; do
  FISR_HIP_SO=$PWD/fisr_amd/libfisr_hip_$tag.so timeout 600 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --precision $pr --no-cpu-baseline --no-fp32-ref > gpurun_out/ab_${tag}_${pr}_$round.log 2>gpurun_out/ab_${tag}_${pr}_$round.err
  python - <<PY
import json
try:
    l=json.loads(open("gpurun_out/ab_${tag}_${pr}_$round.log").read().strip().splitlines()[-1])
    r=l["roofline"]
    print("$tag $pr round $round: fps %.2f  ms/step %.1f  conv TF %.1f (all conv %.1f)  avg_us %.1f" % (l["value"], l["ms_per_step"], r["achieved"], r["all_conv_tflops"], r["avg_launch_us"]))
except Exception as e:
    print("$tag $pr round $round: FAILED", e); print(open("gpurun_out/ab_${tag}_${pr}_$round.err").read()[-600:])
PY
done; done; done
