#!/bin/bash
# One GPU-box round: smoke, GPU parity tests, short bench.  Logs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -6 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -60 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
