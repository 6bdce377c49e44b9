#!/bin/bash
# One GPU-box round: smoke, GPU parity tests, short benches (+ per-layer profile).  Logs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/smoke.log
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -${PYTEST_TAIL:-30} gpurun_out/pytest_gpu.log
fi
for pr in ${BENCH_PRECS:-fp32}; do
for b in ${BENCH_BATCHES:-stack}; do
echo "== bench $pr $b"; timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 --batch $b --precision $pr --layer-profile gpurun_out/layers_${pr}_$b.json ${BENCH_ARGS} > gpurun_out/bench_${pr}_$b.log 2> gpurun_out/bench_${pr}_$b.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_${pr}_$b.log | cut -c1-1800; tail -5 gpurun_out/bench_${pr}_$b.err
done
done
