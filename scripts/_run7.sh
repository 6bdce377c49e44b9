cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 3000 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider --durations=10 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
echo "== bench default"; timeout 1500 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-3000; tail -5 gpurun_out/bench_default.err
