#!/bin/bash
# Winograd fp32 kernel: parity (both U-slab staging variants), per-shape micro-benchmark against the direct fp32
# kernel, whole-step bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 1 0; do
echo "== pytest winograd FISR_WINO_GLDS=$g"
FISR_WINO_GLDS=$g timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -rf -p no:cacheprovider -k "winograd" -s > gpurun_out/pytest_wino_glds$g.log 2>&1; echo "rc=$?"; tail -${PYTEST_TAIL:-25} gpurun_out/pytest_wino_glds$g.log | cut -c1-400
done
if [ -z "$SKIP_CONVBENCH" ]; then
for g in 1 0; do
echo "== conv_bench FISR_WINO_GLDS=$g"
FISR_WINO_GLDS=$g TAG=glds$g timeout 600 python scripts/conv_bench.py fp32w 2>&1 | tee gpurun_out/convbench_wino_glds$g.log
done
echo "== conv_bench direct fp32"
timeout 600 python scripts/conv_bench.py fp32 2>&1 | tee gpurun_out/convbench_fp32.log
fi
if [ -z "$SKIP_BENCH" ]; then
echo "== bench fp32w"
timeout 900 python bench.py --precision fp32w --others "fp32d" --steps 3 --no-cpu-baseline --layer-profile gpurun_out/layers_fp32w.json > gpurun_out/bench_fp32w.log 2> gpurun_out/bench_fp32w.err; echo "rc=$?"; tail -1 gpurun_out/bench_fp32w.log | cut -c1-5000; tail -5 gpurun_out/bench_fp32w.err
fi
