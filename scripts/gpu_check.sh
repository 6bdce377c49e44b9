#!/bin/bash
# GPU-box check of a round: smoke, GPU parity tests, the default bench line, and the N>1 control flow of both
# parallelism modes with every rank on cuda:0 (gloo; a 1-GPU box cannot host several RCCL ranks).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider --durations=8 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -${PYTEST_TAIL:-40} gpurun_out/pytest_gpu.log
fi
if [ -z "$SKIP_BENCH" ]; then
echo "== bench default"; timeout 1200 python bench.py --layer-profile gpurun_out/layers_default.json ${BENCH_ARGS} > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-6000; tail -5 gpurun_out/bench_default.err
fi
if [ -z "$SKIP_DIST" ]; then
export FISR_BENCH_BACKEND=gloo FISR_BENCH_ONE_DEVICE=1
echo "== bench 2 ranks frame-parallel (one device, gloo)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --precision bf16x3 --no-roofline > gpurun_out/bench_2rank_frame.log 2> gpurun_out/bench_2rank_frame.err; echo "rc=$?"; tail -1 gpurun_out/bench_2rank_frame.log | cut -c1-1500; tail -3 gpurun_out/bench_2rank_frame.err
echo "== bench 4 ranks tile-parallel (one device, gloo)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 4 --steps 2 --warmup 1 --precision bf16x3 --parallelism tile --no-roofline > gpurun_out/bench_4rank_tile.log 2> gpurun_out/bench_4rank_tile.err; echo "rc=$?"; tail -1 gpurun_out/bench_4rank_tile.log | cut -c1-1500; tail -3 gpurun_out/bench_4rank_tile.err
fi
