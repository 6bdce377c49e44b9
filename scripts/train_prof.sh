#!/bin/bash
# rocprofv3 kernel trace of the training step (weight gradients on the main stream: the durations add up), binned by kernel and grid size (which map sizes the time goes to), after the
# gradient tests.  bash scripts/train_prof.sh   ->  gpurun_out/train_kernels.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests/test_gpu_train.py -q --no-header -x -p no:cacheprovider > gpurun_out/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_train.log
fi
timeout 600 python scripts/train_bench.py 8 96 5 2>&1 | tail -1
timeout 600 python scripts/train_bench.py 8 96 5 serial 2>&1 | tail -1
rm -rf /tmp/trprof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/trprof -- python scripts/train_bench.py 8 96 3 serial > gpurun_out/train_prof.log 2>&1
python - <<'PY' > gpurun_out/train_kernels.txt
import csv, glob, collections
f = glob.glob('/tmp/trprof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 5
agg = collections.defaultdict(lambda: [0, 0.0])
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r['Kernel_Name'].split('(')[0][:70]
    g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    agg[(name, g)][0] += 1; agg[(name, g)][1] += d
    per[name][0] += 1; per[name][1] += d
tot = sum(v[1] for v in per.values())
print(f"GPU time per step {tot / steps:.1f} ms ({len(rows) // steps} launches)")
for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n:72s} {v[0] / steps:6.0f} launches/step {v[1] / steps:8.2f} ms/step {100 * v[1] / tot:5.1f}%")
print()
for (n, g), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n:60s} wgs {g:6d} {v[0] / steps:6.1f} launches/step {v[1] / steps:8.3f} ms/step  avg {1e3 * v[1] / v[0]:8.1f} us")
PY
head -75 gpurun_out/train_kernels.txt
