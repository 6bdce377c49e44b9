#!/bin/bash
# The round's final measurement on ONE gpurun box (r06): the rocprofv3 trace + PMC passes per engine (scripts/gpu_profile.sh), their
# tables installed under profiles/ ON THE BOX so that the bench lines of the same library read them, then the default bench line and
# the `mixed` one with per-layer tables.  Everything lands in gpurun_out/final/; copy what should be judged into profiles/.
#   gpurun --timeout 3000 -- 'bash scripts/final_measure.sh'
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
F="$REPO/gpurun_out/final"
mkdir -p "$F"
cd "$REPO"
for eng in fp32 mixed f16f8; do
  PREC=$eng OTHERS=none STEPS=2 bash scripts/gpu_profile.sh > "$F/profile_$eng.log" 2>&1
  cp gpurun_out/prof_$eng/pmc_traffic.json profiles/pmc_traffic_$eng.json && cp profiles/pmc_traffic_$eng.json "$F/"
  cp gpurun_out/prof_$eng/summary.txt "$F/rocprofv3_${eng}_summary.txt"
  rm -rf gpurun_out/prof_$eng
done
PREC=fp32 STEPS=2 bash scripts/gpu_profile.sh > "$F/profile_all.log" 2>&1
cp gpurun_out/prof_fp32/pmc_traffic.json profiles/pmc_traffic.json && cp profiles/pmc_traffic.json "$F/"
cp gpurun_out/prof_fp32/summary.txt "$F/rocprofv3_all_engines_summary.txt"
rm -rf gpurun_out/prof_fp32
cd /tmp
timeout 900 python "$REPO/bench.py" --layer-profile "$F/layers_fp32.json" > "$F/bench_default.json" 2> "$F/bench_default.err"; echo "bench default rc=$?"
timeout 600 python "$REPO/bench.py" --precision mixed --others none --no-cpu-baseline --no-train --layer-profile "$F/layers_mixed.json" > "$F/bench_mixed.json" 2> "$F/bench_mixed.err"; echo "bench mixed rc=$?"
tail -c 1500 "$F/bench_default.json"
