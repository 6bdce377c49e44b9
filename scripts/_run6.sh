cd /tmp; export TMPDIR=/tmp; R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out/prof_pwc16
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $R/scripts/pwc_prof.py 3 fp16 > /tmp/pp.log 2>&1; echo rc=$?
python $R/scripts/summarize_prof.py /tmp/pp > $R/gpurun_out/prof_pwc16/summary.txt 2>&1; head -40 $R/gpurun_out/prof_pwc16/summary.txt
