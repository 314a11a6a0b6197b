#!/bin/bash
# Round 5, second half: the corruption evidence again on the final tree (steps 6 and 8 of refresh_profiles_r05.sh + the long-tail kernels) -> gpurun_out/r05/
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/profiles/corruption_sweep.py --events 2>/dev/null > $O/r05_corruption_sweep.txt; cp $R/gpurun_out/corruption_events.json $O/r05_corruption_events.json
rocprofv3 --kernel-trace --stats -d /tmp/corr_kt -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/corr_$c -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1; done
python $R/profiles/summarize_corruptions.py $(find /tmp/corr_kt -name "*.db" | head -1) $(find /tmp/corr_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/corr_WRITE_SIZE -name "*.db" | head -1) $O/r05_corruption_events.json $O/r05_corruption_kernels.csv > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_vit -o v -- python $R/bench.py --workload vit_inc --steps 1 --warmup 1 --no-reference-precision > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vit -name "*.db" | head -1) $O/r05_vit_inc_kernel_stats.csv > /dev/null
bash $R/scratch/r5/profile_tail.sh > $O/r05_tail_kernels.txt 2>&1
ls -la $O
