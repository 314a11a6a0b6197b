#!/bin/bash
# Regenerates the round's evidence under gpurun_out/ (copied into profiles/ afterwards).  Run on the GPU box from the repo root.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_line.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_under_rocprof.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt -name "*.db" | head -1) $O/r02_bench_kernel_stats.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1) $O/r02_pmc_traffic.json > /dev/null
python $R/scratch/prof_engine2.py 2>/dev/null > $O/r02_igemm_per_shape.txt
for w in vit_inc vit_pgd adv_train; do python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_line_$w.json; done
ls -la $O
