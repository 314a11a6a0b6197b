#!/bin/bash
# Regenerates the round-3 evidence under gpurun_out/r03/ (copied into profiles/ afterwards).  Run on the GPU box from the repo root.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_line.json
RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_under_rocprof.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt -name "*.db" | head -1) $O/r03_bench_kernel_stats.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-reference-precision > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1) $O/r03_pmc_traffic.json > /dev/null
python $R/scratch/prof_engine2.py 2>/dev/null > $O/r03_igemm_per_shape.txt
PREC=fp32x python $R/scratch/prof_engine2.py 2>/dev/null > $O/r03_igemm_per_shape_fp32x.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_noise -o n -- python $R/profiles/noise_roofline_target.py 2>/dev/null | tail -1 > $O/r03_noise_roofline_live.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_noise -name "*.db" | head -1) $O/r03_noise_roofline_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_vit -o v -- python $R/scratch/prof_vit_fb.py > $O/r03_vit_fwd_bwd_times.txt 2>/dev/null
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vit -name "*.db" | head -1) $O/r03_vit_fwd_bwd_kernel_stats.csv > /dev/null
for w in vit_inc vit_pgd adv_train; do python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_line_$w.json; done
# corruption sweep: kernel trace, two PMC passes, event timing
rocprofv3 --kernel-trace --stats -d /tmp/corr_kt -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/corr_$c -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1; done
python $R/profiles/corruption_sweep.py --events > $O/r03_corruption_sweep.txt 2>&1
python $R/profiles/summarize_corruptions.py $(find /tmp/corr_kt -name "*.db" | head -1) $(find /tmp/corr_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/corr_WRITE_SIZE -name "*.db" | head -1) $R/gpurun_out/corruption_events.json $O/r03_corruption_kernels.csv > /dev/null
ls -la $O
