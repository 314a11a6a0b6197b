#!/bin/bash
# Regenerates the round-5 evidence under gpurun_out/r05/ (copied into profiles/ afterwards).  Run on the GPU box from the repo root.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's line: reference-precision headline, fast_mode, secondary (configs 4 and 5), cpu_baseline
python $R/bench.py --steps 20 --warmup 5 2>$O/r05_bench_line.err | tail -1 > $O/r05_bench_line.json
# 2. the same command under the kernel trace (both engines in one process: headline + fast_mode)
RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r05_bench_under_rocprof.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt -name "*.db" | head -1) $O/r05_bench_kernel_stats.csv > /dev/null
# 3. HBM traffic: two separate PMC passes of the same command (kernel trace + one counter each)
for c in FETCH_SIZE WRITE_SIZE; do RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1) $O/r05_pmc_traffic.json > /dev/null
# 4. per-launch tables of one gradient evaluation, both engines
python $R/scratch/prof_engine2.py 2>/dev/null > $O/r05_igemm_per_shape.txt
PREC=fp32x python $R/scratch/prof_engine2.py 2>/dev/null > $O/r05_igemm_per_shape_fp32x.txt
# 5. SQ counters of the reference-precision gradient evaluations (after 6a28c0a's LDS layout of the stem pair kernels)
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/prof_sq -o c -- python $R/scratch/r4/one_x3_grad_eval.py > /dev/null 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/prof_sq -name "*.db" | head -1) $O/r05_x3_counters.json > /dev/null
# 6. the corruption sweep at B = 256, severity 3: events, kernel trace, two PMC passes
python $R/profiles/corruption_sweep.py --events 2>/dev/null > $O/r05_corruption_sweep.txt; cp $R/gpurun_out/corruption_events.json $O/r05_corruption_events.json
rocprofv3 --kernel-trace --stats -d /tmp/corr_kt -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/corr_$c -o sweep -- python $R/profiles/corruption_sweep.py > /dev/null 2>&1; done
python $R/profiles/summarize_corruptions.py $(find /tmp/corr_kt -name "*.db" | head -1) $(find /tmp/corr_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/corr_WRITE_SIZE -name "*.db" | head -1) $O/r05_corruption_events.json $O/r05_corruption_kernels.csv > /dev/null
# 7. the gaussian_noise kernels alone under the trace (the live figure of the bench line beside rocprof's average)
rocprofv3 --kernel-trace --stats -d /tmp/prof_noise -o n -- python $R/profiles/noise_roofline_target.py 2>/dev/null | tail -1 > $O/r05_noise_roofline_live.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_noise -name "*.db" | head -1) $O/r05_noise_roofline_kernel_stats.csv > /dev/null
# 8. ViT ImageNet-C sweep under the trace (where the corruption kernels sit inside config 4)
rocprofv3 --kernel-trace --stats -d /tmp/prof_vit -o v -- python $R/bench.py --workload vit_inc --steps 1 --warmup 1 --no-reference-precision > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vit -name "*.db" | head -1) $O/r05_vit_inc_kernel_stats.csv > /dev/null
ls -la $O
