#!/bin/bash
# Regenerates the round-4 evidence under gpurun_out/r04/ (copied into profiles/ afterwards).  Run on the GPU box from the repo root.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04_bench_line.json
RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_under_rocprof.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt -name "*.db" | head -1) $O/r04_bench_kernel_stats.csv > /dev/null
# HBM traffic: two separate PMC passes of the same command (kernel trace + one counter each; the reference-precision block stays IN so that
# k_gemm_pair gets its traffic too)
for c in FETCH_SIZE WRITE_SIZE; do RART_BENCH_NO_4X=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1) $O/r04_pmc_traffic.json > /dev/null
python $R/scratch/prof_engine2.py 2>/dev/null > $O/r04_igemm_per_shape.txt
PREC=fp32x python $R/scratch/prof_engine2.py 2>/dev/null > $O/r04_igemm_per_shape_fp32x.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_noise -o n -- python $R/profiles/noise_roofline_target.py 2>/dev/null | tail -1 > $O/r04_noise_roofline_live.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_noise -name "*.db" | head -1) $O/r04_noise_roofline_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_vitx3 -o v -- python $R/scratch/r4/time_vit_x3.py > $O/r04_vit_x3_times.txt 2>/dev/null
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vitx3 -name "*.db" | head -1) $O/r04_vit_x3_kernel_stats.csv > /dev/null
for w in vit_inc vit_pgd adv_train; do python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_line_$w.json; done
python $R/scratch/r4/time_square.py > /dev/null 2>&1; cp $R/gpurun_out/r04_square_vs_forward.json $O/ 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/prof_adv -o a -- python $R/bench.py --workload adv_train --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_adv -name "*.db" | head -1) $O/r04_adv_train_kernel_stats.csv > /dev/null
python $R/scratch/r4/time_wgrad.py 2>/dev/null > $O/r04_wgrad_per_layer.txt
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_adv_$c -o a -- python $R/bench.py --workload adv_train --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_adv_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_adv_WRITE_SIZE -name "*.db" | head -1) $O/r04_adv_train_pmc_traffic.json > /dev/null
for f in "" "skip=0" "stats=0" "bits=0" "direct=0" "skip=0,stats=0,bits=0,direct=0"; do echo "[$f] $(RART_TRAIN_FLAGS=$f python $R/bench.py --workload adv_train --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)"; done > $O/r04_adv_train_switches.txt
ls -la $O
