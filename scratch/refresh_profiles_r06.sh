#!/bin/bash
# Regenerates the round-6 evidence under gpurun_out/r06/ (copied into profiles/ afterwards).  Run on the GPU box from the repo root.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the driver's line: reference-precision headline, fast_mode, secondary (configs 4 and 5), cpu_baseline
python $R/bench.py --steps 20 --warmup 5 2>$O/r06_bench_line.err | tail -1 > $O/r06_bench_line.json
# 2. the same command under the kernel trace: two streams (the step as it runs) and ONE stream (durations without co-scheduling)
RART_BENCH_NO_4X=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/r06_bench_under_rocprof.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt -name "*.db" | head -1) $O/r06_bench_kernel_stats.csv > /dev/null
RART_BENCH_NO_4X=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt1 -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast-mode --one-stream > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_kt1 -name "*.db" | head -1) $O/r06_bench_kernel_stats_one_stream.csv > /dev/null
# 3. HBM traffic: two separate PMC passes of the same command (kernel trace + one counter each)
for c in FETCH_SIZE WRITE_SIZE; do RART_BENCH_NO_4X=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$c -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1; done
python $R/profiles/summarize_pmc.py $(find /tmp/prof_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/prof_WRITE_SIZE -name "*.db" | head -1) $O/r06_pmc_traffic.json > /dev/null
# 4. per-launch tables of one gradient evaluation, both engines
python $R/scratch/prof_engine2.py 2>/dev/null > $O/r06_igemm_per_shape.txt
PREC=fp32x python $R/scratch/prof_engine2.py 2>/dev/null > $O/r06_igemm_per_shape_fp32x.txt
# 5. SQ counters of the reference-precision gradient evaluations (ResNet-50 and ViT-B/16)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/prof_sq -o c -- python $R/scratch/r4/one_x3_grad_eval.py > /dev/null 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/prof_sq -name "*.db" | head -1) $O/r06_x3_counters.json > /dev/null
# 6. texture-address unit: busy cycles and wave-instructions per kernel of the same gradient evaluations
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum -d /tmp/prof_ta -o c -- python $R/scratch/r6/one_x3_grad_eval.py > /dev/null 2>&1
python $R/profiles/summarize_pmc_generic.py $O/r06_ta_counters.csv $(find /tmp/prof_ta -name "*.db" | head -1) > /dev/null
# 7. every (corruption, severity) pair at B = 256 and the severity-3 sweep
python $R/scratch/r5/sweep_all_severities.py > $O/r06_all_severities.log 2>&1; cp $R/gpurun_out/r05_all_severities.json $O/r06_all_severities.json 2>/dev/null
python $R/profiles/corruption_sweep.py --events 2>/dev/null > $O/r06_corruption_sweep.txt
# 8. the gaussian_noise kernels alone under the trace + their SQ counters
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_noise -o n -- python $R/profiles/noise_roofline_target.py 2>/dev/null | tail -1 > $O/r06_noise_roofline_live.json
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_noise -name "*.db" | head -1) $O/r06_noise_roofline_kernel_stats.csv > /dev/null
# 9. adv_train and the ViT ImageNet-C sweep under the trace
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_at -o a -- python $R/bench.py --workload adv_train --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_at -name "*.db" | head -1) $O/r06_adv_train_kernel_stats.csv > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_vit -o v -- python $R/bench.py --workload vit_inc --steps 1 --warmup 1 --no-reference-precision > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vit -name "*.db" | head -1) $O/r06_vit_inc_kernel_stats.csv > /dev/null
# 10. the lab tables of the ping-pong kernels
python $R/scratch/r6/time_pair_pp.py > $O/r06_pair_pp.log 2>&1; cp $R/gpurun_out/r06_pair_pp.json $O/r06_pair_pp.json
python $R/scratch/r6/time_gemm256.py > $O/r06_gemm256_pp.txt 2>&1
ls -la $O
# 11. (second half of round 6) ViT-B/16 reference precision: per-shape table of the pair GEMM (ping-pong / + remainder split / + 224-row tiles), kernel
#     trace + SQ counters of three gradient evaluations; the pass-quantisation probes
python $R/scratch/r6/time_vit_pair_shapes.py > /dev/null 2>&1; cp $R/gpurun_out/r06_vit_pair_shapes.txt $O/r06_vit_pair_shapes.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_vx -o v -- python $R/scratch/r6/one_vit_x3_grad_eval.py > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/prof_vx -name "*.db" | head -1) $O/r06_vit_x3_kernel_stats.csv > /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/prof_vc -o c -- python $R/scratch/r6/one_vit_x3_grad_eval.py > /dev/null 2>&1
python $R/profiles/summarize_counters.py $(find /tmp/prof_vc -name "*.db" | head -1) $O/r06_vit_x3_counters.json > /dev/null
python $R/scratch/r6/time_pair_rounds.py > $O/r06_pair_rounds.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d /tmp/prof_rd -o c -- python $R/scratch/r6/pair_rounds_target.py > /dev/null 2>&1
python $R/scratch/r6/pmc_per_dispatch.py $(find /tmp/prof_rd -name "*.db" | head -1) gemm_pair > $O/r06_pair_rounds_counters.txt
python $R/scratch/r6/batch_sweep_x3.py 224 240 244 248 252 256 260 272 288 320 > $O/r06_batch_sweep_x3.txt 2>&1
python $R/scratch/r6/vit_pp_trace.py > $O/r06_vit_pp_trace.txt 2>&1
ls -la $O
