#!/bin/bash
# tail kernel with the first wave of workgroups of slots 1 / 2 delayed by S / S2 sleeps of 3.4 us (scratch/r4/conv_tail_stagger.hip)
cd $GRAFT_REPO_ROOT
for C in 64 128; do
for S in "0 0" "4 8" "8 16" "12 24" "16 32" "8 0" "16 0" "24 0"; do
  set -- $S
  echo "C=$C stagger $1 $2: $(C=$C RART_STAGGER=$1 RART_STAGGER2=$2 python scratch/r4/time_tail.py scratch/r4/ko/lib_stagger.so | tail -1)"
done; done
