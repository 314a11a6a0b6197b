#!/bin/bash
# Round 5: per-kernel time of the long tail of the all-severity sweep (run on the GPU box from the repo root) -> gpurun_out/r05_tail_kernels.txt
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in spatter glass_blur elastic_transform frost fog motion_blur snow zoom_blur; do
  rm -rf /tmp/pt
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/pt -o pt -- python $R/scratch/r5/profile_tail.py $c > /dev/null 2>&1
  echo "== $c"
  python $R/profiles/summarize_rocpd.py $(find /tmp/pt -name "*.db" | head -1) /tmp/pt/k.csv > /dev/null
  python - <<EOF
import csv
for r in list(csv.DictReader(open('/tmp/pt/k.csv')))[:12]:
    print('%-100s %4s %9.1f us avg %5.1f%%' % (r['kernel'][:100], r['calls'], float(r['avg_ns']) / 1e3, float(r['percent'])))
EOF
done
