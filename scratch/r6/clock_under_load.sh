#!/bin/bash
# Round 6: shader clock and package power while the reference-precision gradient evaluation (ResNet-50, then ViT-B/16) loops, against idle.
R=$GRAFT_REPO_ROOT
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
for eng in one_x3_grad_eval one_vit_x3_grad_eval; do
  python - <<PY &
import sys, time
sys.path.insert(0, "$R")
import runpy, torch
src = open("$R/scratch/r6/$eng.py").read().replace("for _ in range(3): eng.forward_backward", "t0 = __import__('time').time()\nwhile __import__('time').time() - t0 < 14: eng.forward_backward")
exec(compile(src, "$eng", "exec"))
PY
  PID=$!
  sleep 7
  echo "== $eng under load"
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Socket Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 0.7; done
  wait $PID
done
