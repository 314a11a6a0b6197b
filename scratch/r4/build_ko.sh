#!/bin/bash
# knock-out builds of the tail kernel: full library with conv_tail_pair.o replaced (scratch/r4/ko/lib_<variant>.so)
set -e
R=/root/repo; O=$R/scratch/r4/ko; mkdir -p $O
OBJS=$(ls $R/robustart_amd/csrc/_obj/*.o | grep -v conv_tail_pair)
for v in BASE KO_MAIN KO_TAIL KO_W3 KO_RES KO_STORE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $R/robustart_amd/csrc -D$v -c $R/scratch/r4/conv_tail_ko.hip -o $O/ct_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$v.so $OBJS $O/ct_$v.o
done
ls -la $O/*.so
