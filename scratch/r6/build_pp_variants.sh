#!/bin/bash
# Round 6: lab builds of the ping-pong pair GEMM (scratch/r6/pp/lib_<variant>.so = the product library with gemm_pair_pp.o replaced).  A variant is a
# '+'-joined list of tokens:  LAB (every schedule option, chosen per call by RART_PP_OPT)   STAMPS (s_memtime sums per phase and group, read by
# rart_debug_pp_stamps)   OPT<n> (default option n)   NOMFMA / NOLOAD / NOWAIT / NOREAD (knock-outs of the K loop)
set -e
R=/root/repo; O=$R/scratch/r6/pp; mkdir -p $O
OBJS=$(ls $R/robustart_amd/csrc/_obj/*.o | grep -v '/gemm_pair_pp')
SRC=$R/robustart_amd/csrc/gemm_pair_pp.hip
for v in ${@:-STAMPS NOMFMA NOLOAD NOWAIT NOREAD}; do
  D=""
  for t in ${v//+/ }; do
    case $t in STAMPS) D="$D -DRART_PP_STAMPS";; LAB) D="$D -DRART_PP_LAB";; OPT*) D="$D -DRART_PP_DEFAULT_OPT=${t#OPT}";; *) D="$D -DRART_PP_KO_$t";; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -I $R/include -I $R/robustart_amd/csrc -c $SRC -o $O/pp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$v.so $OBJS $O/pp_$v.o
done
ls -la $O/*.so
