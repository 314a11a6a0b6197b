#!/bin/bash
# Round 5: knock-out builds of k_gemm_pair (scratch/r5/ko/lib_<variant>.so = the product library with gemm_pair.o replaced).  The variants are
# sed patches of the PRODUCT source, so there is no second copy of the kernel to drift:
#   BASE     the product file, rebuilt the same way (control)
#   NOLOAD   the K loop issues no global -> LDS loads after stage 0 (MFMAs + ds_reads + epilogue on stale tiles)
#   NOMFMA   the MFMAs are removed (loads + barriers + epilogue)
#   NOEPI    the epilogue is removed (one guarded store keeps the accumulators alive)
#   NORES    no residual loads      NOSTORE  no destination stores
set -e
R=/root/repo; O=$R/scratch/r5/ko; mkdir -p $O
OBJS=$(ls $R/robustart_amd/csrc/_obj/*.o | grep -v '/gemm_pair')
SRC=$R/robustart_amd/csrc/gemm_pair.hip
for v in BASE NOLOAD NOMFMA NOEPI NORES NOSTORE; do
  T=$O/gp_$v.hip
  case $v in
    BASE) cp $SRC $T;;
    NOLOAD) sed 's/if (kt + 1 < KT) RART_GP_ISSUE(kt + 1, buf ^ 1)/if (kt + 1 < KT \&\& d.K < 0) RART_GP_ISSUE(kt + 1, buf ^ 1)/' $SRC > $T;;
    NOMFMA) sed 's/acc\[i\]\[j\] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(\([a-z]*\)\[i\], \([a-z]*\)\[j\], acc\[i\]\[j\], 0, 0, 0);/acc[i][j][0] += (float)(\1[i][0]) * (float)(\2[j][0]) * 0.f;/' $SRC > $T;;
    NOEPI) sed 's|^  float\* sE = reinterpret_cast<float\*>(lds) + wave \* 32 \* GP_LDE;|  { float s_ = 0.f; for (int i = 0; i < MI; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s_ += acc[i][j][r]; if (s_ == 1.2345e-30f) d.dst_hi[0] = 1; return; }\n  float* sE = reinterpret_cast<float*>(lds) + wave * 32 * GP_LDE;|' $SRC > $T;;
    NORES) sed 's/if (d.res_hi) {/if (d.res_hi \&\& d.K < 0) {/' $SRC > $T;;
    NOSTORE) sed 's|^          \*reinterpret_cast<uint4\*>(d.dst_hi + e) = ph;|          if (v[0] == 1.2345e-30f) *reinterpret_cast<uint4*>(d.dst_hi + e) = ph;|; s|^          \*reinterpret_cast<uint4\*>(d.dst_lo + e) = pl;|          if (v[0] == 1.2345e-30f) *reinterpret_cast<uint4*>(d.dst_lo + e) = pl;|' $SRC > $T;;
  esac
  cmp -s $SRC $T && [ $v != BASE ] && { echo "patch $v did not apply"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $R/robustart_amd/csrc -c $T -o $O/gp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/lib_$v.so $OBJS $O/gp_$v.o

done
ls -la $O/*.so
