#!/bin/bash
# Builds signal_amd/lib/variants/libsignal_<name>.so: the GEMM translation units recompiled with extra -D flags, linked with the
# shipped objects of everything else.  For same-box A/B runs through SIGNAL_HIP_LIB (tools/ab_kernels.sh, tools/ab_multi.sh).
# usage: tools/build_variant.sh <name> "<-DFLAG=...> ..." [unit.hip ...]      (default units: the three GEMM files)
set -e
NAME=$1; DEFS=$2; shift 2 || true
UNITS=${@:-gemm_bf16.hip gemm_nt_persist.hip gemm_tn_grouped.hip}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/signal_amd/csrc
python3 $C/build.py > /dev/null
O=$C/build/variants/$NAME; mkdir -p $O $R/signal_amd/lib/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffast-math -fno-finite-math-only -Wno-unused-function -Wno-unknown-pragmas"
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if echo " $UNITS " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc $FLAGS $DEFS -c $f -o $O/$b.o &
    objs="$objs $O/$b.o"
  else
    objs="$objs $C/build/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/signal_amd/lib/variants/libsignal_$NAME.so $objs -ldl
echo signal_amd/lib/variants/libsignal_$NAME.so
