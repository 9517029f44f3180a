#!/bin/bash
# same-box A/B of library builds on the kernel micro-benchmark (tools/bench_kernels.py), alternating rounds, then
# FETCH_SIZE / WRITE_SIZE of the GEMM kernels per build (separate PMC passes).
# usage: tools/ab_kernels.sh "<lib1.so> <lib2.so> ..." [pmc]      ('-' = the shipped library)
LIBS=$1; PMC=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for round in 1 2; do
  for lib in $LIBS; do
    if [ $lib = - ]; then unset SIGNAL_HIP_LIB; else export SIGNAL_HIP_LIB=$R/$lib; fi
    echo "== $lib round $round"
    timeout -k 10 300 python3 $R/tools/bench_kernels.py 2>/dev/null | grep -E "^(nt_|tn_|ln_|attn_)" | awk '{printf "%s %s  ", $1, $3} END {print ""}'
  done
done
if [ -n "$PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  for lib in $LIBS; do
    if [ $lib = - ]; then unset SIGNAL_HIP_LIB; else export SIGNAL_HIP_LIB=$R/$lib; fi
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/abk_pmc
      rocprofv3 --pmc $c --output-format csv -d /tmp/abk_pmc -- python3 $R/tools/bench_kernels.py > /tmp/abk_pmc.log 2>&1 || tail -3 /tmp/abk_pmc.log
      python3 - "$lib" $c <<'PY'
import csv, glob, sys, collections, re
acc, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob("/tmp/abk_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != sys.argv[2]: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:44] + " grid" + r.get("Grid_Size", "?")
        if "gemm" not in k: continue
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
mul = 2 if sys.argv[2] == "FETCH_SIZE" else 1      # gfx950: FETCH_SIZE reports half of a wide streaming read (KiB units)
for k in sorted(acc):
    print(f"{sys.argv[1]:28s} {sys.argv[2]:10s} {k:64s} {mul * acc[k] / cnt[k] * 1024 / 1e6:9.1f} MB/launch")
PY
    done
  done
fi
