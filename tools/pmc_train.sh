#!/bin/bash
# SQ counters of the train step's kernels, in the model (one rocprofv3 --pmc pass over bench.py, PMC only, no trace domains).
# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs): share of SIMD cycles, at the clock the chip
# held, in which the matrix pipe was busy.  usage: tools/pmc_train.sh <tag> [bench args]
TAG=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_train_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:46]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])[:16]
print(f"{'kernel':46s} launches mfma_util wait_any wait_inst  active lds_conflict")
for k, v in rows:
    n = cnt[(k, "SQ_WAVE_CYCLES")]; wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"] / 32.0 * 1024.0, 1.0)
    print(f"{k:46s} {n:8d} {util:9.3f} {v['SQ_WAIT_ANY'] / wc:8.2f} {v['SQ_WAIT_INST_ANY'] / wc:9.2f} {v['SQ_ACTIVE_INST_ANY'] / wc:7.2f} {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):12.3f}")
PY
