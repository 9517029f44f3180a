#!/bin/bash
# SQ counters for the GEMM kernels of tools/bench_kernels.py (PMC pass only, no trace domains).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmc_gemm_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $R/tools/bench_kernels.py > $OUT/log.txt 2>&1 || tail -5 $OUT/log.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        if "gemm" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in sorted(acc.items()):
    n=cnt[(k,"SQ_WAVE_CYCLES")]
    wc=v["SQ_WAVE_CYCLES"]
    print(f"{k:42s} launches {n:4d} wave_cyc/launch {wc/n:12.0f} wait_any {v['SQ_WAIT_ANY']/wc:5.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:5.2f} active {v['SQ_ACTIVE_INST_ANY']/wc:5.2f} mfma_busy/launch {v['SQ_VALU_MFMA_BUSY_CYCLES']/n:12.0f} lds_conf/lds_active {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):5.3f} busy_cyc/launch {v['SQ_BUSY_CYCLES']/n:10.0f}")
PY
