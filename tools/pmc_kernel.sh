#!/bin/bash
# PMC counters of the isolated GEMM kernels (tools/kernel_bench.py on one layer shape), one rocprofv3 --pmc pass per
# counter group (never combined with trace domains).  usage: tools/pmc_kernel.sh "128,128,5" out_prefix
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${1:-128,128,5}; OUT=${2:-gpurun_out/pmc}
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d ${OUT}_$i -o p -- python3 tools/kernel_bench.py $SHAPE > /dev/null 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("${OUT}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "igemm" not in k and "halo_" not in k: continue
        key = (k, r["Grid_Size"])
        a = agg[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(agg):
    print(key[0], "grid", key[1])
    for c, (v, n) in sorted(agg[key].items()):
        print(f"    {c:28s} {v / n:16.1f}  (avg of {n} launches)")
PY
rm -rf ${OUT}_*
