#!/bin/bash
# HBM-side traffic of the MFMA kernels on one layer shape: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, no trace
# domains).  usage: tools/pmc_traffic.sh "64,51,20,128,128,5"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${1:-64,51,20,128,128,5}
# one counter per pass: FETCH_SIZE + WRITE_SIZE together "exceeds the capabilities of the hardware to collect"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic/$c -o p -- python3 tools/pmc_traffic.py $SHAPE > /dev/null 2>&1 || echo "pass $c failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_traffic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "igemm" not in k and "halo_" not in k and "reduce_slices" not in k and "splitk" not in k: continue
        a = agg[(k, r["Grid_Size"])][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("kernel,grid_threads,launches,FETCH_SIZE_KB,WRITE_SIZE_KB   (shape $SHAPE; read bytes = 2 x FETCH_SIZE on gfx950, MI355X_MICROARCH.md)")
for (k, g), d in sorted(agg.items()):
    f, w = d.get("FETCH_SIZE", [0, 1]), d.get("WRITE_SIZE", [0, 1])
    print(f"{k},{g},{f[1]},{f[0] / max(f[1], 1):.1f},{w[0] / max(w[1], 1):.1f}")
PY
rm -rf gpurun_out/pmc_traffic
