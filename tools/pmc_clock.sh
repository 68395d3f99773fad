#!/bin/bash
# Effective clock and MFMA-pipe occupancy of the heavy GEMM kernels, one rocprofv3 --pmc pass each (no trace domains):
# clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES
# / (SIMDs x cycles).   bash tools/pmc_clock.sh <outdir under gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmc_clock}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for shape in "64,101,40,64,64,5" "64,51,20,128,128,5" "64,26,10,256,256,5" "64,101,40,32,32,5"; do
  tag=$(echo $shape | tr ',' '_')
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/raw/$tag" -o p -- python3 "$ROOT/tools/pmc_traffic.py" $shape > "$OUT/run_$tag.txt" 2> "$OUT/err_$tag.txt" || echo "pass $tag failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys, os
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
cols = None
for f in glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True):
    shape = os.path.relpath(f, os.path.join(out, "raw")).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        cols = list(r.keys())
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cmoop::", "")
        if "igemm" not in k and "halo_" not in k:
            continue
        key = (k, shape, r.get("Dispatch_Id", ""))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c in ("Start_Timestamp", "End_Timestamp"):
            if c in r and r[c]:
                rows[key][c] = [float(r[c])]
print("columns:", cols)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, shape, _), d in rows.items():
    for c, v in d.items():
        agg[(k, shape)][c].append(sum(v) / len(v))
res = []
for (k, shape), d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    row = {"kernel": k, "shape": shape.replace("_", ","), "launches": len(d.get("GRBM_GUI_ACTIVE", [])), **{c: m[c] for c in m if not c.endswith("Timestamp")}}
    if "Start_Timestamp" in m and "End_Timestamp" in m:
        dur_ns = sum(e - s for s, e in zip(d["Start_Timestamp"], d["End_Timestamp"])) / len(d["Start_Timestamp"])
        row["duration_us_under_pmc"] = dur_ns / 1e3
        if "GRBM_GUI_ACTIVE" in m and dur_ns > 0:
            row["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8.0 / dur_ns
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                row["mfma_busy_frac_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    res.append(row)
    print(json.dumps(row))
json.dump(res, open(os.path.join(out, "pmc_clock_summary.json"), "w"), indent=1)
PY
rm -rf "$OUT/raw"
