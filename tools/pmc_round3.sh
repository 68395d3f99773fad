#!/bin/bash
# PMC passes of round 3 (each counter group in its own rocprofv3 --pmc run, no trace domains):
#  * L2 hit rate (TCC_HIT_sum / TCC_MISS_sum) + fabric traffic (FETCH_SIZE, WRITE_SIZE) of the weight-gradient kernels whose
#    fabric-side reads are 6x algorithmic (wgrad<64,128> on 64->64 k5 @101x40, wgrad<128,128> on 128->128 k5 @51x20);
#  * the front end's logmel_kernel: duration, FETCH / WRITE, VALU vs wait cycles (is it VALU-bound as DESIGN says?).
#   bash tools/pmc_round3.sh <outdir under gpurun_out>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/${1:-pmc_r3}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters, program args...
  local name="$1" ctrs="$2"; shift 2
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/raw/$name" -o p -- python3 "$@" > /dev/null 2> "$OUT/raw_$name.err" || echo "pass $name failed" | tee -a "$OUT/failed.txt"
}
for shape in "64,101,40,64,64,5" "64,51,20,128,128,5"; do
  tag=$(echo $shape | tr ',' '_')
  pass "hit_$tag" "TCC_HIT_sum TCC_MISS_sum" "$ROOT/tools/pmc_traffic.py" $shape
  pass "req_$tag" "TCC_REQ_sum TCC_READ_sum" "$ROOT/tools/pmc_traffic.py" $shape
  pass "fetch_$tag" "FETCH_SIZE" "$ROOT/tools/pmc_traffic.py" $shape
  pass "write_$tag" "WRITE_SIZE" "$ROOT/tools/pmc_traffic.py" $shape
  pass "ea_$tag" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "$ROOT/tools/pmc_traffic.py" $shape
done
pass "fe_sq1" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "$ROOT/tools/frontend_pmc.py"
pass "fe_sq2" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "$ROOT/tools/frontend_pmc.py"
pass "fe_fetch" "FETCH_SIZE" "$ROOT/tools/frontend_pmc.py"
pass "fe_write" "WRITE_SIZE" "$ROOT/tools/frontend_pmc.py"
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True):
    passname = os.path.relpath(f, os.path.join(out, "raw")).split(os.sep)[0]
    shape = passname.split("_", 1)[1] if "_" in passname else ""
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cmoop::", "")
        if not ("igemm" in k or "halo_" in k or "logmel" in k):
            continue
        key = (k, r["Grid_Size"], shape if ("igemm" in k or "halo_" in k) else "")
        a = agg[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
res = []
for (k, grid, shape), d in sorted(agg.items()):
    row = {"kernel": k, "grid_threads": int(grid), "shape": shape.replace("_", ","), "counters": {c: v / max(n, 1) for c, (v, n) in sorted(d.items())},
           "launches_averaged": max(n for _, n in d.values())}
    c = row["counters"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c:
        row["fabric_read_MB_corrected"] = 2 * c["FETCH_SIZE"] / 1024.0     # FETCH_SIZE is in KB and reads half on gfx950 (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in c:
        row["fabric_write_MB"] = c["WRITE_SIZE"] / 1024.0
    res.append(row)
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for r in res:
    print(json.dumps(r))
PY
rm -rf "$OUT/raw"
