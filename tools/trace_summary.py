#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV on the GPU box (the raw trace of a headline bench run is hundreds of
MB; only this summary travels back): per-kernel calls / total / average duration, the share of MFMA (igemm_*)
kernels, dispatch count, union busy time of the device and the time-weighted number of kernels in flight.

  python tools/trace_summary.py <kernel_trace.csv> <out.json> [<out_stats.csv>]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("cmoop::", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    src, out_json = sys.argv[1], sys.argv[2]
    out_csv = sys.argv[3] if len(sys.argv) > 3 else None
    per = defaultdict(lambda: [0, 0, 1 << 62, 0])      # calls, total ns, min, max
    events = []
    with open(src, newline="") as f:
        rd = csv.DictReader(f)
        for row in rd:
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            nm = short(row["Kernel_Name"])
            p = per[nm]
            d = e - s
            p[0] += 1
            p[1] += d
            p[2] = min(p[2], d)
            p[3] = max(p[3], d)
            mf = 1 if nm.startswith(("igemm_", "halo_")) else 0
            events.append((s, 1, mf))
            events.append((e, -1, -mf))
    events.sort()
    busy = mfma_busy = 0
    conc_time = defaultdict(int)
    cur = cur_mf = 0
    last = events[0][0] if events else 0
    for t, d, m in events:
        dt = t - last
        if dt > 0:
            if cur > 0:
                busy += dt
            if cur_mf > 0:
                mfma_busy += dt
            conc_time[min(cur, 16)] += dt
        cur += d
        cur_mf += m
        last = t
    span = (events[-1][0] - events[0][0]) if events else 0
    tot = sum(p[1] for p in per.values())
    mfma_tot = sum(p[1] for k, p in per.items() if k.startswith(("igemm_", "halo_")))
    rows = sorted(per.items(), key=lambda kv: -kv[1][1])
    summary = {
        "dispatches": sum(p[0] for p in per.values()),
        "span_ms": span / 1e6, "device_busy_ms": busy / 1e6, "device_busy_frac_of_span": busy / span if span else None,
        "some_mfma_kernel_running_frac_of_span": mfma_busy / span if span else None,
        "sum_kernel_durations_ms": tot / 1e6, "mean_kernels_in_flight_while_busy": tot / busy if busy else None,
        "mfma_share_of_kernel_time": mfma_tot / tot if tot else None,
        "non_mfma_share_of_kernel_time": 1 - mfma_tot / tot if tot else None,
        "time_by_kernels_in_flight_frac": {str(k): v / span for k, v in sorted(conc_time.items())} if span else {},
        "kernels": [{"name": k, "calls": p[0], "total_ms": p[1] / 1e6, "avg_us": p[1] / p[0] / 1e3, "pct": 100.0 * p[1] / tot,
                     "min_us": p[2] / 1e3, "max_us": p[3] / 1e3} for k, p in rows],
    }
    json.dump(summary, open(out_json, "w"), indent=1)
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for k, p in rows:
                w.writerow([k, p[0], p[1], p[1] / p[0], 100.0 * p[1] / tot, p[2], p[3]])
    print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}, indent=1))
    for r in summary["kernels"][:25]:
        print(f"{r['pct']:6.2f}% {r['calls']:8d} x {r['avg_us']:9.1f} us  {r['name']}")


if __name__ == "__main__":
    main()
