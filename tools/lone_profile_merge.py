#!/usr/bin/env python3
"""Join the lone-candidate bench line (per-instantiation launches and algorithmic GFLOP per launch, HIP-event durations)
with the rocprofv3 kernel trace summary of the SAME command: per MFMA instantiation, calls / average duration from the
trace, TFLOP/s = launches x GFLOP per launch / trace time, fraction of the fp32 MFMA peak -- every figure recomputable.

  python tools/lone_profile_merge.py <bench_line_under_rocprofv3.json> <trace_summary.json> [<bench_line_unprofiled.json>]
"""
import json
import sys

PEAK = 157.3


def load_line(path):
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {path}")


def main():
    line = load_line(sys.argv[1])
    trace = json.load(open(sys.argv[2]))
    plain = load_line(sys.argv[3]) if len(sys.argv) > 3 else None
    tk = {k["name"]: k for k in trace["kernels"]}
    lone = line["roofline"]["lone_candidate"]
    plain_k = {k["kernel"]: k for k in (plain["roofline"]["lone_candidate"]["per_kernel"] if plain else [])}
    print(f"lone candidate {tuple(lone['gene'])}: {lone['train_steps']} timed steps (+3 warm-up steps in the trace); "
          f"trace: {trace['dispatches']} dispatches, MFMA share of kernel time {trace['mfma_share_of_kernel_time']:.3f}")
    print(f"{'instantiation':46s} {'GFLOP/launch':>12s} {'trace calls':>11s} {'trace avg us':>12s} {'trace TF/s':>10s} {'frac':>6s} | "
          f"{'events avg us (profiled)':>24s} {'events avg us (plain)':>21s} {'plain TF/s':>10s}")
    tot_fl = tot_ns = 0.0
    for k in sorted(lone["per_kernel"], key=lambda k: -k["gflop_per_launch"] * k["launches"]):
        t = tk.get(k["kernel"])
        if not t:
            continue
        tf = k["gflop_per_launch"] / (t["avg_us"] * 1e-6) / 1e3
        p = plain_k.get(k["kernel"])
        tot_fl += k["gflop_per_launch"] * t["calls"]
        tot_ns += t["total_ms"]
        print(f"{k['kernel']:46s} {k['gflop_per_launch']:12.3f} {t['calls']:11d} {t['avg_us']:12.1f} {tf:10.1f} {tf / PEAK:6.3f} | "
              f"{k['avg_ms'] * 1e3:24.1f} {(p['avg_ms'] * 1e3 if p else float('nan')):21.1f} {(p['tflops'] if p else float('nan')):10.1f}")
    # GFLOP / ms = TFLOP/s
    print(f"all MFMA instantiations together: {tot_fl / tot_ns:.1f} TFLOP/s over their summed trace time = {tot_fl / tot_ns / PEAK:.3f} of peak")
    print("non-MFMA kernels by trace time:")
    for k in trace["kernels"]:
        if not k["name"].startswith(("igemm_", "halo_")) and k["pct"] >= 0.3:
            print(f"   {k['pct']:5.2f} %  {k['calls']:6d} x {k['avg_us']:8.1f} us  {k['name']}")


if __name__ == "__main__":
    main()
