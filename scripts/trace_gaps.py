"""rocprofv3 --kernel-trace CSV -> where the wall time of a run goes: kernel time, and the idle gaps between
consecutive kernels of the stream bucketed by length (a gap above ~6 us is a host round trip: convergence check).
usage: python trace_gaps.py <kernel_trace.csv> [n_steps] [--between NAME] [--json]
--between NAME: only the kernels between the first and the last launch of a kernel whose name contains NAME (bench.py
--trace-markers brackets its timed steps with k_cfl launches); --json: one JSON object (the numbers bench.py quotes)"""
import json
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    argv = [a for a in sys.argv[2:] if not a.startswith("--")]
    steps = float(argv[0]) if argv else 1.0
    if "--between" in sys.argv:
        name = sys.argv[sys.argv.index("--between") + 1]
        argv = [a for a in argv if a != name]
        steps = float(argv[0]) if argv else 1.0
        marks = [k for k, r in enumerate(rows) if name in r[2]]
        if len(marks) >= 2:
            rows = rows[marks[0] + 1:marks[-1]]
    busy = sum(e - s for s, e, _ in rows)
    buckets = [(0, 1.0), (1.0, 2.0), (2.0, 3.0), (3.0, 6.0), (6.0, 12.0), (12.0, 25.0), (25.0, 60.0), (60.0, 1e9)]
    cnt = [0] * len(buckets)
    tot = [0.0] * len(buckets)
    after = {}
    for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
        g = (s1 - e0) / 1e3
        if g < 0:
            g = 0.0
        for k, (lo, hi) in enumerate(buckets):
            if lo <= g < hi:
                cnt[k] += 1
                tot[k] += g
        if 6.0 <= g < 60.0:
            a = after.setdefault(n0, [0, 0.0])
            a[0] += 1
            a[1] += g
    span = (rows[-1][1] - rows[0][0]) / 1e3
    if "--json" in sys.argv:
        small = [(e - s) / 1e3 for s, e, _ in rows if (e - s) < 15000]
        print(json.dumps({"steps": steps, "launches_per_step": len(rows) / steps, "kernel_ms_per_step": busy / 1e6 / steps,
                          "small_launches_per_step": len(small) / steps, "small_launch_ms_per_step": sum(small) / 1e3 / steps,
                          "span_ms_per_step": span / 1e3 / steps,
                          "host_round_trips_per_step": sum(cnt[4:7]) / steps,
                          "idle_ms_per_step_in_round_trips": sum(tot[4:7]) / 1e3 / steps,
                          "idle_ms_per_step_between_launches": sum(tot[:4]) / 1e3 / steps}))
        return
    print("kernels %d  span %.1f us  busy %.1f us (%.1f %%)  per step: span %.1f busy %.1f launches %.1f" % (
        len(rows), span, busy / 1e3, 100 * busy / 1e3 / span, span / steps, busy / 1e3 / steps, len(rows) / steps))
    for k, (lo, hi) in enumerate(buckets):
        print("gap %5.1f .. %7.1f us: %6d  total %9.1f us  per step %7.1f us (%.1f per step)" % (
            lo, hi, cnt[k], tot[k], tot[k] / steps, cnt[k] / steps))
    print("host round trips (6 .. 60 us) by the kernel before them:")
    for n, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-50s %5d  avg %5.1f us  per step %6.1f us" % (n[:50], c, t / c, t / steps))


if __name__ == "__main__":
    main()
