"""Kernel sequence of ONE time step from a rocprofv3 kernel trace (csv): the launches between the last two marker
launches of `--between` (bench.py --trace-markers brackets the timed steps; with --steps 1 that is one step).
usage: trace_sequence.py kernel_trace.csv [--between k_cfl]"""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[sys.argv.index("--between") + 1] if "--between" in sys.argv else "k_cfl"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end = None
for r in rows[a + 1:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = en
    print("%7.1f us  gap %6.1f  grid %9s  %s" % ((en - st) / 1e3, gap, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"),
                                                  r["Kernel_Name"][:70]))
print("launches", b - a - 1)
