"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) statistics.

The multigrid launches one kernel symbol (e.g. k_spmv_stream<1,1,2,3>) on every level, so the
per-symbol average of `--stats` mixes 1,000-row and 1,000,000-row launches; this table separates
them (the finest-level launches are the rows with the largest grid).
usage: python summarize_trace_by_grid.py <kernel_trace.csv> [min_total_us] > by_grid.csv"""
import csv
import statistics
import sys


def main():
    groups = {}
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].split("(")[0]
            key = (name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
            groups.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    floor = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 0.0
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "GridSizeX", "WorkgroupSizeX", "Calls", "TotalDurationNs", "AverageNs", "MedianNs",
                "MinNs", "MaxNs"])
    for (name, grid, wg), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if sum(d) < floor:
            continue
        w.writerow([name, grid, wg, len(d), sum(d), "%.1f" % (sum(d) / len(d)), "%.1f" % statistics.median(d),
                    min(d), max(d)])


if __name__ == "__main__":
    main()
