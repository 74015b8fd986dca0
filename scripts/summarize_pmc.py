"""Per-kernel summary (count, median, max, in KB) of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
counter_collection CSVs: python summarize_pmc.py <fetch_dir> <write_dir> > summary.json
Round 4: additionally `by_grid`: the same per (kernel, grid size) with the MEAN -- a kernel symbol is launched on several
multigrid levels and in several launch shapes; the mean over the launches of one grid is what pairs with an average of
algorithmic bytes per launch (bench.py `roofline.traffic`)."""
import csv
import glob
import json
import os
import statistics
import sys


def summarize(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                per.setdefault(name, []).append(float(row["Counter_Value"]))
    return {k: {"n": len(v), "median_KB": statistics.median(v), "max_KB": max(v)}
            for k, v in per.items()}


def summarize_by_grid(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                grid = row.get("Grid_Size", row.get("Grid_Size_X", "0")) or "0"
                key = "%s @ grid %s" % (row["Kernel_Name"].split("(")[0], grid)
                per.setdefault(key, []).append(float(row["Counter_Value"]))
    return {k: {"n": len(v), "mean_KB": sum(v) / len(v), "median_KB": statistics.median(v), "max_KB": max(v)}
            for k, v in per.items() if sum(v) / len(v) >= 64.0}


if __name__ == "__main__":
    print(json.dumps({"fetch": summarize(sys.argv[1], "FETCH_SIZE"),
                      "write": summarize(sys.argv[2], "WRITE_SIZE"),
                      "by_grid": {"fetch": summarize_by_grid(sys.argv[1], "FETCH_SIZE"),
                                  "write": summarize_by_grid(sys.argv[2], "WRITE_SIZE")}}, indent=1))
