"""Per-kernel summary (count, median, max, in KB) of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
counter_collection CSVs: python summarize_pmc.py <fetch_dir> <write_dir> > summary.json"""
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


if __name__ == "__main__":
    print(json.dumps({"fetch": summarize(sys.argv[1], "FETCH_SIZE"),
                      "write": summarize(sys.argv[2], "WRITE_SIZE")}, indent=1))
