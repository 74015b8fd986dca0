#!/bin/bash
# scripts/pmc_kernel.sh "<counters>" <tag> <python script> [args]: one rocprofv3 --pmc pass, then the
# per-kernel averages of the counters for kernels matching $KERNEL (default: k_spmv_stream)
set -e
CTR="$1"; TAG="$2"; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o $TAG -- python3 $REPO/"$@" > $OUT.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "${KERNEL:-k_spmv_stream}" in k and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= ${MINGRID:-1000000}:
        acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 3) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
