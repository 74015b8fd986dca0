#!/bin/bash
# kernel sequence of the LAST time step of a short timed-only run: scripts/r03_step_sequence.sh <tag>
REPO=$(pwd); TAG=$1; O=$REPO/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --steps 3 --warmup 6 --timed-only > $O/bench.json 2> $O/trace.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts with k_set_bc_residual? find the last occurrence of the first kernel of a step: k_conv_cell<0, 0> preceded by ... use k_spmv<2, 1, 1, 4 (grad) markers
names = [r["Kernel_Name"].split("(")[0].replace("void nsfem::", "").replace("nsfem::", "") for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith("k_spmv<2, 1, 1, 4, 0>")]   # one per step (pressure gradient in the momentum rhs)
start = marks[-1] if marks else 0
# walk back to the previous grad-type marker end to include the beginning of the step
out = open("$O/sequence.txt", "w")
prev_end = None
for r, n in list(zip(rows, names))[start - 3:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.write("%-42s grid %8s  %7.1f us  gap %6.1f us\n" % (n[:42], r["Grid_Size_X"], (e - s) / 1e3, gap))
    prev_end = e
out.close()
PY
rm -rf $O/trace
wc -l $O/sequence.txt
