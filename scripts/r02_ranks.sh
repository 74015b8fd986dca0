#!/bin/bash
cd /root/repo
O=gpurun_out/r02t; mkdir -p $O
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -20 $O/$name.err; exit 1; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run dfg_r3_1 --workload dfg-bdf --dfg-refine 3 --steps 5 --warmup 2 --no-cpu-baseline
run dfg_r3_2 --workload dfg-bdf --dfg-refine 3 --steps 5 --warmup 2 --local-ranks 2
run dfg_r3_4 --workload dfg-bdf --dfg-refine 3 --steps 5 --warmup 2 --local-ranks 4
python - <<'PY'
import json
for n in ("dfg_r3_1","dfg_r3_2","dfg_r3_4"):
    c=json.load(open("gpurun_out/r02t/%s.json"%n))["config"]
    print(n, c["drag_lift_reference_formula"], c["net_boundary_mass_flux"], c["cylinder_perimeter_of_the_mesh"], c["host_setup_s"], c["comm_per_step_rank0"])
PY
