#!/bin/bash
# final state of round 2: GPU test suite, smoke, default bench line
cd /root/repo
O=gpurun_out/r02_d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 900 python bench.py > $O/r02_d_bench_default.json 2> $O/bench.err; python scripts/show_bench.py $O/r02_d_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_d/r02_d_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
