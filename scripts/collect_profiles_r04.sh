#!/bin/bash
# Round-4 profiles; run through gpurun from the repo root: scripts/collect_profiles_r04.sh [part ...]
#   parts: trace pmc sq default configs  (default: all).  Everything lands in gpurun_out/r04/ under the names it
#   keeps in profiles/.
REPO=$(pwd); O=$REPO/gpurun_out/r04
mkdir -p $O
PARTS=${@:-trace pmc sq default configs}
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
trace() {  # name, n_steps_in_trace, bench args...
  local name=$1 steps=$2; shift 2
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$name -o b -- python3 $REPO/bench.py "$@" > $O/r04_$name.json 2> $O/t_$name.err
  local T=$(find $O/t_$name -name "*kernel_trace.csv" | head -1)
  cp $(find $O/t_$name -name "*kernel_stats.csv" | head -1) $O/r04_${name}_kernel_stats.csv
  python3 $REPO/scripts/summarize_trace_by_grid.py $T 100 > $O/r04_${name}_kernel_stats_by_grid.csv
  python3 $REPO/scripts/trace_gaps.py $T $steps --between k_cfl > $O/r04_${name}_gaps.txt
  python3 $REPO/scripts/trace_gaps.py $T $steps --between k_cfl --json > $O/r04_${name}_trace_summary.json
  rm -rf $O/t_$name
  echo "trace $name: $(cat $O/r04_${name}_trace_summary.json)"
}
pmc() {  # name, bench args...
  local name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_${name}_$C -o b -- python3 $REPO/bench.py "$@" > /dev/null 2> $O/p_${name}_$C.err
  done
  python3 $REPO/scripts/summarize_pmc.py $O/p_${name}_FETCH_SIZE $O/p_${name}_WRITE_SIZE > $O/r04_${name}_pmc_fetch_write_size.json
  rm -rf $O/p_${name}_FETCH_SIZE $O/p_${name}_WRITE_SIZE
  echo "pmc $name done"
}
if has trace; then
  trace bench_n512_timed_steps 40 --steps 40 --warmup 10 --timed-only --trace-markers
  trace bench_n1024_timed_steps 10 --cells 1024 --steps 10 --warmup 3 --timed-only --trace-markers
  trace bench_n512_timed_steps_multigrid_cg_poisson 40 --steps 40 --warmup 10 --timed-only --trace-markers --poisson-solver mg
fi
if has pmc; then
  pmc bench_n512 --steps 2 --warmup 1 --timed-only
fi
cd $REPO
if has sq; then
  # SQ counters of the finest-level k_cheb_lattice launches (three passes of eight counters), summed up as one JSON
  KERNEL=k_cheb_lattice MINGRID=800000 bash scripts/r03_pmc_lattice.sh r04_sq > $O/r04_cheb_lattice_sq_counters_n512.txt 2>&1
  python3 - <<PY
import ast, json, re
vals = {}
for line in open("$O/r04_cheb_lattice_sq_counters_n512.txt"):
    m = re.search(r"(\{.*\}) n (\d+)", line)
    if m and "k_cheb_lattice<2, 3, 4" in line:
        vals.update(ast.literal_eval(m.group(1)))
json.dump(vals, open("$O/r04_cheb_lattice_sq_counters_n512.json", "w"), indent=1)
print(vals)
PY
  rm -rf $REPO/gpurun_out/pmc_r04_sq_*
fi
if has default; then
  timeout -k 10 1100 python3 bench.py > $O/r04_bench_default.json 2> $O/bench_default.err
  python3 scripts/show_bench.py $O/r04_bench_default.json
fi
if has configs; then
  timeout -k 10 600 python3 bench.py --cells 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-solver-classes --no-other-configs > $O/r04_bench_n1024.json 2> $O/bench_1024.err
  timeout -k 10 600 python3 bench.py --cells 333 --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > $O/r04_bench_n333.json 2> $O/bench_333.err
  timeout -k 10 600 python3 bench.py --workload tgv3d-ipcs --cells 64 --steps 20 --warmup 3 > $O/r04_bench_tgv3d_n64.json 2> $O/bench_tgv.err
  timeout -k 10 600 python3 bench.py --workload channel3d-bdf --cells 48 --steps 10 --warmup 3 > $O/r04_bench_channel3d_n48.json 2> $O/bench_ch48.err
  timeout -k 10 900 python3 bench.py --workload channel3d-bdf --cells 64 --steps 10 --warmup 3 > $O/r04_bench_channel3d_n64.json 2> $O/bench_ch64.err
  timeout -k 10 600 python3 bench.py --workload dfg-bdf --steps 20 --warmup 3 > $O/r04_bench_dfg.json 2> $O/bench_dfg.err
  # functional rehearsal of the N-rank paths on ONE GPU (thread ranks, in-process communicator): message counts and the
  # kernel set of a strong-scaling step on 8 strips -- NOT a performance figure
  timeout -k 10 600 python3 bench.py --local-ranks 8 --scaling strong --cells 960 --steps 4 --warmup 2 --timed-only > $O/r04_bench_strong_960_thread_ranks_8.json 2> $O/bench_lr8.err
  NSFEM_SETUP_PROFILE=1 NSFEM_DEBUG_SETUP=1 timeout -k 10 600 python3 scripts/r03_setup_profile.py 64 > $O/r04_setup_end_of_round.txt 2>&1      # (profiles/r04_setup.txt = this + the start-of-round run)
  for f in n1024 n333 tgv3d_n64 channel3d_n48 channel3d_n64 dfg strong_960_thread_ranks_8; do python3 scripts/show_bench.py $O/r04_bench_$f.json; done
fi
ls $O
