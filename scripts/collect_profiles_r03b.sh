#!/bin/bash
# Round-3 final profiles (after k_jac_lattice and the three-pass staging of k_cheb_lattice); run through gpurun from
# the repo root: scripts/collect_profiles_r03b.sh <tag>.  The 3D / DFG workloads did not change after r03_a.
TAG=${1:-r03_b}
REPO=$(pwd); O=$REPO/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$name -o b -- python3 $REPO/bench.py "$@" > $O/${TAG}_$name.json 2> $O/t_$name.err
  cp $(find $O/t_$name -name "*kernel_stats.csv" | head -1) $O/${TAG}_${name}_kernel_stats.csv
  python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/t_$name -name "*kernel_trace.csv" | head -1) 500 > $O/${TAG}_${name}_kernel_stats_by_grid.csv
  rm -rf $O/t_$name
  echo "trace $name done: $(tail -c 300 $O/${TAG}_$name.json | head -c 120)"
}
pmc() {  # name, bench args...
  local name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p_${name}_$C -o b -- python3 $REPO/bench.py "$@" > /dev/null 2> $O/p_${name}_$C.err
  done
  python3 $REPO/scripts/summarize_pmc.py $O/p_${name}_FETCH_SIZE $O/p_${name}_WRITE_SIZE > $O/${TAG}_${name}_pmc_fetch_write_size.json
  rm -rf $O/p_${name}_FETCH_SIZE $O/p_${name}_WRITE_SIZE
  echo "pmc $name done"
}
trace bench_n512 --steps 20 --warmup 5 --no-cpu-baseline
trace bench_n512_timed_steps_only --steps 40 --warmup 10 --timed-only
trace bench_n1024 --cells 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-solver-classes
export NSFEM_JAC_LATTICE=0
trace bench_n512_jacobian_launch_pair --steps 40 --warmup 10 --timed-only
unset NSFEM_JAC_LATTICE
pmc bench_n512 --steps 1 --warmup 1 --timed-only
pmc bench_n1024 --cells 1024 --steps 1 --warmup 1 --timed-only
cd $REPO
timeout -k 10 900 python3 bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
timeout -k 10 600 python3 bench.py --cells 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-solver-classes > $O/${TAG}_bench_n1024.json 2> $O/bench_1024.err
timeout -k 10 600 python3 bench.py --cells 336 --steps 50 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_n336.json 2> $O/bench_336.err
timeout -k 10 600 python3 bench.py --workload tgv3d-ipcs --cells 64 --steps 20 --warmup 3 > $O/${TAG}_bench_tgv3d_n64.json 2> $O/bench_tgv.err
timeout -k 10 600 python3 bench.py --workload channel3d-bdf --cells 48 --steps 10 --warmup 3 > $O/${TAG}_bench_channel3d_n48.json 2> $O/bench_ch48.err
timeout -k 10 600 python3 bench.py --workload dfg-bdf --steps 20 --warmup 3 > $O/${TAG}_bench_dfg.json 2> $O/bench_dfg.err
timeout -k 10 300 python3 scripts/gpu_lattice_sweep.py 512 off 0,24 0,32 > $O/${TAG}_lattice_kernel_launch_shapes_n512.txt 2>&1
timeout -k 10 300 python3 scripts/gpu_lattice_sweep.py 1024 off 0,24 0,32 > $O/${TAG}_lattice_kernel_launch_shapes_n1024.txt 2>&1
# the one-launch Jacobian action: knock-outs (NSFEM_JL_DBG: 1 no element kernel, 2 no L product, 4 no node sums, 7 all
# three, 127 + staging loads / stores off, 256 immediate return) and SQ counters
bash scripts/r03_jl_trace.sh ${TAG}_jl 512 0 1 2 4 7 127 256 > $O/${TAG}_jac_lattice_knockouts_n512.txt 2>&1
rm -rf $REPO/gpurun_out/${TAG}_jl
NSFEM_JL_DBG=0 bash scripts/r03_pmc_jl.sh ${TAG}_jl > $O/${TAG}_jac_lattice_sq_counters_n512.txt 2>&1
rm -rf $REPO/gpurun_out/pmc_${TAG}_jl_*
ls $O
