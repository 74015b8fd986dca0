#!/bin/bash
# Round-2 judged profiles (run through gpurun from the repo root): scripts/collect_profiles_r02.sh <tag>
#  1. rocprofv3 --kernel-trace --stats of the default bench command (+ split by grid size)
#  2. the same for --cells 1024 (working set 1.0 GB >> 256 MiB Infinity Cache)
#  3. separate --pmc FETCH_SIZE / WRITE_SIZE passes of the default workload and of tgv3d n=64
#  4. traces of the 3D workloads (tgv3d-ipcs n=64, channel3d-bdf n=48)
#  5. flush-interleaved (cache-cold) smoother figures
TAG=${1:-r02_x}
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
trace bench_n1024 --cells 1024 --steps 5 --warmup 2 --no-cpu-baseline
trace tgv3d_n64 --workload tgv3d-ipcs --cells 64 --steps 5 --warmup 2
trace channel3d_n48 --workload channel3d-bdf --cells 48 --steps 3 --warmup 2
# the same launches without the stencil dictionaries (CSR-stream / SELL-64 kernels of round 1 / early round 2)
export NSFEM_DICT=0
trace bench_n512_csr_kernels --steps 20 --warmup 5 --no-cpu-baseline
trace tgv3d_n64_csr_kernels --workload tgv3d-ipcs --cells 64 --steps 5 --warmup 2
unset NSFEM_DICT
pmc bench_n512 --steps 1 --warmup 1 --timed-only
pmc tgv3d_n64 --workload tgv3d-ipcs --cells 64 --steps 1 --warmup 1
cd $REPO
# plain bench lines (no profiler attached)
timeout -k 10 900 python3 bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
timeout -k 10 600 python3 bench.py --workload channel3d-bdf --cells 64 --steps 10 --warmup 3 > $O/${TAG}_bench_channel3d_n64.json 2> $O/bench_ch64.err
timeout -k 10 600 python3 bench.py --workload channel3d-bdf --cells 48 --steps 10 --warmup 3 > $O/${TAG}_bench_channel3d_n48.json 2> $O/bench_ch48.err
timeout -k 10 600 python3 bench.py --workload tgv3d-ipcs --cells 64 --steps 20 --warmup 3 > $O/${TAG}_bench_tgv3d_n64.json 2> $O/bench_tgv.err
timeout -k 10 600 python3 bench.py --workload dfg-bdf --steps 20 --warmup 3 > $O/${TAG}_bench_dfg.json 2> $O/bench_dfg.err
timeout -k 10 600 python3 bench.py --cells 1024 --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_n1024.json 2> $O/bench_1024.err
timeout -k 10 300 python3 scripts/gpu_smoother_2d.py > $O/${TAG}_smoother_2d_cold_cache.txt 2>&1
timeout -k 10 300 python3 scripts/gpu_sell_tune.py 3 64 parity > $O/${TAG}_smoother_3d_cold_cache.txt 2>&1
NSFEM_DICT=0 timeout -k 10 300 python3 scripts/gpu_sell_tune.py 3 64 parity >> $O/${TAG}_smoother_3d_cold_cache.txt 2>&1
NSFEM_DICT=0 NSFEM_SELL=0 timeout -k 10 300 python3 scripts/gpu_sell_tune.py 3 64 lex >> $O/${TAG}_smoother_3d_cold_cache.txt 2>&1
timeout -k 10 300 python3 scripts/gpu_sell_tune.py 2 512 lex >> $O/${TAG}_smoother_2d_cold_cache.txt 2>&1
NSFEM_DICT=0 timeout -k 10 300 python3 scripts/gpu_sell_tune.py 2 512 lex >> $O/${TAG}_smoother_2d_cold_cache.txt 2>&1
timeout -k 10 300 python3 scripts/gpu_dict_probe.py > $O/${TAG}_dictionary_sizes.txt 2>&1
ls $O
