# One consistent set of measurements from one box: bench lines, per-kernel rocprofv3 statistics, HBM byte counters.
#   gpurun -- bash tools/profile_round.sh        (results under gpurun_out/prof/)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 200 python3 bench.py --workload c2 --rays 1048576 --steps 6 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 200 python3 bench.py --workload c4 --steps 50 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 400 python3 tools/shard_sizes.py $O/shard_sizes.json > $O/shard_sizes.log 2>&1
for f in bf16 mx; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$f -o c3 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --wgrad-operands $f > $O/bench_c3_profiled_$f.json 2> $O/stats_$f.err
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$f -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --wgrad-operands $f > /dev/null 2> $O/pmc_fetch_$f.err
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$f -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --wgrad-operands $f > /dev/null 2> $O/pmc_write_$f.err
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy_$f -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --wgrad-operands $f > /dev/null 2> $O/pmc_busy_$f.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -o c4 -- python3 bench.py --workload c4 --graph off --steps 25 --warmup 5 > $O/bench_c4_profiled.json 2> $O/stats_c4.err
find $O -name "*.csv" | head -40
# summaries (copy what is to be judged into profiles/ afterwards)
python3 tools/pmc_summary.py $O/pmc_traffic.json bf16 $(find $O/pmc_fetch_bf16 -name "*counter_collection.csv") $(find $O/pmc_write_bf16 -name "*counter_collection.csv") mx $(find $O/pmc_fetch_mx -name "*counter_collection.csv") $(find $O/pmc_write_mx -name "*counter_collection.csv") > $O/pmc_traffic.txt
for f in bf16 mx; do python3 tools/chain_counters.py $(find $O/pmc_busy_$f -name "*counter_collection.csv") $O/chain_counters_$f.json > $O/chain_counters_$f.txt; done
