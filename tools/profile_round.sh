cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o c3 -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/prof_bench.json 2> gpurun_out/prof_stats.err && \
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_fetch.err && \
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_write.err && \
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_busy -o c3 -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_busy.err && \
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_busy -name "*.csv" | head -20
