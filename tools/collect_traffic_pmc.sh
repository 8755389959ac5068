#!/bin/bash
# rocprofv3 PMC passes for the HBM-traffic figures of bench.py (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters only,
# as the MI355X guide prescribes): the aggregation micro-benchmark (HBM regime) and the in-step launches.  Run on the GPU box;
# summarise with tools/pmc_traffic.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_r2_fetch -o f --output-format csv -- python3 $R/bench.py --micro-only --micro-select spmm > $R/gpurun_out/pmc_r2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_r2_write -o w --output-format csv -- python3 $R/bench.py --micro-only --micro-select spmm > $R/gpurun_out/pmc_r2_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_r2_sfetch -o f --output-format csv -- python3 $R/bench.py --no-micro --no-cpu-baseline --no-fp32-leg > $R/gpurun_out/pmc_r2_sfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_r2_swrite -o w --output-format csv -- python3 $R/bench.py --no-micro --no-cpu-baseline --no-fp32-leg > $R/gpurun_out/pmc_r2_swrite.log 2>&1
cd $R
find gpurun_out/pmc_r2_fetch gpurun_out/pmc_r2_write gpurun_out/pmc_r2_sfetch gpurun_out/pmc_r2_swrite -name "*counter_collection.csv" | head
