#!/bin/bash
# All rocprofv3 evidence of one round, on the GPU box: usage  tools/collect_profiles.sh r03   (outputs under gpurun_out/<tag>_*;
# copy the summaries into profiles/).  Counter passes carry no trace domain besides --kernel-trace, FETCH_SIZE and WRITE_SIZE
# run in separate passes (MI355X guide), and the profiled program follows `--` directly.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="--no-micro --no-cpu-baseline --no-fp32-leg"
# 1. kernel trace + stats of the default bench command (the roofline's in-step launches) and the per-step breakdown
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o b --output-format csv -- python3 $R/bench.py $B --steps 5 > $O/${TAG}_bench_traced.json 2> $O/${TAG}_trace.log
python3 $R/tools/step_breakdown.py $O/${TAG}_trace/b_kernel_trace.csv 5 60 > $O/${TAG}_step_breakdown.txt
cp $O/${TAG}_trace/b_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
echo "trace done"
# 2. HBM traffic of the aggregation kernels (micro = HBM regime, and in-step)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_fetch -o f --output-format csv -- python3 $R/bench.py --micro-only --micro-select spmm > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_write -o w --output-format csv -- python3 $R/bench.py --micro-only --micro-select spmm > $O/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_sfetch -o f --output-format csv -- python3 $R/bench.py $B > $O/${TAG}_pmc_sfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_swrite -o w --output-format csv -- python3 $R/bench.py $B > $O/${TAG}_pmc_swrite.log 2>&1
echo "traffic done"
# 3. MFMA / issue counters of the attention kernels: CrossAttention geometry at N = 20,804 and N = 5,201, masked MHA L = 512, the
#    in-step packed short-sequence mix; masked MHA at L = 2,048 (the MFMA-bound regime of the d = 64 kernel)
for c in =xattn_N20804 =xattn_N5201 =mha_L512 =mha_L2048_masked packed_mix; do
  rocprofv3 -i $R/tools/ubench/pmc_attn_in.txt --kernel-trace -d $O/${TAG}_pmc_attn/$c -o pmc --output-format csv -- $R/tools/ubench/attn_bench 1 1 0 $c > $O/${TAG}_pmc_attn_$c.log 2>&1
done
cd $R
python3 tools/pmc_attn.py gpurun_out/${TAG}_pmc_attn gpurun_out/${TAG}_attn_pmc > /dev/null
echo "attention counters done"
find gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_sfetch gpurun_out/${TAG}_pmc_swrite -name "*counter_collection.csv"
