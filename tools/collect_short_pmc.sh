#!/bin/bash
# rocprofv3 PMC passes (counters only) over the in-step packed short-sequence attention mix: memory-side counters.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -i $R/tools/ubench/pmc_mem_in.txt --kernel-trace -d $R/gpurun_out/pmc_short_r3 -o pmc --output-format csv -- $R/tools/ubench/attn_bench 1 1 0 packed_mix > $R/gpurun_out/pmc_short_r3.log 2>&1
cd $R
python3 tools/pmc_sum_generic.py gpurun_out/pmc_short_r3 short > gpurun_out/pmc_short_r3.txt
cat gpurun_out/pmc_short_r3.txt
