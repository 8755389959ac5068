#!/bin/bash
# rocprofv3 PMC passes (counters only) over the attention shapes bench.py reports: CrossAttention geometry N = 20,804 (forward +
# backward), masked MHA B = 32 / L = 512, and the in-step packed short-sequence mix.  Summarise with tools/pmc_attn.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in =xattn_N20804 =mha_L512 packed_mix; do
  rocprofv3 -i $R/tools/ubench/pmc_attn_in.txt --kernel-trace -d $R/gpurun_out/pmc_attn_r2/$c -o pmc --output-format csv -- $R/tools/ubench/attn_bench 1 1 0 $c > $R/gpurun_out/pmc_attn_r2_$c.log 2>&1
done
cd $R
python3 tools/pmc_attn.py gpurun_out/pmc_attn_r2 gpurun_out/r02_attn_pmc > /dev/null
grep -E "bf16|unsigned short|pipe|short" gpurun_out/r02_attn_pmc.csv | cut -c1-60 > /dev/null
python3 - <<PY
import json
d = json.load(open("gpurun_out/r02_attn_pmc.json"))
for k, v in d.items():
    if "float" not in k and "delta" not in k:
        print(f"{k:75s} mfma_util {v['mfma_util']:.3f} useful {v['mfma_util_useful']:.3f} valu/mfma {v['valu_per_mfma']:5.1f} wait {v['wait_any']:.2f} inst-stall {v['wait_inst']:.2f} active {v['active']:.2f}")
PY
