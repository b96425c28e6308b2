#!/bin/bash
# usage: scripts/isa_stats.sh <model> : compile plugin with -save-temps in /tmp/sbm and print per-kernel stats
set -e
M=${1:-cascade20}
mkdir -p /tmp/sbm && cd /tmp/sbm && rm -f *.s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSBM_MODEL_HEADER="\"/root/repo/sysbio_modeling_amd/csrc/models/$M.hpp\"" /root/repo/sysbio_modeling_amd/csrc/sbm_plugin_main.hip -o /tmp/sbm/$M.so -save-temps 2>&1 | grep -E "error|warning" | head
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|vgpr_spill_count|name|private_segment_fixed_size):" *gfx950*.s
for k in $(grep -E "^\s+\.name:" *gfx950*.s | awk '{print $2}'); do
  awk "/^$k:/,/s_endpgm/" *gfx950*.s > k.s
  echo "== $k: $(grep -cE '^\s+[a-z_0-9]+ ' k.s) instrs; fp64: $(grep -cE 'v_(fma|fmac|mul|add|rcp|max|min)_f64' k.s) accvgpr: $(grep -c accvgpr k.s) cndmask: $(grep -c v_cndmask k.s) scratch: $(grep -c scratch_ k.s) saveexec: $(grep -c saveexec k.s)"
done
