#!/bin/bash
# Cost attribution of the shade kernel on the GPU box: the bench workload with texture filtering / next-event estimation /
# BSDF sampling compiled out one at a time (builds with -DMI_PT_DIAG_NO_TEX etc. under vk_gltf_renderer_amd/lib/var_*; the
# images are wrong, only the time of k_shade<..., FIRST> is meaningful).  usage: tools/attribution.sh [bench args]
export TMPDIR=/tmp
for v in "" $(cd vk_gltf_renderer_amd/lib && ls -d var_* 2>/dev/null); do
  if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  out=$PWD/gpurun_out/attr_${v:-product}; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/log.txt 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "VARIANT ${v:-product}: $(grep 'k_shade<false, true, true>' $f | awk -F, '{printf "shade<FIRST> avg %.3f ms", $4/1e6}') $(grep 'k_shade<false, true, false>' $f | awk -F, '{printf " shade avg %.3f ms", $4/1e6}')"
done
