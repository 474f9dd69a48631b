#!/bin/bash
# Cost attribution of the shade kernels on the GPU box: a bench workload with texture filtering / next-event estimation /
# BSDF sampling compiled out one at a time (builds with -DMI_PT_DIAG_NO_TEX etc. under vk_gltf_renderer_amd/lib/var_*; the
# images are wrong, only the kernel times are meaningful).  usage: tools/attribution.sh <workload> [bench args]
export TMPDIR=/tmp
w=${1:-helmet}; shift
for v in "" $(cd vk_gltf_renderer_amd/lib && ls -d var_* 2>/dev/null); do
  unset MI_PT_LIB
  if [ -n "$v" ]; then export MI_PT_LIB=$PWD/vk_gltf_renderer_amd/lib/$v/libmi_pt.so; fi
  out=$PWD/gpurun_out/attr_${w}_${v:-product}; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout -k 5 100 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 1 --warmup 1 --frames-per-step 64 --no-cpu-baseline --also none "$@" > $out/log.txt 2>&1)
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "VARIANT ${v:-product} ($w): $(grep -E 'k_shade<false|k_trace_primary<(true|false), false>' $f | awk -F, '{n=$1; gsub(/.*k_/,"k_",n); gsub(/\(.*/,"",n); printf "%s avg %.3f ms x %d | ", n, $4/1e6, $2}')"
done
