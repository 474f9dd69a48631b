#!/bin/bash
# Profiles one bench.py configuration on the GPU box: kernel-trace stats + separate PMC passes (one counter group per run: rocprofv3
# refuses or distorts mixed groups, and FETCH_SIZE / WRITE_SIZE must be alone -- MI355X_MICROARCH.md).
# usage: [PASSES="stats sq sq2 tcc fetch write"] tools/profile.sh <tag> <bench args...>     (outputs under gpurun_out/prof_<tag>/)
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  ( cd /tmp && timeout ${PROFILE_TIMEOUT:-200} rocprofv3 "$@" --output-format csv -d "$out/$name" -- python "$OLDPWD/bench.py" "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 )
}
BENCH_ARGS=("$@" --no-cpu-baseline --also none)
for p in ${PASSES:-stats sq sq2 tcc fetch write}; do
  case $p in
    stats) run stats --kernel-trace --stats ;;
    sq)    run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ;;
    sq2)   run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM ;;
    tcc)   run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum ;;
    fetch) run pmc_fetch --pmc FETCH_SIZE ;;
    write) run pmc_write --pmc WRITE_SIZE ;;
  esac
done
find "$out" -name "*.csv" | head -30
