#!/bin/bash
# Round-3 GPU call C: FETCH_SIZE / WRITE_SIZE calibration, PMC passes of helmet + atrium (-> profiles/pmc_latest_*.json), rank-of-8 check, GPU tests.
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/r03c_gputest.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03c_gputest.txt)"
timeout 300 python -m pytest tests/test_gpu_lobes.py -m gpu -x -q -s -k converged 2>&1 | grep "converged parity\|passed\|failed"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o $O/calib_fetch || exit 1
$O/calib_fetch > $O/r03_calib_stdout.txt; cat $O/r03_calib_stdout.txt
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_calib_fetch -- $O/calib_fetch > /dev/null 2>&1 )
( cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_calib_write -- $O/calib_fetch > /dev/null 2>&1 )
python tools/calib_fetch_report.py $O/r03_calib_stdout.txt $O/r03_calib_fetch $O/r03_calib_write > $O/r03_fetch_calibration.json; python -c "
import json; j=json.load(open('$O/r03_fetch_calibration.json')); print('CALIB', j['factors'])"
for w in helmet atrium; do
  tools/profile.sh r03_$w --workload $w --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarize_pmc.py $O/prof_r03_$w $O/prof_r03_$w/summary.json > /dev/null
  python tools/make_pmc_latest.py $O/prof_r03_$w/summary.json $w 64 3 $O/r03_fetch_calibration.json > $O/pmc_latest_$w.json
  python -c "
import json; j=json.load(open('$O/pmc_latest_$w.json')); print('PMC $w', {k:(v['hbm_bytes_per_launch'], v['hbm_bytes_bounds'], v['avg_us']) for k,v in j['kernels'].items()})"
done
timeout 400 python tools/check_rank_of_8.py atrium 8 > $O/r03_rank_of_8_atrium.txt 2>&1; cat $O/r03_rank_of_8_atrium.txt | grep -v Warning
timeout 500 python tools/check_rank_of_8.py street 8 > $O/r03_rank_of_8_street.txt 2>&1; cat $O/r03_rank_of_8_street.txt | grep -v Warning
