#!/bin/bash
# round 3, run 4 (one box): wide epilogue A/B (KX_DBG bit 16384 = the dword form), GPU suite, reduced-precision report
cd $GRAFT_REPO_ROOT
tools/ab_env.sh r03e "KX_DBG=0" "KX_DBG=16384" || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03e_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r03e_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/reduced_precision_report.py > gpurun_out/r03e_reduced.txt 2> gpurun_out/r03e_reduced.err || { tail gpurun_out/r03e_reduced.err; exit 1; }
grep "audio\|gen.stage.1" gpurun_out/r03e_reduced.txt
KOKOROX_CONV=f16 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-utts 0 --free-run 0 --pcie 0 --serve 0 --reduced 0 > gpurun_out/r03e_bench_f16.json 2> gpurun_out/r03e_bench_f16.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r03e_bench_f16.json').read().strip().splitlines()[-1]); print('KOKOROX_CONV=f16: %.2f ms/step %.0fx' % (d['ms_per_step'], d['value']))"
