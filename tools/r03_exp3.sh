#!/bin/bash
# round 3, run 3 (one box): whole GPU suite with the new tests, default bench line (serve leg), replicas rehearsal
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03c_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r03c_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --cpu-utts 2 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err || { tail -20 gpurun_out/r03c_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03c_bench.json").read().strip().splitlines()[-1])
print("bench: %.2f ms/step, %.0fx, roofline frac %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
print("serve:", json.dumps(d["serve"]))
PY
KX_REPLICA_IDS=0,0 timeout -k 10 300 python bench.py --replicas 2 --steps 3 --warmup 1 > gpurun_out/r03c_replicas.json 2> gpurun_out/r03c_replicas.err || { tail -20 gpurun_out/r03c_replicas.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03c_replicas.json").read().strip().splitlines()[-1])
print("replicas [0,0]: %.2f ms/step, %.0fx, fan-out %.2f s" % (d["ms_per_step"], d["value"], d["weight_broadcast_s"]))
print("serve:", json.dumps(d["serve"]))
PY
