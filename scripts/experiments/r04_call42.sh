#!/bin/bash
# round 4, call 42: the secondary rate tables on the final build (simulators, MIMO schemes, IA solvers, operators)
export TMPDIR=/tmp
timeout 400 python scripts/bench_simulators.py > gpurun_out/simulators.json 2> gpurun_out/simulators.err; echo "simulators rc=$?"
timeout 300 python scripts/bench_ia_solvers.py > gpurun_out/ia_solvers.json 2> gpurun_out/ia_solvers.err; echo "ia rc=$?"
timeout 300 python scripts/bench_mimo_schemes.py > gpurun_out/mimo_schemes.json 2> gpurun_out/mimo_schemes.err; echo "mimo rc=$?"
timeout 300 python scripts/bench_operators.py > gpurun_out/operators.json 2> gpurun_out/operators.err; echo "operators rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/simulators.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, {kk:(('%.3g'%vv) if isinstance(vv,float) else vv) for kk,vv in v.items() if not isinstance(vv,(dict,list))})
PY
