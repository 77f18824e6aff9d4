#!/bin/bash
# the default bench command once, key fields of its line
O=gpurun_out/p2h; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo rc=$rc; tail -3 $O/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print(d["value"], d["window_proofs_per_s"]); print(d.get("poseidon2_hasher")); print(d["poseidon_hashing"])
r=d["roofline"]; print(r["frac"], r.get("traffic"), r.get("valu_roofline"), r.get("counters_stale"))
PY
exit $rc
