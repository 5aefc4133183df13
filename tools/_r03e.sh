#!/bin/bash
mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
timeout 600 python -m pytest tests/test_trpo.py -q -x -m gpu 2>&1 | tail -5 > $O/tests_trpo.txt
cat $O/tests_trpo.txt
DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 25 --out $O/train.json > $O/train.log 2>&1
tail -1 $O/train.log
DM_VF_ONE_LAUNCH=0 DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 25 --out $O/train3.json > $O/train3.log 2>&1
tail -1 $O/train3.log
for f in train train3; do python - <<PY
import json
d=json.load(open("$O/$f.json"))
h=d["history"] if isinstance(d,dict) else d
import numpy as np
keys=h[-1]["profile_ms"].keys()
print("$f", {k: round(float(np.mean([x["profile_ms"][k] for x in h[5:]])),2) for k in keys})
ts=[x["TimeElapsed"] for x in h]; print("iter ms", round(1e3*(ts[-1]-ts[5])/(len(ts)-6),2), "EpLen", h[-1]["EpLenMean"])
PY
done
