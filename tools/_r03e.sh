#!/bin/bash
mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
timeout 600 python -m pytest tests/test_trpo.py tests/test_gpu_rollout.py -q -x -m gpu  2>&1 | tail -5 > $O/tests_trpo.txt
cat $O/tests_trpo.txt
DM_TRPO_PROFILE=1 timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 40 --out $O/train.json > $O/train.log 2>&1
tail -1 $O/train.log
timeout 300 python tools/train_trpo.py --envs 4096 --horizon 128 --iters 40 --out $O/train_noprof.json > $O/train_noprof.log 2>&1
tail -1 $O/train_noprof.log
for f in train train_noprof; do python - <<PY
import json
d=json.load(open("$O/$f.json"))
h=d["history"] if isinstance(d,dict) else d
import numpy as np
if "profile_ms" in h[-1]:
    keys=h[-1]["profile_ms"].keys()
    print("$f", {k: round(float(np.mean([x["profile_ms"][k] for x in h[5:]])),2) for k in keys})
ts=[x["TimeElapsed"] for x in h]; print("$f iter ms", round(1e3*(ts[-1]-ts[5])/(len(ts)-6),2), "EpLen", h[-1]["EpLenMean"], "kl", h[-1]["meankl"])
PY
done
