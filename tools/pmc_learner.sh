# MFMA busy cycles of the learner's kernels (one PMC pass of the training loop, a few iterations):  bash tools/pmc_learner.sh <tag>
TAG=$1; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
( rocprofv3 -L 2>/dev/null | grep -i "mfma" | head -30 ) > gpurun_out/$TAG/mfma_counters.txt
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_pmc -- python $OLDPWD/tools/train_trpo.py --envs 4096 --horizon 128 --iters 6 --out /tmp/t.json > /dev/null 2>&1 )
python - <<'PY' | tee gpurun_out/$TAG/learner_pmc.md
import sqlite3, glob
db = glob.glob("/tmp/p_pmc/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%k_pg%' or kernel_name like '%k_vf_grad%' group by kernel_name, counter_name").fetchall()
dur = dict(cur.execute("select name, avg(end-start) from kernels where name like '%k_pg%' or name like '%k_vf_grad%' group by name").fetchall())
print("| kernel | counter | avg per launch | launches | avg us (this pass) |"); print("|---|---|---|---|---|")
for r in rows:
    print("| %s | %s | %.0f | %d | %.1f |" % (r[0].split("(")[0], r[1], r[2], r[3], dur.get(r[0], 0) / 1e3))
PY
