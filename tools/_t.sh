timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do python bench.py --workload cfg5 --no-pmc --no-cpu-baseline --no-gym-loop | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['packed_redo_env_steps'])"; done
python bench.py --envs 16384 --no-pmc --no-cpu-baseline --no-gym-loop | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['packed_redo_env_steps'])"
python bench.py --envs 4096 --packed 1 --pipeline 1 --no-reorder --no-pmc --no-cpu-baseline --no-gym-loop | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['packed_redo_env_steps'])"
