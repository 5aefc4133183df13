for v in R4 PTX; do
  echo "== $v train"; DMENV_LIB=$PWD/build_ab/$v.so timeout 400 python tools/train_trpo.py --envs 4096 --horizon 128 --seconds 60 --out gpurun_out/f2/trpo_train_60s_$v.json 2>&1 | tail -1
done
