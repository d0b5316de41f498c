cd $GRAFT_REPO_ROOT
python3 tools/check_nccl_alias.py 2>&1 | tail -3
for n in 2 4; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --share-gpu --dist-backend gloo --config 2048x1024 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2
done
python3 bench.py --config 2048x1024 --steps 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
