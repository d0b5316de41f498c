cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_reference_shapes.py -q -x 2>&1 | tail -2
for n in 14 13 12; do python3 tools/ntt_only.py $n 512 60; done
bash tools/exp_env.sh ""
