cd $GRAFT_REPO_ROOT
V=$PWD/lumenos_amd/csrc/variants
echo "== default"; python3 tools/ntt_only.py 14 512 60
echo "== cpl32 (512 threads, 2 waves/SIMD at N=2^14)"; LUMEN_HIP_LIB=$V/cpl32/liblumenos_hip.so python3 tools/ntt_only.py 14 512 60
LUMEN_HIP_LIB=$V/cpl32/liblumenos_hip.so python3 -m pytest tests/test_gpu_parity.py -q -x -k "limb_ntt" 2>&1 | tail -2
bash tools/exp_env.sh "LUMEN_HIP_LIB=$V/cpl32/liblumenos_hip.so"
