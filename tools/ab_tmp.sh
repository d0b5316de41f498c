cd $GRAFT_REPO_ROOT
V=$PWD/lumenos_amd/csrc/variants
python3 -m pytest tests/test_gpu_parity.py -q -x -k "ct_ntt or encode or golden" 2>&1 | tail -2
echo "== ILP 4 (default)"; python3 tools/encode_only.py 16384x4096 3 2>/dev/null | head -1
for v in ilp1 ilp2 ilp8; do echo "== $v"; LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so python3 tools/encode_only.py 16384x4096 3 2>/dev/null | head -1; done
