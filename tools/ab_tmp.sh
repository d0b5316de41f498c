cd $GRAFT_REPO_ROOT
V=$PWD/lumenos_amd/csrc/variants
for v in "" ctw64 ctw64t1024 ctw16; do
  if [ -z "$v" ]; then echo "== default (W=32, 512 thr)"; python3 tools/encode_only.py 16384x4096 3 | head -1
  else echo "== $v"; LUMEN_HIP_LIB=$V/$v/liblumenos_hip.so python3 tools/encode_only.py 16384x4096 3 | head -1; fi
done
