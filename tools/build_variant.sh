#!/bin/bash
# A/B builds of the HIP library: tools/build_variant.sh NAME [--patch tools/exp_X.patch ...] [-DFLAG ...] compiles every
# translation unit with the extra flags into lumenos_amd/csrc/variants/NAME/ and links liblumenos_hip.so there (select it at run
# time with LUMEN_HIP_LIB=<path>; the .so files are git-ignored but travel with gpurun).  --patch: the experiment patches under
# tools/ (code that was measured and is NOT part of the product, e.g. exp_no_butterflies.patch, exp_moddown_r4.patch) are applied to
# a COPY of the sources inside the variant's directory; the product tree is never touched.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/lumenos_amd/csrc/variants/$name
rm -rf "$out"; mkdir -p "$out"
patches=()
while [ "$1" = "--patch" ]; do patches+=("$(cd "$(dirname "$2")" && pwd)/$(basename "$2")"); shift; shift; done
src=$root/lumenos_amd/csrc
if [ ${#patches[@]} -gt 0 ]; then
  mkdir -p "$out/tree/lumenos_amd/csrc" "$out/tree/include"
  cp "$root"/lumenos_amd/csrc/*.hip "$root"/lumenos_amd/csrc/*.h "$out/tree/lumenos_amd/csrc/"
  cp "$root"/include/*.h "$out/tree/include/"
  for p in "${patches[@]}"; do (cd "$out/tree" && patch -p1 --fuzz=3 -s < "$p") || { echo "patch $p does not apply" >&2; exit 1; }; done
  src=$out/tree/lumenos_amd/csrc
fi
echo "${patches[*]##*/} $*" > "$out/FLAGS"   # lumenos_amd/_build.py source_hash() folds the variant's name and flags into the profile stamp
objs=()
for s in "$src"/*.hip; do
  o=$out/$(basename "${s%.hip}").o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result \
    -Rpass-analysis=kernel-resource-usage "$@" -c "$s" -o "$o" > "$o.log" 2>&1 &
  objs+=("$o")
done
wait
# the same gate as the product build: no kernel may spill or use scratch (a variant that does is not an A/B
# of the product, it is a different machine code regime)
python3 - "$out" <<'PY'
import glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "..", "..", ".."))
from lumenos_amd import _build
for log in sorted(glob.glob(os.path.join(sys.argv[1], "*.o.log"))):
    text = open(log, errors="ignore").read()
    if "error:" in text:
        sys.exit(text[-3000:])
    _build.check_resources(_build._resource_usage(text), log)
PY
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/liblumenos_hip.so" "${objs[@]}"
echo "$out/liblumenos_hip.so"
