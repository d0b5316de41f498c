#!/bin/bash
# A/B builds of the HIP library: tools/build_variant.sh NAME [-DFLAG ...] compiles every translation
# unit with the extra flags into lumenos_amd/csrc/variants/NAME/ and links liblumenos_hip.so there
# (select it at run time with LUMEN_HIP_LIB=<path>; the .so files are git-ignored but travel with gpurun).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/lumenos_amd/csrc/variants/$name
mkdir -p "$out"
objs=()
for s in "$root"/lumenos_amd/csrc/*.hip; do
  o=$out/$(basename "${s%.hip}").o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result "$@" -c "$s" -o "$o" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/liblumenos_hip.so" "${objs[@]}"
echo "$out/liblumenos_hip.so"
