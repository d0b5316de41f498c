#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r6_run15; mkdir -p "$out"; cd "$GRAFT_REPO_ROOT"
python -c "
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from tests.test_group_rccl import build_fakes; print(build_fakes())"
FUZZ_GROUP_TRANSPORT=rccl LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tests/cpp/fake_rccl:$LD_LIBRARY_PATH timeout -k 10 600 python tests/dev/fuzz_gpu.py 600 707 > "$out/fuzz_rccl_600.txt" 2>&1 || { tail -5 "$out/fuzz_rccl_600.txt"; exit 1; }
tail -1 "$out/fuzz_rccl_600.txt"
bash tools/pmc_mem_probe.sh gpurun_out/r6_run15/mem_probe 2>&1 | tail -14
