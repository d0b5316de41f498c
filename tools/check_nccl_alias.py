"""One-GPU check of the plumbing the multi-GPU bench relies on but a one-GPU box cannot run at N > 1:
torch tensors that ALIAS the library's device memory (CUDA array interface) handed to RCCL collectives
(world_size 1 process group: the collectives degenerate to copies, the aliasing and stream hand-over are
real).  Prints OK or raises."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lumenos_amd import params as lp  # noqa: E402
from lumenos_amd.hip import Context  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
P = lp.generate_bgv_params_for_ntt(16, 10)
ctx = Context(P.log_n, P.q, P.p, P.psi, P.T)
a = ctx.new_set_lanes(8, 2, 1).fill_random(3)
b = ctx.new_set_lanes(8, 2, 1)
bench.all_to_all_sets(dist, a, b, 1)
assert np.array_equal(a.download(), b.download()), "all_to_all_single on aliased device memory"
s = ctx.new_set(5, 2).fill_random(4)
want = ctx.leaf_digests(s)
ctx.leaf_digests_begin(s)
ptr, n = ctx.leaf_digests_end_device()
root = bench.all_gather_root(dist, ctx, ptr, n, 5, 1)
assert root == ctx.merkle_build(want)[1], "all_gather_into_tensor + device Merkle root"
dist.destroy_process_group()
print("check_nccl_alias OK")
