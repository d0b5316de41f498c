"""Device memory around the scratch-placement selection: free memory after the first key switch (selection), after more of them, after
lumen_ctx_trim and after a second selection must come back to the same levels -- the candidates that lose are freed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def free_gb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**30


def main():
    job = bench.Job("16384x4096", 0, 1, 0)
    ctx = job.ctx
    cols = job.matrix.slice(0, 1024)
    marks = [("job built", free_gb())]
    for i in range(3):
        ctx.matrix_inner_sum(cols, job.r_pt, job.rows).free()
        ctx.sync()
        marks.append((f"inner product {i}", free_gb()))
    ctx.trim()
    marks.append(("after lumen_ctx_trim", free_gb()))
    for i in range(2):
        ctx.matrix_inner_sum(cols, job.r_pt, job.rows).free()
        ctx.sync()
        marks.append((f"inner product after trim {i}", free_gb()))
    for name, g in marks:
        print(f"{name:32s} free {g:8.2f} GiB")
    a = dict(marks)
    assert abs(a["inner product 0"] - a["inner product 2"]) < 0.05, "memory keeps shrinking between calls"
    assert abs(a["inner product 2"] - a["inner product after trim 1"]) < 0.25, "a second selection left candidates behind"
    assert a["after lumen_ctx_trim"] > a["inner product 2"] + 2.0, "trim gave nothing back"
    print("placement leak check OK")
    cols.free()
    job.close()


if __name__ == "__main__":
    main()
