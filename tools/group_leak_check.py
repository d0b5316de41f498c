import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
job = bench.Job("4096x2048", 0, 4, 0, 0, False, [0, 0, 0, 0])
from lumenos_amd.hip import Group
job.group = Group(job.ctxs, transport="copy")
free = []
for it in range(12):
    job.step_group()
    for c in job.ctxs: c.sync()
    free.append(torch.cuda.mem_get_info()[0] / 2**20)
print("free MiB after each step:", [round(x) for x in free])
assert abs(free[-1] - free[3]) < 64, "device memory keeps growing"
job.enable_ring_switch(10)
for it in range(3):
    job.step_group()
print("ring switch steps ok; free MiB", round(torch.cuda.mem_get_info()[0] / 2**20))
job.close()
print("after close free MiB", round(torch.cuda.mem_get_info()[0] / 2**20))
