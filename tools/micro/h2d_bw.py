import torch, time
x = torch.empty(1<<30, dtype=torch.uint8).pin_memory()
d = torch.empty(1<<30, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t
print("H2D pinned: %.1f GB/s" % (5*(1<<30)/dt/1e9))
