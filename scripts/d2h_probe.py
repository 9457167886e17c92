import torch, time
x = torch.randn(11, 1000, 100, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, f in (("cpu()", lambda: x.cpu()),
                ("pinned alloc+copy", lambda: torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x)),):
    for _ in range(2): f()
    t=time.perf_counter()
    for _ in range(10): y=f()
    torch.cuda.synchronize()
    print(name, (time.perf_counter()-t)/10*1e3, "ms")
buf = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
t=time.perf_counter()
for _ in range(10): buf.copy_(x); torch.cuda.synchronize()
print("pinned reuse", (time.perf_counter()-t)/10*1e3, "ms")
big = torch.randn(1000, 100000, dtype=torch.float64, device="cuda")
t=time.perf_counter(); y=big.cpu(); print("800MB cpu()", (time.perf_counter()-t)*1e3)
t=time.perf_counter(); p=torch.empty(big.shape, dtype=big.dtype, pin_memory=True); print("800MB pinned alloc", (time.perf_counter()-t)*1e3)
t=time.perf_counter(); p.copy_(big); torch.cuda.synchronize(); print("800MB pinned copy", (time.perf_counter()-t)*1e3)
