"""Ad-hoc: host->device time of the headline observation array (B=65536, T=10000, m=2 fp32 = 5.24 GB) and
device->host time of one posterior stream slice, to quote the PCIe-inclusive rate in DESIGN.md."""
import time, numpy as np, torch
B, T, m = 65536, 10000, 2
y = np.zeros((B, T, m), np.float32)
torch.cuda.init()
for name, src in (("pageable", torch.from_numpy(y)), ("pinned", torch.from_numpy(y).pin_memory())):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = src.to("cuda", non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"H2D {name:9s}: {dt*1e3:8.1f} ms  {y.nbytes/dt/1e9:6.1f} GB/s")
    del d
means = torch.zeros((B, 1, T, 4), device="cuda")
host = torch.empty(means.shape, dtype=torch.float32).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
host.copy_(means, non_blocking=True); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"D2H pinned  (filtered means, 10.5 GB): {dt*1e3:8.1f} ms  {means.numel()*4/dt/1e9:6.1f} GB/s")
