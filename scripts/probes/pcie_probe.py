"""D2H options for the host-pointer entry: pageable vs pinned destination, allocation costs."""
import time, numpy as np, torch
n = 100000 * 1001  # one of T/z/p for the headline fan (0.8 GB)
d = torch.rand(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def t(f, label, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{label:55s} {best*1e3:8.1f} ms  {n*8/best/1e9:6.1f} GB/s", flush=True)
    return r
t(lambda: np.empty(n), "np.empty (untouched)")
t(lambda: np.zeros(n) + 0, "np fresh + first touch")
h = np.empty(n)
t(lambda: torch.from_numpy(h).copy_(d), "D2H into fresh pageable numpy (first touch)", reps=1)
t(lambda: torch.from_numpy(h).copy_(d), "D2H into touched pageable numpy")
p = t(lambda: torch.empty(n, dtype=torch.float64, pin_memory=True), "allocate pinned (hipHostMalloc)")
t(lambda: p.copy_(d), "D2H into pinned")
t(lambda: np.copyto(h, p.numpy()), "CPU memcpy pinned -> pageable (1 thread)")
hr = torch.from_numpy(h)
t(lambda: torch.cuda.cudart().cudaHostRegister(h.ctypes.data, n * 8, 0), "hipHostRegister existing pageable", reps=1)
t(lambda: hr.copy_(d), "D2H into registered")
t(lambda: torch.cuda.cudart().cudaHostUnregister(h.ctypes.data), "hipHostUnregister", reps=1)
