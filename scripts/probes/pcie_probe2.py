"""D2H of one trajectory array (0.8 GB) into NumPy memory: what each way of receiving it costs on this box.
(fresh = np.empty, never touched; the host-pointer entry gets such buffers from pr.shoot_rays)"""
import time, ctypes, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 100000 * 1001
d = torch.rand(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
rt = torch.cuda.cudart()
def tm(f, label):
    t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:72s} {dt*1e3:8.1f} ms  {n*8/dt/1e9:6.1f} GB/s", flush=True)
    return r
def prefault(h, nt=16):
    b = h.view(np.uint8)
    per = (len(b) + nt - 1) // nt
    def touch(k):
        s = b[k * per:(k + 1) * per:4096]
        s[...] = 0
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(touch, range(nt)))
for rep in range(2):
    print("--- rep", rep)
    h = np.empty(n); tm(lambda: torch.from_numpy(h).copy_(d), "D2H into FRESH pageable (runtime staging + first touch)")
    tm(lambda: torch.from_numpy(h).copy_(d), "D2H into the same, now touched, pageable")
    h = np.empty(n); tm(lambda: prefault(h), "prefault FRESH buffer, 16 threads")
    tm(lambda: torch.from_numpy(h).copy_(d), "D2H into prefaulted pageable")
    h = np.empty(n); tm(lambda: rt.cudaHostRegister(h.ctypes.data, n * 8, 0), "hipHostRegister FRESH buffer")
    hr = torch.from_numpy(h)
    tm(lambda: hr.copy_(d, non_blocking=True), "D2H into registered")
    tm(lambda: rt.cudaHostUnregister(h.ctypes.data), "hipHostUnregister")
    h = np.empty(n); prefault(h); tm(lambda: rt.cudaHostRegister(h.ctypes.data, n * 8, 0), "hipHostRegister PREFAULTED buffer")
    hr = torch.from_numpy(h); tm(lambda: hr.copy_(d, non_blocking=True), "D2H into registered (prefaulted)")
    tm(lambda: rt.cudaHostUnregister(h.ctypes.data), "hipHostUnregister")
    # chunked: register + copy in 64 MB pieces, pipelined over two streams
    h = np.empty(n); prefault(h)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def chunked(piece=64 << 20):
        off = 0; k = 0
        while off < n * 8:
            sz = min(piece, n * 8 - off)
            rt.cudaHostRegister(h.ctypes.data + off, sz, 0)
            with torch.cuda.stream(s1 if k % 2 == 0 else s2):
                torch.from_numpy(h.view(np.uint8)[off:off + sz]).copy_(d.view(torch.uint8)[off:off + sz], non_blocking=True)
            off += sz; k += 1
        torch.cuda.synchronize()
        off = 0
        while off < n * 8:
            rt.cudaHostUnregister(h.ctypes.data + off); off += piece
    tm(chunked, "register + D2H + unregister in 64 MB pieces, 2 streams (prefaulted)")
p = tm(lambda: torch.empty(n, dtype=torch.float64, pin_memory=True), "allocate pinned staging (hipHostMalloc)")
tm(lambda: p.copy_(d), "D2H into pinned staging")
h = np.empty(n); prefault(h)
def par_copy(nt=16):
    src = p.numpy(); per = (n + nt - 1) // nt
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(lambda k: np.copyto(h[k * per:(k + 1) * per], src[k * per:(k + 1) * per]), range(nt)))
tm(par_copy, "CPU memcpy pinned -> prefaulted pageable, 16 threads")
