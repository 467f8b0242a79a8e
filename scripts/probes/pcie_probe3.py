"""hipHostRegister of FRESH (never touched) NumPy memory from several threads at once: does page-locking scale?"""
import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
rt = torch.cuda.cudart()
n = 100000 * 1001
d = torch.rand(n, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
for nt, piece_mb in ((1, 800), (4, 64), (8, 64), (16, 32), (16, 64), (16, 16)):
    h = np.empty(n)
    base, total = h.ctypes.data, n * 8
    piece = piece_mb << 20
    offs = list(range(0, total, piece))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nt) as ex:
        list(ex.map(lambda o: rt.cudaHostRegister(base + o, min(piece, total - o), 0), offs))
    t1 = time.perf_counter()
    torch.from_numpy(h).copy_(d, non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for o in offs: rt.cudaHostUnregister(base + o)
    print(f"{nt:2d} threads x {piece_mb:4d} MB pieces: register fresh 0.8 GB {1e3*(t1-t0):6.1f} ms, then D2H {1e3*(t2-t1):5.1f} ms", flush=True)
