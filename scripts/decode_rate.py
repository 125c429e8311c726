"""Decode throughput of k_zdec on device-resident archives (n x 1 MiB entries compressed by this library)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
cap = n * pna.bound(pna.ALGO_ZSTD, L) + 64
comp = torch.empty(cap, dtype=torch.uint8, device="cuda")
offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, comp.data_ptr(), cap)
back = torch.zeros(n * L + 64, dtype=torch.uint8, device="cuda")
lens = [offs[i + 1] - offs[i] for i in range(n)]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.decompress_batch_device(comp.data_ptr(), offs[:n], lens, back.data_ptr(), [i * L for i in range(n)], [L] * n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"decode {n} x 1 MiB: wall {dt*1e3:.1f} ms = {n*L/dt/2**30:.1f} GiB/s of output; kernel {ctx.timing().ms_lz:.1f} ms; equal {bool(torch.equal(back[:n*L], src[:n*L]))}", flush=True)
