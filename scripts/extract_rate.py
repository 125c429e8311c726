"""Rate of the read-side driver (pna_gpu_extract_archive_host) on the GPU box: python scripts/extract_rate.py [files]"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
names = [f"enwik/part{i:07d}.txt" for i in range(n)]
cap = pna.archive_bound(pna.ALGO_ZSTD, names, [L] * n)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
total, _ = ctx.create_archive_device(names, src.data_ptr(), [i * L for i in range(n)], [L] * n, dst.data_ptr(), cap)
arc = dst[:total].cpu().numpy().tobytes()
seen = [0, 0]
def cb(_u, idx, name, kind, data, ln):
    seen[0] += 1; seen[1] += ln
    return 0
fn = pna.ENTRY_FN(cb)
for it in range(3):
    seen[:] = [0, 0]
    t0 = time.perf_counter()
    rc = ctx._L.pna_gpu_extract_archive_host(ctx._h, arc, len(arc), None, 0, fn, None)
    dt = time.perf_counter() - t0
    assert rc == 0 and seen == [n, n * L], (rc, seen)
    print(f"extract {n} x 1 MiB from a {len(arc) / 2**20:.0f} MiB archive in host memory: {dt * 1e3:.1f} ms = {n * L / dt / 2**30:.2f} GiB/s of output (PCIe + pageable copies included)")
