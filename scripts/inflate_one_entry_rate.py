"""Decode rate of ONE deflate entry of N MiB written by this library (one zlib stream, a sync flush behind every 128 KiB, matches inside 1 MiB segments):
pieces decoded lane-per-piece, execution groups (one per segment) side by side."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pna.Context(0)
L = 1 << 20
n = mib * L
src = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 9700, mib, L, L, src.data_ptr())
cap = pna.bound(pna.ALGO_DEFLATE, n) + 64
comp = torch.empty(cap, dtype=torch.uint8, device="cuda")
offs = ctx.compress_batch_device(src.data_ptr(), [0, n], [n], comp.data_ptr(), cap, algo=pna.ALGO_DEFLATE)
back = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.decompress_batch_device(comp.data_ptr(), [offs[0]], [offs[1] - offs[0]], back.data_ptr(), [0], [n], algo=pna.ALGO_DEFLATE)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    tm = ctx.timing()
    print(f"one deflate entry of {mib} MiB ({offs[1] - offs[0]} B): {dt * 1e3:.1f} ms = {mib / 1024 / dt:.2f} GiB/s (kernels {tm.ms_lz:.1f} ms: walk {tm.ms_stats:.1f}, execution {tm.ms_lit:.1f}), equal {bool(torch.equal(back[:n], src[:n]))}", flush=True)
