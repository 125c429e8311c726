"""One device batch of n x 1 MiB (zstd level 3, inputs and outputs in HBM): time and what the library chose (block size, LZ units), the library's defaults against
128 KiB blocks whatever the batch (latency_max_mib = 0).   python scripts/batch_sizes.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
L = 1 << 20; N = 2048
ctx = pna.Context(0)
src = torch.empty(N * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, N, L, L, src.data_ptr())
cap = N * (L + 65536)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
sizes = [1, 4, 16, 32, 48, 64, 80, 96, 128, 160, 192, 256, 384, 512, 768, 1024, 1536, 2048]
for name, lm in (("defaults", 128), ("latency_max_mib 0", 0)):
    ctx.set_option("latency_max_mib", lm)
    out = []
    for n in sizes:
        best = 1e9; outs = None
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            outs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n)], [L] * n, dst.data_ptr(), cap)
            torch.cuda.synchronize(); best = min(best, (time.time() - t) * 1e3)
        tm = ctx.timing()
        out.append("%d: %.2f ms (2^%d, %d units, ratio %.3f)" % (n, best, tm.blk_log, tm.lz_units, n * L / outs[n]))
    print(name + ": " + "; ".join(out), flush=True)
