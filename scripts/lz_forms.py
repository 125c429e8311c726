"""LZ-stage time of the one-kernel form against the split form (k_lzm + k_lzp) by batch size, N x 1 MiB, latency mode off: python scripts/lz_forms.py"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
L = 1 << 20
nmax = 2048
src = torch.empty(nmax * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(nmax * (L + 1024), dtype=torch.uint8, device="cuda")
ctx = pna.Context(0)
ctx.corpus_fill_device(0, 0, nmax, L, L, src.data_ptr())
ctx.set_option("latency_max_mib", 0)
for n in (32, 64, 128, 256, 512, 768, 1024, 1536, 2048):
    row = []
    for name, smin in (("one-kernel", 1 << 20), ("split", 0)):
        ctx.set_option("lz_split_min", smin)
        best = 1e9
        for it in range(3):
            ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
            tm = ctx.timing(); best = min(best, tm.ms_lz)
        row.append(f"{name} {best:7.3f} ms (match launches {tm.lz_match_launches})")
    print(f"{n:5d} x 1 MiB: " + "   ".join(row), flush=True)
