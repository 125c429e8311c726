"""Diagnostic: per-phase cycle shares of k_lz (mean over the 16 waves of a workgroup), on the GPU box."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 1024, 1 << 20
import sys as _s
extra = int(_s.argv[1], 0) if len(_s.argv) > 1 else 0
ctx = pna.Context(0, flags=pna.F_STD | 0x100 | extra)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
for it in range(2):
    ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    st = ctx.lz_stamps(); tm = ctx.timing()
    tot = sum(st)
    names = ["load", "lookup->B2", "insert+match", "wait B3 (+publish)", "wait B4", "merge+masks (own work)", "emit", "parse loops+coverage (own work)"]
    print("lz ms", round(tm.ms_lz, 3), "cycles per 4096-position tile and wave", round(tot / (n * 256 * 16)), {k: f"{100 * v / tot:.1f}%" for k, v in zip(names, st)})
