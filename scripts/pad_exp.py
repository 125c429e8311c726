import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 1024, 1 << 20
for extra in (0, 0x400, 0x800, 0xC00):
    ctx = pna.Context(0, flags=pna.F_STD | extra)
    src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
    ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
    dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
    for it in range(3):
        ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    print(hex(extra), "lz ms", round(ctx.timing().ms_lz, 3), flush=True)
    ctx.close()
