import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 2048, 1 << 20
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
for extra in (0, 0x400, 0x800):
    ctx = pna.Context(0, flags=pna.F_LAZY | extra)
    ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
    for it in range(2):
        ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel(), algo=pna.ALGO_DEFLATE)
    tm = ctx.timing(); print(hex(extra), "dblock ms", round(tm.ms_seq, 3), "lz", round(tm.ms_lz, 2))
    ctx.close()
