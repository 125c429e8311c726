"""k_lzp's time (LZ stage minus the match kernel) on 4 096 x 1 MiB, for timing variants of the library (PNA_GPU_LIB)."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 4096, 1 << 20
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
ctx = pna.Context(0)
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
best = (1e9, 0)
for it in range(3):
    ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    tm = ctx.timing(); best = min(best, (tm.ms_lz - tm.ms_lz_match, tm.ms_lz_match))
print(f"k_lzp {best[0]:.3f} ms  (k_lzm {best[1]:.3f} ms)")
