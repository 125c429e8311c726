"""One device batch of N x 1 MiB with the encoder flags of $PNA_FLAGS (hex; default 0x77), for profiler runs: python scripts/one_batch.py [N]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1 << 20
fl = int(os.environ.get("PNA_FLAGS", "0x77"), 0)
ctx = pna.Context(0, flags=fl)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
tm = ctx.timing()
print(hex(fl), "k_lzm", round(tm.ms_lz_match, 3), "LZ", round(tm.ms_lz, 3), "out", offs[-1])
