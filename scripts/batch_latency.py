"""Latency of one pna_gpu_compress_batch call (host entries in, host streams out) against the batch size: the fixed cost the group commit of the streaming facade has to amortise."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
L = 1 << 20
ctx = pna.Context(0)
n = 512
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
host = src[:n * L].cpu().numpy().tobytes()
entries = [host[i * L:(i + 1) * L] for i in range(n)]
ctx.compress_batch(entries[:300])
for k in (1, 2, 8, 32, 64, 128, 256, 512):
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter(); ctx.compress_batch(entries[:k]); dt = time.perf_counter() - t0
        best = min(best, dt)
    tm = ctx.timing()
    print(k, f"{best*1e3:.2f} ms", "kernels:", {f: round(getattr(tm, f), 2) for f in ("ms_lz", "ms_stats", "ms_lit", "ms_seq", "ms_pack", "ms_frame")}, flush=True)
