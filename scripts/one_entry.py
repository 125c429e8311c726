"""One 1 MiB entry through pna_gpu_compress_batch, N times (for rocprofv3 --kernel-trace --stats: the kernels of a latency-mode batch)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
L = 1 << 20
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = pna.Context(0)
src = torch.empty(k * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, k, L, L, src.data_ptr())
host = src[:k * L].cpu().numpy().tobytes()
entries = [host[i * L:(i + 1) * L] for i in range(k)]
ctx.compress_batch(entries)
t0 = time.perf_counter()
for _ in range(reps):
    outs = ctx.compress_batch(entries)
dt = (time.perf_counter() - t0) / reps
tm = ctx.timing()
print(f"{k} entries: {dt * 1e3:.3f} ms per batch; blk_log {tm.blk_log}, units {tm.lz_units}; ratio {k * L / sum(map(len, outs)):.4f}; stages",
      {f: round(getattr(tm, f), 3) for f in ("ms_lz", "ms_stats", "ms_lit", "ms_seq", "ms_pack")})
