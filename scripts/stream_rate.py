"""Rate of the streaming facade (the CompressionWriter seam) driven the reference's way: one writer per task on T host threads,
every finish() joining the context's group commit.  python scripts/stream_rate.py [files] [file_mib]  (GPU box)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
host = src[:n * L].cpu().numpy().tobytes()
entries = [host[i * L:(i + 1) * L] for i in range(n)]
del src
print("host threads available:", len(os.sched_getaffinity(0)), flush=True)
ctx.bench_stream_threads(entries[:64], threads=16)                      # warm-up: workspace allocation
for T in (1, 4, 16, 64, 256, 512):
    b0, e0, _ = ctx.stream_stats()
    secs, out = ctx.bench_stream_threads(entries, threads=T)
    b1, e1, mx = ctx.stream_stats()
    print(f"threads {T:4d}: {n * L / secs / 2**30:7.2f} GiB/s of input, ratio {n * L / out:.3f}, {b1 - b0} device batches for {e1 - e0} entries "
          f"(mean {(e1 - e0) / max(1, b1 - b0):.1f}, largest so far {mx})", flush=True)
ctx.close()
