import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 2048, 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
host = src[:n * L].cpu().numpy().tobytes()
entries = [host[i * L:(i + 1) * L] for i in range(n)]
del src
ctx.bench_stream_threads(entries[:64], threads=16)
for T in (int(a) for a in sys.argv[1:]):
    print("==== threads", T, file=sys.stderr, flush=True)
    secs, out = ctx.bench_stream_threads(entries, threads=T)
    print(f"threads {T}: {n * L / secs / 2**30:.2f} GiB/s", file=sys.stderr, flush=True)
    secs, out = ctx.bench_stream_threads(entries, threads=T)
    print(f"threads {T} (again): {n * L / secs / 2**30:.2f} GiB/s", file=sys.stderr, flush=True)
ctx.close()
