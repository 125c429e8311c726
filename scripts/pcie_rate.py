"""PCIe-inclusive archive-create rate (DESIGN.md quotes it; bench.py's `value` is the HBM-resident rate).
(a) pinned host buffers: H2D -> pna_gpu_create_archive_device -> D2H of the archive bytes, one shot
(b) the plain host C ABI pna_create_archive() (pageable memory, host chunk writer + CRC)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = 1 << 20
ctx = pna.Context(0)
dev = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, dev.data_ptr())
host_in = torch.empty(n * L, dtype=torch.uint8).pin_memory()
host_in.copy_(dev[:n * L]); torch.cuda.synchronize()
names = [f"enwik/part{i:07d}.txt" for i in range(n)]
cap = pna.archive_bound(pna.ALGO_ZSTD, names, [L] * n)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
host_out = torch.empty(cap, dtype=torch.uint8).pin_memory()
offs, lens = [i * L for i in range(n)], [L] * n
cache = {}
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev[:n * L].copy_(host_in, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    total, _ = ctx.create_archive_device(names, dev.data_ptr(), offs, lens, dst.data_ptr(), cap, _cache=cache); t2 = time.perf_counter()
    host_out[:total].copy_(dst[:total], non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"(a) pinned one-shot: H2D {1e3*(t1-t0):.1f} ms ({n*L/(t1-t0)/2**30:.1f} GiB/s)  gpu {1e3*(t2-t1):.1f} ms  D2H {1e3*(t3-t2):.1f} ms "
          f"({total/(t3-t2)/2**30:.1f} GiB/s)  total {n*L/(t3-t0)/2**20:.0f} MiB/s", flush=True)
m = min(n, 1024)
ents = [bytes(host_in[i * L:(i + 1) * L].numpy()) for i in range(m)]
for it in range(2):
    t0 = time.perf_counter(); arc = pna.create_archive(ctx, names[:m], ents); t1 = time.perf_counter()
    print(f"(b) host C ABI pna_create_archive, {m} entries: {1e3*(t1-t0):.1f} ms = {m*L/(t1-t0)/2**20:.0f} MiB/s  (archive {len(arc)} B)", flush=True)
# (c) the same C entry point with a sink that only counts (what a C/Rust caller writing to a file would see, minus the write)
import ctypes
Lb = pna.load_library()
m = n
ents = [bytes(host_in[i * L:(i + 1) * L].numpy()) for i in range(m)]
count = [0]
def _sink(_u, buf, k):
    count[0] += k
    return 0
cb = pna.SINK_FN(_sink)
a_names = (ctypes.c_char_p * m)(*[s.encode() for s in names[:m]])
a_src = (ctypes.c_void_p * m)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in ents])
a_len = (ctypes.c_size_t * m)(*[L] * m)
for it in range(3):
    count[0] = 0
    t0 = time.perf_counter()
    rc = Lb.pna_gpu_create_archive_host(ctx._h, pna.ALGO_ZSTD, pna.LEVEL_DEFAULT, m, a_names, a_src, a_len, cb, None)
    t1 = time.perf_counter()
    print(f"(c) pna_gpu_create_archive_host, {m} x 1 MiB pageable host entries, counting sink: rc {rc}  {1e3*(t1-t0):.1f} ms = {m*L/(t1-t0)/2**20:.0f} MiB/s  (archive {count[0]} B)", flush=True)
# (d) the batch API of the seam (pna_gpu_compress_batch): host buffers in, host buffers out
outs = ctx.compress_batch(ents[:2048])
t0 = time.perf_counter(); outs = ctx.compress_batch(ents[:2048]); t1 = time.perf_counter()
print(f"(d) pna_gpu_compress_batch, {len(ents[:2048])} x 1 MiB host entries -> host buffers (incl. the Python wrapper's copies): {1e3*(t1-t0):.1f} ms = {len(ents[:2048])*L/(t1-t0)/2**20:.0f} MiB/s", flush=True)
