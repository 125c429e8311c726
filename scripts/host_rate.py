"""The end-to-end leg on its own: N x 1 MiB entries in PAGEABLE host memory -> pna_gpu_create_archive_host -> counting sink.
python scripts/host_rate.py [files] [option=value ...]   e.g. sub_mib=256 stage_threads=16   (PNA_CREATE_TRACE=1 prints the phases of every sub-batch)"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
L = 1 << 20
ctx = pna.Context(0)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, int(v))
dev = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, dev.data_ptr())
host = dev[:n * L].cpu().numpy()
del dev
ents = [host[i * L:(i + 1) * L].tobytes() for i in range(n)]
names = [f"enwik/part{i:07d}.txt" for i in range(n)]
Lb = pna.load_library()
count = [0]
def _sink(_u, buf, k):
    count[0] += k
    return 0
cb = pna.SINK_FN(_sink)
a_names = (ctypes.c_char_p * n)(*[s.encode() for s in names])
a_src = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in ents])
a_len = (ctypes.c_size_t * n)(*[L] * n)
for it in range(4):
    count[0] = 0
    t0 = time.perf_counter()
    rc = Lb.pna_gpu_create_archive_host(ctx._h, pna.ALGO_ZSTD, pna.LEVEL_DEFAULT, n, a_names, a_src, a_len, cb, None)
    t1 = time.perf_counter()
    print(f"{' '.join(sys.argv[1:])}: rc {rc} {1e3*(t1-t0):.1f} ms = {n*L/(t1-t0)/2**20:.0f} MiB/s of input, archive {count[0]} B", flush=True)
