"""pna_gpu_compress_batch (host buffers in, host buffers out) steady-state rate from the C ABI: python scripts/batch_rate.py [n] [MiB]"""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) << 20
ctx = pna.Context(0)
dev = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, dev.data_ptr())
host = dev[:n * L].cpu().numpy()
lib = pna.load_library()
cap = lib.pna_gpu_bound(2, L)
outs = [ctypes.create_string_buffer(cap) for _ in range(n)]
a_src = (ctypes.c_void_p * n)(*[host.ctypes.data + i * L for i in range(n)])
a_len = (ctypes.c_size_t * n)(*[L] * n)
a_dst = (ctypes.c_void_p * n)(*[ctypes.addressof(o) for o in outs])
a_cap = (ctypes.c_size_t * n)(*[cap] * n)
a_out = (ctypes.c_size_t * n)()
for it in range(4):
    t0 = time.perf_counter()
    rc = lib.pna_gpu_compress_batch(ctx._h, 2, 3, n, a_src, a_len, a_dst, a_cap, a_out)
    dt = time.perf_counter() - t0
    print(f"call {it}: rc {rc} {dt * 1e3:.1f} ms = {n * L / dt / 2**30:.2f} GiB/s, out {sum(a_out)}", flush=True)
