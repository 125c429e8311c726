"""Decode throughput for FOREIGN frames: n x 1 MiB entries compressed by the system libzstd (level 3, one frame per entry, as the reference writes them)."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")
from oracle import codec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
host = src[:n * L].cpu().numpy()
Z = codec.system_libzstd()
cap = Z.ZSTD_compressBound(L); buf = np.empty(cap, dtype=np.uint8)
parts, offs = [], [0]
t0 = time.time()
for i in range(n):
    k = Z.ZSTD_compress(buf.ctypes.data, cap, ctypes.c_char_p(host[i * L:].ctypes.data), L, 3)
    parts.append(buf[:k].copy()); offs.append(offs[-1] + k)
print(f"libzstd: {n} frames, ratio {n * L / offs[-1]:.3f}, {time.time() - t0:.1f} s", flush=True)
comp = torch.from_numpy(np.concatenate(parts)).cuda()
back = torch.zeros(n * L + 64, dtype=torch.uint8, device="cuda")
lens = [offs[i + 1] - offs[i] for i in range(n)]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.decompress_batch_device(comp.data_ptr(), offs[:n], lens, back.data_ptr(), [i * L for i in range(n)], [L] * n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"decode {n} x 1 MiB libzstd frames: wall {dt*1e3:.1f} ms = {n*L/dt/2**30:.1f} GiB/s of output; kernels {ctx.timing().ms_lz:.1f} ms; equal {bool(torch.equal(back[:n*L], src[:n*L]))}", flush=True)
