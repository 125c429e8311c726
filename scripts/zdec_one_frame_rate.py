"""Decode rate of ONE foreign zstd frame (libzstd, one ZSTD_compress call) of N MiB on the device: the one-workgroup-per-frame kernel (k_zdec)."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")
from oracle import codec
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = pna.Context(0)
L = 1 << 20
src = torch.empty(mib * L + 4096, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 9500, mib, L, L, src.data_ptr())
host = src[:mib * L].cpu().numpy()
Z = codec.system_libzstd()
cap = Z.ZSTD_compressBound(host.size); buf = np.empty(cap, dtype=np.uint8)
n = Z.ZSTD_compress(buf.ctypes.data, cap, ctypes.c_char_p(host.ctypes.data), host.size, 1)
comp = torch.from_numpy(buf[:n]).cuda(); back = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda")
for rep in range(2):
    t = time.time(); ctx.decompress_batch_device(comp.data_ptr(), [0], [n], back.data_ptr(), [0], [host.size]); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{mib} MiB in one frame ({n} B): {dt * 1e3:.1f} ms = {mib / dt:.1f} MiB/s, equal {bool(torch.equal(back[:host.size], src[:host.size]))}", flush=True)
