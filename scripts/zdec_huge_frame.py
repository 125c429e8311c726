"""ONE foreign zstd frame (libzstd, one ZSTD_compress call, level 1) of N MiB -- 2 GiB and more: the parallel executor's windows -- decoded on the device, timed, compared in HBM.
The one-workgroup fallback is refused (zdec_fallback_max_mib), so a frame the parallel path does not take fails at once instead of running for minutes."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")
from oracle import codec
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
ctx = pna.Context(0)
ctx.set_option("zdec_fallback_max_mib", 64)
L = 1 << 20
src = torch.empty(mib * L + 4096, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 9500, mib, L, L, src.data_ptr())
host = src[:mib * L].cpu().numpy()
Z = codec.system_libzstd()
cap = Z.ZSTD_compressBound(host.size); buf = np.empty(cap, dtype=np.uint8)
t = time.time()
n = Z.ZSTD_compress(buf.ctypes.data, cap, ctypes.c_char_p(host.ctypes.data), host.size, 1)
print(f"libzstd level 1: {mib} MiB -> {n} B in {time.time() - t:.1f} s", flush=True)
comp = torch.from_numpy(buf[:n]).cuda(); back = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda")
for rep in range(2):
    t = time.time()
    try:
        ctx.decompress_batch_device(comp.data_ptr(), [0], [n], back.data_ptr(), [0], [host.size])
    except pna.PnaGpuError as e:
        print("refused:", e, flush=True); break
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"{mib} MiB in one frame: {dt * 1e3:.1f} ms = {mib / dt:.1f} MiB/s, equal {bool(torch.equal(back[:host.size], src[:host.size]))}", flush=True)
