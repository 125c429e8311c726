"""ONE foreign zlib stream (stdlib zlib, level 1, no sync flushes) of N MiB -- 2 GiB and more: the chunk decoder with the parallel executor's windows -- decoded on the device,
timed, compared in HBM.   python scripts/inflate_huge_stream.py [MiB]"""
import importlib, os, sys, time, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
ctx = pna.Context(0)
ctx.set_option("trace", 1)                                       # (says why, when a stream is not decoded in chunks)
L = 1 << 20
src = torch.empty(mib * L + 4096, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 9700, mib, L, L, src.data_ptr())
host = src[:mib * L].cpu().numpy()
t = time.time()
co = zlib.compressobj(1)
parts = []
for a in range(0, host.size, 256 << 20):
    parts.append(co.compress(host[a:a + (256 << 20)].tobytes())); print(f"  compressed {a >> 20} MiB ({time.time() - t:.0f} s)", flush=True)
parts.append(co.flush())
buf = np.frombuffer(b"".join(parts), dtype=np.uint8); n = buf.size
print(f"zlib level 1: {mib} MiB -> {n} B in {time.time() - t:.1f} s", flush=True)
comp = torch.from_numpy(buf.copy()).cuda(); back = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda")
for rep in range(2):
    t = time.time()
    try:
        ctx.decompress_batch_device(comp.data_ptr(), [0], [n], back.data_ptr(), [0], [host.size], algo=pna.ALGO_DEFLATE)
    except pna.PnaGpuError as e:
        print("refused:", e, flush=True); break
    torch.cuda.synchronize(); dt = time.time() - t
    tm = ctx.timing()
    print(f"{mib} MiB in one zlib stream: {dt * 1e3:.1f} ms = {mib / dt:.1f} MiB/s, streams decoded in chunks {tm.lz_match_launches}, equal {bool(torch.equal(back[:host.size], src[:host.size]))}", flush=True)
