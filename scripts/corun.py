"""experiments/corun.hip beside the library's default workload: does a one-wave, LDS-free, vector-bound kernel find issue slots on CUs the match kernel fills?
   python3 scripts/corun.py [files]   (GPU box; builds nothing: build/libcorun.so travels)"""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
names = [f"f{i}" for i in range(n)]
off = [i * L for i in range(n + 1)]; ln = [L] * n
cap = pna.archive_enc_bound(pna.ALGO_ZSTD, names, ln, None)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
cache = {}
C = ctypes.CDLL(os.path.join(ROOT, "build", "libcorun.so"))
C.corun_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p]
side = torch.cuda.Stream()
out = torch.zeros(16, dtype=torch.int32, device="cuda")
def lib_step():
    ctx.create_archive_device(names, src.data_ptr(), off, ln, dst.data_ptr(), cap, _cache=cache, want_offsets=False)
    t = ctx.timing(); return t.ms_lz_match, t.ms_lz
def spin(wgs, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side); C.corun_launch(ctypes.c_void_p(side.cuda_stream), wgs, iters, ctypes.c_void_p(out.data_ptr())); e1.record(side)
    return e0, e1
lib_step(); lib_step()
m, lz = lib_step(); print(f"library alone: k_lzm {m:.2f} ms, LZ stage {lz:.2f} ms")
for wgs, iters in ((1024 * 4, 40000), (1024 * 8, 30000), (1024 * 2, 60000), (1024, 100000)):
    e0, e1 = spin(wgs, iters); torch.cuda.synchronize(); alone = e0.elapsed_time(e1)
    inst = wgs * iters * 64
    print(f"spin alone: {wgs} waves x {iters * 64} instructions: {alone:.2f} ms = {inst / alone / 1e6:.1f} G wave-instructions/s")
    e0, e1 = spin(wgs, iters)
    m, lz = lib_step(); torch.cuda.synchronize()
    both = e0.elapsed_time(e1)
    print(f"  beside the library's step: spin {both:.2f} ms, k_lzm {m:.2f} ms, LZ stage {lz:.2f} ms")
