"""Stage times of the zstd pipeline at 10 000 x 1 MiB for library builds x sequence-coder forms (flag 0x1000 / 0x2000 force one form):
python scripts/seq_forms.py lib1.so lib2.so ..."""
import os, subprocess, sys
child = r'''
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 10000, 1 << 20
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
for extra in (0, 0x1000, 0x2000):
    ctx = pna.Context(0, flags=pna.F_STD | extra)
    ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
    best = None
    for it in range(3):
        offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
        t = ctx.timing()
        cur = (t.ms_seq, t.ms_lz, t.ms_stats, t.ms_lit, t.ms_pack)
        best = cur if best is None or cur[0] < best[0] else best
    print(hex(extra), "seq %.2f lz %.2f stats %.2f lit %.2f pack %.2f" % best, "out", offs[-1], flush=True)
    ctx.close()
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, PNA_GPU_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print(lib); print(out.stdout.strip() or out.stderr[-600:], flush=True)
