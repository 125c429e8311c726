"""A/B of the per-stage times between library builds on the same box: python scripts/ab_stage.py libA.so libB.so ... [--files N]"""
import os, subprocess, sys
child = r'''
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = int(os.environ.get("AB_FILES", "10000")), 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
dst = torch.empty(n * (L // 2), dtype=torch.uint8, device="cuda")
best = None
for it in range(4):
    offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    t = ctx.timing(); v = (t.ms_lz, t.ms_stats, t.ms_lit, t.ms_seq, t.ms_pack)
    best = v if best is None else tuple(min(a, b) for a, b in zip(best, v))
print("lz %.3f stats %.3f lit %.3f seq %.3f pack %.3f out %d" % (*best, offs[-1]))
'''
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, PNA_GPU_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        print(rnd, os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
