"""A/B timing of k_lz between library builds on the SAME box: python scripts/ab.py libA.so libB.so ... (runs each in a child process, alternating).
An argument of the form K=V[,K=V...] runs the default library under that environment instead (e.g. PNA_LZ_SPLIT=1); `default` = no change."""
import os, subprocess, sys
child = r'''
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = int(os.environ.get("AB_N", "2048")), 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
best = 1e9
for it in range(5):
    offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    best = min(best, ctx.timing().ms_lz)
print(f"{best:.3f} {offs[-1]}")
'''
libs = sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        if lib == "default": env = dict(os.environ)
        elif "=" in lib: env = dict(os.environ, **dict(kv.split("=", 1) for kv in lib.split(",")))
        else: env = dict(os.environ, PNA_GPU_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        print(rnd, lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
