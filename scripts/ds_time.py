import importlib, os, sys, subprocess
child = r'''
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 131072, 4096
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(1, 0, n, L, L, src.data_ptr())
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
best = 1e9
for it in range(4):
    try:
        ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel(), algo=pna.ALGO_DEFLATE)
    except Exception as e:
        pass
    best = min(best, ctx.timing().ms_stats)
print(f"{best:.3f}")
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, PNA_GPU_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print(lib, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
