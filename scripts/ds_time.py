"""Stage-1 time (k_dstats + k_adler) of the deflate path on 131 072 x 4 KiB entries for each library given on the command line: python scripts/ds_time.py lib1.so lib2.so ...
Used for phase timing by early exits: variants of k_deflate.hip with a temporary `return` behind the histogram / the code builds / the code assignment,
built with scripts/build_variant2.sh NAME k_deflate.hip (the exits are not kept in the tree)."""
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
