"""Diagnostic: LZ-stage output (sequences) of the device vs the model for one small input, deflate or zstd parameters."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import codec
pna = importlib.import_module("portable-network-archive_amd")
ctx = pna.Context(0)
algo = sys.argv[1] if len(sys.argv) > 1 else "deflate"
d = bytes(5000) if len(sys.argv) < 3 else (b"ab" * 70000)[: int(sys.argv[2])]
p = codec.deflate_default_params() if algo == "deflate" else codec.default_params()
m = codec.model_lz_segment(d, p)
o = ctx.compress_batch([d], algo=pna.ALGO_DEFLATE if algo == "deflate" else pna.ALGO_ZSTD)
g = ctx.debug_block(0)
print("model nseq", len(m[0][0]), "nlit", len(m[0][1]), "gpu nseq", len(g[0]), "nlit", len(g[1]))
for i, (a, b) in enumerate(zip(m[0][0], g[0])):
    if a != b or i < 6:
        print(i, "model", a, "gpu", b)
    if a != b and i > 40:
        break
