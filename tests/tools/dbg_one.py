import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import codec
pna = importlib.import_module("portable-network-archive_amd")
ctx = pna.Context(0)
p = codec.default_params(); p.flags = 7
for name, d in (("t4096", codec.corpus_file(1, 9, 4096)), ("t5000", codec.corpus_file(1, 19, 5000)), ("t3000", codec.corpus_file(0, 29, 3000))):
    for rep in range(2):
        o = ctx.compress_batch([d])[0]; e = codec.model_compress(d, p)
        diffs = [i for i in range(min(len(o), len(e))) if o[i] != e[i]]
        print(name, rep, len(o), len(e), "ndiff", len(diffs), diffs[:12], o[-60:].hex() if diffs else "", e[-60:].hex() if diffs else "")
