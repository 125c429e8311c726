"""ONE large zlib stream written by the stdlib's zlib (level 6, no sync flushes: what the reference's flate2 writes for a large deflate entry) through the device decoder:
block starts found by trial, chunks walked side by side, records executed by pointer jumping.   python scripts/inflate_foreign_stream_rate.py [MiB] [level]"""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.getcwd())
from oracle import codec
pna = importlib.import_module("portable-network-archive_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
raw = b"".join(codec.corpus_file(i % 2, 7000 + i, 1 << 20) for i in range(mib))
comp = zlib.compress(raw, level)
print(f"zlib -{level}: {len(raw)} -> {len(comp)} B")
with pna.Context(0) as ctx:
    for rep in range(reps):
        t0 = time.time()
        out = ctx.decompress_batch([comp], [len(raw)], algo=pna.ALGO_DEFLATE)
        dt = time.time() - t0
        tm = ctx.timing()
        print(f"one zlib stream of {mib} MiB: {dt * 1e3:.1f} ms wall = {mib / 1024 / dt:.2f} GiB/s incl. host copies; kernels {tm.ms_lz:.1f} ms = {mib / 1.024 / tm.ms_lz:.2f} GiB/s "
              f"(walks {tm.ms_stats:.1f}, execution {tm.ms_lit:.1f}); streams decoded in chunks: {tm.lz_match_launches}; equal {out == [raw]}")
