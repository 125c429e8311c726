"""Stage times of one device batch of N small entries under several encoder flag sets (which part of the per-entry statistics costs what):
python scripts/small_entries_stages.py [N] [BYTES] [KIND]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 1
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 256), dtype=torch.uint8, device="cuda")
for name, fl in (("default", None), ("no huffman", 0x76), ("no fse", 0x75), ("neither", 0x74)):
    ctx = pna.Context(0) if fl is None else pna.Context(0, flags=fl)
    ctx.corpus_fill_device(kind, 0, n, L, L, src.data_ptr())
    for _ in range(2):
        offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    tm = ctx.timing()
    print(f"{name:12s} lz {tm.ms_lz:6.2f} stats {tm.ms_stats:6.2f} lit {tm.ms_lit:6.2f} seq {tm.ms_seq:6.2f} pack {tm.ms_pack:6.2f}  out {offs[-1]}")
    ctx.close()
