"""Diagnostic: per-phase cycle shares of k_lzp (split form of the LZ stage), on the GPU box.  Needs a library built with -DLZP_PROF:
scripts/build_variant.sh lzpprof -DLZP_PROF && PNA_GPU_LIB=portable-network-archive_amd/variants/libpna_gpu_lzpprof.so PNA_LZ_SPLIT=1 python scripts/lzp_stamps.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 2048, 1 << 20
ctx = pna.Context(0)
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
for it in range(2):
    ctx.lz_stamps()
    ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
    st = ctx.lz_stamps(); tm = ctx.timing()
    tot = sum(st)
    names = ["words -> lengths -> masks", "walk", "merge+masks+scans+records", "sequences", "literals", "-", "-", "-"]
    print("lz ms", round(tm.ms_lz, 3), "cycles (100 MHz clock) per tile", round(tot / (n * 256), 1), {k: f"{100 * v / tot:.1f}%" for k, v in zip(names, st) if k != "-"})
