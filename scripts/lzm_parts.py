"""Where k_lzm's time goes: the match kernel's HIP-event time for the encoder's feature subsets (run-time switches) on 4 096 x 1 MiB;
with PNA_GPU_LIB pointing at a -DLZM_EXP_NOK16 build (scripts/build_variant.sh nok16 -DLZM_EXP_NOK16) also without the second 16 bytes."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
pna = importlib.import_module("portable-network-archive_amd")
n, L = 4096, 1 << 20
src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (L + 1024), dtype=torch.uint8, device="cuda")
for name, fl in (("default (far + adoption + even inserts + lazy)", 0x77), ("no far candidates", 0x67), ("no adoption, every position inserted", 0x17), ("neither (fast set + lazy)", 0x07)):
    ctx = pna.Context(0, flags=fl)
    ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
    best = 1e9
    for it in range(3):
        offs = ctx.compress_batch_device(src.data_ptr(), [i * L for i in range(n + 1)], [L] * n, dst.data_ptr(), dst.numel())
        tm = ctx.timing(); best = min(best, tm.ms_lz_match)
    print(f"{name:50s} k_lzm {best:7.3f} ms  LZ stage {tm.ms_lz:7.3f} ms  out {offs[-1]}", flush=True)
    ctx.close()
