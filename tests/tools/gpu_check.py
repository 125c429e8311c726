"""Stage-by-stage GPU-vs-oracle diagnosis (run on the GPU box): python tests/tools/gpu_check.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import codec
pna = importlib.import_module("portable-network-archive_amd")

def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1

def main(extra_flags=0):
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
    ctx = pna.Context(0, flags=(pna.F_HUF | pna.F_FSE | pna.F_LAZY | extra_flags))
    print("ctx flags extra", hex(extra_flags))
    # 1. corpus generator parity
    for kind, n in ((0, 20000), (1, 4096), (2, 8192), (0, 1 << 20)):
        t = torch.empty(n * 2 + 4096, dtype=torch.uint8, device="cuda")
        ctx.corpus_fill_device(kind, 5, 2, n, n, t.data_ptr())
        got = bytes(t[: 2 * n].cpu().numpy().tobytes())
        exp = codec.corpus_file(kind, 5, n) + codec.corpus_file(kind, 6, n)
        print("corpus kind", kind, n, "OK" if got == exp else f"MISMATCH at {first_diff(got, exp)}")
    # 2. compression parity
    p = codec.default_params(); p.flags = codec.F_HUF | codec.F_FSE | codec.F_LAZY
    cases = {
        "txt300k": codec.corpus_file(0, 1, 300000), "empty": b"", "one": b"a", "txt64k": codec.corpus_file(1, 2, 65536),
        "zeros5000": bytes(5000), "rnd10000": codec.corpus_file(2, 3, 10000), "txt1m": codec.corpus_file(0, 3, 1 << 20),
        "txt2m+": codec.corpus_file(0, 4, (2 << 20) + 12345), "zero1m": bytes(1 << 20), "x300k": b"x" * 300000,
        "ab": b"ab" * 70000, "abc": (b"abcdefghij" * 20000)[: 131072 + 77], "tiny7": b"abcdefg", "tiny8": b"abcdefgh",
        "t2047": codec.corpus_file(0, 9, 2047), "t2049": codec.corpus_file(0, 9, 2049), "t4096": codec.corpus_file(1, 9, 4096),
        "mixed": codec.corpus_file(0, 7, 200000) + codec.corpus_file(2, 7, 100000) + bytes(150000) + codec.corpus_file(1, 7, 300000),
    }
    names = list(cases)
    t0 = time.time()
    outs = ctx.compress_batch([cases[k] for k in names])
    print("batch time", time.time() - t0)
    tm = ctx.timing()
    print("timing ms: lz %.3f stats %.3f lit %.3f seq %.3f pack %.3f" % (tm.ms_lz, tm.ms_stats, tm.ms_lit, tm.ms_seq, tm.ms_pack))
    bad = 0
    blk_index = 0
    for k, o in zip(names, outs):
        d = cases[k]
        exp = codec.model_compress(d, p)
        try:
            dec = codec.zstd_decompress(o, len(d)); dec_ok = dec == d
        except Exception as e:
            dec_ok = False; dec = repr(e)
        same = o == exp
        print(f"{k:10s} in {len(d):8d} gpu {len(o):8d} model {len(exp):8d} decode {'OK' if dec_ok else 'FAIL'} bitexact {'OK' if same else 'DIFF@%d' % first_diff(o, exp)}")
        if not (same and dec_ok):
            bad += 1
    # 2b. deflate parity (zlib streams): bit-exact vs the model, inflate with stdlib zlib
    import zlib
    outs = ctx.compress_batch([cases[k] for k in names], algo=pna.ALGO_DEFLATE)
    tm = ctx.timing()
    print("deflate timing ms: lz %.3f stats %.3f lit %.3f seq %.3f pack %.3f" % (tm.ms_lz, tm.ms_stats, tm.ms_lit, tm.ms_seq, tm.ms_pack))
    for k, o in zip(names, outs):
        d = cases[k]
        exp = codec.deflate_model_compress(d)
        try:
            dec_ok = zlib.decompress(o) == d
        except Exception as e:
            dec_ok = False
        same = o == exp
        if not (same and dec_ok):
            bad += 1
            print(f"DEFLATE {k:10s} in {len(d):8d} gpu {len(o):8d} model {len(exp):8d} inflate {'OK' if dec_ok else 'FAIL'} bitexact {'OK' if same else 'DIFF@%d' % first_diff(o, exp)}")
    print("deflate cases done")
    # 3. LZ stage detail for a single-entry batch when something is off
    if bad:
        for k in names:
            d = cases[k]
            if not d:
                continue
            o = ctx.compress_batch([d])[0]
            if o == codec.model_compress(d, p):
                continue
            print("== LZ stage diff for", k)
            gb = 0
            for s0 in range(0, len(d), 1 << 20):
                seg = d[s0:s0 + (1 << 20)]
                model = codec.model_lz_segment(seg, p)
                for b, (ms, ml) in enumerate(model):
                    gs, gl = ctx.debug_block(gb); gb += 1
                    if gs != ms or gl != ml:
                        i = next((i for i in range(min(len(gs), len(ms))) if gs[i] != ms[i]), min(len(gs), len(ms)))
                        print(f"  seg@{s0} blk {b}: nseq gpu {len(gs)} model {len(ms)} first seq diff {i}: gpu {gs[i:i+3]} model {ms[i:i+3]}; nlit gpu {len(gl)} model {len(ml)} litdiff {first_diff(gl, ml)}")
                        pos = sum(x[0] + x[1] for x in ms[:i])
                        print(f"     position of first differing sequence ~{pos} (tile {pos // 2048}, wave {(pos % 2048) // 128}, lane {pos % 64})")
                        break
                else:
                    continue
                break
            else:
                print("  LZ stage identical -> entropy stage differs")
            break
    print("RESULT", "ALL OK" if bad == 0 else f"{bad} FAILED")
    return bad

if __name__ == "__main__":
    rc = main(0)
    rc += main(0x200)   # same cases with the serial form of k_lz's end scan forced
    sys.exit(1 if rc else 0)
