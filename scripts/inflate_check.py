"""Device inflate against Python's zlib on a spread of streams (diagnostics; the pytest cases live in tests/test_gpu_parity.py).
Usage: python scripts/inflate_check.py [--rate]"""
import importlib, os, random, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pna = importlib.import_module("portable-network-archive_amd")


def streams():
    rnd = random.Random(7)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 9))) for _ in range(3000)]
    text = b" ".join(rnd.choice(words) for _ in range(400000))
    cases = []
    for lvl in (0, 1, 6, 9):
        cases.append((f"text-l{lvl}", text[:1 << 20], dict(level=lvl)))
    cases.append(("fixed", text[:70000], dict(level=6, strategy=zlib.Z_FIXED)))
    cases.append(("huffman-only", text[:70000], dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY)))
    cases.append(("rle", bytes([7]) * 300000 + b"xyz" * 50000, dict(level=6)))
    cases.append(("random", rnd.randbytes(200000), dict(level=6)))
    cases.append(("empty", b"", dict(level=6)))
    cases.append(("one", b"a", dict(level=6)))
    cases.append(("binary", bytes(rnd.getrandbits(8) & 0x3F for _ in range(100000)), dict(level=9)))
    cases.append(("wbits9", text[:100000], dict(level=6, wbits=9)))
    out = []
    for name, raw, kw in cases:
        co = zlib.compressobj(kw.get("level", 6), zlib.DEFLATED, kw.get("wbits", 15), 9, kw.get("strategy", zlib.Z_DEFAULT_STRATEGY))
        out.append((name, raw, co.compress(raw) + co.flush()))
    # sync-flushed pieces (stored empty blocks in the middle) and a long stored run (literal-run splitting)
    co = zlib.compressobj(6)
    z = b"".join(co.compress(text[i:i + 50000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, 400000, 50000)) + co.flush()
    out.append(("sync-flush", text[:400000], z))
    big = rnd.randbytes(3 << 20)
    out.append(("stored-3MiB", big, zlib.compress(big, 0)))
    return out


def main():
    ctx = pna.Context()
    cs = streams()
    got = ctx.decompress_batch([z for _, _, z in cs], [len(r) for _, r, _ in cs], algo=pna.ALGO_DEFLATE)
    bad = 0
    for (name, raw, z), g in zip(cs, got):
        ok = g == raw
        bad += not ok
        print(f"{name:14s} raw {len(raw):8d} z {len(z):8d} {'ok' if ok else 'MISMATCH'}")
    # this library's own encoder
    raws = [cs[0][1], cs[6][1], b"", cs[7][1]]
    comp = ctx.compress_batch(raws, algo=pna.ALGO_DEFLATE)
    back = ctx.decompress_batch(comp, [len(r) for r in raws], algo=pna.ALGO_DEFLATE)
    for r, b in zip(raws, back):
        ok = r == b; bad += not ok
        print(f"own encoder    raw {len(r):8d} {'ok' if ok else 'MISMATCH'}")
    # corrupt streams must be refused
    for name, z, raw_len in (("bad adler", cs[0][2][:-1] + bytes([cs[0][2][-1] ^ 1]), len(cs[0][1])), ("truncated", cs[0][2][:1000], len(cs[0][1])),
                             ("bad header", b"\x78\x9d" + cs[0][2][2:], len(cs[0][1])), ("wrong size", cs[0][2], len(cs[0][1]) - 1)):
        try:
            ctx.decompress_batch([z], [raw_len], algo=pna.ALGO_DEFLATE)
            print(f"{name}: accepted (BAD)"); bad += 1
        except pna.PnaGpuError as e:
            print(f"{name}: refused ({e})")
    if "--rate" in sys.argv:
        import torch
        dev = torch.device("cuda:0")
        for n, stride, kind in ((2048, 1 << 20, 0), (8192, 1 << 20, 0), (262144, 4096, 1)):
            src = torch.empty(n * stride + 8192, dtype=torch.uint8, device=dev)
            ctx.corpus_fill_device(kind, 0, n, stride, stride, src.data_ptr())
            cap = pna.bound(pna.ALGO_DEFLATE, stride) * n + 4096
            dst = torch.empty(cap, dtype=torch.uint8, device=dev)
            offs = ctx.compress_batch_device(src.data_ptr(), [i * stride for i in range(n)] + [n * stride], [stride] * n, dst.data_ptr(), cap, algo=pna.ALGO_DEFLATE)
            back = torch.empty(n * stride + 64, dtype=torch.uint8, device=dev)
            so, sl, do, rl = offs[:n], [offs[i + 1] - offs[i] for i in range(n)], [i * stride for i in range(n)], [stride] * n
            for _ in range(2):
                t0 = time.time()
                ctx.decompress_batch_device(dst.data_ptr(), so, sl, back.data_ptr(), do, rl, algo=pna.ALGO_DEFLATE)
                dt = time.time() - t0
                tm = ctx.timing()
                print(f"inflate {n} x {stride} B: {n * stride / dt / 2**30:.1f} GiB/s wall, kernels {tm.ms_lz:.2f} ms = {n * stride / tm.ms_lz / 2**30 * 1e3:.1f} GiB/s "
                      f"(walk {tm.ms_stats:.2f}, exec {tm.ms_lit:.2f})")
            print("round trip", "ok" if torch.equal(back[:n * stride], src[:n * stride]) else "MISMATCH")
            del src, dst, back
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)


main()
