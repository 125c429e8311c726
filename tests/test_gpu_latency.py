"""LATENCY MODE of the device encoder: a small batch -- the CompressionWriter seam (lib/src/compress.rs:32-41) with as
many writers in flight as the reference's rayon pool has threads (cli/src/command/core.rs:505-517), or one entry of a batch call -- is cut
into small blocks inside the same frames and into LZ units (one workgroup each, hash table pre-warmed with everything before the unit), so
that its time is not one workgroup's walk over a whole segment plus one lane's walk over a 128 KiB block's sequences.  The device must equal
the oracle's model run with the block size the device reports (units do not change a byte: pre-warming reproduces the segment-long walk).
"""
import random
import threading
import zlib

import pytest

pytestmark = pytest.mark.gpu

LEVELS_Z = (1, 2, 3, 7, 19)            # zstd levels of every set: fast, light, default, high, max (codec.product_level_flags)


def _ents(codec):
    rnd = random.Random(99)
    rb = lambda n: bytes(rnd.getrandbits(8) for _ in range(n))
    text = codec.corpus_file(0, 4343, 1 << 20)
    return {
        "empty": b"", "one": b"a", "tiny8": b"abcdefgh", "zeros5000": bytes(5000), "txt4k": codec.corpus_file(1, 9, 4096),
        "blk13-1": codec.corpus_file(0, 21, 8191), "blk13": codec.corpus_file(0, 21, 8192), "blk13+1": codec.corpus_file(0, 21, 8193),
        "blk14+1": codec.corpus_file(0, 22, 16385), "txt64k": codec.corpus_file(1, 2, 65536), "txt300k": codec.corpus_file(0, 1, 300000),
        "blk17": codec.corpus_file(0, 11, 131072), "txt1m": text, "seg+1": codec.corpus_file(0, 12, (1 << 20) + 1),
        "txt2m+": codec.corpus_file(0, 4, (2 << 20) + 12345), "zero1m": bytes(1 << 20), "x300k": b"x" * 300000,
        "rnd300k": codec.corpus_file(2, 0, 300000),
        # far candidates across unit borders, long runs across block borders, matches at block / unit ends
        "far": text[:300000] + rb(70000) + text[1000:250000] + rb(1000) + text[123:200123],
        "runs": rb(100000) + bytes(200000) + rb(60000) + bytes(200000) + b"ab" * 50000,
        "ends": rb(16384 - 50) + text[:100] + rb(70000) + text[:100] + rb(32768 - 150) + text[:100],
        "tail": text[:4096 * 3 + 17],
    }


@pytest.mark.parametrize("blk_log,unit_log", [(14, 14), (13, 13), (13, 16), (15, 15), (16, 18), (14, 20), (17, 17), (17, 14)])
def test_latency_mode_equals_the_model(pna, codec, blk_log, unit_log):
    """Every block size the mode may choose, with units of one block, of several blocks and of a whole segment (unit_log 20: no units), and
    units without smaller blocks (17, 17); (17, 14) asks for units below the block size and gets units of one block.  All level sets, both
    codecs; independent decoders on everything."""
    import torch  # noqa: F401
    ents = _ents(codec)
    names = sorted(ents)
    data = [ents[k] for k in names]
    with pna.Context(0) as ctx:
        ctx.set_option("blk_log", blk_log)
        ctx.set_option("unit_log", unit_log)
        if blk_log == 17:
            ctx.set_option("latency_max_mib", 0)            # (17 = "no block size asked for": keep the mode from choosing a smaller one for this small batch)
        for level in LEVELS_Z:
            outs = ctx.compress_batch(data, level=level)
            t = ctx.timing()
            assert t.blk_log == blk_log and (t.lz_units > 0) == (unit_log < 20), (t.blk_log, t.lz_units)
            pz = codec.params_for_level(level, blk_log=blk_log)
            for k, d, o in zip(names, data, outs):
                assert o == codec.model_compress(d, pz), (k, level)
                assert len(o) <= ctx._L.pna_gpu_bound(2, len(d)), k
                if level == 3:
                    assert codec.zstd_decompress(o, len(d)) == d, k
                    if codec.system_libzstd() is not None:
                        assert codec.libzstd_decompress_stream(o, len(d)) == d, k
            if level == 3:
                assert ctx.decompress_batch(outs, [len(d) for d in data]) == data          # the device decoder reads them as well
        for level, fl in ((1, codec.F_LAZY), (6, codec.F_ADOPT | codec.F_INS2 | codec.F_LAZY), (9, codec.F_ADOPT | codec.F_INS2 | codec.F_LAZY | codec.F_STRONG)):
            outs = ctx.compress_batch(data, algo=pna.ALGO_DEFLATE, level=level)
            pd = codec.params_for_flags(fl, deflate=True, blk_log=blk_log)
            for k, d, o in zip(names, data, outs):
                assert o == codec.deflate_model_compress(d, pd), (k, level)
                assert zlib.decompress(o) == d, k
                assert len(o) <= ctx._L.pna_gpu_bound(1, len(d)), k
            if level == 6:
                assert ctx.decompress_batch(outs, [len(d) for d in data], algo=pna.ALGO_DEFLATE) == data


def test_units_do_not_change_a_byte(pna, codec):
    """Units only spread a segment's LZ stage over workgroups: with the block size fixed the streams must not depend on the unit size."""
    import torch  # noqa: F401
    ents = _ents(codec)
    data = [ents[k] for k in sorted(ents)]
    with pna.Context(0) as ctx:
        ctx.set_option("blk_log", 14)
        ref = None
        for ul in (20, 18, 16, 14):
            ctx.set_option("unit_log", ul)
            outs = (ctx.compress_batch(data), ctx.compress_batch(data, algo=pna.ALGO_DEFLATE))
            ref = ref or outs
            assert outs == ref, ul


def test_the_librarys_own_choice(pna, codec, monkeypatch):
    """The default context: one 1 MiB entry runs in 64 units of 16 KiB blocks, a batch of 40 MiB in coarser ones (32 KiB blocks, units that fit one round of the
    chip's CUs), a batch beyond the mode's limit in whole segments -- with blocks that still follow the call's size (zstd: 16 KiB up to 32 MiB of input, 32 KiB up to
    384 MiB, 64 KiB up to 1 GiB, 128 KiB beyond: the headline) --, and with latency_max_mib = 0 in 128 KiB blocks whatever the batch; the model with the reported
    block size reproduces each.  Deflate: 64 KiB blocks between 64 and 384 MiB of input."""
    import torch  # noqa: F401
    one = codec.corpus_file(0, 700, 1 << 20)
    with pna.Context(0) as ctx:
        (o,) = ctx.compress_batch([one])
        t = ctx.timing()
        assert (t.blk_log, t.lz_units) == (14, 64)
        assert o == codec.model_compress(one, codec.params_for_level(3, blk_log=14))
        ents = [codec.corpus_file(0, 701 + i, 1 << 20) for i in range(40)]
        outs = ctx.compress_batch(ents)
        t = ctx.timing()
        assert t.blk_log == 15 and 0 < t.lz_units <= 256, (t.blk_log, t.lz_units)
        pz = codec.params_for_level(3, blk_log=t.blk_log)
        for e, o in zip(ents[:4], outs[:4]):
            assert o == codec.model_compress(e, pz)
        ctx.set_option("latency_max_mib", 8)
        outs2 = ctx.compress_batch(ents[:12])
        t = ctx.timing()
        assert (t.blk_log, t.lz_units) == (14, 0)
        assert outs2[0] == codec.model_compress(ents[0], codec.params_for_level(3, blk_log=14))
        outs3 = ctx.compress_batch(ents)                              # 40 MiB, beyond the mode's limit: 32 KiB blocks, whole segments
        t = ctx.timing()
        assert (t.blk_log, t.lz_units) == (15, 0)
        assert outs3[1] == codec.model_compress(ents[1], codec.params_for_level(3, blk_log=15))
        from oracle import pna_format as pf
        arc = pna.create_archive(ctx, ["m%02d" % i for i in range(len(ents))], ents)    # the archive path follows the same rule: its payloads are the batch call's streams
        assert ctx.timing().blk_log == 15
        items = pf.read_archive(arc)[1]
        assert [it.data for it in items] == outs3
        assert [d for _, _, d in pna.extract_archive(ctx, arc)] == ents
        douts = ctx.compress_batch(ents[:12], algo=pna.ALGO_DEFLATE)   # deflate outside the mode: 64 KiB blocks up to 384 MiB of input (every dynamic block repeats the code description: larger than zstd's)
        assert ctx.timing().blk_log == 16 and zlib.decompress(douts[0]) == ents[0]
        assert douts[0] == codec.deflate_model_compress(ents[0], codec.params_for_level(6, deflate=True, blk_log=16))
        ctx.set_option("latency_max_mib", 0)
        outs4 = ctx.compress_batch(ents[:12])
        t = ctx.timing()
        assert (t.blk_log, t.lz_units) == (17, 0)
        assert outs4[0] == codec.model_compress(ents[0], codec.params_for_level(3))


def test_compression_writers_in_latency_mode(pna, codec, monkeypatch):
    """The seam itself (pna_gpu_stream_*: lib/src/compress.rs:32-41,66-75) under the reference's threading model with the library's default
    settings: 16 writer threads, each stream must equal the model at the block size of the batch that carried it -- a batch of up to 16 MiB
    runs on 16 KiB blocks, so that is the only candidate here -- and decode."""
    import torch  # noqa: F401
    ents = [codec.corpus_file(0, 900 + i, (1 << 20) - 1000 * (i % 5)) for i in range(32)] + [b"", b"x" * 100]
    results = [None] * len(ents)

    class Sink:
        def __init__(self): self.parts = []
        def write(self, b): self.parts.append(bytes(b))

    with pna.Context(0) as ctx:
        def work(t):
            for i in range(t, len(ents), 16):
                s = Sink()
                w = ctx.writer(s)
                w.write(ents[i])
                w.try_into_inner()
                results[i] = b"".join(s.parts)
        th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
        for x in th: x.start()
        for x in th: x.join()
        assert ctx.timing().blk_log in (13, 14)      # (13: a last batch of the two small entries only -- every entry of up to 64 KiB is one block whatever the block size)
    pz = codec.params_for_level(3, blk_log=14)
    for e, r in zip(ents, results):
        assert r == codec.model_compress(e, pz)
        assert codec.zstd_decompress(r, len(e)) == e
