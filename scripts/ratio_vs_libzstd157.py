"""Ratio of the encoder model (== the device output, bit-exact) next to libzstd 1.5.7 -- the version the reference pins (Cargo.lock:3547-3572) --
and the system libzstd, level 3, STREAMING (ZSTD_compressStream2 continue + end, no pledged size: what zstd-rs' Encoder does), on the
benchmark corpus.  Runs in the build container (Pillow bundles libzstd 1.5.7); the GPU box only has 1.4.8.   python scripts/ratio_vs_libzstd157.py [files]"""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import codec

class Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]

def stream_compress(Z, data, level):
    Z.ZSTD_createCCtx.restype = ctypes.c_void_p
    Z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    Z.ZSTD_compressStream2.restype = ctypes.c_size_t
    Z.ZSTD_compressStream2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    Z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    c = Z.ZSTD_createCCtx(); Z.ZSTD_CCtx_setParameter(c, 100, level)
    src = ctypes.create_string_buffer(data, len(data)); dst = ctypes.create_string_buffer(len(data) + 4096)
    i = Buf(ctypes.cast(src, ctypes.c_void_p), len(data), 0); o = Buf(ctypes.cast(dst, ctypes.c_void_p), len(data) + 4096, 0)
    Z.ZSTD_compressStream2(c, ctypes.byref(o), ctypes.byref(i), 0)
    while Z.ZSTD_compressStream2(c, ctypes.byref(o), ctypes.byref(i), 2) != 0:
        pass
    Z.ZSTD_freeCCtx(c)
    return dst.raw[:o.pos]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
files = [codec.corpus_file(0, i, 1 << 20) for i in range(n)]
tot = sum(map(len, files))
libs = {"system": "/usr/lib/x86_64-linux-gnu/libzstd.so.1"}
p157 = glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so.1.5.7")
if p157:
    libs["1.5.7 (Pillow)"] = p157[0]
for name, path in libs.items():
    Z = ctypes.CDLL(path); Z.ZSTD_versionNumber.restype = ctypes.c_uint
    out = [stream_compress(Z, f, 3) for f in files]
    assert codec.zstd_decompress(out[0], 1 << 20) == files[0]
    print(f"libzstd {Z.ZSTD_versionNumber()} [{name}] level 3 streaming: ratio {tot / sum(map(len, out)):.4f}, header {out[0][:6].hex()}")
    if name.startswith("1.5.7"):
        for lv in (1, 2, 5, 7, 9):
            print(f"    ... level {lv}: ratio {tot / sum(len(stream_compress(Z, f, lv)) for f in files):.4f}")
for lv, nm in ((1, "fast"), (2, "light"), (3, "default"), (7, "high"), (19, "max")):
    p = codec.params_for_level(lv)
    print(f"encoder model, zstd {lv} ({nm}: {p.hash_log} slots{' packed' if p.tab3 else ''}, rounds {p.rounds:#x}): ratio {tot / sum(len(codec.model_compress(f, p)) for f in files):.4f}")
import zlib
d = [codec.deflate_model_compress(f) for f in files]
print(f"deflate model (level 6): ratio {tot / sum(map(len, d)):.4f}; zlib {zlib.ZLIB_VERSION} level 6: {tot / sum(len(zlib.compress(f, 6)) for f in files):.4f}")
