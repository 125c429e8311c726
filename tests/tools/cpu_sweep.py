"""CPU baseline thread sweep on the GPU box's host cores (oracle/cpu_baseline.c)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import codec
L = codec.lib()
L.pna_cpu_baseline_zstd.restype = ctypes.c_double
L.pna_cpu_baseline_zstd.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
n = 64
data = b"".join(codec.corpus_file(0, i, 1 << 20) for i in range(n))
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
print("affinity:", len(os.sched_getaffinity(0)))
for th in (1, 4, 16, 32, 64, 128, 256):
    reps = max(1, th * 8 // n)
    buf = data * reps
    out = ctypes.c_uint64()
    s = L.pna_cpu_baseline_zstd(buf, n * reps, 1 << 20, 1 << 20, th, 3, ctypes.byref(out))
    print(th, "threads:", round(n * reps / s, 1), "MiB/s")
