#!/usr/bin/env python3
"""bench.py -- archive-create throughput of the MI355X compression path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over the whole synthetic corpus, ending with the complete `.pna` archive bytes in HBM
(LZ -> entropy -> payloads written at their archive offsets -> chunk framing + FDAT CRC-32; `--framing none` stops at the packed
compressed entries, `--framing solid` is `pna create --solid`).  Default workload = BASELINE.json configs[1]: 10 000 x 1 MiB
enwik-style text, zstd level 3, one GPU.  Inputs are generated on the device and are resident in HBM when the timed region starts.

N > 1 (configs[2]): one process per GPU (torch.distributed, backend nccl = RCCL).  Default `--scaling strong`: the SAME corpus of
`--files` entries is sharded over the ranks in contiguous index ranges (cli/src/command/core.rs:496-537 fan-out semantics), every
rank compresses its shard with no data-path collective, and the compressed shards are gathered in index order onto rank 0 over RCCL
(the ordered gather of the serial PNA stream).  A rank's shard is cut into `--gather-pieces` contiguous pieces (2 when N > 1): piece h
of all ranks forms the h-th stretch of the archive, so the gather of piece h runs while piece h + 1 is being compressed and only the
last piece's gather is exposed.  `--scaling weak` gives every rank `--files` entries of its own.  After the timed region the SURVEY
§8(e) comparison path is timed as well: every rank copies its pieces D2H straight into a pre-offset page-locked host buffer shared by
the ranks (`gather_compare`).

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries `roofline` (the dominant kernel -- k_lzm, the match kernel of the
split LZ stage, or k_lz when the batch went through the one-kernel form -- against the HBM peak), `cpu_baseline` (the
reference's pipeline restated on the host cores, for every --algo / --framing), and at N = 1 `end_to_end`: the same corpus from
PAGEABLE host memory through pna_gpu_create_archive_host to a counting sink (SURVEY §8(d)'s wall-clock metric, PCIe included).
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # what streaming kernels reach of it on MI355X (same guide, HBM section)
def encoder_text(algo: str, level: int, entry_bytes: int = 1 << 20) -> str:
    """The level set behind the (clamped) `level` in words (pna_host.cpp level_flags / set_call_level)."""
    defl = algo == "deflate"
    if not (defl and level == 0) and entry_bytes <= 16384:
        # segments of at most 16 KiB: the short-segment geometry, whatever the level's table
        strong = level >= 9 if defl else (level >= 3 or level == 0)
        fast = level <= 3 if defl else (level < 0 or level == 1)
        high = not defl and level >= 4
        parse = "greedy+lazy3" if fast else ("greedy+lazy3 with backward adoption (4 rounds, 15 back bytes)" if high else "greedy+lazy3 with backward adoption (3 rounds, 7 back bytes)" if strong else "greedy+lazy3 with backward adoption (2 rounds)")
        return (f"GPU encoder, short-segment geometry: one wave per entry, 2048-entry LDS hash table" + ("" if fast else " over the even positions") +
                f", the whole entry as look-back, look-ups and inserts alternating per 256 positions, min_match 6, {parse}, 4096-position parse tiles")
    if defl and level == 0:
        return "GPU encoder: level 0 = Compression::none(): stored blocks only"
    fast = level <= 3 if defl else (level < 0 or level == 1)
    gtab = not defl and level >= 10
    strong = level >= 9 if defl else (level >= 3 or level == 0)
    w16 = not defl and not fast and not gtab and level >= 4
    packed = not defl and not fast and not gtab
    if gtab:
        table = "2^19-slot hash table per segment in global memory over the even positions"
    elif packed:
        table = ("55206" if w16 else "49062") + "-slot PACKED LDS hash table (three 21-bit entries per 64-bit word) over the even positions"
    else:
        table = "24512-entry LDS hash table" + ("" if fast else " over the even positions")
    look = ("look-back 32 KiB (inside the 64 KiB LDS window)" if defl else
            "look-back = the LDS window (56 064 B)" if fast else
            "look-back = the whole 1 MiB segment (the match kernel verifies candidates up to 61 136 B back in its LDS window, older ones in HBM/L2)" if gtab else
            "look-back 512 KiB of the segment (the match kernel verifies candidates up to %s B back in its LDS window, older ones in HBM/L2)" % ("11 984" if w16 else "28 368"))
    high = not defl and level >= 4                                   # FLAG_STRONG2: the packed 16 KiB geometry and the global table
    far1 = packed and not w16                                       # FLAG_FAR1: the light and default sets
    parse = "greedy+lazy3" if fast else ("greedy+lazy3 with backward adoption (4 rounds, 15 back bytes)" if high else "greedy+lazy3 with backward adoption (3 rounds, 7 back bytes)" if strong else "greedy+lazy3 with backward adoption (2 rounds)")
    if far1:
        look += ", at most 63 of those per wave of 256 positions"
    return f"GPU encoder: {table}, {look}, min_match 6, {parse}, a match cut by the merge keeps 6 bytes, 4096-position tiles"


def usable_cores() -> int:
    """Logical CPUs this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def size_label(n: int) -> str:
    for unit, s in ((1 << 30, "GiB"), (1 << 20, "MiB"), (1 << 10, "KiB")):
        if n >= unit and n % unit == 0:
            return f"{n // unit} {s}"
    return f"{n} B"


def recorded_traffic(n_files: int, file_len: int, algo: str, kind: int, framing: str, kernel: str):
    """HBM bytes of the dominant kernel's launches of one step (same span as `algorithmic_bytes`) from the PMC passes committed under
    profiles/ (scripts/pmc_traffic.sh); counters cannot be read from inside this process, so the figure is only reported for the exact
    workload and kernel it was measured on, else null."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        w = d["workload"]
        if (w["files"], w["file_bytes"], w["algo"], w["kind"], w["framing"]) == (n_files, file_len, algo, kind, framing):
            return d[kernel]["hbm_bytes_per_step"]
    except Exception:
        pass
    return None


def recorded_issue(kernel: str):
    """Issue-side counters of the dominant kernel from the committed SQ pass (profiles/r05_sq_issue.json, written by scripts/pmc_sq.sh on the GPU box):
    VALU-active share of a wave's cycles x waves per SIMD = share of the SIMD's cycles with a vector instruction in flight, the LDS unit's share likewise.
    Not measured inside this run (counters cannot be read from within the process)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r05_sq_issue.json")))
        k = d[kernel]
        return {"valu_active_per_wave_cycle": k["valu_active"], "waves_per_simd": k["waves_per_simd"], "issue_frac": round(k["valu_active"] * k["waves_per_simd"], 3),
                "lds_active_per_wave_cycle": k.get("lds_active"), "lds_unit_frac": round(k.get("lds_active", 0) * k["waves_per_simd"] * 4, 3),
                "valu_per_wave_tile": k.get("valu_per_wave_tile"), "source": "profiles/r05_sq_issue.json (rocprofv3 --pmc SQ_* passes of this build, scripts/pmc_sq.sh / pmc_insts.sh)"}
    except Exception:
        return None


def cpu_baseline(algo: str, framing: str, kind: int, file_len: int, sample_bytes: int, unique_files=None) -> dict:
    """The reference's create pipeline restated on the host cores (oracle/cpu_baseline.c), bounded sample of the same workload:
    normal archives = one entry per task on all usable cores (cli/src/command/core.rs:496-537), libzstd level 3 / zlib level 6
    streaming encoders (lib/src/entry/write.rs:257-262); --solid = ONE encoder on ONE thread (lib/src/archive/write.rs:459-463)."""
    from oracle import codec
    L = codec.lib()
    cores = usable_cores()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    # the sample's files: the workload's own -- all different, copied from the device's corpus (the same generator: oracle/corpus_model.c == k_corpus, tests) --
    # when the caller hands them over (unique_files(k) -> the first k files back to back); the generator on the host otherwise (64 files, cycled)
    want = max(1, sample_bytes // max(file_len, 1))
    data = unique_files(want) if unique_files is not None else None
    if data:
        n_unique = len(data) // file_len
    else:
        n_unique = max(1, min(64 if file_len >= (1 << 16) else 4096, want))
        data = b"".join(codec.corpus_file(kind, i, file_len) for i in range(n_unique))
    L.pna_cpu_zstd_version.restype = ctypes.c_uint
    L.pna_cpu_zlib_version.restype = ctypes.c_char_p
    zver = L.pna_cpu_zstd_version()
    lver = (L.pna_cpu_zlib_version() or b"").decode()
    if algo == "zstd" and zver == 0:
        raise RuntimeError("libzstd.so.1 is absent on this host: no CPU baseline (BASELINE.md holds the in-container figures)")
    if algo == "deflate" and not lver:
        raise RuntimeError("libz.so.1 is absent on this host")
    codec_txt = (f"host libzstd {zver // 10000}.{zver // 100 % 100}.{zver % 100} level 3 (the reference pins 1.5.7)" if algo == "zstd"
                 else f"host zlib {lver} level 6 (the reference's flate2 default backend is miniz_oxide; same format and level)")
    kind_txt = "enwik-style" if kind == 0 else "random-text"
    out = ctypes.c_uint64()
    if framing == "solid":
        n_files = max(1, (sample_bytes + file_len - 1) // file_len)
        buf = data * ((n_files + n_unique - 1) // n_unique)
        L.pna_cpu_baseline_solid.restype = ctypes.c_double
        L.pna_cpu_baseline_solid.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                             ctypes.c_int, ctypes.c_int, u64p]
        secs = L.pna_cpu_baseline_solid(buf, n_files, file_len, file_len, 73, 2 if algo == "zstd" else 1, 3 if algo == "zstd" else 6, ctypes.byref(out))
        if secs <= 0:
            raise RuntimeError(f"solid baseline failed ({secs})")
        return {"value": n_files * file_len / secs / 2**20, "unit": "MiB/s", "cores": 1, "kind": "port",
                "sample": f"{n_files} x {file_len} B {kind_txt} inner entries ({n_unique} unique) as ONE stream through ONE streaming encoder on one thread "
                          f"-- the reference's --solid path is single-threaded by construction (lib/src/archive/write.rs:459-463) --, {codec_txt}",
                "ratio": n_files * file_len / max(out.value, 1)}
    per_core = sample_bytes // max(file_len, 1)
    n_files = max(cores, per_core)
    buf = data * ((n_files + n_unique - 1) // n_unique)
    n_files = len(buf) // file_len
    fn = L.pna_cpu_baseline_zstd if algo == "zstd" else L.pna_cpu_baseline_deflate
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u64p]
    level = 3 if algo == "zstd" else 6
    secs = fn(buf, n_files, file_len, file_len, cores, level, ctypes.byref(out))
    if secs <= 0:
        raise RuntimeError(f"baseline failed ({secs})")
    out1 = ctypes.c_uint64()
    n1 = max(1, n_files // max(cores, 1) // 4)
    secs1 = fn(buf, n1, file_len, file_len, 1, level, ctypes.byref(out1))
    # the second bound: + the reference's serial tail (one thread: CRC-32 of every chunk + the write, behind the parallel phase -- the rayon scope ends
    # before drain_entry_results starts, cli/src/command/core.rs:471-537), timed on the sample's compressed size
    with_tail = tail_s = None
    try:
        L.pna_cpu_baseline_tail.restype = ctypes.c_double
        L.pna_cpu_baseline_tail.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        simd = ctypes.c_int(0)
        tail_s = L.pna_cpu_baseline_tail(int(out.value), max(1, int(out.value) // max(n_files, 1)), ctypes.byref(simd))
        if tail_s > 0:
            with_tail = n_files * file_len / (secs + tail_s) / 2**20
    except Exception:
        pass
    return {"value": n_files * file_len / secs / 2**20, "unit": "MiB/s", "cores": cores, "kind": "port",
            "sample": f"{n_files} x {file_len} B {kind_txt} files ({n_unique} unique), {codec_txt} streaming, one entry per task on {cores} threads "
                      f"(= usable cores: {cores} of the host's {os.cpu_count()} logical CPUs -- affinity / cgroup cpu.max); `value` = the parallel compression phase only "
                      f"(favours the CPU); `value_with_serial_tail` adds the reference's single-threaded re-order / CRC-32 / write tail "
                      f"(cli/src/command/core.rs:471-493) as a carry-less-multiplication CRC-32 (the reference's crc32fast) + one copy of the compressed bytes on one thread",
            "ratio": n_files * file_len / max(out.value, 1),
            "value_with_serial_tail": with_tail, "serial_tail_ms": round(tail_s * 1e3, 2) if tail_s and tail_s > 0 else None,
            "host_logical_cpus": os.cpu_count(),
            "single_thread_mib_s": n1 * file_len / secs1 / 2**20 if secs1 > 0 else None}


def end_to_end(pna, ctx, src, n_files: int, file_len: int, stride: int, names, algo: int, level: int, runs: int = 2) -> dict:
    """wall time from the first input byte in (pageable) host RAM to the last archive byte handed to the sink, through
    pna_gpu_create_archive_host (bounded window: staging || H2D || kernels || D2H).  The sink counts the bytes."""
    host = src[:n_files * stride].cpu().numpy()                  # pageable host memory, one entry per pointer
    base = host.ctypes.data
    a_names = (ctypes.c_char_p * n_files)(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * n_files)(*[base + i * stride for i in range(n_files)])
    a_len = (ctypes.c_size_t * n_files)(*[file_len] * n_files)
    count = [0, 0]

    def _sink(_u, _buf, k):
        count[0] += k
        count[1] += 1
        return 0
    cb = pna.SINK_FN(_sink)
    L = pna.load_library()
    best = None
    for it in range(runs + 1):                                   # the first run allocates the page-locked staging slots: not timed
        count[0] = count[1] = 0
        t0 = time.perf_counter()
        rc = L.pna_gpu_create_archive_host(ctx._h, algo, level, n_files, a_names, a_src, a_len, cb, None)
        dt = time.perf_counter() - t0
        if rc:
            raise RuntimeError(f"pna_gpu_create_archive_host failed: {rc}")
        if it > 0:
            best = dt if best is None else min(best, dt)
    in_bytes = n_files * file_len
    # the same with the entries in page-locked slots of the library (pna_gpu_host_alloc: the host reads its files straight into them): no staging copy,
    # one host thread issues the copies
    slots = None
    slot = None
    try:
        import numpy as np
        slot = pna.HostSlot(ctx, n_files * stride + 64)
        np.frombuffer(slot.view, dtype=np.uint8)[:n_files * stride] = host[:n_files * stride]
        s_src = (ctypes.c_void_p * n_files)(*[slot.ptr + i * stride for i in range(n_files)])
        sbest = None
        for it in range(runs + 1):
            count[0] = count[1] = 0
            t0 = time.perf_counter()
            rc = L.pna_gpu_create_archive_host(ctx._h, algo, level, n_files, a_names, s_src, a_len, cb, None)
            dt = time.perf_counter() - t0
            if rc:
                raise RuntimeError(f"pna_gpu_create_archive_host (slots) failed: {rc}")
            if it > 0:
                sbest = dt if sbest is None else min(sbest, dt)
        slots = {"value": round(in_bytes / sbest / 2**20, 1), "unit": "MiB/s", "ms": round(sbest * 1e3, 2), "h2d_GBps": round(in_bytes / sbest / 1e9, 2),
                 "path": "the same entries in page-locked slots of pna_gpu_host_alloc (a host that reads its files straight into them): no staging copy, "
                         "one host thread issues the H2D copies"}
    except Exception as e:
        slots = {"value": None, "error": repr(e)}
    finally:
        if slot is not None:
            slot.free()                                           # (page-locked: not left to the context's close when a run fails)
    return {"value": round(in_bytes / best / 2**20, 1), "unit": "MiB/s", "ms": round(best * 1e3, 2), "archive_bytes": count[0], "sink_calls": count[1],
            "from_host_slots": slots,
            "pcie_bytes": in_bytes + count[0], "pcie_GBps": round((in_bytes + count[0]) / best / 1e9, 2), "runs": runs,
            "h2d_GBps": round(in_bytes / best / 1e9, 2),
            "path": f"{n_files} x {file_len} B entries in pageable host memory -> pna_gpu_create_archive_host (sub-batches of 64 .. 256 MiB over a ring of four "
                    f"page-locked slots: staging || H2D copy engine || kernels || D2H by a throttled copy kernel) -> counting sink; best of {runs} runs after one "
                    f"untimed run.  Bound: the H2D direction of the host link (56.8 GB/s page-locked -> HBM on these boxes, profiles/r03_c_link_duplex.txt)"}


class HostGather:
    """SURVEY §8(e) comparison path: every rank copies its compressed pieces D2H straight to their final offsets in ONE page-locked host
    buffer shared by the ranks of the node (POSIX shared memory registered with the HIP runtime in every process)."""

    def __init__(self, torch, dist, rank: int, world: int, cap: int):
        from multiprocessing import shared_memory
        self.torch, self.rank = torch, rank
        name = f"pna_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}"
        if rank == 0:
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=cap)
        if world > 1:
            dist.barrier()
        if rank != 0:
            self.shm = shared_memory.SharedMemory(name=name)
            try:                                               # rank 0 owns (and unlinks) the segment: keep this process' tracker out of it
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:
                pass
        self.t = torch.frombuffer(self.shm.buf, dtype=torch.uint8)
        self.registered = int(torch.cuda.cudart().cudaHostRegister(self.t.data_ptr(), self.t.numel(), 0)) == 0

    def put(self, dst, total: int, offset: int):
        self.t[offset:offset + total].copy_(dst[:total], non_blocking=True)

    def close(self, dist, world: int):
        try:
            if self.registered:
                self.torch.cuda.cudart().cudaHostUnregister(self.t.data_ptr())
            del self.t
            self.shm.close()
            if world > 1:
                dist.barrier()
            if self.rank == 0:
                self.shm.unlink()
        except Exception:
            pass


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: run the N ranks as children of this process through torch.distributed.run (the very command the
    driver uses), on a free port of 127.0.0.1.  Nothing here touches the GPU -- a process that has initialised HIP must not exec or fork GPU users --;
    the children's stdout / stderr are inherited, so rank 0's one JSON line is this process' stdout, and the launcher's exit code is returned
    (non-zero as soon as one rank fails: torch.distributed.run ends the others)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                      # (what torchrun would set, without its warning on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def launch_check(rank: int, world: int, fail_rank: int) -> int:
    """--launch-check: the launch mechanics alone (no GPU): rendezvous over gloo, every rank reports in, rank 0 prints one JSON line.  tests/ run the
    plain command line through this on CPU-only hosts; --launch-check-fail-rank R makes rank R exit 3 to show the exit code travels."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = os.getpid()
    if world > 1:
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "ranks_seen": int((seen != 0).sum().item()),
                          "distinct_processes": len(set(seen.tolist())), "launcher": "torch.distributed.run"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 3 if rank == fail_rank else 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--files", type=int, default=10000, help="entries of the corpus (strong scaling: in all; weak scaling: per rank)")
    ap.add_argument("--file-mib", type=float, default=1.0)
    ap.add_argument("--algo", choices=["zstd", "deflate"], default="zstd")
    ap.add_argument("--level", type=int, default=None, help="compression level on the reference's scale (default: zstd 3 / deflate 6); zstd 1, "
                    "deflate 0..3 = the fast set, zstd 2 / deflate 4..5 = balanced (no lazy deferral)")
    ap.add_argument("--kind", type=int, default=0, help="corpus kind (0 enwik-style text, 1 random-text)")
    ap.add_argument("--framing", choices=["archive", "none", "solid"], default="archive",
                    help="archive: whole .pna assembled in HBM (default); none: compressed entry streams only; "
                         "solid: `pna create --solid` (BASELINE.json configs[3]: one stream, block-split in the kernels)")
    ap.add_argument("--encrypt", choices=["none", "aes-ctr", "aes-cbc", "aes-gcm"], default="none",
                    help="archive framing only: AES-256 cipher stage between compression and chunk CRC (`pna create --aes [ctr|cbc]`)")
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="N > 1: strong (default) shards the one corpus of --files entries over the ranks (BASELINE.json configs[2]); weak gives every rank --files entries")
    ap.add_argument("--gather-pieces", type=int, default=0,
                    help="archive framing: cut every rank's shard into this many pieces, each compressed and gathered on its own "
                         "(0 = 2 when N > 1, else 1)")
    ap.add_argument("--gather", choices=["lib", "torch"], default="torch",
                    help="N > 1: the ordered gather through the library's own pna_gpu_gather_ordered_start / _wait (RCCL behind the C ABI, include/pna_gpu.h: "
                         "opt-in until an N >= 2 hardware run of it is on record) or through torch.distributed (nccl = RCCL: the default); either way piece h travels "
                         "while piece h + 1 is compressed")
    ap.add_argument("--no-verify", action="store_true", help="skip the device round trip of the last step's archive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-memory-to-sink leg (N = 1, archive framing)")
    ap.add_argument("--no-gather-compare", action="store_true", help="N > 1: skip the direct-D2H comparison path")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)         # launch mechanics only (tests; no GPU)
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-mib", type=int, default=0, help="CPU baseline sample per core in MiB (0 = 256 zstd / 48 deflate; --solid: 1024 zstd / 192 deflate in all)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # the plain command line (`python bench.py --gpus N ...`, no launcher around it): this process becomes the launcher -- it has not
        # touched the GPU (torch is not even imported yet) -- and runs the N ranks as children; rank 0's JSON line is its stdout
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch it plainly (python bench.py --gpus {args.gpus} ...: it starts its own ranks) or with\n"
                 f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...")
    if args.launch_check:
        sys.exit(launch_check(rank, world, args.launch_check_fail_rank))
    import torch
    import torch.distributed as dist
    # PNA_BENCH_REHEARSAL=1: the N > 1 control flow on a box with ONE GPU -- every rank uses cuda:0 and the exchange runs over gloo on host
    # copies (RCCL refuses two ranks on one device).  Timings of such a run mean nothing; it exists to check shard layout, part flags,
    # the ordered gather and the gathered archive.
    rehearsal = bool(os.environ.get("PNA_BENCH_REHEARSAL")) and world > 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (world > 1 and not rehearsal) else 0)
    xdev = torch.device("cpu") if rehearsal else dev           # where the exchange's tensors live

    pna = importlib.import_module("portable-network-archive_amd")
    ctx = pna.Context(dev.index)

    scaling = args.scaling if args.scaling != "auto" else "strong"
    if world == 1:
        scaling = "weak"                                          # one rank: the two coincide; the contract's default word
    file_len = int(args.file_mib * (1 << 20))
    if scaling == "strong" and world > 1:
        if args.files % world:
            sys.exit(f"bench.py: --files {args.files} is not a multiple of the {world} ranks")
        n_files = args.files // world                             # entries of THIS rank
    else:
        n_files = args.files
    files_all = n_files * world
    stride = (file_len + 15) & ~15
    src = torch.empty(n_files * stride + 8192, dtype=torch.uint8, device=dev)
    algo = pna.ALGO_ZSTD if args.algo == "zstd" else pna.ALGO_DEFLATE
    lvl = pna.LEVEL_DEFAULT if args.level is None else args.level
    # pieces: piece h of rank r holds the files [(h * world + r) * n_piece, ... + n_piece) of the corpus -- in archive order all ranks'
    # pieces 0 come first, then all pieces 1, ...: every piece is gathered in rank order as soon as it is compressed
    pieces = args.gather_pieces or (2 if world > 1 else 1)
    if args.framing != "archive" or pieces < 1 or n_files % pieces:
        pieces = 1
    n_piece = n_files // pieces
    shard = importlib.import_module("portable-network-archive_amd.shard")
    first_file = [lo for lo, _ in shard.piece_ranges(rank, world, pieces, n_piece)]
    for h in range(pieces):
        ctx.corpus_fill_device(args.kind, first_file[h], n_piece, file_len, stride, src.data_ptr() + h * n_piece * stride)
    src_off = [i * stride for i in range(n_files)] + [n_files * stride]
    src_len = [file_len] * n_files
    names = [f"enwik/part{first_file[i // n_piece] + i % n_piece:07d}.txt" for i in range(n_files)]
    p_names = [names[h * n_piece:(h + 1) * n_piece] for h in range(pieces)]
    p_off = [src_off[h * n_piece:(h + 1) * n_piece] + [src_off[(h + 1) * n_piece]] for h in range(pieces)]
    p_len = [src_len[h * n_piece:(h + 1) * n_piece] for h in range(pieces)]
    cipher = None
    p_cipher = [None] * pieces
    if args.encrypt != "none":
        if args.framing != "archive":
            ap.error("--encrypt needs --framing archive")
        import hashlib
        # the key a host derives once per WriteOptions (derive_key_material); fixed salt / IV seed: the bench is deterministic
        key = hashlib.pbkdf2_hmac("sha256", b"password", b"saltsaltsalt", 1000, 32)
        iv_seed = hashlib.sha256(b"bench-ivs-%d" % rank).digest()
        mode = {"aes-ctr": pna.MODE_CTR, "aes-cbc": pna.MODE_CBC, "aes-gcm": pna.MODE_GCM}[args.encrypt]
        per = 39 if mode == pna.MODE_GCM else 16               # GCM STREAM: salt(32) || nonce_prefix(7) per entry
        ivs = b"".join((hashlib.sha256(iv_seed + i.to_bytes(4, "little")).digest() + hashlib.sha256(iv_seed + b"x" + i.to_bytes(4, "little")).digest())[:per]
                       for i in range(n_files))
        cipher = pna.Cipher(key, "$pbkdf2-sha256$i=1000,l=32$c2FsdHNhbHRzYWx0", mode, ivs=ivs)
        p_cipher = [cipher if pieces == 1 else pna.Cipher(key, "$pbkdf2-sha256$i=1000,l=32$c2FsdHNhbHRzYWx0", mode, ivs=ivs[h * n_piece * per:(h + 1) * n_piece * per])
                    for h in range(pieces)]
    if args.framing == "archive":
        dst_cap = max(pna.archive_enc_bound(algo, p_names[h], p_len[h], p_cipher[h]) for h in range(pieces))
    elif args.framing == "solid":
        dst_cap = pna.solid_archive_bound(algo, names, src_len)
    else:
        dst_cap = pna.bound(algo, file_len) * n_files + 4096
    # two output buffers when pieces are gathered: the ordered gather of one piece (RCCL send/recv over xGMI) overlaps the compression
    # of the next piece (of this step or of the next one)
    nbuf = 2 if (world > 1 or pieces > 1) else 1
    dsts = [torch.empty(dst_cap, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    arg_cache = [dict() for _ in range(pieces)]

    lib_comm = None
    gather_note = None
    if world > 1 and args.gather == "lib" and not rehearsal:
        # the library's communicator (RCCL behind the C ABI).  Should it not come up on some rank (no librccl to dlopen, an init error), every rank learns of it
        # and the run goes on with the same gather over torch.distributed instead of dying at N > 1 -- the line then says so.
        err = None
        try:
            uid = [pna.Comm.unique_id() if rank == 0 else None]
        except Exception as e:                                  # noqa: BLE001
            uid, err = [None], repr(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                lib_comm = pna.Comm(dev.index, uid[0], world, rank)
            except Exception as e:                              # noqa: BLE001
                err = repr(e)
        ok = torch.tensor([1 if lib_comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if lib_comm is not None:
                lib_comm.close()
                lib_comm = None
            errs = [None] * world
            dist.all_gather_object(errs, err)
            gather_note = "torch.distributed (the library's communicator did not come up: %s)" % next((e for e in errs if e), "unknown")
    gather_out = [None] * pieces                              # rank 0: where the pieces h of all ranks land, in rank order
    piece_sizes = [None] * pieces                             # bytes every rank contributed to piece h (from the last gather of that piece)
    pending = [None] * nbuf                                   # the gather that still reads dsts[b]
    cur = [0]
    mode = ["rccl"]                                           # "rccl": ordered gather onto rank 0's HBM; "d2h": direct copies into the shared host buffer
    host_gather = [None]
    host_pos = [0]

    def finish_gather(b=None):
        for k in (range(nbuf) if b is None else [b]):
            if pending[k] is not None:
                if isinstance(pending[k], tuple) and pending[k][0] == "lib":
                    lib_comm.gather_wait(pending[k][1])       # (the gather that read this buffer; a later one may still be travelling)
                elif pending[k] != "d2h":
                    _, buf_k, sizes_k = shard.gather_ordered_wait(pending[k][0])
                    piece_sizes[pending[k][1]] = sizes_k
                    if rank == 0 and buf_k is not None:
                        gather_out[pending[k][1]] = buf_k         # (the gather allocates a larger buffer when the ranks' pieces outgrow the one it was offered)
                torch.cuda.current_stream().synchronize()     # RCCL work.wait() only orders streams: the buffers are reused by the host-launched kernels
                pending[k] = None

    def part_flags(h):
        # the first piece of rank 0 carries the archive header, the last piece of the last rank AEND: all pieces in order are ONE archive
        return (pna.PART_HEAD if (rank == 0 and h == 0) else 0) | (pna.PART_TAIL if (rank == world - 1 and h == pieces - 1) else 0)

    lz_acc = [0.0, 0.0, 0.0, 0]                             # LZ stage ms, all stages ms, match-kernel ms, match-kernel launches

    def step():
        total_all = 0
        host_pos[0] = 0
        for h in range(pieces):
            b = cur[0]
            finish_gather(b)                                  # the gather that used this buffer two pieces ago
            dst = dsts[b]
            if args.framing == "archive":
                total, _ = ctx.create_archive_device(p_names[h], src.data_ptr(), p_off[h], p_len[h], dst.data_ptr(), dst_cap, algo=algo,
                                                     _cache=arg_cache[h], part=part_flags(h), cipher=p_cipher[h], want_offsets=False, level=lvl)
            elif args.framing == "solid":
                total = ctx.create_solid_archive_device(names, src.data_ptr(), src_off, src_len, dst.data_ptr(), dst_cap, algo=algo, _cache=arg_cache[0], level=lvl)
            else:
                total = ctx.compress_batch_device(src.data_ptr(), src_off, src_len, dst.data_ptr(), dst_cap, algo=algo, level=lvl)[-1]
            tm = ctx.timing()
            lz_acc[0] += tm.ms_lz
            lz_acc[2] += tm.ms_lz_match
            lz_acc[3] += tm.lz_match_launches
            lz_acc[1] += tm.ms_lz + tm.ms_stats + tm.ms_lit + tm.ms_seq + tm.ms_pack + tm.ms_frame + tm.ms_cipher
            if world > 1 and mode[0] == "rccl" and lib_comm is not None:
                # the library's gather: a worst-case destination (every rank's piece is below the archive bound of its entries), one call
                if rank == 0 and gather_out[h] is None:
                    gather_out[h] = torch.empty(dst_cap * world, dtype=torch.uint8, device=dev)
                sizes_h, _ = lib_comm.gather_ordered_start(dst.data_ptr(), total, gather_out[h].data_ptr() if rank == 0 else 0, dst_cap * world if rank == 0 else 0,
                                                           stream=torch.cuda.current_stream().cuda_stream)
                piece_sizes[h] = sizes_h
                pending[b] = ("lib", lib_comm.ticket())       # the transfer runs on the communicator's stream while the next piece is compressed
            elif world > 1 and mode[0] == "rccl":
                if rank == 0 and gather_out[h] is None:
                    gather_out[h] = torch.empty(int(total * world * 1.02) + (1 << 20), dtype=torch.uint8, device=xdev)
                pending[b] = (shard.gather_ordered_start(dst[:total].cpu() if rehearsal else dst, total, rank, world, out=gather_out[h]), h)
            elif mode[0] == "d2h":
                # sizes of piece h of all ranks -> this rank's offset inside the archive image on the host
                sizes_t = torch.zeros(world, dtype=torch.int64, device=xdev)
                if world > 1:
                    dist.all_gather_into_tensor(sizes_t, torch.tensor([total], dtype=torch.int64, device=xdev))
                else:
                    sizes_t[0] = total
                sizes = [int(x) for x in sizes_t.tolist()]
                host_gather[0].put(dst, total, host_pos[0] + sum(sizes[:rank]))
                host_pos[0] += sum(sizes)
                pending[b] = "d2h"
            if nbuf > 1:
                cur[0] ^= 1
            total_all += total
        return total_all

    def timed(n_steps):
        finish_gather()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        lz_acc[0] = lz_acc[1] = lz_acc[2] = 0.0
        lz_acc[3] = 0
        t0 = time.perf_counter()
        out = 0
        for _ in range(n_steps):
            out = step()
        finish_gather()                                       # the last piece's gather is inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    for _ in range(args.warmup):
        step()
    dt, out_total = timed(args.steps)
    lz_ms, stage_ms, lzm_ms, lzm_launches = lz_acc
    if world > 1:
        o = torch.tensor([out_total], dtype=torch.int64, device=xdev)
        dist.all_reduce(o)
        out_all = int(o.item())
    else:
        out_all = out_total
    tm_last = ctx.timing()                                   # stage split of the last timed launch (the checks below run more kernels)

    # ---- outside the timed region: SURVEY §8(e)'s comparison path (N > 1; available at N = 1 with --gather-pieces for rehearsal)
    gather_compare = None
    if args.framing == "archive" and not args.no_gather_compare and (world > 1 or args.gather_pieces > 1):
        try:
            cap_host = int(out_all * 1.05) + (4 << 20)
            host_gather[0] = HostGather(torch, dist, rank, world, cap_host)
            mode[0] = "d2h"
            step()                                             # untimed: first touch of the shared pages
            dt2, _ = timed(args.steps)
            gather_compare = {"rccl_gather_to_rank0_hbm_ms_per_step": round(dt / args.steps * 1e3, 3),
                              "direct_d2h_to_shared_pinned_host_ms_per_step": round(dt2 / args.steps * 1e3, 3),
                              "host_buffer_registered": bool(host_gather[0].registered),
                              "note": "the second path ends with the archive image in HOST memory (where the sink is), the first in rank 0's HBM"}
        except Exception as e:                                 # never lose the headline number because the comparison leg failed
            gather_compare = {"error": repr(e)}
        finally:
            mode[0] = "rccl"
            finish_gather()
            if host_gather[0] is not None:
                host_gather[0].close(dist, world)

    # ---- N > 1: the GATHERED archive (all ranks' pieces in index order, as rank 0 holds it) goes through the extract driver: every name in
    # the corpus order, every length, every 64th entry byte for byte against a freshly generated copy of that corpus file
    gathered_ok = None
    if world > 1 and rank == 0 and args.framing == "archive" and args.encrypt == "none" and not args.no_verify and all(x is not None for x in piece_sizes):
        import numpy as np
        arc = b"".join(gather_out[h][:sum(piece_sizes[h])].cpu().numpy().tobytes() for h in range(pieces))
        order = [lo + i for h in range(pieces) for r in range(world) for lo, _ in [shard.piece_ranges(r, world, pieces, n_piece)[h]] for i in range(n_piece)]
        scratch = torch.empty(stride + 64, dtype=torch.uint8, device=dev)
        good = [0]

        def _gcb(_u, idx, name, kind, data, ln):
            ok1 = idx < len(order) and name.decode() == f"enwik/part{order[idx]:07d}.txt" and kind == 0 and ln == file_len
            if ok1 and idx % 64 == 0 and ln:
                ctx.corpus_fill_device(args.kind, order[idx], 1, file_len, stride, scratch.data_ptr())
                got = np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_ubyte)), shape=(ln,))
                ok1 = bool(np.array_equal(got, scratch[:ln].cpu().numpy()))
            good[0] += 1 if ok1 else 0
            return 0
        gcb = pna.ENTRY_FN(_gcb)
        rcg = ctx._L.pna_gpu_extract_archive_host(ctx._h, arc, len(arc), None, 0, gcb, None)
        gathered_ok = bool(rcg == 0 and good[0] == files_all)
        del arc

    # ---- decode this rank's last archive on the device and compare with the inputs
    verified = None
    if args.framing == "archive" and not args.no_verify and args.encrypt in ("none", "aes-ctr"):
        ok = True
        fs = max(1, (file_len.bit_length() + 7) // 8) if file_len else 0          # fSIZ payload: minimal big-endian
        extra = (12 + len(cipher.phsf.encode()) + 28) if cipher is not None else 0    # PHSF chunk + FDAT(iv) chunk
        back = torch.empty(n_piece * stride + 64, dtype=torch.uint8, device=dev)
        for h in range(pieces):
            total, eoff = ctx.create_archive_device(p_names[h], src.data_ptr(), p_off[h], p_len[h], dsts[0].data_ptr(), dst_cap, algo=algo,
                                                    _cache=arg_cache[h], part=part_flags(h), cipher=p_cipher[h], level=lvl)
            pay_off, pay_len = [], []
            for i in range(n_piece):
                pre = 12 + 6 + len(p_names[h][i].encode()) + 12 + fs + extra + 8
                pay_off.append(eoff[i] + pre); pay_len.append(eoff[i + 1] - eoff[i] - pre - 16)
            if cipher is not None:                            # read side: CTR decrypt in place, then decode
                ctx.cipher_apply_device(p_cipher[h], dsts[0].data_ptr(), pay_off, pay_len, decrypt=True)
            ctx.decompress_batch_device(dsts[0].data_ptr(), pay_off, pay_len, back.data_ptr(), [i * stride for i in range(n_piece)], p_len[h], algo=algo)
            ref = src[h * n_piece * stride:(h + 1) * n_piece * stride]
            ok = ok and all(bool(torch.equal(back[i * stride:i * stride + file_len], ref[i * stride:i * stride + file_len])) for i in range(0, n_piece, max(1, n_piece // 64))) \
                and (stride != file_len or bool(torch.equal(back[:n_piece * stride], ref)))
        verified = bool(ok)
        del back
    elif args.framing == "solid" and not args.no_verify:
        # the archive goes back through the extract driver: SDAT CRCs on the device, frames counted, open-size decode, inner records walked,
        # inner FDAT CRCs on the device; every name / length is checked, every 64th entry (and the last) byte for byte
        import numpy as np
        total = ctx.create_solid_archive_device(names, src.data_ptr(), src_off, src_len, dsts[0].data_ptr(), dst_cap, algo=algo, _cache=arg_cache[0], level=lvl)
        arc = dsts[0][:total].cpu().numpy().tobytes()
        good = [0]
        sample = set(range(0, n_files, 64)) | {n_files - 1}

        def _cb(_u, idx, name, kind, data, ln):
            ok1 = idx < n_files and name.decode() == names[idx] and kind == 0 and ln == file_len
            if ok1 and idx in sample and ln:
                got = np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_ubyte)), shape=(ln,))
                ok1 = bool(np.array_equal(got, src[src_off[idx]:src_off[idx] + ln].cpu().numpy()))
            good[0] += 1 if ok1 else 0
            return 0
        cbf = pna.ENTRY_FN(_cb)
        rc = ctx._L.pna_gpu_extract_archive_host(ctx._h, arc, len(arc), None, 0, cbf, None)
        verified = bool(rc == 0 and good[0] == n_files)
        del arc

    in_rank = n_files * file_len
    in_all = in_rank * world
    e2e = None
    if world == 1 and args.framing == "archive" and args.encrypt == "none" and not args.no_end_to_end:
        try:
            dsts.clear()
            gather_out[:] = [None] * pieces
            torch.cuda.empty_cache()
            e2e = end_to_end(pna, ctx, src, n_files, file_len, stride, names, algo, lvl)
        except Exception as e:
            e2e = {"value": None, "error": repr(e)}
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = in_all / (dt / args.steps) / 2**20
        # roofline of the dominant kernel.  Split LZ stage (the default for runs of >= 1 024 segments): k_lzm, the match kernel, launched once per
        # run of <= 4 GiB of input; its launches of a step together see every input byte, so Σ algorithmic bytes / Σ launch time = the step's
        # algorithmic bytes / the step's k_lzm time.  One-kernel form: k_lz, as before.
        alg_bytes = in_rank + out_total                      # each input byte read once + each output byte written once
        split = lzm_launches > 0 and lzm_ms > 0
        dom_ms_step = (lzm_ms if split else lz_ms) / args.steps
        dom_launches = (lzm_launches / args.steps) if split else float(pieces)
        achieved = alg_bytes / (dom_ms_step / 1e3) / 1e9 if dom_ms_step > 0 else 0.0
        tm = tm_last
        level = pna.clamp_level(algo, lvl)
        wl = (f"{files_all} x {size_label(file_len)}" if args.framing != "solid" else f"--solid, one {size_label(files_all * file_len)} stream of {files_all} x {size_label(file_len)} entries")
        line = {
            "metric": f"archive-create MiB/s (input bytes/sec), {args.algo}-{level}, {wl} {'enwik-style' if args.kind == 0 else 'random-text'} corpus"
                      + (f", {args.encrypt}" if args.encrypt != "none" else "") + ("" if args.framing != "none" else ", compressed streams only (no container)"),
            "value": round(value, 1), "value_hbm_resident": round(value, 1), "value_kind": "hbm_resident (inputs in HBM when the timed region starts, archive bytes left in HBM; the host-RAM-to-sink rate of "
                                                    "SURVEY 8(d) is `end_to_end`)",
            "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"pna create, {files_all} x {file_len} B synthetic {'enwik-style' if args.kind == 0 else 'random'} text"
                                   + (f" ({n_files} per GPU, contiguous index ranges)" if world > 1 else "")
                                   + f", Compression::{'ZStandard' if args.algo == 'zstd' else 'Deflate'} level {level} ({encoder_text(args.algo, level, file_len if args.framing != 'solid' else 1 << 20)}), inputs resident in HBM, "
                                   + ("output = complete .pna archive bytes in HBM (chunk framing + CRC-32 on device)" if args.framing == "archive"
                                      else "--solid: inner STORE records serialised + one compressed stream + SDAT framing, all in HBM" if args.framing == "solid"
                                      else "output = packed compressed entry streams in HBM")
                                   + ("; compressed shards gathered in index order onto rank 0 over RCCL" if world > 1 else ""),
                       "entries": files_all, "entries_per_gpu": n_files, "entry_bytes": file_len, "parallelism": f"entry-sharded x{world}",
                       "gather_pieces": pieces,
                       "gather": (gather_note or ("library (RCCL behind the C ABI: pna_gpu_gather_ordered_start / _wait)" if lib_comm is not None else "torch.distributed")) if world > 1 else None},
            "ratio": round(in_all / max(out_all, 1), 4),
            # N = 1: what ONE rank of the 8-rank strong-scaling run (configs[2]) would hand to the ordered gather per piece (2 pieces per rank): this step's
            # archive bytes / 16 -- to be held against xGMI's ~153 GB/s per link
            "gather_piece_bytes_at_8_ranks": (out_all // 16) if world == 1 and args.framing == "archive" else None,
            "verified": verified,                        # rank 0's archive decoded on the device == its inputs (None: not checked)
            "gathered_archive_verified": gathered_ok,    # N > 1: the archive gathered on rank 0 read back through the extract driver
            # (entries of at most 16 KiB: the match kernel of the short-segment geometry, k_lzms, takes every segment)
            "roofline": {"bound": "hbm", "kernel": ("k_lzms" if (file_len <= 16384 and args.framing != "solid") else "k_lzm") if split else "k_lz", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 5),   # against the ~6.3 TB/s a streaming kernel reaches on this part
                         "issue": recorded_issue("k_lzm" if split else "k_lz"),           # what the kernel is bound by instead: vector issue + LDS (committed SQ counters)
                         "traffic": recorded_traffic(n_files, file_len, args.algo, args.kind, args.framing, "k_lzm" if split else "k_lz"),
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on this build's LZ kernels "
                                           "(scripts/pmc_traffic.sh); not measured inside this run -- counters cannot be read from within the process",
                         "algorithmic_bytes": alg_bytes,                         # per step = over the kernel's launches of one step
                         "launches_per_step": round(dom_launches, 2),
                         "kernel_ms": round(dom_ms_step / max(dom_launches, 1e-9), 3),   # average launch duration (HIP events on the launch stream)
                         "kernel_ms_per_step": round(dom_ms_step, 3),
                         "lz_stage_ms": round(lz_ms / args.steps, 3),           # match + parse kernels (k_lzm + k_lzp) of a step
                         "achieved_lz_stage": round(alg_bytes / max(lz_ms / args.steps / 1e3, 1e-12) / 1e9, 2),
                         "all_kernels_ms": round(stage_ms / args.steps, 3)},
            # the last launch of the last step: with gather_pieces = P that is one piece, 1 / P of a step
            "stages_ms_last_step": {"lz": round(tm.ms_lz, 3), "stats": round(tm.ms_stats, 3), "lit": round(tm.ms_lit, 3),
                                    "seq": round(tm.ms_seq, 3), "pack": round(tm.ms_pack, 3), "frame": round(tm.ms_frame, 3)},
        }
        if cipher is not None:
            line["config"]["workload"] += f", cipher stage {args.encrypt} (AES-256) on the compressed payloads in HBM"
            line["stages_ms_last_step"]["cipher"] = round(tm.ms_cipher, 3)
            line["cipher"] = {"mode": args.encrypt, "ms": round(tm.ms_cipher, 3),
                              "GB_per_s": round(out_total / pieces / max(tm.ms_cipher, 1e-9) / 1e6, 1)}
        if gather_compare is not None:
            line["gather_compare"] = gather_compare
        if e2e is not None:
            line["end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = usable_cores()
                if args.framing == "solid":
                    sample = (args.cpu_sample_mib or (1024 if args.algo == "zstd" else 192)) << 20
                else:
                    sample = (args.cpu_sample_mib or (256 if args.algo == "zstd" else 48)) * cores << 20
                def unique_files(k):                            # the first k files of this rank's corpus (device -> host), back to back
                    k = min(k, n_files)
                    t = src[:k * stride].view(k, stride)[:, :file_len].contiguous() if stride != file_len else src[:k * file_len]
                    return t.cpu().numpy().tobytes()
                line["cpu_baseline"] = cpu_baseline(args.algo, args.framing, args.kind, file_len, sample, unique_files)
                if e2e is not None and e2e.get("value") and line["cpu_baseline"].get("value"):
                    line["end_to_end"]["vs_cpu_baseline"] = round(e2e["value"] / line["cpu_baseline"]["value"], 2)
                    # SURVEY 8(d)'s metric beside the contract's `value` (which stays the HBM-resident rate: the bench contract forbids a PCIe-inclusive `value`
                    # and a `vs_baseline` without a published number): host RAM -> sink, and its ratio to the CPU pipeline timed on this box, both bounds
                    line["value_end_to_end"] = e2e["value"]
                    line["vs_cpu_baseline_end_to_end"] = line["end_to_end"]["vs_cpu_baseline"]
                    if line["cpu_baseline"].get("value_with_serial_tail"):
                        line["vs_cpu_baseline_with_serial_tail_end_to_end"] = round(e2e["value"] / line["cpu_baseline"]["value_with_serial_tail"], 2)
            except Exception as e:  # never lose the GPU number because the CPU leg failed
                line["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()                                        # rank 0 may still be reading the gathered archive back
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
