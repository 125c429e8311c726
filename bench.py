#!/usr/bin/env python3
"""bench.py -- archive-create throughput of the MI355X compression path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole synthetic corpus of this rank, ending with the complete `.pna`
archive bytes in HBM (LZ -> entropy -> payloads written at their archive offsets -> chunk framing + FDAT CRC-32;
`--framing none` stops at the packed compressed entries): `--files` enwik-style text files of `--file-mib` MiB each (BASELINE.json configs[1]:
10 000 x 1 MiB, zstd).  Inputs are generated on the device and are resident in HBM when the timed region starts.
N > 1: one process per GPU (torch.distributed / RCCL), every rank compresses its own shard of the corpus (weak
scaling), then the compressed shards are gathered in rank order onto rank 0 over RCCL (the ordered gather of the
serial PNA stream).  A rank's shard is cut into `--gather-pieces` contiguous pieces (2 when N > 1): piece h of all ranks
forms the h-th stretch of the archive, so the gather of piece h runs while piece h + 1 is being compressed and only the
last piece's gather is exposed at the end of a step.  Rank 0 prints ONE JSON line.
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


def recorded_traffic(n_files: int, file_len: int, algo: str, kind: int, framing: str):
    """HBM bytes per k_lz launch from the PMC passes committed under profiles/ (scripts/pmc_traffic.sh); counters cannot be
    read from inside this process, so the figure is only reported for the exact workload it was measured on, else null."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        w = d["workload"]
        if (w["files"], w["file_bytes"], w["algo"], w["kind"], w["framing"]) == (n_files, file_len, algo, kind, framing):
            return d["k_lz"]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(sample_files: int, file_len: int) -> dict:
    """Reference pipeline restated on the host cores (oracle/cpu_baseline.c): one entry per task, libzstd level 3."""
    from oracle import codec
    L = codec.lib()
    cores = usable_cores()
    data = b"".join(codec.corpus_file(0, i, file_len) for i in range(min(sample_files, 64)))
    n_unique = len(data) // file_len
    L.pna_cpu_zstd_version.restype = ctypes.c_uint
    ver = L.pna_cpu_zstd_version()
    if ver == 0:
        # no libzstd on this host: time the oracle's own encoder model instead (single thread) and say so
        t0 = time.time()
        out = 0
        for i in range(min(4, n_unique)):
            out += len(codec.model_compress(data[i * file_len:(i + 1) * file_len]))
        dt = time.time() - t0
        n = min(4, n_unique) * file_len
        return {"value": n / dt / 2**20, "unit": "MiB/s", "cores": 1, "kind": "port",
                "sample": f"{min(4, n_unique)} x {file_len} B, oracle model encoder (libzstd absent on this host)",
                "ratio": n / max(out, 1)}
    L.pna_cpu_baseline_zstd.restype = ctypes.c_double
    L.pna_cpu_baseline_zstd.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    # replicate the unique files so that every core gets ~2 s of work (files are re-used: identical CPU cost)
    reps = max(1, (sample_files + n_unique - 1) // n_unique)
    buf = data * reps
    n_files = len(buf) // file_len
    out = ctypes.c_uint64()
    secs = L.pna_cpu_baseline_zstd(buf, n_files, file_len, file_len, cores, 3, ctypes.byref(out))
    out1 = ctypes.c_uint64()
    n1 = min(n_files, 16)
    secs1 = L.pna_cpu_baseline_zstd(buf, n1, file_len, file_len, 1, 3, ctypes.byref(out1))
    return {"value": n_files * file_len / secs / 2**20, "unit": "MiB/s", "cores": cores, "kind": "port",
            "sample": f"{n_files} x {file_len} B enwik-style files ({n_unique} unique), host libzstd {ver // 10000}.{ver // 100 % 100}.{ver % 100} "
                      f"level 3 streaming, one entry per task on {cores} threads (= usable cores: affinity / cgroup cpu.max; "
                      f"host has {os.cpu_count()} logical CPUs); parallel compression phase only -- the reference's single-threaded "
                      f"re-order / CRC-32 / write tail (cli/src/command/core.rs:471-493) is not added, which favours the CPU figure",
            "ratio": n_files * file_len / max(out.value, 1),
            "single_thread_mib_s": n1 * file_len / secs1 / 2**20}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--files", type=int, default=10000)
    ap.add_argument("--file-mib", type=float, default=1.0)
    ap.add_argument("--algo", choices=["zstd", "deflate"], default="zstd")
    ap.add_argument("--kind", type=int, default=0, help="corpus kind (0 enwik-style text, 1 random-text)")
    ap.add_argument("--framing", choices=["archive", "none", "solid"], default="archive",
                    help="archive: whole .pna assembled in HBM (default); none: compressed entry streams only; "
                         "solid: `pna create --solid` (BASELINE.json configs[3]: one stream, block-split in the kernels)")
    ap.add_argument("--encrypt", choices=["none", "aes-ctr", "aes-cbc", "aes-gcm"], default="none",
                    help="archive framing only: AES-256 cipher stage between compression and chunk CRC (`pna create --aes [ctr|cbc]`)")
    ap.add_argument("--gather-pieces", type=int, default=0,
                    help="archive framing: cut every rank's shard into this many pieces, each compressed and gathered on its own "
                         "(0 = 2 when N > 1, else 1)")
    ap.add_argument("--no-verify", action="store_true", help="skip the device round trip of the last step's archive (archive framing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-files", type=int, default=0, help="0 = 48 files per core")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    pna = importlib.import_module("portable-network-archive_amd")
    ctx = pna.Context(dev.index)

    n_files, file_len = args.files, int(args.file_mib * (1 << 20))
    stride = (file_len + 15) & ~15
    src = torch.empty(n_files * stride + 8192, dtype=torch.uint8, device=dev)
    algo = pna.ALGO_ZSTD if args.algo == "zstd" else pna.ALGO_DEFLATE
    # pieces: piece h of rank r holds the files [(h * world + r) * n_piece, ... + n_piece) of the corpus -- in archive order all ranks'
    # pieces 0 come first, then all pieces 1, ...: every piece is gathered in rank order as soon as it is compressed
    pieces = args.gather_pieces or (2 if world > 1 else 1)
    if args.framing != "archive" or pieces < 1 or n_files % pieces:
        pieces = 1
    n_piece = n_files // pieces
    shard_mod = importlib.import_module("portable-network-archive_amd.shard")
    first_file = [lo for lo, _ in shard_mod.piece_ranges(rank, world, pieces, n_piece)]
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

    shard = importlib.import_module("portable-network-archive_amd.shard")
    gather_out = [None] * pieces                              # rank 0: where the pieces h of all ranks land, in rank order
    pending = [None] * nbuf                                   # the gather that still reads dsts[b]
    cur = [0]

    def finish_gather(b=None):
        for k in (range(nbuf) if b is None else [b]):
            if pending[k] is not None:
                shard.gather_ordered_wait(pending[k])
                torch.cuda.current_stream().synchronize()     # RCCL work.wait() only orders streams: the buffers are reused by the host-launched kernels
                pending[k] = None

    def part_flags(h):
        # the first piece of rank 0 carries the archive header, the last piece of the last rank AEND: all pieces in order are ONE archive
        return (pna.PART_HEAD if (rank == 0 and h == 0) else 0) | (pna.PART_TAIL if (rank == world - 1 and h == pieces - 1) else 0)

    lz_acc = [0.0, 0.0]

    def step():
        total_all = 0
        for h in range(pieces):
            b = cur[0]
            finish_gather(b)                                  # the gather that used this buffer two pieces ago
            dst = dsts[b]
            if args.framing == "archive":
                total, _ = ctx.create_archive_device(p_names[h], src.data_ptr(), p_off[h], p_len[h], dst.data_ptr(), dst_cap, algo=algo,
                                                     _cache=arg_cache[h], part=part_flags(h), cipher=p_cipher[h], want_offsets=False)
            elif args.framing == "solid":
                total = ctx.create_solid_archive_device(names, src.data_ptr(), src_off, src_len, dst.data_ptr(), dst_cap, algo=algo, _cache=arg_cache[0])
            else:
                total = ctx.compress_batch_device(src.data_ptr(), src_off, src_len, dst.data_ptr(), dst_cap, algo=algo)[-1]
            tm = ctx.timing()
            lz_acc[0] += tm.ms_lz
            lz_acc[1] += tm.ms_lz + tm.ms_stats + tm.ms_lit + tm.ms_seq + tm.ms_pack + tm.ms_frame + tm.ms_cipher
            if world > 1:
                if rank == 0 and gather_out[h] is None:
                    gather_out[h] = torch.empty(int(total * world * 1.02) + (1 << 20), dtype=torch.uint8, device=dev)
                pending[b] = shard.gather_ordered_start(dst, total, rank, world, out=gather_out[h])
            if nbuf > 1:
                cur[0] ^= 1
            total_all += total
        return total_all

    for _ in range(args.warmup):
        step()
    finish_gather()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lz_acc[0] = lz_acc[1] = 0.0
    t0 = time.perf_counter()
    out_total = 0
    for _ in range(args.steps):
        out_total = step()
    finish_gather()                                           # the last piece's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    lz_ms, stage_ms = lz_acc
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        o = torch.tensor([out_total], dtype=torch.int64, device=dev)
        dist.all_reduce(o)
        out_all = int(o.item())
    else:
        out_all = out_total
    # ---- outside the timed region: decode every entry of this rank's last archive on the device and compare with the inputs
    verified = None
    tm_last = ctx.timing()                                   # stage split of the last timed launch (the check below runs more kernels)
    if args.framing == "archive" and not args.no_verify and args.encrypt in ("none", "aes-ctr"):
        ok = True
        fs = max(1, (file_len.bit_length() + 7) // 8) if file_len else 0          # fSIZ payload: minimal big-endian
        extra = (12 + len(cipher.phsf.encode()) + 28) if cipher is not None else 0    # PHSF chunk + FDAT(iv) chunk
        back = torch.empty(n_piece * stride + 64, dtype=torch.uint8, device=dev)
        for h in range(pieces):
            total, eoff = ctx.create_archive_device(p_names[h], src.data_ptr(), p_off[h], p_len[h], dsts[0].data_ptr(), dst_cap, algo=algo,
                                                    _cache=arg_cache[h], part=part_flags(h), cipher=p_cipher[h])
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
    in_rank = n_files * file_len
    in_all = in_rank * world
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = in_all / (dt / args.steps) / 2**20
        lz_avg_s = lz_ms / args.steps / 1e3
        alg_bytes = in_rank + out_total                      # each input byte read once + each output byte written once
        achieved = alg_bytes / lz_avg_s / 1e9 if lz_avg_s > 0 else 0.0
        tm = tm_last
        line = {
            "metric": f"archive-create MiB/s (input bytes/sec), {args.algo}, 10k x 1MiB corpus",
            "value": round(value, 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"pna create, {n_files} x {file_len} B synthetic {'enwik-style' if args.kind == 0 else 'random'} text per GPU, Compression::{'ZStandard' if args.algo == 'zstd' else 'Deflate'} "
                                   f"(GPU encoder: 24512-entry LDS hash table, min_match 6, greedy+lazy1, 4096-position tiles), inputs resident in HBM, "
                                   + ("output = complete .pna archive bytes in HBM (chunk framing + CRC-32 on device)" if args.framing == "archive"
                                      else "--solid: inner STORE records serialised + one compressed stream + SDAT framing, all in HBM" if args.framing == "solid"
                                      else "output = packed compressed entry streams in HBM"),
                       "entries_per_gpu": n_files, "entry_bytes": file_len, "parallelism": f"entry-sharded x{world}",
                       "gather_pieces": pieces},
            "ratio": round(in_all / max(out_all, 1), 4),
            "verified": verified,                        # rank 0's archive decoded on the device == its inputs (None: not checked)
            "roofline": {"bound": "hbm", "kernel": "k_lz", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": recorded_traffic(n_files, file_len, args.algo, args.kind, args.framing),
                         "algorithmic_bytes": alg_bytes,
                         "kernel_ms": round(lz_ms / args.steps, 3), "all_kernels_ms": round(stage_ms / args.steps, 3)},
            # the last launch of the last step: with gather_pieces = P that is one piece, 1 / P of a step
            "stages_ms_last_step": {"lz": round(tm.ms_lz, 3), "stats": round(tm.ms_stats, 3), "lit": round(tm.ms_lit, 3),
                                    "seq": round(tm.ms_seq, 3), "pack": round(tm.ms_pack, 3), "frame": round(tm.ms_frame, 3)},
        }
        if cipher is not None:
            line["config"]["workload"] += f", cipher stage {args.encrypt} (AES-256) on the compressed payloads in HBM"
            line["stages_ms_last_step"]["cipher"] = round(tm.ms_cipher, 3)
            line["cipher"] = {"mode": args.encrypt, "ms": round(tm.ms_cipher, 3),
                              "GB_per_s": round(out_total / pieces / max(tm.ms_cipher, 1e-9) / 1e6, 1)}
        if world == 1 and not args.no_cpu_baseline and args.algo == "zstd":
            try:
                sample = args.cpu_sample_files or 64 * usable_cores()
                line["cpu_baseline"] = cpu_baseline(sample, file_len)
            except Exception as e:  # never lose the GPU number because the CPU leg failed
                line["cpu_baseline"] = {"value": None, "unit": "MiB/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
