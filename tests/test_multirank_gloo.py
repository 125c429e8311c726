"""world_size-2 gloo test of the N>1 path: contiguous sharding + ordered gather + archive assembly on rank 0."""
import io
import os
import socket
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import importlib
    import zlib
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pna = importlib.import_module("portable-network-archive_amd")
    shard = importlib.import_module("portable-network-archive_amd.shard")
    from oracle import codec, pna_format as pf
    n = 23
    sizes = [1000 + 977 * ((i * 7) % 11) for i in range(n)]
    names = [f"corpus/f{i:05d}.txt" for i in range(n)]
    bounds = shard.partition_entries(sizes, world)
    lo, hi = bounds[rank]
    # every rank "compresses" only its shard (stdlib zlib stands in for the GPU codec on this CPU-only host)
    mine = [zlib.compress(codec.corpus_file(1, i, sizes[i]), 6) for i in range(lo, hi)]
    lens = torch.tensor([len(m) for m in mine], dtype=torch.int64)
    blob = torch.frombuffer(bytearray(b"".join(mine) or b"\0"), dtype=torch.uint8)
    got, shard_bytes = shard.gather_ordered(blob, sum(len(m) for m in mine), rank, world)
    # the asynchronous form used by bench.py (gather of shard k overlapped with the compression of shard k + 1)
    h = shard.gather_ordered_start(blob, sum(len(m) for m in mine), rank, world)
    got2, buf2, shard_bytes2 = shard.gather_ordered_wait(h)
    assert shard_bytes2 == shard_bytes and (rank != 0 or bytes(got2.numpy().tobytes()) == bytes(got.numpy().tobytes()))
    assert (got2 is None and buf2 is None) if rank else (got2.numel() == sum(shard_bytes2) <= buf2.numel())
    # a destination that is too small is replaced, and the handle hands the buffer in use back (bench.py keeps it for the read-back and the next step)
    h3 = shard.gather_ordered_start(blob, sum(len(m) for m in mine), rank, world, out=torch.empty(10, dtype=torch.uint8) if rank == 0 else None)
    got3, buf3, sb3 = shard.gather_ordered_wait(h3)
    assert sb3 == shard_bytes and (rank != 0 or (buf3.numel() >= sum(sb3) and got3.numel() == sum(sb3) and bytes(got3.numpy().tobytes()) == bytes(got.numpy().tobytes())))
    # ... unless growing is not allowed: then the verdict is COLLECTIVE -- the capacity travels with the sizes, every rank raises before anything
    # is sent (pna_gpu_gather_ordered's PNA_E_DSTSIZE on all ranks; the review's hang), and the group is still in step for the next gather
    try:
        shard.gather_ordered_start(blob, sum(len(m) for m in mine), rank, world, out=torch.empty(10, dtype=torch.uint8) if rank == 0 else None, grow=False)
        raise AssertionError("an overflow must raise on every rank")
    except shard.GatherOverflow as e:
        assert e.sizes == shard_bytes and e.cap == 10 and e.need == sum(shard_bytes)
    got4, sb4 = shard.gather_ordered(blob, sum(len(m) for m in mine), rank, world, out=torch.empty(sum(shard_bytes), dtype=torch.uint8) if rank == 0 else None, grow=False)
    assert sb4 == shard_bytes and (rank != 0 or bytes(got4.numpy().tobytes()) == bytes(got.numpy().tobytes()))
    # two gathers in flight at once, each into its own destination -- what bench.py does with --gather-pieces 2 (piece 0 travels while
    # piece 1 is compressed into the other buffer)
    half = len(mine) // 2
    pa, pb = b"".join(mine[:half]), b"".join(mine[half:])
    ta = torch.frombuffer(bytearray(pa or b"\0"), dtype=torch.uint8); tb = torch.frombuffer(bytearray(pb or b"\0"), dtype=torch.uint8)
    ha = shard.gather_ordered_start(ta, len(pa), rank, world)
    hb = shard.gather_ordered_start(tb, len(pb), rank, world)
    ga, _, sa = shard.gather_ordered_wait(ha)
    gb, _, sb = shard.gather_ordered_wait(hb)
    la = torch.tensor([len(pa), len(pb)], dtype=torch.int64); alls = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(alls, la)
    assert sa == [int(x[0]) for x in alls] and sb == [int(x[1]) for x in alls]
    if rank == 0:
        assert bytes(ga.numpy().tobytes())[:len(pa)] == pa and bytes(gb.numpy().tobytes())[:len(pb)] == pb
        assert len(ga) == sum(sa) and len(gb) == sum(sb)
    all_lens = [torch.zeros(bounds[r][1] - bounds[r][0], dtype=torch.int64) for r in range(world)]
    dist.all_gather(all_lens, lens) if len({b[1] - b[0] for b in bounds}) == 1 else None
    if len({b[1] - b[0] for b in bounds}) != 1:          # ragged shards: gather lengths by object
        obj = [None] * world
        dist.all_gather_object(obj, lens.tolist())
        all_lens = [torch.tensor(o, dtype=torch.int64) for o in obj]
    if rank == 0:
        stream = bytes(got.numpy().tobytes())
        buf = io.BytesIO(); a = pna.Archive(buf); pos = 0; i = 0
        for r in range(world):
            for L in all_lens[r].tolist():
                a.add_file(names[i], pna.ALGO_DEFLATE, sizes[i], stream[pos:pos + L]); pos += L; i += 1
        a.finalize()
        _, items = pf.read_archive(buf.getvalue())
        ok = [it.name for it in items] == names and all(
            codec.decode_payload(it.compression, it.data, 1 << 20) == codec.corpus_file(1, k, sizes[k]) for k, it in enumerate(items))
        q.put(("ok" if ok and pos == len(stream) and i == n else "bad", bounds, shard_bytes))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_properties():
    import importlib
    shard = importlib.import_module("portable-network-archive_amd.shard")
    for sizes, world in (([1 << 20] * 10000, 8), ([5, 1, 1, 1, 1, 1], 3), ([7], 4), ([], 2), ([3, 3, 3, 3], 4), ([1] * 5, 8)):
        b = shard.partition_entries(sizes, world)
        assert len(b) == world and b[0][0] == 0 and b[-1][1] == len(sizes)
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1)) and all(s <= e for s, e in b)
    b = shard.partition_entries([1 << 20] * 10000, 8)
    assert all(e - s == 1250 for s, e in b)


def test_piece_ranges_tile_the_archive_in_gather_order():
    """bench.py --gather-pieces: walking the pieces in gather order (piece-major, rank-minor) must walk the corpus front to back."""
    import importlib
    shard = importlib.import_module("portable-network-archive_amd.shard")
    for world, pieces, per in ((1, 1, 10000), (2, 2, 5000), (8, 2, 5000), (4, 4, 7), (3, 1, 5)):
        pos = 0
        for h in range(pieces):
            for r in range(world):
                lo, hi = shard.piece_ranges(r, world, pieces, per)[h]
                assert lo == pos and hi == lo + per
                pos = hi
        assert pos == world * pieces * per


@pytest.mark.timeout(180)
def test_two_rank_ordered_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=150)
    for p in procs:
        p.join(60)
    assert res[0] == "ok", res
    assert all(p.exitcode == 0 for p in procs)


def _plain_bench(extra, timeout=150):
    """`python bench.py --gpus N ...` exactly as a driver would type it: no launcher around it, no WORLD_SIZE / RANK in the environment."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(240)
def test_plain_command_line_starts_its_own_ranks():
    """The review's first item: `python bench.py --gpus N` with WORLD_SIZE unset used to sys.exit() before touching a GPU.  Now that process becomes the
    launcher (torch.distributed.run as a child, free port on 127.0.0.1) BEFORE anything initialises the GPU; rank 0's JSON line is its stdout and a
    failing rank makes the exit code non-zero.  --launch-check stops after the rendezvous (gloo), so this runs on a CPU-only host."""
    import json
    r = _plain_bench(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"launch_check": True, "world": 2, "ranks_seen": 2, "distinct_processes": 2, "launcher": "torch.distributed.run"}
    r = _plain_bench(["--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1"])
    assert r.returncode != 0
    # under a launcher of the driver's own the same file must not start a second generation of ranks
    import subprocess
    port = _free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["ranks_seen"] == 2


def test_gather_verdict_is_the_same_on_every_rank(pna):
    """pna_gather_verdict: host arithmetic on the all-gathered (size, capacity) pairs -- an overflow is PNA_E_DSTSIZE whichever rank evaluates it."""
    assert pna.gather_verdict([5, 0, 7], [12, 0, 0]) == (0, [0, 5, 5, 12])
    rc, offs = pna.gather_verdict([5, 0, 7], [11, 99, 99])                 # only the root's capacity counts
    assert rc == pna.E_DSTSIZE and offs[-1] == 12
    assert pna.gather_verdict([5, 0, 7], [0, 0, 12], root=2)[0] == 0
    assert pna.gather_verdict([1 << 63, 1 << 63], [1 << 62, 0])[0] == pna.E_INVAL
