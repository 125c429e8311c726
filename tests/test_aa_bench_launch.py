"""The plain command line of the multi-GPU bench on the ONE-GPU box: `python bench.py --gpus 2 ...` with no launcher around it and no WORLD_SIZE in the
environment -- what the driver types.  The process must start its own ranks before anything touches the GPU, run the whole N > 1 control flow (shard
layout, part flags, size exchange, ordered gather, read-back of the gathered archive) and print rank 0's JSON line; PNA_BENCH_REHEARSAL=1 puts both
ranks on cuda:0 and the exchange on gloo (RCCL refuses two ranks on one device) -- timings of such a run mean nothing.

This module sorts first on purpose: its test starts child processes, and it does so before any test of this pytest process has initialised the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_plain_command_line_two_rank_rehearsal_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env["PNA_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--files", "64", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{\"metric\"")][-1])
    assert d["n_gpus"] == 2 and d["config"]["entries"] == 64 and d["config"]["entries_per_gpu"] == 32 and d["config"]["gather_pieces"] == 2
    assert d["verified"] is True and d["gathered_archive_verified"] is True
    assert d["scaling"] == "strong" and d["value"] > 0
