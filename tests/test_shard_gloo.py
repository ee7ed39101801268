"""CPU, world_size 2 over gloo: the N>1 inference path = static sharding of reference views with no
data-path collective; only the barrier / max-time / item-count reductions touch the process group."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from diffmvs_amd.shard import shard_items, shard_scenes


def test_partition_is_exact():
    for n in (0, 1, 7, 8, 49):
        for world in (1, 2, 3, 8):
            owned = [shard_items(n, r, world) for r in range(world)]
            flat = sorted(i for o in owned for i in o)
            assert flat == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    assert shard_scenes(["a", "b", "c"], 1, 2) == ["b"]


WORKER = textwrap.dedent("""
    import os, sys, time, torch
    sys.path.insert(0, %r)
    from diffmvs_amd import shard
    rank, world, _ = shard.env_rank_world()
    td = shard.init_distributed("gloo")
    mine = shard.shard_items(11, rank, world)
    # stand-in for the per-rank work: deterministic, rank-dependent duration
    elapsed = 0.010 * (rank + 1)
    td.barrier()
    whole = shard.barrier_and_max(elapsed)
    n = shard.total_items(len(mine))
    assert abs(whole - 0.010 * world) < 1e-9, whole
    assert n == 11, n
    # no rank ever needs another rank's inputs or outputs: ownership is disjoint
    gathered = [None] * world
    td.all_gather_object(gathered, mine)
    assert sorted(i for g in gathered for i in g) == list(range(11))
    if rank == 0:
        print("OK", whole, n)
    td.destroy_process_group()
""")


def test_world_size_two_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]
