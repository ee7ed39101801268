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


def test_bench_spawns_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts 2 worker processes itself (env contract of torch.distributed.run,
    rendezvous on 127.0.0.1), times K steps between barriers, takes the max over ranks and prints ONE JSON line from rank 0.
    Here over gloo with a stand-in model (--stub-model); on the GPU box the same code path runs the real model over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                         "--backend", "gloo", "--stub-model", "--batch", "3"], env=env, capture_output=True, text=True, timeout=180)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["world_size_seen"] == 2 and r["steps"] == 5
    # the slowest rank (rank 1 sleeps twice as long) sets the time, value = all ranks' units / that time
    assert r["ms_per_step"] >= 3.9 and r["ms_per_step"] >= r["rank0_ms_per_step"] * 0.999
    assert abs(r["value"] - 3 * 5 * 2 / (r["ms_per_step"] * 5e-3)) < 1e-2 * r["value"]
    # under a launcher (RANK set) the same file does not spawn again; a mismatching --gpus is refused
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cp2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-model", "--backend", "gloo"],
                         env=env2, capture_output=True, text=True, timeout=120)
    assert cp2.returncode != 0 and "WORLD_SIZE=1" in cp2.stderr


def test_bench_spawns_ranks_for_every_line(tmp_path):
    """every line bench.py can print with --gpus N -- the training step (cfg4: one all-reduce of the flat gradient bucket per step), the
    scene-sharded configurations (cfg5, --scene-mode) and cfg3 -- spawned over 2 gloo ranks with the stand-in model, reduced and printed once.
    None of them has run on more than one MI355X (1-GPU leases): the lines say so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    for extra, kind in ((["--config", "cfg4"], "cfg4"), (["--config", "cfg5"], "cfg5"), (["--config", "cfg3"], "cfg3"), (["--scene-mode"], "scene")):
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
                             "--stub-model", "--batch", "2"] + extra, env=env, capture_output=True, text=True, timeout=180)
        assert cp.returncode == 0, (kind, cp.stderr[-2000:])
        lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (kind, cp.stdout)
        r = json.loads(lines[0])
        assert r["stub_of"] == kind and r["n_gpus"] == 2 and r["world_size_seen"] == 2 and "unmeasured on hardware" in r["multi_gpu"]
        assert r["ms_per_step"] >= 3.9                                   # the slower rank sets the time
        if kind == "cfg4":
            assert r["allreduce_bytes"] == 925435 * 4 and r["allreduce_ms_per_step"] > 0
            assert abs(r["bucket_value"] - 1.5) < 1e-6                  # SUM over ranks (1 + 2) / world: both ranks hold the average
        if kind in ("cfg5", "scene"):
            sh = r["scene_sharding"]
            assert sh["scenes_total"] == 4 and sh["scenes_of_rank0"] == ["scene00", "scene02"]
            assert abs(r["value"] - 4 * 2 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-2 * r["value"]      # all ranks' scenes x batch x steps / whole time
