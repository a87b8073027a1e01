"""bench.py's own multi-rank plumbing on a gloo world of 8 (CPU): the shards of the strong-scaling side workloads cover
the 2^20 points exactly, the step time is the MAX over the ranks (one all-reduce, bench.max_over_ranks), the job rate
is the units of ALL ranks over that time, and rank 0 alone builds a line that fits -- so that a `--gpus 8` driver run
cannot fail on arithmetic the single-GPU runs never exercise.  No scaling curve exists."""
import json
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from kyber_amd import dist as kd

N = 1 << 20


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = kd.shard_range(N, rank, world)
        # a rank-dependent "elapsed": rank 5 is the slowest; two figures at once, as other_workloads() sends them
        mine = [0.010 + 0.001 * ((rank * 5) % world), 3.0 + rank]
        got = bench.max_over_ranks(dist, mine, device="cpu")
        dist.barrier()
        q.put((rank, lo, hi, mine, got))
    finally:
        dist.destroy_process_group()


def test_bench_sharding_and_max_time_world8_gloo():
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spans = [(lo, hi) for _, lo, hi, _, _ in res]
    assert spans[0][0] == 0 and spans[-1][1] == N and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sum(hi - lo for lo, hi in spans) == N
    want = [max(m[0] for _, _, _, m, _ in res), max(m[1] for _, _, _, m, _ in res)]
    for rank, _, _, _, got in res:
        assert got == want, (rank, got, want)  # every rank holds the slowest rank's time
    # the job figure of the weak-scaling headline: units of all ranks over the slowest rank's time
    steps, n = 20, N
    value = 2 * n * steps * world / (want[0] * steps)
    assert abs(value - 2 * n * world / want[0]) < 1e-6 * value
    full = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_final_bench.json")))
    full.update({"n_gpus": world, "value": value, "ms_per_step": want[0] * 1e3, "rccl_ranks_seen": world})
    full.pop("cpu_baseline", None)  # N > 1 runs skip the CPU leg
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT and line["n_gpus"] == 8 and line["rccl_ranks_seen"] == 8


def test_max_over_ranks_without_a_process_group():
    assert bench.max_over_ranks(None, [1.5, 2]) == [1.5, 2.0]
