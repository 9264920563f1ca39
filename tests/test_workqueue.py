"""CPU suite: the N>1 path - job assignment and accounting with world_size 2 over gloo (no GPU)."""
import os
import socket
import sys

import pytest

from xevd_amd import workqueue


def test_assignment_is_a_partition_and_balanced():
    costs = [8, 1, 1, 1, 4, 4, 2, 2, 16]
    for world in (1, 2, 4, 8):
        parts = workqueue.assign_jobs(costs, world)
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) <= max(max(costs), -(-sum(costs) // world) + max(costs))
    # identical costs -> sizes differ by at most one
    parts = workqueue.assign_jobs([1] * 10, 4)
    assert sorted(len(p) for p in parts) == [2, 2, 3, 3]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = [{"frames": 5 + i} for i in range(7)]
    costs = [j["frames"] for j in jobs]
    seen = []

    def decode(job):
        seen.append(job["frames"])
        return job["frames"], 0.01 * job["frames"] * (rank + 1)

    frames, secs = workqueue.run_jobs(jobs, costs, decode, dist=dist)
    q.put((rank, frames, secs, sorted(seen)))
    dist.destroy_process_group()


def test_two_ranks_over_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(5 + i for i in range(7))
    assert all(r[1] == total for r in res)                         # every rank sees the global frame count
    assert res[0][2] == res[1][2] and res[0][2] > 0                # and the same max-over-ranks time
    assert sorted(res[0][3] + res[1][3]) == [5 + i for i in range(7)]   # each job decoded exactly once
