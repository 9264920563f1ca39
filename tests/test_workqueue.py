"""CPU suite: the N>1 path - the C work queue of include/xevd_wq.h driven with real parsed jobs (closed GOPs of golden streams) on
a fake backend, and the cross-process ticket queue of bench.py with world_size 2 over gloo.  No GPU."""
import glob
import os
import socket
import sys
import threading
import time

import numpy as np
import pytest

import golden_io
from xevd_amd import stream, workqueue


def _streams():
    out = []
    for path in sorted(glob.glob(os.path.join(golden_io.GOLDEN, "stream_*.npz"))):
        d = np.load(path)
        out.append((os.path.basename(path), d["bytes"].tobytes(), int(d["n"])))
    return out


def _parsed(data):
    """the parser's pictures; streams whose parser refines vectors itself (tool_dmvr with tool_hmvp / tool_mmvd) need the decoded reference samples
    to be parsed at all - those go through parser + oracle"""
    try:
        return list(stream.iter_stream(data))
    except RuntimeError as e:
        if "xhost_parser_set_ref_luma" not in str(e):
            raise
    import stream_util as su
    pics = []
    su.decode_oracle(data, keep_params=pics)
    return pics


def test_gop_split_covers_every_picture_and_units_parse_alone():
    """every golden stream: the closed GOPs partition its slice NAL units, and each unit (its parameter sets prepended) parses on its own into
    exactly the pictures the sequential parse yields for that stretch - same POCs, same CU counts"""
    some_multi = False
    for name, data, n in _streams():
        jobs = workqueue.split_gops(data)
        assert sum(j.n_pictures for j in jobs) == n, name
        assert [j.first_picture for j in jobs] == list(np.cumsum([0] + [j.n_pictures for j in jobs[:-1]])), name
        some_multi |= len(jobs) > 1
        whole = [(p["poc"], len(p["batch"]["x"]), int(p["batch"]["n_coef"])) for p in _parsed(data)]
        k = 0
        for j in jobs:
            unit = [(p["poc"], len(p["batch"]["x"]), int(p["batch"]["n_coef"])) for p in _parsed(workqueue.unit_bytes(data, j))]
            assert unit == whole[k:k + j.n_pictures], (name, j.unit)
            k += j.n_pictures
    assert some_multi, "no golden stream with more than one IDR period"


def test_c_queue_runs_every_job_once_and_balances_dynamically():
    """jobs = the GOPs of all golden streams; three fake devices, one of them 20x slower: every job runs exactly once, the slow device ends up
    with fewer jobs than the fast ones (a static partition would give it a third), a device that fails to come up gets none"""
    jobs = []
    for s, (_, data, _) in enumerate(_streams()):
        jobs += workqueue.split_gops(data, stream=s)
    jobs = jobs * 3                                              # enough work for the balance to show
    q = workqueue.WorkQueue()
    for j in jobs:
        assert q.push(j) == 0
    q.close()
    seen, lock = [], threading.Lock()

    def decode(device, job):
        time.sleep(0.020 if device == 1 else 0.001)
        with lock:
            seen.append((device, job.stream, job.unit))
        return 0
    rc, done = q.run([0, 1, 2, 7], decode, init_ok=lambda d: d != 7)
    q.destroy()
    assert rc == 0 and sum(done) == len(jobs) and done[3] == 0
    assert sorted((s, u) for _, s, u in seen) == sorted((j.stream, j.unit) for j in jobs)
    assert done[1] < done[0] and done[1] < done[2] and done[1] < len(jobs) // 4


def test_c_queue_reports_a_failing_job_and_still_drains():
    jobs = workqueue.split_gops(_streams()[0][1])
    q = workqueue.WorkQueue()
    for j in jobs * 4:
        q.push(j)
    q.close()
    calls = []

    def decode(device, job):
        calls.append(job.unit)
        return -202 if len(calls) == 2 else 0
    rc, done = q.run([0], decode)
    q.destroy()
    assert rc == -202 and len(calls) == 4 * len(jobs) and done[0] == 4 * len(jobs) - 1
    assert q.push(jobs[0]) != 0 if q.q else True


def test_c_queue_reports_a_job_failing_with_the_unexpected_code():
    """-105 (XGPU_ERR_UNEXPECTED: every HIP error and frame_end failure) is a job's failure like any other: reported even when other jobs of the worker succeeded
    and another worker never came up - "worker down" is a flag of its own, not that code"""
    jobs = workqueue.split_gops(_streams()[0][1])
    q = workqueue.WorkQueue()
    for j in jobs * 3:
        q.push(j)
    q.close()
    calls = []

    def decode(device, job):
        calls.append(job.unit)
        return -105 if len(calls) == 2 else 0
    rc, done = q.run([0, 7], decode, init_ok=lambda d: d != 7)
    q.destroy()
    assert rc == -105 and done == [3 * len(jobs) - 1, 0]
    q = workqueue.WorkQueue()
    q.push(jobs[0])
    q.close()
    rc, done = q.run([7, 9], decode, init_ok=lambda d: False)          # no worker at all
    q.destroy()
    assert rc == -105 and done == [0, 0]


def _worker(rank, world, port, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = [{"frames": 5 + i} for i in range(12)]
    seen = []

    def decode(job):
        seen.append(job["frames"])
        time.sleep(0.02 * (1 + 4 * rank))                      # rank 1 is five times slower
        return job["frames"], 0.01 * job["frames"]

    frames, secs, mine = workqueue.run_jobs(jobs, decode, dist=dist)
    out_q.put((rank, frames, secs, mine, sorted(seen)))
    dist.destroy_process_group()


def test_two_ranks_draw_from_one_ticket_queue_over_gloo():
    """bench.py's N>1 path: the ranks share nothing but the job counter in the rendezvous store; every job is decoded exactly once, the
    faster rank draws more of them, both ranks agree on the totals"""
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
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(5 + i for i in range(12))
    assert all(r[1] == total for r in res)                         # every rank sees the global frame count
    assert res[0][2] == res[1][2] and res[0][2] > 0                # and the same max-over-ranks time
    assert sorted(res[0][4] + res[1][4]) == [5 + i for i in range(12)]   # each job decoded exactly once
    assert res[0][3] + res[1][3] == total and len(res[0][4]) > len(res[1][4])      # dynamic: the fast rank took more jobs


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus 2` launched plainly (no torchrun, WORLD_SIZE unset) must run TWO ranks that share the job queue and print one
    line with n_gpus == 2.  The decoder is tests/stub_decoder.py: this checks launcher + queue + accounting."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["XEVD_BENCH_DECODER"] = "stub_decoder"
    env["PYTHONPATH"] = os.path.join(root, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "1", "--batches", "1",
                        "--workload", "cfg2_base_1080p_8b_ippp", "--no-cpu-baseline", "--no-end-to-end"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines      # stdout is the ONE JSON line - no library chatter (gloo reports its connections there unless kept away)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["decoder"] == "stub_decoder"
    assert sorted(p["rank"] for p in out["per_rank"]) == [0, 1]
    assert sum(p["pictures"] for p in out["per_rank"]) == 2 * 12          # weak scaling: world x steps pictures, drawn from one queue
