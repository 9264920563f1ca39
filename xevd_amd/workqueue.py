"""Host work queue for multi-GPU decoding: independent streams / closed GOPs are the unit of work, one
process per GPU, no collective on the data path.

Every rank computes the same deterministic assignment (longest-processing-time first over the job costs), decodes
its own jobs on its own GPU with its own xgpu_ctx/DPB, and only the final accounting (frames, wall time) goes
through torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""


def assign_jobs(costs, world):
    """costs[i] = relative cost of job i (e.g. pictures x samples).  Returns a list of job-index lists, one per rank;
    identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    for r in range(world):
        out[r].sort()
    return out


def run_jobs(jobs, costs, decode_fn, dist=None, device=None):
    """Decode this rank's share of `jobs`; returns (frames decoded by all ranks, max-over-ranks seconds).
    decode_fn(job) -> (frames, seconds)."""
    import torch
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = assign_jobs(costs, world)[rank]
    frames, secs = 0, 0.0
    for i in mine:
        f, s = decode_fn(jobs[i])
        frames += f
        secs += s
    if dist is None:
        return frames, secs
    t = torch.tensor([float(frames), secs], dtype=torch.float64, device=device)
    tf, ts = t[:1].clone(), t[1:].clone()
    dist.all_reduce(tf, op=dist.ReduceOp.SUM)
    dist.all_reduce(ts, op=dist.ReduceOp.MAX)
    return int(tf.item()), float(ts.item())
