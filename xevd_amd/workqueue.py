"""Host work queue for multi-GPU decoding: independent streams / closed GOPs are the unit of work, no collective on the
data path.

Two forms of the same dynamic queue:
  * inside one process - the C queue of include/xevd_wq.h (libxevd_host.so): one worker thread + one xgpu_ctx per device, what
    `examples/evc_decode --gpus N` runs.  WorkQueue below is its ctypes face (used by the tests with a fake backend);
  * across the processes of `torch.distributed.run` (one rank per GPU, the benchmark contract) - TicketQueue: a job counter in the rendezvous
    store; every rank draws the next job index when it is done with the last.  The store is the only thing the ranks share; pictures never
    leave their GPU."""
import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libxevd_host.so")


class Job(C.Structure):
    _fields_ = [("stream", C.c_int), ("unit", C.c_int), ("offset", C.c_uint64), ("size", C.c_uint64), ("first_picture", C.c_int),
                ("n_pictures", C.c_int), ("user", C.c_void_p)]


INIT_FN = C.CFUNCTYPE(C.c_void_p, C.c_int, C.c_void_p)
JOB_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Job))
FINI_FN = C.CFUNCTYPE(None, C.c_void_p)
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB_PATH)
        l.xwq_create.restype = C.c_void_p
        l.xwq_destroy.argtypes = [C.c_void_p]
        l.xwq_push.argtypes = [C.c_void_p, C.POINTER(Job)]
        l.xwq_close.argtypes = [C.c_void_p]
        l.xwq_pop.argtypes = [C.c_void_p, C.POINTER(Job)]
        l.xwq_run.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, INIT_FN, JOB_FN, FINI_FN, C.c_void_p, C.POINTER(C.c_int)]
        l.xwq_split_gops.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Job), C.c_int]
        l.xwq_unit_bytes.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Job), C.c_char_p, C.c_size_t]
        l.xwq_unit_bytes.restype = C.c_size_t
        _lib = l
    return _lib


def split_gops(data, stream=0, max_jobs=4096):
    """closed GOPs (IDR to IDR) of a length-prefixed EVC stream -> list of Job"""
    while True:
        jobs = (Job * max_jobs)()
        n = lib().xwq_split_gops(data, len(data), stream, jobs, max_jobs)
        if n != -203:
            break
        max_jobs *= 4              # more closed GOPs than the array held
    if n < 0:
        raise ValueError(f"damaged NAL length prefix ({n})")
    return [Job.from_buffer_copy(bytes(jobs[i])) for i in range(n)]


def unit_bytes(data, job):
    """the bytes a worker decodes for one unit: the stream's parameter sets before it + the unit"""
    out = C.create_string_buffer(int(job.offset + job.size) + 16)
    n = lib().xwq_unit_bytes(data, len(data), C.byref(job), out, len(out))
    if n == 0:
        raise ValueError("unit outside the stream")
    return out.raw[:n]


class WorkQueue:
    """the C queue with Python callbacks: run(devices, job_fn) calls job_fn(device, Job) -> int on one worker thread per device"""
    def __init__(self):
        self.q = lib().xwq_create()

    def push(self, job):
        return lib().xwq_push(self.q, C.byref(job))

    def close(self):
        lib().xwq_close(self.q)

    def run(self, devices, job_fn, init_ok=lambda device: True):
        states = {}

        def init(device, user):
            if not init_ok(device):
                return None
            states[device + 1] = device
            return device + 1                                   # the worker's state: any non-NULL value

        def job(state, jp):
            return int(job_fn(states[state], jp.contents))
        dev = (C.c_int * len(devices))(*devices)
        done = (C.c_int * len(devices))()
        rc = lib().xwq_run(self.q, dev, len(devices), INIT_FN(init), JOB_FN(job), FINI_FN(lambda s: None), None, done)
        return rc, list(done)

    def destroy(self):
        lib().xwq_destroy(self.q)
        self.q = None


class TicketQueue:
    """Dynamic job queue across the ranks of torch.distributed: next() draws the next job index from a counter in the rendezvous store
    (atomic add), or None when all n_jobs are taken.  No collective; a rank that finishes early simply draws more."""
    def __init__(self, store, n_jobs, key="xevd_amd/jobs"):
        self.store, self.n_jobs, self.key = store, n_jobs, key

    def next(self):
        i = self.store.add(self.key, 1) - 1
        return i if i < self.n_jobs else None


def run_jobs(jobs, decode_fn, dist=None, device=None):
    """Every rank draws jobs until the queue is dry.  decode_fn(job) -> (frames, seconds).  Returns (frames decoded by all ranks,
    max-over-ranks seconds, this rank's frames); the accounting is the only collective (gloo / RCCL all_reduce of three numbers)."""
    import torch
    if dist is None:
        fs = [decode_fn(j) for j in jobs]
        return sum(f for f, _ in fs), sum(s for _, s in fs), sum(f for f, _ in fs)
    import torch.distributed.distributed_c10d as c10d
    q = TicketQueue(c10d._get_default_store(), len(jobs))
    frames, secs = 0, 0.0
    while True:
        i = q.next()
        if i is None:
            break
        f, s = decode_fn(jobs[i])
        frames += f
        secs += s
    t = torch.tensor([float(frames), secs], dtype=torch.float64, device=device)
    tf, ts = t[:1].clone(), t[1:].clone()
    dist.all_reduce(tf, op=dist.ReduceOp.SUM)
    dist.all_reduce(ts, op=dist.ReduceOp.MAX)
    return int(tf.item()), float(ts.item()), frames
