"""Multi-GPU plumbing: one process per MI355X, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm).

The path shards by meta-task: task m runs on rank m % world; the only data-path collective is one
SUM all-reduce of the flat fp32 meta-gradient per meta-step.  Everything else (theta0, Adam moments) is replicated and
updated by identical deterministic kernels, so replicas stay bit-identical without a parameter broadcast.
Works with backend "gloo" on CPU tensors too (used by the world_size-2 tests).
"""
import os

import torch
import torch.distributed as td


def is_on():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_on() else 0


def world_size():
    return td.get_world_size() if is_on() else 1


def _always():
    """MTL_DIST_ALWAYS=1: build the process group and issue the collectives even with ONE rank (lets a 1-GPU box execute the
    RCCL path end to end: tests/test_trainer_gpu.py::test_rccl_collective_path_executes)."""
    return os.environ.get('MTL_DIST_ALWAYS', '0') == '1'


def backend_name():
    return td.get_backend() if is_on() else 'none'


def init_from_env(backend=None):
    """Initialise from torchrun's RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*; no-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if (world <= 1 and not (_always() and 'RANK' in os.environ)) or is_on():
        lr_ = int(os.environ.get('LOCAL_RANK', '0'))
        return lr_ % torch.cuda.device_count() if torch.cuda.is_available() else lr_
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if backend is None:
        # MTL_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the sharded path on a 1-GPU box)
        backend = os.environ.get('MTL_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    td.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=world)
    _scalar_group()
    return local


_HOST_GROUP = None


def _scalar_group():
    """Host-side (gloo) group for the per-iteration log scalars.  On the RCCL communicator they would queue up behind the next
    meta-step's gradient all-reduce -- i.e. behind all of its kernels -- and stall the host, which enqueues one iteration ahead
    (TransientTrainer.enqueue_iteration).  None = the default group is gloo already."""
    global _HOST_GROUP
    if td.get_backend() == 'gloo':
        return None
    if _HOST_GROUP is None:
        if os.environ.get('MASTER_ADDR', '127.0.0.1') in ('127.0.0.1', 'localhost'):
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')      # single node: the container hostname may not resolve
        _HOST_GROUP = td.new_group(backend='gloo')
    return _HOST_GROUP


def shard_tasks(n_tasks, rank_, world):
    """Task ids owned by `rank_`: m with m % world == rank_ (8 tasks on 8 GPUs = one each)."""
    return [m for m in range(n_tasks) if m % world == rank_]


def allreduce_sum_(flat):
    """In-place SUM all-reduce of the flat meta-gradient (56 MB at the north-star size): one collective per meta-step."""
    if is_on() and (world_size() > 1 or _always()):
        td.all_reduce(flat, op=td.ReduceOp.SUM)
    return flat


def collective_on():
    """Is the meta-gradient all-reduce issued at all (several ranks, or MTL_DIST_ALWAYS)?"""
    return is_on() and (world_size() > 1 or _always())


def chunked_on():
    """The all-reduce as one collective per parameter group (decoder, encoder, conv), each started as soon as that group's gradients
    are final, i.e. under the rest of the validation backward (MTL_CHUNKED_ALLREDUCE=0: ONE collective after the backward)."""
    return collective_on() and os.environ.get('MTL_CHUNKED_ALLREDUCE', '1') != '0'


SLICE_ORDER = ('decoder', 'encoder', 'conv')     # the order the validation backward finishes the parameter groups in; EVERY rank posts
#                                                  its slice collectives in this order, whatever schedule its local tasks take


class ChunkedAllReduce:
    """SUM all-reduce of a flat buffer as a fixed sequence of slices, each issued asynchronously (`issue`) as soon as the caller
    knows it is final; `wait()` joins all of them.  Slices are disjoint and always issued in the same order on every rank, so the
    result is reproducible run to run; with two ranks it is bitwise the single all-reduce (a + b), with more ranks it equals it up
    to the ring's summation order inside each slice (parity within tolerance, not bitwise)."""

    def __init__(self):
        self.works = []

    def issue(self, flat_slice):
        if collective_on():
            self.works.append(td.all_reduce(flat_slice, op=td.ReduceOp.SUM, async_op=True))

    def wait(self):
        works, self.works = self.works, []
        for w in works:
            w.wait()               # (device tensors: makes the CURRENT stream wait for the collective's stream; host tensors: blocks)


def allreduce_scalars(values, device):
    if not (is_on() and world_size() > 1):
        return values
    t = torch.tensor(values, dtype=torch.float64)               # host tensor, host collective: never waits for queued GPU work
    td.all_reduce(t, op=td.ReduceOp.SUM, group=_scalar_group())
    return t.tolist()


def barrier():
    if is_on() and world_size() > 1:
        td.barrier()
