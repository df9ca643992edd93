"""Host environment facts the plumbing needs: how many CPUs this process may actually use.

torch sizes its intra-op (OpenMP) pool from the machine's core count.  In a container with a CFS quota that is far too many: on the
MI355X pool a box shows 256 logical CPUs under `cpu.max = 1600000 100000` (16 CPUs' worth of time per 100 ms period), torch starts
128 OpenMP threads, and ONE parallel region -- a 150 KB host-to-host `copy_` in the enqueue path was enough -- wakes all of them;
their spin-waits burn the cgroup's quota and the kernel then throttles EVERY thread of the container, the enqueueing host thread
included, until the next period: 50-100 ms stalls inside arbitrary HIP calls (measured with tools/probe/stallwatch.c +
/sys/fs/cgroup/cpu.stat: nr_throttled rises only while the pool is at 128 threads; OMP_NUM_THREADS=1 -> no stall, profiles/r5/).
The driver's round-4 bench run (host enqueue 80 ms per step, everything else idle) was this."""
import os


def cgroup_cpu_quota():
    """CPUs' worth of time per period granted by the container's CFS quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`), or None
    when there is no limit / it cannot be read."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        quota = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        period = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return quota / period if quota > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus():
    """min(CPUs in the affinity mask, CFS quota), at least 1: what a thread pool in this process can keep busy without being throttled"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(int(q), 1))
    return max(n, 1)


def bound_torch_threads(limit=None):
    """Caps torch's intra-op pool at the CPUs this process may use (never raises it): the enqueue path itself runs no torch CPU
    kernel, but the caller's featurisation thread and the loss / CER read-backs may, and a 128-thread team in a 16-CPU container
    stalls the whole process (see module docstring).  MTL_HOST_THREADS=<n> overrides the limit, 0 leaves torch alone.  -> threads in use"""
    import torch
    env = os.environ.get('MTL_HOST_THREADS')
    if env is not None:
        if int(env) <= 0:
            return torch.get_num_threads()
        limit = int(env)
    if limit is None:
        # three quarters of this process's share (torchrun: LOCAL_WORLD_SIZE ranks share the container): the enqueueing thread, the
        # runtime's own threads and whatever else runs in the container (a monitoring poll) need the rest
        share = effective_cpus() // max(int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1), 1)
        limit = max(1, (3 * share) // 4)
    if torch.get_num_threads() > limit:
        torch.set_num_threads(limit)
    return torch.get_num_threads()
