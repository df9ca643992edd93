"""hostenv: the CPU budget of the container (CFS quota) bounds torch's intra-op pool -- a 128-thread OpenMP team under a 16-CPU quota
got the whole bench process throttled (round 4's host-bound driver run)."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('hostenv_under_test', os.path.join(ROOT, 'meta-transfer-learning_amd', 'hostenv.py'))
hostenv = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hostenv)


def test_quota_parsing(monkeypatch, tmp_path):
    real_open = open

    def fake(v2=None, v1=None):
        def _open(path, *a, **k):
            if path == '/sys/fs/cgroup/cpu.max':
                if v2 is None:
                    raise OSError
                p = tmp_path / 'cpu.max'
                p.write_text(v2)
                return real_open(p)
            if path.startswith('/sys/fs/cgroup/cpu/'):
                if v1 is None:
                    raise OSError
                p = tmp_path / os.path.basename(path)
                p.write_text(v1[os.path.basename(path)])
                return real_open(p)
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr('builtins.open', fake(v2='1600000 100000\n'))
    assert hostenv.cgroup_cpu_quota() == 16.0
    assert hostenv.effective_cpus() == min(16, len(os.sched_getaffinity(0)))
    monkeypatch.setattr('builtins.open', fake(v2='max 100000\n'))
    assert hostenv.cgroup_cpu_quota() is None
    monkeypatch.setattr('builtins.open', fake(v1={'cpu.cfs_quota_us': '400000', 'cpu.cfs_period_us': '100000'}))
    assert hostenv.cgroup_cpu_quota() == 4.0
    monkeypatch.setattr('builtins.open', fake(v1={'cpu.cfs_quota_us': '-1', 'cpu.cfs_period_us': '100000'}))
    assert hostenv.cgroup_cpu_quota() is None


def test_bound_never_raises_the_thread_count(monkeypatch):
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv('MTL_HOST_THREADS', '0')
        assert hostenv.bound_torch_threads() == before                  # 0: leave torch alone
        monkeypatch.setenv('MTL_HOST_THREADS', str(before + 7))
        assert hostenv.bound_torch_threads() == before                  # a larger limit does not raise it
        monkeypatch.setenv('MTL_HOST_THREADS', '1')
        assert hostenv.bound_torch_threads() == 1 and torch.get_num_threads() == 1
        monkeypatch.delenv('MTL_HOST_THREADS')
        monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
        torch.set_num_threads(before)
        n = hostenv.bound_torch_threads()
        assert 1 <= n <= max(1, (3 * (hostenv.effective_cpus() // 8)) // 4) or n == before == 1
    finally:
        torch.set_num_threads(before)


def test_enqueue_path_staging_is_numpy_only():
    """The staging copy of prepare_tasks was a torch host-to-host copy_ of > 32768 elements: an OpenMP region on the whole pool.  The
    enqueue path must fill its pinned staging through numpy views (engine.prepare_tasks); pin the source so it stays that way."""
    src = open(os.path.join(ROOT, 'meta-transfer-learning_amd', 'engine.py')).read()
    body = src[src.index('def prepare_tasks'):src.index('def forward(self, theta')]
    assert "np.copyto(i32_np, meta_np)" in body and "st_i32.copy_(" not in body and "st_ids[0].copy_(" not in body
    # and numpy views of a torch tensor alias its memory (what the staging relies on)
    t = torch.zeros(8, dtype=torch.int32)
    np.copyto(t.numpy(), np.arange(8, dtype=np.int32))
    assert t.tolist() == list(range(8))
