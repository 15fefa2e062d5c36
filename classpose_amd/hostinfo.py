"""What the host really grants this process: CPUs of the affinity mask capped by the cgroup CPU quota.

A container may show 256 CPUs (``os.cpu_count()``) and be throttled to 16 cores of CPU time: thread pools and torch's intra-op pool sized by
the former only fight each other for the latter (round 5, configs[2] through the CLI on such a box: reader threads, the background weight
conversion and GrandQC's thumbnail rendering together ran 2-3x slower EACH than one after the other)."""
from __future__ import annotations

import os


def cgroup_cpu_limit() -> float | None:
    """CPU time this process tree may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:                                   # noqa: BLE001
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except Exception:                                   # noqa: BLE001
        return None


def usable_cpus() -> int:
    """threads worth starting for CPU-bound work: affinity mask, capped by the cgroup quota (at least 1)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:                              # pragma: no cover
        n = os.cpu_count() or 1
    q = cgroup_cpu_limit()
    if q is not None:
        n = min(n, max(1, int(q + 0.5)))
    return max(1, n)


def limit_torch_threads() -> int:
    """cap torch's intra-op pool at usable_cpus(); returns the value in force"""
    import torch
    n = usable_cpus()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
