"""How many CPUs this process may actually use: the scheduler affinity AND the cgroup CPU quota (a pod limited to 16 CPUs on a host
with 256 hardware threads still sees 256 in os.cpu_count() / sched_getaffinity -- and 256 busy threads under a 16-CPU quota are
throttled into a crawl: torch.mm with 256 threads ran 26 GFLOP/s on the round-6 GPU box, profiles/r06_cpu_baseline_torch_256_threads.log).
TEST / MEASUREMENT INFRASTRUCTURE (the CPU comparators of bench.py)."""
import math
import os


def affinity_cpus() -> int:
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def quota_cpus():
    """ceil(quota / period) of the cgroup this process lives in (v2 cpu.max, v1 cpu.cfs_quota_us), or None when unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                return max(1, math.ceil(int(q) / int(p)))
            return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        return max(1, math.ceil(q / p)) if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus() -> int:
    q = quota_cpus()
    a = affinity_cpus()
    return min(a, q) if q else a
