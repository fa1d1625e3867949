#!/bin/bash
# What the host of the GPU box gives this process: logical CPUs, affinity, cgroup quota, NUMA nodes, a STREAM-like copy rate with the oracle's thread count
echo "nproc: $(nproc)  nproc --all: $(nproc --all)"
python - <<'PY'
import os
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
PY
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|MHz" | head -12
free -g | head -2
python - <<'PY'
import numpy as np, time, ctypes, os
from oracle import grb_oracle as O
O.use_all_threads(); print("oracle threads", O.num_threads())
# copy rate of numpy (one thread) as a floor reference
a = np.ones(1 << 28, np.float32); b = np.empty_like(a)
t = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t
print("numpy copy 1 GiB+1 GiB: %.1f GB/s" % (2 * a.nbytes / dt / 1e9))
PY
