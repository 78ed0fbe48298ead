"""Is the validation loop's periodic 60-80 ms stall CPU-quota throttling of the cgroup? (diagnostic)"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def cpustat():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            return dict(l.split() for l in open(p).read().strip().splitlines())
    return {}
def cpumax():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(p):
            return open(p).read().strip()
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "cpu.max", cpumax())
a = cpustat()
nt = sys.argv[1] if len(sys.argv) > 1 else None
env = dict(os.environ)
if nt:
    env["OMP_NUM_THREADS"] = nt
t0 = time.time()
out = subprocess.run([sys.executable, "tools/bench_eval.py"], capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
b = cpustat()
print("OMP_NUM_THREADS", nt, "|", out)
print({k: int(b[k]) - int(a[k]) for k in b if k in a}, f"wall {time.time() - t0:.1f}s")
