import os, time, multiprocessing as mp
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "n/a")
def burn(_):
    t=time.perf_counter(); x=0
    while time.perf_counter()-t < 1.0: x+=1
    return x
for n in (1, 8, 16, 32, 64, 128):
    with mp.Pool(n) as p:
        t=time.perf_counter(); r=p.map(burn, range(n)); dt=time.perf_counter()-t
    print(n, "procs: total work %.1f x single, wall %.2f" % (sum(r)/r[0] if n==1 else sum(r)/base, dt)) if n>1 else None
    if n==1: base=r[0]; print("single", base)
