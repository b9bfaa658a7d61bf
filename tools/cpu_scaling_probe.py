"""CPU oracle throughput vs OpenMP thread count on this box (picks cpu_baseline's thread count)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as O
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
sc = [O.OracleScene("rgbbox"), O.OracleScene("irreg")]
for th in (8, 16, 32, 64, 96, 128, 192, 256):
    if th > os.cpu_count():
        break
    rays, t0 = 0, time.perf_counter()
    for s in sc:
        _, cnt = s.render(1000, 1000, threads=th)
        rays += cnt["rays"]
    dt = time.perf_counter() - t0
    print(f"threads {th}: {rays / dt / 1e6:.1f} Mray/s", flush=True)
