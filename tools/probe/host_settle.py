"""How steady is the host right after a box comes up?  Times a fixed pure-Python workload every ~0.25 s for 15 s (first command of a gpurun call)."""
import time, os
def work():
    t = time.perf_counter(); s = 0
    for i in range(300000): s += i * i
    return time.perf_counter() - t
t0 = time.time(); out = []
while time.time() - t0 < 15:
    out.append((round(time.time() - t0, 2), round(work() * 1e3, 1)))
    time.sleep(0.2)
print('load1', os.getloadavg(), 'cpus', os.cpu_count())
print(' '.join('%s:%s' % p for p in out))
