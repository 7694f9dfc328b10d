import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
os.system('lscpu | head -20; free -g | head -2')
x = torch.randn(8, 64, 150, 150); w = torch.randn(128, 64, 3, 3)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    torch.nn.functional.conv2d(x, w, padding=1)
    t = time.time()
    for _ in range(3): torch.nn.functional.conv2d(x, w, padding=1)
    dt = (time.time() - t) / 3
    print(nt, 'threads conv', round(dt * 1e3, 1), 'ms', round(2 * 8 * 150 * 150 * 128 * 64 * 9 / dt / 1e9, 1), 'GFLOP/s')
