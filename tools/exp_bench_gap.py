import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
cfg = bench.CONFIGS['c2']
e = bench.HipEngine('c2', cfg, 0, 65536, 4096, 'frame', 0, 0)
for _ in range(800): e.step()
e.sync()
for K in (20, 200):
    for mode in ('events', 'plain'):
        e.sync(); t0 = time.perf_counter()
        if mode == 'events':
            d = e.timed_steps(K)
        else:
            for _ in range(K): e.step()
        e.sync(); dt = time.perf_counter() - t0
        print(K, mode, 'ms/step', round(dt / K * 1e3, 4), ('kernel median %.4f' % sorted(d())[K // 2]) if mode == 'events' else '')
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e.sync(); a.record(e.stream)
for _ in range(200): e.step()
b.record(e.stream); e.sync(); print('one event pair over 200 steps: ms/step', round(a.elapsed_time(b) / 200, 4))
