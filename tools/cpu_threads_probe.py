import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
try:
    print("affinity", len(os.sched_getaffinity(0)))
except Exception as e:
    print(e)
a = torch.randn(2056, 1024); w = torch.randn(4096, 1024)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    torch.nn.functional.linear(a, w)
    t0 = time.perf_counter()
    for _ in range(5):
        torch.nn.functional.linear(a, w)
    dt = (time.perf_counter() - t0) / 5
    print(th, "threads", round(2 * 2056 * 1024 * 4096 / dt / 1e9, 1), "GFLOP/s")
