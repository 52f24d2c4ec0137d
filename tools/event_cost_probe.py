"""Host cost of recording a timing event, per process (first GPU process on a fresh box vs later ones)."""
import time, torch
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
s = torch.cuda.current_stream()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(400)]
for rep in range(3):
    t0 = time.perf_counter()
    for e in evs[:200]:
        e.record(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for a, b in zip(evs[:100], evs[100:200]):
        a.elapsed_time(b)
    t3 = time.perf_counter()
    print("rep %d: record %.1f us each, sync %.2f ms, elapsed_time %.1f us each" % (rep, (t1 - t0) / 200 * 1e6, (t2 - t1) * 1e3, (t3 - t2) / 100 * 1e6))
# interleaved with kernels
t0 = time.perf_counter()
for i in range(100):
    x.add_(1.0)
    evs[200 + i].record(s)
    x.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("interleaved with kernels: %.1f us per (2 launches + record)" % ((t1 - t0) / 100 * 1e6))
