"""HBM bandwidth probe on the GPU box: pure write (fill), pure read (sum), copy.  python tools/hbm_bw_probe.py"""
import torch

dev = torch.device("cuda", 0)
n = 1 << 30  # 4 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 4 / 1e9
t = timed(lambda: a.fill_(1.0))
print(f"write (fill_)      : {t:.3f} ms  {gb / t:.2f} TB/s")
t = timed(lambda: a.zero_())
print(f"write (memset)     : {t:.3f} ms  {gb / t:.2f} TB/s")
t = timed(lambda: a.sum())
print(f"read  (sum)        : {t:.3f} ms  {gb / t:.2f} TB/s")
t = timed(lambda: b.copy_(a))
print(f"copy  (read+write) : {t:.3f} ms  {2 * gb / t:.2f} TB/s")
t = timed(lambda: torch.add(a, 1.0, out=b))
print(f"add   (read+write) : {t:.3f} ms  {2 * gb / t:.2f} TB/s")
