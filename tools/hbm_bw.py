import torch, time
dev = torch.device("cuda", 0)
n = 1 << 30   # 1 Gi halfs = 2 GB
a = torch.empty(n, dtype=torch.float16, device=dev); b = torch.empty(n, dtype=torch.float16, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.zero_()); print("fill  2 GB: %.3f ms  %.2f TB/s write" % (ms, 2.147 / ms))
ms = t(lambda: b.copy_(a)); print("copy  2 GB: %.3f ms  %.2f TB/s read + %.2f TB/s write" % (ms, 2.147 / ms, 2.147 / ms))
ms = t(lambda: a.sum()); print("sum   2 GB: %.3f ms  %.2f TB/s read" % (ms, 2.147 / ms))
ms = t(lambda: torch.add(a, 1, out=b)); print("add   2 GB: %.3f ms  %.2f TB/s each way" % (ms, 2.147 / ms))
