import torch, time
n = 1420*1000*1000 // 4
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
a.normal_()
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): b.copy_(a)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("copy 1.42 GB -> 1.42 GB: %.3f ms = %.2f TB/s (read+write)" % (ms, 2 * n * 4 / ms / 1e9))
s.record()
for _ in range(10): b.fill_(1.0)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("fill 1.42 GB: %.3f ms = %.2f TB/s" % (ms, n * 4 / ms / 1e9))
s.record()
for _ in range(10): x = a.sum()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("read-reduce 1.42 GB: %.3f ms = %.2f TB/s" % (ms, n * 4 / ms / 1e9))
