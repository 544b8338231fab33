"""Developer utility: what a plain device-to-device copy reaches on this box (the practical ceiling the G2P launch is compared with)."""
import torch, time
for mb in (64, 85, 128, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.ones(n, dtype=torch.float32, device="cuda")
    for _ in range(5): a.copy_(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps): a.copy_(b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy {mb:5d} MB -> {mb} MB: {ms*1e3:8.1f} us, {2*mb*1.048576/ms:7.1f} GB/s (read + write)")
