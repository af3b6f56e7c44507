"""Does the 256 MB memory-side cache (MALL / Infinity Cache) serve a producer -> consumer hand-off?  A streaming consumer (bf16 multiply,
read n bytes + write n bytes) is timed (a) right after a producer wrote its input and (b) after 2 GB of unrelated traffic flushed the caches,
for tensor sizes around the cache size.  GPU box only."""
import torch

dev = torch.device("cuda:0")
flush = torch.empty(1 << 30, dtype=torch.bfloat16, device=dev)
for mb in (16, 32, 64, 96, 128, 192, 231, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 2
    src = torch.randn(n, device=dev).bfloat16()
    mid = torch.empty_like(src)
    out = torch.empty_like(src)
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(12):
            torch.mul(src, 1.5, out=mid)                 # producer writes `mid`
            if mode == "cold":
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.mul(mid, 0.5, out=out)                 # consumer reads `mid`
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    gb = 2.0 * mb / 1024
    print("%5d MB: consumer after its producer %.3f ms (%.2f TB/s)   after a flush %.3f ms (%.2f TB/s)" % (
        mb, res["warm"], gb / res["warm"], res["cold"], gb / res["cold"]))
