"""Does a collective on a busy stream block the calling thread?  One-rank RCCL communicator; prints the host time
of each call while a long chain of kernels is still queued on the stream."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
g = torch.zeros(38_000_000, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
s = torch.cuda.Stream()
for async_op in (False, True):
    with torch.cuda.stream(s):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                b = a @ a                      # ~20 x 1.5 ms queued
            t1 = time.perf_counter()
            works = []
            for lo, hi in ((0, 16_000_000), (16_000_000, 32_000_000), (32_000_000, 38_000_000)):
                w = dist.all_reduce(g[lo:hi], async_op=async_op)
                works.append(w)
            t2 = time.perf_counter()
            if async_op:
                for w in works: w.wait()
            t3 = time.perf_counter()
            g.mul_(1.0)
            t4 = time.perf_counter()
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            print(f"async={async_op} queue {1e3*(t1-t0):.2f} ms  all_reduce x3 {1e3*(t2-t1):.2f} ms  wait {1e3*(t3-t2):.2f} ms  mul {1e3*(t4-t3):.2f}  drain {1e3*(t5-t4):.2f} ms", flush=True)
dist.destroy_process_group()
