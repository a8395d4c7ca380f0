import sys, torch
sys.path.insert(0, '.')
from mocov2_whisper_flamingo_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for rows, C, with_res in ((375000, 256, True), (375000, 64, False), (101400, 512, True), (101400, 128, False), (29400, 1024, True), (9600, 2048, True)):
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    r = torch.randn(rows, C, device=dev).to(torch.bfloat16) if with_res else None
    sc, sh = torch.randn(C, device=dev), torch.randn(C, device=dev)
    t = timeit(lambda: ops.bn_apply(x, sc, sh, residual=r, relu=True, out=x))
    gb = rows * C * 2 * (3 if with_res else 2) / 1e9
    print(f"bn_apply rows={rows} C={C} res={with_res}: {t*1e6:7.1f} us  {gb/t/1e3:5.2f} TB/s")
