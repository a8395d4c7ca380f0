"""LayerNorm forward on the Whisper-small residual stream ([12000, 768] fp32 -> bf16), per library variant."""
import glob, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, '.')
    from mocov2_whisper_flamingo_amd import ops
    dev = torch.device('cuda:0')
    for rows, D in ((12000, 768), (12000, 1024), (600, 512)):
        x = torch.randn(rows, D, device=dev)
        g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        f = lambda: ops.layernorm_fwd(x, g, b, out_dtype=torch.bfloat16)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        print(f"  LN [{rows},{D}]: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
else:
    for f in sorted(glob.glob('scripts/variants/libavsr_*.so')):
        print(os.path.basename(f))
        print(subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, AVSR_HIP_LIB=os.path.abspath(f)),
                             capture_output=True, text=True).stdout, end="")
