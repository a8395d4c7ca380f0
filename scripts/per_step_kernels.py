"""Launches per benchmark step, by kernel: the difference of two `rocprofv3 --kernel-trace --stats` summaries of bench.py
runs that differ only in --steps (set-up work - synthetic batches, weight packing, the eager passes before the
recordings - cancels out).   python3 scripts/per_step_kernels.py STATS_A.csv STEPS_A STATS_B.csv STEPS_B"""
import csv, json, sys

def load(path):
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}

a, na, b, nb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
out, native = [], 0.0
for name in sorted(set(a) | set(b)):
    ca, ta = a.get(name, (0, 0))
    cb, tb = b.get(name, (0, 0))
    per = (cb - ca) / (nb - na)
    if per == 0:
        continue
    out.append({"kernel": name[:110], "launches_per_step": round(per, 2), "us_per_step": round((tb - ta) / (nb - na) / 1e3, 1)})
    if "at::native" in name or "rocclr" in name:
        native += per
out.sort(key=lambda e: -e["us_per_step"])
print(json.dumps({"steps": [na, nb], "torch_native_or_runtime_copy_launches_per_step": round(native, 2),
                  "launches_per_step": round(sum(e["launches_per_step"] for e in out), 1),
                  "kernel_us_per_step": round(sum(e["us_per_step"] for e in out), 1), "kernels": out}, indent=1))
