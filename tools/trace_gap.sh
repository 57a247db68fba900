#!/bin/bash
# GPU box: kernel + memory-copy + HIP-runtime trace of a few training steps, to see what the host is doing while the GPU idles
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/trace_gap
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $out/t -o g -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline "$@" > $out/bench.json 2> $out/bench.err
cd $R
python3 - $out <<'PY'
import csv, sys, glob
out = sys.argv[1]
def load(pat):
    f = glob.glob(out + "/t/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = load("kernel_trace.csv"); m = load("memory_copy_trace.csv"); h = load("hip_api_trace.csv")
k.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(k) if "adam_kernel" in r["Kernel_Name"]]
a = ad[-1]
prev_end = max(int(r["End_Timestamp"]) for r in k[ad[-2] + 1:a])
start = int(k[a]["Start_Timestamp"])
print("gap before the last adam kernel: %.3f ms" % ((start - prev_end) / 1e6))
print("memory copies inside the gap:")
for r in m:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > prev_end - 200000 and s < start + 200000:
        print("   %s %s bytes  [%.3f .. %.3f ms rel]" % (r.get("Direction", r.get("Kind", "?")), r.get("Bytes", r.get("Size", "?")), (s - prev_end) / 1e6, (e - prev_end) / 1e6))
print("HIP calls of the host overlapping the gap (rel. ms to the end of the last backward kernel):")
for r in h:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > prev_end - 300000 and s < start + 100000 and (e - s) > 20000:
        print("   %-40s %.3f .. %.3f" % (r["Function"], (s - prev_end) / 1e6, (e - prev_end) / 1e6))
# host enqueue position: when was the adam launch call issued relative to the gap?
for r in h:
    if r["Function"] in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel"):
        pass
PY
