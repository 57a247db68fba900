#!/usr/bin/env python3
"""Fail if any kernel whose loop is barrier-paced bursts of 16-bit matrix instructions (gemm_x3*, conv_*_bf16*) ever ran
CONCURRENTLY with a frequency-domain kernel (fft*, ifft*, cgemm*) in a rocprofv3 kernel trace.

Why (profiles/r03_neighbour_mfma.txt): on this hardware such a neighbour changes the results of FFT-type
kernels that run beside it on the same GPU (rocFFT included).  Inside one process the library never runs the two at once --
the tape is one stream; the second stream of the frequency-domain backward only carries frequency-domain kernels and is
joined before the next layer -- and this tool turns that argument into a check on real traces (tools/prof_step.sh runs it
for every configuration it profiles).

usage: check_no_mfma16_beside_fft.py <kernel_trace.csv> [--label TEXT]      exit code 0: no overlap, 1: overlaps found
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_family import FFT, MFMA16  # noqa: E402


def check(path):
    a, b = [], []
    n = 0
    for r in csv.DictReader(open(path)):
        n += 1
        name = r["Kernel_Name"]
        iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name)
        if MFMA16.search(name):
            a.append(iv)
        elif FFT.search(name):
            b.append(iv)
    a.sort()
    b.sort()
    hits = []
    j = 0
    for s, e, na in a:                      # both lists sorted by start: sweep
        while j < len(b) and b[j][1] <= s:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][1] > s:
                hits.append((na, b[k][2], min(e, b[k][1]) - max(s, b[k][0])))
            k += 1
    return n, len(a), len(b), hits


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    label = sys.argv[sys.argv.index("--label") + 1] if "--label" in sys.argv else os.path.basename(os.path.dirname(sys.argv[1]))
    n, na, nb, hits = check(sys.argv[1])
    print("%s: %d dispatches, %d 16-bit-matrix kernels (gemm_x3* / conv_*_bf16*), %d frequency-domain kernels (fft* / cgemm*): "
          "%d overlapping pairs%s" % (label, n, na, nb, len(hits), "" if hits else "  OK"))
    for h in hits[:10]:
        print("   OVERLAP %.1f us: %s  ||  %s" % (h[2] / 1e3, h[0][:60], h[1][:60]))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
