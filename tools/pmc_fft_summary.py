#!/usr/bin/env python3
"""Summarise the PMC passes of tools/pmc_fft_r06.sh per (kernel, grid): mean duration (unprofiled trace), VALU instructions per
wave, VALU-busy share, wait share, LDS instructions / bank conflicts, VMEM instructions, HBM-side traffic (FETCH_SIZE doubled per the
gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE) -> text table on stdout and summary.json beside the passes."""
import collections, csv, glob, json, re, sys

out = sys.argv[1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


want = re.compile(r"fft|cgemm")
rows = collections.defaultdict(dict)
tr = glob.glob(out + "/t/*/*kernel_trace.csv") + glob.glob(out + "/t/*kernel_trace.csv")
if tr:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if want.search(r["Kernel_Name"]):
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
            d[(short(r["Kernel_Name"]), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in d.items():
        v = sorted(v)[: max(1, len(v) - 1)] if len(v) > 2 else v         # drop the slowest (first-touch) launch
        rows[k]["us"] = sum(v) / len(v)
        rows[k]["calls"] = len(v)
for p in ("p1", "p2", "p3", "p4", "p5"):
    files = glob.glob(out + "/" + p + "/*/*counter_collection.csv") + glob.glob(out + "/" + p + "/*counter_collection.csv")
    if not files:
        print("# pass %s: no counter file" % p)
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if want.search(r["Kernel_Name"]):
            agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            rows[k][c] = sum(v) / len(v)
res = {}
print("%-34s %9s %8s %9s %7s %7s %7s %8s %8s %9s %9s %8s" % ("kernel", "grid", "us", "valu/wave", "valuPipe%", "wait%", "salu/wv", "ldsconf%", "vmem/wv", "fetchMB", "writeMB", "TB/s"))
for k in sorted(rows):
    m = rows[k]
    w = max(m.get("SQ_WAVES", 1.0), 1.0)
    busy = max(m.get("SQ_BUSY_CYCLES", 1.0), 1.0)
    wc = max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    fetch = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024 if "FETCH_SIZE" in m else None      # KB -> B, x2 (gfx950)
    write = m.get("WRITE_SIZE", 0.0) * 1024 if "WRITE_SIZE" in m else None
    us = m.get("us")
    rec = {"grid_threads": k[1], "mean_us": None if us is None else round(us, 1), "waves": int(w),
           "valu_insts_per_wave": round(m.get("SQ_INSTS_VALU", 0) / w, 1), "salu_insts_per_wave": round(m.get("SQ_INSTS_SALU", 0) / w, 1),
           "lds_insts_per_wave": round(m.get("SQ_INSTS_LDS", 0) / w, 1),
           "vmem_rd_per_wave": round(m.get("SQ_INSTS_VMEM_RD", 0) / w, 1), "vmem_wr_per_wave": round(m.get("SQ_INSTS_VMEM_WR", 0) / w, 1),
           "valu_busy_pct_of_sq_busy": round(100 * m.get("SQ_ACTIVE_INST_VALU", 0) / busy / 4, 1),
           "wait_inst_any_pct_of_wave_cycles": round(100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 1),
           "wait_any_pct_of_wave_cycles": round(100 * m.get("SQ_WAIT_ANY", 0) / wc, 1) if "SQ_WAIT_ANY" in m else None,
           "lds_active_pct_of_sq_busy": round(100 * m.get("SQ_ACTIVE_INST_LDS", 0) / busy / 4, 1),
           "lds_bank_conflict_pct_of_lds_active": round(100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_ACTIVE_INST_LDS", 1), 1), 1),
           "gui_active_mcycles": round(m.get("GRBM_GUI_ACTIVE", 0) / 1e6, 3),
           # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs
           "valu_pipe_busy_pct": round(100 * m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * max(m.get("GRBM_GUI_ACTIVE", 8) / 8, 1)), 1),
           "clock_ghz_under_counters": None if us is None or "GRBM_GUI_ACTIVE" not in m else round(m["GRBM_GUI_ACTIVE"] / 8 / us / 1e3, 2),
           "active_inst_any_pct_of_wave_cycles": round(100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 1) if "SQ_ACTIVE_INST_ANY" in m else None,
           "fetch_bytes": fetch, "write_bytes": write,
           "hbm_tb_per_s": None if (us is None or fetch is None or write is None) else round((fetch + write) / us / 1e6, 2)}
    res["%s @%d" % k] = rec
    print("%-34s %9d %8.1f %9.0f %7.1f %7.1f %7.1f %8.1f %8.1f %9.1f %9.1f %8s" % (
        k[0][:34], k[1], us or -1, rec["valu_insts_per_wave"], rec["valu_pipe_busy_pct"], rec["wait_inst_any_pct_of_wave_cycles"],
        rec["salu_insts_per_wave"], rec["lds_bank_conflict_pct_of_lds_active"], rec["vmem_rd_per_wave"] + rec["vmem_wr_per_wave"],
        (fetch or 0) / 1e6, (write or 0) / 1e6, rec["hbm_tb_per_s"]))
json.dump({"note": "tools/pmc_fft_r06.sh: separate rocprofv3 --pmc passes over tests/diag/fft_train_kernels.py (B = 20; training plans); "
                   "FETCH_SIZE doubled per the gfx950 correction; valu_pipe_busy_pct = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); "
                   "wait_* and active_* are shares of SQ_WAVE_CYCLES (a wave's resident time)",
           "kernels": res}, open(out + "/summary.json", "w"), indent=1)
