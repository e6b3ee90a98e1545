#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<round>/) into the committed
summaries: profiles/<round>/*.csv (copies), profiles/hbm_traffic.json and profiles/valu_util.json
(read by bench.py).  Usage: python tools/summarize_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats.csv"))
for name in ("fetch", "write", "valu", "wait"):
    shutil.copy(one(f"{name}/**/*counter_collection.csv"), os.path.join(dst, f"pmc_{name}.csv"))
shutil.copy(os.path.join(src, "bench_n1.json"), os.path.join(dst, "bench_n1.json"))


# the dominant kernel: the render_kernel instantiation with the most time in the trace (with the guarded walk
# a second instantiation, the exact re-walk of flagged samples, runs once per pass too)
stats_rows = [r for r in csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))) if "render_kernel" in r["Name"]]
stats_rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
MAIN = stats_rows[0]["Name"]
REWORK = stats_rows[1]["Name"] if len(stats_rows) > 1 else None


def means(name):
    """{kernel kind: {counter: mean over launches}} and launch counts"""
    acc, cnt = {}, {}
    for r in csv.DictReader(open(os.path.join(dst, f"pmc_{name}.csv"))):
        kn = r["Kernel_Name"]
        kind = ("trace" if MAIN.startswith(kn) or kn.startswith(MAIN) else "rework" if "render_kernel" in kn
                else "accumulate_list" if "accumulate_kernel<true>" in kn else "accumulate" if "accumulate_kernel" in kn
                else "primary" if "primary_" in kn else "candidates" if "cand_kernel" in kn else None)
        if kind is None:
            continue
        key = (kind, r["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
        cnt[key] = cnt.get(key, 0) + 1
    return {k: acc[k] / cnt[k] for k in acc}, cnt


f, fc = means("fetch")
w, _ = means("write")
bench = json.loads(open(os.path.join(dst, "bench_n1.json")).read().strip().splitlines()[-1])
launches = fc[("trace", "FETCH_SIZE")]
traffic = {
    "_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) around `python bench.py --steps 1 "
            "--warmup 0 --no-cpu-baseline` (tools/profile_bench.sh); means over the trace launches of the frame; values in KiB as "
            "reported; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports half "
            f"of wide coalesced reads). Raw rows: profiles/{rnd}/pmc_fetch.csv, pmc_write.csv.",
    "trace_kernel": MAIN,
    "trace_launches_per_frame": launches,
    "trace_kernel_fetch_kib": f[("trace", "FETCH_SIZE")],
    "trace_kernel_write_kib": w[("trace", "WRITE_SIZE")],
    "accumulate_kernel_fetch_kib": f[("accumulate", "FETCH_SIZE")],
    "accumulate_kernel_write_kib": w[("accumulate", "WRITE_SIZE")],
    "bytes_per_trace_launch": int((2 * f[("trace", "FETCH_SIZE")] + w[("trace", "WRITE_SIZE")]) * 1024),
    "bytes_per_rework_launch": int((2 * f.get(("rework", "FETCH_SIZE"), 0) + w.get(("rework", "WRITE_SIZE"), 0)) * 1024),
    "bytes_per_accumulate_launch": int((2 * f[("accumulate", "FETCH_SIZE")] + w[("accumulate", "WRITE_SIZE")]) * 1024),
    "bytes_per_accumulate_list_launch": int((2 * f.get(("accumulate_list", "FETCH_SIZE"), 0.0) + w.get(("accumulate_list", "WRITE_SIZE"), 0.0)) * 1024),
    "bytes_per_primary_launch": int((2 * f.get(("primary", "FETCH_SIZE"), 0.0) + w.get(("primary", "WRITE_SIZE"), 0.0)) * 1024),
    "bytes_per_candidates_launch": int((2 * f.get(("candidates", "FETCH_SIZE"), 0.0) + w.get(("candidates", "WRITE_SIZE"), 0.0)) * 1024),
}
rows_bytes = bench["roofline"]["samples_per_launch"] * 12
traffic["slab_rows_bytes_per_launch"] = rows_bytes
# Sky pixels have no rows any more (nobody writes or reads them); the primary pass writes every OTHER row exactly once, in whole
# lines (1.003 x when it still wrote all rows) — its write bytes are the size of the rows the trace kernel then overwrites
traced_rows = w.get(("primary", "WRITE_SIZE"), 0.0) * 1024 or rows_bytes
traffic["traced_rows_bytes_per_launch"] = int(traced_rows)
traffic["trace_write_amplification"] = round(w[("trace", "WRITE_SIZE")] * 1024 / traced_rows, 3)
traffic["frame_total_bytes"] = sum(traffic[k] for k in ("bytes_per_trace_launch", "bytes_per_rework_launch", "bytes_per_accumulate_launch",
                                                         "bytes_per_accumulate_list_launch", "bytes_per_primary_launch", "bytes_per_candidates_launch")) * int(launches and 1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)

v, _ = means("valu")
q, _ = means("wait")
cycles = q[("trace", "GRBM_GUI_ACTIVE")] / 8.0           # summed over the 8 XCDs
issue = 2.0 * v[("trace", "SQ_INSTS_VALU")] / (1024.0 * cycles)
lanes = v[("trace", "SQ_THREAD_CYCLES_VALU")] / (64.0 * v[("trace", "SQ_ACTIVE_INST_VALU")])
launch_ms = bench["roofline"]["launch_ms"]
util = {
    "_how": f"rocprofv3 --pmc passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline` (profiles/{rnd}/pmc_valu.csv, "
            "pmc_wait.csv), means over the trace launches of a frame: issue = 2 cycles x SQ_INSTS_VALU / (1024 SIMDs x "
            "GRBM_GUI_ACTIVE/8); lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); peak = 256 CU x 128 lane-ops/clk x "
            "2.4 GHz = 78.6 T lane-ops/s (MI355X_MICROARCH.md: 157.3 TFLOPS fp32 vector = 2 flops per lane-op).",
    "valu_issue_utilisation": round(issue, 3),
    "valu_lane_utilisation": round(lanes, 3),
    "valu_roofline_frac": round(issue * lanes, 3),
    "valu_wave_instructions_per_launch": v[("trace", "SQ_INSTS_VALU")],
    "shader_cycles_per_launch": cycles,
    "clock_ghz": round(cycles / (launch_ms * 1e6), 3),
    "wave_cycles_in_waitcnt_frac": round(q[("trace", "SQ_WAIT_ANY")] / v[("trace", "SQ_WAVE_CYCLES")], 3),
    "wave_cycles_issue_stalled_frac": round(q[("trace", "SQ_WAIT_INST_ANY")] / v[("trace", "SQ_WAVE_CYCLES")], 3),
    "wave_cycles_issuing_frac": round(q[("trace", "SQ_ACTIVE_INST_ANY")] / v[("trace", "SQ_WAVE_CYCLES")], 3),
    "lds_bank_conflict_cycles_per_lds_active": round(v[("trace", "SQ_LDS_BANK_CONFLICT")] / max(q[("trace", "SQ_LDS_IDX_ACTIVE")], 1.0), 3),
}
json.dump(util, open(os.path.join(ROOT, "profiles", "valu_util.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(util, indent=1))
for r in csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
