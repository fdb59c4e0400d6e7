#!/usr/bin/env python3
"""Turn the profiler output of tools/profile_round.sh (gpurun_out/<tag>_*) into the files kept under
profiles/: <tag>_kernel_stats_{graph,eager}.csv, <tag>_summary.md and <tag>_traffic.json."""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "gpurun_out"
prof = ROOT / "profiles"
prof.mkdir(exist_ok=True)


def one(pattern):
    hits = glob.glob(str(out / pattern), recursive=True)
    return max(hits, key=lambda f: Path(f).stat().st_mtime) if hits else None   # newest run wins


def stats_table(mode):
    f = one("%s_stats_%s/**/*kernel_stats.csv" % (tag, mode))
    if not f:
        return None, None
    shutil.copy(f, prof / ("%s_kernel_stats_%s.csv" % (tag, mode)))
    rows = list(csv.DictReader(open(f)))
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %.3f | %.2f | %s |" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e3, r["Percentage"]))
    return "\n".join(lines), rows


def counter(kind, name, kernel_substr):
    f = one("%s_pmc_%s_%s/**/*counter_collection.csv" % (tag, kind, name))
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals.sort()
    return vals[len(vals) // 2] if vals else None


md = ["# %s - rocprofv3 evidence (MI355X, batch 32 x 30 s, G300)" % tag, ""]
for mode, cmd in (("graph", "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"),
                  ("eager", "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph")):
    table, _ = stats_table(mode)
    if table:
        log = out / ("%s_stats_%s.log" % (tag, mode))
        last = [l for l in log.read_text().splitlines() if l.startswith("{")]
        md += ["## `rocprofv3 --kernel-trace --stats -- %s`" % cmd, "",
               "Warm-up and timed forwards of the resident batch, the same batch through the upload + CTC path, and the",
               "event-timed launches (21 each) of the kernels bench.py reports",
               "roofline lines for: the three GEMM shapes of a block with the kernels that feed them, attention, log-mel.",
               "`gemm_bf16_kernel<E, L>`: E = epilogue (0 bias, 1 bias+Swish, 2 GLU, 3 float32 residual, 4 float32 bias),",
               "L = true for K >= 2048 (the FFN down-projection), false for the K = 1024 projections.", "", table, ""]
        if last:
            md += ["bench line of this run:", "", "```", last[-1], "```", ""]

# BASELINE configs[1] and configs[4]: kernel statistics and the bench line of the profiled run
for mode, title, cmd in (("b1", "BASELINE configs[1]: one 30 s utterance", "python3 bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline"),
                         ("g7b_fp8", "BASELINE configs[4]: 7B geometry, fp8 projections, batch 8 x 30 s",
                          "python3 bench.py --model G7B --batch 8 --fp8 --steps 10 --warmup 3 --no-cpu-baseline")):
    table, _ = stats_table(mode)
    if table:
        log = out / ("%s_stats_%s.log" % (tag, mode))
        last = [l for l in log.read_text().splitlines() if l.startswith("{")]
        md += ["## %s - `rocprofv3 --kernel-trace --stats -- %s`" % (title, cmd), "", table, ""]
        if last:
            (prof / ("%s_bench_%s.json" % (tag, mode))).write_text(last[-1] + "\n")
            md += ["bench line of this run (also `%s_bench_%s.json`):" % (tag, mode), "", "```", last[-1], "```", ""]

traffic = {}
KB = 1024.0
for kind, substr, key, shape, algo in (
        ("gemm", "gemm_bf16_kernel<1", "gemm_bf16_kernel<1>", "48128x4096x1024",
         48128 * 1024 * 2 + 4096 * 1024 * 2 + 4096 * 4 + 48128 * 4096 * 2),
        # the fp8 FFN up-projection of BASELINE configs[4]: e4m3 operands, bf16 output, row scales and bias
        ("gemm8", "gemm_fp8_kernel<1", "gemm_fp8_kernel<1>", "12032x16384x4096",
         12032 * 4096 + 16384 * 4096 + (12032 + 2 * 16384) * 4 + 12032 * 16384 * 2),
        ("logmel", "logmel_kernel", "logmel_kernel", "32x480000", 32 * (480000 * 4 + 2998 * 80 * 4)),
        # the two small kernels after it: per-utterance statistics (32 x 8 partial blocks x 160 floats in, 160 out)
        # and normalise + stack (log-mel in, stacked float32 features [1504][160] out)
        ("logmel", "stats_finalize_kernel", "stats_finalize_kernel", "32 utterances", 32 * (8 * 160 * 4 + 160 * 4)),
        ("logmel", "norm_stack_kernel", "norm_stack_kernel", "32x2998x80", 32 * (2998 * 80 * 4 + 1504 * 160 * 4))):
    fetch = counter(kind, "FETCH_SIZE", substr)
    write = counter(kind, "WRITE_SIZE", substr)
    if fetch is None or write is None:
        continue
    # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies each 128-byte request of a wide
    # streaming read as 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
    traffic[key] = {"shape": shape, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
                    "fetch_bytes_corrected": fetch * KB * 2, "write_bytes": write * KB,
                    "traffic_bytes_per_launch": fetch * KB * 2 + write * KB,
                    "algorithmic_bytes_per_launch": algo}
(prof / ("%s_traffic.json" % tag)).write_text(json.dumps(traffic, indent=1) + "\n")
md += ["## HBM-side bytes per launch (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs)", "",
       "FETCH_SIZE doubled (gfx950 counts a 128-byte request as 64 bytes); Infinity-Cache hits are included.", "",
       "| kernel | shape | FETCH raw KiB | WRITE raw KiB | traffic MB | algorithmic MB |", "|---|---|---|---|---|---|"]
for k, v in traffic.items():
    md.append("| `%s` | %s | %.0f | %.0f | %.1f | %.1f |" % (k, v["shape"], v["FETCH_SIZE_KiB_raw"], v["WRITE_SIZE_KiB_raw"],
                                                          v["traffic_bytes_per_launch"] / 1e6,
                                                          v["algorithmic_bytes_per_launch"] / 1e6))
# SQ / GRBM counter passes (tools/pmc_kernels.sh), when they were collected for this tag
import subprocess
pmc_dir = out / ("%s_pmc" % tag)
if pmc_dir.exists():
    table = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_table.py"), str(pmc_dir), "--md"],
                           capture_output=True, text=True).stdout
    (prof / ("%s_pmc.md" % tag)).write_text(
        "# %s - SQ / GRBM counters per launch (`tools/pmc_kernels.sh`, one `rocprofv3 --pmc` run per pass and kernel)\n\n"
        "Targets: ffn_up = `gemm_bf16_kernel<1, false>` 48128x4096x1024, ffn_down = `gemm_bf16_kernel<3, true>` 48128x1024x4096,\n"
        "dxd_res = `gemm_bf16_kernel<3, false>` 48128x1024x1024, attn64 = `attention_relkey_kernel<64>` 32 x 16 heads x 1499.\n"
        "Median over the launches of a run.  SQ_*_CYCLES of waves count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES and\n"
        "GRBM_GUI_ACTIVE count cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_INSTS_VALU includes the MFMAs.\n\n" % tag + table)
    md += ["## Matrix-pipe utilisation and stall counters", "", "See `%s_pmc.md`." % tag, ""]
(prof / ("%s_summary.md" % tag)).write_text("\n".join(md) + "\n")
print("\n".join(md))
