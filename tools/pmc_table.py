#!/usr/bin/env python3
"""Table of the counter passes collected by tools/pmc_kernels.sh: median per-launch value of every counter
for each target's kernel, plus the derived MFMA utilisation.  Usage: pmc_table.py <dir> [--md]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
targets = {"ffn_up": "gemm_bf16_kernel<1", "ffn_down": "gemm_bf16_kernel<3", "dxd_res": "gemm_bf16_kernel<3", "attn64": "attention",
           "dwconv": "dwconv_ln_swish"}
table = collections.OrderedDict()
for name, needle in targets.items():
    acc = collections.defaultdict(list)
    newest = {}   # one file per pass directory: results of earlier runs merged into the same tree are skipped
    for f in glob.glob(os.path.join(root, name + "_p*", "**", "*counter_collection.csv"), recursive=True):
        key = os.path.relpath(f, root).split(os.sep)[0]
        if key not in newest or os.path.getmtime(f) > os.path.getmtime(newest[key]):
            newest[key] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        table[name] = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
names = sorted({k for t in table.values() for k in t})
md = "--md" in sys.argv
sep = " | " if md else "  "
print(("| " if md else "") + sep.join(["counter".ljust(28)] + [n.rjust(14) for n in table]) + (" |" if md else ""))
if md:
    print("|" + "---|" * (len(table) + 1))
for k in names:
    print(("| " if md else "") + sep.join([k.ljust(28)] + [("%14.4g" % t[k]) if k in t else " " * 14 for t in table.values()]) + (" |" if md else ""))
# derived: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel ran for gui_active / 8 cycles, during
# which the chip's 1024 SIMDs (4 x 256 CUs) each offered one matrix-pipe cycle per cycle
derived = collections.OrderedDict()
for n, t in table.items():
    d = {}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "GRBM_GUI_ACTIVE" in t:
        d["MFMA util = mfma_busy/(gui_active/8*1024)"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / (t["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        d["kernel cycles = gui_active / 8"] = t["GRBM_GUI_ACTIVE"] / 8.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "SQ_BUSY_CU_CYCLES" in t:
        d["mfma_busy / (busy_cu_cycles * 4)"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / (t["SQ_BUSY_CU_CYCLES"] * 4.0)
    if "SQ_WAIT_ANY" in t and "SQ_WAVE_CYCLES" in t:
        d["wait_any / wave_cycles"] = t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]
        d["wait_inst_any / wave_cycles"] = t.get("SQ_WAIT_INST_ANY", 0.0) / t["SQ_WAVE_CYCLES"]
        d["active_inst_any / wave_cycles"] = t.get("SQ_ACTIVE_INST_ANY", 0.0) / t["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in t and "SQ_LDS_IDX_ACTIVE" in t and t["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict / lds_idx_active"] = t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]
    derived[n] = d
keys = sorted({k for d in derived.values() for k in d})
print()
if md:
    print("| derived | " + " | ".join(n.rjust(14) for n in table) + " |")
    print("|" + "---|" * (len(table) + 1))
for k in keys:
    print(("| " if md else "") + sep.join([k.ljust(44)] + [("%14.4g" % d[k]) if k in d else " " * 14 for d in derived.values()]) + (" |" if md else ""))
