#!/usr/bin/env python3
"""Per-opcode VALU issue cost on MI355X from hardware counters -> profiles/<name>/valu_cycles.json (+ .txt).

    tools/ubench_table.py gpurun_out/<dir> profiles/<name>

<dir> holds two rocprofv3 runs of tools/ubench_valu (the binary itself prints a third, clock-dependent estimate):
    ub_pmc/   --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES   (counters only)
    ub_trace/ --kernel-trace --stats                                               (durations only)
Every micro-kernel keeps 8 waves per SIMD busy with 8 independent dependency chains per wave, so the SIMD's VALU
issue port is the only limit.  cycles per wave64 instruction per SIMD =
    (GRBM_GUI_ACTIVE / 8 XCDs) / (SQ_INSTS_VALU / 1024 SIMDs)
both read from the same dispatch: no clock frequency enters.  The effective clock (GRBM_GUI_ACTIVE / 8 / duration) is
listed beside it.  Sequences of more than one instruction are reported per instruction of the sequence.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
pmc = collections.defaultdict(dict)
order = []
for r in csv.DictReader(open(glob.glob(os.path.join(src, "ub_pmc", "*counter_collection.csv"))[0])):
    k = (r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"]))
    if k not in pmc:
        order.append(k)
    pmc[k][r["Counter_Name"]] = float(r["Counter_Value"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(os.path.join(src, "ub_trace", "*kernel_trace.csv"))[0])):
    dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
last = {}
for k in order:  # the last dispatch of every kernel (the first one is the warm-up launch)
    last[k[0]] = pmc[k]
# opcode(s) every micro-kernel issues per sequence: parsed from the source so that the table cannot drift from it
names = {}
for line in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_valu.hip")):
    line = line.strip()
    if line.startswith("KERNEL(") and "REP8_" in line:
        kn = line[len("KERNEL("):].split(",")[0].strip()
        asm = line.split('REP8_', 1)[1].split('("', 1)[1].split('")', 1)[0]  # (REP8_32 / _64 / _32C / _64C)
        names[kn] = [p.strip().split()[0] for p in asm.split("\\n")]
    elif line.startswith("KERNEL(k_mad_u64_u32_asm"):
        names["k_mad_u64_u32_asm"] = ["v_mad_u64_u32"]
rows = []
table = {}
variants = {}
for kn, c in last.items():
    if not kn.startswith("k_") or kn == "k_clock" or c.get("SQ_INSTS_VALU", 0) <= 0:
        continue
    g = c["GRBM_GUI_ACTIVE"] / 8.0
    n = c["SQ_INSTS_VALU"] / 1024.0
    ms = dur[kn][-1] if dur[kn] else float("nan")
    cyc = g / n
    ops = names.get(kn)
    rows.append((kn, ops, ms, g, n, cyc, g / ms / 1e3))
    if kn == "k_cndmask":  # reads a vcc nobody writes: measures a dependency stall, not the issue rate -> not a cost
        continue
    variant = {"k_lshlrev_b32_vgpr": "v_lshlrev_b32 (shift in a VGPR)", "k_lshrrev_b32_vgpr": "v_lshrrev_b32 (shift in a VGPR)",
               "k_lshlrev_b32_e64": "v_lshlrev_b32 (VOP3 encoding)", "k_lshrrev_b32_e64": "v_lshrrev_b32 (VOP3 encoding)",
               "k_lshlrev_b32_by1": "v_lshlrev_b32 (by 1)", "k_ashrrev_i32_vgpr": "v_ashrrev_i32 (shift in a VGPR)",
               "k_fma_f32_3src": "v_fma_f32 (three distinct VGPR sources)",
               **{"k_%s_3src" % n: "v_%s (three distinct VGPR sources)" % n for n in
                  ("bfi_b32", "and_or_b32", "or3_b32", "add3_u32", "xad_u32", "bitop3_b32", "min3_u32", "mad_u32_u24", "alignbit",
                   "lshl_add_u32", "xor3_b32", "perm_b32")},
               "k_mad_u64_u32": None}.get(kn, "")
    if variant is None:
        continue  # (the C++ expression form of round 2; k_mad_u64_u32_asm is the instruction itself)
    # Round 4: an opcode whose cost depends on the operand pattern.  v_fma_f32 and v_bitop3_b32 issue at the fast rate with
    # three distinct sources and 1.3 cycles slower when ONE VGPR feeds two source operands (how round 3 had measured them);
    # every other three-source integer opcode costs the full slot either way.  The kernels use distinct sources: that is
    # the cost of the opcode, the other form is kept as a variant.
    if kn in ("k_fma_f32_3src", "k_bitop3_b32_3src"):
        table[ops[0]] = round(cyc, 3)
        continue
    if kn in ("k_fma_f32", "k_bitop3_b32"):
        variants[ops[0] + " (one VGPR feeding two source operands)"] = round(cyc, 3)
        continue
    if variant:
        variants[variant] = round(cyc, 3)
        continue
    if ops and len(set(ops)) == 1:
        table[ops[0].replace("_e64", "")] = round(cyc, 3)
    elif ops:
        table["+".join(ops)] = round(cyc, 3)
with open(os.path.join(dst, "valu_cycles.txt"), "w") as f:
    f.write("MI355X (gfx950) VALU issue cost, cycles per wave64 instruction per SIMD, from PMC counters\n")
    f.write("(GRBM_GUI_ACTIVE / 8) / (SQ_INSTS_VALU / 1024) of the same dispatch; 8 waves/SIMD x 8 independent chains\n\n")
    f.write("%-22s %-44s %8s %12s %12s %9s %8s\n" % ("micro-kernel", "instruction(s) per sequence", "ms", "cycles/SIMD", "VALU/SIMD",
                                                    "cyc/inst", "eff MHz"))
    for kn, ops, ms, g, n, cyc, mhz in rows:
        f.write("%-22s %-44s %8.3f %12.0f %12.0f %9.3f %8.0f\n" % (kn, " ; ".join(ops) if ops else "(C++ expression)", ms, g, n, cyc, mhz))
# cost classes used by tools/isa_histogram.py for opcodes that have no micro-kernel of their own
simple = [v for k, v in table.items() if k in ("v_add_u32", "v_sub_u32", "v_xor_b32", "v_or_b32", "v_and_b32", "v_not_b32",
                                                "v_lshrrev_b32", "v_mov_b32")]
full = [v for k, v in table.items() if k in ("v_lshlrev_b32", "v_bfi_b32", "v_bfe_i32", "v_and_or_b32", "v_alignbit_b32",
                                              "v_min_u32", "v_lshl_add_u32", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64",
                                              "v_min_f64", "v_max_f64", "v_bfrev_b32", "v_add3_u32", "v_or3_b32")]
out = {"source": os.path.basename(os.path.normpath(src)), "unit": "cycles per wave64 VALU instruction per SIMD",
       "method": "(GRBM_GUI_ACTIVE/8) / (SQ_INSTS_VALU/1024), rocprofv3 --pmc, tools/ubench_valu (8 waves/SIMD, 8 chains/wave)",
       "guide_peak": {"cycles": 2.0, "wave64_Ginst_per_s": 1228.8,
                      "note": "MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues over 2 cycles; 1024 SIMDs x 2.4 GHz / 2"},
       "class_simple": round(sum(simple) / len(simple), 3) if simple else None,
       "class_full": round(sum(full) / len(full), 3) if full else None,
       "opcodes": table, "encoding_variants": variants}
json.dump(out, open(os.path.join(dst, "valu_cycles.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "ubench_valu.txt")):
    shutil.copyfile(os.path.join(src, "ubench_valu.txt"), os.path.join(dst, "ubench_valu.txt"))
print(open(os.path.join(dst, "valu_cycles.txt")).read())
print(json.dumps({k: out[k] for k in ("class_simple", "class_full")}))
