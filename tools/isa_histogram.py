#!/usr/bin/env python3
"""Opcode histogram of a kernel in libpgrhip.so, weighted by measured issue cost -> the VALU-issue lower bound.

    tools/isa_histogram.py [--so pgr-tk_amd/lib/libpgrhip.so] [--kernel level1_tile_kernelILi80ELi56ELb0ELi256E]
                           [--costs profiles/r03_ubench/valu_cycles.json] [--out profiles/<name>/isa_histogram.json]

What it does (works without a GPU: the code object is in the shared library):
  1. llvm-objdump --offloading extracts the gfx950 code objects of the library, llvm-objdump -d disassembles them;
  2. the kernel's instructions are split into basic blocks at branch instructions and branch targets;
  3. every block gets an execution weight per wavefront:
       * blocks of the boundary variant of the tile kernel (they hold the v_bfe_i32 mask expansions that only
         `tile_select<..., MASKED = true>` contains) weigh --masked-weight (default 0: a 10 Mbp contig has 2 boundary
         tiles in 1245);
       * blocks the source marks cold with `s_nop 15` weigh 0, and so does everything behind an `s_nop 14` (the contig's tail at the
         end of the tile kernel, run by one tile per contig);
       * the body of a backward branch (the output loop `while (em)`) weighs --loop-trips (default 2.45 = expected
         maximum over 64 lanes of the number of level-1 minimizers among a lane's 16 positions at density 2/(w+1));
       * everything else weighs 1;
  4. VALU opcodes are priced with the measured cycles per wave64 instruction per SIMD (tools/ubench_table.py);
     opcodes without a micro-benchmark take the class cost (2-cycle "simple" ops: add/sub/and/or/xor/not/lshr/mov;
     everything else full cost).
Output: per-opcode weighted counts and cycles per wavefront, the totals per position, and
    bound_cycles_per_wave = sum(count x cost)   -> the time the SIMDs need just to ISSUE the kernel's VALU work.
The static count is cross-checked against the hardware's SQ_INSTS_VALU per wave when --measured-valu-per-wave is given.
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SIMPLE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_or_b32", "v_and_b32", "v_not_b32", "v_lshrrev_b32",
          "v_mov_b32", "v_nop"}


def disassemble(so):
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copyfile(so, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        text = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                text.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                           capture_output=True, text=True).stdout)
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_lines(text, name):
    out, on = [], False
    for line in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
        if m:
            if on:
                break
            on = name in m.group(2) and not m.group(2).endswith(".kd")
            base = int(m.group(1), 16)
            continue
        if on and line.strip():
            out.append(line)
    if not out:
        raise SystemExit("kernel %r not found" % name)
    return base, out


def parse(base, lines):
    """-> [(address, opcode, operands)] ; addresses from the trailing comment"""
    ins = []
    for line in lines:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def basic_blocks(ins):
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    targets = {}
    for i, (a, op, operands) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(operands.split()[0])
            if off >= 32768:
                off -= 65536
            t = a + 4 + 4 * off
            targets[i] = t
            if t in addr_idx:
                leaders.add(addr_idx[t])
            if i + 1 < len(ins):
                leaders.add(i + 1)
        if op == "s_endpgm" and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    blocks = [(s, (starts[k + 1] if k + 1 < len(starts) else len(ins))) for k, s in enumerate(starts)]
    return blocks, targets, addr_idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "pgr-tk_amd", "lib", "libpgrhip.so"))
    ap.add_argument("--kernel", default="level1_tile_kernelILi80ELi56ELb0ELi256E")
    ap.add_argument("--costs", default=os.path.join(ROOT, "profiles", "r03_ubench", "valu_cycles.json"))
    ap.add_argument("--masked-weight", type=float, default=0.0)
    ap.add_argument("--loop-trips", type=float, default=2.45)
    ap.add_argument("--positions-per-wave", type=int, default=1024)
    ap.add_argument("--core-fraction", type=float, default=(4096 - 2 * 79) // 64 * 64 / 4096.0,
                    help="core positions / extended positions of a tile (256 lanes x 16, w=80: 3904/4096)")
    ap.add_argument("--measured-valu-per-wave", type=float, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    costs = json.load(open(a.costs))
    c_simple, c_full, per_op = costs["class_simple"], costs["class_full"], costs["opcodes"]
    base, lines = kernel_lines(disassemble(a.so), a.kernel)
    ins = parse(base, lines)
    blocks, targets, addr_idx = basic_blocks(ins)
    # loop bodies: a backward branch at instruction i to target t -> blocks in [t, i] repeat
    loops = []
    def is_exit(j):  # straight-line code that ends the program: a shared exit block the compiler placed in front, not a loop head
        while j < len(ins):
            if ins[j][1] == "s_endpgm":
                return True
            if ins[j][1].startswith(("s_cbranch", "s_branch")):
                return False
            j += 1
        return False
    for i, t in targets.items():
        if t in addr_idx and addr_idx[t] <= i and not is_exit(addr_idx[t]):
            loops.append((addr_idx[t], i))
    hist = collections.Counter()
    n_blocks_masked = 0
    # `s_nop 14`: the source marks the start of a cold REGION that runs to the end of the kernel (the contig's tail, run by the last
    # tile of a contig only: 1 tile in 2562 of a 10 Mbp contig)
    cold_from = min([i for i in range(len(ins)) if ins[i][1] == "s_nop" and ins[i][2].strip() == "14"], default=len(ins))
    for s, e in blocks:
        ops = [ins[i][1] for i in range(s, e)]
        w = 1.0
        if any(o.startswith("v_bfe_i32") for o in ops):
            w = a.masked_weight
            n_blocks_masked += 1
        if any(ins[i][1] == "s_nop" and ins[i][2].strip() == "15" for i in range(s, e)):
            w = 0.0  # marked cold in the source (the exact palindrome test behind a wave-uniform, practically never taken branch)
        if s >= cold_from:
            w = 0.0
        for ls, le in loops:
            if s >= ls and e - 1 <= le:
                w *= a.loop_trips
        for o in ops:
            hist[o.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")] += w
    valu = {o: c for o, c in hist.items() if o.startswith("v_") and c > 0}
    rows = []
    tot_n = tot_c = 0.0
    for o, n in sorted(valu.items(), key=lambda kv: -kv[1]):
        cost = per_op.get(o)
        src = "measured"
        if cost is None:
            cost, src = (c_simple, "class simple") if o in SIMPLE else (c_full, "class full")
        rows.append({"opcode": o, "per_wave": round(n, 2), "cycles_each": cost, "cost_source": src, "cycles_per_wave": round(n * cost, 1)})
        tot_n += n
        tot_c += n * cost
    other = {k: round(sum(c for o, c in hist.items() if o.startswith(k)), 1) for k in ("s_", "ds_", "global_", "buffer_", "flat_")}
    core_pos = a.positions_per_wave * a.core_fraction
    out = {
        "kernel": a.kernel, "library": os.path.relpath(a.so, ROOT), "cost_table": os.path.relpath(a.costs, ROOT),
        "weights": {"masked_variant_blocks": n_blocks_masked, "masked_weight": a.masked_weight, "loop_trips": a.loop_trips,
                    "loops_found": len(loops)},
        "valu_insts_per_wave": round(tot_n, 1), "valu_insts_per_core_position": round(tot_n * 64 / core_pos, 2),
        "bound_cycles_per_wave": round(tot_c, 1), "bound_cycles_per_position_row": round(tot_c / (a.positions_per_wave / 64), 2),
        "mean_cycles_per_valu_inst": round(tot_c / tot_n, 3),
        "guide_2cycle_floor_cycles_per_wave": round(tot_n * 2.0, 1),
        "other_insts_per_wave": other, "opcodes": rows,
    }
    if a.measured_valu_per_wave:
        # the static walk counts lane-predicated prologue / epilogue blocks for every wave; the hardware count is exact:
        # price the hardware's instruction count with the static instruction mix
        out["measured_valu_per_wave"] = a.measured_valu_per_wave
        out["static_over_measured"] = round(tot_n / a.measured_valu_per_wave, 4)
        out["bound_cycles_per_wave_at_measured_count"] = round(tot_c / tot_n * a.measured_valu_per_wave, 1)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    print("%-22s %10s %8s %12s  %s" % ("opcode", "per wave", "cycles", "cyc/wave", "cost source"))
    for r in rows:
        print("%-22s %10.1f %8.3f %12.1f  %s" % (r["opcode"], r["per_wave"], r["cycles_each"], r["cycles_per_wave"], r["cost_source"]))
    print("VALU instructions per wave %.1f (per core position %.2f)%s" % (tot_n, out["valu_insts_per_core_position"],
          "" if not a.measured_valu_per_wave else "; hardware SQ_INSTS_VALU per wave %.1f -> static/measured %.3f" %
          (a.measured_valu_per_wave, out["static_over_measured"])))
    print("cycle-weighted VALU issue bound: %.0f cycles per wave = %.1f cycles per position row (16 rows of 64 positions per wave); "
          "2-cycle floor %.0f" % (tot_c, out["bound_cycles_per_position_row"], tot_n * 2.0))
    print("other per wave:", other)


if __name__ == "__main__":
    main()
