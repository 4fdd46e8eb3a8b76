"""The committed evidence hangs together (CPU only): the bench line kept under profiles/ can be recomputed from the tracked
files next to it -- algorithmic bytes / launch time, the VALU-issue bound from the opcode histogram and the measured opcode
costs, the rocprofv3 average of the same kernel -- and no fraction exceeds 1."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TILE, QUERY, UBENCH = "r06_tile", "r06_query", "r04_ubench"   # the round's evidence directories


def _load(*parts):
    return json.load(open(os.path.join(P, *parts)))


def test_tile_kernel_roofline_is_recomputable():
    b = _load(TILE, "bench.json")
    r = b["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    achieved = r["algorithmic_bytes_per_bp"] * r["bp_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(achieved - r["achieved"]) < 1e-6 * achieved
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] >= r["algorithmic_bytes_per_bp"] * r["bp_per_launch"]  # PMC traffic cannot be below the algorithmic bytes
    assert abs(b["value"] - 10.0 / (b["ms_per_step"] * 1e-3) / 1e0) < 1e-6 * b["value"]  # 10 Gbp per step
    v = r["valu_issue"]
    assert 0.0 < v["frac_of_peak"] < 1.0 and 0.0 < v["frac_of_cycle_weighted_bound"] <= 1.0
    # the cycle-weighted bound from tracked files: instructions per launch x mean cycles of the kernel's opcode mix
    t = _load("traffic.json")
    h = _load(*t["isa_histogram"].split("/"))  # (the histogram traffic.json names: r06_tile has none of its own, profiles/README.md says why)
    mean = h["mean_cycles_per_valu_inst"]
    assert abs(mean - h["bound_cycles_per_wave"] / h["valu_insts_per_wave"]) < 1e-3 * mean
    bound_ms = t["valu_wave_insts_per_launch"] * mean / 1024 / 2.4e9 * 1e3
    assert abs(bound_ms - v["cycle_weighted_bound_ms"]) < 0.02 * bound_ms
    assert bound_ms <= r["avg_launch_ms"]  # a kernel cannot beat its own issue bound
    assert t["profile"] == TILE and t["isa_histogram"] in (TILE + "/isa_histogram.json", "r05_tile/isa_histogram.json") and \
        t["valu_cycles"] == UBENCH + "/valu_cycles.json"


def test_valu_busy_is_recomputable_from_the_counters():
    """Round 3: VALU occupancy from counters alone -- issue slots needed by the instructions that went through the first issue
    path (SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2) against the cycles the chip was busy (GRBM_GUI_ACTIVE over 8 shader engines)."""
    b = _load(TILE, "bench.json")
    t = _load("traffic.json")
    pm = _load(TILE, "pmc_summary.json")
    k = [x for x in pm if "level1_tile_kernel<80, 56, false, 256>" in x][0]
    assert abs(pm[k]["SQ_INSTS_VALU"]["mean_per_launch"] - t["valu_wave_insts_per_launch"]) < 1e-6 * t["valu_wave_insts_per_launch"]
    assert abs(pm[k]["SQ_ACTIVE_INST_VALU2"]["mean_per_launch"] - t["valu2_wave_insts_per_launch"]) < 1.0
    c = b["roofline"]["valu_issue"]["busy_by_counters"]
    need = c["issue_slot_cycles"] * (t["valu_wave_insts_per_launch"] - t["valu2_wave_insts_per_launch"]) / 1024
    have = t["gui_active_cycles_per_launch"] / 8
    assert abs(need - c["slots_needed_cycles_per_simd"]) < 1e-3 * need and abs(have - c["elapsed_cycles_per_simd"]) < 1e-3 * have  # (the bench line carries 4.18 rounded)
    assert c["valu_busy"] == min(1.0, round(need / have, 3)) or abs(c["valu_busy"] - min(1.0, need / have)) < 1e-3
    # the slot length is a measurement: the cycles per instruction of the micro-kernels with no second-path instruction
    costs = _load(UBENCH, "valu_cycles.json")
    assert abs(costs["class_full"] - c["issue_slot_cycles"]) < 0.05
    single = [v for k, v in costs["opcodes"].items() if "+" not in k and v > 3.9]
    assert len(single) >= 10 and abs(sum(single) / len(single) - c["issue_slot_cycles"]) < 0.1


def test_rocprof_average_agrees_with_the_bench_line():
    b = _load(TILE, "bench.json")
    rows = list(csv.DictReader(open(os.path.join(P, TILE, "kernel_stats.csv"))))
    tile = [r for r in rows if "level1_tile_kernel<80, 56, false, 256>" in r["Name"]][0]
    avg_ms, min_ms = float(tile["AverageNs"]) / 1e6, float(tile["MinNs"]) / 1e6
    live = b["roofline"]["avg_launch_ms"]
    assert abs(avg_ms - live) < 0.03 * live  # the profiler's average (incl. the first launches at ramping clocks) within 3 %
    assert min_ms <= live * 1.01             # and its fastest launch is not slower than the live average


def test_query_leg_evidence():
    b = _load(TILE, "bench.json")
    q = b["query"]
    r = q["roofline"]
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    algo = 24.0 * q["counts"]["n_hps"] + 0.25 * 10_000 * 10_000 + 17.0 * q["counts"]["n_signatures"]
    assert abs(algo - r["algorithmic_bytes"]) < 1e-6 * algo
    assert q["cpu_baseline"]["content_match"] is True and b["cpu_baseline"]["content_match"] is True
    s = _load(QUERY, "summary.json")
    pm = _load(QUERY, "pmc_summary.json")
    assert pm["per_query_batch"]["hbm_bytes"] > r["algorithmic_bytes"]
    assert s["kernel_ms_total"] * 1e-3 <= q["query_s"] * 1.05  # kernel time fits inside the measured batch time
    # round 3: the per-query kernel behind the shimmer pipeline -- one host wait, at most 20 launches, under a millisecond;
    # round 6: its level-1 form (no list stage of the batch) -- 8 launches, under 0.65 ms
    assert "one wavefront per query" in q["path"] and "ONE" in q["path"] and q["path_id"] == 3
    assert s["launches"] <= 8 and r["launches"] == s["launches"] and q["query_s"] < 0.65e-3
    assert q["stage_ms"]["lookup_ms"] == 0.0 and q["stage_ms"]["chain_ms"] < 0.1  # (nothing waits between the two stages)
