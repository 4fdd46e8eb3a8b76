"""csrc/arena_list.h (the free list under pgr_ctx_reserve) against a byte map: tests/arena_list_harness.cpp"""
import os
import subprocess


def test_arena_list_random_take_and_give_back(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pgr-tk_amd", "csrc")
    exe = str(tmp_path / "arena_list_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, os.path.join(root, "tests", "arena_list_harness.cpp"), "-o", exe],
                   check=True, timeout=300)
    r = subprocess.run([exe, "6000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-500:] + r.stderr[-500:]
