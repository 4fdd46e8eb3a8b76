"""csrc/island_list.h (tile flags -> islands of the exact machine, from the list of flagged tiles) against the tile-by-tile reading,
through a C++ harness built here with g++ (no GPU, no HIP)"""
import os
import subprocess


def test_sparse_island_listing_equals_the_tile_by_tile_reading(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "pgr-tk_amd", "csrc")
    exe = str(tmp_path / "island_list_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, os.path.join(root, "tests", "island_list_harness.cpp"), "-o", exe],
                   check=True, timeout=300)
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "20000 cases, 0 failures" in r.stdout
