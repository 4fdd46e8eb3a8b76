#!/usr/bin/env python3
"""Hunt for the round-5 driver-box hang of `pgr-mdb --ranks 1 --devices 0 --force-exchange`.

    python tools/repro_ranks_hang.py [iterations=30] [outdir=gpurun_out/hang]

Runs the command again and again on this box with the library's watchdogs at 20/30 s, NCCL_DEBUG=INFO and a 60 s limit per run;
a run that does not finish has the stacks of all its processes dumped (tests/procutil.py) before its group is killed.  Every
10 iterations the two tests that precede it in the driver's order (RCCL in-process with one rank, two ranks on one device) are
run in between, because what they leave behind is one of the suspects.  One JSON summary at the end.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import procutil  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "hang")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(ROOT, "pgr-tk_amd", "bin", "pgr-mdb")
    fl = os.path.join(out, "files.txt")
    with open(fl, "w") as f:
        f.write(os.path.join(ROOT, "tests", "golden", "test_seqs.fa") + "\n")
    env = {"NCCL_DEBUG": os.environ.get("NCCL_DEBUG", "INFO"), "PGR_DEBUG": "1"}
    rows = []
    variants = [["--ranks", "1", "--devices", "0", "--force-exchange", "--batch-bp", "60000"],
                ["--ranks", "1", "--force-exchange", "--prepack", "--batch-bp", "100000"]]
    for i in range(n):
        if i % 10 == 0:
            t0 = time.time()
            r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                                "-k", "exchange_abi_one_rank or two_ranks_sharded_build"], capture_output=True, text=True, timeout=900)
            rows.append({"iter": i, "what": "predecessor tests", "rc": r.returncode, "s": round(time.time() - t0, 2),
                         "tail": r.stdout[-400:]})
            print(rows[-1], flush=True)
        v = variants[i % len(variants)]
        t0 = time.time()
        row = {"iter": i, "what": " ".join(v)}
        try:
            r = procutil.run_bounded([exe, fl, os.path.join(out, "o%d" % i)] + v, timeout=60, env=env)
            row.update(rc=r.returncode, s=round(time.time() - t0, 2))
            if r.returncode != 0:
                row["stderr_tail"] = r.stderr[-3000:]
                with open(os.path.join(out, "fail_%d.txt" % i), "w") as f:
                    f.write(r.stderr)
        except AssertionError as e:
            row.update(rc="timeout", s=round(time.time() - t0, 2))
            with open(os.path.join(out, "hang_%d.txt" % i), "w") as f:
                f.write(str(e))
        rows.append(row)
        print(row, flush=True)
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(rows, f, indent=1)
    bad = [r for r in rows if r.get("rc") not in (0,)]
    print("runs: %d, not ok: %d" % (len(rows), len(bad)))


if __name__ == "__main__":
    main()
