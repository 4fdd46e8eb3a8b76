#!/usr/bin/env python3
"""Generates tests/golden/query_sdb_cases.json from the reference's own Python helpers (pgr-tk/pgrtk/__init__.py:
merge_regions :270-328, query_sdb :130-221), run HERE in the build container (the reference is read-only input of this
script; only the resulting input/output vectors are committed).  The module itself cannot be imported (it needs the
compiled Rust extension), so the two pure-Python functions are pulled out of its syntax tree and executed alone.

    python tests/golden/make_query_sdb_fixture.py        # needs /root/reference
"""
import ast
import copy
import json
import os
import random

SRC = "/root/reference/pgr-tk/pgrtk/__init__.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_functions(names):
    tree = ast.parse(open(SRC).read())
    ns = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), SRC, "exec"), ns)
    return ns


class CannedDB:
    """stands in for SeqIndexDB: query_fragment_to_hps returns prepared chains"""

    def __init__(self, r):
        self.r = r

    def query_fragment_to_hps(self, *args):
        return self.r


def random_chains(rng, n_targets):
    out = []
    for sid in rng.sample(range(50), n_targets):
        alns = []
        t0 = rng.randrange(0, 500000)
        for _ in range(rng.randrange(1, 5)):
            n = rng.choice([1, 2, 3, 4, 8, 20])
            qo = rng.randrange(2)
            to = qo if rng.random() < 0.7 else 1 - qo
            q, t = rng.randrange(0, 2000), t0 + rng.randrange(-2000, 30000)
            aln = []
            for _ in range(n):
                ln = rng.randrange(60, 400)
                aln.append(((q, q + ln, qo), (max(t, 0), max(t, 0) + ln, to)))
                q += rng.randrange(50, 900)
                t += rng.randrange(50, 900) if qo == to else -rng.randrange(50, 900)
            alns.append((round(rng.uniform(1, 5000), 3), aln))
        out.append((sid, alns))
    return out


def main():
    ns = load_functions({"merge_regions", "query_sdb", "group_smps_by_principle_bundle_id", "rc", "rc_byte_seq"})
    ns["rc_map"] = dict(zip("ACGTNnactg", "TGCANntgca"))           # module-level tables of the reference (:36-37, :75)
    ns["byte_rc_map"] = dict(zip([ord(c) for c in "ACGTNnacgt"], [ord(c) for c in "TGCANntgca"]))
    rng = random.Random(20240917)
    merge_cases = []
    for _ in range(60):
        rgns = []
        for _ in range(rng.randrange(0, 12)):
            b = rng.randrange(0, 50000)
            e = b + rng.randrange(1, 8000)
            rgns.append((b, e, e - b, rng.randrange(2), [rng.randrange(100) for _ in range(rng.randrange(0, 3))]))
        tol = rng.choice([0, 1, 12, 1000, 100000])
        got = ns["merge_regions"]([tuple(r) for r in copy.deepcopy(rgns)], tol=tol)  # it extends the record lists in place
        merge_cases.append({"rgns": rgns, "tol": tol, "out": got})
    sdb_cases = []
    for _ in range(40):
        r = random_chains(rng, rng.randrange(0, 6))
        tol = rng.choice([0, 12, 1000, 100000])
        got = ns["query_sdb"](CannedDB(copy.deepcopy(r)), b"ACGT", merge_range_tol=tol)
        sdb_cases.append({"r": r, "tol": tol, "out": [[sid, v] for sid, v in got.items()]})
    group_cases = []
    for _ in range(60):
        smps, pos = [], rng.randrange(0, 1000)
        bid, d_run = rng.randrange(6), rng.randrange(2)
        for _ in range(rng.randrange(0, 60)):
            if rng.random() < 0.25:
                bid, d_run = rng.randrange(6), rng.randrange(2)
            ln = rng.randrange(30, 900)
            o = rng.randrange(2)
            smp = (rng.getrandbits(50), rng.getrandbits(50), pos, pos + ln, o)
            info = None if rng.random() < 0.15 else (bid, o if d_run == 0 else 1 - o, rng.randrange(500))
            smps.append((smp, info))
            pos += ln + (rng.randrange(0, 12000) if rng.random() < 0.1 else 0)
        cutoff, merge = rng.choice([0, 50, 500, 2500]), rng.choice([0, 100, 5000, 100000])
        got = ns["group_smps_by_principle_bundle_id"](copy.deepcopy(smps), cutoff, merge)
        group_cases.append({"smps": smps, "len_cutoff": cutoff, "merge_length": merge, "out": got})
    rc_cases = []
    for _ in range(10):
        sq = "".join(rng.choice("ACGTNnacgt") for _ in range(rng.randrange(0, 60)))
        rc_cases.append({"seq": sq, "rc": ns["rc"](sq), "rc_bytes": ns["rc_byte_seq"](list(sq.encode()))})
    with open(os.path.join(HERE, "query_sdb_cases.json"), "w") as f:
        json.dump({"merge_regions": merge_cases, "query_sdb": sdb_cases, "group_smps": group_cases, "rc": rc_cases}, f)
    print(len(merge_cases), "merge_regions cases,", len(sdb_cases), "query_sdb cases,", len(group_cases), "group_smps cases")


if __name__ == "__main__":
    main()
