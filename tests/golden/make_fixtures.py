"""Regenerates the DATA fixtures in tests/golden/ from the reference checkout.

Runs only in the build container (needs /root/reference); the GPU box uses the committed
outputs.  Nothing here copies reference source: the outputs are data files the reference's
own tests hold (FASTA / .mdb / .midx / hit list) and the two input DNA strings of the
known-answer test pgr-db/src/lib.rs:342-363 (expected answer there: 2 shimmers each).
"""
import os
import re
import shutil

REF = "/root/reference/pgr-db"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for f in ["test_seqs.fa", "test_seqs_frag.mdb", "test_seqs_frag.midx", "test_rev.fa", "test_hits"]:
        shutil.copyfile(os.path.join(REF, "test/test_data", f), os.path.join(HERE, f))
    src = open(os.path.join(REF, "src/lib.rs")).read()
    body = src[src.index("fn test_shmmr_reduction_boundary_condition"):]
    seqs = re.findall(r'b"([ACGT]+)"', body)[:2]
    with open(os.path.join(HERE, "boundary_condition_seqs.txt"), "w") as out:
        out.write("# inputs of pgr-db/src/lib.rs:342-363 (spec w=24,k=24,r=12,min_span=24,padding=true; expected len 2)\n")
        for s in seqs:
            out.write(s + "\n")


if __name__ == "__main__":
    main()
