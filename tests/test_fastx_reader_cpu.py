"""CPU: the C++ FASTA / FASTQ reader of the host programs (pgr-tk_amd/host/fastx.hpp) against the Python reader of
the package on tricky inputs (both restate the record semantics of pgr-db/src/fasta_io.rs:79-172)."""
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))

MAIN = r'''
#include <cstdio>
#include "fastx.hpp"
int main(int argc, char **argv) {
    try {
        for (const auto &r : pgrhost::read_fastx(argv[1])) printf("%s\t%zu\t%s\n", r.name.c_str(), r.seq.size(), r.seq.c_str());
    } catch (const std::exception &e) { printf("ERROR\n"); return 3; }
    return 0;
}
'''

# input -> records, derived by hand from pgr-db/src/fasta_io.rs:46-165 (read_until semantics, quirks included)
CASES = {
    "plain.fa": (b">a desc\nACGT\nacgtn\n>b\nTTTT\n", [("a", "ACGTacgtn"), ("b", "TTTT")]),
    "nonl.fa": (b">a\nACGT\n>b\nGG", [("a", "ACGT"), ("b", "GG")]),
    "crlf.fa": (b">a x y\r\nAC\r\nGT\r\n>b\r\n\r\nTT\r\n", [("a", "ACGT"), ("b", "TT")]),
    "empty_rec.fa": (b">a\n>b\nAC\n>c\n", [("a", ""), ("b", "AC"), ("c", "")]),
    "blank_lines.fa": (b">a\n\nAC\n\nGT\n>b\nT\n\n", [("a", "ACGT"), ("b", "T")]),
    "only_header.fa": (b">lonely", [("lonely", "")]),
    # '>' ends a sequence ANYWHERE (:102), not only at a line start; the header is the whole line (:90)
    "gt_inside.fa": (b">a b>c\nAC>GT\n>d\nTT\n", [("a", "AC"), ("GT", ""), ("d", "TT")]),
    "cr_inside.fa": (b">a\nAC\rGT\n", [("a", "ACGT")]),
    # the record read last is dropped when the file ends right after its quality line (:159-162)
    "reads.fq": (b"@r1 c\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n", [("r1", "ACGT")]),
    "reads_trailing_blank.fq": (b"@r1 c\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n\n", [("r1", "ACGT"), ("r2", "GG")]),
    "partial.fq": (b"@r1\nACGT\n+\nIIII\n@r2\nGG\n", [("r1", "ACGT")]),
    "crlf.fq": (b"@r1\r\nACGT\r\n+\r\nIIII\r\n", []),
    # neither '>' nor '@' first: FASTA, and the first byte is consumed all the same (:64-68)
    "other.txt": (b"hello\nAC\n", [("ello", "AC")]),
}


def test_readers_follow_the_reference_reader(tmp_path):
    from pgrtk_amd.seqindexdb import read_fastx
    src = tmp_path / "main.cpp"
    src.write_text(MAIN)
    exe = tmp_path / "rd"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "pgr-tk_amd", "host"), "-o", str(exe), str(src), "-lz"],
                   check=True)
    for name, (data, want) in CASES.items():
        for gz in (False, True):
            path = tmp_path / (name + (".gz" if gz else ""))
            if gz:
                with gzip.open(path, "wb") as f:
                    f.write(data)
            else:
                path.write_bytes(data)
            r = subprocess.run([str(exe), str(path)], capture_output=True, text=True)
            assert r.returncode == 0, (name, gz)
            got = [tuple(l.split("\t")) for l in r.stdout.split("\n") if l]
            got = [(g[0], g[2] if len(g) > 2 else "") for g in got]
            py = [(n, s.decode()) for n, s in read_fastx(str(path))]
            assert got == want, (name, gz, got, want)
            assert py == want, (name, gz, py, want)
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")  # fasta_io.rs:58-63: an error, not an empty list
    assert subprocess.run([str(exe), str(empty)], capture_output=True).returncode == 3
    import pytest
    with pytest.raises(ValueError):
        read_fastx(str(empty))
