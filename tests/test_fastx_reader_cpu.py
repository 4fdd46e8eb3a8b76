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

CASES = {
    "plain.fa": b">a desc\nACGT\nacgtn\n>b\nTTTT\n",
    "nonl.fa": b">a\nACGT\n>b\nGG",
    "crlf.fa": b">a x y\r\nAC\r\nGT\r\n>b\r\n\r\nTT\r\n",
    "empty_rec.fa": b">a\n>b\nAC\n>c\n",
    "blank_lines.fa": b">a\n\nAC\n\nGT\n>b\nT\n\n",
    "only_header.fa": b">lonely",
    "reads.fq": b"@r1 c\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n",
    "partial.fq": b"@r1\nACGT\n+\nIIII\n@r2\nGG\n",
    "crlf.fq": b"@r1\r\nACGT\r\n+\r\nIIII\r\n",
    "empty.fa": b"",
}


def test_cpp_reader_matches_python_reader(tmp_path):
    from pgrtk_amd.seqindexdb import read_fastx
    src = tmp_path / "main.cpp"
    src.write_text(MAIN)
    exe = tmp_path / "rd"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "pgr-tk_amd", "host"), "-o", str(exe), str(src), "-lz"],
                   check=True)
    for name, data in CASES.items():
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
            got = [(g[0], int(g[1]), g[2] if len(g) > 2 else "") for g in got]
            ref = [(n, len(s), s.decode()) for n, s in read_fastx(str(path))]
            assert got == ref, (name, gz, got, ref)
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"hello\n")
    assert subprocess.run([str(exe), str(bad)], capture_output=True).returncode == 3
