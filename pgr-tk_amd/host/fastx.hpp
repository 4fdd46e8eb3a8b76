// fastx.hpp -- FASTA / FASTQ (.gz ok, through zlib) with the record semantics of the reference reader
// (pgr-db/src/fasta_io.rs: id = header up to the first space :94-101, sequence bytes kept as they are,
// line ends dropped :102-106).  Host-side sequence iteration stays with the caller of the C ABI.
#pragma once
#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pgrhost {

struct SeqRec {
    std::string name;
    std::string seq;
};

inline std::string slurp(const std::string &path) {
    gzFile f = gzopen(path.c_str(), "rb");  // transparent for plain files
    if (!f) throw std::runtime_error("can't open " + path);
    gzbuffer(f, 1 << 20);
    std::string data;
    std::vector<char> buf(1 << 22);
    for (;;) {
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) {
            gzclose(f);
            throw std::runtime_error("read error on " + path);
        }
        if (n == 0) break;
        data.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return data;
}

inline std::string first_token(const char *b, const char *e) {
    while (e > b && (e[-1] == '\r')) --e;
    const char *p = b;
    while (p < e && *p != ' ') ++p;
    return std::string(b, p);
}

inline std::vector<SeqRec> read_fastx(const std::string &path) {
    const std::string data = slurp(path);
    std::vector<SeqRec> out;
    if (data.empty()) return out;
    const char *p = data.data(), *end = p + data.size();
    if (*p == '>') {
        while (p < end) {
            // p at '>' of a record
            const char *h = p + 1;
            const char *nl = (const char *)memchr(h, '\n', (size_t)(end - h));
            SeqRec r;
            r.name = first_token(h, nl ? nl : end);
            const char *q = nl ? nl + 1 : end;
            // body: every line up to the next one that starts with '>'
            while (q < end && *q != '>') {
                const char *ln = (const char *)memchr(q, '\n', (size_t)(end - q));
                const char *le = ln ? ln : end;
                const char *ce = le;
                while (ce > q && ce[-1] == '\r') --ce;
                r.seq.append(q, ce);
                q = ln ? ln + 1 : end;
            }
            out.push_back(std::move(r));
            p = q;
        }
    } else if (*p == '@') {
        std::vector<std::pair<const char *, const char *>> lines;
        while (p < end) {
            const char *ln = (const char *)memchr(p, '\n', (size_t)(end - p));
            lines.emplace_back(p, ln ? ln : end);
            p = ln ? ln + 1 : end;
        }
        // four lines per record; a trailing partial record is dropped
        const size_t n_lines = lines.size() + (data.back() == '\n' ? 1 : 0);  // as if split on '\n'
        for (size_t i = 0; i + 3 < n_lines; i += 4) {
            SeqRec r;
            r.name = first_token(lines[i].first + 1, lines[i].second);
            const char *b = lines[i + 1].first, *e = lines[i + 1].second;
            while (e > b && e[-1] == '\r') --e;
            r.seq.assign(b, e);
            out.push_back(std::move(r));
        }
    } else {
        throw std::runtime_error("not a FASTA/FASTQ file: " + path);
    }
    return out;
}

}  // namespace pgrhost
