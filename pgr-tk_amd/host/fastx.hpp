// fastx.hpp -- FASTA / FASTQ (.gz ok, through zlib) with the record semantics of the reference reader
// (pgr-db/src/fasta_io.rs:46-165), INCLUDING its quirks, because sids, .midx lines and query indices follow from them:
//   * the first byte of the file decides the format ('@' = FASTQ, anything else = FASTA) and is consumed (:54-68);
//   * FASTA (:86-119): header = the rest of the line, id = header up to the first ' ' minus '\n' ' ' '\r'; the sequence is
//     every byte up to the next '>' ANYWHERE (not only at a line start), minus '\n' '>' '\r' (so '\r' is dropped
//     everywhere, and a '>' inside a sequence line starts a new record whose header is the rest of that line);
//   * FASTQ (:121-164): id line, sequence line, skip through the next '+', the rest of that line, the quality line,
//     then everything through the next '@'; when that last step reads nothing (end of file right after the quality
//     line) the record is DROPPED -- the reference loses the final record of a FASTQ file that ends after its last
//     quality line; a trailing blank line keeps it.
// Host-side sequence iteration stays with the caller of the C ABI.
#pragma once
#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
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

// BufRead::read_until on an in-memory file: bytes [pos, first delim] (delimiter included), pos moves past them
inline std::pair<const char *, const char *> read_until(const std::string &d, size_t &pos, char delim) {
    const char *b = d.data() + pos, *end = d.data() + d.size();
    const char *q = (pos < d.size()) ? (const char *)memchr(b, delim, (size_t)(end - b)) : nullptr;
    const char *e = q ? q + 1 : end;
    pos = (size_t)(e - d.data());
    return {b, e};
}

inline std::string record_id(const char *b, const char *e) {  // fasta_io.rs:94-101
    std::string id;
    for (const char *p = b; p < e; ++p) {
        if (*p == ' ') break;
        if (*p != '\n' && *p != '\r') id.push_back(*p);
    }
    return id;
}

inline std::vector<SeqRec> read_fastx(const std::string &path) {
    const std::string data = slurp(path);
    std::vector<SeqRec> out;
    if (data.empty()) throw std::runtime_error("empty file: " + path);  // fasta_io.rs:58-63
    const bool fastq = data[0] == '@';
    size_t pos = 1;  // the format byte is consumed
    if (!fastq) {
        for (;;) {
            const auto h = read_until(data, pos, '\n');
            if (h.first == h.second) break;  // read_until returned 0: end of file
            SeqRec r;
            r.name = record_id(h.first, h.second);
            const auto b = read_until(data, pos, '>');
            r.seq.reserve((size_t)(b.second - b.first));
            for (const char *p = b.first; p < b.second; ++p)
                if (*p != '\n' && *p != '>' && *p != '\r') r.seq.push_back(*p);
            out.push_back(std::move(r));
        }
    } else {
        for (;;) {
            const auto h = read_until(data, pos, '\n');
            SeqRec r;
            r.name = record_id(h.first, h.second);
            const auto b = read_until(data, pos, '\n');
            for (const char *p = b.first; p < b.second; ++p)
                if (*p != '\n' && *p != '\r') r.seq.push_back(*p);
            (void)read_until(data, pos, '+');
            (void)read_until(data, pos, '\n');
            (void)read_until(data, pos, '\n');
            const auto t = read_until(data, pos, '@');
            if (t.first == t.second) break;  // fasta_io.rs:159-162: the record read last is dropped
            out.push_back(std::move(r));
        }
    }
    return out;
}

// Path::with_extension of the output prefix (pgr-query.rs:291-302): an extension the prefix's file name already carries
// is REPLACED ("out.v1" -> "out.000.hit"); a leading dot alone is not an extension (".out" -> ".out.000.hit")
inline std::string with_extension(const std::string &prefix, const std::string &ext) {
    const size_t sl = prefix.find_last_of('/');
    const size_t name0 = sl == std::string::npos ? 0 : sl + 1;
    const std::string name = prefix.substr(name0);
    if (name.empty() || name == "..") return prefix;  // no file name: set_extension leaves the path alone
    const size_t dot = name.find_last_of('.');
    const std::string base = (dot == std::string::npos || dot == 0) ? prefix : prefix.substr(0, name0 + dot);
    return base + "." + ext;
}

}  // namespace pgrhost
