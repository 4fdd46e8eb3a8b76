// pgr-mdb counterpart (pgr-bin/src/bin/pgr-mdb.rs:26-111) in C++ above the C ABI of libpgrhip.so:
//   pgr-mdb <filelist> <prefix> [-w 80] [-k 56] [-r 4] [-m 64] [--sketch] [--batch-bp N] [--reference-sid-quirk]
// builds <prefix>.mdb + <prefix>.midx.  The reference iterates an AGC archive; AGC is not available here, so
// <filelist> lists FASTA / FASTQ (.gz) files.  Index-only path (seq_db.rs:541-615): fragment id = pair ordinal in
// the contig; the host owns sequence iteration and the .midx, the GPU computes shimmers and the frag_map.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "fastx.hpp"
#include "pgr_hip.h"

static void die(pgr_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "pgr-mdb: %s failed (%d): %s\n", what, rc, ctx ? pgr_last_error(ctx) : pgr_last_error(nullptr));
    exit(1);
}

int main(int argc, char **argv) {
    pgr_spec spec = {80, 56, 4, 64, 0};
    uint64_t batch_bp = 2000000000ull;
    bool sid_quirk = false;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "pgr-mdb: %s needs a value\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-w") spec.w = (uint32_t)atoi(val("-w"));
        else if (a == "-k") spec.k = (uint32_t)atoi(val("-k"));
        else if (a == "-r") spec.r = (uint32_t)atoi(val("-r"));
        else if (a == "-m" || a == "--min-span") spec.min_span = (uint32_t)atoi(val("-m"));
        else if (a == "--sketch") spec.sketch = 1;
        else if (a == "--batch-bp") batch_bp = strtoull(val("--batch-bp"), nullptr, 10);
        else if (a == "--reference-sid-quirk") sid_quirk = true;  // load_index_from_reader restarts at 0 per input (seq_db.rs:543)
        else pos.push_back(a);
    }
    if (pos.size() != 2) {
        fprintf(stderr, "usage: pgr-mdb <filelist> <prefix> [-w 80 -k 56 -r 4 -m 64 --sketch]\n");
        return 2;
    }
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(0, &ctx);
    if (rc) die(nullptr, "pgr_ctx_create", rc);
    pgr_index *ix = nullptr;
    if ((rc = pgr_index_create(ctx, &spec, &ix))) die(ctx, "pgr_index_create", rc);

    std::ifstream fl(pos[0]);
    if (!fl) {
        fprintf(stderr, "pgr-mdb: can't open %s\n", pos[0].c_str());
        return 1;
    }
    struct Midx {
        uint32_t sid;
        size_t len;
        std::string name, src;
    };
    std::vector<Midx> midx;
    uint32_t sid = 0;
    std::string path;
    while (std::getline(fl, path)) {
        while (!path.empty() && (path.back() == '\r' || path.back() == ' ')) path.pop_back();
        if (path.empty()) continue;
        const std::vector<pgrhost::SeqRec> recs = pgrhost::read_fastx(path);
        if (sid_quirk) sid = 0;
        size_t i = 0;
        while (i < recs.size()) {  // one GPU batch per ~batch_bp (the reference feeds <= 129 contigs, seq_db.rs:549-564)
            size_t j = i;
            uint64_t tot = 0;
            while (j < recs.size() && (j == i || tot + recs[j].seq.size() <= batch_bp)) tot += recs[j++].seq.size();
            std::vector<const uint8_t *> ptrs;
            std::vector<uint64_t> lens;
            std::vector<uint32_t> sids;
            for (size_t q = i; q < j; ++q) {
                ptrs.push_back((const uint8_t *)recs[q].seq.data());
                lens.push_back(recs[q].seq.size());
                sids.push_back(sid + (uint32_t)(q - i));
            }
            if ((rc = pgr_index_add_batch(ctx, ix, (uint32_t)(j - i), ptrs.data(), lens.data(), sids.data())))
                die(ctx, "pgr_index_add_batch", rc);
            for (size_t q = i; q < j; ++q) midx.push_back(Midx{sid++, recs[q].seq.size(), recs[q].name, path});
            i = j;
        }
    }
    if ((rc = pgr_index_finalize(ctx, ix))) die(ctx, "pgr_index_finalize", rc);
    if ((rc = pgr_index_write_mdb(ctx, ix, (pos[1] + ".mdb").c_str()))) die(ctx, "pgr_index_write_mdb", rc);
    {  // seq_db.rs:798-805; written to a temporary name and renamed, every write checked
        const std::string final_path = pos[1] + ".midx", tmp_path = final_path + ".tmp";
        FILE *f = fopen(tmp_path.c_str(), "w");
        bool ok = f != nullptr;
        for (size_t i = 0; ok && i < midx.size(); ++i) {
            const Midx &m = midx[i];
            ok = fprintf(f, "%u\t%zu\t%s\t%s\n", m.sid, m.len, m.name.c_str(), m.src.c_str()) >= 0;
        }
        if (f) ok = (fclose(f) == 0) && ok;
        if (ok) ok = rename(tmp_path.c_str(), final_path.c_str()) == 0;
        if (!ok) {
            (void)remove(tmp_path.c_str());
            fprintf(stderr, "pgr-mdb: can't write %s\n", final_path.c_str());
            return 1;
        }
    }
    fprintf(stderr, "%zu sequences, %llu shimmer pairs, %llu keys -> %s.mdb / .midx\n", midx.size(),
            (unsigned long long)pgr_index_n_records(ix), (unsigned long long)pgr_index_n_keys(ix), pos[1].c_str());
    pgr_index_destroy(ix);
    pgr_ctx_destroy(ctx);
    return 0;
}
