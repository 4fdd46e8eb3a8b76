// index.hip -- ShmmrToFrags index + query entry points (under construction in this commit).
#include "pgr_ctx.h"

struct pgr_index {
    pgr_ctx *ctx = nullptr;
};

#define NOT_YET(ctx) ((ctx) ? (ctx)->fail(PGR_ERR_STATE, "not implemented yet") : PGR_ERR_INVALID_ARG)

extern "C" int pgr_index_create(pgr_ctx *ctx, const pgr_spec *, pgr_index **) { return NOT_YET(ctx); }
extern "C" void pgr_index_destroy(pgr_index *ix) { delete ix; }
extern "C" int pgr_index_add_batch(pgr_ctx *ctx, pgr_index *, uint32_t, const uint8_t *const *, const uint64_t *,
                                   const uint32_t *) { return NOT_YET(ctx); }
extern "C" int pgr_index_add_resident(pgr_ctx *ctx, pgr_index *, const pgr_batch *, const uint32_t *) { return NOT_YET(ctx); }
extern "C" int pgr_index_add_records(pgr_ctx *ctx, pgr_index *, const pgr_frag_rec *, uint64_t, int) { return NOT_YET(ctx); }
extern "C" int pgr_index_finalize(pgr_ctx *ctx, pgr_index *) { return NOT_YET(ctx); }
extern "C" uint64_t pgr_index_n_keys(const pgr_index *) { return 0; }
extern "C" uint64_t pgr_index_n_records(const pgr_index *) { return 0; }
extern "C" int pgr_index_download(pgr_ctx *ctx, const pgr_index *, pgr_frag_rec **, uint64_t *) { return NOT_YET(ctx); }
extern "C" int pgr_query_hps_batch(pgr_ctx *ctx, const pgr_index *, uint32_t, const uint8_t *const *, const uint64_t *,
                                   float, uint32_t, uint32_t, uint32_t, uint32_t, int, uint32_t, int, pgr_hps_result *) {
    return NOT_YET(ctx);
}
extern "C" void pgr_hps_result_free(pgr_hps_result *) {}
extern "C" int pgr_sparse_aln_batch(pgr_ctx *ctx, uint32_t, const pgr_hitpair *, const uint64_t *, uint32_t, float, int,
                                    uint32_t, int, pgr_hps_result *) { return NOT_YET(ctx); }
