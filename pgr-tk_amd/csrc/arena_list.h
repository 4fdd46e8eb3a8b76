// arena_list.h -- the free list of a reserved device block (pgr_ctx_reserve): offsets only, no HIP; the context (ctx.hip) owns the
// memory.  Address-ordered map of free ranges, best fit, a freed range is merged with free neighbours.  Header-only so that the
// CPU test (tests/arena_list_harness.cpp) drives exactly this code against a byte map.
#pragma once
#include <cstddef>
#include <iterator>
#include <map>

namespace pgr {

struct ArenaList {
    static constexpr size_t NONE = ~(size_t)0;
    size_t size = 0;
    std::map<size_t, size_t> free_by_off;  // offset -> length

    void reset(size_t bytes) {
        size = bytes;
        free_by_off.clear();
        if (bytes) free_by_off[0] = bytes;
    }
    // smallest free range that holds `want` bytes: its front part is handed out; NONE when nothing fits
    size_t take(size_t want) {
        auto best = free_by_off.end();
        for (auto it = free_by_off.begin(); it != free_by_off.end(); ++it)
            if (it->second >= want && (best == free_by_off.end() || it->second < best->second)) best = it;
        if (best == free_by_off.end()) return NONE;
        const size_t off = best->first, len = best->second;
        free_by_off.erase(best);
        if (len > want) free_by_off[off + want] = len - want;
        return off;
    }
    void give_back(size_t off, size_t len) {
        auto nx = free_by_off.lower_bound(off);
        if (nx != free_by_off.end() && off + len == nx->first) {  // the range behind is free: one range
            len += nx->second;
            nx = free_by_off.erase(nx);
        }
        if (nx != free_by_off.begin()) {  // ... and the one in front
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {
                pv->second += len;
                return;
            }
        }
        free_by_off[off] = len;
    }
    size_t free_bytes() const {
        size_t s = 0;
        for (const auto &kv : free_by_off) s += kv.second;
        return s;
    }
};

}  // namespace pgr
