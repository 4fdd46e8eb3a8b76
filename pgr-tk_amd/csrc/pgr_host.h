// pgr_host.h -- host-only helpers of libpgrhip.so (csrc/hostpack.cpp): the CPU packer and the staging thread pool.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>

namespace pgr {

// CPUs this process may use (affinity capped by the cgroup quota; PGR_HOST_THREADS overrides)
unsigned host_cpus();

// malloc-compatible (released with free / pgr_free) memory for big results: huge-page advised
void *host_result_alloc(size_t bytes);

// Persistent worker threads for the host side of the staging pipelines (packing, pinned-window copies).  Loops may be
// submitted from several threads at once (the staging thread of a pipelined call and the caller's download); the
// submitting thread always works on its own loop.
class HostPool {
  public:
    static HostPool &instance();
    unsigned workers() const;
    // fn(i) for i in [0, n), on at most max_par threads including the caller (0: no limit)
    void parallel_for(size_t n, const std::function<void(size_t)> &fn, unsigned max_par = 0);
    ~HostPool();

  private:
    explicit HostPool(unsigned n_workers);
    struct Impl;
    Impl *impl;
};

// words [w0, w1) of ONE contig (seq, len) -> planes[0 .. w1-w0), valid[0 .. w1-w0); returns the number of non-ACGT bytes.
// Bits past the contig's end are zero in all three planes.
uint64_t pack_words(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid);
// into a pinned staging window: non-temporal stores + store fence
uint64_t pack_words_stream(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid);
uint64_t pack_words_stream_nofence(const uint8_t *seq, uint64_t len, uint64_t w0, uint64_t w1, uint64_t *planes, uint32_t *valid);
void stream_fence();
void stream_copy(void *dst, const void *src, size_t n);

}  // namespace pgr
