// stage.h - one split_reads2 stage as a job object; also the staged
// multi-GPU entry points (sketch shard / install gathered sketch / run chunk share).
#pragma once
#include <memory>

#include "common.h"

namespace hlmi {
class Job {
  public:
    Job(const char *reads_fa, const char *ref_fa, int nsplit, bool long_mode);
    ~Job();
    size_t num_queries() const;
    size_t num_chunks() const;
    int64_t sketch_bound(int64_t lo, int64_t hi) const;
    int64_t sketch_range(int64_t lo, int64_t hi, void *dev_mz, int64_t cap, void *dev_counts);
    void set_query_sketch(const void *dev_mz, int64_t n, const void *dev_counts);
    void sketch_all_queries();
    void run(int rank, int world, int len_over, int mc, double iden, const char *out_paf);
    struct Impl;
  private:
    std::unique_ptr<Impl> impl_;
};
}  // namespace hlmi
