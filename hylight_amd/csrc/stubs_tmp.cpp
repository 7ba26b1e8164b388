// temporary stubs (replaced as the components land)
#include "ava.h"
#include "graph.h"
#include "stage.h"
namespace hlmi {
hlmi_ava_opts ava_opts_long() { hlmi_ava_opts o{19,5,1,100,10000,2000,3,10,2e-4,2,4,4,2,1}; return o; }
void ava_files(const char*, const char*, const hlmi_ava_opts&, const char*) { fail(HLMI_ESTATE, "ava not built yet"); }
void miniasm_run(const char*, const char*, int,int,int,int,const char*,const char*) { fail(HLMI_ESTATE, "miniasm not built yet"); }
void sfo2overlaps_run(const char*, const char*, int, int) { fail(HLMI_ESTATE, "sfo2overlaps not built yet"); }
struct Job::Impl {};
Job::Job(const char*, const char*, int, bool) { fail(HLMI_ESTATE, "stage not built yet"); }
Job::~Job() {}
size_t Job::num_queries() const { return 0; }
size_t Job::num_chunks() const { return 0; }
int64_t Job::sketch_bound(int64_t, int64_t) const { return 0; }
int64_t Job::sketch_range(int64_t, int64_t, void*, int64_t, void*) { return 0; }
void Job::set_query_sketch(const void*, int64_t, const void*) {}
void Job::sketch_all_queries() {}
void Job::run(int,int,int,int,double,const char*) {}
}
