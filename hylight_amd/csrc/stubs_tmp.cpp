// temporary stubs (replaced as the components land)
#include "graph.h"
namespace hlmi {
void miniasm_run(const char*, const char*, int,int,int,int,const char*,const char*) { fail(HLMI_ESTATE, "miniasm not built yet"); }
void sfo2overlaps_run(const char*, const char*, int, int) { fail(HLMI_ESTATE, "sfo2overlaps not built yet"); }
}
