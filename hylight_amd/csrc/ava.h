// ava.h - the overlapper: sketch -> index -> seed -> chain -> align -> PAF.
#pragma once
#include "common.h"

namespace hlmi {
hlmi_ava_opts ava_opts_long();
// one target file vs one query file, PAF text out (the minimap2 call of filter_overlap_slr2.py:51)
void ava_files(const char *target_fa, const char *query_fa, const hlmi_ava_opts &o, const char *out_paf);
}  // namespace hlmi
