// textpass.h - host text passes next to the hot path; see textpass.cpp
#pragma once
namespace hlmi {
void filter_non_atcg_run(const char *fastx, const char *out_fa, bool fastq);
void gfa2fa_run(const char *gfa, const char *out_fa);
void pick_up_run(const char *paf, const char *fastx, const char *out_fastx, bool fastq);
}  // namespace hlmi
