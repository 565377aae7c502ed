// Translation unit of the fused news-encoder back half (nrl_news_tail.h): the two launches the C ABI uses.
#include "nrl_news_tail.h"

namespace nrl {

// workgroup shape: 4 waves, two workgroups per CU (tools/nt_probe.hip, profiles/r03_news_tail_probe.txt: forward
// 0.25 -> 0.20 ms, backward 0.30 -> 0.25 ms against the 8-wave shape at B = 128)
int news_tail_fwd(const NewsTailArgs& a, hipStream_t st) { return launch_news_tail_fwd<4, 0>(a, st); }
int news_tail_bwd(const NewsTailBwdArgs& a, hipStream_t st) { return launch_news_tail_bwd<4, 0>(a, st); }

}  // namespace nrl
