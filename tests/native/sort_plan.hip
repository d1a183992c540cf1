// CPU check of the host logic of the preparation's sort (eskf_lio_amd/csrc/vgicp_sort.h: plan_for, launches_for,
// split_bytes) — compiled by hipcc, no device call: for sizes across the range the groups of the levels are 2 or 4 (the
// measured optimum), their product covers every wave tile of the scan, no level could be dropped or made smaller, and
// the splitter room holds one key per four pairs of the first level for both alternating arrays.
#include <cstdio>

#include "../../eskf_lio_amd/csrc/vgicp_sort.h"

int main() {
  using namespace vgicp::sortk;
  unsigned long long checked = 0;
  for (unsigned long long n = 1; n <= 20000000ull; n = n < 5000 ? n + 1 : n + n / 97 + 1) {
    const Plan p = plan_for((uint32_t)n);
    const unsigned long long tiles = (n + kTile - 1) / kTile;
    unsigned long long reach = 1;
    for (int l = 0; l < p.levels; ++l) {
      if (p.group[l] != 2 && p.group[l] != 4) { std::printf("n = %llu: level %d merges %d runs\n", n, l, p.group[l]); return 1; }
      reach *= (unsigned long long)p.group[l];
    }
    if (reach < tiles) { std::printf("n = %llu: %d levels reach %llu of %llu tiles\n", n, p.levels, reach, tiles); return 1; }
    if (p.levels > 0) {
      // fewest levels: one level less of the largest group does not reach
      unsigned long long less = 1;
      for (int l = 0; l + 1 < p.levels; ++l) less *= (unsigned long long)kMaxGroup;
      if (less >= tiles) { std::printf("n = %llu: %d levels where %d would do\n", n, p.levels, p.levels - 1); return 1; }
      // smallest groups: halving the last level's group does not reach either
      if (reach / (unsigned long long)p.group[p.levels - 1] * (unsigned long long)(p.group[p.levels - 1] / 2) >= tiles && p.group[p.levels - 1] > 2) {
        std::printf("n = %llu: the last level's group of %d could be smaller\n", n, p.group[p.levels - 1]);
        return 1;
      }
    }
    if (launches_for((uint32_t)n) != 1u + (uint32_t)p.levels) { std::printf("n = %llu: launches_for disagrees with the plan\n", n); return 1; }
    if (split_bytes((uint32_t)n, 8) / 2 < (n / 4 + 1) * 8) { std::printf("n = %llu: too little room for the splitters\n", n); return 1; }
    ++checked;
  }
  if (launches_for(0) != 0u || launches_for(60000) != 5u || launches_for(256) != 1u || launches_for(257) != 2u) { std::printf("known plans differ\n"); return 1; }
  std::printf("ok %llu sizes\n", checked);
  return 0;
}
