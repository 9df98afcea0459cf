// sampler_host.hip -- BAN's adaptive proposal sampling as HOST code inside the C-ABI library (no kernel here; the file is
// .hip only so that the one Makefile rule builds it).  Reference: models/BANlib/model.py:357-435 (`iou`,
// `proposal_selection_with_negative`, `Aaptive_Proposal_Sampling`): per clip a greedy, data-dependent loop over the kept
// cells of the score map in descending score order -- sequential in its <= topk picks and tiny (20 x one IoU sweep over
// ~5 k moments), so it runs on the host between the map stage and the proposal head: the caller copies the [B, C] score
// rows down, this routine fills [B, n_out, 2], the caller copies that up.  The numpy restatement of the same loop
// (vmrframe_amd/ban_sampler.py) took 34 ms for 64 clips x 5376 cells; this takes ~1 ms on 8 threads.
// Ties between equal scores keep cell order (stable sort), as in the numpy version.
#include <algorithm>
#include <numeric>
#include <thread>
#include <vector>

#include "common.h"

namespace {

void sample_clip(const float* sc, const int32_t* cells, int C, float thresh, int topk, int neighbor, int negative, int n_out,
                 int64_t* out, int* n_written) {
  std::vector<int> order(C);
  std::iota(order.begin(), order.end(), 0);
  // NaN scores (a diverged run, or uninitialised memory) sort FIRST, as torch.sort(descending=True) orders them: the
  // key maps NaN to +inf so that the comparator stays a strict weak ordering (std::stable_sort on `a > b` with NaNs is UB)
  auto key = [&](int i) { return sc[i] != sc[i] ? __builtin_huge_valf() : sc[i]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) > key(b); });
  std::vector<float> st(C), en(C);
  for (int r = 0; r < C; ++r) {
    st[r] = (float)cells[2 * order[r]];
    en[r] = (float)(cells[2 * order[r] + 1] + 1);          // (start, end + 1): the reference's `grids[:, 1] += 1`
  }
  std::vector<char> suppressed(C, 0), select(C, 0);
  int count = 0;
  for (int i = 0; i + 1 < C; ++i) {
    if (suppressed[i]) continue;
    const float s = st[i], e = en[i];
    suppressed[i] = 1;
    select[i] = 1;
    int nb = 0;
    for (int r = i + 1; r < C; ++r) {
      const float inter = std::min(en[r], e) - std::max(st[r], s);
      const float uni = std::max(en[r], e) - std::min(st[r], s);
      if (std::max(inter, 0.f) / uni > thresh) {
        if (nb < neighbor) { select[r] = 1; ++nb; }
        suppressed[r] = 1;
      }
    }
    if (++count == topk) break;
  }
  const int total = topk * (neighbor + 1);
  std::vector<int> free_r, sel_r;
  for (int r = 0; r < C; ++r) {
    if (!suppressed[r]) free_r.push_back(r);
    if (select[r]) sel_r.push_back(r);
  }
  std::vector<int> res;
  for (int k = 0; k < negative && k < (int)free_r.size(); ++k) res.push_back(free_r[free_r.size() - 1 - k]);
  if ((int)sel_r.size() < total)
    for (int k = 0; k < total - (int)sel_r.size() && k < (int)free_r.size(); ++k) res.push_back(free_r[k]);
  for (int r : sel_r) res.push_back(r);
  *n_written = (int)res.size();
  for (int k = 0; k < (int)res.size() && k < n_out; ++k) {
    out[2 * k] = (int64_t)st[res[k]];
    out[2 * k + 1] = (int64_t)en[res[k]];
  }
}

}  // namespace

// scores [B][C] host float (score_pred at the kept cells, in mask.nonzero() row-major order), cells [C][2] host int32 (i, j);
// out [B][n_out][2] host int64 receives (start, end + 1) in the reference's order [negatives | padding | selected by rank].
// Returns 0, or -22 when a clip yields a count != n_out (the reference's .view(B, prop_num, ...) would fail as well).
extern "C" int vmr_ban_sample_host(const float* scores, const int32_t* cells, int B, int C, float thresh, int topk, int neighbor,
                                   int negative, int n_out, int64_t* out) {
  VMR_CHECK(scores && cells && out, "vmr_ban_sample_host: null pointer");
  VMR_CHECK(B >= 0 && C > 0 && topk > 0 && neighbor >= 0 && negative >= 0 && n_out > 0, "vmr_ban_sample_host: bad arguments");
  std::vector<int> written(B, 0);
  const int nthreads = std::max(1, std::min({B, 8, (int)std::thread::hardware_concurrency()}));
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t)
    pool.emplace_back([&, t]() {
      for (int b = t; b < B; b += nthreads)
        sample_clip(scores + (int64_t)b * C, cells, C, thresh, topk, neighbor, negative, n_out, out + (int64_t)b * n_out * 2,
                    &written[b]);
    });
  for (auto& th : pool) th.join();
  for (int b = 0; b < B; ++b)
    VMR_CHECK(written[b] == n_out, "vmr_ban_sample_host: clip %d yields %d proposals, expected %d", b, written[b], n_out);
  return 0;
}
