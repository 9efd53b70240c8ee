// similarity.hip -- the motif similarity grid behind Peng::merge_iupac_patterns (src/peng.cpp:237-313): for every pair
// of motifs the best score of IUPACPattern::calculate_S (src/iupac_pattern.cpp:568-615) over all shifts with at least
// MIN_MERGE_OVERLAP = 6 overlapping columns and, under both strands, the reverse complement of the motif with fewer
// sites -- O(n^2 * shifts * W) Jensen-Shannon terms, recomputed by the reference after every merge.
//
// The device result is an fp64 evaluation rounded to float: the reference rounds its running sums to float after every
// term, so the two agree to ~1e-4, not to the bit.  That is enough for what the grid is used for: the merge loop needs
// the pair with the highest score, so the host mirror takes every pair within a margin of the device maximum and
// evaluates only those with the reference's own arithmetic (host/peng.cpp) -- decisions and printed scores stay the
// reference's, the n^2 grid moves here.
//
// One wave per pair.  s(shift) = 0.5 (d_bg(big) + d_bg(small)) - d(big, small) over the overlapping columns
// (calculate_s, :551-566), d(x, y) = sum of  x' lg x' + y' lg y' - (x' + y') lg((x' + y') / 2)  with x' = x + 1e-4.
// Column sums of the single-motif parts are prefix-summed in LDS; a lane owns one (strand, shift) and pays one log2 per
// overlapping PWM entry for the cross term.
#include <math.h>

#include "pengk_internal.h"

namespace pengk {
namespace {

constexpr int MAXL = PENGK_MAX_MOTIF_LEN;
constexpr int MIN_OVERLAP = 6;  // MIN_MERGE_OVERLAP, src/iupac_pattern.h
constexpr double EPS = (double)1e-4f;

__device__ __forceinline__ double xlgx(double v) { return v * log2(v); }

// pairs (i, j), i < j, j >= first_new; blockIdx.x enumerates them row-major over j
__global__ __launch_bounds__(64) void similarity_kernel(const float* __restrict__ pwm, const float* __restrict__ comp,
                                                        const int32_t* __restrict__ len, const uint64_t* __restrict__ sites,
                                                        int n, int both, float bg0, float bg1, float bg2, float bg3,
                                                        int first_new, float* __restrict__ out) {
  // decode the pair: for j = first_new .. n-1 there are j pairs (i = 0 .. j-1)
  long long q = blockIdx.x;
  int j = first_new;
  while (q >= j) {  // at most n - first_new steps; wave-uniform
    q -= j;
    ++j;
  }
  const int i = (int)q;
  if (j >= n) return;
  const int lane = threadIdx.x;
  // big / small as calculate_S assigns them: p1 = motif i, p2 = motif j; the longer is "big", ties keep p1
  const int li = len[i], lj = len[j];
  const int b = li < lj ? j : i, s = li < lj ? i : j;
  const int lb = len[b], ls = len[s];
  const float bgv[4] = {bg0, bg1, bg2, bg3};

  __shared__ float PB[2][MAXL][4], PS[2][MAXL][4];   // [orientation][column][letter]
  __shared__ double SB[2][MAXL + 1], SS[2][MAXL + 1];  // prefix sums of the single-motif parts of s, per orientation
  const int n_orient = both ? 2 : 1;
  for (int o = 0; o < n_orient; ++o) {
    const bool comp_big = o == 1 && sites[b] < sites[s];
    const bool comp_small = o == 1 && !comp_big;
    const float* src_b = (comp_big ? comp : pwm) + (size_t)b * MAXL * 4;
    const float* src_s = (comp_small ? comp : pwm) + (size_t)s * MAXL * 4;
    for (int e = lane; e < MAXL * 4; e += 64) {
      (&PB[o][0][0])[e] = e < lb * 4 ? src_b[e] : 0.0f;
      (&PS[o][0][0])[e] = e < ls * 4 ? src_s[e] : 0.0f;
    }
  }
  __syncthreads();
  // per column c of a motif: 0.5 * d_bg(column) - sum_a x' lg x'   (the part of s that does not depend on the partner)
  for (int o = 0; o < n_orient; ++o) {
    for (int which = 0; which < 2; ++which) {
      const int L = which ? ls : lb;
      double v = 0.0;
      if (lane < L) {
        for (int a = 0; a < 4; ++a) {
          const double x = (double)(which ? PS[o][lane][a] : PB[o][lane][a]) + EPS;
          const double y = (double)bgv[a] + EPS;
          const double self = xlgx(x);
          const double dbg = self + xlgx(y) - (x + y) * log2(0.5 * (x + y));
          v += 0.5 * dbg - self;
        }
      }
      // inclusive scan over the lanes (columns), written as prefix[c + 1]
      for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
      }
      double* dst = which ? SS[o] : SB[o];
      if (lane < L) dst[lane + 1] = v;
      if (lane == 0) dst[0] = 0.0;
    }
  }
  __syncthreads();
  // lane -> (orientation, shift)
  const int n_shift = lb + ls - 2 * MIN_OVERLAP + 1;  // shift = MIN_OVERLAP - ls .. lb - MIN_OVERLAP
  double best = -INFINITY;
  for (int t = lane; t < n_orient * (n_shift > 0 ? n_shift : 0); t += 64) {
    const int o = t / n_shift, shift = MIN_OVERLAP - ls + t % n_shift;
    const int off_small = shift < 0 ? -shift : 0, off_big = shift > 0 ? shift : 0;
    const int overlap = min(lb - off_big, ls - off_small);
    double cross = 0.0;  // sum of (x' + y') lg((x' + y') / 2)
    for (int c = 0; c < overlap; ++c)
      for (int a = 0; a < 4; ++a) {
        const double m = (double)PB[o][off_big + c][a] + (double)PS[o][off_small + c][a] + 2.0 * EPS;
        cross += m * log2(0.5 * m);
      }
    const double sc = (SB[o][off_big + overlap] - SB[o][off_big]) + (SS[o][off_small + overlap] - SS[o][off_small]) + cross;
    best = fmax(best, sc);
  }
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
  if (lane == 0) out[blockIdx.x] = (float)best;  // pair order: j = first_new .. n-1, i = 0 .. j-1
}

}  // namespace

int launch_similarity(pengk_ctx* ctx, int n, const float* d_pwm, const float* d_comp, const int32_t* d_len,
                      const uint64_t* d_sites, int both, const float* h_bg, int first_new, float* d_out) {
  long long pairs = 0;
  for (int j = first_new; j < n; ++j) pairs += j;
  if (pairs == 0) return PENGK_OK;
  if (pairs > 0x7FFFFFFFll) return fail(PENGK_ERR_RANGE, "pengk_motif_similarity: %lld pairs in one call", pairs);
  hipLaunchKernelGGL(similarity_kernel, dim3((unsigned)pairs), dim3(64), 0, ctx->stream, d_pwm, d_comp, d_len, d_sites, n, both,
                     h_bg[0], h_bg[1], h_bg[2], h_bg[3], first_new, d_out);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

// (pengk_warmup: loads this translation unit's code object ahead of its first launch)
int warm_similarity() {
  hipFuncAttributes a;
  PENGK_HIP(hipFuncGetAttributes(&a, (const void*)similarity_kernel));
  return PENGK_OK;
}

}  // namespace pengk
