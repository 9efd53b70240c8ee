// peng_oracle.cpp -- CPU restatement of PEnG-motif's k-mer-enrichment / PWM-EM hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
// (libpengk.so) never links, loads or falls back to it.
//
// Parity status: PINNED.  Every function below is compared bit-for-bit (integers, float32 bit
// patterns) against dumps of the compiled reference (oracle/_ref, built by oracle/Makefile from
// the sources in /root/reference) by tests/golden/make_golden.py; the dumps are committed under
// tests/golden/*.npz and re-checked by tests/test_oracle_golden.py on every run.
//
// Each function cites the reference file:line whose behaviour it restates.  Nothing here is
// copied: the scan is re-expressed as "segments + greedy spacing" with 64-bit positions, the
// revcomp as bit arithmetic, the background recursion as a flat per-pattern product.
//
// Conventions: base codes are the reference's Alphabet codes (0 = other, A,C,G,T = 1..4,
// src/shared/Alphabet.cpp:33-41).  Pattern ids are little-endian base 4 (first base = least
// significant digit, src/base_pattern.h:24-29); BaMM (k+1)-mer ids are big-endian
// (src/shared/Sequence.cpp:21-33).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#define PO_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------
// encodings
// ---------------------------------------------------------------------------------------------

// src/shared/Alphabet.cpp:33-41 -- ACGT in either case -> 1..4, everything else 0.
PO_API int po_base_code(int ch) {
  switch (ch) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    case 'T': case 't': return 4;
    default: return 0;
  }
}

// src/base_pattern.cpp:119-144 -- reverse complement of a W-digit little-endian base-4 id.
PO_API uint64_t po_revcomp(uint64_t id, int W) {
  uint64_t r = 0;
  for (int p = 0; p < W; ++p) {
    r = (r << 2) | (3u - (id & 3u));
    id >>= 2;
  }
  return r;
}

// src/base_pattern.h:88-103 -- BaMM id of digits [i-k, i] of a PEnG id (big-endian).
static inline unsigned bamm_id(uint64_t x, int i, int k) {
  unsigned y = 0;
  for (int q = i - k; q <= i; ++q) y = (y << 2) | (unsigned)((x >> (2 * q)) & 3u);
  return y;
}

// ---------------------------------------------------------------------------------------------
// FASTA reader (src/shared/SequenceSet.cpp:285-447).  Returns the number of kept records, or a
// negative error mirroring the reference's exit(1) cases: -1 file, -2 space in sequence,
// -3 sequence before header.  Output: concatenated codes + offsets (n+1 entries).
// The caller frees *codes_out / *offs_out with po_free.
// ---------------------------------------------------------------------------------------------
PO_API void po_free(void* p) { free(p); }

PO_API int64_t po_read_fasta(const char* path, uint8_t** codes_out, int64_t** offs_out) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  std::vector<char> buf;
  {
    char tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  }
  fclose(f);
  std::vector<uint8_t> codes;
  std::vector<int64_t> offs(1, 0);
  bool have_header = false;
  int64_t cur_start = 0;
  size_t pos = 0;
  const size_t n = buf.size();
  auto close_record = [&]() {
    // :317-349 / :383-421 -- a header without sequence is dropped and not counted.
    if (have_header && (int64_t)codes.size() > cur_start) offs.push_back((int64_t)codes.size());
    have_header = false;
  };
  while (pos < n) {
    // getline(...).good() (:304): a final line that is not newline-terminated is never seen.
    const char* nl = (const char*)memchr(buf.data() + pos, '\n', n - pos);
    if (!nl) break;
    size_t len = (size_t)(nl - (buf.data() + pos));
    const char* line = buf.data() + pos;
    pos += len + 1;
    if (len == 0) continue;  // blank (:306)
    if (line[0] == '>') {
      close_record();
      have_header = true;
      cur_start = (int64_t)codes.size();
    } else if (have_header) {
      if (memchr(line, ' ', len)) return -2;  // :361-364
      for (size_t i = 0; i < len; ++i) {
        unsigned char c = (unsigned char)line[i];
        codes.push_back((uint8_t)(c < 128 ? po_base_code(c) : 0));
      }
    } else {
      return -3;  // :369-373
    }
  }
  close_record();
  int64_t nrec = (int64_t)offs.size() - 1;
  *codes_out = (uint8_t*)malloc(codes.size() ? codes.size() : 1);
  memcpy(*codes_out, codes.data(), codes.size());
  *offs_out = (int64_t*)malloc(offs.size() * sizeof(int64_t));
  memcpy(*offs_out, offs.data(), offs.size() * sizeof(int64_t));
  return nrec;
}

// ---------------------------------------------------------------------------------------------
// k-mer count (src/base_pattern.cpp:331-441).
//
// Restated as: every sequence is cut into "visited runs" by the scan rule below; inside one
// sequence an occurrence of canonical id c at window-end position e is counted iff the last
// COUNTED occurrence of c ended at <= e - W (greedy spacing).  Positions are 64-bit here (the
// reference's `unsigned int j` / `last_match_pos` wrap at 2^32 positions, :335-336,:399-400;
// 64-bit is the intended semantics and identical below 2^32).
//
// Scan rule (incl. the off-by-one after an invalid base, :347-381): at i, try to read W valid
// bases; if an invalid base (or the end) is met first, resume right after it.  Otherwise visit
// windows one by one until the base to the right of the window is invalid (index q) or the
// sequence ends; then resume at q+2 -- the base at q+1 is skipped (:369,:380 plus the outer i++).
// ---------------------------------------------------------------------------------------------
PO_API void po_count(const uint8_t* codes, const int64_t* offs, int64_t nseq, int W, int both,
                     uint64_t* counts /* 4^W, zeroed here */, uint64_t* ltot_out) {
  const uint64_t NP = 1ull << (2 * W);
  const uint64_t mask = NP - 1;
  std::vector<uint64_t> last(NP, 0);
  memset(counts, 0, NP * sizeof(uint64_t));
  uint64_t ltot = 0;
  uint64_t base = (uint64_t)W;  // window-end coordinate of index 0 of the current sequence
  for (int64_t s = 0; s < nseq; ++s) {
    const uint8_t* seq = codes + offs[s];
    const int64_t L = offs[s + 1] - offs[s];
    int64_t i = 0;
    while (i < L) {
      int p = 0;
      uint64_t id = 0;
      while (p < W && i < L && seq[i] > 0) {
        id |= (uint64_t)(seq[i] - 1) << (2 * p);
        ++p;
        ++i;
      }
      if (p < W) {  // hit an invalid base (at i) or the end: resume at i+1
        ++i;
        continue;
      }
      uint64_t rc = po_revcomp(id, W);
      for (;;) {
        const uint64_t can = both ? std::min(id, rc) : id;
        const uint64_t e = base + (uint64_t)i;
        if (last[can] + (uint64_t)W <= e) {
          ++counts[can];
          last[can] = e;
        }
        ++ltot;
        if (i >= L || seq[i] == 0) break;
        const uint64_t c = (uint64_t)(seq[i] - 1);
        id = (id >> 2) | (c << (2 * (W - 1)));
        rc = ((rc << 2) & mask) | (3u - c);
        ++i;
      }
      i += 2;
    }
    base += (uint64_t)L + (uint64_t)W + 2u;  // >= W apart: no suppression across sequences (:382)
  }
  if (both) {  // :387-392 mirror to the twin id
    for (uint64_t x = 0; x < NP; ++x) {
      const uint64_t r = po_revcomp(x, W);
      if (x > r) counts[x] = counts[r];
    }
  }
  *ltot_out = ltot;
}

// ---------------------------------------------------------------------------------------------
// Background model counts (src/shared/Sequence.cpp:28-33, src/shared/BackgroundModel.cpp:60-84).
// n[k][y], y big-endian (k+1)-mer ending at i, i >= k.  With an invalid base among the (up to)
// 9 positions [i-8, i] the reference's kmer_ goes negative and `kmer_ % 4^(k+1) >= 0` holds only
// when the last k+1 digits are all zero, where an invalid base contributes digit 0: such a
// (k+1)-mer is counted as poly-A (y = 0), every other one is skipped.  K <= 2 here; out = 4+16+64.
// ---------------------------------------------------------------------------------------------
PO_API void po_bg_counts(const uint8_t* codes, const int64_t* offs, int64_t nseq, int K,
                         int64_t* out /* sum_{k<=K} 4^(k+1) entries, order k=0,1,2 */) {
  int64_t tot = 0;
  for (int k = 0; k <= K; ++k) tot += 1ll << (2 * (k + 1));
  memset(out, 0, (size_t)tot * sizeof(int64_t));
  for (int64_t s = 0; s < nseq; ++s) {
    const uint8_t* seq = codes + offs[s];
    const int64_t L = offs[s + 1] - offs[s];
    int64_t* nk = out;
    for (int k = 0; k <= K; ++k) {
      const unsigned m = (1u << (2 * (k + 1))) - 1u;
      for (int64_t i = k; i < L; ++i) {
        bool any_invalid = false;
        unsigned y = 0;
        const int64_t lo = i < 8 ? 0 : i - 8;
        for (int64_t q = lo; q <= i; ++q) {
          const unsigned d = seq[q] ? (unsigned)(seq[q] - 1) : 0u;
          if (!seq[q]) any_invalid = true;
          y = (y << 2) | d;
        }
        y &= m;
        if (!any_invalid || y == 0) ++nk[y];
      }
      nk += 1ll << (2 * (k + 1));
    }
  }
}

// calculateV, src/shared/BackgroundModel.cpp:490-530 (interpolate = true).  The reference keeps
// `int` counters (:54-56); counts are converted int -> float exactly as static_cast<float> does.
template <class I>
static void bg_V_impl(const int64_t* n, int K, const float* alpha, float* V) {
  const int64_t* nk[3];
  float* vk[3];
  {
    const int64_t* p = n;
    float* q = V;
    for (int k = 0; k <= K; ++k) {
      nk[k] = p;
      vk[k] = q;
      p += 1ll << (2 * (k + 1));
      q += 1ll << (2 * (k + 1));
    }
  }
  I base_counts = 0;
  for (int y = 0; y < 4; ++y) base_counts += (I)nk[0][y];
  for (int y = 0; y < 4; ++y)
    vk[0][y] = ((float)(I)nk[0][y] + alpha[0] * 0.25f) / ((float)base_counts + alpha[0]);
  for (int k = 1; k <= K; ++k) {
    const int ny = 1 << (2 * (k + 1));
    const int yk = 1 << (2 * k);
    for (int y = 0; y < ny; ++y) {
      const int y2 = y % yk;  // drop the oldest base
      const int yp = y / 4;   // drop the newest base
      vk[k][y] = ((float)(I)nk[k][y] + alpha[k] * vk[k - 1][y2]) / ((float)(I)nk[k - 1][yp] + alpha[k]);
    }
    for (int g = 0; g < ny; g += 4) {  // :519-528 normalise each context group by its running sum
      float factor = 0.0f;
      for (int a = 0; a < 4; ++a) factor += vk[k][g + a];
      for (int a = 3; a >= 0; --a) vk[k][g + a] /= factor;
    }
  }
}

// the reference's arithmetic: `int` counters and an `int` base count (src/shared/BackgroundModel.cpp:492-495) -- which WRAP
// beyond 2^31 bases, i.e. for every shard of BASELINE configs[3] (2.5e9 bases) -- ...
PO_API void po_bg_V(const int64_t* n, int K, const float* alpha, float* V /* same layout as n */) { bg_V_impl<int>(n, K, alpha, V); }
// ... and the intended semantics the product computes (64-bit counters; identical below 2^31 bases): what the
// reference-derived rows of sets beyond that size are built with (tests/golden/make_shard_golden.py), labelled so there
PO_API void po_bg_V64(const int64_t* n, int K, const float* alpha, float* V) { bg_V_impl<int64_t>(n, K, alpha, V); }

// ---------------------------------------------------------------------------------------------
// Per-pattern background probability of order k (src/base_pattern.cpp:285-325), float32 product
// in position order; V laid out as po_bg_V.  `both` applies the strand aggregation of :268-283:
// both twins get p[min] + p[max]; palindromes are left alone.
// ---------------------------------------------------------------------------------------------
// Threads for the two pattern-space sweeps below -- the loops the reference runs under OpenMP (src/base_pattern.cpp:232,
// 253, 261 static; :289 over the initial (k+1)-mers): every pattern is independent, so the values do not depend on the
// thread count.  1 by default (the reference's --threads default, src/Global.cpp:52); bench.py's cpu_baseline times both.
static int g_sweep_threads = 1;
PO_API void po_set_threads(int n) { g_sweep_threads = n < 1 ? 1 : n; }

PO_API void po_bgprob(int W, int k, const float* V, int both, float* out /* 4^W */) {
  const uint64_t NP = 1ull << (2 * W);
  const float* vk[3];
  {
    const float* q = V;
    for (int j = 0; j <= 2; ++j) {
      vk[j] = q;
      q += 1ll << (2 * (j + 1));
    }
  }
#pragma omp parallel for schedule(static) num_threads(g_sweep_threads)
  for (uint64_t x = 0; x < NP; ++x) {
    float pr = 1.0f;
    for (int j = 0; j <= k && j < W; ++j) pr *= vk[j][bamm_id(x, j, j)];
    for (int i = k + 1; i < W; ++i) pr *= vk[k][bamm_id(x, i, k)];
    out[x] = pr;
  }
  if (both) {
    for (uint64_t x = 0; x < NP; ++x) {
      const uint64_t r = po_revcomp(x, W);
      if (x < r) {
        const float s = out[x] + out[r];
        out[x] = s;
        out[r] = s;
      }
    }
  }
}

// expected / log-p / z (src/base_pattern.cpp:231-265).  Unqualified sqrt/log are the double
// overloads in the reference translation unit; mu/n and 1-mu/(n+1) follow the C++ promotions.
PO_API void po_stats(int W, const uint64_t* counts, const float* bgp, uint64_t ltot, float* expected,
                     float* logp, float* z) {
  const uint64_t NP = 1ull << (2 * W);
  const float fl = (float)ltot;
#pragma omp parallel for schedule(static) num_threads(g_sweep_threads)
  for (uint64_t x = 0; x < NP; ++x) {
    const float mu = bgp[x] * fl;
    expected[x] = mu;
    const uint64_t n = counts[x];
    if (n == 0) {
      logp[x] = std::numeric_limits<float>::infinity();
    } else {
      const float frac = (float)(1.0 - (double)(mu / (float)(n + 1)));
      if ((float)n > mu && n > 5) {
        const double v = (double)n * log((double)(mu / (float)n)) + (double)n - (double)mu -
                         0.5 * log(6.283 * (double)n * (double)frac * (double)frac);
        logp[x] = (float)v;
      } else {
        logp[x] = 0.0f;
      }
    }
    z[x] = (float)((double)((float)n - mu) / sqrt((double)mu));
  }
}

// Seed selection (src/base_pattern.cpp:443-515): non-stable std::sort of all ids by z descending
// (ties between revcomp twins are decided by libstdc++'s introsort, so the same call is used),
// then the threshold walk with the Hamming-1 neighbourhood filter.  Returns the number of seeds.
PO_API int64_t po_select(int W, const float* z, const uint64_t* counts, float z_thr, uint64_t count_thr,
                         int single_stranded, int filter_neighbors, uint64_t* seeds, int64_t max_seeds) {
  const size_t NP = (size_t)1 << (2 * W);
  std::vector<size_t> order(NP);
  for (size_t i = 0; i < NP; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [z](const size_t a, const size_t b) { return z[a] > z[b]; });
  std::vector<uint8_t> seen(NP, 0);
  int64_t ns = 0;
  for (size_t r = 0; r < NP; ++r) {
    const size_t x = order[r];
    if (z[x] < z_thr) break;
    if (counts[x] < count_thr) continue;
    if (seen[x]) continue;
    if (!single_stranded && seen[po_revcomp(x, W)]) continue;
    if (ns < max_seeds) seeds[ns] = x;
    ++ns;
    seen[x] = 1;
    if (filter_neighbors)
      for (int p = 0; p < W; ++p)
        for (uint64_t c = 0; c < 4; ++c) seen[(x & ~(3ull << (2 * p))) | (c << (2 * p))] = 1;
  }
  return ns;
}

// ---------------------------------------------------------------------------------------------
// IUPAC patterns (src/iupac_alphabet.cpp:138-180, src/iupac_pattern.cpp:331-473,806-833).
// Letters A0 C1 G2 T3 S4 W5 R6 Y7 M8 K9 N10; ids little-endian base 11.
// ---------------------------------------------------------------------------------------------
static const int kRepN[11] = {1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 4};
static const int kRep[11][4] = {{0}, {1}, {2}, {3}, {1, 2}, {0, 3}, {0, 2}, {1, 3}, {0, 1}, {2, 3}, {0, 1, 2, 3}};

// Expansion in the reference's emission order: the explicit LIFO stack of :341-365 yields an
// odometer in which the LAST degenerate position moves fastest and each position runs through its
// representatives in the order rep[0], rep[n-1], rep[n-2], ..., rep[1].  both=1 canonicalises and
// sorts ascending (:364,:367).  Returns the number of ids written (product of set sizes).
PO_API int64_t po_iupac_expand(uint64_t iupac, int W, int both, uint64_t* out, int64_t cap) {
  int letter[32], step[32];
  uint64_t t = iupac;
  int64_t total = 1;
  for (int p = 0; p < W; ++p) {
    letter[p] = (int)(t % 11);
    t /= 11;
    step[p] = 0;
    total *= kRepN[letter[p]];
  }
  if (total > cap) return -total;
  for (int64_t n = 0; n < total; ++n) {
    uint64_t id = 0;
    for (int p = 0; p < W; ++p) {
      const int cnt = kRepN[letter[p]];
      const int j = step[p] == 0 ? 0 : cnt - step[p];
      id |= (uint64_t)kRep[letter[p]][j] << (2 * p);
    }
    out[n] = both ? std::min(id, po_revcomp(id, W)) : id;
    for (int p = W - 1; p >= 0; --p) {  // advance: last position fastest
      if (++step[p] < kRepN[letter[p]]) break;
      step[p] = 0;
    }
  }
  if (both) std::sort(out, out + total);
  return total;
}

struct po_iupac_stats {
  uint64_t sites;    // sum of counts (n_sites)
  float bg_p;        // float32 sequential sum of background probabilities
  float expected;    // float32 sequential sum of expected counts
  float zscore;      // :446
  float log_pvalue;  // :453-470 incl. Bonferroni term
};

// log_bonferroni table (src/iupac_pattern.cpp:199-210): float(log(double)).
static float log_bonf(int letter) {
  if (letter < 4) return (float)log(8.0);
  if (letter < 8) return (float)log(16.0);
  if (letter < 10) return (float)log(24.0);
  return (float)log(6.0);
}

// aggregate_attributes_from_basepatterns, src/iupac_pattern.cpp:410-473: sequential float32 sums
// over the expansion, skipping an element equal to its predecessor.
PO_API int po_iupac_aggregate(uint64_t iupac, int W, int both, const uint64_t* counts, const float* bgp,
                              const float* expected, po_iupac_stats* out) {
  int64_t total = 1;
  {
    uint64_t t = iupac;
    for (int p = 0; p < W; ++p) {
      total *= kRepN[t % 11];
      t /= 11;
    }
  }
  std::vector<uint64_t> ids((size_t)total);
  po_iupac_expand(iupac, W, both, ids.data(), total);
  uint64_t last = ids[0];
  float sum_bg = bgp[last];
  uint64_t sum_c = counts[last];
  float sum_e = expected[last];
  for (int64_t i = 1; i < total; ++i) {
    const uint64_t x = ids[i];
    if (x != last) {
      sum_bg += bgp[x];
      sum_c += counts[x];
      sum_e += expected[x];
    }
    last = x;
  }
  out->sites = sum_c;
  out->bg_p = sum_bg;
  out->expected = sum_e;
  out->zscore = (float)((double)((float)sum_c - sum_e) / sqrt((double)sum_e));
  if (sum_c == 0) {
    out->log_pvalue = std::numeric_limits<float>::infinity();
  } else {
    const float mu = sum_e;
    const float frac = 1 - mu / (float)(sum_c + 1);  // float arithmetic here (:457), unlike BasePattern
    float lp = 0;
    if ((float)sum_c > mu && sum_c > 5 && out->zscore > 2) {
      lp = (float)((double)sum_c * log((double)(mu / (float)sum_c)) + (double)sum_c - (double)mu -
                   0.5 * log(6.283 * (double)sum_c * (double)frac * (double)frac));
    }
    uint64_t t = iupac;
    for (int p = 0; p < W; ++p) {
      lp += log_bonf((int)(t % 11));
      t /= 11;
    }
    out->log_pvalue = lp;
  }
  return 0;
}

// count_combined_occurences, src/iupac_pattern.cpp:806-833.
PO_API uint64_t po_iupac_count(uint64_t iupac, int W, int both, const uint64_t* counts) {
  int64_t total = 1;
  {
    uint64_t t = iupac;
    for (int p = 0; p < W; ++p) {
      total *= kRepN[t % 11];
      t /= 11;
    }
  }
  std::vector<uint64_t> ids((size_t)total);
  po_iupac_expand(iupac, W, both, ids.data(), total);
  uint64_t sum = counts[ids[0]];
  for (int64_t i = 1; i < total; ++i)
    if (!both || ids[i] != ids[i - 1]) sum += counts[ids[i]];
  return sum;
}

// Mutual-information optimisation score (src/utils.h:25-37, src/iupac_pattern.cpp:652-669,
// src/base_pattern.cpp:184-200): float variables, exp/log in double, lower is better.
static float entropy_nat(float p) { return (float)(-(double)p * log((double)p) - (double)(1 - p) * log((double)(1 - p))); }
static float mi_fast(float obs, float exp_, unsigned nseq, float prior) {
  const float p_obs = (float)(1 - exp((double)(-obs / (float)nseq)));
  const float p_exp = (float)(1 - exp((double)(-exp_ / (float)nseq)));
  const float q = prior;
  const float p = p_obs * q + p_exp * (1 - q);
  return -q * entropy_nat(p_obs) - (1 - q) * entropy_nat(p_exp) + entropy_nat(p);
}
PO_API float po_mi_score(float observed, float expected, unsigned nseq) {
  if (observed < expected) return 0;
  float score = 0;
  const float qs[3] = {0.5f, 0.1f, 0.01f};
  for (int i = 0; i < 3; ++i) score += mi_fast(observed, expected, nseq, qs[i]) / entropy_nat(qs[i]);
  return -score;
}

// ---------------------------------------------------------------------------------------------
// EM (src/peng.cpp:48-197, src/iupac_pattern.cpp:291-303).  mode 0: float32 serial accumulation
// in pattern order == the reference bit for bit.  mode 1: identical per-element float32 terms,
// accumulated in double (the order-independent target the device path is held to).
// pwm is W x 4 row-major, updated in place; returns the iteration count.  `change_out` receives
// the last computed change.  The trailing re-normalisation of the IUPACPattern(ori, pwm)
// constructor (src/iupac_pattern.cpp:61) is applied when final_norm != 0.
// ---------------------------------------------------------------------------------------------
PO_API int po_em(int W, const uint64_t* counts, const float* bg, float* pwm, float saturation,
                 float threshold, int max_iter, int mode, int final_norm, float* change_out) {
  const uint64_t NP = 1ull << (2 * W);
  float oldp[32][4], newp[32][4];
  for (int p = 0; p < W; ++p)
    for (int a = 0; a < 4; ++a) oldp[p][a] = pwm[p * 4 + a];
  float change = (float)W;
  int it = 0;
  while (!(change <= threshold || it >= max_iter)) {
    ++it;
    double acc[32][4];
    float accf[32][4];
    for (int p = 0; p < W; ++p)
      for (int a = 0; a < 4; ++a) {
        acc[p][a] = 0.0;
        accf[p][a] = 0.0f;
      }
    for (uint64_t x = 0; x < NP; ++x) {
      float pr = 1.0f;  // :180-197: ((1*pwm[0][x0])*pwm[1][x1])...
      for (int p = 0; p < W; ++p) pr = pr * oldp[p][(x >> (2 * p)) & 3];
      const float odds = pr / bg[x];
      const float w = ((float)counts[x] * saturation) / (1 + saturation / odds);  // :124-125
      if (mode == 0) {
        for (int p = 0; p < W; ++p) accf[p][(x >> (2 * p)) & 3] += w;
      } else {
        for (int p = 0; p < W; ++p) acc[p][(x >> (2 * p)) & 3] += (double)w;
      }
    }
    for (int p = 0; p < W; ++p) {
      for (int a = 0; a < 4; ++a) newp[p][a] = mode == 0 ? accf[p][a] : (float)acc[p][a];
      float sum = 0;
      for (int a = 0; a < 4; ++a) sum += newp[p][a];
      for (int a = 0; a < 4; ++a) newp[p][a] /= sum;
    }
    change = 0;
    for (int p = 0; p < W; ++p)
      for (int a = 0; a < 4; ++a) change += std::abs(newp[p][a] - oldp[p][a]);
    memcpy(oldp, newp, sizeof oldp);
  }
  if (final_norm) {
    for (int p = 0; p < W; ++p) {
      float sum = 0;
      for (int a = 0; a < 4; ++a) sum += oldp[p][a];
      for (int a = 0; a < 4; ++a) oldp[p][a] /= sum;
    }
  }
  for (int p = 0; p < W; ++p)
    for (int a = 0; a < 4; ++a) pwm[p * 4 + a] = oldp[p][a];
  if (change_out) *change_out = change;
  return it;
}

// One EM accumulation pass only (no normalisation): raw new_pwm sums in double; used to check the
// device kernel's accumulators element for element.
PO_API void po_em_accumulate(int W, const uint64_t* counts, const float* bg, const float* pwm, float saturation,
                             double* acc /* W*4 */) {
  const uint64_t NP = 1ull << (2 * W);
  for (int i = 0; i < W * 4; ++i) acc[i] = 0.0;
  for (uint64_t x = 0; x < NP; ++x) {
    float pr = 1.0f;
    for (int p = 0; p < W; ++p) pr = pr * pwm[p * 4 + ((x >> (2 * p)) & 3)];
    const float odds = pr / bg[x];
    const float w = ((float)counts[x] * saturation) / (1 + saturation / odds);
    for (int p = 0; p < W; ++p) acc[p * 4 + ((x >> (2 * p)) & 3)] += (double)w;
  }
}

// ---------------------------------------------------------------------------------------------
// Synthetic input of SURVEY.md section 8d (counter-based; any shard can be generated alone).
// Writes base codes 1..4 for sequences [seq0, seq0+nseq) of length L.
// ---------------------------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
PO_API void po_synth(uint64_t seed, uint64_t seq0, uint64_t nseq, uint32_t L, uint8_t* codes /* nseq*L */) {
  static const char motif[11] = "GCTGAGTCAT";
  for (uint64_t s = 0; s < nseq; ++s) {
    const uint64_t n = seq0 + s;
    uint8_t* out = codes + s * L;
    for (uint32_t j = 0; j < L; ++j)
      out[j] = (uint8_t)(1 + (mix64(seed + 0x9E3779B97F4A7C15ull * (n * (uint64_t)L + j + 1)) >> 62));
    if (L >= 10 && mix64(seed ^ 0xA5A5A5A5ull ^ (n + 1)) % 10 == 0) {
      const uint64_t q = mix64(seed ^ 0x5A5A5A5Aull ^ (n + 1)) % (L - 9);
      for (int t = 0; t < 10; ++t) out[q + t] = (uint8_t)po_base_code(motif[t]);
    }
  }
}
