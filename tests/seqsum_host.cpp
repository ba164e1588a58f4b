// Host build of graphlearning_amd/csrc/seqsum_exact.h for tests/test_seqsum_exact.py: the plain chain, and the same sum through
// block records prepared from an approximate (optionally perturbed) prefix -- the scalar twin of the kernels in cg_seqsum.hip.
#include "../graphlearning_amd/csrc/seqsum_exact.h"
#include <vector>

extern "C" double ss_host_chain(const double* x, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s = s + x[i];
  return s;
}

// records of nb blocks: plain sums of the runs of SS_SUB rows, their plain prefix (perturbed by `noise`, relative), one record per
// run, merged per block in the device's tree order
static void host_records(const double* x, int64_t n, int64_t nb, double noise, std::vector<SsRec>& rec) {
  const int64_t nsub = nb * SS_Q;
  std::vector<double> apx(nsub + 1, 0.0);
  for (int64_t q = 0; q < nsub; ++q) {
    double t = 0.0;
    for (int64_t i = q * SS_SUB; i < n && i < (q + 1) * SS_SUB; ++i) t += x[i];
    apx[q + 1] = apx[q] + t;
  }
  for (int64_t q = 0; q < nsub; ++q) apx[q] *= 1.0 + noise * ((q * 2654435761u) % 1000 / 500.0 - 1.0);
  rec.resize(nb);
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t left = n - b * SS_BLOCK;
    const int len = left <= 0 ? 0 : (left < SS_BLOCK ? (int)left : SS_BLOCK);
    ss_block_record(x + (len ? b * SS_BLOCK : 0), 1, len, &apx[b * SS_Q], &rec[b]);
  }
}

// stats: [0] blocks applied in integer form, [1] blocks taken row by row, [2] rows added exactly inside applied blocks
extern "C" double ss_host_blocks(const double* x, int64_t n, double noise, int64_t* stats) {
  const int64_t nb = (n + SS_BLOCK - 1) / SS_BLOCK;
  std::vector<SsRec> rec;
  host_records(x, n, nb, noise, rec);
  double s = 0.0;
  stats[0] = stats[1] = stats[2] = 0;
  for (int64_t b = 0; b < nb; ++b) {
    if (ss_apply_record(&s, &rec[b])) { stats[0]++; stats[2] += rec[b].nsplit; continue; }
    stats[1]++;
    for (int64_t i = b * SS_BLOCK; i < n && i < (b + 1) * SS_BLOCK; ++i) s = s + x[i];
  }
  return s;
}

// The walk of cg_seqsum.hip's ss_walk_kernel, lane loops written out: chunks of 64 block records with the chunk-local prefix of
// the plain blocks' totals, the first block that is not a plain same-binade block found per round, its record or its rows, on.
// stats: [0] plain, [1] through the record, [2] row by row.
extern "C" double ss_host_walk(const double* x, int64_t n, double noise, int64_t* stats) {
  const int64_t nchunks = (n + (int64_t)SS_BLOCK * 64 - 1) / ((int64_t)SS_BLOCK * 64);
  const int64_t nb = nchunks * 64;
  std::vector<SsRec> rec;
  host_records(x, n, nb, noise, rec);
  std::vector<uint64_t> excl(nb, 0);
  for (int64_t c = 0; c < nchunks; ++c) {
    uint64_t e = 0;
    for (int l = 0; l < 64; ++l) {
      const SsRec& r = rec[c * 64 + l];
      excl[c * 64 + l] = e;
      if (r.nsplit == 0 && r.E[0] >= 0) e += (uint64_t)r.R[0];
    }
  }
  double s = 0.0;
  stats[0] = stats[1] = stats[2] = 0;
  for (int64_t c = 0; c < nchunks; ++c) {
    const SsRec* cur = &rec[c * 64];
    const uint64_t* ex = &excl[c * 64];
    const bool plain63 = cur[63].nsplit == 0 && cur[63].E[0] >= 0;
    const uint64_t total = ex[63] + (plain63 ? (uint64_t)cur[63].R[0] : 0);
    int start = 0;
    while (start < 64) {
      const bool valid = ss_valid(s);
      const int E = ss_expo(s);
      const int64_t K = ss_mant(s);
      const uint64_t ex0 = ex[start];
      int f = 64;
      for (int l = start; l < 64; ++l) {
        const bool plain = cur[l].nsplit == 0 && cur[l].E[0] >= 0;
        const int64_t Kl = (int64_t)((uint64_t)K + (ex[l] - ex0));
        const bool ok = cur[l].E[0] == SS_E_ANY || (plain && valid && cur[l].E[0] == E && ss_range_ok(Kl, cur[l].lo[0], cur[l].hi[0]));
        if (!ok) { f = l; break; }
      }
      if (f > start) {
        const uint64_t upto = f < 64 ? ex[f] : total;
        const uint64_t d = upto - ex0;
        if (valid && d) s = ss_compose(E, (int64_t)((uint64_t)K + d));
        stats[0] += f - start;
      }
      if (f == 64) break;
      if (cur[f].E[0] != SS_E_BAD && ss_apply_record(&s, &cur[f])) {
        stats[1]++;
      } else {
        const int64_t b = c * 64 + f;
        for (int64_t i = b * SS_BLOCK; i < (b + 1) * SS_BLOCK; ++i) s = s + (i < n ? x[i] : 0.0);
        stats[2]++;
      }
      start = f + 1;
    }
  }
  return s;
}

// A census of the blocks of one column (tests/seqsum_census.py): out[0] plain, [1] empty, [2] through a record with splits, [3] record
// refused by the exact state, [4] prepared as "row by row", [5] plain but refused.  verbose: why the row-by-row blocks are what they are.
#include <stdio.h>
// Per column walk: classify blocks. Emulates device prefix: group sums (4 blocks) prefix + in-group run prefix.
extern "C" void ss_host_census(const double* x, int64_t n, int64_t stride, int64_t* out /*[8]*/, int verbose) {
  const int64_t nchunks = (n + (int64_t)SS_BLOCK * 64 - 1) / ((int64_t)SS_BLOCK * 64);
  const int64_t nb = nchunks * 64, nsub = nb * SS_Q;
  std::vector<double> xs(nb * SS_BLOCK, 0.0);
  for (int64_t i = 0; i < n; ++i) xs[i] = x[i * stride];
  std::vector<double> apx(nsub + 1, 0.0);
  for (int64_t q = 0; q < nsub; ++q) { double t = 0; for (int i = 0; i < SS_SUB; ++i) t += xs[q * SS_SUB + i]; apx[q + 1] = apx[q] + t; }
  double s = 0.0;
  // out: 0 plain, 1 any(empty), 2 rec ok, 3 rec failed->rows, 4 BAD->rows, 5 plain-but-mismatch->rows...
  for (int k = 0; k < 8; ++k) out[k] = 0;
  for (int64_t b = 0; b < nb; ++b) {
    SsRec rec;
    const int64_t left = n - b * SS_BLOCK;
    const int len = left <= 0 ? 0 : (left < SS_BLOCK ? (int)left : SS_BLOCK);
    ss_block_record(&xs[b * SS_BLOCK], 1, len, &apx[b * SS_Q], &rec);
    int cls;
    double s0 = s;
    if (rec.E[0] == SS_E_ANY) cls = 1;
    else if (rec.E[0] == SS_E_BAD) cls = 4;
    else if (ss_apply_record(&s, &rec)) cls = rec.nsplit ? 2 : 0;
    else cls = rec.nsplit ? 3 : 5;
    if (cls >= 3) {
      for (int i = 0; i < SS_BLOCK; ++i) s = s + xs[b * SS_BLOCK + i];
      if (verbose) {
        // why BAD: count sub-records bad, splits
        int nbad = 0, nspl = 0;
        for (int q = 0; q < SS_Q; ++q) { SsRec r; ss_sub_record(&xs[b * SS_BLOCK + q * SS_SUB], SS_SUB, apx[b * SS_Q + q], &r); if (r.E[0] == SS_E_BAD) nbad++; else if (r.E[0] >= 0) nspl += r.nsplit; }
        // count binade changes of exact chain inside the block and sign changes
        double t = s0; int nbin = 0;
        for (int i = 0; i < SS_BLOCK; ++i) { double u = t + xs[b * SS_BLOCK + i]; if (ss_expo(u) != ss_expo(t) || (u < 0) != (t < 0)) nbin++; t = u; }
        printf("  blk %lld cls %d s0 %.6e -> %.6e  subBAD %d splits %d binade changes %d  apx %.6e\n", (long long)b, cls, s0, s, nbad, nspl, nbin, apx[b * SS_Q]);
      }
    }
    out[cls]++;
  }
}
