// hawk_hostutil.hip — host-side helpers of the report assembly.  No device code: the guide report's
// `samples` and `haplotype_id` columns list every carrier of every report row (C3: 28 M entries, ~330 MB of text);
// joining them is a ragged byte gather that Python cannot do at memory speed, so it lives here, multi-threaded.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/hawk.h"

namespace {
template <class F> void par_groups(uint64_t n_groups, F f) {
  unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (n_groups < 4096) nt = 1;
  const uint64_t per = (n_groups + nt - 1) / nt;
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t)
    th.emplace_back([=] { f(std::min<uint64_t>(n_groups, t * per), std::min<uint64_t>(n_groups, (t + 1) * per)); });
  f(0, std::min<uint64_t>(n_groups, per));
  for (auto& x : th) x.join();
}
template <class F> void par_groups_any(uint64_t n, F f) {  // the same split for few, heavy items
  const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n, std::min(32u, std::max(1u, std::thread::hardware_concurrency()))));
  const uint64_t per = (n + nt - 1) / nt;
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back([=] { f(std::min<uint64_t>(n, t * per), std::min<uint64_t>(n, (t + 1) * per)); });
  f(0, std::min<uint64_t>(n, per));
  for (auto& x : th) x.join();
}
inline unsigned dec_len(uint32_t v) { unsigned n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
inline uint8_t* put_dec(uint8_t* w, uint32_t v) {
  char tmp[12]; int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *w++ = (uint8_t)tmp[--n];
  return w;
}
}  // namespace

extern "C" {

// For every group g the items item_label[group_off[g] .. group_off[g+1]) name byte strings pool[pool_off[l] .. pool_off[l+1]);
// the group's output is those strings joined by `sep`.  out_off[n_groups + 1] receives the byte offsets of the groups in
// `out` (always written); the bytes are written when `out` is non-NULL and out_cap suffices, else HAWK_E_CAPACITY with
// the required size in out_off[n_groups].
int hawk_host_ragged_join(const uint32_t* item_label, const uint64_t* group_off, uint64_t n_groups, const uint8_t* pool,
                          const uint64_t* pool_off, uint64_t n_labels, uint8_t sep, uint8_t* out, uint64_t out_cap,
                          uint64_t* out_off) {
  if (!group_off || !out_off || (group_off[n_groups] && (!item_label || !pool || !pool_off))) return HAWK_E_INVALID;
  const uint64_t n_items = group_off[n_groups];
  for (uint64_t g = 0; g < n_groups; ++g)
    if (group_off[g + 1] < group_off[g]) return HAWK_E_INVALID;
  unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (n_groups < 4096) nt = 1;
  const uint64_t per = (n_groups + nt - 1) / nt;
  // pass 1: sizes
  std::vector<int> bad(nt, 0);
  auto size_pass = [&](unsigned t) {
    const uint64_t g0 = std::min<uint64_t>(n_groups, t * per), g1 = std::min<uint64_t>(n_groups, g0 + per);
    for (uint64_t g = g0; g < g1; ++g) {
      uint64_t sz = 0;
      for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i) {
        const uint32_t l = item_label[i];
        if (l >= n_labels) { bad[t] = 1; return; }
        sz += pool_off[l + 1] - pool_off[l] + 1;
      }
      out_off[g + 1] = sz ? sz - 1 : 0;  // separators between items only
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(size_pass, t);
    size_pass(0);
    for (auto& x : th) x.join();
  }
  for (int b : bad) if (b) return HAWK_E_INVALID;
  out_off[0] = 0;
  for (uint64_t g = 0; g < n_groups; ++g) out_off[g + 1] += out_off[g];
  (void)n_items;
  if (!out) return HAWK_OK;
  if (out_off[n_groups] > out_cap) return HAWK_E_CAPACITY;
  auto write_pass = [&](unsigned t) {
    const uint64_t g0 = std::min<uint64_t>(n_groups, t * per), g1 = std::min<uint64_t>(n_groups, g0 + per);
    for (uint64_t g = g0; g < g1; ++g) {
      uint8_t* w = out + out_off[g];
      for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i) {
        const uint32_t l = item_label[i];
        const uint64_t n = pool_off[l + 1] - pool_off[l];
        if (i != group_off[g]) *w++ = sep;
        memcpy(w, pool + pool_off[l], n);
        w += n;
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(write_pass, t);
  write_pass(0);
  for (auto& x : th) x.join();
  return HAWK_OK;
}


// Per group: the sorted set of the items its member haplotypes carry, joined by `sep` (collapse_haplotype_ids,
// reports.py:845-857, and the unphased form of collapse_samples).  Haplotype h carries items hap_item[hap_item_off[h] ..
// hap_item_off[h+1]), each an index into the label pool whose order IS the sort order.
int hawk_host_group_join(const uint64_t* member_off, const uint32_t* member_hap, uint64_t n_groups, const uint64_t* hap_item_off,
                         const uint32_t* hap_item, uint64_t n_haps, const uint8_t* pool, const uint64_t* pool_off, uint64_t n_labels,
                         uint8_t sep, uint8_t* out, uint64_t out_cap, uint64_t* out_off) {
  if (!member_off || !out_off || !hap_item_off || (member_off[n_groups] && (!member_hap || !pool_off))) return HAWK_E_INVALID;
  for (uint64_t i = 0; i < member_off[n_groups]; ++i) if (member_hap[i] >= n_haps) return HAWK_E_INVALID;
  for (uint64_t i = 0; i < hap_item_off[n_haps]; ++i) if (hap_item[i] >= n_labels) return HAWK_E_INVALID;
  auto run = [&](bool write) {
    par_groups(n_groups, [&, write](uint64_t g0, uint64_t g1) {
      std::vector<uint32_t> items;
      for (uint64_t g = g0; g < g1; ++g) {
        items.clear();
        bool sorted = true;
        for (uint64_t i = member_off[g]; i < member_off[g + 1]; ++i) {
          const uint32_t h = member_hap[i];
          for (uint64_t k = hap_item_off[h]; k < hap_item_off[h + 1]; ++k) {
            if (!items.empty() && hap_item[k] < items.back()) sorted = false;
            items.push_back(hap_item[k]);
          }
        }
        if (!sorted) std::sort(items.begin(), items.end());
        items.erase(std::unique(items.begin(), items.end()), items.end());
        if (!write) {
          uint64_t sz = 0;
          for (uint32_t l : items) sz += pool_off[l + 1] - pool_off[l] + 1;
          out_off[g + 1] = sz ? sz - 1 : 0;
        } else {
          uint8_t* w = out + out_off[g];
          bool first = true;
          for (uint32_t l : items) {
            if (!first) *w++ = sep;
            first = false;
            const uint64_t n = pool_off[l + 1] - pool_off[l];
            memcpy(w, pool + pool_off[l], n);
            w += n;
          }
        }
      }
    });
  };
  run(false);
  out_off[0] = 0;
  for (uint64_t g = 0; g < n_groups; ++g) out_off[g + 1] += out_off[g];
  if (!out) return HAWK_OK;
  if (out_off[n_groups] > out_cap) return HAWK_E_CAPACITY;
  run(true);
  return HAWK_OK;
}

// The `samples` column of a phased panel (reports.py:767-810: sorted unique `sample:a|b` entries, then one entry per sample
// with the per-copy maxima, samples in the order of their first entry).  Haplotype h carries entries hap_ent[...]; entry e
// belongs to sample ent_sample[e] (samples numbered in that first-entry order) with alleles ent_a1[e] | ent_a2[e];
// ent_ok[e] = 0 marks an entry that is not `name:int|int`.  group_flags[g]: bit 0 = the group holds an ok (phased) entry,
// bit 1 = it holds one that is not.  A group with only such entries gets them joined as they are (entry strings in ent_pool);
// a group mixing both kinds gets an empty string (the caller resolves it).
int hawk_host_group_samples(const uint64_t* member_off, const uint32_t* member_hap, uint64_t n_groups, const uint64_t* hap_ent_off,
                            const uint32_t* hap_ent, uint64_t n_haps, const uint32_t* ent_sample, const uint16_t* ent_a1,
                            const uint16_t* ent_a2, const uint8_t* ent_ok, uint64_t n_entries, const uint8_t* name_pool,
                            const uint64_t* name_off, uint64_t n_samples, const uint8_t* ent_pool, const uint64_t* ent_pool_off,
                            uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint8_t* group_flags) {
  if (!member_off || !out_off || !hap_ent_off || !group_flags || (member_off[n_groups] && (!member_hap || !name_off))) return HAWK_E_INVALID;
  for (uint64_t i = 0; i < member_off[n_groups]; ++i) if (member_hap[i] >= n_haps) return HAWK_E_INVALID;
  for (uint64_t i = 0; i < hap_ent_off[n_haps]; ++i) if (hap_ent[i] >= n_entries) return HAWK_E_INVALID;
  for (uint64_t e = 0; e < n_entries; ++e) if (ent_sample[e] >= n_samples) return HAWK_E_INVALID;
  struct Ent { uint32_t s; uint16_t a, b; };
  auto run = [&](bool write) {
    par_groups(n_groups, [&, write](uint64_t g0, uint64_t g1) {
      std::vector<Ent> v;
      std::vector<uint32_t> raw;
      for (uint64_t g = g0; g < g1; ++g) {
        v.clear();
        raw.clear();
        bool sorted = true;
        uint8_t fl = 0;
        for (uint64_t i = member_off[g]; i < member_off[g + 1]; ++i) {
          const uint32_t h = member_hap[i];
          for (uint64_t k = hap_ent_off[h]; k < hap_ent_off[h + 1]; ++k) {
            const uint32_t e = hap_ent[k];
            fl |= ent_ok[e] ? 1 : 2;
            if (!v.empty() && ent_sample[e] < v.back().s) sorted = false;
            v.push_back(Ent{ent_sample[e], ent_a1[e], ent_a2[e]});
            raw.push_back(e);
          }
        }
        group_flags[g] = fl;
        if (fl == 2) {  // no phased entry at all (e.g. the REF group): the sorted unique entries as they are
          std::sort(raw.begin(), raw.end());
          raw.erase(std::unique(raw.begin(), raw.end()), raw.end());
          if (!write) {
            uint64_t sz = 0;
            for (uint32_t e : raw) sz += ent_pool_off[e + 1] - ent_pool_off[e] + 1;
            out_off[g + 1] = sz ? sz - 1 : 0;
          } else {
            uint8_t* w = out + out_off[g];
            bool first = true;
            for (uint32_t e : raw) {
              if (!first) *w++ = ',';
              first = false;
              const uint64_t len = ent_pool_off[e + 1] - ent_pool_off[e];
              memcpy(w, ent_pool + ent_pool_off[e], len);
              w += len;
            }
          }
          continue;
        }
        if (fl & 2) { if (!write) out_off[g + 1] = 0; continue; }
        if (!sorted) std::stable_sort(v.begin(), v.end(), [](const Ent& x, const Ent& y) { return x.s < y.s; });
        size_t n = 0;  // merge neighbours of one sample: per-copy maxima
        for (size_t i = 0; i < v.size(); ++i) {
          if (n && v[n - 1].s == v[i].s) { v[n - 1].a = std::max(v[n - 1].a, v[i].a); v[n - 1].b = std::max(v[n - 1].b, v[i].b); }
          else v[n++] = v[i];
        }
        if (!write) {
          uint64_t sz = 0;
          for (size_t i = 0; i < n; ++i) sz += name_off[v[i].s + 1] - name_off[v[i].s] + 1 + dec_len(v[i].a) + 1 + dec_len(v[i].b) + 1;
          out_off[g + 1] = sz ? sz - 1 : 0;
        } else {
          uint8_t* w = out + out_off[g];
          for (size_t i = 0; i < n; ++i) {
            if (i) *w++ = ',';
            const uint64_t len = name_off[v[i].s + 1] - name_off[v[i].s];
            memcpy(w, name_pool + name_off[v[i].s], len);
            w += len;
            *w++ = ':';
            w = put_dec(w, v[i].a);
            *w++ = '|';
            w = put_dec(w, v[i].b);
          }
        }
      }
    });
  };
  run(false);
  out_off[0] = 0;
  for (uint64_t g = 0; g < n_groups; ++g) out_off[g + 1] += out_off[g];
  if (!out) return HAWK_OK;
  if (out_off[n_groups] > out_cap) return HAWK_E_CAPACITY;
  run(true);
  return HAWK_OK;
}

// Position-map segments of every row of an expansion (haplotype.py:90-159 as unit-slope segments), from the carried
// indels alone: row r's list entries are [hv_off[r], hv_off[r+1]); indel_entry = the ascending entry indices whose
// variant changes the length (hawk_gt_lists_indels).  A carried deletion opens one segment behind the deleted bases, a
// carried insertion of n bases n + 1 (the inserted bases all map to the anchor position); every row starts with the
// identity segment (rel 0 -> startp); rows that alias another row (collapsed onto it) keep only that one; segments
// starting at or behind the row's end are dropped.  seg_start[n_rows + 1] is always written (so a first call with
// cap = 0 tells the size), seg_rel / seg_gen when they hold seg_start[n_rows] entries.
int hawk_host_build_segments(const uint32_t* indel_entry, uint64_t n_indel, const uint32_t* hv_idx, const int32_t* hv_o,
                             const uint64_t* hv_off, uint32_t n_rows, const int64_t* var_r0, const int64_t* var_chain, int64_t startp,
                             const uint32_t* hap_len, const int64_t* alias, uint64_t* seg_start, uint32_t* seg_rel, int64_t* seg_gen,
                             uint64_t cap) {
  if (!hv_off || !hap_len || !alias || !seg_start || !n_rows || (n_indel && (!indel_entry || !hv_idx || !hv_o || !var_r0 || !var_chain)))
    return HAWK_E_INVALID;
  for (uint64_t e = 1; e < n_indel; ++e)
    if (indel_entry[e] <= indel_entry[e - 1]) return HAWK_E_INVALID;  // ascending: a row's indels are one run of the list
  if (n_indel && indel_entry[n_indel - 1] >= hv_off[n_rows]) return HAWK_E_INVALID;
  // one row's segments: the identity segment, then those of its carried indels (entries [hv_off[r], hv_off[r + 1]) of the
  // list, found by bisection so that rows can be walked by several threads); `at` = where the row's segments go, or
  // nullptr to count them
  auto row = [&](uint32_t r, uint64_t at, bool write) -> uint64_t {
    uint64_t cnt = 1;
    if (write) { seg_rel[at] = 0; seg_gen[at] = startp; }
    if (alias[r] != (int64_t)r) return cnt;
    const uint32_t* lo = std::lower_bound(indel_entry, indel_entry + n_indel, hv_off[r],
                                          [](uint32_t a, uint64_t b) { return (uint64_t)a < b; });
    const int64_t len = (int64_t)hap_len[r];
    for (const uint32_t* p = lo; p < indel_entry + n_indel && *p < hv_off[r + 1]; ++p) {
      const uint32_t v = hv_idx[*p];
      const int64_t o = hv_o[*p], pos = var_r0[v] + startp, ch = var_chain[v];
      const int64_t nseg = ch < 0 ? 1 : ch + 1;
      for (int64_t k = 0; k < nseg; ++k) {
        const int64_t rel = o + 1 + k;
        if (rel >= len) break;
        if (write) {
          seg_rel[at + cnt] = (uint32_t)rel;
          seg_gen[at + cnt] = ch < 0 ? pos + 1 - ch : (k < ch ? pos : pos + 1);
        }
        ++cnt;
      }
    }
    return cnt;
  };
  par_groups(n_rows, [&](uint64_t a, uint64_t b) { for (uint64_t r = a; r < b; ++r) seg_start[r + 1] = row((uint32_t)r, 0, false); });
  seg_start[0] = 0;
  for (uint32_t r = 0; r < n_rows; ++r) seg_start[r + 1] += seg_start[r];
  if (!seg_rel || !seg_gen || cap < seg_start[n_rows]) return cap ? HAWK_E_CAPACITY : HAWK_OK;
  par_groups(n_rows, [&](uint64_t a, uint64_t b) { for (uint64_t r = a; r < b; ++r) (void)row((uint32_t)r, seg_start[r], true); });
  return HAWK_OK;
}

// posmap_rev[g] of every row (the reference rebuilds the reverse dict by overwrite, haplotype.py:159): the LAST relative
// position whose genomic position is g, -1 where g is deleted from the row or outside it.  Segments as built above.
int hawk_host_posmap_rev(const uint64_t* seg_start, const uint32_t* seg_rel, const int64_t* seg_gen, const uint32_t* hap_len,
                         uint32_t n_rows, int64_t g, int64_t* out) {
  if (!seg_start || !seg_rel || !seg_gen || !hap_len || !out) return HAWK_E_INVALID;
  for (uint32_t r = 0; r < n_rows; ++r) {
    int64_t best = -1;
    for (uint64_t k = seg_start[r]; k < seg_start[r + 1]; ++k) {
      const int64_t end = k + 1 < seg_start[r + 1] ? (int64_t)seg_rel[k + 1] : (int64_t)hap_len[r];
      const int64_t last_gen = seg_gen[k] + (end - (int64_t)seg_rel[k]) - 1;
      if (seg_gen[k] <= g && g <= last_gen) best = (int64_t)seg_rel[k] + (g - seg_gen[k]);  // later segments overwrite
    }
    out[r] = best;
  }
  return HAWK_OK;
}

}  // extern "C"

// ---- the guide report as TSV text, written straight to its file ---------------------------------------------------------
// The report of a C3 search is 2.2 x 10^5 rows and 0.64 GB of text (the `samples` / `haplotype_id` columns list every carrier of
// every row).  Python needs seconds to turn columns into row strings and the rows into a file; here the columns arrive as they
// are kept - constants, fixed-width byte matrices, ragged byte columns, integers, vocabulary indices - and the rows are
// laid out in two parallel passes (lengths, then bytes) directly in a shared mapping of the output file.
namespace {
inline unsigned dec_len64(int64_t v) {
  unsigned n = v < 0 ? 1u : 0u;
  uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
  do { u /= 10; ++n; } while (u);
  return n;
}
inline uint8_t* put_dec64(uint8_t* w, int64_t v) {
  char tmp[24]; int n = 0;
  uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
  do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *w++ = '-';
  while (n) *w++ = (uint8_t)tmp[--n];
  return w;
}
inline uint64_t field_len(const hawk_tsv_col& c, uint64_t r) {
  switch (c.kind) {
    case HAWK_TSV_CONST: return c.width;
    case HAWK_TSV_FIXED: return c.width;
    case HAWK_TSV_RAGGED: return c.off[r + 1] - c.off[r];
    case HAWK_TSV_INT64: return dec_len64(static_cast<const int64_t*>(c.data)[r]);
    default: { const uint32_t k = static_cast<const uint32_t*>(c.data)[r]; return c.off[k + 1] - c.off[k]; }
  }
}
inline uint8_t* field_put(const hawk_tsv_col& c, uint64_t r, uint8_t* w) {
  switch (c.kind) {
    case HAWK_TSV_CONST: memcpy(w, c.data, c.width); return w + c.width;
    case HAWK_TSV_FIXED: memcpy(w, static_cast<const uint8_t*>(c.data) + r * c.width, c.width); return w + c.width;
    case HAWK_TSV_RAGGED: { const uint64_t n = c.off[r + 1] - c.off[r]; memcpy(w, static_cast<const uint8_t*>(c.data) + c.off[r], n); return w + n; }
    case HAWK_TSV_INT64: return put_dec64(w, static_cast<const int64_t*>(c.data)[r]);
    default: {
      const uint32_t k = static_cast<const uint32_t*>(c.data)[r];
      const uint64_t n = c.off[k + 1] - c.off[k];
      memcpy(w, static_cast<const uint8_t*>(c.pool) + c.off[k], n);
      return w + n;
    }
  }
}
}  // namespace
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
extern "C" {

int hawk_host_tsv_write(const char* path, const char* header, uint64_t header_len, uint64_t n_rows, const uint64_t* order, uint32_t n_cols,
                        const hawk_tsv_col* cols, uint64_t* bytes_out) {
  if (!path || !n_cols || !cols || (header_len && !header)) return HAWK_E_INVALID;
  for (uint32_t c = 0; c < n_cols; ++c) {
    const hawk_tsv_col& k = cols[c];
    if (k.kind > HAWK_TSV_VOCAB) return HAWK_E_INVALID;
    if (n_rows && k.kind != HAWK_TSV_CONST && !k.data) return HAWK_E_INVALID;
    if ((k.kind == HAWK_TSV_RAGGED || k.kind == HAWK_TSV_VOCAB) && !k.off) return HAWK_E_INVALID;
    if (k.kind == HAWK_TSV_VOCAB && !k.pool && k.n_vocab) return HAWK_E_INVALID;
    if (k.kind == HAWK_TSV_CONST && k.width && !k.data) return HAWK_E_INVALID;
  }
  if (order) for (uint64_t i = 0; i < n_rows; ++i) if (order[i] >= n_rows) return HAWK_E_INVALID;
  for (uint32_t c = 0; c < n_cols; ++c)
    if (cols[c].kind == HAWK_TSV_VOCAB) {
      const uint32_t* idx = static_cast<const uint32_t*>(cols[c].data);
      for (uint64_t i = 0; i < n_rows; ++i) if (idx[i] >= cols[c].n_vocab) return HAWK_E_INVALID;
    }
  // pass 1: bytes of every output row (fields + tabs + newline)
  std::vector<uint64_t> row_off(n_rows + 1, 0);
  par_groups(n_rows, [&](uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; ++i) {
      const uint64_t r = order ? order[i] : i;
      uint64_t sz = n_cols;  // n_cols - 1 tabs + the newline
      for (uint32_t c = 0; c < n_cols; ++c) sz += field_len(cols[c], r);
      row_off[i + 1] = sz;
    }
  });
  row_off[0] = header_len;
  for (uint64_t i = 0; i < n_rows; ++i) row_off[i + 1] += row_off[i];
  const uint64_t total = row_off[n_rows];
  if (bytes_out) *bytes_out = total;
  const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return HAWK_E_INVALID;
  if (total == 0) { close(fd); return HAWK_OK; }
  // The rows are laid out in anonymous memory (huge pages where the system grants them: first-touch faults of 4 KB pages were
  // a third of the pass) and leave through parallel pwrite()s.  Measured on the GPU box's overlay file system, 0.63 GB of report:
  // a shared mapping of the file 0.47 s, this 0.22 s, the layout alone 0.16 s (tools/_tsv_probe.py, round 4).
  uint8_t* out = static_cast<uint8_t*>(mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (out == MAP_FAILED) { close(fd); return HAWK_E_INVALID; }
#ifdef MADV_HUGEPAGE
  (void)madvise(out, total, MADV_HUGEPAGE);
#endif
  if (header_len) memcpy(out, header, header_len);
  par_groups(n_rows, [&](uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; ++i) {
      const uint64_t r = order ? order[i] : i;
      uint8_t* w = out + row_off[i];
      for (uint32_t c = 0; c < n_cols; ++c) {
        w = field_put(cols[c], r, w);
        *w++ = c + 1 == n_cols ? '\n' : '\t';
      }
    }
  });
  int rc = HAWK_OK;
  const uint64_t n_chunks = (total + (1u << 22) - 1) >> 22;  // 4 MB pieces
  std::vector<int> bad(1, 0);
  auto put = [&](uint64_t c0, uint64_t c1) {
    uint64_t done = c0 << 22;
    const uint64_t end = std::min<uint64_t>(total, c1 << 22);
    while (done < end) {
      const ssize_t k = pwrite(fd, out + done, end - done, (off_t)done);
      if (k <= 0) { bad[0] = 1; return; }
      done += (uint64_t)k;
    }
  };
  if (n_chunks >= 16) par_groups_any(n_chunks, put); else put(0, n_chunks);
  if (bad[0]) rc = HAWK_E_INVALID;
  munmap(out, total);
  if (close(fd) != 0) rc = HAWK_E_INVALID;
  return rc;
}

// ---- the line index of a VCF text (f3: readers.VCF) ------------------------------------------------------------------------
// A 2504-sample VCF is ~10 kB per record; the reader's index pass (where does every record start, what is its POS, where do its
// sample columns begin) walks every byte of it once.  numpy did that at ~1.5 GB/s on one core; here the text - the reader maps the
// file - is cut into pieces, one thread each: newlines counted, then every line's fields located.
//   line_start[i]   offset of line i (header lines included), line_start[n_lines] = end of the last line (= len)
//   pos[i]          POS of a record, -1 for a header line ('#'), -2 for a malformed record (no second tab / non-digit POS)
//   gt_off[i]       offset of the record's first sample column (behind its 9th tab), 0 when the line has fewer than 9 tabs
//   chrom_len[i]    bytes of the record's CHROM field
// *n_lines: lines found (the text must end with a newline); with cap too small nothing but *n_lines is written and
// HAWK_E_CAPACITY is returned.  *multi_contig: 1 when two records differ in CHROM.
int hawk_host_vcf_index(const uint8_t* text, uint64_t len, uint64_t cap, uint64_t* line_start, int64_t* pos, uint64_t* gt_off,
                        uint32_t* chrom_len, uint64_t* n_lines, uint32_t* multi_contig) {
  if (!n_lines || (len && !text)) return HAWK_E_INVALID;
  if (len && text[len - 1] != '\n') return HAWK_E_INVALID;
  unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (len < (1u << 22)) nt = 1;
  const uint64_t per = (len + nt - 1) / nt;
  std::vector<uint64_t> cnt(nt + 1, 0);
  auto count = [&](unsigned t) {
    const uint64_t a = std::min<uint64_t>(len, t * per), b = std::min<uint64_t>(len, a + per);
    uint64_t c = 0;
    const uint8_t* p = text + a;
    const uint8_t* e = text + b;
    while (p < e) { const void* q = memchr(p, '\n', (size_t)(e - p)); if (!q) break; ++c; p = static_cast<const uint8_t*>(q) + 1; }
    cnt[t + 1] = c;
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(count, t);
    count(0);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t) cnt[t + 1] += cnt[t];
  const uint64_t n = cnt[nt];
  *n_lines = n;
  if (n > cap) return HAWK_E_CAPACITY;
  if (!n) return HAWK_OK;
  if (!line_start || !pos || !gt_off || !chrom_len) return HAWK_E_INVALID;
  auto fill = [&](unsigned t) {  // line k + 1 starts behind the k-th newline
    const uint64_t a = std::min<uint64_t>(len, t * per), b = std::min<uint64_t>(len, a + per);
    uint64_t k = cnt[t];
    const uint8_t* p = text + a;
    const uint8_t* e = text + b;
    while (p < e) {
      const void* q = memchr(p, '\n', (size_t)(e - p));
      if (!q) break;
      p = static_cast<const uint8_t*>(q) + 1;
      line_start[++k] = (uint64_t)(p - text);
    }
  };
  line_start[0] = 0;
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(fill, t);
    fill(0);
    for (auto& x : th) x.join();
  }
  par_groups(n, [&](uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; ++i) {
      const uint8_t* p = text + line_start[i];
      const uint8_t* e = text + line_start[i + 1] - 1;  // the newline
      gt_off[i] = 0; chrom_len[i] = 0;
      if (p == e) { pos[i] = -1; continue; }        // empty line: skipped like a header
      if (*p == '#') { pos[i] = -1; continue; }
      const uint8_t* t1 = static_cast<const uint8_t*>(memchr(p, '\t', (size_t)(e - p)));
      const uint8_t* t2 = t1 ? static_cast<const uint8_t*>(memchr(t1 + 1, '\t', (size_t)(e - t1 - 1))) : nullptr;
      if (!t1 || !t2 || t2 == t1 + 1 || t2 - t1 > 19) { pos[i] = -2; continue; }
      int64_t v = 0;
      bool ok = true;
      for (const uint8_t* q = t1 + 1; q < t2; ++q) { if (*q < '0' || *q > '9') { ok = false; break; } v = v * 10 + (*q - '0'); }
      if (!ok) { pos[i] = -2; continue; }
      pos[i] = v;
      chrom_len[i] = (uint32_t)(t1 - p);
      const uint8_t* q = t2;  // the 2nd tab; seven more end the FORMAT column
      int tabs = 2;
      while (tabs < 9) {
        q = static_cast<const uint8_t*>(memchr(q + 1, '\t', (size_t)(e - q - 1)));
        if (!q) break;
        ++tabs;
      }
      if (q && tabs == 9) gt_off[i] = (uint64_t)(q + 1 - text);
    }
  });
  if (multi_contig) {
    *multi_contig = 0;
    uint64_t f = n;
    for (uint64_t i = 0; i < n; ++i) if (pos[i] >= 0) { f = i; break; }
    if (f < n) {
      const uint8_t* c0 = text + line_start[f];
      const uint32_t l0 = chrom_len[f];
      par_groups(n, [&](uint64_t i0, uint64_t i1) {
        for (uint64_t i = i0; i < i1; ++i)
          if (pos[i] >= 0 && (chrom_len[i] != l0 || memcmp(text + line_start[i], c0, l0) != 0)) { *multi_contig = 1; return; }
      });
    }
  }
  return HAWK_OK;
}


// ---- which of a haplotype's variants a guide shows (annotation.py:246-284, polish_guide_variants) ----------------------------
// The report's variant_id / af columns: for a row whose candidate variants are all SNVs the assembly answers in bulk (numpy); a
// row with an indel among them - or a position map that is not linear over the guide - needs the reference's own walk over the
// guide's positions, which took 25 us of Python per row (C3: 10^4 rows, a quarter of a second).  Here the walk itself, for all
// such rows at once:
//   row k: spacer + PAM `cores[k]` (L cased bytes, + strand), haplotype row hap[k] whose position map is segments
//   [seg_start[h], seg_start[h + 1]) of (seg_rel, seg_gen), first guide position pivot[k], genomic `stop[k]`, candidates
//   cand_var[cand_off[k] .. cand_off[k + 1]) - variant indices, ascending, no duplicates.
//   variant v: adjusted position t_pos[v], alleles ref / alt (pools), name_rank[v] = rank of its id in string order.
// Output: out_var[out_off[k] .. out_off[k + 1]) = the variants the guide shows, in id order (out_var holds cand_off[n] entries at
// most); need_python[k] = 1 where the reference's code would trip its own assertion (annotation.py:185-189) - the caller runs
// its Python mirror on those rows, which raises as the reference does.
namespace {
inline bool is_upper(uint8_t c) { return c >= 'A' && c <= 'Z'; }
inline bool is_lower(uint8_t c) { return c >= 'a' && c <= 'z'; }
inline uint8_t to_upper(uint8_t c) { return is_lower(c) ? (uint8_t)(c - 32) : c; }
}  // namespace
namespace {
struct PolishTables {
  uint32_t L;
  const uint64_t* seg_start; const uint32_t* seg_rel; const int64_t* seg_gen;
  const int64_t* t_pos; const uint8_t* ref_pool; const uint64_t* ref_off; const uint8_t* alt_pool; const uint64_t* alt_off;
  const uint32_t* name_rank;
};
// one row of annotation.polish_guide_variants: the candidates `cand` (ascending variant indices, no duplicates) the guide g shows,
// written to w in id order; returns their number, or -1 where the reference's own assertion would fire
inline int polish_one(const PolishTables& T, const uint8_t* g, uint32_t hap, int64_t pivot, int64_t stop, const uint32_t* cand, uint32_t n_cand,
                      uint32_t* w) {
  const uint32_t L = T.L;
  int64_t gen[64];
  // genomic position of every guide position: PosSegments.lookup (last segment with rel <= p)
  const uint64_t s0 = T.seg_start[hap], s1 = T.seg_start[hap + 1];
  uint64_t j = s0;
  {
    uint64_t lo = s0, hi = s1;  // last j in [s0, s1) with seg_rel[j] <= pivot (seg_rel[s0] = 0)
    const int64_t p0 = pivot < 0 ? 0 : pivot;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if ((int64_t)T.seg_rel[mid] <= p0) lo = mid; else hi = mid; }
    j = lo;
  }
  for (uint32_t i = 0; i < L; ++i) {
    const int64_t p = pivot + i;
    while (j + 1 < s1 && (int64_t)T.seg_rel[j + 1] <= p) ++j;
    gen[i] = T.seg_gen[j] + (p - (int64_t)T.seg_rel[j]);
  }
  uint32_t m = 0;
  for (uint32_t i = 0; i < L; ++i) {
    uint32_t offset = 0;
    for (uint32_t c = 0; c < n_cand; ++c) {
      const uint32_t v = cand[c];
      if (T.t_pos[v] != gen[i]) continue;
      const uint64_t rl = T.ref_off[v + 1] - T.ref_off[v], al = T.alt_off[v + 1] - T.alt_off[v];
      const uint8_t* alt = T.alt_pool + T.alt_off[v];
      const bool is_snv = rl == al;
      if (!is_snv) offset = rl < al ? (uint32_t)(al - rl) : 0u;
      const uint32_t sl = std::min<uint32_t>(offset + 1, L - i);  // seg = guidepam[i : i + offset + 1]
      const uint8_t* seg = g + i;
      bool show = false;
      if (!is_snv) {  // _check_insertion
        if (i == 0) {
          // _find_insertion_stop asserts a lower-case first base and at least one base that is not upper case
          bool all_up = true;
          for (uint32_t q = 0; q < sl; ++q) all_up = all_up && is_upper(seg[q]);
          if (is_upper(seg[0]) || all_up) return -1;
          uint32_t st = 0;
          for (uint32_t q = 0; q < sl; ++q) if (is_upper(seg[q])) { st = q; break; }
          bool ends = st <= al;  // alt.endswith(seg.upper()[:st])
          for (uint32_t q = 0; q < st && ends; ++q) ends = alt[al - st + q] == to_upper(seg[q]);
          show = ends;
        }
        if (!show && gen[i] == stop && sl <= al) {  // alt.startswith(seg.upper())
          bool starts = true;
          for (uint32_t q = 0; q < sl && starts; ++q) starts = alt[q] == to_upper(seg[q]);
          show = starts;
        }
      }
      if (!show) {  // seg.islower() and seg.upper() == alt
        bool low = true;
        for (uint32_t q = 0; q < sl; ++q) low = low && is_lower(seg[q]);
        if (low && sl == al) {
          bool eq = true;
          for (uint32_t q = 0; q < sl && eq; ++q) eq = alt[q] == to_upper(seg[q]);
          show = eq;
        }
      }
      if (show) {
        bool dup = false;
        for (uint32_t q = 0; q < m; ++q) dup = dup || w[q] == v;
        if (!dup) w[m++] = v;
      }
    }
  }
  std::sort(w, w + m, [&](uint32_t x, uint32_t y) { return T.name_rank[x] < T.name_rank[y]; });
  return (int)m;
}
// out_var[off[k] ..) -> out_var[out_off[k] ..), in place (rows only move towards the front)
inline void polish_compact(uint64_t n, const uint64_t* off, const std::vector<uint32_t>& cnt, uint64_t* out_off, uint32_t* out_var) {
  out_off[0] = 0;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t src = off[k], dst = out_off[k];
    if (dst != src) memmove(out_var + dst, out_var + src, (size_t)cnt[k] * 4);
    out_off[k + 1] = dst + cnt[k];
  }
}
}  // namespace
int hawk_host_polish_rows(uint64_t n, uint32_t L, const uint8_t* cores, const uint32_t* hap, const int64_t* pivot, const int64_t* stop,
                          const uint64_t* cand_off, const uint32_t* cand_var, const uint64_t* seg_start, const uint32_t* seg_rel,
                          const int64_t* seg_gen, uint64_t n_haps, const int64_t* t_pos, const uint8_t* ref_pool, const uint64_t* ref_off,
                          const uint8_t* alt_pool, const uint64_t* alt_off, uint32_t n_var, const uint32_t* name_rank, uint64_t* out_off,
                          uint32_t* out_var, uint8_t* need_python) {
  if (!n) { if (out_off) out_off[0] = 0; return HAWK_OK; }
  if (!cores || !hap || !pivot || !stop || !cand_off || !seg_start || !seg_rel || !seg_gen || !t_pos || !ref_off || !alt_off || !name_rank ||
      !out_off || !out_var || !need_python || !L || L > 64)
    return HAWK_E_INVALID;
  for (uint64_t k = 0; k < n; ++k) if (hap[k] >= n_haps || cand_off[k + 1] < cand_off[k]) return HAWK_E_INVALID;
  for (uint64_t i = 0; i < cand_off[n]; ++i) if (cand_var[i] >= n_var) return HAWK_E_INVALID;
  const PolishTables T{L, seg_start, seg_rel, seg_gen, t_pos, ref_pool, ref_off, alt_pool, alt_off, name_rank};
  std::vector<uint32_t> cnt(n, 0);
  par_groups(n, [&](uint64_t k0, uint64_t k1) {
    for (uint64_t k = k0; k < k1; ++k) {
      const int m = polish_one(T, cores + k * L, hap[k], pivot[k], stop[k], cand_var + cand_off[k], (uint32_t)(cand_off[k + 1] - cand_off[k]),
                               out_var + cand_off[k]);
      need_python[k] = m < 0;
      cnt[k] = m < 0 ? 0u : (uint32_t)m;
    }
  });
  polish_compact(n, cand_off, cnt, out_off, out_var);
  return HAWK_OK;
}

int hawk_host_variant_window(uint64_t n, const uint32_t* hap, const int64_t* p_lo, const int64_t* p_hi, const uint64_t* var_off,
                             const int64_t* var_idx, const int64_t* t_pos, uint64_t n_haps, uint32_t n_var, uint64_t* first, uint32_t* count) {
  if (!n) return HAWK_OK;
  if (!hap || !p_lo || !p_hi || !var_off || !var_idx || !t_pos || !first || !count) return HAWK_E_INVALID;
  std::atomic<int> bad{0};
  par_groups(n_haps, [&](uint64_t h0, uint64_t h1) {  // every row's list: valid indices, ascending positions
    for (uint64_t h = h0; h < h1 && !bad.load(std::memory_order_relaxed); ++h) {
      if (var_off[h + 1] < var_off[h]) { bad = 1; break; }
      int64_t prev = INT64_MIN;
      for (uint64_t k = var_off[h]; k < var_off[h + 1]; ++k) {
        const int64_t v = var_idx[k];
        if (v < 0 || v >= (int64_t)n_var) { bad = 1; break; }
        if (t_pos[v] < prev) { bad = 2; break; }
        prev = t_pos[v];
      }
    }
  });
  if (bad.load() == 1) return HAWK_E_INVALID;
  if (bad.load() == 2) return HAWK_E_UNSUPPORTED;  // (the caller sorts, or takes its own route)
  for (uint64_t k = 0; k < n; ++k) if (hap[k] >= n_haps) return HAWK_E_INVALID;
  par_groups(n, [&](uint64_t k0, uint64_t k1) {
    for (uint64_t k = k0; k < k1; ++k) {
      const uint64_t b0 = var_off[hap[k]], b1 = var_off[hap[k] + 1];
      uint64_t lo = b0, hi = b1;  // first entry with position >= p_lo
      while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (t_pos[var_idx[mid]] < p_lo[k]) lo = mid + 1; else hi = mid; }
      const uint64_t a = lo;
      hi = b1;                    // first entry with position > p_hi
      while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (t_pos[var_idx[mid]] <= p_hi[k]) lo = mid + 1; else hi = mid; }
      first[k] = a;
      count[k] = (uint32_t)(lo - a);
    }
  });
  return HAWK_OK;
}

int hawk_host_polish_windows(uint64_t n, uint32_t L, const uint8_t* cores, const uint32_t* hap, const int64_t* pivot, const int64_t* stop,
                             const uint64_t* first, const uint64_t* cand_off, const int64_t* var_idx, uint64_t n_idx, const uint64_t* seg_start,
                             const uint32_t* seg_rel, const int64_t* seg_gen, uint64_t n_haps, const int64_t* t_pos, const uint8_t* ref_pool,
                             const uint64_t* ref_off, const uint8_t* alt_pool, const uint64_t* alt_off, uint32_t n_var, const uint32_t* name_rank,
                             uint64_t* out_off, uint32_t* out_var, uint8_t* need_python) {
  if (!n) { if (out_off) out_off[0] = 0; return HAWK_OK; }
  if (!cores || !hap || !pivot || !stop || !first || !cand_off || !var_idx || !seg_start || !seg_rel || !seg_gen || !t_pos || !ref_off || !alt_off ||
      !name_rank || !out_off || !out_var || !need_python || !L || L > 64)
    return HAWK_E_INVALID;
  for (uint64_t k = 0; k < n; ++k)
    if (hap[k] >= n_haps || cand_off[k + 1] < cand_off[k] || first[k] + (cand_off[k + 1] - cand_off[k]) > n_idx) return HAWK_E_INVALID;
  const PolishTables T{L, seg_start, seg_rel, seg_gen, t_pos, ref_pool, ref_off, alt_pool, alt_off, name_rank};
  std::vector<uint32_t> cnt(n, 0);
  std::atomic<int> bad{0};
  par_groups(n, [&](uint64_t k0, uint64_t k1) {
    std::vector<uint32_t> cand;
    for (uint64_t k = k0; k < k1; ++k) {
      const uint64_t nc = cand_off[k + 1] - cand_off[k];
      cand.resize(nc);
      for (uint64_t c = 0; c < nc; ++c) {
        const int64_t v = var_idx[first[k] + c];
        if (v < 0 || v >= (int64_t)n_var) { bad = 1; cand[c] = 0; } else cand[c] = (uint32_t)v;
      }
      std::sort(cand.begin(), cand.end());
      const uint32_t nu = (uint32_t)(std::unique(cand.begin(), cand.end()) - cand.begin());
      const int m = polish_one(T, cores + k * L, hap[k], pivot[k], stop[k], cand.data(), nu, out_var + cand_off[k]);
      need_python[k] = m < 0;
      cnt[k] = m < 0 ? 0u : (uint32_t)m;
    }
  });
  if (bad.load()) return HAWK_E_INVALID;
  polish_compact(n, cand_off, cnt, out_off, out_var);
  return HAWK_OK;
}

}  // extern "C"
