// hawk_api_offtarget.hip - C ABI: the off-target scan (K7)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "hawk_host.h"

extern "C" {

// ---------------------------------------------------------------------------- K7 off-targets
int hawk_genome_finalize(hawk_hapset* rows) {
  if (!rows || rows->vplan) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(rows->ctx->device));
  hawk_launch_ot_onehot(rows->ctx->stream, rows->plane, (uint64_t)rows->n_hap * rows->S);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(rows->ctx->stream));
  return HAWK_OK;
}

int hawk_offtarget_scan(hawk_hapset* hs, const hawk_ot_params* p, const uint64_t* guides2, uint32_t n_guides,
                        uint32_t* out_guide, uint32_t* out_row, uint32_t* out_q, uint8_t* out_strand, uint8_t* out_mm,
                        uint64_t* out_code, uint32_t* out_nmask, uint64_t cap, uint64_t* n_out, hawk_ot_timing* timing) {
  if (!hs || !p || !hs->has_meta || !n_out || (n_guides && !guides2)) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  if (p->guidelen + p->pamlen > 32 || p->guidelen == 0) return HAWK_E_UNSUPPORTED;  // window code = 2 bits x 32
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, p->pam_fwd, p->pam_rev, p->pamlen, p->guidelen, p->right, false, &sp);
  if (rc) return rc;
  // windows are indexed by their start q: strand 0 stores the + strand as the guide reads it,
  // strand 1 the mirror image (same convention as the search, search_guides.py:538)
  sp.poF = p->right ? 0 : (int32_t)p->guidelen;
  sp.poR = p->right ? (int32_t)p->guidelen : 0;
  const HapSetDev d = make_dev(hs);
  const size_t words = (size_t)hs->n_hap * hs->S;
  const uint64_t ncnt = (uint64_t)hs->n_hap * 2 * sp.bph;
  if ((rc = hs->keepF.reserve(words * 4)) || (rc = hs->keepR.reserve(words * 4)) || (rc = hs->counts.reserve(ncnt * 4)) ||
      (rc = hs->offsets.reserve((ncnt + 1) * 8)) || (rc = hs->totals.reserve(sizeof(ScanTotals))) ||
      (rc = hs->partial.reserve((ncnt / 1024 + 2) * 8)) || (rc = hs->misc.reserve(512 * 8 + 64)) ||
      (rc = hs->guides.reserve(std::max<size_t>((size_t)n_guides * 8, 16))))
    return rc;
  // Pigeonhole seeds when they pay: enough guides to bucket, and blocks of at least two bases.  HAWK_OT_ALLPAIRS=1
  // keeps the all-pairs kernel (A/B measurements, and the parity test runs both).
  static const bool force_allpairs = [] { const char* e = getenv("HAWK_OT_ALLPAIRS"); return e && e[0] == '1'; }();
  const int G = (int)p->guidelen, nb = (int)p->max_mm + 1;
  const bool seeded = !force_allpairs && n_guides >= 64 && nb <= OT_MAX_BLOCKS && nb * 2 <= G;
  OtSeeds sd;
  memset(&sd, 0, sizeof(sd));
  // LDS variant: guides in chunks of OT_LDS_CHUNK, 4 key bases per block (nb * 8.5 KB of LDS must leave room for a few
  // workgroups per CU); HAWK_OT_SEED_GLOBAL=1 keeps the single-table global-gather kernel
  static const bool seed_global = [] { const char* e = getenv("HAWK_OT_SEED_GLOBAL"); return e && e[0] == '1'; }();
  // Pair seeds (max_mm + 2 blocks, buckets per pair of blocks: hawk_offtarget.hip k_ot_match_pairs) are the default where they
  // apply: HAWK_OT_PAIRS=0 keeps the single-block seeds (the parity tests run every kernel against the brute force)
  const char* e_pairs = getenv("HAWK_OT_PAIRS");
  const int nb2 = (int)p->max_mm + 2;
  const bool pairs = seeded && !seed_global && !(e_pairs && e_pairs[0] == '0') && nb2 <= OT_MAX_BLOCKS && nb2 <= G;
  const bool seed_lds = seeded && !pairs && !seed_global && nb <= 6;
  const uint32_t chunk = seed_lds ? OT_LDS_CHUNK : n_guides;
  const uint32_t n_chunks = seeded ? (n_guides + chunk - 1) / chunk : 0;
  OtPairSeeds ps;
  memset(&ps, 0, sizeof(ps));
  if (pairs) {
    ps.nb = nb2;
    int startb = 0;
    for (int b = 0; b < nb2; ++b) {  // blocks of G / nb2 bases (the first G % nb2 one longer); key = a block's first <= 4 bases
      const int len = G / nb2 + (b < G % nb2 ? 1 : 0), kl = std::min(len, 4);
      ps.start[b] = startb; ps.klen[b] = kl;
      for (int t = 0; t < kl; ++t) ps.pmask2[b] |= 1ull << (2 * (startb + t));
      startb += len;
    }
    for (int bi = 0; bi < nb2; ++bi)
      for (int bj = bi + 1; bj < nb2; ++bj) { ps.pi[ps.n_pairs] = (uint8_t)bi; ps.pj[ps.n_pairs] = (uint8_t)bj; ++ps.n_pairs; }
    std::vector<uint32_t> goff;
    std::vector<uint64_t> gcode((size_t)ps.n_pairs * n_guides, 0);
    std::vector<uint32_t> gid((size_t)ps.n_pairs * n_guides, 0);
    std::vector<uint32_t> keys(n_guides);
    for (int q = 0; q < ps.n_pairs; ++q) {
      const int bi = ps.pi[q], bj = ps.pj[q];
      const uint32_t mi = (1u << (2 * ps.klen[bi])) - 1u, mj = (1u << (2 * ps.klen[bj])) - 1u;
      const uint32_t nkeys = 1u << (2 * (ps.klen[bi] + ps.klen[bj]));
      ps.off_base[q] = (uint32_t)goff.size();
      std::vector<uint32_t> cnt(nkeys + 1, 0);
      for (uint32_t g = 0; g < n_guides; ++g) {
        keys[g] = ((uint32_t)(guides2[g] >> (2 * ps.start[bi])) & mi) | (((uint32_t)(guides2[g] >> (2 * ps.start[bj])) & mj) << (2 * ps.klen[bi]));
        ++cnt[keys[g] + 1];
      }
      for (uint32_t v = 0; v < nkeys; ++v) cnt[v + 1] += cnt[v];
      goff.insert(goff.end(), cnt.begin(), cnt.end());
      std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
      for (uint32_t g = 0; g < n_guides; ++g) {  // counting sort: guides of one bucket stay in input order
        const uint32_t slot = cur[keys[g]]++;
        gcode[(size_t)q * n_guides + slot] = guides2[g];
        gid[(size_t)q * n_guides + slot] = g;
      }
    }
    if ((rc = hs->otoff.reserve(goff.size() * 4)) || (rc = hs->otcode.reserve(gcode.size() * 8)) || (rc = hs->otid.reserve(gid.size() * 4)))
      return rc;
    HIPCHK(hipMemcpyAsync(hs->otoff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otcode.p, gcode.data(), gcode.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otid.p, gid.data(), gid.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
  } else if (seeded) {
    sd.nb = nb;
    const int kmax = seed_lds ? 4 : 6;
    int startb = 0;
    for (int b = 0; b < nb; ++b) {
      const int len = G / nb + (b < G % nb ? 1 : 0), kl = std::min(len, kmax);
      sd.start[b] = startb; sd.klen[b] = kl;
      for (int t = 0; t < kl; ++t) sd.pmask2[b] |= 1ull << (2 * (startb + t));
      startb += len;
    }
    // tables per (chunk, block): bucket offsets (inside the chunk), codes and guide ids in bucket order
    std::vector<uint32_t> goff;
    std::vector<uint64_t> gcode((size_t)n_chunks * nb * chunk, 0);
    std::vector<uint32_t> gid((size_t)n_chunks * nb * chunk, 0);
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const uint32_t g0 = c * chunk, ng = std::min<uint32_t>(chunk, n_guides - g0);
      for (int b = 0; b < nb; ++b) {
        const uint32_t nkeys = seed_lds ? OT_LDS_KEYS : (1u << (2 * sd.klen[b])), kmask = (1u << (2 * sd.klen[b])) - 1u;
        if (c == 0) sd.off_base[b] = (uint32_t)goff.size();  // global variant: one chunk, per-block table sizes differ
        std::vector<uint32_t> cnt(nkeys + 1, 0);
        for (uint32_t g = 0; g < ng; ++g) ++cnt[((uint32_t)(guides2[g0 + g] >> (2 * sd.start[b])) & kmask) + 1];
        for (uint32_t v = 0; v < nkeys; ++v) cnt[v + 1] += cnt[v];
        goff.insert(goff.end(), cnt.begin(), cnt.end());
        std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
        const size_t base = ((size_t)c * nb + b) * chunk;
        for (uint32_t g = 0; g < ng; ++g) {  // counting sort: guides of one bucket stay in input order
          const uint32_t slot = cur[(uint32_t)(guides2[g0 + g] >> (2 * sd.start[b])) & kmask]++;
          gcode[base + slot] = guides2[g0 + g];
          gid[base + slot] = g0 + g;
        }
      }
    }
    if ((rc = hs->otoff.reserve(goff.size() * 4)) || (rc = hs->otcode.reserve(gcode.size() * 8)) || (rc = hs->otid.reserve(gid.size() * 4)))
      return rc;
    HIPCHK(hipMemcpyAsync(hs->otoff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otcode.p, gcode.data(), gcode.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otid.p, gid.data(), gid.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
  }
  hipEvent_t* ev = ctx->ev;
  if (n_guides) HIPCHK(hipMemcpyAsync(hs->guides.p, guides2, (size_t)n_guides * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(hs->misc.p, 0, 64, ctx->stream));
  HIPCHK(hipEventRecord(ev[0], ctx->stream));
  hawk_launch_scan_raw(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(), hs->counts.as<uint32_t>());
  hawk_launch_mscan(ctx->stream, hs->counts.as<uint32_t>(), ncnt, hs->partial.as<unsigned long long>(), nullptr,
                    hs->offsets.as<uint64_t>(), hs->totals.as<ScanTotals>());
  HIPCHK(hipEventRecord(ev[1], ctx->stream));
  HIPCHK(hipGetLastError());
  ScanTotals tot;
  HIPCHK(hipMemcpyAsync(&tot, hs->totals.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t nsites = tot.n_keep;
  if ((rc = hs->sites.reserve(std::max<uint64_t>(nsites, 1) * sizeof(OtSite))) ||
      (rc = hs->hits.reserve(std::max<uint64_t>(cap, 1) * sizeof(OtHit))))
    return rc;
  unsigned long long* d_nhits = hs->misc.as<unsigned long long>();
  HIPCHK(hipEventRecord(ev[2], ctx->stream));
  if (nsites) hawk_launch_ot_sites(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(),
                                   hs->offsets.as<uint64_t>(), hs->sites.as<OtSite>());
  HIPCHK(hipEventRecord(ev[3], ctx->stream));
  if (pairs) {
    hawk_launch_ot_match_pairs(ctx->stream, hs->sites.as<OtSite>(), nsites, ps, hs->otoff.as<uint32_t>(), hs->otcode.as<uint64_t>(),
                               hs->otid.as<uint32_t>(), n_guides, G, p->right ? (int)p->pamlen : 0, (int)p->max_mm, hs->hits.as<OtHit>(), cap, d_nhits);
  } else if (seeded) {
    if (seed_lds)
      hawk_launch_ot_match_seeded_lds(ctx->stream, hs->sites.as<OtSite>(), nsites, sd, hs->otoff.as<uint32_t>(),
                                      hs->otcode.as<uint64_t>(), hs->otid.as<uint32_t>(), n_guides, n_chunks, G,
                                      p->right ? (int)p->pamlen : 0, (int)p->max_mm, hs->hits.as<OtHit>(), cap, d_nhits);
    else
      hawk_launch_ot_match_seeded(ctx->stream, hs->sites.as<OtSite>(), nsites, sd, hs->otoff.as<uint32_t>(), hs->otcode.as<uint64_t>(),
                                  hs->otid.as<uint32_t>(), n_guides, G, p->right ? (int)p->pamlen : 0, (int)p->max_mm,
                                  hs->hits.as<OtHit>(), cap, d_nhits);
  } else {
    hawk_launch_ot_match(ctx->stream, hs->sites.as<OtSite>(), nsites, hs->guides.as<uint64_t>(), n_guides, (int)p->guidelen,
                         p->right ? (int)p->pamlen : 0, (int)p->max_mm, hs->hits.as<OtHit>(), cap, d_nhits);
  }
  HIPCHK(hipEventRecord(ev[4], ctx->stream));
  HIPCHK(hipGetLastError());
  unsigned long long nh = 0;
  HIPCHK(hipMemcpyAsync(&nh, d_nhits, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n_out = nh;
  if (timing) {
    memset(timing, 0, sizeof(*timing));
    (void)hipEventElapsedTime(&timing->scan_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&timing->sites_ms, ev[2], ev[3]);
    (void)hipEventElapsedTime(&timing->match_ms, ev[3], ev[4]);
    (void)hipEventElapsedTime(&timing->total_ms, ev[0], ev[4]);
    timing->n_sites = nsites;
    uint64_t pos = 0;
    for (uint32_t h = 0; h < hs->n_hap; ++h) pos += (uint64_t)std::max(0, hs->scan_stop[h] - hs->scan_start[h]);
    timing->scanned_positions = pos;
  }
  if (nh > cap) return HAWK_E_CAPACITY;
  if (!nh) return HAWK_OK;
  std::vector<OtHit> hh(nh);
  HIPCHK(hipMemcpy(hh.data(), hs->hits.p, nh * sizeof(OtHit), hipMemcpyDeviceToHost));
  // the sites of the hits: gathered into a compact array on the device, one download
  std::vector<OtSite> ss(nh);
  if ((rc = hs->othit.reserve(nh * sizeof(OtSite)))) return rc;
  hawk_launch_ot_gather(ctx->stream, hs->sites.as<OtSite>(), hs->hits.as<OtHit>(), nh, hs->othit.as<OtSite>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ss.data(), hs->othit.p, nh * sizeof(OtSite), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint64_t i = 0; i < nh; ++i) {
    if (out_guide) out_guide[i] = hh[i].guide;
    if (out_row) out_row[i] = ss[i].row;
    if (out_q) out_q[i] = ss[i].q & 0x7fffffffu;
    if (out_strand) out_strand[i] = (uint8_t)(ss[i].q >> 31);
    if (out_mm) out_mm[i] = (uint8_t)hh[i].mm;
    if (out_code) out_code[i] = ss[i].code;
    if (out_nmask) out_nmask[i] = ss[i].nmask;
  }
  return HAWK_OK;
}

}  // extern "C"
