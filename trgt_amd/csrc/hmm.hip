// trgt_amd/csrc/hmm.hip -- motif-segmentation HMM on gfx950: log-space Viterbi
// fill, traceback and MS/MC/AP decoding, one allele per workgroup.
//
// Replaces (PacificBiosciences/trgt v3.0.0): build_hmm (src/hmm/builder.rs:4-173),
// Hmm::label (hmm_model.rs:54-156), calc_purity/get_events (purity.rs:6-41,
// events.rs:17-117), remove_imperfect_motifs (operations.rs:6-80), label_motifs
// (hmm_model.rs:158-200), count_motifs/collapse_labels (utils.rs:3-27) as composed
// by label_with_hmm (src/trgt/workflows/tr.rs:454-492).
//
// Kernel shape (see DESIGN.md "K4/K5"): one workgroup per allele, one thread per
// HMM state (workgroup = 64*ceil(S/64) threads; S = 17..26 for one STR motif is a
// single wavefront).  The two live score columns sit in LDS as f64; per base the
// emitting states update in parallel, then the silent states are evaluated level
// by level of their (acyclic) dependency order, so every f64 sum is formed exactly
// as the reference forms it: (prev + ln p) + emission, first strict maximum wins.
// No log() is evaluated on the device: the ln tables come from the host libm.
// Back-pointers are one byte per (base, state), written once to HBM (coalesced
// across the states of a column) and re-read in LDS-staged chunks by the traceback,
// which also classifies events (purity) and collects motif visits on the fly.
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <limits>
#include <thread>

#include "common.hpp"

namespace trgt {

// ------------------------------------------------------------ host model

}  // namespace trgt
#include <memory>

#include "hmm_host.hpp"
namespace trgt {

struct HmmJobDev {
  uint32_t set, seq_len, job_index, path_cap;
  uint64_t seq_off, bp_off, path_off, span_off, count_off, visit_off;
  uint64_t map_off;  // words into the visit workspace: chunk maps of the parallel trace-back of long alleles (0: none, the fill kernel traces back itself)
};

static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Everything about a motif set's model that follows from the motif LENGTHS: number of states, block table (start state, end state,
// motif length, offset of the motif's bytes), the lane -> state table of models of more than one wave, and where the tables sit in the
// set's blob (offsets relative to it).  Shared by the host builder (build_set) and the device builder (hmm_model_build_kernel).
static uint64_t layout_set(const uint32_t* mlen, uint32_t n_motifs, HmmSetDev& d, std::vector<uint32_t>& blocks, std::vector<uint16_t>& perm) {
  uint32_t S = 7, max_mlen = 0, mbytes = 0;
  for (uint32_t i = 0; i < n_motifs; ++i) { S += 3 * mlen[i] + 1; max_mlen = std::max(max_mlen, mlen[i]); }
  const uint32_t nb = n_motifs + 1;
  blocks.assign(4 * (size_t)nb, 0);
  uint32_t ms = 2;
  for (uint32_t mi = 0; mi < n_motifs; ++mi) {
    const uint32_t n = mlen[mi], me = ms + 3 * n;
    blocks[0 * nb + mi] = ms; blocks[1 * nb + mi] = me; blocks[2 * nb + mi] = n; blocks[3 * nb + mi] = mbytes;
    mbytes += n;
    ms = me + 1;
  }
  blocks[0 * nb + nb - 1] = ms; blocks[1 * nb + nb - 1] = ms + 2; blocks[2 * nb + nb - 1] = 0; blocks[3 * nb + nb - 1] = mbytes;  // skip block (builder.rs:41-53)
  // The chain of deletion states of a motif block ends in the block end (d0 .. d0 + n - 2, me = d0 + n - 1: consecutive states, each
  // with the one before it as its LAST predecessor): the kernel walks it across LANES, and a chain that runs over a wave boundary
  // costs it another round.  Models of several waves therefore get a lane -> state table that keeps every chain inside one wave
  // (first fit, longest chain first; the other states fill the lanes left over; a chain longer than a wave starts a wave of its own).
  uint32_t chain_rounds = 1, n_lanes = 0;
  perm.clear();
  if (S > 64) {
    struct Chain { uint32_t first, len; };
    std::vector<Chain> chains;
    std::vector<uint8_t> in_chain(S, 0);
    for (uint32_t b = 0; b + 1 < nb; ++b) {
      const uint32_t n = blocks[2 * nb + b], me = blocks[1 * nb + b];
      if (n > 1) { chains.push_back({me - (n - 1), n}); for (uint32_t s = me - (n - 1); s <= me; ++s) in_chain[s] = 1; }
    }
    std::stable_sort(chains.begin(), chains.end(), [](const Chain& a, const Chain& b) { return a.len > b.len; });
    std::vector<uint32_t> fill;  // lanes used per wave
    auto place = [&](uint32_t lane, uint32_t st) { if (perm.size() <= lane) perm.resize(((size_t)lane / 64 + 1) * 64, 0xFFFF); perm[lane] = (uint16_t)st; };
    for (const Chain& ch : chains) {
      uint32_t w = 0;
      if (ch.len > 64) { w = (uint32_t)fill.size(); }  // whole waves of its own
      else { while (w < fill.size() && fill[w] + ch.len > 64) ++w; }
      const uint32_t waves = (ch.len + 63) / 64;
      if (w + waves > fill.size()) fill.resize(w + waves, 0);
      const uint32_t lane0 = 64 * w + fill[w];
      for (uint32_t k = 0; k < ch.len; ++k) place(lane0 + k, ch.first + k);
      for (uint32_t k = 0; k < waves; ++k) fill[w + k] = k + 1 < waves ? 64 : std::max(fill[w + k], (lane0 + ch.len) - 64 * (w + k));
      chain_rounds = std::max(chain_rounds, (lane0 + ch.len - 1) / 64 - lane0 / 64 + 1);
    }
    uint32_t w = 0;
    for (uint32_t st = 0; st < S; ++st) {
      if (in_chain[st]) continue;
      while (w < fill.size() && fill[w] >= 64) ++w;
      if (w == fill.size()) fill.push_back(0);
      place(64 * w + fill[w], st);
      ++fill[w];
    }
    n_lanes = (uint32_t)perm.size();
  }
  d.S = S; d.n_blocks = nb; d.chain_rounds = chain_rounds; d.max_mlen = max_mlen; d.n_lanes = n_lanes;
  d.ppl_lanes = mbytes + 1 <= 64 ? mbytes + 1 : 0;  // (mbytes = the motif lengths summed up: one lane per motif position, one for the skip block)
  uint64_t o = 0;  // offsets relative to this set's blob; the caller rebases them
  d.off_inlp = o; o += 8ull * 4 * S;
  d.off_em = o; o += 8ull * 5 * S;
  d.off_inst = o; o += 2ull * 4 * S;
  d.off_block = o; o += 2ull * S; o = align_up(o, 4);
  d.off_blocks = o; o += 4ull * 4 * nb;
  d.off_nin = o; o += S;
  d.off_level = o; o += S;
  d.off_flags = o; o += S;
  d.off_motifs = o; o += mbytes; o = align_up(o, 2);
  d.off_perm = o; o += 2ull * perm.size();
  return align_up(o, 16);
}

// build_hmm restated as flat tables (builder.rs:4-173); predecessor ORDER is part of the contract
// because the first strict maximum wins in calc_viterbi_score (hmm_model.rs:79-88).  (Host version: the product builds the tables
// on the device, hmm_model_build_kernel below; this one is what trgt_hmm_models_check compares it with, byte for byte.)
static void build_set(const std::vector<std::string>& motifs, std::vector<uint8_t>& blob, HmmSetDev& d) {
  std::vector<uint32_t> mlens, blocks;
  std::vector<uint16_t> perm;
  for (auto& m : motifs) mlens.push_back((uint32_t)m.size());
  const uint64_t bytes = layout_set(mlens.data(), (uint32_t)mlens.size(), d, blocks, perm);
  const uint32_t S = d.S, nb = d.n_blocks;
  const double NINF = -std::numeric_limits<double>::infinity();
  std::vector<double> inlp(4 * (size_t)S, NINF), em(5 * (size_t)S, NINF);
  std::vector<uint16_t> inst(4 * (size_t)S, 0);
  std::vector<int16_t> block(S, -1);
  std::vector<uint8_t> nin(S, 0), level(S, 0), flags(S, 0);
  std::string mbytes;
  auto set_ems = [&](uint32_t st, const double (&p)[5]) { for (int i = 0; i < 5; ++i) em[(size_t)i * S + st] = std::log(p[i]); };
  auto set_trans = [&](uint32_t st, std::initializer_list<uint32_t> ins, std::initializer_list<double> ps) {
    nin[st] = (uint8_t)ins.size();
    int j = 0;
    for (uint32_t s : ins) inst[(size_t)(j++) * S + st] = (uint16_t)s;
    j = 0;
    for (double p : ps) inlp[(size_t)(j++) * S + st] = std::log(p);
  };
  const double SILENT[5] = {0, 0, 0, 0, 0}, TERM[5] = {1, 0, 0, 0, 0}, UNI[5] = {0.00, 0.25, 0.25, 0.25, 0.25};
  const uint32_t start = 0, end = S - 1, rs = 1, re = S - 2;
  set_ems(start, TERM); set_ems(end, TERM);
  set_trans(end, {re}, {0.10});
  set_ems(rs, SILENT);
  set_trans(rs, {start, re}, {1.00, 1.00});
  const double rs_to_ms = 1.00, me_to_re = 0.50;
  uint32_t ms = rs + 1;
  for (size_t mi = 0; mi < motifs.size(); ++mi) {
    const std::string& motif = motifs[mi];
    const uint32_t n = (uint32_t)motif.size(), me = ms + 3 * n;
    set_ems(ms, SILENT);
    set_trans(ms, {rs, me}, {rs_to_ms, 1.0 - me_to_re});
    // define_motif_block (builder.rs:80-173)
    const uint32_t m0 = ms + 1, i0 = m0 + n, d0 = i0 + n;
    const double match_prob = 0.90, ins_to_ins = 0.25, match_to_indel = (1.00 - match_prob) / 2.00, del_to_match = 0.50;
    const double seed = 2.00 * (1.00 - match_prob) / (double)((size_t)n * (size_t)(n - 1));
    for (uint32_t k = 0; k < n; ++k) {
      double e[5] = {0.00, 0.03, 0.03, 0.03, 0.03};
      switch (motif[k]) {
        case 'A': e[1] = 0.90; break;
        case 'T': e[2] = 0.90; break;
        case 'C': e[3] = 0.90; break;
        case 'G': e[4] = 0.90; break;
        default: e[1] = e[2] = e[3] = e[4] = 0.25; break;  // 'N'
      }
      set_ems(m0 + k, e);
      if (k == 0) set_trans(m0, {ms}, {match_prob});
      else if (k == 1) set_trans(m0 + 1, {m0, ms, i0}, {match_prob, seed * (double)(n - k), 1.0 - ins_to_ins});
      else set_trans(m0 + k, {m0 + k - 1, ms, i0 + k - 1, d0 + k - 2}, {match_prob, seed * (double)(n - k), 1.0 - ins_to_ins, del_to_match});
    }
    for (uint32_t k = 0; k < n; ++k) { set_ems(i0 + k, UNI); set_trans(i0 + k, {i0 + k, m0 + k}, {ins_to_ins, match_to_indel}); }
    for (uint32_t k = 0; k + 1 < n; ++k) {
      set_ems(d0 + k, SILENT);
      if (k == 0) set_trans(d0, {m0}, {match_to_indel});
      else set_trans(d0 + k, {m0 + k, d0 + k - 1}, {match_to_indel, 1.0 - del_to_match});
    }
    set_ems(me, SILENT);
    if (n > 1) set_trans(me, {m0 + n - 1, i0 + n - 1, d0 + n - 2}, {match_prob, 1.0 - ins_to_ins, 1.0});
    else set_trans(me, {m0 + n - 1, i0 + n - 1}, {match_prob, 1.0 - ins_to_ins});
    for (uint32_t s = ms; s <= me; ++s) block[s] = (int16_t)mi;
    mbytes += motif;
    ms = me + 1;
  }
  // skip block (builder.rs:41-53)
  const uint32_t skip = ms + 1, me_skip = ms + 2;
  set_ems(ms, SILENT);
  set_trans(ms, {rs, me_skip}, {rs_to_ms, 1.0 - me_to_re});
  set_ems(skip, UNI);
  set_trans(skip, {ms, skip}, {1.0, 0.5});
  set_ems(me_skip, SILENT);
  set_trans(me_skip, {skip}, {1.0 - 0.5});
  for (uint32_t s = ms; s <= me_skip; ++s) block[s] = (int16_t)(nb - 1);
  // run end: predecessors are the block ends in block order, each ln(me_to_re) (builder.rs:55-57)
  set_ems(re, SILENT);
  nin[re] = 0xFF;
  inlp[re] = std::log(me_to_re);
  // flags; silent states are evaluated after the emitting ones of a column (level 1; any topological order of the silent states
  // gives identical values, hmm_model.rs:206-240 -- the kernel's passes are one)
  for (uint32_t s = 0; s < S; ++s) {
    bool any = false, base = false;
    for (int i = 0; i < 5; ++i) if (std::isfinite(em[(size_t)i * S + s])) { any = true; if (i) base = true; }
    flags[s] = (any ? 1 : 0) | (base ? 2 : 0);
    level[s] = any ? 0 : 1;
  }
  // serialise
  blob.assign((size_t)bytes, 0);
  std::memcpy(&blob[d.off_inlp], inlp.data(), 8ull * 4 * S);
  std::memcpy(&blob[d.off_em], em.data(), 8ull * 5 * S);
  std::memcpy(&blob[d.off_inst], inst.data(), 2ull * 4 * S);
  std::memcpy(&blob[d.off_block], block.data(), 2ull * S);
  std::memcpy(&blob[d.off_blocks], blocks.data(), 4ull * 4 * nb);
  std::memcpy(&blob[d.off_nin], nin.data(), S);
  std::memcpy(&blob[d.off_level], level.data(), S);
  std::memcpy(&blob[d.off_flags], flags.data(), S);
  if (!mbytes.empty()) std::memcpy(&blob[d.off_motifs], mbytes.data(), mbytes.size());
  if (!perm.empty()) std::memcpy(&blob[d.off_perm], perm.data(), 2ull * perm.size());
}

// ---- the same tables built on the device: one workgroup per motif set.  The natural logarithms are the HOST's (the reference's are
// Rust f64::ln = the platform libm, and the scores are sums of them: bit-exact parity needs the very same values): the handful of
// constants comes in by value, ln(seed(n) * (n - k)) from a table with one row per distinct motif length of the batch.
struct HmmBuildConsts {
  double ln_match, ln_1m_ins, ln_del2m, ln_ins2ins, ln_m2indel, ln_1m_del2m, ln_one, ln_1m_me2re, ln_me2re, ln_end, ln_skip_self, ln_skip_out,
         em_hit, em_miss, em_n, em_uni;
};
static HmmBuildConsts hmm_build_consts() {
  const double match_prob = 0.90, ins_to_ins = 0.25, match_to_indel = (1.00 - match_prob) / 2.00, del_to_match = 0.50, me_to_re = 0.50;
  HmmBuildConsts k;
  k.ln_match = std::log(match_prob); k.ln_1m_ins = std::log(1.0 - ins_to_ins); k.ln_del2m = std::log(del_to_match); k.ln_ins2ins = std::log(ins_to_ins);
  k.ln_m2indel = std::log(match_to_indel); k.ln_1m_del2m = std::log(1.0 - del_to_match); k.ln_one = std::log(1.00); k.ln_1m_me2re = std::log(1.0 - me_to_re);
  k.ln_me2re = std::log(me_to_re); k.ln_end = std::log(0.10); k.ln_skip_self = std::log(0.5); k.ln_skip_out = std::log(1.0 - 0.5);
  k.em_hit = std::log(0.90); k.em_miss = std::log(0.03); k.em_n = std::log(0.25); k.em_uni = std::log(0.25);
  return k;
}
struct HmmBuildArgs {
  const HmmSetDev* sets; uint8_t* blob;
  const uint8_t* motif_bytes;       // sanitised motifs of all sets, back to back
  const uint32_t* motif_off;        // [n_motifs_total + 1] into motif_bytes
  const uint32_t* set_motif_begin;  // [n_sets + 1]
  const uint32_t* seed_row;         // [n_motifs_total] first entry of the motif's row of seed_tab (entry k: ln(seed(n) * (n - k)))
  const double* seed_tab;
  const uint16_t* perm_all; const uint64_t* perm_src;  // lane -> state tables of the sets that have one, [n_sets] offsets into perm_all
  uint64_t blob_bytes;              // all sets' tables together (the end of the last set's)
  HmmBuildConsts k;
};
__global__ void __launch_bounds__(64) hmm_model_build_kernel(const HmmBuildArgs a) {
  const int s = (int)blockIdx.x, tid = (int)threadIdx.x;
  const HmmSetDev d = a.sets[s];
  const int S = (int)d.S, nb = (int)d.n_blocks;
  const uint32_t mb = a.set_motif_begin[s];
  uint8_t* const blob = a.blob;
  double* inlp = reinterpret_cast<double*>(blob + d.off_inlp);
  double* em = reinterpret_cast<double*>(blob + d.off_em);
  uint16_t* inst = reinterpret_cast<uint16_t*>(blob + d.off_inst);
  int16_t* block = reinterpret_cast<int16_t*>(blob + d.off_block);
  uint32_t* blocks = reinterpret_cast<uint32_t*>(blob + d.off_blocks);
  uint8_t *nin = blob + d.off_nin, *level = blob + d.off_level, *flags = blob + d.off_flags, *mot = blob + d.off_motifs;
  __shared__ uint32_t b_start[256];  // first state of every block (n_blocks <= 254)
  if (tid == 0) {
    uint32_t ms = 2, mbytes = 0;
    for (int mi = 0; mi + 1 < nb; ++mi) {
      const uint32_t n = a.motif_off[mb + mi + 1] - a.motif_off[mb + mi], me = ms + 3 * n;
      blocks[0 * nb + mi] = ms; blocks[1 * nb + mi] = me; blocks[2 * nb + mi] = n; blocks[3 * nb + mi] = mbytes;
      b_start[mi] = ms;
      mbytes += n; ms = me + 1;
    }
    blocks[0 * nb + nb - 1] = ms; blocks[1 * nb + nb - 1] = ms + 2; blocks[2 * nb + nb - 1] = 0; blocks[3 * nb + nb - 1] = mbytes;
    b_start[nb - 1] = ms;
  }
  __syncthreads();
  const double NINF = -__builtin_huge_val();
  const HmmBuildConsts& k = a.k;
  for (int st = tid; st < S; st += 64) {
    double lp[4] = {NINF, NINF, NINF, NINF}, e[5] = {NINF, NINF, NINF, NINF, NINF};
    uint32_t in[4] = {0, 0, 0, 0};
    int n_in = 0, blk = -1;
    if (st == 0) { e[0] = k.ln_one; }                                                   // start: emits '#'
    else if (st == S - 1) { e[0] = k.ln_one; n_in = 1; in[0] = (uint32_t)S - 2; lp[0] = k.ln_end; }   // end <- run end
    else if (st == 1) { n_in = 2; in[0] = 0; in[1] = (uint32_t)S - 2; lp[0] = k.ln_one; lp[1] = k.ln_one; }  // run start <- {start, run end}
    else if (st == S - 2) { n_in = 0xFF; lp[0] = k.ln_me2re; }                          // run end <- the block ends
    else {
      int lo = 0, hi = nb - 1;  // the block of the state
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b_start[mid] <= (uint32_t)st) lo = mid; else hi = mid - 1; }
      blk = lo;
      const uint32_t ms = b_start[blk];
      if (blk == nb - 1) {  // skip block
        const uint32_t skip = ms + 1, me = ms + 2;
        if ((uint32_t)st == ms) { n_in = 2; in[0] = 1; in[1] = me; lp[0] = k.ln_one; lp[1] = k.ln_1m_me2re; }
        else if ((uint32_t)st == skip) { e[1] = e[2] = e[3] = e[4] = k.em_uni; n_in = 2; in[0] = ms; in[1] = skip; lp[0] = k.ln_one; lp[1] = k.ln_skip_self; }
        else { n_in = 1; in[0] = skip; lp[0] = k.ln_skip_out; }
      } else {
        const uint32_t mo = a.motif_off[mb + blk], n = a.motif_off[mb + blk + 1] - mo;
        const uint32_t me = ms + 3 * n, m0 = ms + 1, i0 = m0 + n, d0 = i0 + n;
        const double* seed = a.seed_tab + a.seed_row[mb + blk];
        if ((uint32_t)st == ms) { n_in = 2; in[0] = 1; in[1] = me; lp[0] = k.ln_one; lp[1] = k.ln_1m_me2re; }
        else if ((uint32_t)st == me) {
          if (n > 1) { n_in = 3; in[0] = m0 + n - 1; in[1] = i0 + n - 1; in[2] = d0 + n - 2; lp[0] = k.ln_match; lp[1] = k.ln_1m_ins; lp[2] = k.ln_one; }
          else { n_in = 2; in[0] = m0 + n - 1; in[1] = i0 + n - 1; lp[0] = k.ln_match; lp[1] = k.ln_1m_ins; }
        } else if ((uint32_t)st < i0) {  // match state
          const uint32_t kk = (uint32_t)st - m0;
          const uint8_t c = a.motif_bytes[mo + kk];
          if (c == 'A' || c == 'T' || c == 'C' || c == 'G') { e[1] = e[2] = e[3] = e[4] = k.em_miss; e[c == 'A' ? 1 : c == 'T' ? 2 : c == 'C' ? 3 : 4] = k.em_hit; }
          else e[1] = e[2] = e[3] = e[4] = k.em_n;
          if (kk == 0) { n_in = 1; in[0] = ms; lp[0] = k.ln_match; }
          else if (kk == 1) { n_in = 3; in[0] = m0; in[1] = ms; in[2] = i0; lp[0] = k.ln_match; lp[1] = seed[1]; lp[2] = k.ln_1m_ins; }
          else { n_in = 4; in[0] = m0 + kk - 1; in[1] = ms; in[2] = i0 + kk - 1; in[3] = d0 + kk - 2; lp[0] = k.ln_match; lp[1] = seed[kk]; lp[2] = k.ln_1m_ins; lp[3] = k.ln_del2m; }
        } else if ((uint32_t)st < d0) {  // insertion state
          const uint32_t kk = (uint32_t)st - i0;
          e[1] = e[2] = e[3] = e[4] = k.em_uni;
          n_in = 2; in[0] = i0 + kk; in[1] = m0 + kk; lp[0] = k.ln_ins2ins; lp[1] = k.ln_m2indel;
        } else {  // deletion state
          const uint32_t kk = (uint32_t)st - d0;
          if (kk == 0) { n_in = 1; in[0] = m0; lp[0] = k.ln_m2indel; }
          else { n_in = 2; in[0] = m0 + kk; in[1] = d0 + kk - 1; lp[0] = k.ln_m2indel; lp[1] = k.ln_1m_del2m; }
        }
      }
    }
    bool any = false, base = false;
#pragma unroll
    for (int i = 0; i < 5; ++i) { em[(size_t)i * S + st] = e[i]; if (e[i] > NINF) { any = true; if (i) base = true; } }
#pragma unroll
    for (int j = 0; j < 4; ++j) { inlp[(size_t)j * S + st] = lp[j]; inst[(size_t)j * S + st] = (uint16_t)in[j]; }
    block[st] = (int16_t)blk; nin[st] = (uint8_t)n_in; flags[st] = (uint8_t)((any ? 1 : 0) | (base ? 2 : 0)); level[st] = any ? 0 : 1;
  }
  {  // motif bytes, lane table, padding
    const uint32_t m_lo = a.motif_off[mb], m_hi = a.motif_off[mb + nb - 1];
    if (tid == 0) {  // the alignment gaps of the set's tables (layout_set): zero, as in the host builder's blob
      const uint64_t end = (uint32_t)s + 1u < gridDim.x ? a.sets[s + 1].off_inlp : a.blob_bytes;
      for (uint64_t q = d.off_block + 2ull * (uint64_t)S; q < d.off_blocks; ++q) blob[q] = 0;
      for (uint64_t q = d.off_motifs + (uint64_t)(m_hi - m_lo); q < d.off_perm; ++q) blob[q] = 0;
      for (uint64_t q = d.off_perm + 2ull * d.n_lanes; q < end; ++q) blob[q] = 0;
    }
    for (uint32_t i = (uint32_t)tid; i < m_hi - m_lo; i += 64) mot[i] = a.motif_bytes[m_lo + i];
    if (d.n_lanes) {
      uint16_t* perm = reinterpret_cast<uint16_t*>(blob + d.off_perm);
      const uint16_t* src = a.perm_all + a.perm_src[s];
      for (uint32_t i = (uint32_t)tid; i < d.n_lanes; i += 64) perm[i] = src[i];
    }
  }
}

// --------------------------------------------------------------- kernel
#ifdef TRGT_HMM_PROF
// developer build (make HMMPROF=1): lane 0 of every job splits its shader-clock time over the phases of the kernel
__device__ unsigned long long g_hmm_prof[16];
#define HP_DECL unsigned long long hp_t = clock64()
#define HP_MARK(i) do { const unsigned long long n_ = clock64(); if (tid == 0) atomicAdd(&g_hmm_prof[i], n_ - hp_t); hp_t = n_; } while (0)
#define HP_FILL_DECL unsigned long long hf_t = clock64(), hf_acc[6] = {0, 0, 0, 0, 0, 0}
#define HP_FILL(i) do { const unsigned long long n_ = clock64(); hf_acc[i] += n_ - hf_t; hf_t = n_; } while (0)
#define HP_FILL_END do { if (tid == 0) for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&g_hmm_prof[8 + q_], hf_acc[q_]); } while (0)
#define HP_LONG_DECL unsigned long long hl_t = clock64()
#define HP_LONG(i) do { const unsigned long long n_ = clock64(); if (tid == 0) atomicAdd(&g_hmm_prof[i], n_ - hl_t); hl_t = n_; } while (0)
#else
#define HP_DECL
#define HP_MARK(i)
#define HP_FILL_DECL
#define HP_FILL(i)
#define HP_FILL_END
#define HP_LONG_DECL
#define HP_LONG(i)
#endif
constexpr int HMM_LONG_MIN = 1536;   // columns from which an allele's trace-back goes to hmm_traceback_long_kernel (round 5, tried 512: the cfg3 call 9.5 -> 9.3 ms, but the hundreds of 0.5-2 kb VNTR alleles of a cfg4 step through the many-wave kernel 9.1 -> 11.5 ms)
constexpr int HMM_LONG_CHUNK = 64;   // columns per chunk map there
constexpr int HMM_STAGE_BYTES = 1024;  // LDS staging window for back-pointer columns during traceback (one-wave models)
// ... models of several waves (rows of 80 to 320 bytes) stage 16 columns at a time: 1 KB held five columns of a 192-state model, and every
// chunk costs a global round trip and two barriers
__host__ __device__ constexpr int hmm_stage_bytes(int spad) { return spad > 64 ? (16 * spad > HMM_STAGE_BYTES ? 16 * spad : HMM_STAGE_BYTES) : HMM_STAGE_BYTES; }
constexpr int HMM_LDS_PER_STATE = 16 + 16 + 40 + 4 + 16 + 2 + 1 + 1;  // two score columns, lp[2], em[5], info, inst[4] (32-bit entries), block, flags, bp column

__device__ __forceinline__ int hmm_code(const uint8_t* __restrict__ seq, int i, int L) {
  // '#'+seq+'#' with encode_base (hmm_model.rs:243-252) after replace_invalid_bases(seq, ATCG) (utils.rs:29-42)
  if (i == 0 || i == L - 1) return 0;
  const uint8_t b = seq[i - 1];
  return b == 'A' ? 1 : b == 'T' ? 2 : b == 'C' ? 3 : b == 'G' ? 4 : ((i - 1) & 3) + 1;
}

// "#ATCG"[code] without a memory access (a string constant is a global load: a round trip per trace-back step)
__device__ __forceinline__ int hmm_code_char(int code) { return (int)((0x4743544123ull >> (8 * code)) & 0xFFull); }

// Workgroup synchronisation of the Viterbi kernel.  Most motif sets need at most 64 states, i.e. a single wavefront: its lanes run in
// lockstep and the LDS unit serves one wave's accesses in order, so a compiler-level fence is all that is needed -- whereas
// __syncthreads() also waits for every global store in flight (the back-pointer column written at the end of each step), a
// memory round trip per step.
// Several waves: an LDS-only barrier.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier, and the vmcnt(0) makes every
// one of the four barriers of a column wait for the back-pointer stores of the column before (a round trip to L2 each, most of a
// column's time for the 170- to 190-state models); those bytes are read again only by the trace-back, behind hmm_sync_mem().
__device__ __forceinline__ void hmm_sync(int nthr) {
  if (nthr == 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// ... and the full one: global stores of the whole workgroup are visible afterwards
__device__ __forceinline__ void hmm_sync_mem(int nthr) {
  if (nthr == 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  else __syncthreads();
}

// LDS of a job does not depend on the length of its allele (it bounds the waves per SIMD of a kernel that is bound by instruction
// issue and dependent LDS round trips): the fill and the trace-back see the allele through a WINDOW of symbol codes, the trace-back
// keeps the first HMM_VIS_LDS motif visits in LDS and the rest in the job's global workspace (read back in staged chunks), the
// check of the motif copies (remove_imperfect_motifs) happens in the trace-back, where the window holds their bases.
constexpr int HMM_CODE_WINDOW = 256;  // columns of symbol codes kept in LDS (+ HMM_CODE_PAD behind them)
constexpr int HMM_CODE_PAD = 16;      // the column after the window (fill), the bases of a motif copy that starts at its end (trace-back)
constexpr int HMM_VIS_LDS = 64;       // motif visits kept in LDS (block | dropped << 15, first base, one past the last base)
// SUB: lanes per allele.  64 (or more: one thread per state, several waves for large motif sets) is the general case; SUB = 32
// packs TWO alleles into one wave when the model has at most 32 states (a single STR motif of up to 8 bases): the kernel is bound
// by instruction issue and most passes of a column keep only one or two lanes busy, so halving the waves nearly halves its time.
// The two halves run the same code on their own LDS regions; their control flow may diverge (different allele lengths).
// the value of the lane before (lane 0: its own)
__device__ __forceinline__ double wave_shr1_f64(double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const int lo = (int)(u & 0xFFFFFFFFull), hi = (int)(u >> 32);
  const int slo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);  // wave_shr:1
  const int shi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)shi << 32) | (unsigned)slo));
}

// max of two scores: the value of (a > b ? a : b) -- scores are never NaN or -0.0 -- in ONE v_max_f64 (fmax() comes with a canonicalising
// v_max_f64 x, x in front of it: 16 more cycles in a loop that runs once per base of the longest motif per column)
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// f64 from another lane of the wave (byte address of the source lane: lane << 2)
__device__ __forceinline__ double bperm_f64(int addr, double x) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(u & 0xFFFFFFFFull)), hi = __builtin_amdgcn_ds_bpermute(addr, (int)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// ---- the packed back-pointer rows of hmm_fill_ppl_kernel<G, true> (hmm_ppl.hpp): one byte per LANE (motif position) and column.
__host__ __device__ inline int hmm_ppl_group(uint32_t positions) { return positions == 0 || positions > 64 ? 0 : positions <= 8 ? 8 : positions <= 16 ? 16 : positions <= 32 ? 32 : 64; }
// Where the back-pointer of state `st` lies in such a row: lane | shift << 6 | mask << 9 (mask 0: a state whose back-pointer is a constant 0 --
// the end state -- or is never read; the run end's block index is spread over the lanes 1-3: hmm_bp_run_end).  blocks: [4][nb] start state,
// end state, motif length, first position (= lane) of every block; blk: the state's block (-1: outside the blocks).
__device__ __forceinline__ uint32_t hmm_bp_loc(int st, int nb, int blk, const uint32_t* blocks) {
  if (st == 1) return 0u | (6u << 6) | (3u << 9);               // run start: lane 0, bits 6-7
  if (blk < 0) return 0u;
  const int ms = (int)blocks[blk], n = (int)blocks[2 * nb + blk], off = st - ms;
  const uint32_t pos0 = blocks[3 * nb + blk];
  if (off == 0) return pos0 | (5u << 6) | (1u << 9);            // block start: bit 5 of the block's first lane
  if (blk == nb - 1) return off == 1 ? (pos0 | (0u << 6) | (3u << 9)) : (pos0 | (3u << 6) | (3u << 9));  // skip state | skip block's end
  if (off <= n) return (pos0 + (uint32_t)(off - 1)) | (0u << 6) | (3u << 9);              // match state k
  if (off <= 2 * n) return (pos0 + (uint32_t)(off - n - 1)) | (2u << 6) | (1u << 9);      // insertion state k
  return (pos0 + (uint32_t)(off - 2 * n - 1)) | (3u << 6) | (3u << 9);                    // deletion state k / block end (k = n - 1)
}
__device__ __forceinline__ uint32_t hmm_bp_unpack(uint32_t byte, uint32_t loc) { return (byte >> ((loc >> 6) & 7u)) & (loc >> 9); }
__device__ __forceinline__ uint32_t hmm_bp_run_end(uint32_t row_word) { return ((row_word >> 14) & 3u) | ((row_word >> 20) & 0xCu) | ((row_word >> 26) & 0x30u); }
// loc of the end state of block b (the deletion slot of its last position; the skip block has one position)
__device__ __forceinline__ uint32_t hmm_bp_loc_block_end(int b, int nb, const uint32_t* blocks) {
  const uint32_t n = blocks[2 * nb + b];
  return (blocks[3 * nb + b] + (n ? n - 1u : 0u)) | (3u << 6) | (3u << 9);
}
// predecessor entries of the trace-backs: predecessor state | its "emits" bit << 15 | its hmm_bp_loc << 16 (what the chase needs to know
// about a state is there when it arrives in it: one LDS round trip per step)
__device__ __forceinline__ uint32_t hmm_pred_entry(uint32_t pr, int S, int nb, const uint8_t* g_flags, const int16_t* g_block, const uint32_t* g_blocks) {
  if (pr >= (uint32_t)S) return pr;
  return pr | ((uint32_t)(g_flags[pr] & 1) << 15) | (hmm_bp_loc((int)pr, nb, (int)g_block[pr], g_blocks) << 16);
}

// ONE_WAVE (models of at most 64 states: one state per lane of ONE wave, or two alleles of at most 32 states in its halves): the score
// columns of the fill live in REGISTERS -- a state's predecessors are fetched from their lanes (ds_bpermute: one LDS-crossbar round per
// pass, nothing written) instead of through two LDS arrays (a write, a fence and a read per pass), and what the run-end lane worked out
// alone (the maximum over the block ends, the run start behind it) every lane of the job works out for itself from the same fetched
// values.  Same sums, same predecessor order, same strict '>'.
template <int SUB, bool ONE_WAVE>
__global__ void hmm_viterbi_kernel(const HmmJobDev* __restrict__ jobs, const HmmSetDev* __restrict__ sets,
                                   const uint8_t* __restrict__ model, const uint8_t* __restrict__ seq_blob,
                                   uint8_t* __restrict__ bp_ws, uint32_t* __restrict__ visit_ws,
                                   uint16_t* __restrict__ path, uint32_t* __restrict__ path_len,
                                   int32_t* __restrict__ spans3, uint32_t* __restrict__ n_spans,
                                   uint32_t* __restrict__ counts, double* __restrict__ purity,
                                   int32_t* __restrict__ edit_out, int32_t* __restrict__ maxd_out,
                                   uint32_t n_launch_jobs, uint32_t lds_per_job, const uint32_t* __restrict__ n_jobs_dev, uint32_t* __restrict__ long_list) {
  extern __shared__ __align__(16) unsigned char lds_all[];
  if (n_jobs_dev) n_launch_jobs = *n_jobs_dev;  // a job list resolved on the device (hmm_resolve_kernel): the grid covers all candidates
  const bool four_rounds = (lds_per_job >> 31) != 0u;  // (TRGT_HMM_FOUR_ROUNDS, see the register fill)
  const bool ppl_filled = ((lds_per_job >> 30) & 1u) != 0u;
  const bool ppl_packed = ((lds_per_job >> 23) & 1u) != 0u;  // ... in rows of one byte per lane (hmm_fill_ppl_kernel<G, true>)
  const int long_min = ((lds_per_job >> 24) & 0x3Fu) ? (int)((lds_per_job >> 24) & 0x3Fu) * 256 : HMM_LONG_MIN;  // (by the size of the class, hmm_long_min)  // the back-pointers of sets with ppl_lanes are there already (hmm_fill_ppl_kernel ran in front)
  lds_per_job &= 0x007FFFFFu;
  const int grp = SUB == 32 ? (int)(threadIdx.x >> 5) : 0;
  const int tid = SUB == 32 ? (int)(threadIdx.x & 31) : (int)threadIdx.x, nthr = SUB == 32 ? 32 : (int)blockDim.x;
  const int sync_n = SUB == 32 ? 64 : (int)blockDim.x;  // see hmm_sync: a single wave needs no barrier
  const uint32_t jidx = SUB == 32 ? blockIdx.x * 2u + (uint32_t)grp : blockIdx.x;
  if (jidx >= n_launch_jobs) return;
  HP_DECL;
  unsigned char* const lds = lds_all + (size_t)grp * lds_per_job;
  const HmmJobDev job = jobs[jidx];
  const HmmSetDev set = sets[job.set];
  const int S = (int)set.S, nb = (int)set.n_blocks, n_motifs = nb - 1;
  const int qlen = (int)job.seq_len, L = qlen + 2;
  const int Spad = (S + 15) & ~15;
  const double NINF = -__builtin_huge_val();

  for (int m = tid; m < n_motifs; m += nthr) counts[job.count_off + m] = 0;
  if (qlen == 0) {  // Hmm::label returns an empty path; calc_purity returns NaN (hmm_model.rs:145-147, purity.rs:7-9)
    if (tid == 0) {
      if (path_len) path_len[job.job_index] = 0;
      n_spans[job.job_index] = 0;
      purity[job.job_index] = __builtin_nan("");
      if (edit_out) edit_out[job.job_index] = 0;
      if (maxd_out) maxd_out[job.job_index] = 0;
    }
    return;
  }
  // ---- LDS carve-up (all dynamic, 16-byte aligned pieces; no static LDS in front of it)
  int* tb = reinterpret_cast<int*>(lds);  // traceback state shared between the walker and the stagers
  int &tb_state = tb[0], &tb_idx = tb[1], &tb_done = tb[2], &tb_npath = tb[3], &tb_nvisit = tb[4], &tb_edit = tb[5],
      &tb_ref = tb[6], &tb_next = tb[7], &tb_vb1 = tb[8];
  uint32_t* l_inst = reinterpret_cast<uint32_t*>(lds + 64);           // [S][4] hmm_pred_entry (predecessor b of state s at 4 s + b: 16 bytes per state, one read)
  double* sc0 = reinterpret_cast<double*>(l_inst + 4 * S);
  double* sc1 = sc0 + S;
  double* l_lp = sc1 + S;                                             // [2][S] ln transition probabilities of predecessors 0 and 1
  double* l_em = l_lp + 2 * S;                                        // [5][S] ln emission probabilities by symbol code: one LDS read per column instead of a select tree
  uint32_t* l_info = reinterpret_cast<uint32_t*>(l_em + 5 * S);       // [S] what the traceback needs to know about a state, in one word
  int16_t* l_block = reinterpret_cast<int16_t*>(l_info + S);         // [S]
  uint8_t* l_flags = reinterpret_cast<uint8_t*>(l_block + S);        // [S]
  uint8_t* l_bpcol = l_flags + S;                                     // [S] back-pointers of states evaluated by another lane
  uint32_t* l_blocks = reinterpret_cast<uint32_t*>(lds + 64 + (((size_t)HMM_LDS_PER_STATE * S + 15) & ~(size_t)15));  // [4][nb]
  uint8_t* l_stage = lds + 64 + (((size_t)HMM_LDS_PER_STATE * S + 15) & ~(size_t)15) + (((size_t)16 * nb + 15) & ~(size_t)15);

  const double* g_inlp = reinterpret_cast<const double*>(model + set.off_inlp);
  const double* g_em = reinterpret_cast<const double*>(model + set.off_em);
  const uint16_t* g_inst = reinterpret_cast<const uint16_t*>(model + set.off_inst);
  const int16_t* g_block = reinterpret_cast<const int16_t*>(model + set.off_block);
  const uint32_t* g_blocks = reinterpret_cast<const uint32_t*>(model + set.off_blocks);
  const uint8_t* g_motifs = model + set.off_motifs;
  // behind the back-pointer staging window: window of symbol codes | motif bytes | motif visits | motif counts
  uint8_t* l_seq = l_stage + hmm_stage_bytes(Spad);
  uint8_t* l_mot = l_seq + HMM_CODE_WINDOW + HMM_CODE_PAD;
  const int mot_bytes = (S - 7 - n_motifs) / 3;
  uint32_t* l_vis = reinterpret_cast<uint32_t*>(l_mot + ((mot_bytes + 15) & ~15));
  uint32_t* l_cnt = l_vis + 3 * HMM_VIS_LDS;
  uint32_t* l_rec = l_cnt + ((nb + 1) & ~1);  // [HMM_REC_MAX][2] (state, column) of the steps of a trace-back round
  for (int i = tid; i < mot_bytes; i += nthr) l_mot[i] = g_motifs[i];
  for (int i = tid; i < n_motifs; i += nthr) l_cnt[i] = 0;
  const uint8_t* const motif_bytes = l_mot;
  // (bit 15 of a predecessor entry: that state emits a base -- the trace-back then knows it on arrival, without a look-up of its own)
  for (int i = tid; i < 4 * S; i += nthr) l_inst[4 * (i % S) + i / S] = hmm_pred_entry(g_inst[i], S, nb, model + set.off_flags, g_block, g_blocks);
  // (a job whose back-pointers the position-per-lane fill has written is only traced back here: the transition / emission tables and the
  //  per-state registers of this kernel's own fill are not loaded for it -- a dozen rounds of global loads per job)
  const bool own_fill = !(ppl_filled && set.ppl_lanes != 0u);
  for (int i = tid; i < S; i += nthr) { l_block[i] = g_block[i]; l_flags[i] = model[set.off_flags + i]; if (own_fill) { l_lp[i] = g_inlp[i]; l_lp[S + i] = g_inlp[S + i]; } }
  if (own_fill) for (int i = tid; i < 5 * S; i += nthr) l_em[i] = g_em[i];
  for (int i = tid; i < 4 * nb; i += nthr) l_blocks[i] = g_blocks[i];

  // ---- my state's tables in registers
  // (models of several waves: the lane -> state table of build_set, which keeps every deletion chain inside one wave)
  const int st_lane = set.n_lanes ? (int)reinterpret_cast<const uint16_t*>(model + set.off_perm)[tid] : tid;
  const bool act = st_lane < S;
  const int st = act ? st_lane : 0;
  const int n_in = own_fill ? (int)model[set.off_nin + st] : 0;
  const int level = own_fill ? (int)model[set.off_level + st] : 0;
  double lp0 = 0.0, lp1 = 0.0, lp2 = 0.0, lp3 = 0.0;
  int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  if (own_fill) {
    lp0 = g_inlp[0 * S + st]; lp1 = g_inlp[1 * S + st]; lp2 = g_inlp[2 * S + st]; lp3 = g_inlp[3 * S + st];
    p0 = g_inst[0 * S + st]; p1 = g_inst[1 * S + st]; p2 = g_inst[2 * S + st]; p3 = g_inst[3 * S + st];
  }
  // predecessor slots that do not exist read score 0 of state 0 and are ignored (n_in guards the comparison)
  const int q0 = (n_in != 0xFF && n_in > 0) ? p0 : 0, q1 = (n_in != 0xFF && n_in > 1) ? p1 : 0, q2 = (n_in != 0xFF && n_in > 2) ? p2 : 0, q3 = (n_in != 0xFF && n_in > 3) ? p3 : 0;
  const uint8_t* __restrict__ seq = seq_blob + job.seq_off;
  // Symbol code of column i ('#' + allele + '#', hmm_code): from the window l_seq, columns [win0, win0 + HMM_CODE_WINDOW + pad).  A
  // load per column from global memory put a memory round trip -- and, the loads and the back-pointer stores sharing one in-order
  // counter, the store of the column before -- into every step.
  int win0 = 0;
  auto code_at = [&](int i) -> int { return (int)l_seq[i - win0]; };
  uint8_t* __restrict__ bp = bp_ws + job.bp_off;
  hmm_sync_mem(sync_n);  // (also orders the zeroing of the motif counts above before thread 0 counts into them)
  // traceback word of my state: kind (0 outside any block, 1 block start, 2 block end, 3 skip state, 4 match, 5 insertion,
  // 6 deletion) | emits << 3 | block << 8 | expected motif base << 16 (match states)
  if (act) {
    const int blk = (int)l_block[st];
    uint32_t kind = 0, expected = 0;
    if (blk >= 0) {
      const int bstart = (int)l_blocks[0 * nb + blk], bend = (int)l_blocks[1 * nb + blk];
      if (st == bstart) kind = 1;
      else if (st == bend) kind = 2;
      else if (blk == nb - 1) kind = 3;
      else {
        const int mlen = (int)l_blocks[2 * nb + blk], off = st - bstart - 1, k = off / mlen;
        kind = 4u + (uint32_t)k;
        if (k == 0) expected = motif_bytes[l_blocks[3 * nb + blk] + off];
      }
    }
    l_info[st] = kind | ((uint32_t)(l_flags[st] & 1) << 3) | ((uint32_t)(blk & 0xFF) << 8) | (expected << 16);
  }
  // roles in the evaluation of the silent states of a column (see the fill loop)
  const int my_blk = act ? (int)l_block[st] : -1;
  const bool role_end = act && my_blk >= 0 && st == (int)l_blocks[1 * nb + my_blk];
  const bool role_start = act && my_blk >= 0 && st == (int)l_blocks[0 * nb + my_blk];
  const bool role_del = act && level > 0 && !role_end && !role_start && n_in != 0xFF && my_blk >= 0;  // deletion states
  const bool role_chain = role_del || role_end;
  // a chain state's LAST predecessor is the state before it (deletion state k: {match k, deletion k - 1}; block end: {last match,
  // last insertion, last deletion}) -- the lane before it, or the last lane of the wave before
  const bool chain_prev = role_del ? n_in > 1 : role_end ? n_in > 2 : false;
  const bool chain_xwave = chain_prev && (threadIdx.x & 63u) == 0;
  // (every lane runs the chain steps: -inf as the transition term of the lanes that take nothing from their neighbour -- and of the
  //  first lane of a wave, which takes its predecessor from LDS before the steps -- makes their candidate lose every comparison)
  const double lp_chain = role_del ? lp1 : lp2, lp_step = chain_prev && !chain_xwave ? lp_chain : NINF;
  const int chain_steps = __builtin_amdgcn_readfirstlane(min(63, max((int)set.max_mlen, SUB == 32 ? __shfl_xor((int)set.max_mlen, 32) : 0) - 1));  // (the same in every lane: a scalar loop count)
  const int chain_rounds = (int)set.chain_rounds;
  // (waves of a multi-wave model that hold no chain state skip the steps: they only cost issue slots of their SIMD)
  const bool wave_chain = __ballot(lp_step > NINF) != 0ull;

  HP_MARK(0);
  // the run-end state's predecessors (the block ends, in block order): the first BE_REG of them by index in registers
  const double lp_rs0 = l_lp[1], lp_rs1 = l_lp[S + 1];  // transition terms of the run start (state 1), evaluated by the run-end lane
  constexpr int BE_REG = 8;
  int be[BE_REG];
#pragma unroll
  for (int b = 0; b < BE_REG; ++b) be[b] = (int)l_blocks[1 * nb + (b < nb ? b : 0)];
  // ---- Viterbi fill (generate_mats, hmm_model.rs:99-114)
  // Two rounds of cross-lane traffic per column instead of four (register fill: crossbar fetches; LDS fill: barriers).  What the silent states of a column need is final early: the deletion chains
  // and block ends after the chain pass, and run end / run start / block starts are then plain functions of THOSE -- every lane
  // works them out for itself from one round of fetches (block ends, start state, its own block's end): the run end as before, the
  // run start behind it, and the start of its own block {run start, own block end}.  The same round fetches the predecessors of
  // the NEXT column's emitting states (emitting, deletion and block-end states: final by then); a predecessor that is a block
  // start or the run end is taken from the lane's own copy.  Same sums, same order, same strict '>'.  The topology this rests on
  // (builder.rs:4-173) is checked per job: block start = {run start, own block end}; an emitting state's silent predecessors are
  // deletion states, block ends, its OWN block's start (slot 0 or 1) or the run end (slot 0) -- anything else takes the loop
  // with one round per pass below.
  const int my_ms = my_blk >= 0 ? (int)l_blocks[0 * nb + my_blk] : 0, my_me = my_blk >= 0 ? (int)l_blocks[1 * nb + my_blk] : 0;
  const double ms_lp0 = my_blk >= 0 ? l_lp[my_ms] : NINF, ms_lp1 = my_blk >= 0 ? l_lp[S + my_ms] : NINF;
  bool bad = false, use_loc0 = false, use_loc1 = false;
  if (own_fill && act && my_blk >= 0) bad = model[set.off_nin + my_ms] != 2 || g_inst[0 * S + my_ms] != 1 || g_inst[1 * S + my_ms] != my_me;
  if (own_fill && act && level == 0 && n_in != 0xFF) {
    for (int b = 0; b < n_in && b < 4; ++b) {
      const int pb = b == 0 ? p0 : b == 1 ? p1 : b == 2 ? p2 : p3;
      const int pblk = pb < S ? (int)l_block[pb] : -1;
      const bool p_start = pblk >= 0 && pb == (int)l_blocks[0 * nb + pblk];
      if (pb == 1) bad = true;                                   // the run start feeds block starts only
      else if (pb == S - 2) { if (b == 0) use_loc0 = true; else bad = true; }
      else if (p_start) {
        if (pblk == my_blk && b == 0) use_loc0 = true;
        else if (pblk == my_blk && b == 1) use_loc1 = true;
        else bad = true;
      }
    }
  }
  const bool two_rounds = !(sync_n == 64 ? __ballot(bad) != 0ull : __syncthreads_or(bad ? 1 : 0) != 0) && !four_rounds;
  double* prev = sc0;
  double* cur = sc1;
  HP_FILL_DECL;
  int sym_next = 0;
  if (own_fill) {
  if constexpr (ONE_WAVE) {
    const int hw = (int)(threadIdx.x & 63u), lane_base = hw & ~(SUB - 1);
    const int a_q0 = (lane_base + q0) << 2, a_q1 = (lane_base + q1) << 2, a_q2 = (lane_base + q2) << 2, a_q3 = (lane_base + q3) << 2, a_st0 = lane_base << 2;
    int a_be[BE_REG];
#pragma unroll
    for (int b = 0; b < BE_REG; ++b) a_be[b] = (lane_base + be[b]) << 2;
    const double lp_re = l_lp[S - 2];  // the run end's transition term (one for all its predecessors' slot 0 ... see below)
    const int a_myend = (lane_base + my_me) << 2;
    const bool is_run_end = act && n_in == 0xFF, is_run_start = act && st == 1;
    if (two_rounds) {
      // (What a column costs the one wave is issue slots: a ds_bpermute_b32 of a permutation ~32 cycles, 14 when all lanes read one
      //  lane, an f64 add or compare ~16, tools/micro/bperm_cost.hip.  Tried on top of this loop and dropped, each slower or even:
      //  block ends by v_readlane, predecessor 0 by a DPP shift, the own block end picked from the block ends (each also on its own:
      //  6.0 -> 6.5 / 6.25 ms for a class of a cfg3 call), every "first maximum" as a tournament with -inf transition terms in the
      //  invalid slots -- fewer crossbar passes, more selects and uniform branches that split the column's one basic block.)
      double em_next = 0.0;
      const bool loc_is_re = act && use_loc0 && p0 == S - 2;
      double pe0 = NINF, pe1 = NINF, pe2 = NINF, pe3 = NINF;  // the predecessors' scores of the column before
      uint8_t* __restrict__ bp_col = bp;
      // (column 0 is a copy of its own: the start state's special case and the run start's first candidate cost no branch per column)
      auto column = [&](const int i, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // (symbol codes two columns ahead, emission terms one: neither LDS round trip is waited for in the column that starts it)
        if ((i % HMM_CODE_WINDOW) == 0) {
          hmm_sync(sync_n);
          win0 = i;
          for (int k = tid; k < HMM_CODE_WINDOW + 2 && i + k < L; k += nthr) l_seq[k] = (uint8_t)hmm_code(seq, i + k, L);
          hmm_sync(sync_n);
          sym_next = code_at(i);
          em_next = l_em[sym_next * S + st];
          sym_next = i + 1 < L ? code_at(i + 1) : 0;
        }
        const double em = em_next;
        em_next = l_em[__umul24((unsigned)sym_next, (unsigned)S) + st];
        sym_next = i + 2 < L ? code_at(i + 2) : 0;
        HP_FILL(0);
        double best = NINF;
        int bpi = 0xFF;
        if (act && level == 0) {
          if constexpr (FIRST) {
            if (n_in == 0 && em > NINF) { best = em; bpi = 0xFE; }  // the start state (hmm_model.rs:91-94)
          } else {
            const double v0 = (pe0 + lp0) + em, v1 = (pe1 + lp1) + em, v2 = (pe2 + lp2) + em, v3 = (pe3 + lp3) + em;
            if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
            if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
            if (n_in > 2 && v2 > best) { best = v2; bpi = 2; }
            if (n_in > 3 && v3 > best) { best = v3; bpi = 3; }
          }
        }
        double cur_v = best;
        HP_FILL(1);
        {  // round 1: the chains d0 <- d1 <- ... <- block end take this column's emitting states
          const double s0 = bperm_f64(a_q0, cur_v), s1 = bperm_f64(a_q1, cur_v);
          if (role_chain) {
            const double v0 = (s0 + lp0), v1 = (s1 + lp1);
            if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
            if (role_end && n_in > 1 && v1 > best) { best = v1; bpi = 1; }
          }
          double val = best, cand = NINF;
          for (int t = 0; wave_chain && t < chain_steps; ++t) {
            cand = (wave_shr1_f64(val) + lp_step);
            val = max_f64(cand, best);
          }
          if (cand > best) bpi = role_del ? 1 : 2;
          best = val;
          if (role_chain) cur_v = val;
        }
        HP_FILL(2);
        {  // round 2: block ends, start state, own block end -- and the next column's predecessors
          const double f0 = bperm_f64(a_q0, cur_v), f1 = bperm_f64(a_q1, cur_v), f2 = bperm_f64(a_q2, cur_v), f3 = bperm_f64(a_q3, cur_v);
          const double myend = bperm_f64(a_myend, cur_v);
          double re = NINF; int re_bp = 0xFF;
          auto block_ends = [&](auto n_const) {
            constexpr int N = decltype(n_const)::value;
            double e[N];
#pragma unroll
            for (int b = 0; b < N; ++b) e[b] = bperm_f64(a_be[b], cur_v);
#pragma unroll
            for (int b = 0; b < N; ++b) {
              const double v = (e[b] + lp_re);
              if (b < nb && v > re) { re = v; re_bp = b; }
            }
          };
          if (nb <= 2) block_ends(std::integral_constant<int, 2>());
          else if (nb <= 4) block_ends(std::integral_constant<int, 4>());
          else block_ends(std::integral_constant<int, BE_REG>());
          for (int b = BE_REG; b < nb; ++b) {
            const double v = (bperm_f64((lane_base + (int)l_blocks[1 * nb + b]) << 2, cur_v) + lp_re);
            if (v > re) { re = v; re_bp = b; }
          }
          // run start {start state, run end}: the start state has a score in column 0 only (no predecessors, hmm_model.rs:91-94), so
          // from column 1 on its candidate is -inf + lp = -inf and loses the strict '>': the run start is what the run end gives
          double br = NINF; int pr = 0xFF;
          if constexpr (FIRST) {
            const double start_now = bperm_f64(a_st0, cur_v);
            const double v0 = (start_now + lp_rs0), v1 = (re + lp_rs1);
            if (v0 > br) { br = v0; pr = 0; }
            if (v1 > br) { br = v1; pr = 1; }
          } else {
            const double v1 = (re + lp_rs1);
            if (v1 > br) { br = v1; pr = 1; }
          }
          double msv = NINF; int ms_bp = 0xFF;  // the start of my block: {run start, own block end}
          {
            const double v0 = (br + ms_lp0), v1 = (myend + ms_lp1);
            if (v0 > msv) { msv = v0; ms_bp = 0; }
            if (v1 > msv) { msv = v1; ms_bp = 1; }
          }
          if (is_run_end) bpi = re_bp;
          if (is_run_start) bpi = pr;
          if (role_start) bpi = ms_bp;
          const double loc = loc_is_re ? re : msv;
          pe0 = use_loc0 ? loc : f0; pe1 = use_loc1 ? msv : f1; pe2 = f2; pe3 = f3;
        }
        HP_FILL(4);
        if (act) bp_col[st] = (uint8_t)bpi;  // (a uniform base that moves by a column, the lane's state as the offset: no 64-bit address arithmetic per lane)
        bp_col += Spad;
        HP_FILL(5);
      };
      column(0, std::true_type());
      for (int i = 1; i < L; ++i) column(i, std::false_type());
    } else {
    double prev_v = NINF;
    double em_next = 0.0;
    for (int i = 0; i < L; ++i) {
      if ((i % HMM_CODE_WINDOW) == 0) {
        hmm_sync(sync_n);
        win0 = i;
        for (int k = tid; k < HMM_CODE_WINDOW + 1 && i + k < L; k += nthr) l_seq[k] = (uint8_t)hmm_code(seq, i + k, L);
        hmm_sync(sync_n);
        sym_next = code_at(i);
        em_next = l_em[sym_next * S + st];
      }
      const double em = em_next;
      if (i + 1 < L) { sym_next = code_at(i + 1); em_next = l_em[sym_next * S + st]; }  // (a column ahead: off this column's critical path)
      HP_FILL(0);
      double best = NINF;
      int bpi = 0xFF;
      {
        const double s0 = bperm_f64(a_q0, prev_v), s1 = bperm_f64(a_q1, prev_v), s2 = bperm_f64(a_q2, prev_v), s3 = bperm_f64(a_q3, prev_v);
        if (act && level == 0) {
          if (i == 0) {
            if (n_in == 0 && em > NINF) { best = em; bpi = 0xFE; }  // the start state (hmm_model.rs:91-94)
          } else {
            const double v0 = (s0 + lp0) + em, v1 = (s1 + lp1) + em, v2 = (s2 + lp2) + em, v3 = (s3 + lp3) + em;
            if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
            if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
            if (n_in > 2 && v2 > best) { best = v2; bpi = 2; }
            if (n_in > 3 && v3 > best) { best = v3; bpi = 3; }
          }
        }
      }
      double cur_v = best;  // emitting states: final; silent ones: -inf until their pass
      HP_FILL(1);
      {  // the chains d0 <- d1 <- ... <- block end (see the LDS variant below for the argument)
        const double s0 = bperm_f64(a_q0, cur_v), s1 = bperm_f64(a_q1, cur_v);
        if (role_chain) {
          const double v0 = (s0 + lp0), v1 = (s1 + lp1);
          if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
          if (role_end && n_in > 1 && v1 > best) { best = v1; bpi = 1; }
        }
        double val = best, cand = NINF;
        for (int t = 0; wave_chain && t < chain_steps; ++t) {
          cand = (wave_shr1_f64(val) + lp_step);
          val = cand > best ? cand : best;
        }
        if (cand > best) bpi = role_del ? 1 : 2;  // (never for the lanes whose transition term is -inf)
        best = val;
        if (role_chain) cur_v = val;
      }
      HP_FILL(2);
      {  // run end: the block ends in block order; the run start {start state, run end} behind it -- by every lane, kept by the two
        const double start_now = bperm_f64(a_st0, cur_v);
        double re = NINF; int re_bp = 0xFF;
        auto block_ends = [&](auto n_const) {
          constexpr int N = decltype(n_const)::value;
          double e[N];
#pragma unroll
          for (int b = 0; b < N; ++b) e[b] = bperm_f64(a_be[b], cur_v);
#pragma unroll
          for (int b = 0; b < N; ++b) {
            const double v = (e[b] + lp_re);
            if (b < nb && v > re) { re = v; re_bp = b; }
          }
        };
        if (nb <= 2) block_ends(std::integral_constant<int, 2>());
        else if (nb <= 4) block_ends(std::integral_constant<int, 4>());
        else block_ends(std::integral_constant<int, BE_REG>());
        for (int b = BE_REG; b < nb; ++b) {
          const double v = (bperm_f64((lane_base + (int)l_blocks[1 * nb + b]) << 2, cur_v) + lp_re);
          if (v > re) { re = v; re_bp = b; }
        }
        double br = NINF; int pr = 0xFF;
        const double v0 = (start_now + lp_rs0), v1 = (re + lp_rs1);
        if (v0 > br) { br = v0; pr = 0; }
        if (v1 > br) { br = v1; pr = 1; }
        if (is_run_end) { cur_v = re; best = re; bpi = re_bp; }
        if (is_run_start) { cur_v = br; best = br; bpi = pr; }
      }
      HP_FILL(3);
      {  // block starts: {run start, own block end}
        const double s0 = bperm_f64(a_q0, cur_v), s1 = bperm_f64(a_q1, cur_v);
        if (role_start) {
          const double v0 = (s0 + lp0), v1 = (s1 + lp1);
          if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
          if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
          cur_v = best;
        }
      }
      HP_FILL(4);
      if (act) bp[(size_t)i * Spad + st] = (uint8_t)bpi;
      prev_v = cur_v;
      HP_FILL(5);
    }
    }
  } else if (two_rounds) {
    // The LDS fill with two barriers per column (plus one per extra chain round): emitting states | chains | -- and what used to be
    // two more passes with a barrier each (run end + run start by one lane, then the block starts) every lane works out for itself
    // from the block ends, the start state and its own block's end, as in the register fill above.  Run end, run start and block
    // starts are not written to the score columns any more: the only readers were those passes and the next column's emitting states,
    // which take them from their own copy.
    const double lp_re = l_lp[S - 2];
    const bool is_run_end = act && n_in == 0xFF, is_run_start = act && st == 1;
    const bool loc_is_re = act && use_loc0 && p0 == S - 2;
    double loc = NINF, msv_prev = NINF;  // my copies of the column before: run end or own block start | own block start
    uint8_t* __restrict__ bp_col = bp;
    for (int i = 0; i < L; ++i) {  // (column 0 as a copy of its own, as in the register fill, measured slower here: 5.45 -> 5.64 ms)
      if ((i % HMM_CODE_WINDOW) == 0) {
        hmm_sync(sync_n);
        win0 = i;
        for (int k = tid; k < HMM_CODE_WINDOW + 1 && i + k < L; k += nthr) l_seq[k] = (uint8_t)hmm_code(seq, i + k, L);
        hmm_sync(sync_n);
        sym_next = code_at(i);
      }
      const int sym = sym_next;
      if (i + 1 < L) sym_next = code_at(i + 1);
      HP_FILL(0);
      double best = NINF;
      int bpi = 0xFF;
      if (act && level == 0) {
        const double em = l_em[sym * S + st];
        if (i == 0) {
          if (n_in == 0 && em > NINF) { best = em; bpi = 0xFE; }  // the start state (hmm_model.rs:91-94)
        } else {
          const double r0 = prev[q0], r1 = prev[q1], s2 = prev[q2], s3 = prev[q3];
          const double s0 = use_loc0 ? loc : r0, s1 = use_loc1 ? msv_prev : r1;
          const double v0 = (s0 + lp0) + em, v1 = (s1 + lp1) + em, v2 = (s2 + lp2) + em, v3 = (s3 + lp3) + em;
          if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
          if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
          if (n_in > 2 && v2 > best) { best = v2; bpi = 2; }
          if (n_in > 3 && v3 > best) { best = v3; bpi = 3; }
        }
        cur[st] = best;
      }
      hmm_sync(sync_n);
      HP_FILL(1);
      if (role_chain) {
        const double s0 = cur[q0], s1 = cur[q1];
        const double v0 = (s0 + lp0), v1 = (s1 + lp1);
        if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
        if (role_end && n_in > 1 && v1 > best) { best = v1; bpi = 1; }
      }
      {
        const double own = best;
        int own_bp = bpi;
        double val = best, cand = NINF;
        for (int r = 0; r < chain_rounds; ++r) {
          if (chain_xwave) {
            cand = (cur[st - 1] + lp_chain);
            best = cand > own ? cand : own;
            bpi = cand > own ? (role_del ? 1 : 2) : own_bp;
            val = best;
          }
          for (int t = 0; wave_chain && t < chain_steps; ++t) {
            cand = (wave_shr1_f64(val) + lp_step);
            val = max_f64(cand, best);
          }
          if (role_chain) cur[st] = val;
          if (r + 1 < chain_rounds) hmm_sync(sync_n);
        }
        if (cand > best) bpi = role_del ? 1 : 2;
        best = val;
      }
      hmm_sync(sync_n);
      HP_FILL(2);
      {
        const double myend = cur[my_me];
        double re = NINF; int re_bp = 0xFF;
        auto block_ends = [&](auto n_const) {
          constexpr int N = decltype(n_const)::value;
          double e[N];
#pragma unroll
          for (int b = 0; b < N; ++b) e[b] = cur[be[b]];
#pragma unroll
          for (int b = 0; b < N; ++b) {
            const double v = (e[b] + lp_re);
            if (b < nb && v > re) { re = v; re_bp = b; }
          }
        };
        if (nb <= 2) block_ends(std::integral_constant<int, 2>());
        else if (nb <= 4) block_ends(std::integral_constant<int, 4>());
        else block_ends(std::integral_constant<int, BE_REG>());
        for (int b = BE_REG; b < nb; ++b) {
          const double v = (cur[l_blocks[1 * nb + b]] + lp_re);
          if (v > re) { re = v; re_bp = b; }
        }
        double br = NINF; int pr = 0xFF;  // (the start state has a score in column 0 only: see the register fill)
        if (i == 0) {
          const double start_now = cur[0];
          const double v0 = (start_now + lp_rs0), v1 = (re + lp_rs1);
          if (v0 > br) { br = v0; pr = 0; }
          if (v1 > br) { br = v1; pr = 1; }
        } else {
          const double v1 = (re + lp_rs1);
          if (v1 > br) { br = v1; pr = 1; }
        }
        double msv = NINF; int ms_bp = 0xFF;  // the start of my block: {run start, own block end}
        {
          const double v0 = (br + ms_lp0), v1 = (myend + ms_lp1);
          if (v0 > msv) { msv = v0; ms_bp = 0; }
          if (v1 > msv) { msv = v1; ms_bp = 1; }
        }
        if (is_run_end) bpi = re_bp;
        if (is_run_start) bpi = pr;
        if (role_start) bpi = ms_bp;
        loc = loc_is_re ? re : msv; msv_prev = msv;
      }
      HP_FILL(4);
      if (act) bp_col[st] = (uint8_t)bpi;
      bp_col += Spad;
      double* t = prev; prev = cur; cur = t;
      HP_FILL(5);
    }
  } else
  for (int i = 0; i < L; ++i) {
    if ((i % HMM_CODE_WINDOW) == 0) {  // next window of symbol codes (one column more than the window: the look-ahead below)
      hmm_sync(sync_n);
      win0 = i;
      for (int k = tid; k < HMM_CODE_WINDOW + 1 && i + k < L; k += nthr) l_seq[k] = (uint8_t)hmm_code(seq, i + k, L);
      hmm_sync(sync_n);
      sym_next = code_at(i);
    }
    const int sym = sym_next;
    if (i + 1 < L) sym_next = code_at(i + 1);  // (fetched a column ahead: its LDS round trip is off this column's critical path)
    HP_FILL(0);
    double best = NINF;
    int bpi = 0xFF;
    if (act && level == 0) {
      const double em = l_em[sym * S + st];
      if (i == 0) {
        if (n_in == 0 && em > NINF) { best = em; bpi = 0xFE; }  // the start state (hmm_model.rs:91-94)
      } else {
        // all four predecessor scores are fetched first (one LDS wait instead of four); same sums, same strict '>' in the same order
        const double s0 = prev[q0], s1 = prev[q1], s2 = prev[q2], s3 = prev[q3];
        const double v0 = (s0 + lp0) + em, v1 = (s1 + lp1) + em, v2 = (s2 + lp2) + em, v3 = (s3 + lp3) + em;
        if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
        if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
        if (n_in > 2 && v2 > best) { best = v2; bpi = 2; }
        if (n_in > 3 && v3 > best) { best = v3; bpi = 3; }
      }
      cur[st] = best;
    }
    hmm_sync(sync_n);
    HP_FILL(1);
    // (Silent states emit nothing: the reference adds an emission term of 0.0 to their sums, which changes no value -- scores are sums of
    //  logarithms of probabilities, never -0.0 -- and is left out here.)
    // Silent states of the column in three passes instead of one per topological level (motif length + 3 of them).  First the chains
    // d0 <- d1 <- ... <- block end of all motif blocks at once: a chain state is max(what its other predecessors give, the state before
    // it + lp) with ties to the former, and the state before it is the LANE before it -- every lane takes its neighbour's value
    // (DPP wave_shr:1: no LDS round trip), and after k steps the k-th state of a chain has its final value (steps beyond that
    // recompute it from the same inputs).  A chain that runs over a wave boundary takes another round: the first lane of a wave reads
    // the state before it from LDS.  Then the run-end lane takes the maximum over the block ends and evaluates the run start behind
    // it; then every block start.  Same sums, same predecessor order, same strict '>' as level by level (any topological order gives
    // identical values, hmm_model.rs:206-240).
    if (role_chain) {
      const double s0 = cur[q0], s1 = cur[q1];
      const double v0 = (s0 + lp0), v1 = (s1 + lp1);
      if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
      if (role_end && n_in > 1 && v1 > best) { best = v1; bpi = 1; }
    }
    {
      const double own = best;  // what the predecessors other than the chain's give
      int own_bp = bpi;
      double val = best, cand = NINF;
      for (int r = 0; r < chain_rounds; ++r) {
        if (chain_xwave) {  // the state before sits in the wave before: final after the round before (round 0: a throw-away value)
          cand = (cur[st - 1] + lp_chain);
          best = cand > own ? cand : own;
          bpi = cand > own ? (role_del ? 1 : 2) : own_bp;
          val = best;
        }
        for (int t = 0; wave_chain && t < chain_steps; ++t) {
          cand = (wave_shr1_f64(val) + lp_step);
          val = cand > best ? cand : best;
        }
        if (role_chain) cur[st] = val;
        if (r + 1 < chain_rounds) hmm_sync(sync_n);
      }
      if (cand > best) bpi = role_del ? 1 : 2;  // (never for the lanes whose transition term is -inf)
      best = val;
    }
    hmm_sync(sync_n);
    HP_FILL(2);
    if (act && n_in == 0xFF) {  // run end: block ends in block order; then the run start {start state, run end}
      const double start_now = cur[0];  // (the start state of this column, for the run start below: fetched with the block ends)
      // (their indices sit in registers: one round trip for all of them; a model of one or two motifs -- two or three blocks, nearly every
      //  locus of a genome-wide catalog -- fetches and compares only what it has: 8 fetches and compares per column were 200 of its 2 400 cycles)
      auto block_ends = [&](auto n_const) {
        constexpr int N = decltype(n_const)::value;
        double e[N];
#pragma unroll
        for (int b = 0; b < N; ++b) e[b] = cur[be[b]];
#pragma unroll
        for (int b = 0; b < N; ++b) {
          const double v = (e[b] + lp0);
          if (b < nb && v > best) { best = v; bpi = b; }
        }
      };
      if (nb <= 2) block_ends(std::integral_constant<int, 2>());
      else if (nb <= 4) block_ends(std::integral_constant<int, 4>());
      else block_ends(std::integral_constant<int, BE_REG>());
      for (int b = BE_REG; b < nb; ++b) {
        const double v = (cur[l_blocks[1 * nb + b]] + lp0);
        if (v > best) { best = v; bpi = b; }
      }
      cur[st] = best;
      double br = NINF; int pr = 0xFF;
      const double v0 = (start_now + lp_rs0), v1 = (best + lp_rs1);
      if (v0 > br) { br = v0; pr = 0; }
      if (v1 > br) { br = v1; pr = 1; }
      cur[1] = br; l_bpcol[1] = (uint8_t)pr;
    }
    hmm_sync(sync_n);
    HP_FILL(3);
    if (role_start) {  // block start: {run start, own block end}
      const double s0 = cur[q0], s1 = cur[q1];
      const double v0 = (s0 + lp0), v1 = (s1 + lp1);
      if (n_in > 0 && v0 > best) { best = v0; bpi = 0; }
      if (n_in > 1 && v1 > best) { best = v1; bpi = 1; }
      cur[st] = best;
    }
    hmm_sync(sync_n);
    HP_FILL(4);
    if (act && st == 1) bpi = l_bpcol[1];  // the run start was evaluated by the run-end lane
    if (act) bp[(size_t)i * Spad + st] = (uint8_t)bpi;
    double* t = prev; prev = cur; cur = t;
    HP_FILL(5);
  }
  }  // (fill)
  HP_FILL_END;
  HP_MARK(1);
  // long alleles are traced back by hmm_traceback_long_kernel (many waves per allele, behind this launch on the same stream)
  if (long_list && job.map_off && L >= long_min) {
    hmm_sync_mem(sync_n);
    if (tid == 0) long_list[1 + atomicAdd(long_list, 1u)] = jidx;
    return;
  }
  if (tid == 0) {
    tb_state = S - 1; tb_idx = L - 1; tb_done = 0; tb_npath = 0; tb_nvisit = 0; tb_edit = 0; tb_ref = 0; tb_next = -1; tb_vb1 = 0;
  }
  hmm_sync_mem(sync_n);  // the back-pointer columns written by all waves are read back from here on

  // ---- traceback (hmm_model.rs:125-142) fused with get_events/calc_purity (events.rs:17-86, purity.rs:6-41)
  //      and motif-visit collection (operations.rs:26-40); back-pointer columns are staged through LDS.
  //      Thread 0 only CHASES the back-pointers (state, column -> predecessor: two LDS round trips and a dozen instructions per step)
  //      and notes the states it passes, HMM_REC at a time; what each step means -- its events, its part of the edit count, the motif
  //      visit it closes -- is then worked out for all noted steps at once, one lane per step.  (Everything in one loop on one lane was
  //      90 instructions per step: 750 cycles, a third of the kernel.)  What a step needs from its neighbours is little: the state
  //      walked just before it (the implied leading deletions of a block start) and the column of the last block end before it (the
  //      bases of the visit a block start closes): a lane shift and a ballot.
  // rows of the back-pointer workspace: one byte per state (this kernel's own fill, the round-5 layout of the position-per-lane fill) or
  // one byte per lane of the job's group (hmm_fill_ppl_kernel<G, true>)
  const bool packed = ppl_filled && ppl_packed && set.ppl_lanes != 0u;
  const int rstride = packed ? hmm_ppl_group(set.ppl_lanes) : Spad;
  const int cols_per_chunk = packed ? hmm_stage_bytes(Spad) / rstride - 2 : max(1, hmm_stage_bytes(Spad) / Spad);
  int& tb_loc = tb[11];  // hmm_bp_loc of tb_state
  if (tid == 0) tb_loc = 0;  // (the end state: a constant 0)
  uint16_t* pbuf = path ? path + job.path_off : nullptr;
  uint32_t* const g_vis = visit_ws + job.visit_off;  // visits HMM_VIS_LDS, HMM_VIS_LDS + 1, ... at their own index
  const int pcap = (int)job.path_cap;
  constexpr int HMM_REC = SUB == 32 ? 32 : 64;  // steps per round: the lanes of the job's first wave
  int &tb_nrec = tb[9], &tb_more = tb[10];      // steps noted in this round; 1: the chunk has more, 0: it is exhausted, 2: the walk is over
  const int hwlane = (int)(threadIdx.x & 63u);
  const unsigned long long gmask = SUB == 32 ? (0xFFFFFFFFull << (hwlane & 32)) : ~0ull;  // the lanes of my job in this wave
  const unsigned long long below = gmask & ((1ull << hwlane) - 1ull);
  while (true) {
    if (tb_done) break;
    const int c1 = tb_idx + 1, c0 = packed ? (max(0, c1 - cols_per_chunk) & ~1) : max(0, c1 - cols_per_chunk);  // (rows of 8 bytes: an even column starts a 16-byte piece)
    {
      const uint4* src = reinterpret_cast<const uint4*>(bp + (size_t)c0 * rstride);
      uint4* dst = reinterpret_cast<uint4*>(l_stage);
      const int n16 = ((c1 - c0) * rstride + 15) / 16;
      for (int i = tid; i < n16; i += nthr) dst[i] = src[i];
      // ... and the symbol codes of the same columns, plus those a motif copy starting in the last of them reaches into
      win0 = c0;
      for (int k = tid; k < c1 - c0 + HMM_CODE_PAD && c0 + k < L; k += nthr) l_seq[k] = (uint8_t)hmm_code(seq, c0 + k, L);
    }
    hmm_sync(sync_n);
    for (;;) {
      if (tid == 0) {  // ---- the chase
        int state = tb_state, idx = tb_idx, n = 0;
        int row = (idx - c0) * rstride;  // offset of column idx in the staged chunk
        int emits = (int)((l_info[state] >> 3) & 1u);
        uint32_t loc = (uint32_t)tb_loc;
        while (state != 0 && idx >= c0 && n < HMM_REC) {
          l_rec[2 * n] = (uint32_t)state; l_rec[2 * n + 1] = (uint32_t)idx; ++n;
          const uint4 pred4 = *reinterpret_cast<const uint4*>(l_inst + 4 * state);  // all four predecessors: no second round trip behind b
          int b;
          if (packed) {
            b = (int)hmm_bp_unpack(l_stage[row + (int)(loc & 63u)], loc);
            if (state == S - 2) b = (int)hmm_bp_run_end(*reinterpret_cast<const uint32_t*>(l_stage + row));
          } else b = l_stage[row + state];
          uint32_t pe = (b & 2) ? ((b & 1) ? pred4.w : pred4.z) : ((b & 1) ? pred4.y : pred4.x);  // predecessor | its "emits" bit << 15 | its hmm_bp_loc << 16
          if (state == S - 2) { const uint32_t be_ = l_blocks[1 * nb + b]; pe = be_ | ((uint32_t)(l_flags[be_] & 1) << 15) | (hmm_bp_loc_block_end(b, nb, l_blocks) << 16); }  // the run end: from a block end
          if (emits) { --idx; row -= rstride; }
          emits = (int)((pe >> 15) & 1u);
          state = (int)(pe & 0x7FFFu);
          loc = pe >> 16;
        }
        tb_state = state; tb_idx = idx; tb_nrec = n; tb_loc = (int)loc;
        tb_more = state == 0 ? 2 : (idx >= c0 ? 1 : 0);
      }
      hmm_sync(sync_n);
      const int more = tb_more;  // (read before the next barrier: thread 0 writes it again right behind that one)
      if (tid < HMM_REC) {  // ---- what the noted steps mean (events.rs:17-86, purity.rs:6-41, operations.rs:26-57), one lane per step
        const int n = tb_nrec, np0 = tb_npath, nv0 = tb_nvisit, nxt0 = tb_next, vb0 = tb_vb1;
        const bool valid = tid < n;
        const int state = valid ? (int)l_rec[2 * tid] : 0, idx = valid ? (int)l_rec[2 * tid + 1] : 0;
        const uint32_t inf = valid ? l_info[state] : 0u;
        const int kind = (int)(inf & 7u), blk = (int)((inf >> 8) & 0xFFu), expected = (int)((inf >> 16) & 0xFFu);
        if (valid && pbuf && np0 + tid < pcap) pbuf[pcap - 1 - (np0 + tid)] = (uint16_t)state;
        // the state walked just before this one (the step before: the lane before)
        const int up = __shfl_up(state, 1);
        const int nxt = tid == 0 ? nxt0 : up;
        // MotifStart (1) adds the implied leading deletions, Skip (3) / Mismatch / Ins (5) / Del (6) are edits, Skip / Match-state /
        // Del consume a reference base
        const int qbase = valid ? hmm_code_char(code_at(idx)) : 0;
        const int dels = kind == 1 ? nxt - state - 1 : 0;
        const int mism = kind == 4 && !(qbase == expected || expected == 'N');  // events.rs:66-73
        int edit = valid ? dels + (kind == 3) + mism + (kind == 5) + (kind == 6) : 0;
        int ref = valid ? dels + (kind == 3) + (kind == 4) + (kind == 6) : 0;
        // the last block end (2) walked before this step: the bases of the visit a block start (1) closes are query[idx .. vb1)
        const unsigned long long ends = __ballot(valid && kind == 2) & gmask, starts = __ballot(valid && kind == 1) & gmask;
        const unsigned long long ends_below = ends & below;
        const int src_end = ends_below ? 63 - (int)__builtin_clzll(ends_below) : hwlane;
        const int idx_end = __shfl(idx, src_end);
        const int vb1 = ends_below ? idx_end : vb0;
        if (valid && kind == 1) {  // a motif visit
          // remove_imperfect_motifs(.., 6) (operations.rs:45-57): only copies of STR motifs can be dropped -- short ones, and ones
          // whose bases differ from the motif (its bases are columns idx + 1 .. idx + mlen: in the window)
          uint32_t drop = 0;
          const int mlen = (int)l_blocks[2 * nb + blk];
          if (blk != nb - 1 && mlen <= 6) {
            if (vb1 - idx < mlen) drop = 1;
            else {
              const uint8_t* mot = motif_bytes + l_blocks[3 * nb + blk];
              for (int j = 0; j < mlen; ++j) {
                const int obs = hmm_code_char(code_at(idx + j + 1));
                if (mot[j] != 'N' && obs != mot[j]) drop = 1;
              }
            }
          }
          const int nv = nv0 + (int)__builtin_popcountll(starts & below);
          uint32_t* vrec = nv < HMM_VIS_LDS ? l_vis + 3 * nv : g_vis + 3 * (size_t)nv;
          vrec[0] = (uint32_t)blk | (drop << 15); vrec[1] = (uint32_t)idx; vrec[2] = (uint32_t)vb1;
        }
        // sums over the round (butterfly inside the job's lanes)
#pragma unroll
        for (int o = HMM_REC / 2; o >= 1; o >>= 1) { edit += __shfl_xor(edit, o); ref += __shfl_xor(ref, o); }
        const int src_last_end = ends ? 63 - (int)__builtin_clzll(ends) : hwlane;
        const int idx_last_end = __shfl(idx, src_last_end);
        const int last_state = __shfl(state, (hwlane & ~(HMM_REC - 1)) + max(n - 1, 0));
        if (tid == 0) {
          int np = np0 + n;
          if (more == 2) { if (pbuf && np < pcap) pbuf[pcap - 1 - np] = 0; ++np; tb_done = 1; }
          tb_npath = np; tb_nvisit = nv0 + (int)__builtin_popcountll(starts); tb_edit += edit; tb_ref += ref;
          if (n > 0) tb_next = last_state;
          if (ends) tb_vb1 = idx_last_end;
        }
      }
      hmm_sync(sync_n);
      if (more != 1) break;
    }
  }
  const int np = tb_npath;
  HP_MARK(2);
  // ---- state path: shift the reversed tail to the front (forward order)
  if (pbuf) {
    const int n = min(np, pcap), shift = pcap - n;
    hmm_sync_mem(sync_n);  // (the path was written by thread 0)
    for (int base = 0; base < n; base += nthr) {
      const int f = base + tid;
      uint16_t v = 0;
      if (f < n) v = pbuf[shift + f];
      hmm_sync_mem(sync_n);
      if (f < n) pbuf[f] = v;
      hmm_sync_mem(sync_n);
    }
  }
  // ---- decode (thread 0): purity, label_motifs over the kept copies, skip filter, counts, collapse.  Visits were recorded back to
  //      front: the last ones recorded (the first of the allele) sit in global memory and come through LDS in chunks.
  int ns = 0, cum = 0, last_motif = -1, last_end = -1;
  int32_t* const sp = spans3 + 3 * job.span_off;
  auto take_visit = [&](const uint32_t* vrec) {
    const int blk = (int)(vrec[0] & 0x7FFFu), b0 = (int)vrec[1], b1 = (int)vrec[2];
    const bool keep = (vrec[0] >> 15) == 0;
    const int cnt = b1 - b0;
    const int start = cum, end = cum + cnt;
    cum = end;
    const int motif = keep ? blk : nb - 1;
    if (motif < n_motifs) {
      l_cnt[motif] += 1;
      if (ns > 0 && last_motif == motif && last_end == start) { sp[3 * (ns - 1) + 2] = end; }
      else { sp[3 * ns + 0] = motif; sp[3 * ns + 1] = start; sp[3 * ns + 2] = end; ++ns; last_motif = motif; }
      last_end = end;
    }
  };
  if (tid == 0) {
    if (path_len) path_len[job.job_index] = (uint32_t)np;
    const int edit = tb_edit, mx = max(tb_ref, qlen);
    purity[job.job_index] = ((double)mx - (double)edit) / (double)mx;
    if (edit_out) edit_out[job.job_index] = edit;
    if (maxd_out) maxd_out[job.job_index] = mx;
  }
  int v = tb_nvisit - 1;
  constexpr int VIS_CHUNK = HMM_STAGE_BYTES / 12;
  uint32_t* const l_vchunk = reinterpret_cast<uint32_t*>(l_stage);  // (the staging window of the back-pointers is free now)
  while (v >= HMM_VIS_LDS) {
    const int n = min(v - HMM_VIS_LDS + 1, VIS_CHUNK), v0 = v - n + 1;
    hmm_sync_mem(sync_n);  // (the visits were written by thread 0; the chunk before has been consumed)
    for (int k = tid; k < 3 * n; k += nthr) l_vchunk[k] = g_vis[3 * (size_t)v0 + k];
    hmm_sync(sync_n);
    if (tid == 0) for (int k = n - 1; k >= 0; --k) take_visit(l_vchunk + 3 * k);
    v -= n;
  }
  if (tid == 0) {
    for (; v >= 0; --v) take_visit(l_vis + 3 * v);
    n_spans[job.job_index] = (uint32_t)ns;
  }
  hmm_sync(sync_n);
  for (int m = tid; m < n_motifs; m += nthr) counts[job.count_off + m] = l_cnt[m];
  HP_MARK(3);
}

#include "hmm_ppl.hpp"

// ---- trace-back of LONG alleles on many waves (DESIGN_HISTORY 7.3: "block-wise composed trace-back").  The chase of the back-pointers
// is a serial chain of dependent look-ups -- a third of a long allele's time on one lane.  Here the columns are cut into chunks of
// HMM_LONG_CHUNK; (A) for every chunk and EVERY state it could be entered in, a wave (lane = entry state) walks the chunk and notes where
// the walk leaves it and what it passed (steps, motif visits, the last block end, the last state); (B) one thread strings the chunks
// together from the end state -- a look-up per chunk; (C) every chunk is walked again from its now known entry state with the full
// decoding of the steps (events, purity counts, motif visits: the code of hmm_viterbi_kernel), the waves of the workgroup side by side.
// Exact by construction: (C) walks the very path the serial chase walks, and what a step needs from the steps before it (the state
// walked last, the column of the last block end, how many steps / visits came before) is handed over by (B).
struct HmmChunkMap { uint16_t exit_state, last_state, n_steps, n_starts; int32_t last_end; };  // 12 bytes = 3 words
struct HmmChunkRec { uint32_t entry, np0, nv0; int32_t vb0, nxt0; uint32_t pad[3]; };          // 8 words
constexpr int HMM_LONG_THREADS = 1024, HMM_LONG_STG = 2048;
constexpr int HMM_LONG_MAP_LDS = 48 * 1024;  // chunk maps and records that fit stay in LDS (a 10-kb allele of a 16-state model: 35 KB), else in the job's workspace

__global__ void __launch_bounds__(HMM_LONG_THREADS) hmm_traceback_long_kernel(
    const HmmJobDev* __restrict__ jobs, const HmmSetDev* __restrict__ sets, const uint8_t* __restrict__ model, const uint8_t* __restrict__ seq_blob,
    const uint8_t* __restrict__ bp_ws, uint32_t* __restrict__ visit_ws, uint16_t* __restrict__ path, uint32_t* __restrict__ path_len,
    int32_t* __restrict__ spans3, uint32_t* __restrict__ n_spans, uint32_t* __restrict__ counts, double* __restrict__ purity,
    int32_t* __restrict__ edit_out, int32_t* __restrict__ maxd_out, const uint32_t* __restrict__ long_list, const int phase_arg, const int G) {
  const int phase = phase_arg & 0xFF;
  const bool ppl_packed = ((phase_arg >> 8) & 1) != 0;  // the position-per-lane fill wrote rows of one byte per lane (hmm_fill_ppl_kernel<G, true>)
  // phase 0: the whole trace-back by one workgroup per allele.  Phases 1 / 2 / 3 are the same code as three launches with G workgroups
  // per allele in (A) and (C) -- an allele's chunks are independent there, and a class has few long alleles for 256 CUs: 1: (A) with
  // the maps in the job's workspace; 2: (B) by one workgroup; 3: (C), the totals in the four words in front of the maps, and the
  // workgroup that finishes last does the rest (path order, purity, visits).
  extern __shared__ __align__(16) unsigned char lds_long[];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = HMM_LONG_THREADS / 64;
  const int part = (int)(blockIdx.x % (unsigned)G);
  __shared__ int l_last;
  for (uint32_t li = blockIdx.x / (unsigned)G; li < long_list[0]; li += gridDim.x / (unsigned)G) {
    __syncthreads();
    const HmmJobDev job = jobs[long_list[1 + li]];
    const HmmSetDev set = sets[job.set];
    const int S = (int)set.S, nb = (int)set.n_blocks, n_motifs = nb - 1;
    const int qlen = (int)job.seq_len, L = qlen + 2;
    const int Spad = (S + 15) & ~15;
    const int C = HMM_LONG_CHUNK, n_chunks = (L + C - 1) / C;
    // ---- LDS: totals | tables | per wave: staged back-pointer columns, the steps of a round
    int* tot = reinterpret_cast<int*>(lds_long);  // [0] edit, [1] ref, [2] np, [3] nv
    uint32_t* l_inst = reinterpret_cast<uint32_t*>(lds_long + 64);                       // [S][4] hmm_pred_entry
    uint32_t* l_info = l_inst + 4 * S;                                                   // [S]
    uint32_t* l_blocks = l_info + S;                                                     // [4][nb]
    uint32_t* l_cnt = l_blocks + 4 * nb;                                                 // [nb]
    uint8_t* l_flags = reinterpret_cast<uint8_t*>(l_cnt + nb);                           // [S]
    const int mot_bytes = (S - 7 - n_motifs) / 3;
    uint8_t* l_mot = l_flags + ((S + 3) & ~3);
    unsigned char* wave_base = lds_long + ((64 + (size_t)16 * S + 4 * S + 16 * nb + 4 * nb + ((S + 3) & ~3) + ((mot_bytes + 15) & ~15) + 15) & ~(size_t)15);
    uint8_t* l_stage = wave_base + (size_t)wave * (HMM_LONG_STG + 512);
    uint32_t* const l_map = reinterpret_cast<uint32_t*>(wave_base + (size_t)NW * (HMM_LONG_STG + 512));
    const bool map_lds = phase == 0 && ((size_t)3 * S + 8) * (size_t)n_chunks * 4 <= (size_t)HMM_LONG_MAP_LDS;
    uint32_t* const hdr = visit_ws + job.map_off - 4;                                    // [0] edit, [1] ref, [2] workgroups done (phase 3)
    uint32_t* l_rec = reinterpret_cast<uint32_t*>(l_stage + HMM_LONG_STG);               // [64][2]
    const uint16_t* g_inst = reinterpret_cast<const uint16_t*>(model + set.off_inst);
    const int16_t* g_block = reinterpret_cast<const int16_t*>(model + set.off_block);
    const uint32_t* g_blocks = reinterpret_cast<const uint32_t*>(model + set.off_blocks);
    const uint8_t* g_motifs = model + set.off_motifs;
    for (int i = tid; i < 4 * S; i += HMM_LONG_THREADS) l_inst[4 * (i % S) + i / S] = hmm_pred_entry(g_inst[i], S, nb, model + set.off_flags, g_block, g_blocks);
    for (int i = tid; i < S; i += HMM_LONG_THREADS) l_flags[i] = model[set.off_flags + i];
    for (int i = tid; i < 4 * nb; i += HMM_LONG_THREADS) l_blocks[i] = g_blocks[i];
    for (int i = tid; i < mot_bytes; i += HMM_LONG_THREADS) l_mot[i] = g_motifs[i];
    for (int i = tid; i < nb; i += HMM_LONG_THREADS) l_cnt[i] = 0;
    for (int m = tid; m < n_motifs; m += HMM_LONG_THREADS) counts[job.count_off + m] = 0;
    if (tid < 4) tot[tid] = 0;
    __syncthreads();
    for (int st = tid; st < S; st += HMM_LONG_THREADS) {  // the trace-back word of a state (as in hmm_viterbi_kernel)
      const int blk = (int)g_block[st];
      uint32_t kind = 0, expected = 0;
      if (blk >= 0) {
        const int bstart = (int)l_blocks[0 * nb + blk], bend = (int)l_blocks[1 * nb + blk];
        if (st == bstart) kind = 1;
        else if (st == bend) kind = 2;
        else if (blk == nb - 1) kind = 3;
        else {
          const int mlen = (int)l_blocks[2 * nb + blk], off = st - bstart - 1, k = off / mlen;
          kind = 4u + (uint32_t)k;
          if (k == 0) expected = l_mot[l_blocks[3 * nb + blk] + off];
        }
      }
      l_info[st] = kind | ((uint32_t)(l_flags[st] & 1) << 3) | ((uint32_t)(blk & 0xFF) << 8) | (expected << 16);
    }
    __syncthreads();
    const uint8_t* __restrict__ seq = seq_blob + job.seq_off;
    const uint8_t* __restrict__ bp = bp_ws + job.bp_off;
    uint32_t* const g_vis = visit_ws + job.visit_off;
    uint32_t* const g_map = map_lds ? l_map : visit_ws + job.map_off;                 // [n_chunks][S] HmmChunkMap
    HmmChunkRec* const g_crec = reinterpret_cast<HmmChunkRec*>(g_map + (size_t)3 * S * n_chunks);  // [n_chunks]
    const bool packed = ppl_packed && set.ppl_lanes != 0u;
    const int rstride = packed ? hmm_ppl_group(set.ppl_lanes) : Spad;   // bytes per column (rows of one byte per state, or per lane of the job's group)
    const int sub_cols = max(1, HMM_LONG_STG / rstride);
    // the back-pointer of `state` (loc: its hmm_bp_loc) in the staged row at `row`
    auto bp_at = [&](int row, int state, uint32_t loc) -> int {
      if (!packed) return (int)l_stage[row + state];
      if (state == S - 2) return (int)hmm_bp_run_end(*reinterpret_cast<const uint32_t*>(l_stage + row));
      return (int)hmm_bp_unpack(l_stage[row + (int)(loc & 63u)], loc);
    };
    // one step of the chase: the predecessor of `state` in column `idx` (b: its back-pointer)
    // (all four predecessor entries of the state are read before its back-pointer is known -- one LDS round trip per step, not two)
    auto pred_of = [&](int state, int b, const uint4& p4) -> uint32_t {
      uint32_t pe = (b & 2) ? ((b & 1) ? p4.w : p4.z) : ((b & 1) ? p4.y : p4.x);
      if (state == S - 2) { const int bb = b < nb ? b : 0; const uint32_t be_ = l_blocks[1 * nb + bb]; pe = be_ | ((uint32_t)(l_flags[be_] & 1) << 15) | (hmm_bp_loc_block_end(bb, nb, l_blocks) << 16); }
      return pe;
    };
    // wave-cooperative staging of the back-pointer columns [c0, c1) (16-byte pieces; c0 is a multiple of the chunk length: 16-byte aligned rows)
    auto stage_cols = [&](int c0, int c1) {
      const uint4* src = reinterpret_cast<const uint4*>(bp + (size_t)c0 * rstride);
      uint4* dst = reinterpret_cast<uint4*>(l_stage);
      const int n16 = ((c1 - c0) * rstride + 15) / 16;
      for (int i = lane; i < n16; i += 64) dst[i] = src[i];
    };
    HP_LONG_DECL;
    // ---- (A) chunk maps: every (chunk, block of 64 entry states) is a task of one wave
    const int passes = (S + 63) / 64;
    if (phase <= 1)
    for (int task = wave + NW * part; task < n_chunks * passes; task += NW * G) {
      const int j = task / passes, q = task % passes;
      const int bot = j * C, top = min(L - 1, bot + C - 1);
      const int s0 = 64 * q + lane;
      int state = s0 < S ? s0 : 0, idx = top;
      uint32_t loc = packed && s0 < S ? hmm_bp_loc(s0, nb, (int)g_block[s0], l_blocks) : 0u;
      int nsteps = 0, nstarts = 0, last_end = -1, last_state = state;
      const int step_cap = (top - bot + 1) * (int)(set.max_mlen + 8) + S;  // (entry states the path cannot be in may hold arbitrary back-pointers)
      for (int c1 = top + 1; c1 > bot;) {
        const int c0 = max(bot, c1 - sub_cols);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stage_cols(c0, c1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        while (__ballot(state != 0 && idx >= c0 && nsteps < step_cap) != 0ull) {
          if (state != 0 && idx >= c0 && nsteps < step_cap) {
            const uint4 p4 = *reinterpret_cast<const uint4*>(l_inst + 4 * state);
            const uint32_t inf = l_info[state];
            const int b = bp_at((idx - c0) * rstride, state, loc);
            const int kind = (int)(inf & 7u);
            ++nsteps; nstarts += kind == 1; if (kind == 2) last_end = idx; last_state = state;
            const uint32_t pe = pred_of(state, b, p4);
            if ((inf >> 3) & 1u) --idx;
            const int nx = (int)(pe & 0x7FFFu);
            state = nx < S ? nx : 0;
            loc = nx < S ? pe >> 16 : 0u;
          }
        }
        c1 = c0;
      }
      if (s0 < S) {
        uint32_t* m = g_map + (size_t)3 * ((size_t)j * S + s0);
        m[0] = (uint32_t)(nsteps >= step_cap ? 0 : state) | ((uint32_t)last_state << 16);
        m[1] = (uint32_t)(nsteps & 0xFFFF) | ((uint32_t)(nstarts & 0xFFFF) << 16);
        m[2] = (uint32_t)last_end;
      }
    }
    __syncthreads();  // (+ the maps are in global memory: read back by thread 0 of this workgroup)
    __threadfence();
    HP_LONG(4);
    if (phase == 1) continue;
    // ---- (B) the chunks strung together from the end state
    uint32_t* const g_tail = reinterpret_cast<uint32_t*>(g_crec + n_chunks);  // [0] steps, [1] visits of the whole walk (phases 2 -> 3)
    if (phase == 2 && (size_t)3 * S * (size_t)n_chunks * 4 <= (size_t)HMM_LONG_MAP_LDS) {  // the maps through LDS: the chain is a dependent look-up per chunk
      for (int i = tid; i < 3 * S * n_chunks; i += HMM_LONG_THREADS) l_map[i] = __builtin_nontemporal_load(g_map + i);
      __syncthreads();
    }
    const bool b_lds = map_lds || (phase == 2 && (size_t)3 * S * (size_t)n_chunks * 4 <= (size_t)HMM_LONG_MAP_LDS);
    if (tid == 0 && phase != 3) {
      uint32_t e = (uint32_t)(S - 1), np = 0, nv = 0; int vb = 0, nxt = -1;
      for (int j = n_chunks - 1; j >= 0; --j) {
        HmmChunkRec r; r.entry = e; r.np0 = np; r.nv0 = nv; r.vb0 = vb; r.nxt0 = nxt; r.pad[0] = r.pad[1] = r.pad[2] = 0;
        g_crec[j] = r;
        const uint32_t* m = (b_lds ? l_map : g_map) + (size_t)3 * ((size_t)j * S + e);
        const uint32_t m0 = b_lds ? m[0] : __builtin_nontemporal_load(m), m1 = b_lds ? m[1] : __builtin_nontemporal_load(m + 1), m2 = b_lds ? m[2] : __builtin_nontemporal_load(m + 2);
        const uint32_t ns = m1 & 0xFFFFu;
        np += ns; nv += m1 >> 16;
        if ((int32_t)m2 >= 0) vb = (int32_t)m2;
        if (ns) nxt = (int)(m0 >> 16);
        e = m0 & 0xFFFFu;
      }
      tot[2] = (int)np; tot[3] = (int)nv;
      if (phase == 2) { g_tail[0] = np; g_tail[1] = nv; hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; }
    }
    __syncthreads();
    __threadfence();
    HP_LONG(5);
    if (phase == 2) continue;
    // ---- (C) every chunk again, from its entry state, with the decoding of the steps (the round loop of hmm_viterbi_kernel)
    uint16_t* pbuf = path ? path + job.path_off : nullptr;
    const int pcap = (int)job.path_cap;
    auto code_at = [&](int i) -> int { return hmm_code(seq, i, L); };
    const unsigned long long below = (1ull << lane) - 1ull;
    int edit_acc = 0, ref_acc = 0;
    for (int j = wave + NW * part; j < n_chunks; j += NW * G) {
      const HmmChunkRec cr = g_crec[j];
      const int bot = j * C, top = min(L - 1, bot + C - 1);
      int state = (int)cr.entry, idx = top, np_c = (int)cr.np0, nv_c = (int)cr.nv0, vb_c = cr.vb0, nxt_c = cr.nxt0;
      uint32_t loc = packed && state < S ? hmm_bp_loc(state, nb, (int)g_block[state], l_blocks) : 0u;
      for (int c1 = top + 1; c1 > bot && state != 0;) {
        const int c0 = max(bot, c1 - sub_cols);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stage_cols(c0, c1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        while (state != 0 && idx >= c0) {
          // the chase of up to 64 steps: on wave-uniform values (state and column in SGPRs: scalar arithmetic, scalar branches)
          int n = 0;
          int emits = __builtin_amdgcn_readfirstlane((int)((l_info[state] >> 3) & 1u));  // (afterwards: bit 15 of the predecessor entry)
          while (state != 0 && idx >= c0 && n < 64) {
            l_rec[2 * n] = (uint32_t)state; l_rec[2 * n + 1] = (uint32_t)idx; ++n;
            const uint4 p4 = *reinterpret_cast<const uint4*>(l_inst + 4 * state);
            const int b = __builtin_amdgcn_readfirstlane(bp_at((idx - c0) * rstride, state, loc));
            const uint32_t pe = (uint32_t)__builtin_amdgcn_readfirstlane((int)pred_of(state, b, p4));
            if (emits) --idx;
            emits = (int)((pe >> 15) & 1u);
            state = (int)(pe & 0x7FFFu);
            loc = pe >> 16;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          {  // ---- what the noted steps mean, one lane per step (events.rs:17-86, purity.rs:6-41, operations.rs:26-57)
            const bool valid = lane < n;
            const int st = valid ? (int)l_rec[2 * lane] : 0, ix = valid ? (int)l_rec[2 * lane + 1] : 0;
            const uint32_t inf = valid ? l_info[st] : 0u;
            const int kind = (int)(inf & 7u), blk = (int)((inf >> 8) & 0xFFu), expected = (int)((inf >> 16) & 0xFFu);
            if (valid && pbuf && np_c + lane < pcap) pbuf[pcap - 1 - (np_c + lane)] = (uint16_t)st;
            const int up = __shfl_up(st, 1);
            const int nxt = lane == 0 ? nxt_c : up;
            const int qbase = valid ? hmm_code_char(code_at(ix)) : 0;
            const int dels = kind == 1 ? nxt - st - 1 : 0;
            const int mism = kind == 4 && !(qbase == expected || expected == 'N');
            int edit = valid ? dels + (kind == 3) + mism + (kind == 5) + (kind == 6) : 0;
            int ref = valid ? dels + (kind == 3) + (kind == 4) + (kind == 6) : 0;
            const unsigned long long ends = __ballot(valid && kind == 2), starts = __ballot(valid && kind == 1);
            const unsigned long long ends_below = ends & below;
            const int src_end = ends_below ? 63 - (int)__builtin_clzll(ends_below) : lane;
            const int idx_end = __shfl(ix, src_end);
            const int vb1 = ends_below ? idx_end : vb_c;
            if (valid && kind == 1) {
              uint32_t drop = 0;
              const int mlen = (int)l_blocks[2 * nb + blk];
              if (blk != nb - 1 && mlen <= 6) {
                if (vb1 - ix < mlen) drop = 1;
                else {
                  const uint8_t* mot = l_mot + l_blocks[3 * nb + blk];
                  for (int jj = 0; jj < mlen; ++jj) {
                    const int obs = hmm_code_char(code_at(ix + jj + 1));
                    if (mot[jj] != 'N' && obs != mot[jj]) drop = 1;
                  }
                }
              }
              const int nv = nv_c + (int)__builtin_popcountll(starts & below);
              uint32_t* vrec = g_vis + 3 * (size_t)nv;
              vrec[0] = (uint32_t)blk | (drop << 15); vrec[1] = (uint32_t)ix; vrec[2] = (uint32_t)vb1;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { edit += __shfl_xor(edit, o); ref += __shfl_xor(ref, o); }
            const int src_last_end = ends ? 63 - (int)__builtin_clzll(ends) : lane;
            const int idx_last_end = __shfl(ix, src_last_end);
            const int last_state = __shfl(st, max(n - 1, 0));
            np_c += n; nv_c += (int)__builtin_popcountll(starts); edit_acc += edit; ref_acc += ref;
            if (n > 0) nxt_c = last_state;
            if (ends) vb_c = idx_last_end;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        c1 = c0;
      }
    }
    if (lane == 0) { atomicAdd(&tot[0], edit_acc); atomicAdd(&tot[1], ref_acc); }
    __syncthreads();
    __threadfence();
    HP_LONG(6);
    if (phase == 3) {  // totals of all workgroups of the allele; the one that arrives last goes on
      if (tid == 0) {
        atomicAdd(hdr + 0, (uint32_t)tot[0]); atomicAdd(hdr + 1, (uint32_t)tot[1]);
        __threadfence();
        l_last = atomicAdd(hdr + 2, 1u) == (uint32_t)(G - 1) ? 1 : 0;
      }
      __syncthreads();
      if (!l_last) continue;
      __threadfence();
      if (tid == 0) {
        tot[0] = (int)__hip_atomic_load(hdr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tot[1] = (int)__hip_atomic_load(hdr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[2] = (int)__builtin_nontemporal_load(g_tail); tot[3] = (int)__builtin_nontemporal_load(g_tail + 1);
      }
      __syncthreads();
    }
    // ---- the end of the walk (the start state closes the path), then as in hmm_viterbi_kernel: path order, purity, the visits
    int np = tot[2];
    if (tid == 0 && pbuf && np < pcap) pbuf[pcap - 1 - np] = 0;
    ++np;
    if (pbuf) {
      const int n = min(np, pcap), shift = pcap - n;
      __syncthreads();
      for (int base = 0; base < n; base += HMM_LONG_THREADS) {
        const int f = base + tid;
        uint16_t v = 0;
        if (f < n) v = pbuf[shift + f];
        __syncthreads();
        if (f < n) pbuf[f] = v;
        __syncthreads();
      }
    }
    if (tid == 0) {
      if (path_len) path_len[job.job_index] = (uint32_t)np;
      const int edit = tot[0], mx = max(tot[1], qlen);
      purity[job.job_index] = ((double)mx - (double)edit) / (double)mx;
      if (edit_out) edit_out[job.job_index] = edit;
      if (maxd_out) maxd_out[job.job_index] = mx;
    }
    // label_motifs over the kept copies, skip filter, counts, collapse (hmm_model.rs:158-200, operations.rs:6-80, utils.rs:3-27), all threads:
    // the visits were recorded back to front, one lane takes one visit; base ranges by a prefix sum of the copy lengths, "the kept
    // visit before me" by a look-back (a span goes on when that one has my motif and ends where I start), span numbers by a prefix
    // count of the span heads, span ends by an atomic max (they grow along a span).  (One thread walking the list paid three
    // dependent global loads per visit: 43 % of this kernel on a 10-kb CAG allele.)
    {
      int32_t* const sp = spans3 + 3 * job.span_off;
      int* const sc_sum = reinterpret_cast<int*>(wave_base);  // [NW] | [NW] | [NW][3]: the staging areas are free now
      int* const sc_heads = sc_sum + NW;
      int* const sc_last = sc_heads + NW;
      const int nvis = tot[3];
      int carry_cum = 0, carry_ns = 0, carry_motif = -1, carry_end = -1;  // (the same in every thread)
      const unsigned long long below_m = (1ull << lane) - 1ull;
      auto scan_add = [&](int x, int* lds_w, int& total) -> int {  // inclusive prefix sum over the workgroup
        int v = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
        if (lane == 63) lds_w[wave] = v;
        __syncthreads();
        int woff = 0, tt = 0;
        for (int w = 0; w < NW; ++w) { const int t = lds_w[w]; if (w < wave) woff += t; tt += t; }
        __syncthreads();
        total = tt;
        return v + woff;
      };
      for (int base = 0; base < nvis; base += HMM_LONG_THREADS) {
        const int i = base + tid;
        const bool in = i < nvis;
        uint32_t v0 = 0, v1 = 0, v2 = 0;
        if (in) {
          const uint32_t* vr = g_vis + 3 * (size_t)(nvis - 1 - i);
          v0 = __builtin_nontemporal_load(vr); v1 = __builtin_nontemporal_load(vr + 1); v2 = __builtin_nontemporal_load(vr + 2);
        }
        const int blk = (int)(v0 & 0x7FFFu), cnt = in ? (int)v2 - (int)v1 : 0;
        const int motif = (v0 >> 15) == 0 ? blk : nb - 1;
        const bool valid = in && motif < n_motifs;
        int total_cnt = 0;
        const int end = carry_cum + scan_add(cnt, sc_sum, total_cnt), start = end - cnt;
        // the kept visit before me: in my wave, else the last one of the nearest wave before that has any, else the chunk before
        const unsigned long long vm = __ballot(valid);
        const int last_lane = vm ? 63 - (int)__builtin_clzll(vm) : 0;
        if (lane == last_lane) { sc_last[3 * wave] = vm ? 1 : 0; sc_last[3 * wave + 1] = motif; sc_last[3 * wave + 2] = end; }
        __syncthreads();
        const unsigned long long vb = vm & below_m;
        const int src = vb ? 63 - (int)__builtin_clzll(vb) : lane;
        int pm = __shfl(motif, src), pe = __shfl(end, src);
        if (!vb) {
          pm = carry_motif; pe = carry_end;
          for (int w = wave - 1; w >= 0; --w) if (sc_last[3 * w]) { pm = sc_last[3 * w + 1]; pe = sc_last[3 * w + 2]; break; }
        }
        int new_motif = carry_motif, new_end = carry_end;
        for (int w = NW - 1; w >= 0; --w) if (sc_last[3 * w]) { new_motif = sc_last[3 * w + 1]; new_end = sc_last[3 * w + 2]; break; }
        const bool head = valid && !(pm == motif && pe == start && pm >= 0);
        int total_heads = 0;
        const int hincl = scan_add(head ? 1 : 0, sc_heads, total_heads);
        const int si = carry_ns + hincl - 1;  // my span (a head: the one it opens)
        if (head) { sp[3 * si + 0] = motif; sp[3 * si + 1] = start; sp[3 * si + 2] = end; }
        if (valid) atomicAdd(&l_cnt[motif], 1u);
        __syncthreads();
        __threadfence();
        if (valid && !head) atomicMax(&sp[3 * si + 2], end);
        carry_cum += total_cnt; carry_ns += total_heads; carry_motif = new_motif; carry_end = new_end;
        __syncthreads();
      }
      if (tid == 0) n_spans[job.job_index] = (uint32_t)carry_ns;
    }
    __syncthreads();
    for (int m = tid; m < n_motifs; m += HMM_LONG_THREADS) counts[job.count_off + m] = l_cnt[m];
    HP_LONG(7);
  }
}

// Job list of a launch class resolved on the device: candidate (locus, allele) slots with everything the host knows ahead of the
// genotyper (motif set, where the allele will be, output places, workspace for its longest possible sequence) become jobs once the
// genotyper has written how many alleles a locus has and how long they are -- no host round trip between the genotyper and this
// kernel's launch.  One workgroup per launch class (results land at the slot's index: the host needs no job order).
struct HmmResolveArgs {
  const HmmJobDev* cand; uint32_t n;
  const uint8_t* skip_locus; const int32_t* n_alleles; const uint32_t* allele_len;  // genotyper results (device), by locus / slot
  HmmJobDev* jobs; uint32_t* n_jobs; uint32_t* n_spans; double* purity;
  uint32_t len_shift;  // alleles are binned by length >> len_shift (64 bins, the last one open)
};
// ... longest alleles first (bins of similar length, descending): workgroups start in list order and a launch ends with its last
// allele -- a long allele started behind thousands of short ones is pure tail.  Three sweeps over the candidates: count per bin,
// exclusive scan over the bins from the longest down, scatter (the order inside a bin is whatever the atomics make it; results do
// not depend on the job order).
constexpr uint32_t HMM_RESOLVE_LDS = 24576;  // candidates whose (active, length) verdict is kept in LDS between the sweeps (96 KB)
__global__ void __launch_bounds__(1024) hmm_resolve_kernel(const HmmResolveArgs a) {
  __shared__ uint32_t hist[64], cursor[64];
  __shared__ uint32_t verdict[HMM_RESOLVE_LDS];  // length + 1 of an active candidate, 0 otherwise
  const int tid = (int)threadIdx.x;
  if (tid < 64) hist[tid] = 0;
  __syncthreads();
  auto probe = [&](uint32_t i) -> uint32_t {  // length + 1 if candidate i is a job, else 0
    const uint32_t slot = a.cand[i].job_index, l = slot >> 1, al = slot & 1u;
    const bool on = !a.skip_locus[l] && (int32_t)al < a.n_alleles[l];
    if (!on) { a.n_spans[slot] = 0; a.purity[slot] = __builtin_nan(""); }  // what the caller's arrays hold for an allele that is not there
    return on ? a.allele_len[slot] + 1u : 0u;
  };
  auto bin_of = [&](uint32_t len) { const uint32_t b = len >> a.len_shift; return 63u - (b < 63u ? b : 63u); };  // bin 0 = the longest
  // One LDS atomic per wave and bin, not per candidate: the alleles of an STR catalog fall into two or three bins, and 20 k atomics on
  // three LDS words were most of this kernel's 0.14 ms (it sits between the genotyper and the HMM on every call's critical path).
  // Returns, for a lane with a bin, the lane's rank among the wave's lanes of the same bin, their number, and whether it leads them.
  auto wave_bins = [&](bool on, uint32_t bin, uint32_t& rank, uint32_t& count, uint32_t& leader) {
    const int lane = tid & 63;
    unsigned long long todo = __ballot(on);
    rank = 0; count = 0; leader = 0;
    while (todo) {
      const int lead = (int)__builtin_ctzll(todo);
      const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)bin, lead);
      const unsigned long long same = __ballot(on && bin == bb);
      if (on && bin == bb) { rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); count = (uint32_t)__popcll(same); leader = (uint32_t)lead; }
      todo &= ~same;
    }
  };
  const uint32_t n_round = (a.n + 1023u) & ~1023u;  // (whole waves walk the loops: the ballots need every lane)
  for (uint32_t i = (uint32_t)tid; i < n_round; i += 1024) {
    const uint32_t v = i < a.n ? probe(i) : 0u;
    if (i < a.n && i < HMM_RESOLVE_LDS) verdict[i] = v;
    uint32_t rank, count, leader;
    const uint32_t bin = v ? bin_of(v - 1u) : 0u;
    wave_bins(v != 0u, bin, rank, count, leader);
    if (v && rank == 0u) atomicAdd(&hist[bin], count);
  }
  __syncthreads();
  if (tid == 0) { uint32_t run = 0; for (int b = 0; b < 64; ++b) { cursor[b] = run; run += hist[b]; } *a.n_jobs = run; }
  __syncthreads();
  for (uint32_t i = (uint32_t)tid; i < n_round; i += 1024) {
    const uint32_t v = i < a.n ? (i < HMM_RESOLVE_LDS ? verdict[i] : probe(i)) : 0u;
    uint32_t rank, count, leader;
    const uint32_t bin = v ? bin_of(v - 1u) : 0u;
    wave_bins(v != 0u, bin, rank, count, leader);
    uint32_t base = 0;
    if (v && rank == 0u) base = atomicAdd(&cursor[bin], count);
    base = (uint32_t)__shfl((int)base, (int)leader);  // (the first lane of a bin's group holds its base; lanes without a job read lane 0)
    if (!v) continue;
    HmmJobDev jd = a.cand[i];
    jd.seq_len = v - 1u;
    a.jobs[base + rank] = jd;
  }
}

// The same in two launches of many workgroups, all classes of a call at once (round 3: the one-workgroup kernel above took 0.14 ms per
// class -- twenty dependent rounds of probes on one CU -- between the genotyper and the HMM of every call).  First launch: verdict per
// candidate, counts per (class, length bin) through one atomic per wave and key; second launch: every workgroup scans the 8 x 64 counts
// itself (longest bin first within a class) and reserves the places of its candidates, again one atomic per wave and key.
struct HmmResolveAllArgs {
  const HmmJobDev* cand; uint32_t n;        // all candidates, classes back to back
  uint32_t class_begin[9];                  // class k = candidates [class_begin[k], class_begin[k + 1])
  const uint8_t* skip_locus; const int32_t* n_alleles; const uint32_t* allele_len;
  HmmJobDev* jobs; uint32_t* n_jobs;        // job list of class k at jobs + class_begin[k], its length at n_jobs[k]
  uint32_t* n_spans; double* purity;
  uint32_t* verdict; uint32_t* hist; uint32_t* taken;  // [n], [512], [512] (hist and taken cleared by the caller)
  uint32_t len_shift;
  const uint8_t* seq_blob; uint8_t* dup;   // dup [n] (optional): 1 = the second allele of a locus equals the first -- no job, its results are copied (hmm_dup_copy_kernel)
};
__device__ __forceinline__ void hmm_wave_keys(bool on, uint32_t key, int lane, uint32_t& rank, uint32_t& count, uint32_t& leader) {
  unsigned long long todo = __ballot(on);
  rank = 0; count = 0; leader = 0;
  while (todo) {
    const int lead = (int)__builtin_ctzll(todo);
    const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, lead);
    const unsigned long long same = __ballot(on && key == kk);
    if (on && key == kk) { rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull)); count = (uint32_t)__popcll(same); leader = (uint32_t)lead; }
    todo &= ~same;
  }
}
__device__ __forceinline__ uint32_t hmm_class_of(const HmmResolveAllArgs& a, uint32_t i) {
  uint32_t k = 0;
#pragma unroll
  for (int c = 1; c < 8; ++c) k += i >= a.class_begin[c] ? 1u : 0u;
  return k;
}
__global__ void __launch_bounds__(256) hmm_resolve_count_kernel(const HmmResolveAllArgs a) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  uint32_t v = 0, key = 0;
  if (i < a.n) {
    const uint32_t slot = a.cand[i].job_index, l = slot >> 1, al = slot & 1u;
    const bool on = !a.skip_locus[l] && (int32_t)al < a.n_alleles[l];
    if (!on) { a.n_spans[slot] = 0; a.purity[slot] = __builtin_nan(""); }  // what the caller's arrays hold for an allele that is not there
    v = on ? a.allele_len[slot] + 1u : 0u;
    // a homozygous locus labels the same sequence twice in the reference (label_with_hmm per allele, tr.rs:454-492): here the second
    // allele gets no job when it equals the first byte for byte (the candidates of a locus sit next to each other)
    if (a.dup) {
      uint8_t d = 0;
      if (on && al == 1u && i > 0 && a.cand[i - 1].job_index == slot - 1u && a.allele_len[slot - 1u] == a.allele_len[slot]) {
        const uint8_t* p = a.seq_blob + a.cand[i - 1].seq_off; const uint8_t* q = a.seq_blob + a.cand[i].seq_off;
        const uint32_t n = a.allele_len[slot];
        // (eight bases per step, all steps' loads independent of the compare: byte by byte this was 200 dependent loads per thread and
        //  0.18 ms between the genotyper and the HMM of every cfg4 call)
        uint64_t diff = 0;
        uint32_t t = 0;
        for (; t + 8 <= n; t += 8) { uint64_t x, y; __builtin_memcpy(&x, p + t, 8); __builtin_memcpy(&y, q + t, 8); diff |= x ^ y; }
        for (; t < n; ++t) diff |= (uint64_t)(p[t] ^ q[t]);
        d = diff == 0 ? 1 : 0;
      }
      a.dup[i] = d;
      if (d) v = 0;
    }
    a.verdict[i] = v;
    if (v) { const uint32_t b = (v - 1u) >> a.len_shift; key = hmm_class_of(a, i) * 64u + (63u - (b < 63u ? b : 63u)); }
  }
  uint32_t rank, count, leader;
  hmm_wave_keys(v != 0u, key, lane, rank, count, leader);
  if (v && rank == 0u) atomicAdd(&a.hist[key], count);
}
__global__ void __launch_bounds__(256) hmm_resolve_scatter_kernel(const HmmResolveAllArgs a) {
  __shared__ uint32_t base[512];
  for (uint32_t t = threadIdx.x; t < 512; t += 256) base[t] = a.hist[t];
  __syncthreads();
  if (threadIdx.x < 8) {  // exclusive scan of a class's bins, bin 0 (the longest alleles) first
    uint32_t run = 0;
    for (int b = 0; b < 64; ++b) { const uint32_t h = base[threadIdx.x * 64 + b]; base[threadIdx.x * 64 + b] = run; run += h; }
    if (blockIdx.x == 0) a.n_jobs[threadIdx.x] = run;
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t v = i < a.n ? a.verdict[i] : 0u;
  uint32_t key = 0, k = 0;
  if (v) { k = hmm_class_of(a, i); const uint32_t b = (v - 1u) >> a.len_shift; key = k * 64u + (63u - (b < 63u ? b : 63u)); }
  uint32_t rank, count, leader;
  hmm_wave_keys(v != 0u, key, lane, rank, count, leader);
  uint32_t at = 0;
  if (v && rank == 0u) at = atomicAdd(&a.taken[key], count);
  at = (uint32_t)__shfl((int)at, (int)leader);
  if (!v) return;
  HmmJobDev jd = a.cand[i];
  jd.seq_len = v - 1u;
  a.jobs[a.class_begin[k] + base[key] + at + rank] = jd;
}

// Behind the HMM kernels: the results of a locus' first allele copied to its second one where hmm_resolve_count_kernel found them equal
__global__ void __launch_bounds__(256) hmm_dup_copy_kernel(const HmmJobDev* __restrict__ cand, uint32_t n, const uint8_t* __restrict__ dup, const HmmSetDev* __restrict__ sets,
                                                           int32_t* __restrict__ spans3, uint32_t* __restrict__ n_spans, uint32_t* __restrict__ counts, double* __restrict__ purity) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !dup[i]) return;
  const HmmJobDev a = cand[i - 1], b = cand[i];
  const uint32_t ns = n_spans[a.job_index];
  n_spans[b.job_index] = ns; purity[b.job_index] = purity[a.job_index];
  for (uint32_t k = 0; k < 3 * ns; ++k) spans3[3 * b.span_off + k] = spans3[3 * a.span_off + k];
  const uint32_t nm = sets[b.set].n_blocks - 1;
  for (uint32_t k = 0; k < nm; ++k) counts[b.count_off + k] = counts[a.count_off + k];
}

// Compaction of the per-job span lists (each job owns a worst-case region) into one dense array for the D2H copy.
__global__ void hmm_pack_spans_kernel(const int32_t* __restrict__ spans3, const uint64_t* __restrict__ job_span_off,
                                      const uint32_t* __restrict__ n_spans, const uint64_t* __restrict__ packed_off,
                                      int32_t* __restrict__ packed, uint64_t n_jobs) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_jobs) return;
  const int32_t* src = spans3 + 3 * job_span_off[j];
  int32_t* dst = packed + 3 * packed_off[j];
  for (uint32_t i = 0; i < 3 * n_spans[j]; ++i) dst[i] = src[i];
}

// exclusive prefix of the span counts (one workgroup): where every job's spans go in the packed array; off[n] = their total
__global__ void __launch_bounds__(1024) hmm_span_prefix_kernel(const uint32_t* __restrict__ n_spans, uint64_t* __restrict__ off, uint64_t n) {
  __shared__ uint64_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint64_t per = (n + 1023) / 1024, b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
  uint64_t sum = 0;
  for (uint64_t i = b; i < e; ++i) sum += n_spans[i];
  part[t] = sum;
  __syncthreads();
  if (t == 0) { uint64_t run = 0; for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; } off[n] = run; }
  __syncthreads();
  uint64_t run = part[t];
  for (uint64_t i = b; i < e; ++i) { off[i] = run; run += n_spans[i]; }
}


// ---- host side of the long trace-back: room for the chunk maps of a job that may be long, and the launch behind a class's fill kernel
// From how many columns on an allele's trace-back goes to the many-wave kernel: 1 536 in a class of thousands of jobs (every long job
// costs that kernel three launches' worth of chunk maps: the hundreds of 0.5-2 kb VNTR alleles of a catalog step were 27 % slower
// through it), 512 in a small class (a pathogenic catalog: the one lane of the fill kernel chasing 1 500 columns was 1.2 ms of a call).
static inline uint32_t hmm_long_min(uint64_t class_jobs) { return class_jobs <= 256 ? 512u : (uint32_t)HMM_LONG_MIN; }  // (2 048 was too many: the 2 000 VNTR candidates of a cfg4 step, 9.2 -> 10.6 ms)
static inline uint64_t hmm_map_words(uint32_t S, uint64_t max_len, uint32_t long_min) {
  const uint64_t L = max_len + 2;
  if (L < (uint64_t)long_min) return 0;
  const uint64_t n_chunks = (L + HMM_LONG_CHUNK - 1) / HMM_LONG_CHUNK;
  return n_chunks * (3ull * S + 8ull) + 8;
}
static size_t hmm_long_lds_bytes(uint32_t S, uint32_t nb) {
  size_t o = 64 + (size_t)20 * S + (size_t)20 * nb + ((S + 3) & ~3u) + (((size_t)S / 3 + 15) & ~(size_t)15) + 32;
  return ((o + 15) & ~(size_t)15) + (size_t)(HMM_LONG_THREADS / 64) * (HMM_LONG_STG + 512) + HMM_LONG_MAP_LDS;
}
static size_t hmm_lds_bytes(uint32_t S, uint32_t nb) {
  size_t o = 64 + (((size_t)HMM_LDS_PER_STATE * S + 15) & ~(size_t)15) + (((size_t)16 * nb + 15) & ~(size_t)15) + (size_t)hmm_stage_bytes((int)((S + 15) & ~15u));
  const size_t spad = (S + 15) & ~15u;
  (void)spad;
  o += HMM_CODE_WINDOW + HMM_CODE_PAD + (((size_t)S / 3 + 15) & ~(size_t)15) + 12 * (size_t)HMM_VIS_LDS + 4 * (size_t)nb;
  o += 8 + 8 * 64;  // the steps of a trace-back round (l_rec)
  return o + 64;
}


// The position-per-lane fill (hmm_ppl.hpp) of one class's job list, in front of the class's trace-back launch on the same stream: one
// launch per group width the class's sets need (lanes_mask: bit 0 = 8 lanes, 1 = 16, 2 = 32, 3 = 64); a launch walks the whole list
// and takes the jobs of its width (groups of other widths idle: a wave without a job of its own returns at once).
static int hmm_launch_ppl(trgt_hip_ctx* c, int bset, int class_slot, hipStream_t ls, unsigned lanes_mask, const HmmJobDev* d_jobs, const HmmSetDev* d_sets, const uint8_t* d_model,
                          const uint8_t* d_seq, uint8_t* d_bp, const ppl::PplSegs& segs) {
  if (!segs.begin[segs.n_seg]) return TRGT_OK;
  auto launch = [&](int g, hipStream_t s) {
    const uint32_t nj = segs.end_slot[g] > segs.first_slot[g] ? segs.end_slot[g] - segs.first_slot[g] : 0u;  // job slots this width looks at
    if (!nj) return;
    // (rows of one byte per lane; TRGT_HMM_PPL_WIDE in `make DEV=1` builds: the round-5 rows of one byte per state)
#define TRGT_PPL_LAUNCH(GG, NB)                                                                                                                         \
    do { if (c->knobs.hmm_ppl_wide) hipLaunchKernelGGL((ppl::hmm_fill_ppl_kernel<GG, false>), dim3(NB), dim3(64), 0, s, d_jobs, d_sets, d_model, d_seq, d_bp, segs); \
         else hipLaunchKernelGGL((ppl::hmm_fill_ppl_kernel<GG, true>), dim3(NB), dim3(64), 0, s, d_jobs, d_sets, d_model, d_seq, d_bp, segs); } while (0)
    if (g == 0) TRGT_PPL_LAUNCH(8, (nj + 7) / 8);
    else if (g == 1) TRGT_PPL_LAUNCH(16, (nj + 3) / 4);
    else if (g == 2) TRGT_PPL_LAUNCH(32, (nj + 1) / 2);
    else TRGT_PPL_LAUNCH(64, nj);
#undef TRGT_PPL_LAUNCH
  };
  // the widest groups on the class's own stream; the other widths (disjoint jobs) next to it on side streams forked off that stream and
  // joined back into it: one behind the other they added up their tails (a cfg4 class with 32- and 64-lane sets: 0.53 + 0.47 ms in
  // front of its trace-back)
  const bool side_ok = !c->knobs.hmm_ppl_serial && class_slot >= 0 && class_slot < 4 && (lanes_mask & (lanes_mask - 1u)) != 0u;
  if (!side_ok) {
    for (int g = 3; g >= 0; --g) if (lanes_mask & (1u << g)) launch(g, ls);
    return TRGT_OK;
  }
  hipEvent_t& f = c->hmm_ppl_fork[bset][class_slot];
  if (!f) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&f, hipEventDisableTiming));
  TRGT_HIP_TRY(c, hipEventRecord(f, ls));
  int n_side = 0;
  bool first = true;
  for (int g = 3; g >= 0; --g) {
    if (!(lanes_mask & (1u << g))) continue;
    if (first || n_side >= 3) { launch(g, ls); first = false; continue; }
    hipStream_t& s = c->hmm_ppl_side[bset][class_slot][n_side];
    hipEvent_t& j = c->hmm_ppl_join[bset][class_slot][n_side];
    if (!s) TRGT_HIP_TRY(c, trgt::make_side_stream(c, &s));
    if (!j) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&j, hipEventDisableTiming));
    TRGT_HIP_TRY(c, hipStreamWaitEvent(s, f, 0));
    launch(g, s);
    TRGT_HIP_TRY(c, hipEventRecord(j, s));
    ++n_side;
  }
  for (int i = 0; i < n_side; ++i) TRGT_HIP_TRY(c, hipStreamWaitEvent(ls, c->hmm_ppl_join[bset][class_slot][i], 0));
  return TRGT_OK;
}
static inline ppl::PplSegs hmm_ppl_one_segment(uint32_t nj, const uint32_t* n_jobs_dev) { ppl::PplSegs g{}; g.begin[1] = nj; g.n_seg = 1; g.counts = n_jobs_dev; for (int w = 0; w < 4; ++w) { g.first_slot[w] = 0; g.end_slot[w] = nj; } return g; }
static inline unsigned hmm_ppl_bit(const HmmSetDev& sd) { const int g = ppl::lanes_for(sd.ppl_lanes); return g == 8 ? 1u : g == 16 ? 2u : g == 32 ? 4u : g == 64 ? 8u : 0u; }

// All motif-set models of a batch (host side).  Thread-safe: touches no ctx state.
int hmm_build_models(int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off, const uint32_t* set_motif_begin, HmmModels& out) {
  char msg[160];
  out.rc = 0;
  std::vector<HmmSetDev>& sets = out.sets;
  sets.assign((size_t)n_sets, HmmSetDev());
  std::vector<std::vector<uint8_t>> blobs((size_t)n_sets);
  std::vector<int> set_err((size_t)n_sets, 0);
  {
    const int nthr = (int)std::min<int64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), std::max(1, n_sets / 64));
    auto work = [&](int t) {
      for (int s = t; s < n_sets; s += nthr) {
        std::vector<std::string> motifs;
        for (uint32_t m = set_motif_begin[s]; m < set_motif_begin[s + 1]; ++m) {
          std::string mot((const char*)motif_blob + motif_off[m], motif_off[m + 1] - motif_off[m]);
          if (mot.empty()) { set_err[s] = 1; break; }
          static const char allowed[] = "ATCGN";  // replace_invalid_bases(m, ATCGN)
          for (size_t i = 0; i < mot.size(); ++i)
            if (mot[i] == 0 || !std::strchr("ATCGN", mot[i])) mot[i] = allowed[i % 5];
          motifs.push_back(mot);
        }
        if (set_err[s]) continue;
        if (motifs.empty()) { set_err[s] = 2; continue; }
        build_set(motifs, blobs[s], sets[s]);
        if (sets[s].S > 1024) set_err[s] = 3;
        else if (sets[s].n_blocks > 254) set_err[s] = 4;
      }
    };
    if (nthr <= 1) work(0);
    else {
      std::vector<std::thread> th;
      for (int t = 0; t < nthr; ++t) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
  }
  std::vector<uint8_t>& blob = out.blob;
  {
    uint64_t total = 0;
    for (int s = 0; s < n_sets; ++s) {
      if (set_err[s] == 1) { snprintf(msg, sizeof msg, "trgt_hmm_batch: empty motif in set %d", s); out.err = msg; return out.rc = TRGT_ERR_INVALID; }
      if (set_err[s] == 2) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has no motif", s); out.err = msg; return out.rc = TRGT_ERR_INVALID; }
      if (set_err[s] == 3) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has %u HMM states (kernel limit 1024)", s, sets[s].S); out.err = msg; return out.rc = TRGT_ERR_UNSUPPORTED; }
      if (set_err[s] == 4) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has too many motifs", s); out.err = msg; return out.rc = TRGT_ERR_UNSUPPORTED; }
      total += blobs[s].size();
    }
    blob.resize((size_t)total);
    uint64_t pos = 0;
    for (int s = 0; s < n_sets; ++s) {
      std::memcpy(blob.data() + pos, blobs[s].data(), blobs[s].size());
      HmmSetDev& d = sets[s];
      d.off_inlp += pos; d.off_em += pos; d.off_inst += pos; d.off_block += pos; d.off_nin += pos; d.off_level += pos;
      d.off_flags += pos; d.off_blocks += pos; d.off_motifs += pos; d.off_perm += pos;
      pos += blobs[s].size();
      std::vector<uint8_t>().swap(blobs[s]);
    }
  }
  return 0;
}


// All motif-set models of a batch, built ON THE DEVICE (hmm_model_build_kernel): the host lays the sets out (sizes, offsets, lane
// tables: a few microseconds per thousand sets), uploads the motifs and the logarithm table (kilobytes) and launches one workgroup per
// set -- instead of building ~2 KB of tables per set on host threads and uploading them (6 ms + 22 MB for a 10k-locus batch: it had
// become the critical path of stage C).  Uploads and the kernel go to `up` (a stream with nothing in front of it); `done` is recorded
// behind them.  out.sets / out.d_sets / out.d_blob are valid on return (the device side once `done` has fired).
int hmm_models_on_device(trgt_hip_ctx* c, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off, const uint32_t* set_motif_begin,
                         HmmModels& out, hipStream_t up, hipEvent_t done) {
  out.rc = 0; out.blob.clear(); out.d_sets = out.d_blob = nullptr;
  std::vector<HmmSetDev>& sets = out.sets;
  sets.assign((size_t)n_sets, HmmSetDev());
  if (n_sets <= 0) return TRGT_OK;
  const uint32_t m_total = set_motif_begin[n_sets];
  const uint32_t byte0 = motif_off[0], bytes_total = motif_off[m_total] - byte0;
  // layout: per set
  std::vector<uint16_t> perm_all, perm;
  std::vector<uint64_t> perm_src((size_t)n_sets, 0);
  std::vector<uint32_t> mlens, blocks, seed_row((size_t)m_total, 0);
  std::vector<double> seed_tab;
  std::vector<int64_t> row_of_len;  // motif length -> first entry of its row (-1: none yet)
  uint64_t pos = 0;
  char msg[160];
  for (int s = 0; s < n_sets; ++s) {
    const uint32_t m0 = set_motif_begin[s], m1 = set_motif_begin[s + 1];
    if (m1 <= m0) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has no motif", s); out.err = msg; return out.rc = TRGT_ERR_INVALID; }
    if (m1 - m0 + 1 > 254) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has too many motifs", s); out.err = msg; return out.rc = TRGT_ERR_UNSUPPORTED; }
    mlens.clear();
    for (uint32_t m = m0; m < m1; ++m) {
      const uint32_t n = motif_off[m + 1] - motif_off[m];
      if (n == 0) { snprintf(msg, sizeof msg, "trgt_hmm_batch: empty motif in set %d", s); out.err = msg; return out.rc = TRGT_ERR_INVALID; }
      mlens.push_back(n);
      if (row_of_len.size() <= n) row_of_len.resize((size_t)n + 1, -1);
      if (row_of_len[n] < 0) {  // ln(seed * (n - k)), k = 1 .. n - 1 (builder.rs:80-173), entry 0 unused
        row_of_len[n] = (int64_t)seed_tab.size();
        const double seed = 2.00 * (1.00 - 0.90) / (double)((size_t)n * (size_t)(n - 1));
        seed_tab.push_back(0.0);
        for (uint32_t k = 1; k < n; ++k) seed_tab.push_back(std::log(seed * (double)(n - k)));
      }
      seed_row[m] = (uint32_t)row_of_len[n];
    }
    HmmSetDev& d = sets[(size_t)s];
    const uint64_t bytes = layout_set(mlens.data(), (uint32_t)mlens.size(), d, blocks, perm);
    if (d.S > 1024) { snprintf(msg, sizeof msg, "trgt_hmm_batch: set %d has %u HMM states (kernel limit 1024)", s, d.S); out.err = msg; return out.rc = TRGT_ERR_UNSUPPORTED; }
    d.off_inlp += pos; d.off_em += pos; d.off_inst += pos; d.off_block += pos; d.off_nin += pos; d.off_level += pos;
    d.off_flags += pos; d.off_blocks += pos; d.off_motifs += pos; d.off_perm += pos;
    pos += bytes;
    if (!perm.empty()) { perm_src[(size_t)s] = perm_all.size(); perm_all.insert(perm_all.end(), perm.begin(), perm.end()); }
  }
  out.blob_bytes = pos;
  // one pinned slab -> one device slab: sets | motif_off (rebased) | set_motif_begin | seed_row | perm_src | seed_tab | perm_all | motif bytes
  struct Lay { size_t total = 0; size_t add(size_t b) { const size_t o = total; total += (b + 15) & ~(size_t)15; return o; } } lay;
  const size_t o_sets = lay.add(sizeof(HmmSetDev) * (size_t)n_sets), o_moff = lay.add(4 * ((size_t)m_total + 1)), o_smb = lay.add(4 * ((size_t)n_sets + 1)),
               o_srow = lay.add(4 * (size_t)m_total), o_psrc = lay.add(8 * (size_t)n_sets), o_stab = lay.add(8 * seed_tab.size()),
               o_perm = lay.add(2 * perm_all.size()), o_mot = lay.add(bytes_total);
  void *h = nullptr, *dv = nullptr, *d_blob = nullptr, *d_sets = nullptr;
  int rc;
  if ((rc = pin_get(c, P_HMM_BUILD, lay.total, &h)) || (rc = dev_get(c, S_HMM_BUILD, lay.total, &dv)) ||
      (rc = dev_get(c, S_HMM_MODEL, (size_t)pos + 16, &d_blob)))
    return out.rc = rc;
  uint8_t* hb = (uint8_t*)h;
  std::memcpy(hb + o_sets, sets.data(), sizeof(HmmSetDev) * (size_t)n_sets);
  { uint32_t* q = (uint32_t*)(hb + o_moff); for (uint32_t m = 0; m <= m_total; ++m) q[m] = motif_off[m] - byte0; }
  std::memcpy(hb + o_smb, set_motif_begin, 4 * ((size_t)n_sets + 1));
  std::memcpy(hb + o_srow, seed_row.data(), 4 * (size_t)m_total);
  std::memcpy(hb + o_psrc, perm_src.data(), 8 * (size_t)n_sets);
  std::memcpy(hb + o_stab, seed_tab.data(), 8 * seed_tab.size());
  if (!perm_all.empty()) std::memcpy(hb + o_perm, perm_all.data(), 2 * perm_all.size());
  {  // replace_invalid_bases(m, ATCGN) (utils.rs:29-42), position inside the motif
    uint8_t* q = hb + o_mot;
    static const char allowed[] = "ATCGN";
    for (uint32_t m = 0; m < m_total; ++m) {
      const uint8_t* src = motif_blob + motif_off[m];
      const uint32_t n = motif_off[m + 1] - motif_off[m];
      uint8_t* dst = q + (motif_off[m] - byte0);
      for (uint32_t i = 0; i < n; ++i) { const uint8_t ch = src[i]; dst[i] = (ch == 'A' || ch == 'T' || ch == 'C' || ch == 'G' || ch == 'N') ? ch : (uint8_t)allowed[i % 5]; }
    }
  }
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  if ((rc = h2d_small(c, dv, h, lay.total, up, -1))) return out.rc = rc;  // (pinned source; a kernel copy: not queued behind bulk uploads)
  // (the descriptors stay where they were uploaded -- the build slab lives as long as the models; the padding between a set's tables is
  //  cleared by the build kernel itself, so that the blob compares equal to the host builder's: a D2D copy and a 20-MB fill per call less)
  d_sets = (uint8_t*)dv + o_sets;
  HmmBuildArgs a;
  const uint8_t* db = (const uint8_t*)dv;
  a.sets = (const HmmSetDev*)(db + o_sets); a.blob = (uint8_t*)d_blob; a.motif_bytes = db + o_mot; a.motif_off = (const uint32_t*)(db + o_moff);
  a.set_motif_begin = (const uint32_t*)(db + o_smb); a.seed_row = (const uint32_t*)(db + o_srow); a.seed_tab = (const double*)(db + o_stab);
  a.perm_all = (const uint16_t*)(db + o_perm); a.perm_src = (const uint64_t*)(db + o_psrc); a.blob_bytes = pos; a.k = hmm_build_consts();
  hipLaunchKernelGGL(hmm_model_build_kernel, dim3((unsigned)n_sets), dim3(64), 0, up, a);
  TRGT_HIP_TRY(c, hipGetLastError());
  if (done) TRGT_HIP_TRY(c, hipEventRecord(done, up));
  out.d_sets = d_sets; out.d_blob = d_blob;
  return TRGT_OK;
}

}  // namespace trgt

using namespace trgt;

extern "C" uint64_t trgt_hmm_path_capacity(uint32_t seq_len, uint32_t max_motif_len) {
  if (seq_len == 0) return 1;
  return (uint64_t)(seq_len + 2) * (max_motif_len + 4) + 8;
}

// State of an enqueued HMM batch between hmm_enqueue (uploads + kernel launches, no host wait for the kernels) and
// hmm_collect (result copies).  One batch per ctx may be pending: the device buffers are the ctx's pool slots.
struct trgt::HmmPending {
  int64_t n_jobs = 0;
  int set = 0; hipStream_t stream = nullptr;  // scratch-buffer set and stream of this batch
  std::vector<uint64_t> cnt_off; std::vector<uint32_t> cnt_n; uint32_t* cnt_user = nullptr; uint64_t cnt_total = 0;  // motif counts go back job by job
  bool spans_on_host = false;
  // the packed copy of the spans is made behind the kernels at enqueue time (pack_behind_kernels): collect only copies
  const int32_t* d_packed = nullptr; const uint64_t* d_poff = nullptr; uint64_t packed_cap_bytes = 0;
  std::vector<uint64_t> tight_off;
  std::vector<uint32_t> slot_nm;  // hmm_enqueue_slots: motifs of every slot's set (the job list is only known once the genotyper is back)
  std::vector<uint64_t> slot_cnt_off;
  std::vector<HmmJobDev> jobs;   // upload sources stay alive until the batch is collected
  HmmModels local_models;
  int32_t* spans3 = nullptr; const uint64_t* span_off = nullptr;
  DevOut<uint16_t> o_path; DevOut<uint32_t> o_plen, o_nsp, o_cnt; DevOut<int32_t> o_spans, o_edit, o_maxd; DevOut<double> o_pur;
};

void trgt::hmm_pending_free(HmmPending* p) { delete p; }

// Behind the kernels of a batch whose spans go to host memory: prefix of the span counts + compaction of the per-job span lists into
// the back-pointer workspace (free once the kernels are through; at least 16 B per (base, state-row) >= 12 B per possible span).
// hmm_collect then copies counts, total and the first HMM_PACKED_FIRST bytes of the packed spans in ONE go -- the host used to wait for
// the counts, build the offsets, upload them, launch the compaction and wait again (0.5 ms of a 10k-locus call's tail).
constexpr size_t HMM_PACKED_FIRST = 1u << 20;
// (the offsets of the per-job span lists go up on `up` BEFORE the kernels are enqueued -- with the job list, on the copy stream in the
//  slots path -- and not behind them on the batch's stream, where that upload sat on the critical path of every call's tail: round 5)
static int pack_tables_upload(trgt_hip_ctx* c, const uint64_t* tight_off_host, int64_t n_jobs, int so, hipStream_t up, void** d_tabs) {
  int rc;
  if ((rc = dev_get(c, S_HMM_MOTIFS + so, (size_t)n_jobs * 16 + 16, d_tabs))) return rc;
  uint64_t* d_toff = (uint64_t*)*d_tabs + (n_jobs + 1);
  return h2d_small(c, d_toff, tight_off_host, (size_t)n_jobs * 8, up, S_HMM_MOTIFS + so);
}
static int pack_behind_kernels(trgt_hip_ctx* c, trgt::HmmPending* P, void* d_tabs, int64_t n_jobs, void* d_bp, uint64_t bp_bytes,
                               const int32_t* d_spans, const uint32_t* d_nsp, int so) {
  uint64_t* d_poff = (uint64_t*)d_tabs; uint64_t* d_toff = d_poff + (n_jobs + 1);
  hipLaunchKernelGGL(hmm_span_prefix_kernel, dim3(1), dim3(1024), 0, c->stream, d_nsp, d_poff, (uint64_t)n_jobs);
  hipLaunchKernelGGL(hmm_pack_spans_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, c->stream, d_spans, (const uint64_t*)d_toff, d_nsp,
                     (const uint64_t*)d_poff, (int32_t*)d_bp, (uint64_t)n_jobs);
  TRGT_HIP_TRY(c, hipGetLastError());
  P->d_packed = (const int32_t*)d_bp; P->d_poff = d_poff; P->packed_cap_bytes = bp_bytes;
  return TRGT_OK;
}

int trgt::hmm_batch_impl(trgt_hip_ctx* c, const HmmModels* premade, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                         const uint32_t* set_motif_begin, int64_t n_jobs, const uint32_t* job_set,
                         const uint8_t* seq_blob, const uint64_t* seq_off, const uint32_t* seq_len, uint16_t* path,
                         const uint64_t* path_off, uint32_t* path_len, int32_t* spans3, const uint64_t* span_off,
                         uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off, double* purity,
                         int32_t* edit_dist, int32_t* max_dist) {
  HmmPending* pend = nullptr;
  int rc = hmm_enqueue(c, premade, n_sets, motif_blob, motif_off, set_motif_begin, n_jobs, job_set, seq_blob, seq_off, seq_len, path, path_off,
                       path_len, spans3, span_off, n_spans, motif_counts, count_off, purity, edit_dist, max_dist, &pend);
  if (rc) return rc;
  return hmm_collect(c, pend);
}

int trgt::hmm_enqueue(trgt_hip_ctx* c, const HmmModels* premade, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                      const uint32_t* set_motif_begin, int64_t n_jobs, const uint32_t* job_set,
                      const uint8_t* seq_blob, const uint64_t* seq_off, const uint32_t* seq_len, uint16_t* path,
                      const uint64_t* path_off, uint32_t* path_len, int32_t* spans3, const uint64_t* span_off,
                      uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off, double* purity,
                      int32_t* edit_dist, int32_t* max_dist, HmmPending** out_pending, int buffer_set) {
  *out_pending = nullptr;
  const int so = buffer_set ? (int)S_HMM_B_BASE - (int)S_HMM_SEQ : 0;  // slot offset of the buffer set
  if (!c) return TRGT_ERR_INVALID;
  if (n_sets < 0 || n_jobs < 0 || (n_jobs > 0 && (!motif_blob || !motif_off || !set_motif_begin || !job_set || !seq_off ||
                                                    !seq_len || (spans3 && !span_off) || !n_spans || !motif_counts ||
                                                    !count_off || !purity)))
    return fail(c, TRGT_ERR_INVALID, "trgt_hmm_batch: null argument");
  if (path && !path_off) return fail(c, TRGT_ERR_INVALID, "trgt_hmm_batch: path without path_off");
  if (n_jobs == 0) return TRGT_OK;
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  std::unique_ptr<HmmPending> P(new HmmPending());
  P->n_jobs = n_jobs; P->spans3 = spans3; P->span_off = span_off; P->set = buffer_set; P->stream = c->stream;
  // ---- models (host libm ln tables): built here unless the caller prepared them ahead of time (trgt_locus_batch does,
  //      concurrently with the flank-location stage)
  HmmModels& local_models = P->local_models;
  const HmmModels* mp = premade;
  if (!mp) {
    const int mrc = hmm_models_on_device(c, n_sets, motif_blob, motif_off, set_motif_begin, local_models, c->stream, nullptr);
    if (mrc) return fail(c, mrc, "%s", local_models.err.empty() ? trgt_hip_last_error(c) : local_models.err.c_str());
    mp = &local_models;
  } else if (mp->rc) return fail(c, mp->rc, "%s", mp->err.c_str());
  const std::vector<HmmSetDev>& sets = mp->sets;
  const std::vector<uint8_t>& blob = mp->blob;
  // ---- jobs, grouped by workgroup size (64 * ceil(S/64))
  std::vector<HmmJobDev>& jobs = P->jobs;
  jobs.resize((size_t)n_jobs);
  uint64_t bp_total = 0, visit_total = 0, seq_total = 0, span_total = 0, count_total = 0, path_total = 0;
  int64_t cells = 0;
  auto set_class_of = [&](const HmmSetDev& sd_) { return sd_.S <= 32 ? 0u : (std::max(sd_.S, sd_.n_lanes) + 63) / 64; };
  uint64_t class_jobs[32] = {};
  for (int64_t j = 0; j < n_jobs; ++j) {
    if ((int32_t)job_set[j] >= n_sets) return fail(c, TRGT_ERR_INVALID, "trgt_hmm_batch: job %lld bad set", (long long)j);
    class_jobs[std::min<uint32_t>(set_class_of(sets[job_set[j]]), 31u)] += 1;
  }
  for (int64_t j = 0; j < n_jobs; ++j) {
    if ((int32_t)job_set[j] >= n_sets) return fail(c, TRGT_ERR_INVALID, "trgt_hmm_batch: job %lld bad set", (long long)j);
    const HmmSetDev& sd = sets[job_set[j]];
    HmmJobDev& jd = jobs[(size_t)j];
    jd.set = job_set[j]; jd.seq_len = seq_len[j]; jd.job_index = (uint32_t)j;
    jd.seq_off = seq_off[j]; jd.span_off = spans3 ? span_off[j] : 0; jd.count_off = count_off[j];
    jd.path_off = path ? path_off[j] : 0;
    jd.path_cap = (uint32_t)std::min<uint64_t>(trgt_hmm_path_capacity(seq_len[j], sd.max_mlen), 0xFFFFFFFFull);
    const uint64_t spad = (sd.S + 15) & ~15u;
    // Invariant (hmm_ppl.hpp): every job's rows start 16 bytes into its piece -- bytes [bp_off - 16, bp_off) take the stores of roles a lane of the
    // position-per-lane fill does not have, and a wave's lanes without a job store to bytes 0..15 of the workspace through the fake
    // bp_off = 16; no row of any job may ever start below 16 (tests/test_hmm_gpu.py compares both fills on long alleles)
    jd.bp_off = bp_total + 16; bp_total += 16 + align_up(spad * ((uint64_t)seq_len[j] + 2), 16);
    jd.visit_off = visit_total; visit_total += 3ull * ((uint64_t)seq_len[j] + 2);
    { const uint64_t mw = c->knobs.hmm_no_long_tb ? 0 : hmm_map_words(sd.S, seq_len[j], hmm_long_min(class_jobs[std::min<uint32_t>(set_class_of(sd), 31u)])); jd.map_off = mw ? visit_total + 4 : 0; visit_total += mw ? mw + 4 : 0; }
    seq_total = std::max<uint64_t>(seq_total, seq_off[j] + seq_len[j]);
    if (spans3) span_total = std::max<uint64_t>(span_total, span_off[j] + seq_len[j] + 1);
    count_total = std::max<uint64_t>(count_total, count_off[j] + (sd.n_blocks - 1));
    P->cnt_off.push_back(count_off[j]); P->cnt_n.push_back(sd.n_blocks - 1);
    if (path) path_total = std::max<uint64_t>(path_total, path_off[j] + jd.path_cap);
    cells += (int64_t)sd.S * ((int64_t)seq_len[j] + 2);
  }
  if (bp_total > c->ws_limit) return fail(c, TRGT_ERR_NOMEM, "trgt_hmm_batch: back-pointer workspace %llu B exceeds limit", (unsigned long long)bp_total);
  // class 0: at most 32 states (two alleles per wave); else the number of waves per allele
  auto job_class = [&](const HmmJobDev& j) { const HmmSetDev& sd_ = sets[j.set]; return sd_.S <= 32 ? 0u : (std::max(sd_.S, sd_.n_lanes) + 63) / 64; };
  {  // (usually one class: skip the sort then)
    bool mixed = false;
    const uint32_t c0 = job_class(jobs[0]);
    for (const auto& jd : jobs) if (job_class(jd) != c0) { mixed = true; break; }
    if (mixed) std::stable_sort(jobs.begin(), jobs.end(), [&](const HmmJobDev& a, const HmmJobDev& b) { return job_class(a) < job_class(b); });
    // Inside a class the longest alleles go first: workgroups start in list order, and a launch ends with its last allele -- a 10-kb
    // allele started behind a thousand short ones is pure tail.  (Classes of many alleles of similar length -- the single STR motif
    // of a whole-genome catalog -- are left alone: sorting twenty thousand jobs costs more than it saves.)
    size_t b0 = 0;
    while (b0 < jobs.size()) {
      size_t b1 = b0; uint32_t mx = 0; uint64_t sum = 0;
      const uint32_t jc = job_class(jobs[b0]);
      while (b1 < jobs.size() && job_class(jobs[b1]) == jc) { mx = std::max(mx, jobs[b1].seq_len); sum += jobs[b1].seq_len; ++b1; }
      const size_t n = b1 - b0;
      if (n > 1 && (n <= 4096 || (uint64_t)mx * n > 4 * sum))
        std::stable_sort(jobs.begin() + (ptrdiff_t)b0, jobs.begin() + (ptrdiff_t)b1, [](const HmmJobDev& a, const HmmJobDev& b) { return a.seq_len > b.seq_len; });
      b0 = b1;
    }
  }
  // ---- device buffers
  const uint8_t* d_seq = nullptr;
  int rc;
  if (is_device_ptr(seq_blob)) d_seq = seq_blob;
  else {
    // host sequences: upload them packed (the caller's blob is usually sparse: one max-length slot per allele)
    uint64_t tight = 0;
    for (int64_t j = 0; j < n_jobs; ++j) tight += seq_len[j];
    void *h_tight = nullptr, *d_tight = nullptr;
    if ((rc = pin_get(c, buffer_set ? P_HMM_SEQ_B : P_HMM_SEQ, (size_t)tight + 16, &h_tight)) || (rc = dev_get(c, S_HMM_SEQ + so, (size_t)tight + 16, &d_tight))) return rc;
    std::vector<uint64_t> toff((size_t)n_jobs);
    uint64_t o = 0;
    for (int64_t j = 0; j < n_jobs; ++j) { toff[(size_t)j] = o; std::memcpy((uint8_t*)h_tight + o, seq_blob + seq_off[j], seq_len[j]); o += seq_len[j]; }
    for (auto& jd : jobs) jd.seq_off = toff[jd.job_index];
    if ((rc = h2d_small(c, d_tight, h_tight, (size_t)tight, c->stream, -1))) return rc;
    d_seq = (const uint8_t*)d_tight;
  }
  void *d_sets = nullptr, *d_model = nullptr, *d_jobs = nullptr, *d_bp = nullptr, *d_visits = nullptr;
  if (mp->d_sets && mp->d_blob) { d_sets = const_cast<void*>(mp->d_sets); d_model = const_cast<void*>(mp->d_blob); }
  else {
    if ((rc = dev_get(c, S_HMM_DESC, sets.size() * sizeof(HmmSetDev), &d_sets))) return rc;
    if ((rc = dev_get(c, S_HMM_MODEL, blob.size(), &d_model))) return rc;
    TRGT_HIP_TRY(c, hipMemcpyAsync(d_sets, sets.data(), sets.size() * sizeof(HmmSetDev), hipMemcpyHostToDevice, c->stream));
    TRGT_HIP_TRY(c, hipMemcpyAsync(d_model, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
  }
  if ((rc = dev_get(c, S_HMM_JOBS + so, jobs.size() * sizeof(HmmJobDev), &d_jobs))) return rc;
  if ((rc = dev_get(c, S_HMM_BP + so, (size_t)bp_total, &d_bp))) return rc;
  if ((rc = dev_get(c, S_HMM_VISITS + so, (size_t)visit_total * 4, &d_visits))) return rc;
  auto &o_path = P->o_path; auto &o_plen = P->o_plen, &o_nsp = P->o_nsp, &o_cnt = P->o_cnt; auto &o_spans = P->o_spans, &o_edit = P->o_edit, &o_maxd = P->o_maxd; auto& o_pur = P->o_pur;
  // Spans: when the caller's buffer is host memory the kernel writes a tight per-job layout on the device and only the
  // spans actually produced are copied back (packed); a device buffer is written in the caller's layout directly.
  // spans3 == NULL: the caller only wants purity / counts (filter_impure_trs); the spans stay in the tight device layout
  const bool discard_spans = spans3 == nullptr;
  const bool spans_on_host = discard_spans || !is_device_ptr(spans3);
  P->spans_on_host = spans_on_host && !discard_spans;
  std::vector<uint64_t>& tight_off = P->tight_off;
  uint64_t tight_total = 0;
  if (spans_on_host) {
    tight_off.resize((size_t)n_jobs);
    for (int64_t j = 0; j < n_jobs; ++j) { tight_off[(size_t)j] = tight_total; tight_total += (uint64_t)seq_len[j] + 1; }
    for (auto& jd : jobs) jd.span_off = tight_off[jd.job_index];
  }
  {  // the job list goes up once, from pinned memory (a pageable source makes the "async" copy a blocking staged one)
    void* h_jobs = nullptr;
    if ((rc = pin_get(c, buffer_set ? P_HMM_JOBS_B : P_HMM_JOBS, jobs.size() * sizeof(HmmJobDev), &h_jobs))) return rc;
    std::memcpy(h_jobs, jobs.data(), jobs.size() * sizeof(HmmJobDev));
    if ((rc = h2d_small(c, d_jobs, h_jobs, jobs.size() * sizeof(HmmJobDev), c->stream, -1))) return rc;
  }
  void* d_pack_tabs = nullptr;
  if (P->spans_on_host && (rc = pack_tables_upload(c, tight_off.data(), n_jobs, so, c->stream, &d_pack_tabs))) return rc;
  if ((rc = o_path.init(c, S_HMM_PATH + so, path, (size_t)path_total))) return rc;
  if ((rc = o_plen.init(c, S_HMM_PLEN + so, path_len, (size_t)n_jobs))) return rc;
  if (spans_on_host) {
    void* d = nullptr;
    if ((rc = dev_get(c, S_HMM_SPANS + so, (size_t)tight_total * 12, &d))) return rc;
    o_spans.user = spans3; o_spans.dev = (int32_t*)d; o_spans.count = 0; o_spans.staged = false;  // copied back packed, below
  } else if ((rc = o_spans.init(c, S_HMM_SPANS + so, spans3, (size_t)span_total * 3))) return rc;
  if ((rc = o_nsp.init(c, S_HMM_NSP + so, n_spans, (size_t)n_jobs))) return rc;
  if ((rc = o_cnt.init(c, S_HMM_CNT + so, motif_counts, (size_t)count_total))) return rc;
  if (o_cnt.staged) {  // host array shared with other batches: only the ranges of this batch's jobs may be written back (hmm_collect)
    P->cnt_user = motif_counts; P->cnt_total = count_total; o_cnt.staged = false;
  }
  if ((rc = o_pur.init(c, S_HMM_PUR + so, purity, (size_t)n_jobs))) return rc;
  if ((rc = o_edit.init(c, S_HMM_EDIT + so, edit_dist, (size_t)n_jobs))) return rc;
  if ((rc = o_maxd.init(c, S_HMM_MAXD + so, max_dist, (size_t)n_jobs))) return rc;
  // ---- one launch per workgroup-size class; the first on the batch's stream, the others on side streams forked off it, so that
  //      the classes run next to each other (a class is bounded by its longest allele: one behind the other they add up their tails)
  // the fork event sits behind the uploads and IN FRONT of the first launch: recorded behind it (as it was until the cfg3 trace showed
  // it), every other class waited for the first class to finish -- 17.6 + 15.2 ms instead of 17.6 for the 10-kb alleles of cfg3
  if (!c->hmm_fork[buffer_set ? 1 : 0]) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->hmm_fork[buffer_set ? 1 : 0], hipEventDisableTiming));
  TRGT_HIP_TRY(c, hipEventRecord(c->hmm_fork[buffer_set ? 1 : 0], c->stream));
  size_t i = 0;
  int n_class = 0;
  unsigned side_used = 0;
  while (i < jobs.size()) {
    const uint32_t jc = job_class(jobs[i]), cls = jc;
    size_t e = i;
    uint32_t maxS = 0, maxnb = 0, maxq = 0;
    unsigned ppl_mask = 0;
    while (e < jobs.size() && job_class(jobs[e]) == jc) {
      maxS = std::max(maxS, sets[jobs[e].set].S); maxnb = std::max(maxnb, sets[jobs[e].set].n_blocks); maxq = std::max(maxq, jobs[e].seq_len);
      ppl_mask |= hmm_ppl_bit(sets[jobs[e].set]); ++e;
    }
    if (c->knobs.hmm_no_ppl) ppl_mask = 0;
    const bool half = cls == 0;  // two alleles per wave
    const size_t lds_job = (hmm_lds_bytes(maxS, maxnb) + 15) & ~(size_t)15;
    const size_t lds = half ? 2 * lds_job : lds_job;
    if (lds > 160 * 1024) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_hmm_batch: LDS need %zu B", lds);
    const bool regs = !c->knobs.hmm_lds_fill;  // one-wave classes keep the score columns in registers (TRGT_HMM_LDS_FILL=1: in LDS like the others)
    const void* kfn = half ? (regs ? (const void*)hmm_viterbi_kernel<32, true> : (const void*)hmm_viterbi_kernel<32, false>)
                           : (regs && cls == 1 ? (const void*)hmm_viterbi_kernel<64, true> : (const void*)hmm_viterbi_kernel<64, false>);
    if (lds > 64 * 1024) TRGT_HIP_TRY(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t ls = c->stream;
    if (n_class > 0) {
      const int sidx = (n_class - 1) % 3 + (buffer_set ? 3 : 0);
      if (!c->hmm_side[sidx]) TRGT_HIP_TRY(c, trgt::make_side_stream(c, &c->hmm_side[sidx]));
      if (!c->hmm_join[sidx]) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->hmm_join[sidx], hipEventDisableTiming));
      ls = c->hmm_side[sidx];
      TRGT_HIP_TRY(c, hipStreamWaitEvent(ls, c->hmm_fork[buffer_set ? 1 : 0], 0));
      side_used |= 1u << sidx;
    }
    ++n_class;
    KTimer t(c, TRGT_K_HMM, ls);
    const uint32_t nj = (uint32_t)(e - i);
    const dim3 grid(half ? (nj + 1) / 2 : nj), block(half ? 64 : 64 * cls);
#define TRGT_HMM_LAUNCH(SB, OW)                                                                                                    \
    hipLaunchKernelGGL((hmm_viterbi_kernel<SB, OW>), grid, block, lds, ls, (const HmmJobDev*)d_jobs + i, (const HmmSetDev*)d_sets,   \
                       (const uint8_t*)d_model, d_seq, (uint8_t*)d_bp, (uint32_t*)d_visits, o_path.dev, o_plen.dev, o_spans.dev, \
                       o_nsp.dev, o_cnt.dev, o_pur.dev, o_edit.dev, o_maxd.dev, nj, (uint32_t)lds_job | (c->knobs.hmm_four_rounds ? 0x80000000u : 0u) | (ppl_mask ? 0x40000000u : 0u) | (ppl_mask && !c->knobs.hmm_ppl_wide ? 0x00800000u : 0u) | ((long_min_cls / 256u) << 24), (const uint32_t*)nullptr, d_long_cls)
    // (the class's list of long alleles: filled by the fill kernel, worked off by the trace-back kernel right behind it)
    uint32_t* d_long_cls = nullptr;
    const uint32_t long_min_cls = hmm_long_min(e - i);
    if (!c->knobs.hmm_no_long_tb && (uint64_t)maxq + 2 >= (uint64_t)long_min_cls) {
      if (void* z = zero_take(c, ((size_t)nj + 16) * 4)) d_long_cls = (uint32_t*)z;  // [count | job indices] of this class, the count cleared
      else {
        void* dl = nullptr;
        if ((rc = dev_get(c, S_HMM_LONG + so, 8 * ((size_t)jobs.size() + 16) * 4, &dl))) return rc;
        d_long_cls = (uint32_t*)dl + (size_t)n_class * 0 + (i + 8 * (size_t)(n_class - 1));
        TRGT_HIP_TRY(c, hipMemsetAsync(d_long_cls, 0, 4, ls));
      }
    }
    if (ppl_mask && (rc = hmm_launch_ppl(c, buffer_set ? 1 : 0, (n_class - 1) & 3, ls, ppl_mask, (const HmmJobDev*)d_jobs + i, (const HmmSetDev*)d_sets, (const uint8_t*)d_model, d_seq, (uint8_t*)d_bp, hmm_ppl_one_segment(nj, nullptr)))) return rc;
    if (half) { if (regs) TRGT_HMM_LAUNCH(32, true); else TRGT_HMM_LAUNCH(32, false); }
    else if (regs && cls == 1) TRGT_HMM_LAUNCH(64, true);
    else TRGT_HMM_LAUNCH(64, false);
#undef TRGT_HMM_LAUNCH
    TRGT_HIP_TRY(c, hipGetLastError());
    if (d_long_cls) {
      const size_t llds = hmm_long_lds_bytes(maxS, maxnb);
      if (llds > 64 * 1024) TRGT_HIP_TRY(c, hipFuncSetAttribute((const void*)hmm_traceback_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
      const int G = c->knobs.hmm_long_wgs;  // workgroups per long allele (1: one launch, the whole trace-back by one workgroup)
      for (int ph = G > 1 ? 1 : 0; ph <= (G > 1 ? 3 : 0); ++ph)
        hipLaunchKernelGGL(hmm_traceback_long_kernel, dim3((unsigned)std::min<uint32_t>(nj, 64u) * (unsigned)(ph == 1 || ph == 3 ? G : 1)), dim3(HMM_LONG_THREADS), llds, ls, (const HmmJobDev*)d_jobs + i, (const HmmSetDev*)d_sets,
                           (const uint8_t*)d_model, d_seq, (const uint8_t*)d_bp, (uint32_t*)d_visits, o_path.dev, o_plen.dev, o_spans.dev, o_nsp.dev, o_cnt.dev, o_pur.dev,
                           o_edit.dev, o_maxd.dev, (const uint32_t*)d_long_cls, ph | (ppl_mask && !c->knobs.hmm_ppl_wide ? 0x100 : 0), ph == 1 || ph == 3 ? G : 1);
      TRGT_HIP_TRY(c, hipGetLastError());
    }
    t.stop(i == 0 ? cells : 0);
    i = e;
  }
  for (int sidx = 0; sidx < 6; ++sidx)
    if (side_used & (1u << sidx)) {
      TRGT_HIP_TRY(c, hipEventRecord(c->hmm_join[sidx], c->hmm_side[sidx]));
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->hmm_join[sidx], 0));
    }
  if (P->spans_on_host && (rc = pack_behind_kernels(c, P.get(), d_pack_tabs, n_jobs, d_bp, bp_total, o_spans.dev, o_nsp.dev, so))) return rc;
  *out_pending = P.release();
  return TRGT_OK;
}


// ---- stage C behind a device-side genotyper: the kernels are enqueued before the host knows which alleles exist (see
//      hmm_resolve_kernel); hmm_slots_resolved tells the pending batch afterwards, hmm_collect is the same as for a host-built list.
int trgt::hmm_enqueue_slots(trgt_hip_ctx* c, const HmmModels* mp, const HmmSlots& in, int32_t* spans3, const uint64_t* span_off,
                            uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off, double* purity,
                            HmmPending** out_pending, int buffer_set) {
  *out_pending = nullptr;
  const int so = buffer_set ? (int)S_HMM_B_BASE - (int)S_HMM_SEQ : 0;
  if (!c) return TRGT_ERR_INVALID;
  if (!mp || mp->rc) return fail(c, mp ? mp->rc : TRGT_ERR_INVALID, "%s", mp ? mp->err.c_str() : "no models");
  const int64_t nl = in.n_loci, n_slots = 2 * nl;
  if (nl <= 0) return TRGT_OK;
  if (!mp->d_sets || !mp->d_blob) return fail(c, TRGT_ERR_INVALID, "hmm_enqueue_slots: the models must be in device memory");
  if (!is_device_ptr(in.seq_blob_dev) || (spans3 && is_device_ptr(spans3))) return fail(c, TRGT_ERR_INVALID, "hmm_enqueue_slots: device alleles, host results");
  TRGT_HIP_TRY(c, hipSetDevice(c->device));
  const std::vector<HmmSetDev>& sets = mp->sets;
  std::unique_ptr<HmmPending> P(new HmmPending());
  P->n_jobs = n_slots; P->spans3 = spans3; P->span_off = span_off; P->set = buffer_set; P->stream = c->stream;
  auto set_class = [&](const HmmSetDev& sd) { return sd.S <= 32 ? 0u : (std::max(sd.S, sd.n_lanes) + 63) / 64; };
  // candidates by class (a counting sort: loci keep their order inside a class)
  uint32_t class_n[8] = {0, 0, 0, 0, 0, 0, 0, 0}, class_begin[9];
  for (int64_t l = 0; l < nl; ++l) {
    if (in.host_skip && in.host_skip[l]) continue;
    const uint32_t k = set_class(sets[(size_t)l]);
    if (k >= 8) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_hmm_batch: a model of %u states", sets[(size_t)l].S);
    class_n[k] += 2;
  }
  class_begin[0] = 0;
  for (int k = 0; k < 8; ++k) class_begin[k + 1] = class_begin[k] + class_n[k];
  const uint32_t n_cand = class_begin[8];
  P->slot_nm.assign((size_t)n_slots, 0); P->slot_cnt_off.assign(count_off, count_off + n_slots);
  P->tight_off.assign(span_off, span_off + n_slots);
  P->spans_on_host = spans3 != nullptr;
  uint64_t count_total = 0, span_total = 0;
  for (int64_t sl = 0; sl < n_slots; ++sl) {
    const HmmSetDev& sd = sets[(size_t)(sl >> 1)];
    P->slot_nm[(size_t)sl] = sd.n_blocks - 1;
    count_total = std::max<uint64_t>(count_total, count_off[sl] + (sd.n_blocks - 1));
    span_total = std::max<uint64_t>(span_total, span_off[sl] + in.cap[sl >> 1] + 1);
  }
  tl_mark(c, "hmm slots: counted");
  if (n_cand == 0) { P->n_jobs = 0; return TRGT_OK; }
  void* h_cand = nullptr;
  int rc;
  if ((rc = pin_get(c, buffer_set ? P_HMM_JOBS_B : P_HMM_JOBS, (size_t)n_cand * sizeof(HmmJobDev), &h_cand))) return rc;
  HmmJobDev* cand = (HmmJobDev*)h_cand;
  uint64_t bp_total = 0, visit_total = 0;
  {
    uint32_t at[8];
    for (int k = 0; k < 8; ++k) at[k] = class_begin[k];
    for (int64_t l = 0; l < nl; ++l) {
      if (in.host_skip && in.host_skip[l]) continue;
      const HmmSetDev& sd = sets[(size_t)l];
      const uint32_t k = set_class(sd);
      const uint64_t spad = (sd.S + 15) & ~15u;
      for (int a = 0; a < 2; ++a) {
        HmmJobDev& jd = cand[at[k]++];
        const int64_t sl = 2 * l + a;
        jd.set = (uint32_t)l; jd.seq_len = 0; jd.job_index = (uint32_t)sl; jd.path_cap = 0;
        jd.seq_off = in.seq_off[sl]; jd.path_off = 0; jd.span_off = span_off[sl]; jd.count_off = count_off[sl];
        jd.bp_off = bp_total + 16; bp_total += 16 + align_up(spad * ((uint64_t)in.cap[l] + 2), 16);  // room for the longest allele the locus can have (+ the dump slot of hmm_fill_ppl_kernel)
        jd.visit_off = visit_total; visit_total += 3ull * ((uint64_t)in.cap[l] + 2);
        { const uint64_t mw = c->knobs.hmm_no_long_tb ? 0 : hmm_map_words(sd.S, in.cap[l], hmm_long_min(class_n[k])); jd.map_off = mw ? visit_total + 4 : 0; visit_total += mw ? mw + 4 : 0; }
      }
    }
  }
  tl_mark(c, "hmm slots: candidates");
  if (bp_total > c->ws_limit) return fail(c, TRGT_ERR_NOMEM, "trgt_hmm_batch: back-pointer workspace %llu B exceeds limit", (unsigned long long)bp_total);
  void *d_jobs = nullptr, *d_bp = nullptr, *d_visits = nullptr;
  const size_t jobs_bytes = (size_t)n_cand * sizeof(HmmJobDev);
  // (S_HMM_JOBS: candidates | job lists | 8 counts | 512 + 512 resolve counters | verdicts)
  if ((rc = dev_get(c, S_HMM_JOBS + so, 2 * jobs_bytes + 64 + 4096 + 4 * (size_t)n_cand + (size_t)n_cand + 16, &d_jobs)) || (rc = dev_get(c, S_HMM_BP + so, (size_t)bp_total, &d_bp)) ||
      (rc = dev_get(c, S_HMM_VISITS + so, (size_t)visit_total * 4, &d_visits)))
    return rc;
  HmmJobDev* const d_cand = (HmmJobDev*)d_jobs;
  void* d_pack_tabs = nullptr;
  HmmJobDev* const d_list = d_cand + n_cand;
  // counts | histogram | places taken (64 + 4096 bytes, cleared): a piece of the call's zero arena when there is one
  void* const z_count = zero_take(c, 64 + 4096);
  uint32_t* const d_count = z_count ? (uint32_t*)z_count : (uint32_t*)((uint8_t*)d_jobs + 2 * jobs_bytes);
  uint32_t* const d_verdict = (uint32_t*)((uint8_t*)d_jobs + 2 * jobs_bytes + 64 + 4096);
  {  // on the copy stream: a copy queued on the batch's stream would sit in the copy engine's queue until the genotyper in front of it
     // has run, and hold up every copy issued after it (the next batch's reads)
    hipStream_t us = c->stream_copy ? c->stream_copy : c->stream;
    if ((rc = h2d_small(c, d_cand, h_cand, jobs_bytes, us, -1))) return rc;
    if (P->spans_on_host && (rc = pack_tables_upload(c, P->tight_off.data(), n_slots, so, us, &d_pack_tabs))) return rc;
    if (us != c->stream) {
      if (!c->ev_upload) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
      TRGT_HIP_TRY(c, hipEventRecord(c->ev_upload, us));
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_upload, 0));
    }
  }
  auto &o_path = P->o_path; auto &o_plen = P->o_plen, &o_nsp = P->o_nsp, &o_cnt = P->o_cnt; auto &o_spans = P->o_spans, &o_edit = P->o_edit, &o_maxd = P->o_maxd; auto& o_pur = P->o_pur;
  if ((rc = o_path.init(c, S_HMM_PATH + so, (uint16_t*)nullptr, 0)) || (rc = o_plen.init(c, S_HMM_PLEN + so, (uint32_t*)nullptr, 0))) return rc;
  {
    void* d = nullptr;
    if ((rc = dev_get(c, S_HMM_SPANS + so, (size_t)span_total * 12 + 16, &d))) return rc;
    o_spans.user = spans3; o_spans.dev = (int32_t*)d; o_spans.count = 0; o_spans.staged = false;  // copied back packed (hmm_collect)
  }
  if ((rc = o_nsp.init(c, S_HMM_NSP + so, n_spans, (size_t)n_slots)) || (rc = o_cnt.init(c, S_HMM_CNT + so, motif_counts, (size_t)count_total)) ||
      (rc = o_pur.init(c, S_HMM_PUR + so, purity, (size_t)n_slots)) || (rc = o_edit.init(c, S_HMM_EDIT + so, (int32_t*)nullptr, 0)) ||
      (rc = o_maxd.init(c, S_HMM_MAXD + so, (int32_t*)nullptr, 0)))
    return rc;
  if (o_cnt.staged) { P->cnt_user = motif_counts; P->cnt_total = count_total; o_cnt.staged = false; }
  uint32_t max_cap = 1, len_shift = 0;
  for (int64_t l = 0; l < nl; ++l) max_cap = std::max(max_cap, in.cap[l]);
  while ((max_cap >> len_shift) >= 64) ++len_shift;
  // slots that are no candidates at all (loci left to the host path) hold "no allele" too
  if (void* z = o_nsp.staged && (size_t)n_slots * 4 <= (512u << 10) ? zero_take(c, (size_t)n_slots * 4) : nullptr) o_nsp.dev = (uint32_t*)z;
  else TRGT_HIP_TRY(c, hipMemsetAsync(o_nsp.dev, 0, (size_t)n_slots * 4, c->stream));
  const uint8_t* d_dup = nullptr;
  if (c->knobs.hmm_resolve_one_wg) {
    for (int k = 0; k < 8; ++k) {
      if (!class_n[k]) continue;
      HmmResolveArgs ra{d_cand + class_begin[k], class_n[k], in.d_skip, in.d_n_alleles, in.d_allele_len, d_list + class_begin[k], d_count + k, o_nsp.dev, o_pur.dev, len_shift};
      hipLaunchKernelGGL(hmm_resolve_kernel, dim3(1), dim3(1024), 0, c->stream, ra);
    }
  } else {
    HmmResolveAllArgs ra;
    ra.cand = d_cand; ra.n = (uint32_t)n_cand;
    for (int k = 0; k < 8; ++k) ra.class_begin[k] = (uint32_t)class_begin[k];
    ra.class_begin[8] = (uint32_t)n_cand;
    ra.skip_locus = in.d_skip; ra.n_alleles = in.d_n_alleles; ra.allele_len = in.d_allele_len;
    ra.jobs = d_list; ra.n_jobs = d_count; ra.n_spans = o_nsp.dev; ra.purity = o_pur.dev;
    ra.hist = d_count + 16; ra.taken = ra.hist + 512; ra.verdict = d_verdict; ra.len_shift = len_shift;
    ra.seq_blob = in.seq_blob_dev; ra.dup = c->knobs.hmm_no_dedupe ? nullptr : reinterpret_cast<uint8_t*>(ra.verdict + n_cand);
    d_dup = ra.dup;
    if (!z_count) TRGT_HIP_TRY(c, hipMemsetAsync(d_count, 0, 64 + 4096, c->stream));
    const dim3 rg((unsigned)((n_cand + 255) / 256));
    hipLaunchKernelGGL(hmm_resolve_count_kernel, rg, dim3(256), 0, c->stream, ra);
    hipLaunchKernelGGL(hmm_resolve_scatter_kernel, rg, dim3(256), 0, c->stream, ra);
  }
  TRGT_HIP_TRY(c, hipGetLastError());
  tl_mark(c, "hmm slots: resolve launched");
  // ---- the position-per-lane fills of ALL classes: one launch per group width over the whole list (a segment per class), the widths next
  //      to each other (hmm_launch_ppl), the classes' trace-backs behind them.  Per class -- up to four launches on as many streams for
  //      each of up to four classes -- the stage had more streams than HIP has hardware queues, and what shares a queue runs one after
  //      the other: in a cfg4 trace the second 64-lane fill began 0.7 ms after the stage did (1.47 ms for the stage; 0.57 + 0.3 are its
  //      longest fill and trace-back).  TRGT_HMM_PPL_PER_CLASS=1: as before.
  unsigned class_ppl_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, all_ppl_mask = 0;
  if (!c->knobs.hmm_no_ppl)
    for (uint32_t k = 0; k < 8; ++k) {
      for (uint32_t i = class_begin[k]; i < class_begin[k + 1]; i += 2) class_ppl_mask[k] |= hmm_ppl_bit(sets[cand[i].set]);
      all_ppl_mask |= class_ppl_mask[k];
    }
  const bool ppl_merged = all_ppl_mask != 0 && !c->knobs.hmm_ppl_per_class;
  if (ppl_merged) {
    ppl::PplSegs segs{};
    for (int k = 0; k <= 8; ++k) segs.begin[k] = k < 8 ? (uint32_t)class_begin[k] : (uint32_t)n_cand;
    segs.n_seg = 8; segs.counts = (const uint32_t*)d_count;
    // a width's launch spans the classes that hold sets of that width, not the whole list: a workgroup that finds no job of its width
    // ends at once, but 17 000 of them in front of 3 000 that work made the 64-lane launch of cfg4 0.73 ms instead of 0.54
    for (int w = 0; w < 4; ++w) {
      segs.first_slot[w] = segs.end_slot[w] = 0;
      bool any = false;
      for (int k = 0; k < 8; ++k)
        if (class_ppl_mask[k] & (1u << w)) { if (!any) segs.first_slot[w] = segs.begin[k]; segs.end_slot[w] = segs.begin[k + 1]; any = true; }
    }
    KTimer tf(c, TRGT_K_HMM, c->stream);
    if ((rc = hmm_launch_ppl(c, buffer_set ? 1 : 0, 0, c->stream, all_ppl_mask, (const HmmJobDev*)d_list, (const HmmSetDev*)mp->d_sets, (const uint8_t*)mp->d_blob, in.seq_blob_dev, (uint8_t*)d_bp, segs))) return rc;
    TRGT_HIP_TRY(c, hipGetLastError());
    tf.stop(0);
  }
  if (!c->hmm_fork[buffer_set ? 1 : 0]) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->hmm_fork[buffer_set ? 1 : 0], hipEventDisableTiming));
  TRGT_HIP_TRY(c, hipEventRecord(c->hmm_fork[buffer_set ? 1 : 0], c->stream));  // (behind the resolve kernel and the merged fills, in front of the first class launch)
  int n_class = 0;
  unsigned side_used = 0;
  for (uint32_t k = 0; k < 8; ++k) {
    if (!class_n[k]) continue;
    uint32_t maxS = 0, maxnb = 0;
    for (uint32_t i = class_begin[k]; i < class_begin[k + 1]; i += 2) { const HmmSetDev& sd = sets[cand[i].set]; maxS = std::max(maxS, sd.S); maxnb = std::max(maxnb, sd.n_blocks); }
    const unsigned ppl_mask = class_ppl_mask[k];
    const bool half = k == 0;
    const size_t lds_job = (hmm_lds_bytes(maxS, maxnb) + 15) & ~(size_t)15;
    const size_t lds = half ? 2 * lds_job : lds_job;
    if (lds > 160 * 1024) return fail(c, TRGT_ERR_UNSUPPORTED, "trgt_hmm_batch: LDS need %zu B", lds);
    const bool regs = !c->knobs.hmm_lds_fill;  // one-wave classes keep the score columns in registers (TRGT_HMM_LDS_FILL=1: in LDS like the others)
    const void* kfn = half ? (regs ? (const void*)hmm_viterbi_kernel<32, true> : (const void*)hmm_viterbi_kernel<32, false>)
                           : (regs && k == 1 ? (const void*)hmm_viterbi_kernel<64, true> : (const void*)hmm_viterbi_kernel<64, false>);
    if (lds > 64 * 1024) TRGT_HIP_TRY(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t ls = c->stream;
    if (n_class > 0) {
      const int sidx = (n_class - 1) % 3 + (buffer_set ? 3 : 0);
      if (!c->hmm_side[sidx]) TRGT_HIP_TRY(c, trgt::make_side_stream(c, &c->hmm_side[sidx]));
      if (!c->hmm_join[sidx]) TRGT_HIP_TRY(c, hipEventCreateWithFlags(&c->hmm_join[sidx], hipEventDisableTiming));
      ls = c->hmm_side[sidx];
      TRGT_HIP_TRY(c, hipStreamWaitEvent(ls, c->hmm_fork[buffer_set ? 1 : 0], 0));
      side_used |= 1u << sidx;
    }
    ++n_class;
    KTimer t(c, TRGT_K_HMM, ls);
    const uint32_t nj = class_n[k];
    const dim3 grid(half ? (nj + 1) / 2 : nj), block(half ? 64 : 64 * k);
#define TRGT_HMM_LAUNCH(SB, OW)                                                                                                    \
    hipLaunchKernelGGL((hmm_viterbi_kernel<SB, OW>), grid, block, lds, ls, (const HmmJobDev*)d_list + class_begin[k], (const HmmSetDev*)mp->d_sets, \
                       (const uint8_t*)mp->d_blob, in.seq_blob_dev, (uint8_t*)d_bp, (uint32_t*)d_visits, o_path.dev, o_plen.dev, o_spans.dev, \
                       o_nsp.dev, o_cnt.dev, o_pur.dev, o_edit.dev, o_maxd.dev, nj, (uint32_t)lds_job | (c->knobs.hmm_four_rounds ? 0x80000000u : 0u) | (ppl_mask ? 0x40000000u : 0u) | (ppl_mask && !c->knobs.hmm_ppl_wide ? 0x00800000u : 0u) | ((long_min_cls / 256u) << 24), (const uint32_t*)(d_count + k), d_long_cls)
    uint32_t* d_long_cls = nullptr;
    uint32_t max_cap_cls = 0;
    for (uint32_t i = class_begin[k]; i < class_begin[k + 1]; i += 2) max_cap_cls = std::max(max_cap_cls, in.cap[cand[i].set]);
    const uint32_t long_min_cls = hmm_long_min(class_n[k]);
    if (!c->knobs.hmm_no_long_tb && (uint64_t)max_cap_cls + 2 >= (uint64_t)long_min_cls) {
      if (void* z = zero_take(c, ((size_t)nj + 16) * 4)) d_long_cls = (uint32_t*)z;  // [count | job indices] of this class, the count cleared
      else {
        void* dl = nullptr;
        if ((rc = dev_get(c, S_HMM_LONG + so, ((size_t)n_cand + 8 * 16) * 4, &dl))) return rc;
        d_long_cls = (uint32_t*)dl + class_begin[k] + 8 * k;
        TRGT_HIP_TRY(c, hipMemsetAsync(d_long_cls, 0, 4, ls));
      }
    }
    if (ppl_mask && !ppl_merged && (rc = hmm_launch_ppl(c, buffer_set ? 1 : 0, (n_class - 1) & 3, ls, ppl_mask, (const HmmJobDev*)d_list + class_begin[k], (const HmmSetDev*)mp->d_sets, (const uint8_t*)mp->d_blob, in.seq_blob_dev, (uint8_t*)d_bp, hmm_ppl_one_segment(nj, (const uint32_t*)(d_count + k))))) return rc;
    if (half) { if (regs) TRGT_HMM_LAUNCH(32, true); else TRGT_HMM_LAUNCH(32, false); }
    else if (regs && k == 1) TRGT_HMM_LAUNCH(64, true);
    else TRGT_HMM_LAUNCH(64, false);
#undef TRGT_HMM_LAUNCH
    TRGT_HIP_TRY(c, hipGetLastError());
    if (d_long_cls) {
      const size_t llds = hmm_long_lds_bytes(maxS, maxnb);
      if (llds > 64 * 1024) TRGT_HIP_TRY(c, hipFuncSetAttribute((const void*)hmm_traceback_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
      const int G = c->knobs.hmm_long_wgs;
      for (int ph = G > 1 ? 1 : 0; ph <= (G > 1 ? 3 : 0); ++ph)
        hipLaunchKernelGGL(hmm_traceback_long_kernel, dim3((unsigned)std::min<uint32_t>(nj, 64u) * (unsigned)(ph == 1 || ph == 3 ? G : 1)), dim3(HMM_LONG_THREADS), llds, ls, (const HmmJobDev*)d_list + class_begin[k], (const HmmSetDev*)mp->d_sets,
                           (const uint8_t*)mp->d_blob, in.seq_blob_dev, (const uint8_t*)d_bp, (uint32_t*)d_visits, o_path.dev, o_plen.dev, o_spans.dev, o_nsp.dev, o_cnt.dev, o_pur.dev,
                           o_edit.dev, o_maxd.dev, (const uint32_t*)d_long_cls, ph | (ppl_mask && !c->knobs.hmm_ppl_wide ? 0x100 : 0), ph == 1 || ph == 3 ? G : 1);
      TRGT_HIP_TRY(c, hipGetLastError());
    }
    t.stop(0);
  }
  for (int sidx = 0; sidx < 6; ++sidx)
    if (side_used & (1u << sidx)) {
      TRGT_HIP_TRY(c, hipEventRecord(c->hmm_join[sidx], c->hmm_side[sidx]));
      TRGT_HIP_TRY(c, hipStreamWaitEvent(c->stream, c->hmm_join[sidx], 0));
    }
  if (d_dup) {  // homozygous loci: the second allele's results are the first one's
    hipLaunchKernelGGL(hmm_dup_copy_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, c->stream, (const HmmJobDev*)d_cand, (uint32_t)n_cand, d_dup, (const HmmSetDev*)mp->d_sets,
                       o_spans.dev, o_nsp.dev, o_cnt.dev, o_pur.dev);
    TRGT_HIP_TRY(c, hipGetLastError());
  }
  tl_mark(c, "hmm slots: classes launched");
  if (P->spans_on_host && (rc = pack_behind_kernels(c, P.get(), d_pack_tabs, n_slots, d_bp, bp_total, o_spans.dev, o_nsp.dev, so))) return rc;
  *out_pending = P.release();
  return TRGT_OK;
}

int64_t trgt::hmm_slots_resolved(trgt_hip_ctx* c, HmmPending* P, const HmmModels* mp, const uint8_t* skip, const int32_t* n_alleles, const uint32_t* allele_len) {
  if (!P) return 0;
  int64_t jobs = 0, cells = 0;
  P->cnt_off.clear(); P->cnt_n.clear();
  for (int64_t sl = 0; sl < P->n_jobs; ++sl) {
    const int64_t l = sl >> 1;
    if (skip[l] || (int32_t)(sl & 1) >= n_alleles[l]) continue;
    P->cnt_off.push_back(P->slot_cnt_off[(size_t)sl]); P->cnt_n.push_back(P->slot_nm[(size_t)sl]);
    cells += (int64_t)mp->sets[(size_t)l].S * ((int64_t)allele_len[sl] + 2);
    ++jobs;
  }
  if (c->timing) c->k_cells[TRGT_K_HMM] += cells;
  return jobs;
}

int trgt::hmm_collect(trgt_hip_ctx* c, HmmPending* pend) {
  if (!pend) return TRGT_OK;  // an empty batch
  std::unique_ptr<HmmPending> P(pend);
  struct StreamSwap { trgt_hip_ctx* c; hipStream_t saved; ~StreamSwap() { c->stream = saved; } } stream_swap{c, c->stream};
  c->stream = P->stream;
  const int64_t n_jobs = P->n_jobs;
  const bool spans_on_host = P->spans_on_host;
  std::vector<uint64_t>& tight_off = P->tight_off;
  int32_t* spans3 = P->spans3; const uint64_t* span_off = P->span_off;
  auto &o_path = P->o_path; auto &o_plen = P->o_plen, &o_nsp = P->o_nsp, &o_cnt = P->o_cnt; auto &o_spans = P->o_spans, &o_edit = P->o_edit, &o_maxd = P->o_maxd; auto& o_pur = P->o_pur;
  int rc;
  const bool tl_on = c->knobs.timeline;
  const auto tl0 = std::chrono::steady_clock::now();
  auto HTL = [&](const char* name, double mb) { if (tl_on) fprintf(stderr, "[tl]   hmm collect %-22s +%6.2f ms  (%.2f MB)\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count(), mb); };
  if (spans_on_host) {
    // counts, total and the head of the packed spans in one round trip (the compaction ran behind the kernels: pack_behind_kernels)
    std::vector<uint32_t> h_nsp((size_t)n_jobs);
    uint64_t ptotal = 0;
    const size_t first = (size_t)std::min<uint64_t>(HMM_PACKED_FIRST, P->packed_cap_bytes);
    std::vector<int32_t> h_packed(first / 4 + 4);
    { const int d2h_rc = trgt::d2h(c, h_nsp.data(), o_nsp.dev, (size_t)n_jobs * 4, c->stream); if (d2h_rc) return d2h_rc; }
    { const int d2h_rc = trgt::d2h(c, &ptotal, P->d_poff + n_jobs, 8, c->stream); if (d2h_rc) return d2h_rc; }
    { const int d2h_rc = trgt::d2h(c, h_packed.data(), P->d_packed, first, c->stream); if (d2h_rc) return d2h_rc; }
    TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
    HTL("counts + packed spans on host", (double)((size_t)n_jobs * 4 + first) / 1e6);
    if (ptotal * 12 > P->packed_cap_bytes) return fail(c, TRGT_ERR_INVALID, "trgt_hmm_batch: %llu spans do not fit the packing workspace", (unsigned long long)ptotal);
    if (ptotal * 12 > first) {  // the rest (many spans: long alleles of interrupted repeats)
      h_packed.resize((size_t)ptotal * 3 + 4);
      { const int d2h_rc = trgt::d2h(c, (uint8_t*)h_packed.data() + first, (const uint8_t*)P->d_packed + first, (size_t)ptotal * 12 - first, c->stream); if (d2h_rc) return d2h_rc; }
      TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
      HTL("rest of the packed spans", (double)(ptotal * 12 - first) / 1e6);
    }
    uint64_t at = 0;
    for (int64_t j = 0; j < n_jobs; ++j) {
      std::memcpy(spans3 + 3 * span_off[j], h_packed.data() + 3 * at, (size_t)h_nsp[(size_t)j] * 12);
      at += h_nsp[(size_t)j];
    }
    (void)tight_off; (void)rc;
  }
  std::vector<uint32_t> h_cnt;
  if (P->cnt_user && P->cnt_total) {
    h_cnt.resize((size_t)P->cnt_total);
    { const int d2h_rc = trgt::d2h(c, h_cnt.data(), o_cnt.dev, (size_t)P->cnt_total * 4, c->stream); if (d2h_rc) return d2h_rc; }
  }
  if ((rc = o_path.finish(c)) || (rc = o_plen.finish(c)) || (rc = o_spans.finish(c)) || (rc = o_nsp.finish(c)) ||
      (rc = o_cnt.finish(c)) || (rc = o_pur.finish(c)) || (rc = o_edit.finish(c)) || (rc = o_maxd.finish(c)))
    return rc;
  HTL("copies enqueued", (double)((o_path.staged ? o_path.count * 2 : 0) + (o_plen.staged ? o_plen.count * 4 : 0) + (o_spans.staged ? o_spans.count * 4 : 0) + (o_nsp.staged ? o_nsp.count * 4 : 0) +
                                  (o_cnt.staged ? o_cnt.count * 4 : 0) + (o_pur.staged ? o_pur.count * 8 : 0) + (o_edit.staged ? o_edit.count * 4 : 0) + (o_maxd.staged ? o_maxd.count * 4 : 0)) / 1e6);
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  HTL("synced", 0.0);
#ifdef TRGT_HMM_PROF
  {
    unsigned long long h[16], z[16] = {0};
    TRGT_HIP_TRY(c, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hmm_prof), sizeof h));
    TRGT_HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(g_hmm_prof), z, sizeof h));
    const double t = (double)(h[0] + h[1] + h[2] + h[3]) + 1e-9;
    fprintf(stderr, "[hmm prof] jobs=%lld | setup %.1f%% fill %.1f%% traceback %.1f%% path+decode %.1f%% | kcycles/job %.1f\n", (long long)n_jobs,
            100 * h[0] / t, 100 * h[1] / t, 100 * h[2] / t, 100 * h[3] / t, t / 1e3 / (double)n_jobs);
    const double f = (double)(h[8] + h[9] + h[10] + h[11] + h[12] + h[13]) + 1e-9;
    fprintf(stderr, "[hmm prof]   long trace-back (kcycles of thread 0 over all alleles): chunk maps %.0f, stringing %.0f, re-walk %.0f, path order + visits %.0f\n", h[4] / 1e3, h[5] / 1e3, h[6] / 1e3, h[7] / 1e3);
    fprintf(stderr, "[hmm prof]   fill (%.0f kcycles on the profiled lanes): symbol %.1f%% emitting+sync %.1f%% chains+ends+sync %.1f%% run end+sync %.1f%% block starts+sync %.1f%% store %.1f%%\n",
            f / 1e3, 100 * h[8] / f, 100 * h[9] / f, 100 * h[10] / f, 100 * h[11] / f, 100 * h[12] / f, 100 * h[13] / f);
  }
#endif
  if (!h_cnt.empty())
    for (size_t j = 0; j < P->cnt_off.size(); ++j)
      std::memcpy(P->cnt_user + P->cnt_off[j], h_cnt.data() + P->cnt_off[j], (size_t)P->cnt_n[j] * 4);
  return TRGT_OK;
}

// Developer / test entry: builds the models of the motif sets both ways -- on the device (what every other entry point uses) and with
// the host builder -- and counts the bytes in which the set descriptors and the table blobs differ (0 expected).
extern "C" int trgt_hmm_models_check(trgt_hip_ctx* c, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                                     const uint32_t* set_motif_begin, int64_t* n_diff) {
  if (!c || !n_diff) return TRGT_ERR_INVALID;
  *n_diff = -1;
  HmmModels dev, host;
  int rc = hmm_models_on_device(c, n_sets, motif_blob, motif_off, set_motif_begin, dev, c->stream, nullptr);
  if (rc) return fail(c, rc, "%s", dev.err.empty() ? trgt_hip_last_error(c) : dev.err.c_str());
  rc = hmm_build_models(n_sets, motif_blob, motif_off, set_motif_begin, host);
  if (rc) return fail(c, rc, "%s", host.err.c_str());
  std::vector<uint8_t> got((size_t)dev.blob_bytes);
  if (dev.blob_bytes) { const int d2h_rc = trgt::d2h(c, got.data(), dev.d_blob, (size_t)dev.blob_bytes, c->stream); if (d2h_rc) return d2h_rc; }
  TRGT_HIP_TRY(c, trgt::stream_wait(c, c->stream));
  int64_t diff = dev.blob_bytes > host.blob.size() ? (int64_t)(dev.blob_bytes - host.blob.size()) : (int64_t)(host.blob.size() - dev.blob_bytes);
  const size_t n = std::min((size_t)dev.blob_bytes, host.blob.size());
  for (size_t i = 0; i < n; ++i) diff += got[i] != host.blob[i];
  const uint8_t *a = (const uint8_t*)dev.sets.data(), *b = (const uint8_t*)host.sets.data();
  for (size_t i = 0; i < sizeof(HmmSetDev) * (size_t)n_sets; ++i) diff += a[i] != b[i];
  *n_diff = diff;
  return TRGT_OK;
}

extern "C" int trgt_hmm_batch(trgt_hip_ctx* c, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                              const uint32_t* set_motif_begin, int64_t n_jobs, const uint32_t* job_set,
                              const uint8_t* seq_blob, const uint64_t* seq_off, const uint32_t* seq_len, uint16_t* path,
                              const uint64_t* path_off, uint32_t* path_len, int32_t* spans3, const uint64_t* span_off,
                              uint32_t* n_spans, uint32_t* motif_counts, const uint64_t* count_off, double* purity,
                              int32_t* edit_dist, int32_t* max_dist) {
  return hmm_batch_impl(c, nullptr, n_sets, motif_blob, motif_off, set_motif_begin, n_jobs, job_set, seq_blob, seq_off, seq_len, path,
                        path_off, path_len, spans3, span_off, n_spans, motif_counts, count_off, purity, edit_dist, max_dist);
}
