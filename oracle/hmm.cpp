// oracle/hmm.cpp -- CPU ORACLE (test infrastructure, never shipped / never on
// the product path).  Scalar restatement of the motif HMM of
// PacificBiosciences/trgt v3.0.0, src/hmm/*.rs and tr.rs:454-492.
// Compile with -ffp-contract=off: Rust never fuses mul-add and the Viterbi
// tie-breaks depend on the exact f64 sums.
#include "oracle_internal.h"

#include <cassert>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace orc {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

// ---- Hmm::new / set_trans / set_ems (hmm_model.rs:28-52) -----------------
Hmm::Hmm(int n) : num_states(n), ems(n, std::array<double, 5>{NEG_INF, NEG_INF, NEG_INF, NEG_INF, NEG_INF}),
                  in_states(n), in_lps(n) {}

void Hmm::set_trans(int target, std::vector<int> ins, std::vector<double> probs) {
  in_states[target] = std::move(ins);
  in_lps[target].clear();
  for (double p : probs) in_lps[target].push_back(std::log(p));  // f64::ln -> libm log
}

void Hmm::set_ems(int target, const std::array<double, 5>& e) {
  for (int i = 0; i < 5; ++i) ems[target][i] = std::log(e[i]);
}

bool Hmm::is_silent(int s) const {
  for (double e : ems[s])
    if (!std::isinf(e)) return false;
  return true;
}
bool Hmm::emits_any(int s) const {  // traceback: any finite emission (hmm_model.rs:134)
  for (double e : ems[s])
    if (std::isfinite(e)) return true;
  return false;
}
bool Hmm::emits_base(int s) const {  // hmm_model.rs:202-204 (skips '#')
  for (int i = 1; i < 5; ++i)
    if (std::isfinite(ems[s][i])) return true;
  return false;
}

// ---- get_match_emissions (builder.rs:175-184) ---------------------------
static std::array<double, 5> match_emissions(uint8_t b) {
  switch (b) {
    case 'A': return {0.00, 0.90, 0.03, 0.03, 0.03};
    case 'T': return {0.00, 0.03, 0.90, 0.03, 0.03};
    case 'C': return {0.00, 0.03, 0.03, 0.90, 0.03};
    case 'G': return {0.00, 0.03, 0.03, 0.03, 0.90};
    case 'N': return {0.00, 0.25, 0.25, 0.25, 0.25};
    default: assert(!"unknown motif base"); return {0, 0, 0, 0, 0};
  }
}

// ---- define_motif_block (builder.rs:80-173) -----------------------------
static void define_motif_block(Hmm& hmm, int ms, const std::string& motif) {
  const int n = (int)motif.size();
  std::vector<int> match_states, ins_states, del_states;
  for (int i = 0; i < n; ++i) match_states.push_back(ms + 1 + i);
  const int first_ins = match_states.back() + 1;
  for (int i = 0; i < n; ++i) ins_states.push_back(first_ins + i);
  const int first_del = ins_states.back() + 1;
  for (int i = 0; i < n - 1; ++i) del_states.push_back(first_del + i);

  const double match_prob = 0.90;
  const double ins_to_ins = 0.25;
  const double match_to_indel = (1.00 - match_prob) / 2.00;
  const double del_to_match = 0.50;
  // builder.rs:93 -- the divisor is an integer product cast to f64
  const double mismatch_seed_prob = 2.00 * (1.00 - match_prob) / (double)((size_t)n * (size_t)(n - 1));

  for (int mi = 0; mi < n; ++mi) {
    const int st = match_states[mi];
    hmm.set_ems(st, match_emissions((uint8_t)motif[mi]));
    if (mi == 0) {
      hmm.set_trans(st, {ms}, {match_prob});
    } else if (mi == 1) {
      const double mismatch_prob = mismatch_seed_prob * (double)(n - mi);
      hmm.set_trans(st, {st - 1, ms, ins_states[mi - 1]}, {match_prob, mismatch_prob, 1.0 - ins_to_ins});
    } else {
      const double mismatch_prob = mismatch_seed_prob * (double)(n - mi);
      hmm.set_trans(st, {st - 1, ms, ins_states[mi - 1], del_states[mi - 2]},
                    {match_prob, mismatch_prob, 1.0 - ins_to_ins, del_to_match});
    }
  }
  for (int ii = 0; ii < n; ++ii) {
    hmm.set_ems(ins_states[ii], {0.00, 0.25, 0.25, 0.25, 0.25});
    hmm.set_trans(ins_states[ii], {ins_states[ii], match_states[ii]}, {ins_to_ins, match_to_indel});
  }
  for (int di = 0; di < n - 1; ++di) {
    hmm.set_ems(del_states[di], {0, 0, 0, 0, 0});
    if (di == 0)
      hmm.set_trans(del_states[di], {match_states[di]}, {match_to_indel});
    else
      hmm.set_trans(del_states[di], {match_states[di], del_states[di - 1]}, {match_to_indel, 1.0 - del_to_match});
  }
  const int me = ms + 3 * n;
  hmm.set_ems(me, {0, 0, 0, 0, 0});
  if (!del_states.empty())
    hmm.set_trans(me, {match_states.back(), ins_states.back(), del_states.back()}, {match_prob, 1.0 - ins_to_ins, 1.0});
  else
    hmm.set_trans(me, {match_states.back(), ins_states.back()}, {match_prob, 1.0 - ins_to_ins});
}

// ---- build_hmm (builder.rs:4-78) ----------------------------------------
Hmm build_hmm(const std::vector<std::string>& motifs) {
  int num_states = 7;
  for (auto& m : motifs) num_states += 3 * (int)m.size() + 1;
  Hmm hmm(num_states);
  const int start = 0, end = num_states - 1, rs = 1, re = end - 1;

  hmm.set_ems(start, {1, 0, 0, 0, 0});
  hmm.set_ems(end, {1, 0, 0, 0, 0});
  hmm.set_trans(end, {re}, {0.10});
  hmm.set_ems(rs, {0, 0, 0, 0, 0});
  hmm.set_trans(rs, {start, re}, {1.00, 1.00});

  const double rs_to_ms = 1.00, me_to_re = 0.50;
  std::vector<int> mes;
  int ms = rs + 1;
  for (auto& motif : motifs) {
    const int nst = 3 * (int)motif.size() + 1;
    const int me = ms + nst - 1;
    hmm.set_ems(ms, {0, 0, 0, 0, 0});
    hmm.set_trans(ms, {rs, me}, {rs_to_ms, 1.0 - me_to_re});
    define_motif_block(hmm, ms, motif);
    mes.push_back(me);
    ms += nst;
  }
  assert(ms + 3 == re);
  const int skip_state = ms + 1, me = ms + 2;
  hmm.set_ems(ms, {0, 0, 0, 0, 0});
  hmm.set_trans(ms, {rs, me}, {rs_to_ms, 1.0 - me_to_re});
  const double skip_to_skip = 0.5;
  hmm.set_ems(skip_state, {0.00, 0.25, 0.25, 0.25, 0.25});
  hmm.set_trans(skip_state, {ms, skip_state}, {1.0, skip_to_skip});
  hmm.set_ems(me, {0, 0, 0, 0, 0});
  hmm.set_trans(me, {skip_state}, {1.0 - skip_to_skip});
  mes.push_back(me);

  hmm.set_ems(re, {0, 0, 0, 0, 0});
  hmm.set_trans(re, mes, std::vector<double>(motifs.size() + 1, me_to_re));

  for (size_t mi = 0; mi < motifs.size(); ++mi) {
    const int mend = mes[mi];
    hmm.motifs.push_back({mend - 3 * (int)motifs[mi].size(), mend, (int)mi});
  }
  hmm.motifs.push_back({skip_state - 1, skip_state + 1, (int)motifs.size()});
  return hmm;
}

// ---- order_states (hmm_model.rs:206-240) --------------------------------
static std::vector<int> order_states(const Hmm& h) {
  std::vector<int> normal, silent;
  for (int s = 0; s < h.num_states; ++s) (h.is_silent(s) ? silent : normal).push_back(s);
  std::vector<int> sorted;
  while (!silent.empty()) {
    std::vector<int> unused;
    for (int s : silent) {
      bool has_in = false;
      for (int p : h.in_states[s])
        for (int q : silent)
          if (q == p) has_in = true;
      (has_in ? unused : sorted).push_back(s);
    }
    assert(unused.size() < silent.size());
    silent.swap(unused);
  }
  normal.insert(normal.end(), sorted.begin(), sorted.end());
  return normal;
}

static inline uint8_t encode_base(uint8_t b) {  // hmm_model.rs:243-252
  switch (b) {
    case '#': return 0;
    case 'A': return 1;
    case 'T': return 2;
    case 'C': return 3;
    case 'G': return 4;
    default: assert(!"encode_base: invalid base"); return 0;
  }
}

// ---- Hmm::label = generate_mats + traceback (hmm_model.rs:54-156) --------
std::vector<int> hmm_label(const Hmm& h, const std::string& seq, int64_t* cells) {
  if (seq.empty()) return {};
  std::vector<uint8_t> q;
  q.push_back(encode_base('#'));
  for (char c : seq) q.push_back(encode_base((uint8_t)c));
  q.push_back(encode_base('#'));
  const int L = (int)q.size(), S = h.num_states;
  const std::vector<int> order = order_states(h);
  std::vector<uint8_t> silent(S);
  for (int s = 0; s < S; ++s) silent[s] = h.is_silent(s);

  std::vector<double> score((size_t)S * L, NEG_INF);  // state-major like the reference
  std::vector<int> bp((size_t)S * L, -1);
  int64_t evals = 0;
  for (int index = 0; index < L; ++index) {
    for (int st : order) {
      const bool sil = silent[st];
      const double em = sil ? 0.0 : h.ems[st][q[index]];
      const int lookback = sil ? 0 : 1;
      const auto& ins = h.in_states[st];
      if (index == 0 && !ins.empty() && lookback == 1) continue;
      double mx = NEG_INF;
      int best = -1;
      for (size_t j = 0; j < ins.size(); ++j) {
        const double prev = score[(size_t)ins[j] * L + (index - lookback)];
        const double v = prev + h.in_lps[st][j] + em;  // (prev + lp) + em, left-assoc (hmm_model.rs:82)
        ++evals;
        if (v > mx) { mx = v; best = ins[j]; }
      }
      if (index == 0 && ins.empty() && std::isfinite(em)) { mx = em; best = st; }
      if (best >= 0) { score[(size_t)st * L + index] = mx; bp[(size_t)st * L + index] = best; }
    }
  }
  if (cells) *cells += (int64_t)S * L;
  (void)evals;
  // traceback (hmm_model.rs:125-142)
  std::vector<int> tb;
  int st = S - 1, index = L - 1;
  while (st != 0) {
    tb.push_back(st);
    const int prev = bp[(size_t)st * L + index];
    assert(prev >= 0);
    if (h.emits_any(st)) index -= 1;
    st = prev;
  }
  tb.push_back(0);
  std::reverse(tb.begin(), tb.end());
  return tb;
}

// ---- label_motifs (hmm_model.rs:158-200) --------------------------------
std::vector<Span> hmm_label_motifs(const Hmm& h, const std::vector<int>& states) {
  std::vector<Span> spans;
  size_t si = 0;
  while (si < states.size()) {
    const int st = states[si];
    int mi = -1;
    for (size_t j = 0; j < h.motifs.size(); ++j)
      if (h.motifs[j].start_state == st) mi = (int)j;
    if (mi >= 0) {
      const HmmMotif& m = h.motifs[mi];
      int span = 0;
      while (states[si] != m.end_state) { span += h.emits_base(states[si]); ++si; }
      while (si < states.size() && states[si] == m.end_state) { span += h.emits_base(states[si]); ++si; }
      const int start = spans.empty() ? 0 : spans.back().end;
      spans.push_back({mi, start, start + span});
    } else {
      assert(!h.emits_base(st));
      ++si;
    }
  }
  return spans;
}

// ---- get_base_match (events.rs:88-117) ----------------------------------
int hmm_base_match(const Hmm& h, int state) {
  const auto& e = h.ems[state];
  if (!h.emits_base(state)) return ' ';
  double mx = e[0];
  for (int i = 1; i < 5; ++i)
    if (e[i] > mx) mx = e[i];
  int cnt = 0, first = -1;
  for (int i = 0; i < 5; ++i)
    if (e[i] == mx) { ++cnt; if (first < 0) first = i; }
  if (cnt == 1) return "#ATCG"[first];
  if (cnt == 4) return 'N';
  return ' ';
}

// ---- get_events (events.rs:17-86) ---------------------------------------
std::vector<uint8_t> hmm_events(const Hmm& h, const std::vector<std::string>& motifs,
                                const std::vector<int>& states, const std::string& query) {
  std::vector<int> s2m(h.num_states, -1);
  for (size_t mi = 0; mi < h.motifs.size(); ++mi)
    for (int s = h.motifs[mi].start_state; s <= h.motifs[mi].end_state; ++s) s2m[s] = (int)mi;
  std::vector<uint8_t> ev;
  size_t base_index = 0;
  for (size_t si = 0; si < states.size(); ++si) {
    const int st = states[si];
    const int mi = s2m[st];
    if (mi == -1) { ev.push_back(EV_TRANS); continue; }
    const HmmMotif& hm = h.motifs[mi];
    if (st == hm.start_state) {
      ev.push_back(EV_MOTIF_START);
      const int next = states[si + 1];
      for (int d = 0; d < next - st - 1; ++d) ev.push_back(EV_DEL);
      continue;
    }
    if (st == hm.end_state) { ev.push_back(EV_MOTIF_END); continue; }
    if ((size_t)mi + 1 == h.motifs.size()) { ev.push_back(EV_SKIP); ++base_index; continue; }
    const int offset = st - hm.start_state - 1;
    const int mlen = (int)motifs[hm.motif_index].size();
    uint8_t e;
    switch (offset / mlen) {
      case 0: {
        const int base = (uint8_t)query[base_index];
        const int expected = hmm_base_match(h, st);
        e = (base == expected || expected == 'N') ? EV_MATCH : EV_MISMATCH;
        break;
      }
      case 1: e = EV_INS; break;
      case 2: e = EV_DEL; break;
      default: assert(!"Event decoding error"); e = EV_TRANS;
    }
    if (e == EV_MATCH || e == EV_MISMATCH || e == EV_INS || e == EV_SKIP) ++base_index;
    ev.push_back(e);
  }
  return ev;
}

// ---- calc_purity (purity.rs:6-41) ---------------------------------------
double hmm_purity(const Hmm& h, const std::vector<std::string>& motifs, const std::vector<int>& states,
                  const std::string& query, int* edit_out, int* max_out) {
  if (edit_out) *edit_out = 0;
  if (max_out) *max_out = 0;
  if (query.empty()) return std::numeric_limits<double>::quiet_NaN();
  const auto ev = hmm_events(h, motifs, states, query);
  int edit = 0, ref_len = 0;
  for (uint8_t e : ev) {
    if (e == EV_DEL || e == EV_INS || e == EV_MISMATCH || e == EV_SKIP) ++edit;
    if (e == EV_MATCH || e == EV_MISMATCH || e == EV_DEL || e == EV_SKIP) ++ref_len;
  }
  const int mx = std::max(ref_len, (int)query.size());
  if (edit_out) *edit_out = edit;
  if (max_out) *max_out = mx;
  const double max_dist = (double)mx;
  return (max_dist - (double)edit) / max_dist;
}

// ---- remove_imperfect_motifs (operations.rs:6-80) -----------------------
std::vector<int> hmm_remove_imperfect(const Hmm& h, const std::vector<std::string>& motifs,
                                      const std::vector<int>& states, const std::string& query, int max_motif_len) {
  if (states.empty()) return {};
  assert(states.size() > 4);
  std::vector<int> out{states[0], states[1]};
  auto is_start = [&](int s) { for (auto& m : h.motifs) if (m.start_state == s) return true; return false; };
  auto is_end = [&](int s) { for (auto& m : h.motifs) if (m.end_state == s) return true; return false; };
  const int run_end = h.num_states - 2;
  size_t si = 2, base_index = 0;
  while (si != states.size()) {
    assert(is_start(states[si]));
    (void)is_start;
    std::vector<int> ms;
    std::string mseq;
    while (!is_end(states[si])) {
      ms.push_back(states[si]);
      if (h.emits_base(states[si])) { mseq.push_back(query[base_index]); ++base_index; }
      ++si;
    }
    ms.push_back(states[si]);
    ++si;
    const HmmMotif* rec = nullptr;
    for (auto& m : h.motifs) if (m.start_state == ms.front()) rec = &m;
    const int mlen = (rec->end_state - rec->start_state) / 3;
    bool keep = true;
    const bool skip_motif = (size_t)rec->motif_index + 1 == h.motifs.size();
    if (!skip_motif && mlen <= max_motif_len) {
      const std::string& motif = motifs[rec->motif_index];
      if (mseq.size() < motif.size()) keep = false;
      else
        for (size_t i = 0; i < motif.size(); ++i)
          if (motif[i] != 'N' && mseq[i] != motif[i]) keep = false;
    }
    if (keep) out.insert(out.end(), ms.begin(), ms.end());
    else {
      int consumed = 0;
      for (int s : ms) consumed += h.emits_base(s);
      const HmmMotif& sk = h.motifs.back();
      out.push_back(sk.start_state);
      for (int i = 0; i < consumed; ++i) out.push_back(sk.start_state + 1);
      out.push_back(sk.end_state);
    }
    if (states[si] == run_end) { out.push_back(states[si]); out.push_back(states[si + 1]); si += 2; }
  }
  return out;
}

// ---- replace_invalid_bases (utils.rs:29-42) -----------------------------
std::string replace_invalid_bases(const std::string& seq, const std::string& allowed) {
  std::string out = seq;
  for (size_t i = 0; i < out.size(); ++i)
    if (allowed.find(out[i]) == std::string::npos) out[i] = allowed[i % allowed.size()];
  return out;
}

// ---- label_with_hmm for one allele (tr.rs:463-488) ----------------------
Annotation annotate_allele(const Hmm& h, const std::vector<std::string>& motifs, const std::string& raw_seq,
                           std::vector<int>* path_out, int64_t* cells) {
  Annotation a;
  const std::string seq = replace_invalid_bases(raw_seq, "ATCG");
  std::vector<int> labels = hmm_label(h, seq, cells);
  if (path_out) *path_out = labels;
  a.purity = hmm_purity(h, motifs, labels, seq, &a.edit_dist, &a.max_dist);
  labels = hmm_remove_imperfect(h, motifs, labels, seq, 6);
  std::vector<Span> spans = hmm_label_motifs(h, labels);
  std::vector<Span> kept;
  for (auto& s : spans)
    if (s.motif_index < (int)motifs.size()) kept.push_back(s);
  a.motif_counts.assign(motifs.size(), 0);  // count_motifs (utils.rs:3-9)
  for (auto& s : kept) a.motif_counts[s.motif_index] += 1;
  for (auto& s : kept) {  // collapse_labels (utils.rs:11-27)
    if (!a.labels.empty() && a.labels.back().motif_index == s.motif_index && a.labels.back().end == s.start)
      a.labels.back().end = s.end;
    else
      a.labels.push_back(s);
  }
  return a;
}

std::vector<std::string> motifs_from_blob(const uint8_t* blob, const uint32_t* off, int n) {
  std::vector<std::string> m;
  for (int i = 0; i < n; ++i) m.emplace_back((const char*)blob + off[i], off[i + 1] - off[i]);
  return m;
}

}  // namespace orc

// =========================== C API =======================================
using namespace orc;

static std::vector<int> to_vec(const int32_t* p, int n) { return std::vector<int>(p, p + n); }

extern "C" {

int orc_hmm_num_states(const uint32_t* motif_off, int n_motifs) {
  int s = 7;
  for (int i = 0; i < n_motifs; ++i) s += 3 * (int)(motif_off[i + 1] - motif_off[i]) + 1;
  return s;
}

int orc_hmm_label(const uint8_t* mb, const uint32_t* mo, int nm, const uint8_t* seq, int n, int32_t* path, int cap) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  auto p = hmm_label(h, std::string((const char*)seq, n), nullptr);
  if ((int)p.size() > cap) return -1;
  for (size_t i = 0; i < p.size(); ++i) path[i] = p[i];
  return (int)p.size();
}

int orc_hmm_remove_imperfect(const uint8_t* mb, const uint32_t* mo, int nm, const int32_t* path, int pl,
                             const uint8_t* seq, int n, int max_motif_len, int32_t* out, int cap) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  auto r = hmm_remove_imperfect(h, motifs, to_vec(path, pl), std::string((const char*)seq, n), max_motif_len);
  if ((int)r.size() > cap) return -1;
  for (size_t i = 0; i < r.size(); ++i) out[i] = r[i];
  return (int)r.size();
}

int orc_hmm_label_motifs(const uint8_t* mb, const uint32_t* mo, int nm, const int32_t* path, int pl, int32_t* spans3, int cap) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  auto sp = hmm_label_motifs(h, to_vec(path, pl));
  if ((int)sp.size() > cap) return -1;
  for (size_t i = 0; i < sp.size(); ++i) {
    spans3[3 * i] = sp[i].motif_index; spans3[3 * i + 1] = sp[i].start; spans3[3 * i + 2] = sp[i].end;
  }
  return (int)sp.size();
}

int orc_hmm_events(const uint8_t* mb, const uint32_t* mo, int nm, const int32_t* path, int pl,
                   const uint8_t* seq, int n, uint8_t* events, int cap) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  auto ev = hmm_events(h, motifs, to_vec(path, pl), std::string((const char*)seq, n));
  if ((int)ev.size() > cap) return -1;
  std::memcpy(events, ev.data(), ev.size());
  return (int)ev.size();
}

double orc_hmm_purity(const uint8_t* mb, const uint32_t* mo, int nm, const int32_t* path, int pl,
                      const uint8_t* seq, int n, int32_t* edit, int32_t* maxd) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  int e = 0, m = 0;
  double p = hmm_purity(h, motifs, to_vec(path, pl), std::string((const char*)seq, n), &e, &m);
  if (edit) *edit = e;
  if (maxd) *maxd = m;
  return p;
}

int orc_hmm_base_match(const uint8_t* mb, const uint32_t* mo, int nm, int state) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  Hmm h = build_hmm(motifs);
  return hmm_base_match(h, state);
}

// get_base_match on a hand-made model: Hmm::new(n_states) + set_ems(state, probs) for every state (events.rs:138-145 builds one)
int orc_hmm_base_match_ems(int n_states, const double* ems_probs /* 5 per state */, int state) {
  Hmm h(n_states);
  for (int s = 0; s < n_states; ++s) h.set_ems(s, {ems_probs[5 * s], ems_probs[5 * s + 1], ems_probs[5 * s + 2], ems_probs[5 * s + 3], ems_probs[5 * s + 4]});
  return hmm_base_match(h, state);
}

void orc_replace_invalid_bases(uint8_t* seq, int len, const char* allowed) {
  std::string r = replace_invalid_bases(std::string((const char*)seq, len), allowed);
  std::memcpy(seq, r.data(), len);
}

int orc_hmm_annotate(const uint8_t* mb, const uint32_t* mo, int nm, const uint8_t* seq, int n,
                     int32_t* path, int path_cap, int32_t* path_len, int32_t* spans3, int span_cap, int32_t* n_spans,
                     int32_t* motif_counts, double* purity, int32_t* edit, int32_t* maxd, int64_t* cells) {
  auto motifs = motifs_from_blob(mb, mo, nm);
  for (auto& m : motifs) m = replace_invalid_bases(m, "ATCGN");  // tr.rs:455-460
  Hmm h = build_hmm(motifs);
  std::vector<int> p;
  int64_t c = 0;
  Annotation a = annotate_allele(h, motifs, std::string((const char*)seq, n), &p, &c);
  if ((int)p.size() > path_cap || (int)a.labels.size() > span_cap) return -1;
  for (size_t i = 0; i < p.size(); ++i) path[i] = p[i];
  *path_len = (int)p.size();
  for (size_t i = 0; i < a.labels.size(); ++i) {
    spans3[3 * i] = a.labels[i].motif_index; spans3[3 * i + 1] = a.labels[i].start; spans3[3 * i + 2] = a.labels[i].end;
  }
  *n_spans = (int)a.labels.size();
  for (int i = 0; i < nm; ++i) motif_counts[i] = a.motif_counts[i];
  *purity = a.purity;
  if (edit) *edit = a.edit_dist;
  if (maxd) *maxd = a.max_dist;
  if (cells) *cells = c;
  return 0;
}

int orc_hmm_batch(int n_sets, const uint8_t* motif_blob, const uint32_t* motif_off, const uint32_t* set_motif_begin,
                  int64_t n_jobs, const uint32_t* job_set, const uint8_t* seq_blob, const uint64_t* seq_off,
                  const uint32_t* seq_len, uint16_t* path, const uint64_t* path_off, uint32_t* path_len,
                  int32_t* spans3, const uint64_t* span_off, uint32_t* n_spans, uint32_t* motif_counts,
                  const uint64_t* count_off, double* purity, int32_t* edit_dist, int32_t* max_dist,
                  int64_t* cells_total, int n_threads) {
  // build every model once (label_with_hmm builds once per locus, tr.rs:461)
  std::vector<std::vector<std::string>> set_motifs(n_sets);
  std::vector<Hmm> hmms;
  hmms.reserve(n_sets);
  for (int s = 0; s < n_sets; ++s) {
    for (uint32_t m = set_motif_begin[s]; m < set_motif_begin[s + 1]; ++m)
      set_motifs[s].push_back(replace_invalid_bases(
          std::string((const char*)motif_blob + motif_off[m], motif_off[m + 1] - motif_off[m]), "ATCGN"));
    hmms.push_back(build_hmm(set_motifs[s]));
  }
  if (n_threads < 1) n_threads = 1;
  std::vector<int64_t> cells(n_threads, 0);
  auto work = [&](int t) {
    for (int64_t j = t; j < n_jobs; j += n_threads) {
      const int s = (int)job_set[j];
      std::vector<int> p;
      Annotation a = annotate_allele(hmms[s], set_motifs[s], std::string((const char*)seq_blob + seq_off[j], seq_len[j]), &p, &cells[t]);
      if (path) for (size_t i = 0; i < p.size(); ++i) path[path_off[j] + i] = (uint16_t)p[i];
      if (path_len) path_len[j] = (uint32_t)p.size();
      for (size_t i = 0; i < a.labels.size(); ++i) {
        int32_t* o = spans3 + 3 * (span_off[j] + i);
        o[0] = a.labels[i].motif_index; o[1] = a.labels[i].start; o[2] = a.labels[i].end;
      }
      n_spans[j] = (uint32_t)a.labels.size();
      for (size_t i = 0; i < a.motif_counts.size(); ++i) motif_counts[count_off[j] + i] = (uint32_t)a.motif_counts[i];
      purity[j] = a.purity;
      if (edit_dist) edit_dist[j] = a.edit_dist;
      if (max_dist) max_dist[j] = a.max_dist;
    }
  };
  if (n_threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  if (cells_total) { *cells_total = 0; for (auto c : cells) *cells_total += c; }
  return 0;
}

}  // extern "C"
