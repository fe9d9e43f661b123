// oracle/wfa.cpp -- CPU ORACLE (test infrastructure, never shipped / never on
// the product path).  Scalar restatement of the wavefront alignment algorithm
// (WFA / BiWFA) that PacificBiosciences/trgt v3.0.0 reaches through
// src/wfaligner.rs -> wfa2-sys -> WFA2-lib (C).  WFA2-lib is an UN-VENDORED
// third-party dependency (wfa2-sys 0.1.0, git ctsa/rust-wfa2 rev 4342b3b0,
// Cargo.toml:36, Cargo.lock:1839-1845) whose sources are not under
// /root/reference; this file restates its published algorithm (Marco-Sola et
// al. 2021 "Fast gap-affine pairwise alignment using the wavefront
// algorithm"; 2023 "Optimal gap-affine alignment in O(s) space") following
// SURVEY.md Appendix A and is anchored on the reference's own call sites and
// known-answer tests (wfaligner.rs:1136-1828 -> tests/golden/wfa_kats.json).
//
// Pinned: every exact unidirectional mode (all KATs).  UNPINNED: BiWFA CIGAR
// tie-breaking and the wfadaptive heuristic (only wfaligner.rs:1437-1454 and
// the commented-out test :1754-1792 give evidence; both are reproduced).
#include "oracle_internal.h"

#include <cassert>
#include <climits>
#include <cstring>
#include <thread>

namespace orc {

static const int32_t OFF_NULL = INT32_MIN / 2;  // WAVEFRONT_OFFSET_NULL
enum { CM = 0, CI1 = 1, CI2 = 2, CD1 = 3, CD2 = 4 };  // affine2p_matrix_type
enum { ST_OK = 0, ST_END_REACHED = 1, ST_END_UNREACHABLE = 2 };
enum { WF_COMPLETED = 0, WF_PARTIAL = 1, WF_MAX_STEPS = -100, WF_OOM = -200, WF_UNATTAINABLE = -300 };
enum { M_INDEL = 0, M_EDIT = 1, M_LINEAR = 2, M_AFFINE = 3, M_AFFINE2P = 4 };

struct SeqView {
  const uint8_t* p; int len; bool rev;
  inline uint8_t at(int i) const { return rev ? p[len - 1 - i] : p[i]; }
};

struct Wf {
  bool exists = false;  // pointer != NULL in WFA2-lib
  bool null = true;     // ->null
  int lo = 1, hi = -1, base = 0;
  std::vector<int32_t> off;
  void alloc(int l, int h) {
    exists = true; null = false; lo = l; hi = h; base = l;
    off.assign((size_t)std::max(0, h - l + 1), OFF_NULL);
  }
  inline int32_t get(int k) const { return (k >= lo && k <= hi) ? off[k - base] : OFF_NULL; }
  inline int32_t& at(int k) { return off[k - base]; }
};
static const Wf WF_NULL;  // lo = 1, hi = -1, all offsets NULL

struct Breakpoint {
  int score = INT_MAX, score_forward = 0, score_reverse = 0, k_forward = 0, k_reverse = 0;
  int32_t offset_forward = 0, offset_reverse = 0; int component = CM;
};

// One unidirectional aligner (wavefront_aligner_t with memory high / score-only stepwise use).
class Uni {
 public:
  const orc_wfa_params& P_;
  SeqView pat, txt;
  int plen, tlen, metric, x, o1, e1, o2, e2;
  int span, pbf, pef, tbf, tef, comp_begin, comp_end;
  std::vector<Wf> wf[5];
  int num_null_steps = 0, steps_wait = 0, max_score_scope = 2, status = ST_OK;
  int end_score = -1, end_k = 0; int32_t end_offset = 0;
  int64_t cells = 0;
  std::string ops;  // backtrace result (forward order)

  Uni(const orc_wfa_params& p, SeqView pa, SeqView te, int span_, int pbf_, int pef_, int tbf_, int tef_, int cb, int ce)
      : P_(p), pat(pa), txt(te), plen(pa.len), tlen(te.len), metric(p.metric), span(span_), pbf(pbf_), pef(pef_),
        tbf(tbf_), tef(tef_), comp_begin(cb), comp_end(ce) {
    switch (metric) {
      case M_INDEL: x = -1; o1 = 1; e1 = -1; o2 = e2 = -1; max_score_scope = 2; break;
      case M_EDIT: x = 1; o1 = 1; e1 = -1; o2 = e2 = -1; max_score_scope = 2; break;
      case M_LINEAR: x = p.mismatch; o1 = p.gap_ext1; e1 = -1; o2 = e2 = -1; max_score_scope = std::max(x, o1) + 1; break;
      case M_AFFINE: x = p.mismatch; o1 = p.gap_open1; e1 = p.gap_ext1; o2 = e2 = -1;
        max_score_scope = std::max(x, o1 + e1) + 1; break;
      default: x = p.mismatch; o1 = p.gap_open1; e1 = p.gap_ext1; o2 = p.gap_open2; e2 = p.gap_ext2;
        max_score_scope = std::max(x, std::max(o1 + e1, o2 + e2)) + 1; break;
    }
    init();
  }

  Wf& slot(int c, int s) {
    if ((int)wf[c].size() <= s) wf[c].resize(s + 1);
    return wf[c][s];
  }
  const Wf& fetch(int c, int s) const {  // wavefront_compute_get_*wavefront
    if (s < 0 || s >= (int)wf[c].size() || !wf[c][s].exists || wf[c][s].null) return WF_NULL;
    return wf[c][s];
  }
  const Wf* ptr(int c, int s) const {  // raw pointer semantics (may be null-flagged)
    if (s < 0 || s >= (int)wf[c].size() || !wf[c][s].exists) return nullptr;
    return &wf[c][s];
  }

  // wavefront_unialign_init (+ init_end2end / init_endsfree, heuristic_clear)
  void init() {
    num_null_steps = 0; status = ST_OK; end_score = -1;
    steps_wait = P_.h_steps_between_cutoffs;
    if (span == 0) {
      Wf& w = slot(comp_begin, 0);
      w.alloc(0, 0); w.at(0) = 0; cells += 1;
    } else {
      Wf& w = slot(CM, 0);
      w.alloc(-pbf, tbf);
      w.at(0) = 0;
      for (int h = 1; h <= tbf; ++h) w.at(h) = h;
      for (int v = 1; v <= pbf; ++v) w.at(-v) = 0;
      cells += tbf + pbf + 1;
    }
  }

  // wavefront_compute_trim_ends
  void trim_ends(Wf& w) const {
    int k;
    for (k = w.hi; k >= w.lo; --k) {
      const int32_t off = w.at(k);
      const uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
      if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
    }
    w.hi = k;
    for (k = w.lo; k <= w.hi; ++k) {
      const int32_t off = w.at(k);
      const uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
      if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
    }
    w.lo = k;
    w.null = (w.lo > w.hi);
  }
  inline int32_t bound(int32_t mx, int k) const {  // "Adjust offset out of boundaries"
    const uint32_t h = (uint32_t)mx, v = (uint32_t)(mx - k);
    if (h > (uint32_t)tlen) mx = OFF_NULL;
    if (v > (uint32_t)plen) mx = OFF_NULL;
    return mx;
  }

  void ensure(int s) {
    for (int c = 0; c < 5; ++c)
      if ((int)wf[c].size() <= s) wf[c].resize((size_t)s + 1);
  }
  void compute(int s) {
    ensure(s);  // after this, references into wf[c] stay valid for the whole step
    switch (metric) {
      case M_INDEL: case M_EDIT: compute_edit(s); break;
      case M_LINEAR: compute_linear(s); break;
      case M_AFFINE: compute_affine(s); break;
      default: compute_affine2p(s); break;
    }
  }

  // wavefront_compute_edit (edit and indel share it)
  void compute_edit(int s) {
    const Wf* prevp = ptr(CM, s - 1);
    const Wf& prev = prevp ? *prevp : WF_NULL;
    const int lo = prev.lo - 1, hi = prev.hi + 1;
    Wf& out = slot(CM, s);
    out.alloc(lo, hi);
    cells += std::max(0, hi - lo + 1);
    for (int k = lo; k <= hi; ++k) {
      const int32_t ins = prev.get(k - 1), del = prev.get(k + 1), mis = prev.get(k);
      int32_t mx = (metric == M_EDIT) ? std::max(del, std::max(ins, mis) + 1) : std::max(del, ins + 1);
      out.at(k) = bound(mx, k);
    }
    trim_ends(out);
    if (out.null) num_null_steps = INT_MAX;
  }

  static inline void lim(const Wf& w, int dlo, int dhi, int& lo, int& hi) {
    if (lo > w.lo + dlo) lo = w.lo + dlo;
    if (hi < w.hi + dhi) hi = w.hi + dhi;
  }

  void null_step(int s, int ncomp) {
    ++num_null_steps;
    static const int comps[5] = {CM, CI1, CD1, CI2, CD2};
    for (int i = 0; i < ncomp; ++i) { Wf& w = slot(comps[i], s); w = Wf(); }
  }

  void compute_linear(int s) {
    const Wf &m_misms = fetch(CM, s - x), &m_open = fetch(CM, s - o1);
    if (m_misms.null && m_open.null) { null_step(s, 1); return; }
    num_null_steps = 0;
    int lo = m_misms.lo, hi = m_misms.hi;
    lim(m_open, -1, +1, lo, hi);
    Wf& out = slot(CM, s);
    out.alloc(lo, hi);
    cells += std::max(0, hi - lo + 1);
    for (int k = lo; k <= hi; ++k) {
      const int32_t ins = m_open.get(k - 1), del = m_open.get(k + 1), mis = m_misms.get(k);
      out.at(k) = bound(std::max(del, std::max(mis, ins) + 1), k);
    }
    trim_ends(out);
  }

  void compute_affine(int s) {
    const Wf &m_misms = fetch(CM, s - x), &m_open = fetch(CM, s - o1 - e1), &i_ext = fetch(CI1, s - e1), &d_ext = fetch(CD1, s - e1);
    if (m_misms.null && m_open.null && i_ext.null && d_ext.null) { null_step(s, 3); return; }
    num_null_steps = 0;
    int lo = m_misms.lo, hi = m_misms.hi;  // wavefront_compute_limits_input
    lim(m_open, -1, +1, lo, hi);
    lim(i_ext, +1, +1, lo, hi);
    lim(d_ext, -1, -1, lo, hi);
    const bool has_i = !m_open.null || !i_ext.null, has_d = !m_open.null || !d_ext.null;
    Wf out_m, out_i, out_d;
    out_m.alloc(lo, hi); out_i.alloc(lo, hi); out_d.alloc(lo, hi);
    cells += 3 * (int64_t)std::max(0, hi - lo + 1);
    for (int k = lo; k <= hi; ++k) {
      const int32_t ins = std::max(m_open.get(k - 1), i_ext.get(k - 1)) + 1;
      const int32_t del = std::max(m_open.get(k + 1), d_ext.get(k + 1));
      const int32_t mis = m_misms.get(k) + 1;
      out_i.at(k) = ins; out_d.at(k) = del;
      out_m.at(k) = bound(std::max(del, std::max(mis, ins)), k);
    }
    trim_ends(out_m);
    slot(CM, s) = std::move(out_m);
    if (has_i) { trim_ends(out_i); slot(CI1, s) = std::move(out_i); } else slot(CI1, s) = Wf();
    if (has_d) { trim_ends(out_d); slot(CD1, s) = std::move(out_d); } else slot(CD1, s) = Wf();
  }

  void compute_affine2p(int s) {
    const Wf &m_misms = fetch(CM, s - x), &m_open1 = fetch(CM, s - o1 - e1), &m_open2 = fetch(CM, s - o2 - e2);
    const Wf &i1_ext = fetch(CI1, s - e1), &i2_ext = fetch(CI2, s - e2), &d1_ext = fetch(CD1, s - e1), &d2_ext = fetch(CD2, s - e2);
    if (m_misms.null && m_open1.null && m_open2.null && i1_ext.null && i2_ext.null && d1_ext.null && d2_ext.null) {
      null_step(s, 5); return;
    }
    num_null_steps = 0;
    int lo = m_misms.lo, hi = m_misms.hi;
    lim(m_open1, -1, +1, lo, hi); lim(i1_ext, +1, +1, lo, hi); lim(d1_ext, -1, -1, lo, hi);
    lim(m_open2, -1, +1, lo, hi); lim(i2_ext, +1, +1, lo, hi); lim(d2_ext, -1, -1, lo, hi);
    const bool has_i1 = !m_open1.null || !i1_ext.null, has_d1 = !m_open1.null || !d1_ext.null;
    const bool has_i2 = !m_open2.null || !i2_ext.null, has_d2 = !m_open2.null || !d2_ext.null;
    Wf om, oi1, oi2, od1, od2;
    om.alloc(lo, hi); oi1.alloc(lo, hi); oi2.alloc(lo, hi); od1.alloc(lo, hi); od2.alloc(lo, hi);
    cells += 5 * (int64_t)std::max(0, hi - lo + 1);
    for (int k = lo; k <= hi; ++k) {
      const int32_t ins1 = std::max(m_open1.get(k - 1), i1_ext.get(k - 1)) + 1;
      const int32_t ins2 = std::max(m_open2.get(k - 1), i2_ext.get(k - 1)) + 1;
      const int32_t ins = std::max(ins1, ins2);
      const int32_t del1 = std::max(m_open1.get(k + 1), d1_ext.get(k + 1));
      const int32_t del2 = std::max(m_open2.get(k + 1), d2_ext.get(k + 1));
      const int32_t del = std::max(del1, del2);
      const int32_t mis = m_misms.get(k) + 1;
      oi1.at(k) = ins1; oi2.at(k) = ins2; od1.at(k) = del1; od2.at(k) = del2;
      om.at(k) = bound(std::max(del, std::max(mis, ins)), k);
    }
    trim_ends(om); slot(CM, s) = std::move(om);
    if (has_i1) { trim_ends(oi1); slot(CI1, s) = std::move(oi1); } else slot(CI1, s) = Wf();
    if (has_i2) { trim_ends(oi2); slot(CI2, s) = std::move(oi2); } else slot(CI2, s) = Wf();
    if (has_d1) { trim_ends(od1); slot(CD1, s) = std::move(od1); } else slot(CD1, s) = Wf();
    if (has_d2) { trim_ends(od2); slot(CD2, s) = std::move(od2); } else slot(CD2, s) = Wf();
  }

  // wavefront_termination_endsfree
  bool term_endsfree(int k, int32_t off) const {
    const int h = off, v = off - k;
    if (h >= tlen && plen - v <= pef) return true;
    if (v >= plen && tlen - h <= tef) return true;
    return false;
  }
  // wavefront_termination_end2end
  bool term_end2end(int s) {
    const int ak = tlen - plen; const int32_t aoff = tlen;
    const Wf* w = ptr(comp_end, s);
    if (comp_end != CM && w == nullptr) return false;
    if (w == nullptr) return false;
    if (w->lo > ak || ak > w->hi) return false;
    if (w->off[ak - w->base] < aoff) return false;
    end_score = s; end_k = ak; end_offset = aoff;
    return true;
  }

  // wavefront_heuristic_cufoff (wfadaptive only) -- SURVEY Appendix A.7 / F.4
  void heuristic_cutoff(int s) {
    Wf* m = (s < (int)wf[CM].size() && wf[CM][s].exists) ? &wf[CM][s] : nullptr;
    if (m == nullptr || m->lo > m->hi) return;
    --steps_wait;
    if (steps_wait <= 0) {
      const int base_lo = m->lo, base_hi = m->hi;
      if (base_hi - base_lo + 1 >= P_.h_min_wavefront_length) {
        std::vector<int> dist((size_t)(base_hi - base_lo + 1));
        int min_d = std::max(plen, tlen);
        for (int k = base_lo; k <= base_hi; ++k) {
          const int32_t off = m->at(k);
          int d;
          if (off < 0) d = -OFF_NULL;
          else if (span == 0) d = std::max(plen - (off - k), tlen - off);
          else {  // wf_compute_distance_endsfree
            const int left_v = plen - (off - k), left_h = tlen - off;
            d = std::min(std::max(left_h, left_v - pef), std::max(left_v, left_h - tef));
          }
          dist[k - base_lo] = d;
          min_d = std::min(min_d, d);
        }
        const int ak = tlen - plen, thr = P_.h_max_distance_threshold;
        // wf_heuristic_wfadaptive_reduce (preserve target diagonal)
        const int top_limit = std::min(ak, m->hi);
        int lo_red = m->lo;
        for (int k = m->lo; k < top_limit; ++k) {
          if (dist[k - base_lo] - min_d <= thr) break;
          ++lo_red;
        }
        m->lo = lo_red;
        const int bottom_limit = std::max(ak, m->lo);
        int hi_red = m->hi;
        for (int k = m->hi; k > bottom_limit; --k) {
          if (dist[k - base_lo] - min_d <= thr) break;
          --hi_red;
        }
        m->hi = hi_red;
        steps_wait = P_.h_steps_between_cutoffs;
      }
    }
    if (m->lo > m->hi) m->null = true;
    if (metric <= M_LINEAR) return;
    auto equate = [&](int c) {  // wavefront_heuristic_equate
      if (s >= (int)wf[c].size() || !wf[c][s].exists) return;
      Wf& d = wf[c][s];
      if (m->lo > d.lo) d.lo = m->lo;
      if (m->hi < d.hi) d.hi = m->hi;
      if (d.lo > d.hi) d.null = true;
    };
    equate(CI1); equate(CD1);
    if (metric == M_AFFINE2P) { equate(CI2); equate(CD2); }
  }

  // wavefront_extend_{end2end,end2end_max,endsfree}: returns 1 when done
  int extend(int s, int* max_ak, bool act_on_end = true) {
    if (max_ak) *max_ak = 0;
    Wf* m = (s < (int)wf[CM].size() && wf[CM][s].exists) ? &wf[CM][s] : nullptr;
    if (m == nullptr || m->null) {
      if (m == nullptr || metric <= M_EDIT) {
        if (num_null_steps > max_score_scope) { status = ST_END_UNREACHABLE; end_score = s; return 1; }
      }
      if (m == nullptr) return 0;
    }
    int32_t mak = 0;
    bool end_reached = false;
    for (int k = m->lo; k <= m->hi; ++k) {
      int32_t off = m->at(k);
      if (off < 0) continue;
      int v = off - k, h = off;
      while (v < plen && h < tlen && pat.at(v) == txt.at(h)) { ++v; ++h; }
      off = h;
      m->at(k) = off;
      const int32_t ad = 2 * off - k;
      if (mak < ad) mak = ad;
      if (span == 1 && term_endsfree(k, off)) {
        end_score = s; end_k = k; end_offset = off; end_reached = true; break;
      }
    }
    if (span == 0) end_reached = term_end2end(s);
    if (end_reached && act_on_end) { status = ST_END_REACHED; return 1; }
    if (P_.heuristic != 0) heuristic_cutoff(s);
    if (max_ak) *max_ak = mak;
    return 0;
  }

  // wavefront_unialign
  int run() {
    int s = 0;
    while (true) {
      if (extend(s, nullptr)) return status;
      ++s;
      compute(s);
    }
  }

  // ---- backtrace (wavefront_backtrace_{linear,affine}) -- Appendix A.6 / F.3
  inline int64_t bt(int c, int s, int k, int add, int type) const {
    const Wf* w = ptr(c, s);
    if (s < 0 || w == nullptr || k < w->lo || k > w->hi) return (int64_t)OFF_NULL;
    return (((int64_t)(w->off[k - w->base] + add)) << 4) | type;  // BACKTRACE_TYPE_BITS_SET
  }

  void backtrace() {
    std::string rev;  // ops pushed in reverse
    int mt = comp_end, s = end_score, k = end_k;
    int32_t off = end_offset;
    int h = off, v = off - k;
    if (comp_end == CM) {  // ending insertions/deletions (ends-free)
      for (int i = plen - v; i > 0; --i) rev.push_back('D');
      for (int i = tlen - h; i > 0; --i) rev.push_back('I');
    }
    const bool lin = metric <= M_LINEAR;
    while (v > 0 && h > 0 && s > 0) {
      int64_t best = (int64_t)OFF_NULL;
      auto take = [&](int64_t c) { if (c > best) best = c; };
      if (lin) {
        if (metric != M_INDEL) take(bt(CM, s - x, k, +1, 9));
        take(bt(CM, s - o1, k - 1, +1, 1));
        take(bt(CM, s - o1, k + 1, 0, 5));
      } else {
        if (mt == CM) take(bt(CM, s - x, k, +1, 9));
        if (mt == CM || mt == CD1) { take(bt(CD1, s - e1, k + 1, 0, 6)); take(bt(CM, s - o1 - e1, k + 1, 0, 5)); }
        if (mt == CM || mt == CI1) { take(bt(CI1, s - e1, k - 1, +1, 2)); take(bt(CM, s - o1 - e1, k - 1, +1, 1)); }
        if (metric == M_AFFINE2P) {
          if (mt == CM || mt == CD2) { take(bt(CD2, s - e2, k + 1, 0, 8)); take(bt(CM, s - o2 - e2, k + 1, 0, 7)); }
          if (mt == CM || mt == CI2) { take(bt(CI2, s - e2, k - 1, +1, 4)); take(bt(CM, s - o2 - e2, k - 1, +1, 3)); }
        }
      }
      if (best < 0) break;
      const int32_t best_off = (int32_t)(best >> 4);
      const int type = (int)(best & 0xF);
      if (mt == CM) {
        for (int i = off - best_off; i > 0; --i) rev.push_back('M');
        off = best_off; h = off; v = off - k;
        if (v <= 0 || h <= 0) break;
      }
      switch (type) {
        case 9: rev.push_back('X'); s -= x; mt = CM; --off; break;
        case 1: rev.push_back('I'); s -= lin ? o1 : (o1 + e1); mt = CM; --k; --off; break;
        case 2: rev.push_back('I'); s -= e1; mt = CI1; --k; --off; break;
        case 3: rev.push_back('I'); s -= o2 + e2; mt = CM; --k; --off; break;
        case 4: rev.push_back('I'); s -= e2; mt = CI2; --k; --off; break;
        case 5: rev.push_back('D'); s -= lin ? o1 : (o1 + e1); mt = CM; ++k; break;
        case 6: rev.push_back('D'); s -= e1; mt = CD1; ++k; break;
        case 7: rev.push_back('D'); s -= o2 + e2; mt = CM; ++k; break;
        case 8: rev.push_back('D'); s -= e2; mt = CD2; ++k; break;
        default: assert(!"bad backtrace type");
      }
      h = off; v = off - k;
    }
    if (mt == CM && v > 0 && h > 0) {
      const int n = std::min(v, h);
      for (int i = 0; i < n; ++i) rev.push_back('M');
      v -= n; h -= n;
    }
    for (; v > 0; --v) rev.push_back('D');
    for (; h > 0; --h) rev.push_back('I');
    ops.assign(rev.rbegin(), rev.rend());
  }
};

static inline int classic_score(int metric, int s) { return metric <= M_EDIT ? s : -s; }

// ---------------------------------------------------------------- BiWFA
struct Bi {
  const orc_wfa_params& p;
  const uint8_t* P; const uint8_t* T;
  std::string ops; int status = WF_COMPLETED; int score = INT32_MIN; int64_t cells = 0;

  int gap_open_adjust() const {  // o_max
    if (p.metric == M_AFFINE) return p.gap_open1;
    if (p.metric == M_AFFINE2P) return std::max(p.gap_open1, p.gap_open2);
    return 0;
  }

  // wavefront_bialign_breakpoint_{indel2indel,m2m}
  static void bp_check(const Uni& a0, bool fwd, int s0, int s1, const Wf* w0, const Wf* w1, int comp, int gap_open,
                       Breakpoint& bp) {
    const int plen = a0.plen, tlen = a0.tlen;
    const int lo0 = w0->lo, hi0 = w0->hi;
    const int lo1 = (tlen - plen) - w1->hi, hi1 = (tlen - plen) - w1->lo;
    if (hi1 < lo0 || hi0 < lo1) return;
    const int min_hi = std::min(hi0, hi1), max_lo = std::max(lo0, lo1);
    for (int k0 = max_lo; k0 <= min_hi; ++k0) {
      const int k1 = (tlen - plen) - k0;
      const int32_t h0 = w0->off[k0 - w0->base], h1 = w1->off[k1 - w1->base];
      if (h0 + h1 >= tlen && s0 + s1 - gap_open < bp.score) {
        if (fwd) {
          const int v = h0 - k0, h = h0;
          if (v > plen || h > tlen) continue;
          bp.score_forward = s0; bp.score_reverse = s1; bp.k_forward = k0; bp.k_reverse = k1;
          bp.offset_forward = h0; bp.offset_reverse = h1;
        } else {
          const int v = h1 - k1, h = h1;
          if (v > plen || h > tlen) continue;
          bp.score_forward = s1; bp.score_reverse = s0; bp.k_forward = k1; bp.k_reverse = k0;
          bp.offset_forward = h1; bp.offset_reverse = h0;
        }
        bp.score = s0 + s1 - gap_open; bp.component = comp;
        return;
      }
    }
  }

  // wavefront_bialign_overlap
  void overlap(const Uni& a0, const Uni& a1, int s0, int s1, bool fwd, Breakpoint& bp) const {
    const int scope = a0.max_score_scope, metric = p.metric;
    const Wf* m0 = a0.ptr(CM, s0);
    if (m0 == nullptr) return;
    const Wf *d10 = nullptr, *i10 = nullptr, *d20 = nullptr, *i20 = nullptr;
    if (metric >= M_AFFINE) { d10 = a0.ptr(CD1, s0); i10 = a0.ptr(CI1, s0); }
    if (metric == M_AFFINE2P) { d20 = a0.ptr(CD2, s0); i20 = a0.ptr(CI2, s0); }
    for (int i = 0; i < scope; ++i) {
      const int si = s1 - i;
      if (si < 0) break;
      if (metric == M_AFFINE2P && s0 + si - p.gap_open2 < bp.score) {
        const Wf* d21 = a1.ptr(CD2, si);
        if (d20 && d21) bp_check(a0, fwd, s0, si, d20, d21, CD2, p.gap_open2, bp);
        const Wf* i21 = a1.ptr(CI2, si);
        if (i20 && i21) bp_check(a0, fwd, s0, si, i20, i21, CI2, p.gap_open2, bp);
      }
      if (metric >= M_AFFINE && s0 + si - p.gap_open1 < bp.score) {
        const Wf* d11 = a1.ptr(CD1, si);
        if (d10 && d11) bp_check(a0, fwd, s0, si, d10, d11, CD1, p.gap_open1, bp);
        const Wf* i11 = a1.ptr(CI1, si);
        if (i10 && i11) bp_check(a0, fwd, s0, si, i10, i11, CI1, p.gap_open1, bp);
      }
      if (s0 + si >= bp.score) continue;
      const Wf* m1 = a1.ptr(CM, si);
      if (m1) bp_check(a0, fwd, s0, si, m0, m1, CM, 0, bp);
    }
  }

  // wavefront_bialign_find_breakpoint; returns ST_OK / ST_END_REACHED / ST_END_UNREACHABLE
  int find_breakpoint(int pb, int pl, int tb, int tl, int cb, int ce, Breakpoint& bp) {
    orc_wfa_params ps = p;
    Uni F(ps, SeqView{P + pb, pl, false}, SeqView{T + tb, tl, false}, 0, 0, 0, 0, 0, cb, CM);
    Uni R(ps, SeqView{P + pb, pl, true}, SeqView{T + tb, tl, true}, 0, 0, 0, 0, 0, ce, CM);
    struct Acc { Bi* b; Uni* f; Uni* r; ~Acc() { b->cells += f->cells + r->cells; } } acc{this, &F, &R};
    const int max_antidiagonal = pl + tl - 1;
    int sf = 0, sr = 0, fak = 0, rak = 0, mak = 0;
    if (F.extend(sf, &fak)) return F.status;
    if (R.extend(sr, &rak)) return R.status;
    bool last_forward = false;
    while (true) {
      if (fak + rak >= max_antidiagonal) break;
      ++sf; F.compute(sf);
      if (F.extend(sf, &mak)) return F.status;
      if (fak < mak) fak = mak;
      last_forward = true;
      if (fak + rak >= max_antidiagonal) break;
      ++sr; R.compute(sr);
      if (R.extend(sr, &mak)) return R.status;
      if (rak < mak) rak = mak;
      last_forward = false;
    }
    const int scope = F.max_score_scope, gap_opening = gap_open_adjust();
    bp.score = INT_MAX;
    while (true) {
      if (last_forward) {
        const int min_sr = (sr > scope - 1) ? sr - (scope - 1) : 0;
        if (sf + min_sr - gap_opening >= bp.score) break;
        overlap(F, R, sf, sr, true, bp);
        ++sr; R.compute(sr);
        if (R.extend(sr, nullptr, false)) return R.status;  // phase 2: only a dead front ends the search (A.7/F.5)
      }
      const int min_sf = (sf > scope - 1) ? sf - (scope - 1) : 0;
      if (min_sf + sr - gap_opening >= bp.score) break;
      overlap(R, F, sr, sf, false, bp);
      ++sf; F.compute(sf);
      if (F.extend(sf, nullptr, false)) return F.status;
      last_forward = true;
    }
    return ST_OK;
  }

  // wavefront_bialign_base
  void base(int pb, int pl, int tb, int tl, int cb, int ce) {
    Uni U(p, SeqView{P + pb, pl, false}, SeqView{T + tb, tl, false}, 0, 0, 0, 0, 0, cb, ce);
    const int st = U.run();
    cells += U.cells;
    if (st != ST_END_REACHED) { status = WF_UNATTAINABLE; return; }
    U.backtrace();
    ops += U.ops;
  }

  // wavefront_bialign_alignment
  void align(int pb, int pl, int tb, int tl, int cb, int ce, int score_remaining, int level) {
    if (status != WF_COMPLETED) return;
    if (tl == 0) { ops.append((size_t)pl, 'D'); return; }
    if (pl == 0) { ops.append((size_t)tl, 'I'); return; }
    if (score_remaining <= p.bialign_min_score) { base(pb, pl, tb, tl, cb, ce); return; }
    Breakpoint bp;
    const int st = find_breakpoint(pb, pl, tb, tl, cb, ce, bp);
    if (st == ST_END_REACHED) { base(pb, pl, tb, tl, cb, ce); return; }
    if (st != ST_OK) { status = WF_UNATTAINABLE; return; }
    const int bh = bp.offset_forward, bv = bp.offset_forward - bp.k_forward;
    align(pb, bv, tb, bh, cb, bp.component, bp.score_forward, level + 1);
    align(pb + bv, pl - bv, tb + bh, tl - bh, bp.component, ce, bp.score_reverse, level + 1);
    score = classic_score(p.metric, bp.score);
  }

  // wavefront_bialign_compute_score
  void score_only(int pl, int tl) {
    Breakpoint bp;
    const int st = find_breakpoint(0, pl, 0, tl, CM, CM, bp);
    if (st == ST_END_REACHED) {
      Uni U(p, SeqView{P, pl, false}, SeqView{T, tl, false}, 0, 0, 0, 0, 0, CM, CM);
      const int s2 = U.run();
      cells += U.cells;
      if (s2 != ST_END_REACHED) { status = WF_UNATTAINABLE; return; }
      score = classic_score(p.metric, U.end_score);
      return;
    }
    if (st != ST_OK) { status = WF_UNATTAINABLE; return; }
    score = classic_score(p.metric, bp.score);
  }
};

// ---------------------------------------------------------- entry point
WfaResult wfa_align(const orc_wfa_params& p, const uint8_t* pattern, int plen, const uint8_t* text, int tlen) {
  WfaResult r;
  if (p.memory_mode == 3) {  // MemoryUltraLow -> BiWFA (wavefront_bialign)
    Bi b{p, pattern, text, std::string(), WF_COMPLETED, INT32_MIN, 0};
    if (p.scope == 0) b.score_only(plen, tlen);
    else {
      const bool min_length = std::max(plen, tlen) <= p.bialign_min_length;
      b.align(0, plen, 0, tlen, CM, CM, min_length ? 0 : INT_MAX, 0);
    }
    r.cells = b.cells;
    r.status = b.status;
    if (b.status == WF_COMPLETED) { r.score = b.score; if (p.scope != 0) r.ops = b.ops; }
    else r.score = INT32_MIN;
    return r;
  }
  const int span = p.span;
  auto fr = [](int v, int len) { return v < 0 ? len : v; };
  Uni U(p, SeqView{pattern, plen, false}, SeqView{text, tlen, false}, span, span ? fr(p.pattern_begin_free, plen) : 0,
        span ? fr(p.pattern_end_free, plen) : 0, span ? fr(p.text_begin_free, tlen) : 0, span ? fr(p.text_end_free, tlen) : 0,
        CM, CM);
  const int st = U.run();
  r.cells = U.cells;
  if (st != ST_END_REACHED) { r.status = WF_UNATTAINABLE; r.score = INT32_MIN; return r; }
  r.status = WF_COMPLETED;
  r.score = classic_score(p.metric, U.end_score);
  if (p.scope != 0) { U.backtrace(); r.ops = U.ops; }
  return r;
}

int cigar_count_matches(const std::string& ops) {  // cigar_count_matches (wfaligner.rs:988-1000)
  int n = 0;
  for (char c : ops) n += (c == 'M');
  return n;
}

// get_alignment_span (wfaligner.rs:864-908)
void alignment_span(const orc_wfa_params& p, const std::string& ops, int plen, int tlen, uint32_t s4[4]) {
  if (p.span == 0) { s4[0] = 0; s4[1] = plen; s4[2] = 0; s4[3] = tlen; return; }
  uint32_t pi = 0, ti = 0, ps = 0, pe = 0, ts = 0, te = 0; bool started = false;
  for (char c : ops) {
    if (c == 'I') ++ti;
    else if (c == 'D') ++pi;
    else { if (!started) { ps = pi; ts = ti; started = true; } ++pi; ++ti; pe = pi; te = ti; }
  }
  s4[0] = ps; s4[1] = pe; s4[2] = ts; s4[3] = te;
}

// cigar_get_CIGAR (Appendix A.8; wfaligner.rs:1609-1676)
std::vector<uint32_t> cigar_rle(const std::string& ops, bool show_mismatches) {
  std::vector<uint32_t> out;
  auto code = [&](char c) -> uint32_t {
    switch (c) { case 'M': return show_mismatches ? 7u : 0u; case 'X': return show_mismatches ? 8u : 0u; case 'I': return 1u; default: return 2u; }
  };
  size_t i = 0;
  while (i < ops.size()) {
    const uint32_t c = code(ops[i]);
    size_t j = i;
    while (j < ops.size() && code(ops[j]) == c) ++j;
    out.push_back((uint32_t)((j - i) << 4) | c);
    i = j;
  }
  return out;
}

static int op_score(const orc_wfa_params& p, char op, int len) {  // wfaligner.rs:534-593
  switch (p.metric) {
    case M_INDEL: case M_EDIT: return op == 'M' ? 0 : len;
    case M_LINEAR: return op == 'M' ? 0 : (op == 'X' ? len * p.mismatch : len * p.gap_ext1);
    case M_AFFINE: return op == 'M' ? 0 : (op == 'X' ? len * p.mismatch : p.gap_open1 + p.gap_ext1 * len);
    default:
      return op == 'M' ? 0 : (op == 'X' ? len * p.mismatch
                                        : std::min(p.gap_open1 + p.gap_ext1 * len, p.gap_open2 + p.gap_ext2 * len));
  }
}
static int score_range(const orc_wfa_params& p, const uint8_t* ops, int b, int e) {
  if (b >= e) return 0;
  int score = 0, i = b;
  while (i < e) {
    int j = i;
    while (j < e && ops[j] == ops[i]) ++j;
    const int s = op_score(p, (char)ops[i], j - i);
    score += (p.metric <= M_EDIT) ? s : -s;
    i = j;
  }
  return score;
}

}  // namespace orc

using namespace orc;
extern "C" {

void orc_wfa_default_params(orc_wfa_params* p) {  // wavefront_aligner_attr_default (Appendix A.7 "Defaults")
  std::memset(p, 0, sizeof(*p));
  p->metric = M_AFFINE; p->mismatch = 4; p->gap_open1 = 6; p->gap_ext1 = 2; p->gap_open2 = 24; p->gap_ext2 = 1;
  p->span = 0; p->scope = 1; p->memory_mode = 0;
  p->heuristic = 1; p->h_min_wavefront_length = 10; p->h_max_distance_threshold = 50; p->h_steps_between_cutoffs = 1;
  p->bialign_min_score = 250; p->bialign_min_length = 100;
}

int orc_wfa_align(const orc_wfa_params* p, const uint8_t* pattern, int plen, const uint8_t* text, int tlen, int32_t* score,
                  uint8_t* ops, int32_t* ops_len, int32_t* n_match, uint32_t* span4, int64_t* cells) {
  WfaResult r = wfa_align(*p, pattern, plen, text, tlen);
  if (score) *score = r.score;
  if (ops) std::memcpy(ops, r.ops.data(), r.ops.size());
  if (ops_len) *ops_len = (int32_t)r.ops.size();
  if (n_match) *n_match = cigar_count_matches(r.ops);
  if (span4) alignment_span(*p, r.ops, plen, tlen, span4);
  if (cells) *cells = r.cells;
  return r.status;
}

int orc_wfa_batch(const orc_wfa_params* p, int64_t n_jobs, const uint8_t* seqs, const uint64_t* pat_off,
                  const uint32_t* pat_len, const uint64_t* txt_off, const uint32_t* txt_len, int32_t* status, int32_t* score,
                  int32_t* n_match, uint32_t* span4, uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len,
                  uint8_t* ops, const uint64_t* ops_off, uint32_t* ops_len, int64_t* cells_total, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  std::vector<int64_t> cells(n_threads, 0);
  auto work = [&](int t) {
    for (int64_t j = t; j < n_jobs; j += n_threads) {
      WfaResult r = wfa_align(*p, seqs + pat_off[j], (int)pat_len[j], seqs + txt_off[j], (int)txt_len[j]);
      cells[t] += r.cells;
      if (status) status[j] = r.status;
      if (score) score[j] = r.score;
      if (n_match) n_match[j] = cigar_count_matches(r.ops);
      if (span4) alignment_span(*p, r.ops, (int)pat_len[j], (int)txt_len[j], span4 + 4 * j);
      if (cigar) {
        auto c = cigar_rle(r.ops, true);
        std::memcpy(cigar + cigar_off[j], c.data(), c.size() * 4);
        cigar_len[j] = (uint32_t)c.size();
      }
      if (ops) { std::memcpy(ops + ops_off[j], r.ops.data(), r.ops.size()); }
      if (ops_len) ops_len[j] = (uint32_t)r.ops.size();
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

int orc_cigar_rle(const uint8_t* ops, int n, int show_mismatches, uint32_t* out, int cap) {
  auto c = cigar_rle(std::string((const char*)ops, n), show_mismatches != 0);
  if ((int)c.size() > cap) return -1;
  std::memcpy(out, c.data(), c.size() * 4);
  return (int)c.size();
}
int orc_cigar_score(const orc_wfa_params* p, const uint8_t* ops, int n) { return score_range(*p, ops, 0, n); }
int orc_cigar_score_clipped(const orc_wfa_params* p, const uint8_t* ops, int n, int flank) {
  const int b = flank, e = std::max(b, n - flank);
  return score_range(*p, ops, b, e);
}

}  // extern "C"
