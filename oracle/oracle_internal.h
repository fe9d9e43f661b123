// oracle/oracle_internal.h -- CPU ORACLE internals (test infrastructure only).
#ifndef TRGT_ORACLE_INTERNAL_H
#define TRGT_ORACLE_INTERNAL_H
#include <algorithm>
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "oracle.h"

namespace orc {

// ------------------------------------------------------------------- HMM
struct HmmMotif { int start_state, end_state, motif_index; };  // hmm_model.rs:21-26

struct Hmm {  // hmm_model.rs:13-19
  int num_states;
  std::vector<std::array<double, 5>> ems;
  std::vector<std::vector<int>> in_states;
  std::vector<std::vector<double>> in_lps;
  std::vector<HmmMotif> motifs;
  explicit Hmm(int n);
  void set_trans(int target, std::vector<int> ins, std::vector<double> probs);
  void set_ems(int target, const std::array<double, 5>& e);
  bool is_silent(int s) const;
  bool emits_any(int s) const;
  bool emits_base(int s) const;
};

struct Span { int motif_index, start, end; };  // spans.rs:1-6
struct Annotation {                           // spans.rs:20-25 (labels empty == None)
  std::vector<Span> labels;
  std::vector<int> motif_counts;
  double purity = 0;
  int edit_dist = 0, max_dist = 0;
};

enum : uint8_t { EV_MATCH = 0, EV_MISMATCH, EV_INS, EV_DEL, EV_TRANS, EV_SKIP, EV_MOTIF_START, EV_MOTIF_END };

Hmm build_hmm(const std::vector<std::string>& motifs);
std::vector<int> hmm_label(const Hmm& h, const std::string& seq, int64_t* cells);
std::vector<Span> hmm_label_motifs(const Hmm& h, const std::vector<int>& states);
int hmm_base_match(const Hmm& h, int state);
std::vector<uint8_t> hmm_events(const Hmm& h, const std::vector<std::string>& motifs, const std::vector<int>& states,
                                const std::string& query);
double hmm_purity(const Hmm& h, const std::vector<std::string>& motifs, const std::vector<int>& states,
                  const std::string& query, int* edit_out, int* max_out);
std::vector<int> hmm_remove_imperfect(const Hmm& h, const std::vector<std::string>& motifs, const std::vector<int>& states,
                                      const std::string& query, int max_motif_len);
std::string replace_invalid_bases(const std::string& seq, const std::string& allowed);
Annotation annotate_allele(const Hmm& h, const std::vector<std::string>& motifs, const std::string& raw_seq,
                           std::vector<int>* path_out, int64_t* cells);
std::vector<std::string> motifs_from_blob(const uint8_t* blob, const uint32_t* off, int n);

// ------------------------------------------------------------------- WFA
struct WfaResult {
  int status = 0;
  int score = INT32_MIN;       // cigar.score (classic score; INT32_MIN when failed)
  std::string ops;             // M X I D between begin_offset and end_offset
  int64_t cells = 0;           // wavefront offsets computed (all components, all levels, all sub-aligners)
};
WfaResult wfa_align(const orc_wfa_params& p, const uint8_t* pattern, int plen, const uint8_t* text, int tlen);
int cigar_count_matches(const std::string& ops);
void alignment_span(const orc_wfa_params& p, const std::string& ops, int plen, int tlen, uint32_t span4[4]);
std::vector<uint32_t> cigar_rle(const std::string& ops, bool show_mismatches);

// ---------------------------------------------------------------- callers
struct SpanOpt { int start = -1, end = -1; bool some() const { return start >= 0; } };
SpanOpt find_span(const uint8_t* piece, int piece_len, const uint8_t* s, int slen, const orc_wfa_params& flank_params,
                  double threshold, bool* used_wfa, int64_t* cells);
}  // namespace orc
#endif
