/*
 * oracle/oracle.h -- C API of the CPU ORACLE.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain, scalar CPU
 * restatement of the reference algorithms on the hot path of
 * PacificBiosciences/trgt v3.0.0 (the motif HMM in src/hmm/ and the WFA2-lib
 * wavefront aligner reached through src/wfaligner.rs).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (trgt_amd/libtrgt_hip.so) never links, loads or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - HMM: pinned by every live reference KAT (builder.rs:208-273,
 *     purity.rs:48-96, events.rs:124-145) -> tests/golden/hmm_kats.json.
 *   - WFA, exact unidirectional modes: pinned by wfaligner.rs:1136-1381,
 *     1383-1421, 1589-1676, 1718-1828 -> tests/golden/wfa_kats.json.
 *   - WFA, BiWFA (MemoryUltraLow) + default wfadaptive heuristic: PARITY
 *     UNPINNED -- WFA2-lib is an un-vendored git dependency (wfa2-sys 0.1.0,
 *     git ctsa/rust-wfa2 rev 4342b3b0, Cargo.lock:1839-1845) whose source is
 *     absent from /root/reference; the only evidence the reference holds is
 *     wfaligner.rs:1437-1454 (Unattainable / -881), reproduced here.
 */
#ifndef TRGT_ORACLE_H
#define TRGT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ HMM */
/* motifs: n_motifs byte strings concatenated in `motif_blob`, motif i is
 * motif_blob[motif_off[i] .. motif_off[i+1]).  Motifs must already be over
 * ATCGN (callers sanitise with orc_replace_invalid_bases, tr.rs:455-460). */

/* Hmm::label (hmm_model.rs:144-156): state path of '#'+seq+'#'.  seq must be
 * over ATCG.  Returns path length (0 for empty seq), or -1 if cap too small. */
int orc_hmm_label(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                  const uint8_t* seq, int seq_len, int32_t* path, int cap);
int orc_hmm_num_states(const uint32_t* motif_off, int n_motifs);
/* remove_imperfect_motifs (operations.rs:6-80) */
int orc_hmm_remove_imperfect(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                             const int32_t* path, int path_len, const uint8_t* seq, int seq_len,
                             int max_motif_len, int32_t* out, int cap);
/* Hmm::label_motifs (hmm_model.rs:158-200): spans3 = (motif_index,start,end)* */
int orc_hmm_label_motifs(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                         const int32_t* path, int path_len, int32_t* spans3, int cap);
/* get_events (events.rs:17-86): codes 0 Match 1 Mismatch 2 Ins 3 Del 4 Trans
 * 5 Skip 6 MotifStart 7 MotifEnd */
int orc_hmm_events(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                   const int32_t* path, int path_len, const uint8_t* seq, int seq_len,
                   uint8_t* events, int cap);
/* calc_purity (purity.rs:6-41); also returns the two integers it divides */
double orc_hmm_purity(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                      const int32_t* path, int path_len, const uint8_t* seq, int seq_len,
                      int32_t* edit_dist, int32_t* max_dist);
/* get_base_match (events.rs:88-117) */
int orc_hmm_base_match(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs, int state);
/* get_base_match on a hand-made model (Hmm::new + set_ems with probabilities, hmm_model.rs:28-52; events.rs:138-145) */
int orc_hmm_base_match_ems(int n_states, const double* ems_probs /* 5 per state */, int state);
/* replace_invalid_bases (utils.rs:29-42); allowed is a NUL-terminated string */
void orc_replace_invalid_bases(uint8_t* seq, int len, const char* allowed);

/* label_with_hmm for ONE allele (tr.rs:454-492): sanitise -> label -> purity
 * -> remove_imperfect(6) -> label_motifs -> drop skip spans -> count ->
 * collapse.  Outputs: path (unmodified label() result), collapsed spans,
 * per-motif counts, purity.  Returns 0 or -1 on capacity error. */
int orc_hmm_annotate(const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                     const uint8_t* seq, int seq_len,
                     int32_t* path, int path_cap, int32_t* path_len,
                     int32_t* spans3, int span_cap, int32_t* n_spans,
                     int32_t* motif_counts, double* purity, int32_t* edit_dist, int32_t* max_dist,
                     int64_t* viterbi_cells);

/* Batch form with the same array layout as the product ABI trgt_hmm_batch
 * (include/trgt_hip.h); used by differential tests and the cpu_baseline. */
int orc_hmm_batch(int n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                  const uint32_t* set_motif_begin,
                  int64_t n_jobs, const uint32_t* job_set,
                  const uint8_t* seq_blob, const uint64_t* seq_off, const uint32_t* seq_len,
                  uint16_t* path, const uint64_t* path_off, uint32_t* path_len,
                  int32_t* spans3, const uint64_t* span_off, uint32_t* n_spans,
                  uint32_t* motif_counts, const uint64_t* count_off,
                  double* purity, int32_t* edit_dist, int32_t* max_dist,
                  int64_t* viterbi_cells_total, int n_threads);

/* ------------------------------------------------------------------ WFA */
typedef struct orc_wfa_params {
  int32_t metric;            /* 0 indel 1 edit 2 gap-linear 3 gap-affine 4 gap-affine-2p */
  int32_t mismatch;          /* x (linear/affine/affine2p) */
  int32_t gap_open1;         /* o1 (affine); unused for linear */
  int32_t gap_ext1;          /* e1 (affine) ; indel penalty for gap-linear */
  int32_t gap_open2, gap_ext2;
  int32_t span;              /* 0 end-to-end, 1 ends-free */
  int32_t pattern_begin_free, pattern_end_free, text_begin_free, text_end_free; /* -1 => sequence length */
  int32_t scope;             /* 0 score only, 1 full alignment */
  int32_t memory_mode;       /* 0 high 1 med 2 low (identical results) 3 ultralow (BiWFA) */
  int32_t heuristic;         /* 0 none, 1 wfadaptive */
  int32_t h_min_wavefront_length, h_max_distance_threshold, h_steps_between_cutoffs;
  int32_t bialign_min_score;  /* WF_BIALIGN_FALLBACK_MIN_SCORE  (250) */
  int32_t bialign_min_length; /* WF_BIALIGN_FALLBACK_MIN_LENGTH (100; 0 disables) */
} orc_wfa_params;

void orc_wfa_default_params(orc_wfa_params* p); /* wavefront_aligner_attr_default */

/* One alignment.  ops receives M/X/I/D bytes (capacity plen+tlen);
 * span4 = pattern_start, pattern_end, text_start, text_end
 * (get_alignment_span, wfaligner.rs:864-908).  Returns WF status
 * (0 completed, 1 partial, -100 max steps, -200 OOM, -300 unattainable). */
int orc_wfa_align(const orc_wfa_params* p, const uint8_t* pattern, int plen,
                  const uint8_t* text, int tlen,
                  int32_t* score, uint8_t* ops, int32_t* ops_len,
                  int32_t* n_match, uint32_t* span4, int64_t* cells);

/* Batch form, same layout as trgt_wfa_batch. */
int orc_wfa_batch(const orc_wfa_params* p, int64_t n_jobs, const uint8_t* seqs,
                  const uint64_t* pat_off, const uint32_t* pat_len,
                  const uint64_t* txt_off, const uint32_t* txt_len,
                  int32_t* status, int32_t* score, int32_t* n_match, uint32_t* span4,
                  uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len,
                  uint8_t* ops, const uint64_t* ops_off, uint32_t* ops_len,
                  int64_t* cells_total, int n_threads);

/* CIGAR utilities (Appendix A.8) */
int orc_cigar_rle(const uint8_t* ops, int n, int show_mismatches, uint32_t* out, int cap); /* cigar_get_CIGAR */
int orc_cigar_score(const orc_wfa_params* p, const uint8_t* ops, int n);                   /* cigar_score_* */
int orc_cigar_score_clipped(const orc_wfa_params* p, const uint8_t* ops, int n, int flank); /* wfaligner.rs:595-705 */

/* ------------------------------------------------- callers / locus path */
/* find_spans (span_locater.rs:7-30) for one piece over many reads. */
int orc_find_spans(const uint8_t* piece, int piece_len, int64_t n_reads, const uint8_t* read_blob,
                   const uint64_t* read_off, const uint32_t* read_len,
                   int mism, int gapo, int gape, double threshold,
                   int32_t* start, int32_t* end, int32_t* used_wfa, int64_t* cells);

typedef struct orc_locus_params {
  int32_t flank_len;          /* --flank-len 250 (cli.rs:295-302) */
  double  min_flank_id_frac;  /* 0.7 */
  int32_t max_depth;          /* 250 */
  int32_t mism, gapo, gape;   /* --aln-scoring 2,5,1 */
  int32_t ploidy;             /* 1 or 2 */
  int32_t genotyper;          /* 0 Genotyper::Size, 1 Genotyper::Cluster (locus.rs:25-29) */
  double  min_read_qual;      /* Params::min_read_qual; < 0.9 switches filter_impure_trs on (tr.rs:37-50) */
} orc_locus_params;

/* analyze_tr restricted to pre-clipped reads, no HP/SNV/meth (tr.rs:24-109
 * minus BAM); size or cluster genotyper.  Text outputs are written as
 * NUL-terminated strings into caller buffers.  read_qual: per input read,
 * NaN = None; NULL = every read None (HiFiRead::read_qual, reads/read.rs). */
int orc_locus_analyze(const orc_locus_params* p,
                      const uint8_t* left_flank, int lf_len, const uint8_t* right_flank, int rf_len,
                      const uint8_t* ref_tr, int ref_tr_len,
                      const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                      int64_t n_reads, const uint8_t* read_blob, const uint64_t* read_off, const uint32_t* read_len,
                      int32_t* span_start, int32_t* span_end,      /* per input read, -1 = None (find_tr_spans) */
                      int32_t* n_alleles, char* allele0, char* allele1, int allele_cap,
                      int32_t* gt_size, int32_t* gt_ci,            /* [2], [4] */
                      int32_t* n_spanning, int32_t* kept_read, int32_t* classification, /* per spanning read */
                      int32_t* num_spanning_by_hap,                /* [2] */
                      char* mc, char* ms, char* ap, int str_cap,   /* VCF encodings write_vcf.rs:286-343 */
                      int64_t* stats /* [8]: wfa_cells, viterbi_cells, n_wfa_flank, n_wfa_cons, bytes_io, n_wfa_ed, n_purity */,
                      const double* read_qual);

/* ... with the per-read fields genotype_flank::genotype reads (HiFiRead::hp_tag, start_offset, end_offset, mismatch_offsets; reads/read.rs,
 * genotype_flank.rs:9-290): the locus is re-genotyped from them when its two alleles differ by at most 10 bases (tr.rs:69-75).
 * Arrays per INPUT read; any of them may be NULL (no HP tags / offsets 0 / no mismatches); meta NULL = orc_locus_analyze. */
typedef struct orc_read_meta {
  const int16_t* hp_tag;            /* -1 = None */
  const int32_t* start_offset; const int32_t* end_offset;
  const int32_t* mismatch_offsets; const uint64_t* mismatch_off;  /* CSR, [n_reads + 1] */
} orc_read_meta;
int orc_locus_analyze_meta(const orc_locus_params* p,
                           const uint8_t* left_flank, int lf_len, const uint8_t* right_flank, int rf_len,
                           const uint8_t* ref_tr, int ref_tr_len,
                           const uint8_t* motif_blob, const uint32_t* motif_off, int n_motifs,
                           int64_t n_reads, const uint8_t* read_blob, const uint64_t* read_off, const uint32_t* read_len,
                           int32_t* span_start, int32_t* span_end, int32_t* n_alleles, char* allele0, char* allele1, int allele_cap,
                           int32_t* gt_size, int32_t* gt_ci, int32_t* n_spanning, int32_t* kept_read, int32_t* classification,
                           int32_t* num_spanning_by_hap, char* mc, char* ms, char* ap, int str_cap, int64_t* stats,
                           const double* read_qual, const orc_read_meta* meta);

/* flank re-genotyping alone (genotype_flank.rs:9-42) on n reads given by their repeat sequences and metadata: returns 1 and the
 * genotype (sizes[2], ci[4], alleles, assignment[n]) or 0 for None */
int orc_genotype_flank(int n, const uint8_t* tr_blob, const uint64_t* tr_off, const uint32_t* tr_len, const orc_read_meta* meta,
                       int32_t* sizes, int32_t* ci, char* allele0, char* allele1, int allele_cap, int32_t* assignment);

/* orc_locus_analyze over loci [first, first + n) of a batch in the trgt_locus_batch_in layout, on n_threads threads (static
 * partition).  Returns the number of loci analysed; cpu_baseline helper of bench.py. */
int64_t orc_locus_analyze_many(const orc_locus_params* p, int64_t first, int64_t n, const uint8_t* flank_blob, const uint64_t* lf_off,
                               const uint32_t* lf_len, const uint64_t* rf_off, const uint32_t* rf_len, const uint8_t* tr_blob,
                               const uint64_t* tr_off, const uint32_t* tr_len, const uint8_t* motif_blob, const uint32_t* motif_off,
                               const uint32_t* set_motif_begin, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                               const uint64_t* read_off, const uint32_t* read_len, int n_threads, int64_t* alleles_out);

/* The same walk, writing one text record per locus (NUL-terminated, rec_stride bytes apart; "OVERFLOW" if it does not fit):
 *   S:<span_start>,<span_end>;..|A:<alleles>|K:<kept reads>|C:<classification>|ALLR:..|SD:..|MC:..|MS:..|AP:..
 * genotyper / ploidy: one byte per locus, NULL = the values in *p.  Whole-catalog parity sweeps (tests/tools/parity_sweep.py). */
int64_t orc_locus_analyze_records(const orc_locus_params* p, int64_t first, int64_t n, const uint8_t* flank_blob, const uint64_t* lf_off,
                                  const uint32_t* lf_len, const uint64_t* rf_off, const uint32_t* rf_len, const uint8_t* tr_blob,
                                  const uint64_t* tr_off, const uint32_t* tr_len, const uint8_t* motif_blob, const uint32_t* motif_off,
                                  const uint32_t* set_motif_begin, const uint64_t* locus_read_begin, const uint8_t* read_blob,
                                  const uint64_t* read_off, const uint32_t* read_len, int n_threads, const uint8_t* genotyper,
                                  const uint8_t* ploidy, char* rec_blob, uint64_t rec_stride,
                                  const double* read_qual /* one per read of the batch (NaN = no rq tag), or NULL */);

/* haploid::genotype / diploid::genotype on a histogram of repeat lengths (haploid.rs:3-15, diploid.rs:5-49): gt3 receives
 * (size, ci_lo, ci_hi) per allele; returns the number of alleles (1 or 2). */
int orc_genotype_sizes(int ploidy, const int32_t* sizes, const int32_t* counts, int n, int32_t* gt3);

/* Ward linkage as kodama 0.3.0's linkage(.., Method::Ward) performs it (PARITY UNPINNED, see locus.cpp): dists is
 * the condensed matrix, overwritten as kodama overwrites it.  Returns the number of steps (n-1). */
int orc_ward_linkage(double* dists, int n, int32_t* steps3 /* cluster1, cluster2, size */, double* dissimilarity);
int orc_cluster_groups(double* dists, int n, int32_t* group_of);

#ifdef __cplusplus
}
#endif
#endif
