/*
 * include/trgt_hip.h -- C ABI of libtrgt_hip.so: the MI355X (gfx950) batch
 * implementation of TRGT's per-locus alignment/DP hot path.
 *
 * What each entry point replaces in PacificBiosciences/trgt v3.0.0
 * (paths relative to the reference root):
 *
 *   trgt_wfa_batch        the WFA2-lib C calls bound by wfa2-sys and wrapped by
 *                         src/wfaligner.rs: wavefront_aligner_new (:371),
 *                         wavefront_aligner_set_alignment_end_to_end (:467),
 *                         wavefront_aligner_set_alignment_free_ends (:479-485),
 *                         wavefront_align (:492-498, :519-525),
 *                         cigar_get_CIGAR (:944-949), cigar_count_matches (:999)
 *                         and the direct struct reads cigar->{operations,
 *                         begin_offset,end_offset,score} (:531,:600-611,:875-881)
 *                         -- one call per *batch* instead of one per alignment.
 *   trgt_find_spans_batch find_spans / find_tr_spans, src/trgt/genotype/
 *                         span_locater.rs:7-68 (exact 250-mer search + WFA
 *                         ends-free fallback through THREAD_WFA_FLANK,
 *                         src/commands/genotype.rs:66-80).
 *   trgt_hmm_batch        build_hmm + Hmm::label + calc_purity +
 *                         remove_imperfect_motifs + label_motifs + count_motifs
 *                         + collapse_labels as composed by label_with_hmm,
 *                         src/trgt/workflows/tr.rs:454-492 (src/hmm/, all files).
 *   trgt_locus_batch      analyze_tr for pre-clipped reads,
 *                         src/trgt/workflows/tr.rs:24-109: get_spanning_reads,
 *                         filter_impure_trs (:400-452), Genotyper::Size
 *                         (genotype_size.rs:6-64) or Genotyper::Cluster
 *                         (genotype_cluster.rs:58-152), label_with_hmm.
 *
 * Conventions
 *   - Plain C, caller-owned buffers, no exceptions cross the boundary.  Every
 *     function returns TRGT_OK (0) or a negative TRGT_ERR_*; the message is
 *     available from trgt_hip_last_error().
 *   - "blob" pointers (sequence bytes) and all OUTPUT pointers may be HOST or
 *     DEVICE (HBM) pointers; the library detects which (hipPointerGetAttributes)
 *     and uses device memory in place.  Offset / length / index arrays that
 *     describe the batch are HOST pointers (the planner reads them).
 *   - A ctx is bound to one GPU and one HIP stream and is single-threaded
 *     (mirrors the thread_local aligners of src/commands/genotype.rs:94-103).
 *     Calls are synchronous from the caller's point of view.
 *   - There is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point fails with TRGT_ERR_NO_DEVICE.
 */
#ifndef TRGT_HIP_H
#define TRGT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TRGT_HIP_ABI_VERSION 11

#define TRGT_OK 0
#define TRGT_ERR_INVALID (-1)     /* bad argument */
#define TRGT_ERR_HIP (-2)         /* HIP runtime error */
#define TRGT_ERR_UNSUPPORTED (-3) /* valid request outside the kernels' limits */
#define TRGT_ERR_NO_DEVICE (-4)   /* no GPU / extension built without device code */
#define TRGT_ERR_NOMEM (-5)

/* per-job alignment status == WF_STATUS_* of WFA2-lib (wfaligner.rs:126-159) */
#define TRGT_WF_COMPLETED 0
#define TRGT_WF_PARTIAL 1
#define TRGT_WF_MAX_STEPS (-100)
#define TRGT_WF_OOM (-200)
#define TRGT_WF_UNATTAINABLE (-300)

typedef struct trgt_hip_ctx trgt_hip_ctx;

int trgt_hip_abi_version(void);
/* device: HIP ordinal (>= 0).  There is no CPU device.
 * Process-wide side effect: the first context of a process calls mallopt() -- mmap threshold 32 MB, no heap trimming -- because every
 * munmap (what free() of a large block ends in) delays the next GPU submission of the process by 10-30 ms on this stack (DESIGN.md
 * section 5, INTEGRATION.md "Host allocator").  TRGT_MALLOC_TUNE=0 in the environment leaves malloc alone. */
int trgt_hip_create(int device, trgt_hip_ctx** out);
void trgt_hip_destroy(trgt_hip_ctx* ctx);
const char* trgt_hip_last_error(const trgt_hip_ctx* ctx); /* ctx may be NULL: last create() error */
/* Use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = ctx-owned stream. */
int trgt_hip_set_stream(trgt_hip_ctx* ctx, void* hip_stream);
/* Upper bound (bytes) for the per-call device workspace (wavefront history, back-pointers). 0 = default (32 GiB). */
int trgt_hip_set_workspace_limit(trgt_hip_ctx* ctx, uint64_t bytes);

/* ---- kernel timing (HIP events on the ctx stream, for bench.py's roofline) ---- */
#define TRGT_K_FLANK_SCAN 0   /* exact flank search (+ the segment search for the seeded windows of the fallback alignments) */
#define TRGT_K_WFA 1          /* wavefront alignment kernel: trgt_wfa_batch / consensus alignments */
#define TRGT_K_HMM 2          /* Viterbi + traceback + decode   */
#define TRGT_K_WFA_FLANK 3    /* wavefront alignment kernel: flank fallback inside trgt_find_spans_batch -- the launch over the
                                 reads too short to span their locus (the expensive alignments), or the only launch */
#define TRGT_K_WFA_FLANK_REST 4 /* ... the launch(es) over the remaining fallback alignments */
#define TRGT_K_WFA_FILTER 5   /* register-resident pre-filter of those expensive alignments (penalty + match bound, no back-trace);
                                 TRGT_K_WFA_FLANK then covers only the alignments the filter keeps */
#define TRGT_K_FLANK_WINDOW 6 /* seed search + seeded windows of the fallback alignments (flank_window_kernel).  A slot of its own since ABI 9: its launches
                                 sit on the stream next to the pre-filter's and mostly WAIT for that kernel's persistent workgroups (0.08-0.14 ms of work
                                 bracketed by up to 2.5 ms): event time here is not execution time, and it must not be added to the scan's */
#define TRGT_K_COUNT 7
int trgt_hip_timing_enable(trgt_hip_ctx* ctx, int on);
int trgt_hip_timing_reset(trgt_hip_ctx* ctx);
/* accumulated device time (ms), number of launches, and DP work items (wavefront offsets / Viterbi cells) */
int trgt_hip_timing_get(trgt_hip_ctx* ctx, int kernel, double* ms, int64_t* launches, int64_t* cells);

/* ------------------------------------------------------------------ WFA */
/* Mirrors the attribute set the wrapper configures (wfaligner.rs:161-380). */
typedef struct trgt_wfa_params {
  int32_t metric;      /* 0 indel, 1 edit, 2 gap-linear, 3 gap-affine, 4 gap-affine-2p (DistanceMetric, :79-85) */
  int32_t mismatch;    /* x */
  int32_t gap_open1;   /* o1 */
  int32_t gap_ext1;    /* e1 (gap-linear: the indel penalty) */
  int32_t gap_open2;
  int32_t gap_ext2;
  int32_t span;        /* 0 end-to-end (:489-501), 1 ends-free (:503-528) */
  int32_t pattern_begin_free, pattern_end_free, text_begin_free, text_end_free; /* -1 = that sequence's length */
  int32_t scope;       /* AlignmentScope: 0 Score, 1 Alignment (:52-56) */
  int32_t memory_mode; /* MemoryModel: 0 High, 1 Med, 2 Low, 3 UltraLow = BiWFA (:4-10) */
  int32_t heuristic;   /* 0 Heuristic::None, 1 Heuristic::WFadaptive (:68-77).  CONTRACT for the rest of the enum -- 2 WFmash, 3 XDrop, 4 ZDrop,
                          5 BandedStatic, 6 BandedAdaptive (set_heuristic, :707-780): trgt_wfa_batch returns TRGT_ERR_UNSUPPORTED for the whole
                          batch and touches no output.  No call site of the genotype path uses them (genotype.rs:66-92 builds its three aligners
                          with None / the default WFadaptive), WFA2-lib's sources are not in the reference tree and no reference test pins what
                          they compute (wfaligner.rs:1428-1434 only sets them): an implementation here could not be checked against anything */
  int32_t h_min_wavefront_length, h_max_distance_threshold, h_steps_between_cutoffs;
  int32_t bialign_min_score;  /* WF_BIALIGN_FALLBACK_MIN_SCORE, 250 */
  int32_t bialign_min_length; /* WF_BIALIGN_FALLBACK_MIN_LENGTH, 100 (0 disables) */
} trgt_wfa_params;

/* wavefront_aligner_attr_default: affine(4,6,2), alignment scope, end-to-end, wfadaptive(10,50,1), memory high */
void trgt_wfa_default_params(trgt_wfa_params* p);

/* n_jobs alignments of pattern j = seqs[pat_off[j] .. +pat_len[j]) vs text j.
 * Outputs (each may be NULL):
 *   status[j]    WF status;  score[j] cigar.score (INT32_MIN when failed / never set)
 *   n_match[j]   cigar_count_matches;  span4[4j..] = pattern_start,pattern_end,text_start,text_end
 *   cigar        run-length CIGAR as cigar_get_CIGAR(show_mismatches=true): len<<4 | {7 '=',8 'X',1 'I',2 'D'},
 *                job j at cigar[cigar_off[j] ..], capacity pat_len+txt_len+1 entries; cigar_len[j] entries used
 *   ops          expanded M/X/I/D bytes at ops[ops_off[j] ..], capacity pat_len+txt_len; ops_len[j]
 */
int trgt_wfa_batch(trgt_hip_ctx* ctx, const trgt_wfa_params* p, int64_t n_jobs,
                   const uint8_t* seqs, const uint64_t* pat_off, const uint32_t* pat_len,
                   const uint64_t* txt_off, const uint32_t* txt_len,
                   int32_t* status, int32_t* score, int32_t* n_match, uint32_t* span4,
                   uint32_t* cigar, const uint64_t* cigar_off, uint32_t* cigar_len,
                   uint8_t* ops, const uint64_t* ops_off, uint32_t* ops_len);

/* --------------------------------------------------------- flank location */
typedef struct trgt_span_params {
  int32_t flank_len;         /* Params::search_flank_len (tr.rs:20), 250 */
  double min_flank_id_frac;  /* Params::min_flank_id_frac, 0.7 */
  int32_t mism, gapo, gape;  /* --aln-scoring 2,5,1 (genotype.rs:75-77) */
} trgt_span_params;

/* find_tr_spans for n_loci loci.  Locus l owns reads [locus_read_begin[l], locus_read_begin[l+1]).
 * flank_blob holds, per locus, the left flank then the right flank (any length >= flank_len):
 * left flank of locus l = flank_blob[lf_off[l] .. +lf_len[l]), right = flank_blob[rf_off[l] .. +rf_len[l]).
 * span_start/span_end: per read, -1/-1 = None.  lf_hit/rf_hit (optional, per read): 0 none, 1 exact, 2 WFA. */
int trgt_find_spans_batch(trgt_hip_ctx* ctx, const trgt_span_params* p, int64_t n_loci,
                          const uint8_t* flank_blob, const uint64_t* lf_off, const uint32_t* lf_len,
                          const uint64_t* rf_off, const uint32_t* rf_len,
                          const uint64_t* locus_read_begin,
                          const uint8_t* read_blob, const uint64_t* read_off, const uint32_t* read_len,
                          int32_t* span_start, int32_t* span_end, uint8_t* lf_hit, uint8_t* rf_hit);

/* Pre-filter of the fallback alignments of find_spans (span_locater.rs:14-22), as trgt_find_spans_batch runs it in front of the
 * back-tracing kernel.  Job j = align_ends_free(pattern j, 0, 0, text j, |text|, |text|) with gap-affine (mism, gapo, gape),
 * which must be (2, 5, 1) (--aln-scoring default; anything else: TRGT_ERR_UNSUPPORTED).  Outputs (each may be NULL):
 *   score[j]        the optimal alignment score (-penalty), exactly WFA2-lib's; INT32_MIN if the job was not judged
 *   match_bound[j]  an upper bound on count_matches() of the alignment the reference's back-trace returns (-1: not judged)
 *   keep[j]         1 iff match_bound >= min_matches, or the job was not judged (pattern longer than 254, text shorter than the
 *                   pattern or longer than the kernel's diagonals, sequence bytes 0x01 / 0x02 / 0xFF): only kept jobs need the
 *                   back-trace -- for the others count_matches() < min_matches is certain
 *   offsets_computed  (one value) wavefront offsets computed, equal to WFA2-lib's count for the judged jobs
 * early_reject != 0 (what trgt_find_spans_batch uses): an alignment is given up at the first score level (checked every 16th)
 * at which no cell of its wavefronts can end in min_matches matches any more -- matched so far + pattern bases left < min_matches
 * for all of them, a number no alignment step raises -- so count_matches() < min_matches is certain whatever the optimal
 * alignment turns out to be.  Such a job gets keep 0, score INT32_MIN + 1 (not computed), match_bound min_matches - 1, and
 * offsets_computed counts only the levels that were computed. */
int trgt_flank_filter_batch(trgt_hip_ctx* ctx, const trgt_span_params* p, int64_t n_jobs,
                            const uint8_t* seqs, const uint64_t* pat_off, const uint32_t* pat_len,
                            const uint64_t* txt_off, const uint32_t* txt_len, int32_t min_matches, int32_t early_reject,
                            int32_t* score, int32_t* match_bound, uint8_t* keep, int64_t* offsets_computed);

/* ------------------------------------------------------------------ HMM */
/* n_sets motif sets (one per locus): set s owns motifs [set_motif_begin[s], set_motif_begin[s+1]);
 * motif m = motif_blob[motif_off[m] .. motif_off[m+1]).  Motif bytes outside ATCGN are replaced as
 * replace_invalid_bases(m, ATCGN) does (tr.rs:455-460).
 * Job j labels seq j with the HMM of set job_set[j]; sequence bytes outside ATCG are replaced as
 * replace_invalid_bases(seq, ATCG) does (tr.rs:465).
 * Outputs (path may be NULL):
 *   path        Hmm::label state path (u16), job j at path[path_off[j] ..], path_len[j] entries;
 *               capacity per job >= trgt_hmm_path_capacity(seq_len, longest motif of the set)
 *   spans3      collapsed MS spans (motif_index,start,end) at spans3[3*span_off[j] ..], capacity seq_len[j]+1;
 *               n_spans[j] == 0 means labels = None.  spans3 may be NULL when only purity / counts are wanted
 *               (filter_impure_trs, tr.rs:400-452, needs nothing else)
 *   motif_counts  MC per motif at motif_counts[count_off[j] .. + #motifs of the set]
 *   purity      AP (NaN for an empty allele); edit_dist / max_dist: the integers calc_purity divides
 */
int trgt_hmm_batch(trgt_hip_ctx* ctx, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                   const uint32_t* set_motif_begin,
                   int64_t n_jobs, const uint32_t* job_set,
                   const uint8_t* seq_blob, const uint64_t* seq_off, const uint32_t* seq_len,
                   uint16_t* path, const uint64_t* path_off, uint32_t* path_len,
                   int32_t* spans3, const uint64_t* span_off, uint32_t* n_spans,
                   uint32_t* motif_counts, const uint64_t* count_off,
                   double* purity, int32_t* edit_dist, int32_t* max_dist);
uint64_t trgt_hmm_path_capacity(uint32_t seq_len, uint32_t max_motif_len);
/* Self-check of the model builder (not a reference interface).  The tables of build_hmm / define_motif_block (src/hmm/builder.rs:4-173)
 * are built on the device by every entry point; this call builds them a second time with the host-side builder and returns in
 * *n_diff the number of bytes in which descriptors and tables differ (0 expected). */
int trgt_hmm_models_check(trgt_hip_ctx* ctx, int32_t n_sets, const uint8_t* motif_blob, const uint32_t* motif_off,
                          const uint32_t* set_motif_begin, int64_t* n_diff);

/* ----------------------------------------------------------------- locus */
typedef struct trgt_locus_params {
  int32_t flank_len;         /* 250 */
  double min_flank_id_frac;  /* 0.7 */
  int32_t max_depth;         /* 250 */
  int32_t mism, gapo, gape;  /* 2,5,1 */
  int32_t host_threads;      /* threads for the host glue between GPU stages (0 = hardware concurrency) */
  double min_read_qual;      /* Params::min_read_qual (tr.rs:19, --min-read-quality, default 0.98): below MIN_RQ_FOR_PURITY = 0.9
                                the purity filter filter_impure_trs runs on the spanning reads (tr.rs:37-50) */
} trgt_locus_params;
/* flank_len 250, min_flank_id_frac 0.7, max_depth 250, scoring 2,5,1, host_threads 0, min_read_qual 0.98 (cli.rs:271-344) */
void trgt_locus_default_params(trgt_locus_params* p);

typedef struct trgt_locus_batch_in {   /* Locus (locus.rs:13-23) x n_loci, reads already clipped (tr.rs:33-34) */
  int64_t n_loci;
  const uint8_t* flank_blob;           /* host or device */
  const uint64_t* lf_off; const uint32_t* lf_len;   /* left_flank  */
  const uint64_t* rf_off; const uint32_t* rf_len;   /* right_flank */
  const uint8_t* tr_blob; const uint64_t* tr_off; const uint32_t* tr_len;  /* locus.tr (reference allele), host */
  const uint8_t* motif_blob; const uint32_t* motif_off; const uint32_t* set_motif_begin; /* motifs, host */
  const uint8_t* ploidy;               /* per locus 1 or 2 (0 = skip, tr.rs:29-31) */
  const uint64_t* locus_read_begin;    /* CSR over reads */
  const uint8_t* read_blob;            /* host or device */
  const uint64_t* read_off; const uint32_t* read_len;
  const uint8_t* genotyper;            /* optional, per locus: 0 Genotyper::Size, 1 Genotyper::Cluster (locus.rs:25-29); NULL = all Size */
  const double* read_qual;             /* optional, per read: HiFiRead::read_qual, NaN = None; NULL = None for every read.
                                          Only looked at when min_read_qual < 0.9. */
  /* optional, per read: what genotype_flank::genotype reads (src/trgt/genotype/genotype_flank.rs:9-290).  A locus with two alleles at
   * most 10 bases apart is genotyped again from the haplotype tags or from heterozygous SNVs of the flanks (tr.rs:69-75) when hp_tag or
   * mismatch_offsets is given; with both NULL the step cannot change anything (no tags, no mismatches: it returns None) and is skipped. */
  const int16_t* hp_tag;               /* HiFiRead::hp_tag, -1 = None */
  const int32_t* start_offset;         /* alignment start - region start; NULL = 0 */
  const int32_t* end_offset;           /* alignment end - region end; NULL = 0 */
  const int32_t* mismatch_offsets;     /* HiFiRead::mismatch_offsets (snp.rs:51-79), ascending per read */
  const uint64_t* mismatch_off;        /* [n_reads + 1] into mismatch_offsets */
  /* How read_blob holds the bases.  0: one base per byte (ASCII, what HiFiRead::bases is, reads.rs).  1 (TRGT_READS_BAM4): the BAM
   * record's own 4-bit codes, two bases per byte, first base in the high nibble ("=ACMGRSVTWYHKDBN", SAM spec 4.2.3; rust-htslib
   * hands them out as record.seq().encoded): read_off[r] is the BYTE offset of read r's first pair, every read starts on a byte
   * boundary, read_len[r] stays the number of bases.  The library expands the codes to the same ASCII bytes in HBM (one kernel, 0.5 B
   * read + 1 B written per base) -- half the bytes cross PCIe, the results are identical.  Host or device, like encoding 0. */
  int32_t read_encoding;
} trgt_locus_batch_in;
#define TRGT_READS_ASCII 0
#define TRGT_READS_BAM4 1

typedef struct trgt_locus_batch_out {  /* LocusResult (locus_result.rs:16-22) x n_loci; all HOST, caller-allocated */
  int32_t* span_start; int32_t* span_end;   /* per input read (find_tr_spans), -1 = None */
  int32_t* n_alleles;                       /* per locus: 0 (LocusResult::empty), 1 or 2 */
  uint8_t* allele_blob; const uint64_t* allele_off; /* per locus 2 slots, each of capacity allele_cap[l] */
  const uint32_t* allele_cap; uint32_t* allele_len; /* allele_len[2l+a] */
  int32_t* ci;                              /* [4 per locus]: lo,hi of allele 0; lo,hi of allele 1 */
  int32_t* num_spanning;                    /* [2 per locus] */
  int32_t* classification;                  /* per input read: allele index of each kept spanning read, -1 otherwise */
  int32_t* read_rank;                       /* per input read: position in LocusResult.reads order, -1 otherwise */
  /* annotations (Annotation, spans.rs:20-25) */
  int32_t* spans3; const uint64_t* span_off; uint32_t* n_spans;   /* span_off[2l+a], capacity allele_cap[l]+1 */
  uint32_t* motif_counts; const uint64_t* count_off;              /* count_off[2l+a] */
  double* purity;                           /* [2 per locus] */
  int64_t* stats;                           /* optional [24]: see DESIGN.md */
  /* optional (NULL = not wanted): what get_meth / assign_read (tr.rs:196-262) need beyond the fields above */
  int32_t* gt_size;                         /* [2 per locus] TrSize::size of the genotype, output order (the allele length unless it was repaired) */
  uint8_t* flipped;                         /* per locus: 1 = "reference allele first" swapped the two alleles (tr.rs:95-101) */
} trgt_locus_batch_out;

int trgt_locus_batch(trgt_hip_ctx* ctx, const trgt_locus_params* p, const trgt_locus_batch_in* in,
                     trgt_locus_batch_out* out);

/* Host helper for callers that hold ASCII reads and want TRGT_READS_BAM4 (a BAM reader has the codes already: trgt_ingest_params.
 * keep_bam4): packs read r (ascii + read_off[r], read_len[r] bases) to packed + packed_off[r], where packed_off[r] is written as the
 * running sum of ceil(read_len / 2).  Letters outside "=ACMGRSVTWYHKDBN" become N, as in htslib.  packed needs
 * sum(ceil(read_len[r] / 2)) bytes; returns that sum, or a negative TRGT_ERR_* code. */
int64_t trgt_reads_pack_bam4(const uint8_t* ascii, int64_t n_reads, const uint64_t* read_off, const uint32_t* read_len,
                             uint8_t* packed, uint64_t* packed_off);

/* Pipelined form (the consumer side of the reference's per-locus channel, src/commands/genotype.rs:140-187): the upload of batch
 * k + 1 runs next to the kernels of batch k.
 *   trgt_locus_batch_submit  starts copying the read and flank bytes of the batch to HBM on the context's copy stream (one of two
 *                            staging sets) and returns at once with a ticket;
 *   trgt_locus_batch_wait    analyses that batch exactly as trgt_locus_batch does and returns when `out` is complete.
 * Tickets are waited for in submission order; at most two may be outstanding.  `in`, `out` and everything they point to must stay
 * valid and unchanged until the wait returns.  The copy is asynchronous when the blobs are in pinned host memory (hipHostMalloc /
 * hipHostRegister); from pageable memory it still works, the submit then blocks for the copy.  Blobs already in HBM are used in
 * place.  Usage: submit(b0); for k: { submit(b[k+1]); wait(b[k]); }  */
int trgt_locus_batch_submit(trgt_hip_ctx* ctx, const trgt_locus_params* p, const trgt_locus_batch_in* in,
                            trgt_locus_batch_out* out, int64_t* ticket);
int trgt_locus_batch_wait(trgt_hip_ctx* ctx, int64_t ticket);

/* ---- several contexts, one queue of batches (the consumer side of the reference's locus channel, src/commands/genotype.rs:140-187, one level
 * up: a context is single-threaded and a call is synchronous, so K contexts keep K batches in flight -- the tail of one call overlaps the
 * flank location of the next ones).  devices[i] = HIP ordinal of context i; ordinals may repeat (several contexts per GPU: 4 measure
 * 1.3x one context on the 10k-locus STR batch) and differ (one process driving several GPUs).  trgt_locus_batch_many hands the batches
 * to one worker thread per context through a shared counter (dynamic: a batch of long alleles does not hold the others up) and returns
 * when all are done; the first error of any batch is returned (message: trgt_hip_pool_last_error), batches not yet started are
 * skipped then.  out[i] belongs to batch i -- or, with out_per_context != 0, out has one entry per CONTEXT and every batch a context
 * runs overwrites that context's entry (throughput measurements).  ran_on (optional, per batch): the context that took it. */
typedef struct trgt_hip_pool trgt_hip_pool;
int trgt_hip_pool_create(const int32_t* devices, int32_t n_contexts, trgt_hip_pool** out);
void trgt_hip_pool_destroy(trgt_hip_pool* pool);
int32_t trgt_hip_pool_size(const trgt_hip_pool* pool);
trgt_hip_ctx* trgt_hip_pool_context(trgt_hip_pool* pool, int32_t i);   /* e.g. for trgt_hip_timing_* */
const char* trgt_hip_pool_last_error(const trgt_hip_pool* pool);
int trgt_locus_batch_many(trgt_hip_pool* pool, const trgt_locus_params* p, int64_t n_batches, const trgt_locus_batch_in* const* in,
                          trgt_locus_batch_out* const* out, int32_t out_per_context, int32_t* ran_on);

/* ------------------------------------------------------------ read ingestion (the step in front of trgt_locus_batch)
 * Repeat catalog + indexed FASTA + indexed BAM -> the arrays of trgt_locus_batch_in, i.e. what analyze_tr does before get_spanning_reads:
 *   Locus::new / get_tr_and_flanks (src/trgt/locus.rs:31-98, 168-190), extract_reads (src/trgt/workflows/tr.rs:268-361: region
 *   +- flank_len, secondary / supplementary records dropped, rq < min_read_qual dropped and counted, at most 3 * max_depth reads kept
 *   -- beyond that a reservoir driven by StdRng::seed_from_u64(42)), HiFiRead::from_hts_rec (src/trgt/reads/read.rs:98-141: bases, base
 *   qualities, rq / HP tags, 5mC calls of the MM / ML tags per CpG, mismatch offsets of snp.rs:51-79) and clip_reads (tr.rs:186-196 ->
 *   clip_region.rs:19-184, radius 2 * flank_len).  Host code (zlib + the .bai / .fai indexes), loci read by a pool of threads -- or, with
 *   trgt_ingest_params.ingest_device, kernels behind the device inflate (the inflated BAM bytes never leave HBM). */
typedef struct trgt_ingest trgt_ingest;
typedef struct trgt_ingest_params {
  int32_t flank_len;       /* 250  --flank-len: flanks of the Locus and the search flank of extract_reads */
  int32_t max_depth;       /* 250  reads kept per locus: 3 * max_depth */
  double min_read_qual;    /* 0.98 */
  int32_t threads;         /* 0 = min(32, cores); a worker takes runs of consecutive catalog lines and keeps the last BGZF blocks it inflated */
  int32_t genotyper;       /* 0 size, 1 cluster: copied into trgt_ingest_batch::genotyper for every locus */
  int32_t default_ploidy;  /* 2 (the karyotype logic of locus.rs:216-240 stays with the caller: overwrite ploidy[] for X / Y loci) */
  int32_t keep_bam4;       /* 0; 1 = also fill read_bam4 / read_bam4_off: the clipped reads as 4-bit codes for TRGT_READS_BAM4 */
  int32_t ingest_device;   /* ABI 10 (replaces ABI 8's inflate_device, whose inflated bytes went back to the host workers): -1 (default) = the host
                              path, worker threads with zlib.  >= 0 = GPU ordinal: the BGZF blocks the .bai names for the loci of the call are read
                              into pinned memory, inflated on that GPU (trgt_inflate_blocks' kernel) with their CRC-32 / ISIZE checked, and the record
                              walk of extract_reads, the filters, HiFiRead::from_hts_rec (rq / HP / MM / ML tags, mismatch offsets) and clip_to_region
                              run as kernels on the inflated bytes in HBM (trgt_amd/csrc/ingest_dev.hip): only the clipped reads come back, and the
                              ASCII read blob stays in HBM as well (trgt_ingest_batch::read_blob_dev) for trgt_locus_batch.  Same arrays, bit for
                              bit, as the host path.  A call the kernels do not take -- a block that fails its CRC or does not inflate, a record that
                              leaves its range, MM strings beyond the kernel's caps (a read longer than 65 535 bases, more than 4 096 calls in one MM
                              entry) -- is redone as a whole by the host path, which yields the data or the error
                              (trgt_ingest_device_stats counts them).  A device that cannot be used fails the call: no silent host-only run.
                              Calls from several host threads on one reader overlap (six slots of device state) */
  int32_t inflate_waves_per_cu; /* ABI 10, with ingest_device: BGZF blocks in flight per CU of the inflate kernel (one wave each, 10 KB of LDS; its waves
                              live as long as the launch).  0 = 12.  A block takes its wave the same ~ 3.8 ms whatever the occupancy, so a launch
                              lasts ceil(blocks / waves) block times: for the ~ 4 200 blocks of a 1 000-locus chunk anything from 9 per CU on is
                              two rounds; 12 leave LDS and registers for the kernels of trgt_locus_batch on the same GPU to start next to it
                              instead of behind it; 16 = the whole CU for a GPU that only ingests (values above 16 are taken as 16) */
} trgt_ingest_params;
typedef struct trgt_ingest_batch {  /* everything host memory owned by the batch; free with trgt_ingest_free */
  int64_t n_loci, n_reads, n_motifs;
  uint64_t flank_bytes, tr_bytes, motif_bytes, read_bytes;
  /* -- the fields of trgt_locus_batch_in, same names and meaning */
  const uint8_t* flank_blob; const uint64_t* lf_off; const uint32_t* lf_len; const uint64_t* rf_off; const uint32_t* rf_len;
  const uint8_t* tr_blob; const uint64_t* tr_off; const uint32_t* tr_len;
  const uint8_t* motif_blob; const uint32_t* motif_off; const uint32_t* set_motif_begin;
  const uint8_t* ploidy; const uint8_t* genotyper; const uint64_t* locus_read_begin;
  const uint8_t* read_blob; const uint64_t* read_off; const uint32_t* read_len; const double* read_qual;  /* NaN = no rq tag */
  /* -- per locus: catalog fields and counters */
  const char* contig_blob; const uint64_t* contig_off; const char* id_blob; const uint64_t* id_off;      /* [n_loci + 1] offsets */
  const char* struc_blob; const uint64_t* struc_off; const int64_t* region_start; const int64_t* region_end;
  const int32_t* n_quality_filtered; const int64_t* n_reads_seen;   /* dropped for rq; reads that passed the filters (> kept: reservoir) */
  /* -- per clipped read: the rest of HiFiRead, for the writers and genotype_flank */
  const uint8_t* qual_blob;                       /* base qualities, same offsets as read_blob */
  const char* name_blob; const uint64_t* name_off;  /* [n_reads + 1] */
  const uint8_t* is_reverse; const uint8_t* mapq; const int16_t* hp_tag;   /* hp_tag -1 = None */
  const int32_t* start_offset; const int32_t* end_offset;                  /* alignment start - region start, alignment end - region end */
  const int32_t* mismatch_offsets; const uint64_t* mismatch_off;           /* [n_reads + 1] */
  const uint8_t* meth; const uint64_t* meth_off; const uint8_t* has_meth;  /* [n_reads + 1]; has_meth 0 = None */
  const uint32_t* cigar; const uint64_t* cigar_off; const int64_t* cigar_ref_pos;  /* clipped CIGAR (len << 4 | op), [n_reads + 1] */
  /* -- with trgt_ingest_params.keep_bam4: the reads once more as BAM 4-bit codes (trgt_locus_batch_in: read_blob = read_bam4, read_off =
   *    read_bam4_off, read_encoding = TRGT_READS_BAM4; read_len is shared).  NULL otherwise. */
  const uint8_t* read_bam4; const uint64_t* read_bam4_off; uint64_t read_bam4_bytes;
  /* -- catalog lines that gave no locus: one "Error at BED line N: ..." message each (locus.rs:93-137 reports them and goes on) */
  int64_t n_skipped; const char* skipped_blob; const uint64_t* skipped_off;   /* [n_skipped + 1] offsets into skipped_blob */
  /* -- ABI 10, with trgt_ingest_params.ingest_device: the ASCII reads once more in HBM of GPU read_blob_device (same bytes and offsets as
   *    read_blob; owned by the batch): trgt_locus_batch_in::read_blob may point here, nothing is uploaded then.  NULL / -1 otherwise. */
  const uint8_t* read_blob_dev; int32_t read_blob_device;
  void* owner;
} trgt_ingest_batch;
void trgt_ingest_default_params(trgt_ingest_params* p);
int trgt_ingest_open(const char* bam_path, const char* fasta_path, trgt_ingest** out);  /* needs <bam>.bai and <fasta>.fai */
void trgt_ingest_close(trgt_ingest* h);
const char* trgt_ingest_last_error(const trgt_ingest* h);
/* catalog LINES [first_locus, first_locus + max_loci) (max_loci < 0: to the end); a line that gives no Locus -- wrong field count
 * (blank lines too), bad coordinates, flanks leaving the contig, unknown contig -- is skipped and reported in skipped_blob, as
 * stream_loci_into_channel does (src/trgt/locus.rs:93-137) */
int trgt_ingest_batch_from_catalog(trgt_ingest* h, const trgt_ingest_params* p, const char* bed_path, int64_t first_locus,
                                   int64_t max_loci, trgt_ingest_batch** out);
void trgt_ingest_free(trgt_ingest_batch* b);   /* batches of a reader are freed before trgt_ingest_close when they hold device memory */
/* ABI 10: what ingest_device did so far: out[0] calls that asked for the device, [1] of those, calls redone by the host path, [2] the reason of
 * the last one (1 a BGZF block, 2 the record walk, 4 MM / ML caps), [3] BGZF blocks inflated for the device path, [4] of
 * those, blocks its kernel declined (zlib took them) */
void trgt_ingest_device_stats(const trgt_ingest* h, int64_t out[5]);

/* the header of the BAM behind a reader: its text and its reference sequences (what the writers take over) */
const char* trgt_ingest_header_text(const trgt_ingest* h);
int32_t trgt_ingest_n_contigs(const trgt_ingest* h);
const char* trgt_ingest_contig_name(const trgt_ingest* h, int32_t i);
uint32_t trgt_ingest_contig_length(const trgt_ingest* h, int32_t i);

/* ------------------------------------------------------------ writers (the step behind trgt_locus_batch)
 * VcfWriter (src/trgt/writers/write_vcf.rs:19-397): header lines, one record per locus -- REF / ALT with the padding base, GT by
 * set_gt, AL / ALLR / SD / MC / MS / AP and AM (get_meth / assign_read / get_tr_meth, src/trgt/workflows/tr.rs:196-262, 363-398) -- as text,
 * BGZF-compressed when the path ends in ".gz".  BamWriter (src/trgt/writers/write_bam.rs:33-144): the spanning reads of every locus clipped
 * to output_flank_len bases around the repeat (HiFiRead::clip_bases, src/trgt/reads/clip_bases.rs:9-120) with the tags TR, rq, MC, MO, HP, SO,
 * EO, AL, FL, in a BAM whose header is the input's plus a @PG record.  bam_path NULL: no BAM. */
typedef struct trgt_writer trgt_writer;
typedef struct trgt_writer_params {
  int32_t output_flank_len;   /* 50   min(--flank-len, --output-flank-len) (genotype.rs:129) */
  const char* sample_name;    /* VCF sample column */
  const char* program;        /* "trgt": ##<program>Version= / ##<program>Command= / @PG ID, PN */
  const char* version;
  const char* command_line;
  int32_t keep_unmapped_flag; /* 1 (default since ABI 7): flag 0x4 stays set on every record (flags 4 / 20) -- what write_bam.rs:96-111 produces with
                                 rust-htslib 0.46 (Cargo.lock), whose Record::new() initialises a record as unmapped (set_unmapped(), tid / pos /
                                 mtid / mpos = -1: the mate fields this writer mirrors already) and whose set() / set_pos() / set_mapq() / set_reverse()
                                 do not clear it; write_bam.rs never calls unset_unmapped().  0: aligned reads are written with flag 0 / 0x10.
                                 rust-htslib is un-vendored and no reference-produced BAM is on disk: parity of this bit is UNPINNED, hence the switch */
  int32_t threads;            /* 0 = min(32, cores): workers that format the loci of a batch (contiguous ranges, written in locus order) and
                                 deflate its BGZF blocks; the files do not depend on it */
  int32_t bam_compress_level; /* ABI 8: 6 (default: htslib's level for BAM) ... 1 (fast), 0 = stored DEFLATE blocks: the same records in a
                                 larger file, for pipelines whose spanning BAM is transient; the VCF (.gz) always uses 6 */
  int32_t deflate_device;     /* ABI 9: -1 (default) = the BGZF blocks of the spanning BAM are deflated by zlib on host threads (what htslib does
                                 for the reference, write_bam.rs:72-144); >= 0 = GPU ordinal: the full blocks of a batch are deflated on that GPU in
                                 one go (trgt_deflate_blocks: fixed Huffman codes, ratio about zlib's level 1), a block the device declines by zlib.
                                 The records are the same; the compressed bytes are not zlib's */
  int32_t write_behind;       /* ABI 11: 0 (default) = trgt_writer_write returns when the batch's bytes have been handed to the files.  1 = it
                                 returns once the batch is FORMATTED (nothing of the batch or of the results is referenced afterwards); a thread
                                 of the writer's own appends, deflates and writes the pieces while the caller goes on -- one batch in flight,
                                 the files are the same byte for byte.  An error of that part (a file that cannot be written, a device deflate
                                 that fails) is returned by the NEXT trgt_writer_write or by trgt_writer_close */
} trgt_writer_params;
void trgt_writer_default_params(trgt_writer_params* p);
int trgt_writer_open(const trgt_ingest* src, const trgt_writer_params* p, const char* vcf_path, const char* bam_path, trgt_writer** out);
/* the loci of one ingested batch with the results trgt_locus_batch filled for it (gt_size / flipped wanted for an exact AM) */
int trgt_writer_write(trgt_writer* w, const trgt_ingest_batch* b, const trgt_locus_batch_out* out);
int trgt_writer_close(trgt_writer* w);   /* flushes, writes the BGZF end-of-file blocks, frees the handle */
const char* trgt_writer_last_error(const trgt_writer* w);
/* ABI 10: BGZF blocks of the spanning BAM so far: out[0] deflated on the device (deflate_device), [1] declined by it (zlib took them), [2] by
 * zlib because no device was named or a flush held fewer than 16 full blocks.  A device deflate that FAILS fails trgt_writer_write. */
void trgt_writer_device_stats(const trgt_writer* w, int64_t out[3]);

/* ------------------------------------------------- per-read helpers of the ingestion / writer steps, exported on their own
 * (host code; the functions trgt_ingest_* and trgt_writer_* use internally -- a host that keeps its own BAM reader can call them, and the
 * reference's unit tests for them run through these entry points: tests/test_read_helper_kats.py).  CIGAR operations are BAM words
 * (len << 4 | op, op: 0 M, 1 I, 2 D, 3 N, 4 S, 5 H, 6 P, 7 =, 8 X).  Every function returns the number of elements it produced, -1 for
 * "None" in the reference's sense, or a TRGT_ERR_* code (<= -2 ... see the values above) when an output capacity is too small. */
/* CigarOpExt::get_ref_len / get_query_len (src/trgt/reads/cigar.rs:8-31) and Cigar::query_len (:41-45) */
int64_t trgt_cigar_ref_len(uint32_t op);
int64_t trgt_cigar_query_len(uint32_t op);
int64_t trgt_cigar_total_query_len(const uint32_t* cigar, int64_t n_ops);
/* extract_snps_offset (src/trgt/reads/snp.rs:51-79): offsets of the X runs outside [region_start, region_end] */
int64_t trgt_read_mismatch_offsets(const uint32_t* cigar, int64_t n_ops, int64_t ref_pos, int64_t region_start, int64_t region_end,
                                   int32_t* out, int64_t cap);
/* get_meth (src/trgt/reads/read.rs:55-96) from the MM text and ML bytes of a record: one value per CpG of `bases`; -1 = None */
int64_t trgt_read_meth(const uint8_t* bases, int64_t n_bases, const char* mm, const uint8_t* ml, int64_t n_ml, int32_t is_reverse,
                       uint8_t* out, int64_t cap);
/* HiFiRead::clip_to_region (src/trgt/reads/clip_region.rs:19-184): bases / quals / per-CpG meth (n_meth < 0: None) / CIGAR of the part of
 * the read aligned inside [region_start, region_end).  Returns the clipped number of bases, -1 = None (no overlap); out_meth_n gets
 * the number of meth values kept (-1 = None), out_ref_pos / out_cigar / out_n_ops the clipped alignment.  Output capacities: n_bases
 * bases and quals, n_meth meth values, n_ops + 1 operations. */
int64_t trgt_read_clip_to_region(const uint8_t* bases, const uint8_t* quals, int64_t n_bases, const uint8_t* meth, int64_t n_meth,
                                 const uint32_t* cigar, int64_t n_ops, int64_t ref_pos, int64_t region_start, int64_t region_end,
                                 uint8_t* out_bases, uint8_t* out_quals, uint8_t* out_meth, int64_t* out_meth_n,
                                 uint32_t* out_cigar, int64_t* out_n_ops, int64_t* out_ref_pos);
/* HiFiRead::clip_bases (src/trgt/reads/clip_bases.rs:9-120): the read without its first left_len and last right_len bases; same outputs */
int64_t trgt_read_clip_bases(const uint8_t* bases, const uint8_t* quals, int64_t n_bases, const uint8_t* meth, int64_t n_meth,
                             const uint32_t* cigar, int64_t n_ops, int64_t ref_pos, int64_t left_len, int64_t right_len,
                             uint8_t* out_bases, uint8_t* out_quals, uint8_t* out_meth, int64_t* out_meth_n,
                             uint32_t* out_cigar, int64_t* out_n_ops, int64_t* out_ref_pos);
/* utils::math::median (src/utils/math.rs:73-98): f32 median of i32 values as simple_consensus uses it (genotype_flank.rs:147); 0 = None */
int32_t trgt_median_i32(const int32_t* data, int64_t n, float* out);

/* One raw DEFLATE stream (a BGZF block's payload) into exactly n_out bytes: mode 0 = the library's decoder (trgt_amd/csrc/inflate_fast.hpp;
 * 1 = done, 0 = declined -- ingestion then lets zlib decide), mode 1 = zlib.  Exported for tests/test_inflate.py; htslib's bgzf_read_block
 * is what it stands in for (the reference reads BAM through rust-htslib). */
int32_t trgt_inflate_raw(const uint8_t* in, int64_t n_in, uint8_t* out, int64_t n_out, int32_t mode);

/* ABI 8 -- device-side BGZF inflate (trgt_amd/csrc/inflate_dev.hip): n_blocks independent raw DEFLATE streams (the payloads of BGZF
 * blocks: what htslib's bgzf_read_block inflates one by one for bam::IndexedReader, src/trgt/workflows/tr.rs:268-305), block b from
 * src + src_off[b] (src_len[b] bytes) into dst + dst_off[b] (exactly dst_len[b] <= 65536 bytes); src / dst host or device memory.
 * status[b] = 1: inflated; 0: declined (a stream the device decoder does not take, or a damaged one: hand it to zlib, which yields
 * the data or the error).  One wave per block, Huffman tables and the LZ77 window in LDS.  Synchronous.  trgt_ingest_params.
 * inflate_device makes the ingestion use it for the blocks of a whole batch of loci at once. */
int trgt_inflate_blocks(trgt_hip_ctx* ctx, int64_t n_blocks, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_len,
                        uint8_t* dst, const uint64_t* dst_off, const uint32_t* dst_len, uint8_t* status);

/* ABI 9 -- device-side BGZF deflate (trgt_amd/csrc/deflate_dev.hip): n_blocks independent raw DEFLATE streams (the payloads of BGZF
 * blocks: what htslib's bgzf_write / bgzf_flush deflate one by one for bam::Writer, src/trgt/writers/write_bam.rs:72-144), block b
 * from src + src_off[b] (src_len[b] <= 65536 bytes) into dst + dst_off[b]; dst_off[b] a multiple of 4, the regions at least
 * dst_cap[b] + 8 bytes apart (the encoder clears and ORs whole words).  dst_len[b] = bytes written (<= dst_cap[b]: ONE final block with
 * fixed Huffman codes, readable by any inflate), or 0: declined (the encoding does not fit dst_cap[b] -- data that does not compress):
 * deflate that block with zlib.  src / dst / dst_len host memory.  One wave per block, one slice of the block per lane.  Synchronous.
 * trgt_writer_params.deflate_device makes the writer use it for the spanning BAM. */
int trgt_deflate_blocks(trgt_hip_ctx* ctx, int64_t n_blocks, const uint8_t* src, const uint64_t* src_off, const uint32_t* src_len,
                        uint8_t* dst, const uint64_t* dst_off, const uint32_t* dst_cap, uint32_t* dst_len);

/* ------------------------------------------------- synthetic workload */
typedef struct trgt_synth_params {
  uint64_t seed;           /* 20250509 */
  int32_t config;          /* 2 = single-motif STR loci (cfg2), 4 = genome-wide catalog stand-in (70 % STR, 20 % 2-5 motifs,
                              10 % VNTR motifs of 7-60 bp), 5 = compound / N-motif loci for the cluster genotyper */
  int32_t reads_per_locus; /* 30 */
  int32_t context_len;     /* 500 */
  int32_t flank_len;       /* 250 */
  int32_t max_allele_bp;   /* 200 (cfg2) / 300 (cfg5); larger values give cfg3-like long alleles */
  double sub_rate, del_rate, ins_rate, stutter_rate, truncate_rate; /* 5e-4, 2.5e-4, 2.5e-4, 0.05, 0.10 */
} trgt_synth_params;
void trgt_synth_default_params(trgt_synth_params* p, int config);
/* Library-owned synthetic batch in exactly the trgt_locus_batch_in layout (host memory).  Locus indices are
 * global (first_locus .. first_locus+n_loci): the per-locus RNG is splitmix64 seeded with
 * seed ^ (locus_index+1)*0x9E3779B97F4A7C15, so any shard regenerates exactly its own loci. */
typedef struct trgt_synth_batch {
  int64_t n_loci, n_reads, n_motifs;
  uint64_t flank_bytes, tr_bytes, motif_bytes, read_bytes;
  uint8_t* flank_blob; uint64_t* lf_off; uint32_t* lf_len; uint64_t* rf_off; uint32_t* rf_len;
  uint8_t* tr_blob; uint64_t* tr_off; uint32_t* tr_len;
  uint8_t* motif_blob; uint32_t* motif_off; uint32_t* set_motif_begin; uint8_t* ploidy;
  uint64_t* locus_read_begin; uint8_t* read_blob; uint64_t* read_off; uint32_t* read_len;
  uint32_t* true_allele_len;   /* [2 * n_loci] ground truth (bp) */
  uint8_t* read_hap;           /* per read: haplotype 0/1 */
  uint8_t* read_truncated;     /* per read: 1 if cut (must end up with span None) */
  uint8_t* genotyper;          /* per locus: 0 size, 1 cluster (cfg5) */
} trgt_synth_batch;
int trgt_synth_generate(const trgt_synth_params* p, int64_t first_locus, int64_t n_loci, int threads, trgt_synth_batch** out);
void trgt_synth_free(trgt_synth_batch* b);

#ifdef __cplusplus
}
#endif
#endif
