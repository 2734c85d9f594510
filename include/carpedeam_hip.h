/*
 * carpedeam_hip.h -- C ABI of the MI355X (gfx950) implementation of CarpeDeam's hot path
 *
 *     kmermatcher -> rescorediagonal -> ancient_correction -> ancient_read_assemble
 *
 * This is the drop-in boundary.  The reference has no FFI: its "plugin" surface is the
 * module function  int fn(int argc, const char **argv, const Command&)  (lib/mmseqs/src/commons/Command.h:92-103)
 * whose body is an OpenMP loop over a DBReader.  Each entry point below replaces one such loop; the reference
 * lines it replaces are cited per function.  INTEGRATION.md shows the few lines a maintainer adds to the
 * reference's module bodies to call them.
 *
 * Conventions
 *   - plain C, POD structs, opaque handles; no exceptions cross the boundary.
 *   - every function returns 0 on success, a negative cdm_status otherwise; cdm_last_error() gives the text
 *     (thread local).  There is NO CPU fallback: without a usable gfx950 device cdm_ctx_create fails.
 *   - handles own device memory (HBM); *_download copies into caller-allocated host buffers.
 *   - sequences are addressed by *index* (position in the key-sorted DB index, i.e. DBReader's local id,
 *     lib/mmseqs/src/commons/DBReader.cpp:238-) on the device; keys travel alongside for the host codecs.
 *   - a context is bound to one device and one HIP stream; one context per host thread.
 */
#ifndef CARPEDEAM_HIP_H
#define CARPEDEAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cdm_status {
    CDM_OK = 0,
    CDM_ERR_NO_DEVICE = -1,   /* no gfx950 device / HIP runtime failure at create */
    CDM_ERR_HIP = -2,         /* HIP API or kernel launch failure */
    CDM_ERR_INVALID = -3,     /* invalid argument / inconsistent input */
    CDM_ERR_UNSUPPORTED = -4, /* input outside what the device path implements (see cdm_last_error) */
    CDM_ERR_IO = -5           /* damage profile could not be read / parsed */
} cdm_status;

typedef struct cdm_ctx cdm_ctx;
typedef struct cdm_seqdb cdm_seqdb; /* 2-bit packed sequence DB resident in HBM */
typedef struct cdm_hits cdm_hits;   /* prefilter hits (kmermatcher output), CSR by query */
typedef struct cdm_alns cdm_alns;   /* ungapped alignments (rescorediagonal output), CSR by query */

const char *cdm_last_error(void);
int cdm_ctx_create(int device_ordinal, cdm_ctx **out);
void cdm_ctx_destroy(cdm_ctx *ctx);
/* blocks until all work queued on the context's stream is done */
int cdm_ctx_sync(cdm_ctx *ctx);
/* the hipStream_t the context launches on (as void*), for callers that time with HIP events */
void *cdm_ctx_stream(cdm_ctx *ctx);
/* device time in ms of the dominant kernel(s) of the last stage call, measured with HIP events on the context stream;
 * which = 0: ancient_correction pile-up/call kernel, 1: rescore kernel, 2: kmermatcher sorts, 3: kmer extraction,
 * 4: extension kernel, 5: kmermatcher sort 1 on the k-mer slots (the library's own onesweep radix passes, csrc/radix.h: histogram +
 * pass launches), 6: sort 2 (run records, their radix sort, aggregation / unit sorters), 7: sort 1 on the whole-sequence hash tuples;
 * 8..11: the WHOLE stage call (everything it launched, host round trips between kernels included) of kmermatcher, rescorediagonal,
 * ancient_correction, ancient_read_assemble; 13: the radix PASS launches of sort 1 on the k-mer slots alone, summed, 14: how many
 * launches that sum covers, 15: the bytes those launches move at the least, in GB - every pair or tuple they sort read once and
 * written once per launch (bench.py's roofline figure).
 * Returns a negative value when that stage has not run. */
float cdm_ctx_last_kernel_ms(cdm_ctx *ctx, int which);

/* ---------------------------------------------------------------------------------------------------------
 * Sequence DB.  Replaces DBReader<unsigned int>::getData/getSeqLen/getDbKey/getExtData random access
 * (lib/mmseqs/src/commons/DBReader.h:193-213, DBReader.cpp:1001-1012) inside the four stage loops.
 *
 * data      : the DB data file as mapped (entries "SEQUENCE\n\0")
 * offsets[i]: byte offset of entry i in data;  lengths[i]: sequence length WITHOUT "\n\0"
 * keys[i]   : DB key of entry i; entries must be ordered by key (the order DBReader's index has after open()).
 * ext[i]    : the fork's 4th index column "wasExtended" (DBReader.cpp:808-817); may be NULL (= all 0)
 * 'N' is kept as an exception bit beside the 2-bit codes.  Any other byte (lower case, IUPAC codes, ...) is mapped as
 * NucleotideMatrix::setupLetterMapping does (lib/mmseqs/src/commons/NucleotideMatrix.cpp:17-61) for kmermatcher, the
 * diagonal score and reverse complements, and the sequence keeps its original bytes in a side plane for the consumers
 * that look at them (nucleotideMap[c] of the assembler modules, letter identity, the letters copied to the output);
 * cdm_seqdb_download gives them back; the multi-GPU exchange carries them in a section of its own (cdm_seqdb_copy_raw).
 */
int cdm_seqdb_upload(cdm_ctx *ctx, const char *data, const uint64_t *offsets, const uint32_t *lengths,
                     const uint32_t *keys, const uint8_t *ext, uint64_t n, cdm_seqdb **out);
/* Synthetic reads generated on the device with the counter-based generator specified in carpedeam_amd/synth.py
 * (SURVEY.md 8(d)): n reads, fixed length len_lo == len_hi or uniform in [len_lo, len_hi], 20x coverage genome,
 * dhigh damage.  Keys are 0..n-1, ext = 0.  first_read lets a rank generate its own shard [first_read, first_read+n)
 * of a larger corpus (the genome is that of the whole corpus of n_total reads). */
int cdm_seqdb_synth(cdm_ctx *ctx, uint64_t n_total, uint64_t first_read, uint64_t n, uint32_t len_lo, uint32_t len_hi,
                    uint64_t seed, cdm_seqdb **out);
uint64_t cdm_seqdb_size(const cdm_seqdb *db);
uint64_t cdm_seqdb_residues(const cdm_seqdb *db); /* sum of sequence lengths (DBReader::getAminoAcidDBSize) */
uint32_t cdm_seqdb_max_len(const cdm_seqdb *db);
/* lengths/keys/ext: caller arrays of size n (any may be NULL) */
int cdm_seqdb_meta(cdm_ctx *ctx, const cdm_seqdb *db, uint32_t *lengths, uint32_t *keys, uint8_t *ext);
/* ASCII download: out must hold sum(len[i] + 1) bytes; entry i is written at out_offsets[i] followed by '\n' */
int cdm_seqdb_download(cdm_ctx *ctx, const cdm_seqdb *db, char *out, const uint64_t *out_offsets);
/* the same blob in consecutive pieces of (about) piece_bytes through pinned staging buffers of the library: sink(user, data, offset,
 * bytes) gets piece i while piece i + 1 is on its way (data is valid during the call; non-zero return: the download ends with an error).
 * out_offsets must ascend.  What a module process writes its sequence DB with (csrc/host/main.cpp). */
int cdm_seqdb_download_stream(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *out_offsets, uint64_t piece_bytes,
                              int (*sink)(void *user, const char *data, uint64_t offset, uint64_t bytes), void *user);
void cdm_seqdb_free(cdm_seqdb *db);
/* Multi-GPU hand-off (one process per GPU, RCCL all-gather of per-shard contigs): the sequences with wasExtended == 1
 * (the contigs an ancient_read_assemble pass produced) as a new DB; its packed form copied into caller-provided DEVICE
 * buffers (e.g. torch tensors that RCCL then all-gathers); and a DB rebuilt from such packed device buffers.
 * Packed form: codes = 16 bases per uint32 (A,C,G,T = 0..3), every sequence starting on a word boundary, nmask = one
 * uint16 per code word (bit j = base j is 'N'), lengths and keys one uint32 per sequence. */
int cdm_seqdb_select_ext(cdm_ctx *ctx, const cdm_seqdb *db, cdm_seqdb **out);
uint64_t cdm_seqdb_words(const cdm_seqdb *db);
int cdm_seqdb_copy_packed(cdm_ctx *ctx, const cdm_seqdb *db, void *dev_codes, void *dev_nmask16, void *dev_lengths, void *dev_keys);
int cdm_seqdb_from_packed(cdm_ctx *ctx, const void *dev_codes, const void *dev_nmask16, const void *dev_lengths, const void *dev_keys,
                          uint64_t n, uint64_t words, uint8_t ext_value, cdm_seqdb **out);
/* A DB with letters beyond ACGTN (cdm_seqdb_has_raw != 0) has more to hand over than the packed form holds: the original bytes,
 * 16 per code word (dev_raw: words * 16 bytes; rows of sequences without such letters are undefined), and one byte per sequence
 * saying whether its row counts (dev_flags: n bytes).  cdm_seqdb_attach_raw gives them to a DB rebuilt by cdm_seqdb_from_packed*;
 * an exchange that leaves them out would silently turn 'acgt' into what NucleotideMatrix maps it to. */
int cdm_seqdb_has_raw(const cdm_seqdb *db);
int cdm_seqdb_copy_raw(cdm_ctx *ctx, const cdm_seqdb *db, void *dev_raw, void *dev_flags);
int cdm_seqdb_attach_raw(cdm_ctx *ctx, cdm_seqdb *db, const void *dev_raw, const void *dev_flags);
/* the same with one wasExtended flag per sequence (device array of n bytes, NULL = all 0), and the flags of a DB copied out:
 * the query-sharded stages of a multi-GPU run hand whole DB slices around, flags included */
int cdm_seqdb_from_packed_ext(cdm_ctx *ctx, const void *dev_codes, const void *dev_nmask16, const void *dev_lengths, const void *dev_keys,
                              const void *dev_ext, uint64_t n, uint64_t words, cdm_seqdb **out);
int cdm_seqdb_copy_ext(cdm_ctx *ctx, const cdm_seqdb *db, void *dev_ext);
/* The same packed form to and from HOST buffers: what a module process leaves in a binary side-car next to the sequence DB it read or wrote
 * (csrc/host/sidecar.cpp), so that the next module of data/nuclassemble.sh:100-146 uploads 2 bits per base instead of parsing and packing
 * the text DB again (lib/mmseqs/src/commons/DBReader.cpp:108-133, DBWriter.cpp:322-427 stay the format every reference module reads).
 * nmask16 / raw / raw_flags may be NULL: export - not wanted; import - no letter beyond ACGT / no raw plane.  raw_flags of an export:
 * one byte per sequence, bit 0 = the sequence has a letter beyond ACGT, bit 1 = its row of the raw plane counts. */
int cdm_seqdb_export_packed(cdm_ctx *ctx, const cdm_seqdb *db, void *codes, void *nmask16, void *lengths, void *keys, void *ext, void *raw, void *raw_flags);
int cdm_seqdb_import_packed(cdm_ctx *ctx, const void *codes, const void *nmask16, const void *lengths, const void *keys, const void *ext, const void *raw,
                            const void *raw_flags, uint64_t n, uint64_t words, cdm_seqdb **out);

/* ---------------------------------------------------------------------------------------------------------
 * Damage model.  Replaces the per-thread initDeamProbabilities + getSeqErrorProf calls
 * (src/assembler/correction.cpp:184-197, ancientReadsResults.cpp:161-173; nuclassembleUtil.cpp:821-1007,49-65).
 * prefix is --ancient-damage: files <prefix>5p.prof and <prefix>3p.prof; "" means no damage (zero matrices).
 * The host side of the library builds the 11+11 substitution matrices in the reference's mixed
 * long double/double arithmetic and from them the log look-up tables the kernels use.
 */
int cdm_damage_load(cdm_ctx *ctx, const char *prefix);
/* the 2 x 11 x 4 x 4 matrices as long double (fwd then rev), for tests */
int cdm_damage_get(cdm_ctx *ctx, long double *out352);

/* ---------------------------------------------------------------------------------------------------------
 * kmermatcher.  Replaces doComputation + writeKmerMatcherResult + the fill-in loop
 * (lib/mmseqs/src/linclust/kmermatcher.cpp:391-451, 453-562, 815-930, 717-729) in single-split semantics.
 */
typedef struct cdm_kmer_params {
    int32_t kmer_size;               /* -k (20 for reads, 22 for contigs) */
    int32_t kmers_per_seq;           /* --kmer-per-seq */
    float kmers_per_seq_scale;       /* --kmer-per-seq-scale */
    uint64_t hash_shift;             /* --hash-shift (xxhash seed) */
    int32_t ignore_multi_kmer;       /* --ignore-multi-kmer */
    int32_t include_only_extendable; /* --include-only-extendable */
    int32_t cov_mode;                /* --cov-mode */
    float cov_thr;                   /* -c */
} cdm_kmer_params;

/* one prefilter line "targetKey \t score \t diagonal" (lib/mmseqs/src/prefiltering/QueryMatcher.h:35-39,114-126);
 * score < 0 means the query has to be reverse-complemented */
typedef struct cdm_hit {
    uint32_t target; /* sequence index (not key) */
    int32_t score;
    int32_t diagonal; /* already truncated to short as the reference stores it */
} cdm_hit;

int cdm_kmermatch(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out);
/* offsets: n+1 entries (CSR by query index); every query has at least its self hit as first record */
int cdm_hits_upload(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *offsets, const cdm_hit *hits, cdm_hits **out);
uint64_t cdm_hits_count(const cdm_hits *h);
int cdm_hits_download(cdm_ctx *ctx, const cdm_hits *h, uint64_t *offsets, cdm_hit *hits);
void cdm_hits_free(cdm_hits *h);

/* kmermatcher split over several GPUs without changing its result (the reference's MPI path splits the k-mer
 * space the same way, lib/mmseqs/src/linclust/kmermatcher.cpp:634-663, but merges per-split results lossily - here the
 * single-split semantics are kept).  Every rank holds the whole sequence DB.
 *   cdm_kmermatch_part   doComputation's first half (:391-430) on the k-mer range part/nparts (ranges are equal slices of the
 *                        2k-bit k-mer space in k-mer order; the whole-sequence hash tuples belong to the last range): extraction,
 *                        sort 1, assignGroup -> (rep, id, diagonal, strand) group keys of this range
 *   cdm_kpart_info       info[0] real tuples of the range, [1] kept group keys, [2] 1 if a real tuple lies below the range, [3] n
 *   cdm_kpart_stale      the left-over list of the reference's run-past-the-end scan (kmermatcher.cpp:875-887) from local
 *                        k-mer-order index J of this range: out[0] count, out[1] sequence id, out[2..63] positions, out[66] = 1 if
 *                        the scan consumed every tuple up to the end of the range (then it goes on in the next range)
 *   cdm_kpart_gather     the group keys grouped by representative (k-mer order inside); offsets[r]..offsets[r+1] is the slice
 *                        for the rank that owns representatives [r n/nranks, (r+1) n/nranks); *dev_keys is a DEVICE pointer
 *                        owned by the handle (send buffer of the all-to-all)
 *   cdm_kpart_sort       second half, sort 2 (:431), on the keys received from all ranks, concatenated in rank order (device
 *                        buffer).  head (cdm_kpart_cont_cap() + 3 values): the start of the sorted array while the target id
 *                        stays that of the first tuple - what the per-target scan of the rank in front runs into
 *                        (head[0] count, head[1] that id, head[2] = 1 if it is the whole array, head[3..] entries);
 *                        info[0] = sorted tuples, info[1] = target id of the last one
 *   cdm_kpart_vote       writeKmerMatcherResult (:815-930): hits of the representatives this rank owns (every other sequence gets
 *                        its self hit only).  cont: what this rank's last scan runs into, built from the later ranks' heads
 *                        (cont[0] entries, cont[1] = this rank's last target id, cont[2] = 1 if the scan then goes on into the
 *                        left-over list, cont[3..] entries; NULL = nothing but the left-over list behind this rank's tuples);
 *                        stale = the combined left-over list (65 values as above)
 */
typedef struct cdm_kpart cdm_kpart;
int cdm_kmermatch_part(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, int part, int nparts, cdm_kpart **out);
/* The same first half with the extraction split by READS (what cdm_kmermatch_dist does): rank r extracts the k-mers of ITS block of the
 * sequences only - block r of nranks of the (length descending, id ascending) order fillKmerPositionArray's result is sorted into
 * (kmermatcher.cpp:391-430), the blocks cut so that they hold about the same number of k-mer slots - and orders its tuples by
 * CDM_KPART_SLICES equal slices of the k-mer space (its top 8 bits: one radix pass).  cdm_kpart_outgoing names the send buffers (offsets[s]..offsets[s+1] = the tuples of
 * slice s, keys 8 bytes and values *val_bytes each; the whole-sequence hash tuples sort behind every k-mer).  The CALLER cuts the
 * ranks' k-mer ranges as runs of slices from all ranks' counts (canonical k-mers crowd the low end of the k-mer space: equal slices
 * are not equal shares), sends every range to its rank and the hash tuples to the last one; cdm_kmermatch_split_finish takes what
 * arrived, CONCATENATED IN RANK ORDER (device buffers; `below` = 1 if any rank holds a tuple of a range in front of this one), and
 * leaves the handle where cdm_kmermatch_part leaves it (rank r of nranks ranges). */
#define CDM_KPART_SLICES 256
int cdm_kmermatch_split_begin(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, int rank, int nranks, cdm_kpart **out);
int cdm_kpart_outgoing(const cdm_kpart *h, uint64_t *offsets, const void **keys, const void **vals, int *val_bytes,
                       const void **hash_keys, const void **hash_vals, uint64_t *n_hash);
int cdm_kpart_set_range(cdm_kpart *h, int rank, int nranks);     /* before split_finish: finish as range `rank` of `nranks` (a handle begun as block 0 of 1: all sequences) */
int cdm_kmermatch_split_finish(cdm_ctx *ctx, cdm_kpart *h, const void *keys, const void *vals, uint64_t m,
                               const void *hash_keys, const void *hash_vals, uint64_t n_hash, int below);
int cdm_kpart_info(const cdm_kpart *h, uint64_t info[4]);
int cdm_kpart_stale(cdm_ctx *ctx, cdm_kpart *h, uint64_t J, uint32_t out[67]);
int cdm_kpart_gather(cdm_ctx *ctx, cdm_kpart *h, int nranks, uint64_t *offsets, const void **dev_keys);
/* the same with the owners' ranges given: offsets[t] = first key whose representative is >= bounds[t] (nb ascending sequence ids; the
 * keys are grouped by representative on the first call).  cdm_kmermatch_dist calls it twice: with 4097 equal bounds for a histogram of
 * keys per id range, then with the nranks + 1 bounds it cut from all ranks' histograms. */
int cdm_kpart_gather_at(cdm_ctx *ctx, cdm_kpart *h, int nb, const uint64_t *bounds, uint64_t *offsets, const void **dev_keys);
int cdm_kpart_sort(cdm_ctx *ctx, cdm_kpart *h, const void *dev_keys, uint64_t n_keys, uint32_t *head, uint64_t info[2]);
int cdm_kpart_vote(cdm_ctx *ctx, cdm_kpart *h, const uint32_t *cont, const uint32_t *stale, cdm_hits **out);
int cdm_kpart_cont_cap(void);
void cdm_kpart_free(cdm_kpart *h);
/* tuning: head room of the library's device-memory cache for callers whose inputs grow from call to call (the contig iterations of
 * the workflow loop): large blocks are allocated `factor` times the request, so the next, larger request fits a cached block instead of
 * mapping device memory anew.  1 = off (default); process-wide. */
void cdm_pool_headroom(float factor);
/* the allocator's counters since the process began: [0] requests, [1] served without the driver, [2] calls that asked the driver for
 * memory, [3] their bytes, [4] the nanoseconds they took, [5] times free memory was given back after an out-of-memory; and now:
 * [6] bytes mapped into the arenas of all threads, [7] bytes of them in blocks that are in use (head room included) */
void cdm_pool_stats(uint64_t out[8]);
/* The library reads its CDM_* switches (A/B and test aids) from the environment ONCE per process, not with getenv() on
 * its call paths (getenv is not safe beside a setenv elsewhere in the process).  cdm_env_refresh() reads them again - for tests and A/B
 * runs that change a switch inside one process; call it while no other thread is inside the library. */
void cdm_env_refresh(void);
/* device-to-device copy on the context's stream (synchronises it): moves library-owned buffers into caller tensors */
int cdm_dev_copy(cdm_ctx *ctx, void *dst, const void *src, uint64_t bytes);

/* ---------------------------------------------------------------------------------------------------------
 * rescorediagonal (--rescore-mode 3, query DB == target DB).  Replaces the loop at
 * lib/mmseqs/src/alignment/rescorediagonal.cpp:145-356 (DistanceCalculator.h:93-175,204-220).
 */
typedef struct cdm_rescore_params {
    float seq_id_thr; /* --min-seq-id */
    double eval_thr;  /* -e */
    int32_t cov_mode; /* --cov-mode */
    float cov_thr;    /* -c */
    int32_t seq_id_mode; /* --seq-id-mode (0 only) */
    int32_t min_aln_len; /* --min-aln-len */
} cdm_rescore_params;

/* one alignment record = the ten text columns of Matcher::resultToBuffer (lib/mmseqs/src/alignment/Matcher.cpp:356-404)
 * minus lengths (taken from the seq DB); raw_score/ident let the host derive bit score, E-value and seq.id. text.
 * q_start > q_end encodes a reverse-strand hit as in the reference (rescorediagonal.cpp:294-297). */
typedef struct cdm_aln {
    uint32_t target;  /* sequence index */
    int32_t raw_score;
    int32_t ident;    /* identical columns */
    int32_t q_start, q_end, db_start, db_end;
    float seq_id;     /* value a reader of the text record gets: (float)strtod(fastSeqIdToBuffer(ident/alnLen)) */
} cdm_aln;

int cdm_rescore(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_rescore_params *par, cdm_alns **out);
/* rescorediagonal --rescore-mode 0 --wrapped-scoring 1 (query DB == target DB): linclust's Hamming-distance pre-clustering of the
 * assembled contigs (lib/mmseqs/data/workflow/linclust.sh:27-31; rescorediagonal.cpp:145-356 in that mode,
 * DistanceCalculator::computeUngappedWrappedAlignment, DistanceCalculator.h:57-91).  Output: the hits that pass, as prefilter
 * records (target, 100 * seq. id. signed by the strand, diagonal), same CSR over the queries. */
typedef struct cdm_hamming_params {
    float seq_id_thr;   /* --min-seq-id */
    double eval_thr;    /* -e (the mode's E-value is 0: only a negative threshold drops everything) */
    int32_t cov_mode;   /* --cov-mode */
    float cov_thr;      /* -c */
    int32_t seq_id_mode; /* --seq-id-mode 0 | 1 | 2 */
    int32_t min_aln_len; /* --min-aln-len */
    int32_t reverse_prefilter; /* the prefilter DB's type is DBTYPE_PREFILTER_REV_RES (kmermatcher's): a negative score = reverse-strand hit */
} cdm_hamming_params;
int cdm_rescore_hamming(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_hamming_params *par, cdm_hits **out);
int cdm_alns_upload(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *offsets, const cdm_aln *alns, cdm_alns **out);
uint64_t cdm_alns_count(const cdm_alns *a);
int cdm_alns_download(cdm_ctx *ctx, const cdm_alns *a, uint64_t *offsets, cdm_aln *alns);
void cdm_alns_free(cdm_alns *a);
/* host helpers for the text codec of the alignment DB: E-value, bit score (EvalueComputation.h:18-40 over ALP) */
double cdm_evalue(double raw_score, double query_len, uint64_t db_residues);
/* E-value and bit score of a GAPPED nucleotide alignment (the `align` step of linclust; Alignment.cpp:273: EvalueComputation with gap
 * costs): implemented for --gap-open 5 --gap-extend 2, whose Gumbel parameters are ALP's estimate taken from the reference's object code */
int cdm_gapped_evalue(int gap_open, int gap_extend, double raw_score, double query_len, uint64_t db_residues, double *evalue, int *bit_score);
int cdm_bit_score(double raw_score);

/* ---------------------------------------------------------------------------------------------------------
 * ancient_correction.  Replaces the loop at src/assembler/correction.cpp:200-476 (mostLikeliBaseRead :7-123).
 * Output: a new sequence DB with identical keys/lengths/flags and corrected bases.
 */
typedef struct cdm_ancient_params {
    float seq_id_thr;            /* --min-seq-id */
    float corr_reads_ry_seq_id;  /* --min-ryseq-id-corr-reads */
    float ry_seq_id_thr;         /* rySeqIdThr (not settable on the module's command line; 0.99) */
    float rand_align_penal;      /* --ext-random-align */
    float excess_penal;          /* --excess-penalty */
    float likelihood_threshold;  /* --likelihood-ratio-threshold */
    int32_t unsafe;              /* --unsafe: 1 = consensusCaller's majority vote over the extending targets (nuclassembleUtil.cpp:570-702) */
    int32_t min_cov_safe;        /* --min-cov-safe */
    uint64_t max_seq_len;        /* --max-seq-len */
} cdm_ancient_params;

int cdm_correct(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out);

/* ---------------------------------------------------------------------------------------------------------
 * ancient_read_assemble.  Replaces the loops at src/assembler/ancientReadsResults.cpp:178-581.
 * db must be the corrected DB.  Output: new sequence DB (extended sequences get ext=1, others are copied through).
 * scores (optional, may be NULL): for tests, the per-candidate likelihood of the first scoring round
 * (calcLikelihoodConsensus, nuclassembleUtil.cpp:203-374): one double sLenNorm per alignment record, NaN where the
 * record is not a candidate.
 */
int cdm_extend(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out,
               double *scores);

/* ---------------------------------------------------------------------------------------------------------
 * ancient_contig_merge (contig phase of the workflow, data/nuclassemble.sh:148-196).  Replaces the loop at
 * src/assembler/ancientContigsResults.cpp:94-509.  db must be the corrected contig DB.  The per-alignment column work
 * (orientation, identities, the counts of updateSeqIdConsensus / ancientMatchCount) runs on the device, and since round 5 so do the
 * queue - a Beta-posterior comparator built on the C library's lgammaf / logf (:25-70), not a strict weak ordering: the device reads
 * those two functions from tables of the library's own values and replays libstdc++'s heap step for step (csrc/contigqueue.hip) -
 * and the extension loop (:276-470).  The library's host code (the same libstdc++ priority queue as the reference) runs for
 * par->unsafe = 1, for small calls in a process that has not filled the tables yet, and for the rare query the device hands back
 * (CDM_CONTIG_QUEUE=host|device pins either).  Same result either way.
 * merge_seq_id_thr is --min-merge-seq-id; par->ry_seq_id_thr, max_seq_len, unsafe, min_cov_safe are used from par.
 */
int cdm_contig_merge(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, float merge_seq_id_thr,
                     cdm_seqdb **out);

/* ---------------------------------------------------------------------------------------------------------
 * cyclecheck (data/nuclassemble.sh:19-60, after every contig iteration).  Replaces the loop at src/assembler/cyclecheck.cpp:75-258
 * (k = 22 as setCycleCheckDefaults fixes it): *cyclic receives the contigs the module writes - cut at the split diagonal when
 * chop_cycle (--chop-cycle) is set, whole otherwise, wasExtended 0 - and, if rest is not NULL, *rest the others (what the workflow's
 * "_noneCycle" index selects), both in the order of db.  Contigs of max_seq_len (--max-seq-len) letters or more are never cyclic
 * (:107-112).  split (host, db size entries, may be NULL) receives the split diagonal of every contig, 0 = not cyclic.
 */
int cdm_cyclecheck(cdm_ctx *ctx, const cdm_seqdb *db, uint32_t max_seq_len, int chop_cycle, cdm_seqdb **cyclic, cdm_seqdb **rest, uint32_t *split);

/* ---------------------------------------------------------------------------------------------------------
 * Multi-GPU inside the library: one rank per device (one process per GPU, or one host thread per GPU of one process), RCCL over
 * xGMI called directly (csrc/dist.hip).  The reference shards inside its modules - kmermatcher by k-mer range
 * (lib/mmseqs/src/linclust/kmermatcher.cpp:634-663, merged :742-784), rescorediagonal by query range
 * (lib/mmseqs/src/alignment/rescorediagonal.cpp:399-421) - and so do these calls: every rank holds the whole sequence DB, the result is
 * bit-identical to the single-device calls (kmermatcher's first half: every rank extracts its block of the sequences, all-to-alls
 * carry the k-mer tuples to the rank of their k-mer range - cdm_kmermatch_split_* -; ONE all-to-all of the group keys to the owners of
 * their representatives; the owned ranges of the stages' result DBs all-gathered).
 *   cdm_comm_unique_id       rank 0: the 128 bytes of ncclGetUniqueId, to be handed to every rank by whatever launched them
 *   cdm_comm_create_rccl     a rank's communicator (ncclCommInitRank on the context's device; librccl is loaded on first use)
 *   cdm_comm_create_ops      the same over collectives the caller supplies (tests: W ranks on one device; another collective library)
 *   cdm_kmermatch_dist       kmermatcher over the ranks: the hits of the representatives this rank OWNS, self hits for all others; the
 *                            union is cdm_kmermatch's result.  Owned = a range of sequence ids per rank, cut by this call so that every
 *                            rank owns about the same number of group keys (representatives are the longest, then lowest ids: equal id
 *                            ranges gave the first of 8 ranks 88 % of them at 50 M reads); cdm_comm_owned reports the ranges
 *   cdm_comm_owned           bounds[world + 1]: rank r owns the sequences [bounds[r], bounds[r + 1]) of a DB of n sequences - as the
 *                            communicator's last cdm_kmermatch_dist on such a DB cut them, equal ranges before any; CDM_ERR_INVALID for a
 *                            DB of another size than the one the ranges were cut for
 * A failure on one rank (out of memory for an exchange buffer, a refusal that depends on the rank's data) is agreed on before the next
 * collective is entered: every rank returns an error from the same call, none is left waiting in RCCL.
 *   cdm_seqdb_allgather_owned   the owned ranges (cdm_comm_owned) of the ranks' DBs (same number of sequences on every rank) -> the complete DB
 *   cdm_reads_iteration_dist    one iteration of the reads loop (data/nuclassemble.sh:100-146) over the ranks; hits / alns hold the
 *                            owned queries' records, corr / next are complete on every rank and equal the single-device DBs
 *   cdm_contig_iteration_dist   one iteration of the contig loop (data/nuclassemble.sh:148-196) over the ranks, up to ancient_contig_merge:
 *                            the reference's loop over independent queries (ancientContigsResults.cpp:94-509) sharded by owned queries like
 *                            the other stages; corr / next complete on every rank; the script's cyclecheck step is the caller's
 */
typedef struct cdm_comm cdm_comm;
typedef struct cdm_comm_ops {
    void *user;
    /* host buffers: recv[p * bytes ..] = rank p's send (bytes each) */
    int (*all_gather_host)(void *user, const void *send, void *recv, uint64_t bytes);
    /* device buffers, byte offsets [world + 1]: peer p gets send[send_off[p], send_off[p + 1]) and fills recv[recv_off[p], recv_off[p + 1]);
     * enqueued on `stream` (a hipStream_t) or complete on return */
    int (*all_to_all_dev)(void *user, const void *send, const uint64_t *send_off, void *recv, const uint64_t *recv_off, void *stream);
    /* device buffers: recv[recv_off[p], recv_off[p + 1]) = rank p's send (send_bytes of them differ between the ranks) */
    int (*all_gather_dev)(void *user, const void *send, uint64_t send_bytes, void *recv, const uint64_t *recv_off, void *stream);
} cdm_comm_ops;
int cdm_comm_unique_id(void *id128);
int cdm_comm_create_rccl(cdm_ctx *ctx, int rank, int world, const void *id128, cdm_comm **out);
int cdm_comm_create_ops(cdm_ctx *ctx, int rank, int world, const cdm_comm_ops *ops, cdm_comm **out);
/* tests: the RCCL transport's own code (its buffers, its pieces of 256 MB, its offsets) over a stand-in for RCCL's send / recv /
 * all-gather inside ONE process - the ranks are host threads that share a device, which RCCL itself refuses.  group: from
 * cdm_comm_standin_group(world), shared by the world's ranks. */
void *cdm_comm_standin_group(int world);
int cdm_comm_create_standin(cdm_ctx *ctx, void *group, int rank, cdm_comm **out);
void cdm_comm_free(cdm_comm *c);
int cdm_comm_rank(const cdm_comm *c);
int cdm_comm_world(const cdm_comm *c);
int cdm_comm_owned(const cdm_comm *c, uint64_t n, uint64_t *bounds);
/* what the communicator's last cdm_kmermatch_dist did: 0 nothing yet, 1 every rank ran kmermatcher whole (two ranks; a DB that takes the wide
 * group key), 2 every rank extracted all reads and kept its range of the k-mer space, 3 the reads were split and the k-mer tuples travelled,
 * 4 equal slices of the k-mer space by value (cdm_kmermatch_part), 5 the first half by ranges of the k-mer space, the kept group keys (wide form
 * included) all-gathered, sort 2 and the vote on every rank (DBs that take the wide group key) */
int cdm_comm_last_path(const cdm_comm *c);
int cdm_kmermatch_dist(cdm_ctx *ctx, cdm_comm *comm, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out);
int cdm_seqdb_allgather_owned(cdm_ctx *ctx, cdm_comm *comm, const cdm_seqdb *local, cdm_seqdb **out);
int cdm_reads_iteration_dist(cdm_ctx *ctx, cdm_comm *comm, const cdm_seqdb *db, const cdm_kmer_params *kpar, const cdm_rescore_params *rpar,
                             const cdm_ancient_params *apar, cdm_hits **hits, cdm_alns **alns, cdm_seqdb **corr, cdm_seqdb **next);
int cdm_contig_iteration_dist(cdm_ctx *ctx, cdm_comm *comm, const cdm_seqdb *db, const cdm_kmer_params *kpar, const cdm_rescore_params *rpar,
                              const cdm_ancient_params *apar, float merge_seq_id_thr, cdm_alns **alns, cdm_seqdb **corr, cdm_seqdb **next);

#ifdef __cplusplus
}
#endif
#endif
