/*
 * arachne_amd.h -- C ABI of libarachne_amd.so, the MI355X (gfx950) implementation of Arachne's per-barcode
 * alignment hot path.  Plain pointers and sizes only; no torch / HIP types cross this boundary.
 *
 * What it replaces in the reference (pdimens/arachne @ 2025-09-05), whose Go code reaches the BWA C core one
 * read / one candidate at a time through cgo (src/gobwa/gobwa.go:7-13, src/gobwa/bwa_bridge.h:35-39):
 *
 *   arx_open           <- bwa_idx_load(path, BWA_IDX_ALL) + mem_opt_init()      gobwa.go:128-152 (GoBwaLoadReference, GoBwaAllocSettings)
 *   arx_contigs        <- direct field reads of bwaidx_t.bns / bntann1_t        gobwa.go:28-39,420-432 (GetReferenceContigsInfo, EnumerateContigs)
 *   arx_batch_create   <- SequenceConvert (nst_nt4_table) per read              gobwa.go:159-167; the caller passes 0..4 codes
 *   arx_batch_run      <- per pair: mem_align1_core x2, the two mem_matesw loops, InterpretAlign           gobwa.go:226-337 (GoBwaMemMateSW)
 *                         per candidate: mem_reg2aln                                                      gobwa.go:400-415 (GoBwaSmithWaterman)
 *                         i.e. loops A and B of DoRFAForOneBarcode               src/aligner/aligner.go:1633-1715,1484-1501
 *   arx_batch_fetch    <- mem_alnreg_v / mem_aln_t returned by value + Arena     gobwa.go:107-126,191,326-327,411-412
 *   arx_batch_free     <- Arena.Free                                             aligner.go:475,500
 *
 * Errors: every entry returns ARX_OK (0) or a negative code and never aborts the process (the reference asserts /
 * err_fatals); arx_last_error() gives the text.  A context may be shared by threads that each own their batches.
 * Results are bit-identical to the reference C core: same regions in the same order, same CIGAR/NM/pos/strand.
 */
#ifndef ARACHNE_AMD_H
#define ARACHNE_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct arx_ctx arx_ctx;
typedef struct arx_batch arx_batch;

enum { ARX_OK = 0, ARX_E_OPEN = -1, ARX_E_ARG = -2, ARX_E_DEVICE = -3, ARX_E_TOO_LARGE = -4, ARX_E_IO = -5 };

/* last_stage values for arx_batch_run: stop after a stage to inspect intermediate results */
enum { ARX_STAGE_SEED = 1, ARX_STAGE_CHAIN = 2, ARX_STAGE_EXTEND = 3, ARX_STAGE_RESCUE = 4, ARX_STAGE_ALN = 5 };

/* mem_alnreg_t (bwa/bwamem.h:66-87) without `hash` (always 0 on this path); frac_rep keeps its float bits */
typedef struct {
	int64_t rb, re;                 /* [rb,re) on the forward+reverse reference */
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	float frac_rep;
	int32_t pad;
} arx_reg;

/* mem_aln_t (bwa/bwamem.h:97-108) without XA and mapq (never read by Arachne) */
typedef struct {
	int64_t pos;                    /* 0-based leftmost position on contig rid */
	int32_t rid, flag, is_rev, is_alt, NM, n_cigar, cigar_off /* into the cigar array of the batch */, score, sub, alt_sc;
} arx_aln;

typedef struct { int64_t pos; int32_t rid, n, seed_off, w, kept, first, is_alt, head, tail; float frac_rep; } arx_chain; /* mem_chain_t */
typedef struct { int64_t rbeg; int32_t qbeg, len; } arx_seed;                                                         /* mem_seed_t  */

/* Builds <prefix>.{bwt,sa,pac,ann,amb} from a plain-text FASTA, byte-identical to the reference's `bwa index`
 * (bwa/bwtindex.c:251-316 bwa_idx_build).  FASTA parsing and the .pac / .ann / .amb files are host work; BWT and suffix array are sorted
 * in HBM when a HIP device is visible (any genome size the HBM holds: GRCh38 in ~10 s) and by a 32-bit host sorter otherwise (2 * l_pac < 2^31;
 * ARX_INDEX_HOST=1 forces it, ARX_INDEX_DEVICE=k picks the device) -- same bytes either way.  msg (may be NULL) receives the error text. */
int arx_index_build(const char *fasta, const char *prefix, char *msg, int32_t msg_cap);

/* Loads <prefix>.{bwt,sa,pac,ann,amb,alt} (files written by `bwa index`) into HBM of `device`; derives there, once, what the kernels use
 * beside the files' content: the Occ blocks re-packed for one popcount per count; the k-mer tables of the seeding passes (ARX_KMER_K /
 * ARX_KMER_FWD: up to 69 GB + 5.7 GB at GRCh38 size, never more than a third of the free memory); the WHOLE suffix array and its inverse,
 * 40 bits per entry (2 x 31 GB at GRCh38 size, when they take no more than half of what is free after the tables; ARX_TEXT_INDEX=0: never),
 * or else the suffix-array sample every 4th row (ARX_SA_DENSE).  arx_index_info says what was built.  ~5 s for GRCh38. */
int arx_open(const char *prefix, int device, arx_ctx **out);
void arx_close(arx_ctx *ctx); /* also frees the context's batches that are still alive: their handles are invalid afterwards */
const char *arx_last_error(arx_ctx *ctx);      /* ctx may be NULL after a failed arx_open */
const char *arx_backend(void);                 /* "hip:gfx950" for the product library */
/* info[0] symbols of the index (2 x l_pac); [1] K of the k-mer table of the third seeding pass (0: none); [2] deepest per-depth table of the
 * forward extensions (0: none); [3] rows per resident suffix-array entry (1: the whole array); [4] 1 if the inverse suffix array is resident
 * (text mode of the seeding passes); [5] bytes of device memory the index holds; [6], [7] reserved (0) */
int arx_index_info(arx_ctx *ctx, int64_t *info /* 8 */);

/* Page-locks host memory the caller hands to arx_batch_create / arx_batch_reset (bases) or receives results in (arx_batch_fetch,
 * arx_batch_rfa_fetch, arx_batch_post_fetch), for as long as it keeps reusing those arrays: copies then run at PCIe speed without a
 * staging copy (the reference recycles its per-work-unit buffers the same way, aligner.go:234 ReturnBuffer, gobwa.go:107-126 Arena).
 * Optional: unregistered memory works, through the library's own staging.  Unregister before freeing.  ARX_E_ARG when the range cannot
 * be (un)registered (e.g. registered already). */
int arx_host_register(void *ptr, int64_t bytes);
int arx_host_unregister(void *ptr);

int arx_contigs(arx_ctx *ctx, int32_t *n, const char *const **names, const int64_t **offsets, const int32_t **lens,
                const int32_t **is_alt, int64_t *l_pac);

/* bases: concatenated reads as codes 0..4 (A,C,G,T,N); lens[n_reads]; read 2i and 2i+1 are mates (either may be empty).
 * Reads of one barcode are contiguous; barcode boundaries do not matter to this stage (pairs are independent). */
int arx_batch_create(arx_ctx *ctx, int32_t n_reads, const uint8_t *bases, const int32_t *lens, arx_batch **out);
/* Replaces the reads of an existing batch: same stream, same work memory, same input buffers when the new reads fit -- a caller in
 * steady state allocates nothing (what the reference does with its per-work-unit buffers: aligner.go:234 ReturnBuffer, gobwa.go:107-126
 * Arena).  Results of the previous run are gone; the uploads are asynchronous on the batch's stream (pinned staging inside). */
int arx_batch_reset(arx_ctx *ctx, arx_batch *b, int32_t n_reads, const uint8_t *bases, const int32_t *lens);
/* Runs the stages up to last_stage.  Stages already done are kept: run(ARX_STAGE_SEED) followed by run(ARX_STAGE_ALN) resumes after
 * seeding; asking for a stage that is already done (or an earlier one) restarts the batch from its reads.  All work memory of the
 * batch is reused by the next run, so results (arx_batch_fetch, arx_batch_rfa_fetch) must be fetched before it. */
int arx_batch_run(arx_ctx *ctx, arx_batch *b, int32_t last_stage);
/* counts[8] = n_reads, n_regs, n_cigar_words, seed occurrences, extension rounds, extension DPs, rescue rounds, rescue SWs */
int arx_batch_counts(arx_ctx *ctx, arx_batch *b, int64_t *counts);
/* ---- device-resident boundary (multi-GPU dataflow): a batch whose reads are in the memory of this context's GPU already
 * -- received from the ingest GPU over RCCL -- and result slabs handed out where they lie, for a send without a host copy.
 * arx_batch_reset_device: arx_batch_reset from device pointers (d_bases: n_bases codes 0..4; d_lens: n_reads lengths); the caller's
 * buffers must be complete when it is called and may be reused when it returns.  arx_batch_device_view: the dense result arrays of
 * arx_batch_fetch / arx_batch_rfa_fetch in device memory (cand_off / cands NULL before arx_batch_rfa); valid until the batch is run, reset
 * or freed; the call waits for the batch's stream.  (In the host test double "device" memory is host memory.) */
typedef struct {
	int64_t n_reads, n_regs, n_cigar, n_cands;
	const int32_t *reg_off; const arx_reg *regs; const arx_aln *alns; const uint32_t *cigars;
	const int32_t *cand_off; const struct arx_cand_ *cands;
} arx_device_view;
int arx_batch_reset_device(arx_ctx *ctx, arx_batch *b, int32_t n_reads, int64_t n_bases, const uint8_t *d_bases, const int32_t *d_lens);
int arx_batch_device_view(arx_ctx *ctx, arx_batch *b, arx_device_view *view);

/* reg_off[n_reads+1], regs[n_regs], alns[n_regs], cigars[n_cigar_words]: caller-allocated from arx_batch_counts */
int arx_batch_fetch(arx_ctx *ctx, arx_batch *b, int32_t *reg_off, arx_reg *regs, arx_aln *alns, uint32_t *cigars);
void arx_batch_free(arx_ctx *ctx, arx_batch *b);

/* ---- the Go half of the per-barcode path (src/aligner/aligner.go): candidates per read (GetChains :1633, GetAlignments :1484),
 * tagBestAlignments :1397, inferMolecules :1300 ... optimizer.Optimize (src/optimizer/optimizer.go:15) and estimateMapQualities :797.
 * Needs arx_batch_run(..., ARX_STAGE_ALN) first.  One candidate per region, or one placeholder (reg = -1, pos = -1) for a read
 * without regions; `active` marks the placement chosen for the read, `mapq` is set on active candidates. */
typedef struct arx_cand_ {
	int64_t pos, aend;              /* Alignment.pos / .aend (0-based, reverse-strand candidates swapped +1, aligner.go:1577-1582) */
	double sum_move;                /* 1 + sum of 10^fastScore over sink molecules (method 2) */
	int32_t reg, read, rid, reversed, score, mismatches, indels, soft_clipped, soft_clipped_length;
	int32_t lap2;                   /* log_alignment_probability * 2 */
	int32_t active, is_proper, mapq, molecule_id, active_molecule, in_filtered /* score >= best - 17 */, best_in_mol, pad;
} arx_cand;
/* bc_pair_off[n_barcodes+1]: pair offsets of the (whole) barcodes in the batch; do_rfa[b]: worthRunningRFA (aligner.go:1018-1030),
 * decided by the caller from the barcode string; penalty: the reference's -i flag (a float64, main.go:28; default -4).  Only integer
 * values are accepted (ARX_E_ARG with a message otherwise): every score term is then a multiple of 0.5 and the sums are exact in any
 * order, which is what makes the result well defined; cen_start/cen_end per contig or NULL. */
int arx_batch_rfa(arx_ctx *ctx, arx_batch *b, int32_t n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa, double penalty,
                  const int64_t *cen_start, const int64_t *cen_end, int64_t *n_cands);
int arx_batch_rfa_fetch(arx_ctx *ctx, arx_batch *b, int32_t *cand_off /* n_reads+1 */, arx_cand *cands /* n_cands */);

/* Overlapping the way home with the next super-batch (a worker with two host threads per handle).  arx_batch_detach copies the dense results of
 * the run -- and of arx_batch_rfa when it has run -- aside on the device (memory of the handle's own, outside its work memory; < 1 ms) and says
 * how large they are: sizes[0..3] = reads, regions, CIGAR words, candidates.  From then on the handle may be reset and run again by one thread
 * while ANOTHER thread calls arx_batch_fetch_detached, which copies them to host arrays on a stream of its own (any pointer may be NULL: not
 * wanted).  The next arx_batch_detach of the same handle must wait until that call has returned (the caller's ordering).  The reference's
 * workers overlap the same way: results of one work unit are written out while the next is aligned (aligner.go:335-371). */
int arx_batch_detach(arx_ctx *ctx, arx_batch *b, int64_t *sizes /* 4 */);
int arx_batch_fetch_detached(arx_ctx *ctx, arx_batch *b, int32_t *reg_off, arx_reg *regs, arx_aln *alns, uint32_t *cigars, int32_t *cand_off, arx_cand *cands);

/* ---- what the reference computes per barcode between placement and the BAM records, on the candidates arx_batch_rfa left on the
 * device (needs arx_batch_rfa first; uses its barcodes, penalty and centromeres):
 *   the CIGAR walk of GetAlignments (aligner.go:1505-1570, against GetSeq gobwa.go:50-80): matches and the mismatch locations in
 *   reference (contig) and read coordinates -- the host never re-fetches the reference; readmap_s/_e (gobwa.go:368-369);
 *   markDuplicates (aligner.go:611-641); CheckSplitReads / GetSplitAlignment (split.go:31-163). */
typedef struct {
	int32_t qb, qe;                 /* Alignment.readmap_s / readmap_e */
	int32_t matches;                /* Alignment.matches */
	int32_t n_mm, mm_off;           /* Alignment.mismatchLocs / mismatchReadLocs = mm_ref / mm_read[mm_off .. mm_off + n_mm) */
	int32_t duplicate;              /* Alignment.duplicate */
} arx_cand_post;
typedef struct {                    /* per read: Alignment.secondary of its active candidate */
	int32_t split;                  /* candidate index, -1 for none */
	int32_t mapq, is_proper;        /* split.mapq, split.is_proper */
	int32_t n_split_cand;           /* candidates that passed split.go:85-97 */
	int32_t order_pinned;           /* 0: > 12 such candidates with a score tie that decides the result -- Go's unstable sort.Sort picks there */
	int32_t second_best2, score2;   /* split.mapq_data.second_best_score * 2, .score * 2 */
	int32_t pad;
} arx_split;
int arx_batch_post(arx_ctx *ctx, arx_batch *b, int64_t *n_mm);
/* post[n_cands], split[n_reads], mm_ref[n_mm], mm_read[n_mm]; any of them may be NULL */
int arx_batch_post_fetch(arx_ctx *ctx, arx_batch *b, arx_cand_post *post, arx_split *split, int32_t *mm_ref, int32_t *mm_read);

/* ---- in front of the path: the reference's paired FASTQ reader (src/fastqreader/reader.go) re-shaped to deliver super-batches of
 * whole barcode sets -- what its producer loop (aligner.go:335-358) hands to one worker per set, one read pair per cgo call.
 * A set is what ReadBarcodeSet (reader.go:209-300) returns: consecutive records of one barcode, at most 30000, 201-record chunks
 * while a barcode continues across sets; `unique` is its third result (WorkUnit.unique_barcode), `do_rfa` = worthRunningRFA
 * (aligner.go:1018-1030).  bases/lens/set_pair_off/do_rfa go straight into arx_batch_create and arx_batch_rfa; the rest is what
 * the BAM records need.  Host code only: works without a GPU.  Pointers stay valid until the next call on the same feeder. */
typedef struct arx_feeder arx_feeder;
typedef struct {
	int32_t n_sets, pad;
	int64_t n_pairs;
	int64_t bad_lines;              /* lines skipped while looking for a record start (reader.go:156-159), since open */
	const int64_t *set_pair_off;    /* n_sets + 1 */
	const uint8_t *unique, *do_rfa; /* n_sets */
	const uint8_t *bases;           /* codes 0..4 (nst_nt4_table); read 2i / 2i+1 = Read1 / Read2 of pair i */
	const char *quals;              /* one byte per base, same layout */
	const int32_t *lens;            /* 2 * n_pairs */
	const uint8_t *valid;           /* n_pairs: FastQRecord.Valid (VX:i:1) */
	const int64_t *name_off; const char *names;       /* n_pairs + 1: FastQRecord.ReadInfo */
	const int64_t *rg_off; const char *rgs;           /* n_pairs + 1: FastQRecord.ReadGroupId */
	const int64_t *barcode_off; const char *barcodes; /* n_sets + 1: FastQRecord.Barcode of the set */
} arx_super_batch;
int arx_feeder_open(const char *r1_path, const char *r2_path, arx_feeder **out, char *msg, int32_t msg_cap); /* plain or gzip */
/* appends whole sets until at least target_pairs pairs are held (at least one set); returns the number of sets, 0 at the end of the
 * input, < 0 on a read error */
int arx_feeder_next(arx_feeder *f, int64_t target_pairs, arx_super_batch *out);
void arx_feeder_close(arx_feeder *f);

/* ---- behind the path: the BAM sink.  The reference builds one biogo sam.Record per alignment on a single goroutine and
 * writes it twice (BamThread / AppendBams, src/aligner/bamwriter.go:615-627,279-282), two BGZF goroutines per writer (:118).  Here records
 * arrive in batches as flat arrays -- what AppendBam (:284-566) computes per alignment stays with the caller -- and are encoded and
 * BGZF-compressed on `threads` host threads, blocks written in order.  One arx_bam per output file (the barcode-sorted BAM, each position
 * bucket); host code only.  SAM/BAM specification v1 encoding; aux bytes are passed through as the caller encoded them. */
typedef struct arx_bam arx_bam;
typedef struct {
	int64_t n_records;
	const int64_t *name_off; const char *names;       /* n + 1 offsets; read names, 1..254 bytes each, no NUL */
	const int32_t *flag, *rid, *pos;                  /* rid = -1 and pos = -1 for an unmapped record (bamwriter.go:352-356) */
	const uint8_t *mapq;
	const int32_t *mate_rid, *mate_pos, *tlen;
	const int64_t *cigar_off; const uint32_t *cigars; /* BAM words: length << 4 | op, op in MIDNSHP=X = 0..8 */
	const int64_t *seq_off; const uint8_t *seq;       /* ASCII bases as written (already reverse-complemented where the caller does, :372-375) */
	const uint8_t *qual; int32_t qual_offset;         /* same offsets as seq; qual_offset is subtracted (33: fixQual, :240-247); 255: no qualities (0xff) */
	const int64_t *aux_off; const uint8_t *aux;       /* BAM-encoded aux fields of each record, back to back */
} arx_bam_batch;
/* extra_header: further header lines (e.g. @RG, @PG), each ending in '\n', or NULL; level: zlib 0..9 (-1: 6) */
int arx_bam_open(const char *path, int32_t n_contigs, const char *const *names, const int32_t *lens, const char *extra_header, int32_t threads, int32_t level,
                 arx_bam **out, char *msg, int32_t msg_cap);
int arx_bam_write(arx_bam *w, const arx_bam_batch *batch);
/* stats[4] (may be NULL): records, BGZF blocks, uncompressed bytes, file bytes */
int arx_bam_close(arx_bam *w, int64_t *stats);
const char *arx_bam_error(arx_bam *w);

/* ---- between the path and the sink: the placed candidate of every read of a super-batch as BAM records -- the part of DumpToBams /
 * AppendBam (src/aligner/bamwriter.go:283-568, 635-658) that decides flags, position, MAPQ, mate fields, template length, CIGAR op codes,
 * strand of bases and qualities and the RG / AS / XM / AM / XT / BX / VX tags of the primary record (csrc/bam_records.h lists what is left
 * to the caller: split records and their tags).  sb: the super-batch the batch was created from; cand_off / cands: arx_batch_rfa_fetch;
 * alns / cigars: arx_batch_fetch; post: arx_batch_post_fetch's per-candidate records or NULL (no duplicate flags).  The view points
 * into the buffer and stays valid until the next build on it; hand it to arx_bam_write.  Host code only. */
typedef struct arx_recbuf arx_recbuf;
int arx_recbuf_create(arx_recbuf **out);
int arx_recbuf_build(arx_recbuf *rb, const arx_super_batch *sb, const int32_t *cand_off, const arx_cand *cands, const arx_aln *alns, const uint32_t *cigars,
                     const arx_cand_post *post, int32_t threads, arx_bam_batch *view);
const char *arx_recbuf_error(arx_recbuf *rb);
void arx_recbuf_free(arx_recbuf *rb);

/* ---- several GPUs behind one handle (arx_open(prefix, n_devices, ...)): one index replica per device, a super-batch of whole
 * barcodes cut by pair count (greedy longest-processing-time), every device's share on a host thread of its own, the result slabs
 * renumbered into the order of the read set -- byte for byte what ONE batch over everything returns.  devices: HIP device indices or NULL
 * for 0 .. n_devices - 1.  The result arrays are host memory owned by the handle, valid until the next call on it; device_of_barcode says
 * where each barcode ran.  (csrc/arx_multi.cpp: host code on the single-device entry points above; the reads reach every GPU from host
 * memory, so nothing travels between GPUs here -- arachne_amd/shard.py is the RCCL form for reads that arrive on one GPU.) */
typedef struct arx_multi arx_multi;
typedef struct {
	int64_t n_reads, n_regs, n_cigar, n_cands;
	const int32_t *reg_off; const arx_reg *regs; const arx_aln *alns; const uint32_t *cigars;
	const int32_t *cand_off; const arx_cand *cands;
	const int32_t *device_of_barcode; /* n_barcodes: index into the handle's devices */
} arx_multi_result;
int arx_multi_open(const char *prefix, int32_t n_devices, const int32_t *devices, arx_multi **out, char *msg, int32_t msg_cap);
int arx_multi_contigs(arx_multi *m, int32_t *n, const char *const **names, const int64_t **offsets, const int32_t **lens, const int32_t **is_alt, int64_t *l_pac);
int arx_multi_run(arx_multi *m, int32_t n_reads, const uint8_t *bases, const int32_t *lens, int32_t n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa,
                  double penalty, const int64_t *cen_start, const int64_t *cen_end, arx_multi_result *out);
const char *arx_multi_error(arx_multi *m);
void arx_multi_close(arx_multi *m);

/* intermediate results for parity tests (device -> host copies of stage outputs) */
#define ARX_CAP_INTV 256
int arx_batch_debug_intv(arx_ctx *ctx, arx_batch *b, int32_t *n_intv, uint64_t *intv4 /* n_reads*ARX_CAP_INTV*4 */);
int arx_batch_debug_chains(arx_ctx *ctx, arx_batch *b, int32_t *occ_off /* n_reads+1 */, int32_t *n_chain, arx_chain *chains, arx_seed *seeds /* counts[3] each */);
int arx_batch_debug_core(arx_ctx *ctx, arx_batch *b, int32_t *n_core, arx_reg *regs /* counts[3] */);

/* self-test of a device routine that has no host twin (tests): klib's introsort as a wavefront reproduces it (csrc/dev_regs_wave.h:
 * w_introsort -- its order of equal keys is part of every result that passes mem_sort_dedup_patch or mem_chain_flt, ksort.h:176-226)
 * against the one-thread ks_introsort, on n_cases random index arrays of 2..832 entries whose keys come from small ranges, so that ties
 * abound.  *n_bad = arrays on which the two differ. */
int arx_selftest_wave_sort(int32_t device, int32_t n_cases, int64_t seed, int64_t *n_bad);

/* per-kernel device time (HIP events on the launch stream), accumulated since the last reset */
int arx_kernel_times(arx_ctx *ctx, int32_t cap, char *names, int32_t name_w, double *ms, int64_t *calls, int64_t *items);
void arx_kernel_times_reset(arx_ctx *ctx, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif
