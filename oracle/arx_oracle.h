/*
 * oracle/arx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, written from scratch) of the reference's per-pair hot path:
 * FM-index seeding, chaining, banded extension, mate rescue and CIGAR generation, exactly as
 * pdimens/arachne drives the vendored BWA 0.7.16a core through its cgo bridge
 * (/root/reference/src/gobwa/gobwa.go:226-337,400-415).  Every function cites the reference
 * file:line it follows.  The product (arachne_amd/, libarachne_amd.so) never includes, links or
 * calls anything in this directory; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * Parity pin: this restatement is checked bit-for-bit against oracle/_ref/libbwaref.so (the
 * reference's own C sources compiled in place) by tests/test_oracle_vs_ref.py, and against the
 * committed vectors in tests/golden/ (generated from that .so by tests/golden/make_golden.py).
 */
#ifndef ARX_ORACLE_H
#define ARX_ORACLE_H
#include <stdint.h>

#define ORA_REG_W 20 /* same row layout as oracle/ref_driver.c */
#define ORA_ALN_W 12

typedef struct ora_ctx ora_ctx_t;

/* instrumentation: algorithmic work counters accumulated over a batch */
typedef struct {
	int64_t n_reads;
	int64_t ext_same_block;  /* E1: bwt_extend calls whose k-1 and k-1+size share one Occ block */
	int64_t ext_two_block;   /* E2: ... fall in two blocks */
	int64_t sa_lookups;      /* N_sa */
	int64_t sa_lf_steps;     /* S: LF steps over all bwt_sa calls */
	int64_t n_regs;          /* regs produced (post rescue) */
	int64_t cells_extend;    /* ksw_extend2 cell updates */
	int64_t cells_u8;        /* ksw_u8 cell updates (qlen_padded x rows) */
	int64_t cells_global;    /* ksw_global2 cell updates */
	int64_t n_extend_calls, n_u8_calls, n_global_calls;
	int64_t ext3_same_block, ext3_two_block; /* the share of E1 / E2 spent in the third seeding pass (bwt_seed_strategy1) */
	int64_t extb_same_block, extb_two_block; /* the share spent in backward extensions (the backward sweeps of bwt_smem1a) */
	int64_t n_smem_calls;                    /* bwt_smem1a calls (first pass + re-seeding) */
	int64_t sa_lf_steps8;                    /* LF steps of the same lookups up to the first row that is a multiple of 8 */
	int64_t sa_lf_steps4;                    /* ... of 4 (the product's suffix-array sample since round 3) */
	int64_t ext_rows_qlen, ext_rows_cols;    /* ksw_extend2: rows computed x query length, and x the columns of the device kernel's length class (16 lanes x C) */
} ora_counters_t;

#ifdef __cplusplus
extern "C" {
#endif

ora_ctx_t *ora_open(const char *prefix);            /* reads <prefix>.{bwt,sa,pac,ann,amb,alt} */
void ora_close(ora_ctx_t *c);
int64_t ora_l_pac(ora_ctx_t *c);
int64_t ora_seq_len(ora_ctx_t *c);
int64_t ora_primary(ora_ctx_t *c);
int ora_n_seqs(ora_ctx_t *c);
void ora_counters(ora_ctx_t *c, ora_counters_t *out, int reset);

/* per-function known-answer entry points (same signatures as the ref_* ones in ref_driver.c) */
void ora_occ4(ora_ctx_t *c, int n, const uint64_t *k, uint64_t *out);
void ora_extend(ora_ctx_t *c, int n, const uint64_t *ik3, int is_back, uint64_t *ok12);
void ora_sa(ora_ctx_t *c, int n, const uint64_t *k, uint64_t *out);
int ora_collect_intv(ora_ctx_t *c, int len, const uint8_t *seq, uint64_t *out, int cap);
int ora_chains(ora_ctx_t *c, int len, const uint8_t *seq, int do_flt, int64_t *chains, int cap_c, int64_t *seeds, int cap_s, int *n_seeds_out, uint32_t *frac_rep_bits);
void ora_ksw_extend2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int end_bonus, int zdrop, int h0, int *out);
void ora_ksw_align2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int xtra, int *out);
int ora_ksw_global2(ora_ctx_t *c, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int w, int *score, uint32_t *cigar, int cap);
int ora_align1(ora_ctx_t *c, int len, const uint8_t *seq, int64_t *regs, int cap);
int64_t ora_fetch_seq(ora_ctx_t *c, int64_t *beg, int64_t mid, int64_t *end, int *rid, uint8_t *out, int64_t cap);

/* the per-pair path */
double ora_batch_run(ora_ctx_t *c, int64_t n_pairs, const uint8_t *seqs, const int32_t *lens, int score_delta, int n_threads);
void ora_batch_get(ora_ctx_t *c, int64_t *n_reads, int64_t *n_regs, int64_t *n_cig, int64_t **reg_off, int64_t **regs, int64_t **alns, uint32_t **cigars);

/* the Go half (arx_oracle_rfa.c): candidate post-processing, RFA placement, MAPQ.  Row layout ORA_CAND_W = 18 int64:
 * reg read pos aend reversed rid score mismatches indels soft_clipped soft_clipped_length lap2 active is_proper mapq
 * molecule_id active_molecule in_filtered */
#define ORA_CAND_W 18
int64_t ora_rfa(int64_t n_reads, const int64_t *reg_off, const int64_t *regs, const int64_t *alns, const uint32_t *cigars, const int32_t *lens,
                int n_barcodes, const int64_t *bc_pair_off, const uint8_t *do_rfa, int penalty, int64_t l_pac, const int64_t *ann_off,
                const int64_t *cen_start, const int64_t *cen_end, int64_t *cand_rows, int64_t *cand_off);

/* passes between placement and the BAM records (CIGAR walk, markDuplicates, CheckSplitReads); rows documented at the function */
#define ORA_POST_W 6
#define ORA_SPLIT_W 7
int64_t ora_post(ora_ctx_t *ctx, int64_t n_reads, const int64_t *regs, const int64_t *alns, const uint32_t *cigars, const uint8_t *bases, const int32_t *lens,
                 int n_barcodes, const int64_t *bc_pair_off, int penalty, const int64_t *ann_off, const int64_t *cen_start, const int64_t *cen_end,
                 const int64_t *cand_rows, const int64_t *cand_off, int64_t *post_rows, int64_t *split_rows, int32_t *mm_ref, int32_t *mm_read, int64_t mm_cap);

#ifdef __cplusplus
}
#endif
#endif
