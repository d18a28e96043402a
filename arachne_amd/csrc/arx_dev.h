// arx_dev.h -- common device-side types, constants and helpers of the MI355X hot path.
//
// Every routine cites the reference function it must agree with bit-for-bit
// (paths relative to /root/reference/src/gobwa/bwa/ unless stated otherwise).
// Kernel bodies are written as functors `void operator()(int tid)`; the HIP runtime layer
// (hip_rt.h) wraps them into __global__ launches.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifndef ARX_DEV
#define ARX_DEV __device__
#define ARX_DEVI __device__ __forceinline__
#define ARX_HDI __host__ __device__ __forceinline__ // also used by the host tail of a stage
#define ARX_ATOMIC_OR(p, v) atomicOr((unsigned int *)(p), (unsigned int)(v))
#define ARX_ATOMIC_INC(p) atomicAdd((int *)(p), 1)
#define ARX_ATOMIC_ADD(p, v) atomicAdd((int *)(p), (int)(v))
#define ARX_ATOMIC_MIN(p, v) atomicMin((int *)(p), (int)(v))
#define ARX_ATOMIC_CAS(p, c, v) atomicCAS((int *)(p), (int)(c), (int)(v))
#define ARX_ATOMIC_ADD64(p, v) atomicAdd((unsigned long long *)(p), (unsigned long long)(v))
#define ARX_ATOMIC_MIN64(p, v) atomicMin((long long *)(p), (long long)(v))
#define ARX_ATOMIC_MAX64(p, v) atomicMax((long long *)(p), (long long)(v))
// plain read of a word other lanes of the workgroup update with atomics (which execute in L2): bypass the per-CU L1
#define ARX_LOAD_SHARED(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

namespace arx {

// ---- scoring / heuristics: compile-time defaults only (bwamem.c:48-84 mem_opt_init; no flag reaches them)
constexpr int OPT_A = 1, OPT_B = 4, OPT_O_DEL = 6, OPT_E_DEL = 1, OPT_O_INS = 6, OPT_E_INS = 1;
constexpr int OPT_W = 100, OPT_ZDROP = 100, OPT_PEN_CLIP5 = 5, OPT_PEN_CLIP3 = 5;
constexpr int OPT_MAX_MEM_INTV = 20, OPT_MIN_SEED_LEN = 19, OPT_SPLIT_WIDTH = 10, OPT_MAX_OCC = 500;
constexpr int OPT_MAX_CHAIN_GAP = 10000, OPT_MAX_BAND_TRY = 2, OPT_SPLIT_LEN = 28; // (int)(19*1.5+.499)
constexpr float OPT_MASK_LEVEL = 0.50f, OPT_DROP_RATIO = 0.50f, OPT_MASK_LEVEL_REDUN = 0.95f;
constexpr int KSW_XBYTE = 0x10000, KSW_XSTOP = 0x20000, KSW_XSUBO = 0x40000, KSW_XSTART = 0x80000;
// Arachne's fixed insert model: only FR valid, [-35, 500] (/root/reference/src/gobwa/gobwa.go:229-237)
constexpr int PES_LOW = -35, PES_HIGH = 500, MAX_RESCUE = 50;

constexpr int MAX_READ_LEN = 255;   // 249 until round 3: mates of 250 bases and more take ksw_i16 (bwamem_pair.c:150, ksw.c:232-334) -- its eight-stripe
                                    // form of the rescue SW exists since; 255 is what the byte-wide fields and the 16-lane tilings of the DP kernels hold
// Extensions are binned by query length so that a wavefront's four 16-lane groups run the same register tiling
// (hip_sw_coop.h: C columns per lane, 16 * C > qlen; C = 2, 3, 4, 6, 8, 10, 16)
constexpr int EXT_CLASSES = 7;
ARX_DEVI int ext_class(int qlen) { return qlen < 32 ? 0 : qlen < 48 ? 1 : qlen < 64 ? 2 : qlen < 96 ? 3 : qlen < 128 ? 4 : qlen < 160 ? 5 : 6; }
constexpr int CAP_INTV = 256;       // SMEM intervals kept per read (overflow is reported, never truncated silently)

// error bits raised by kernels into Pipeline::d_err
enum : uint32_t { ERR_INTV_OVERFLOW = 1, ERR_READ_TOO_LONG = 2, ERR_POOL_OVERFLOW = 4, ERR_CIGAR_OVERFLOW = 8, ERR_INTERNAL = 16 };

// ---- index in HBM (layouts of F0/F16 in SURVEY.md §8a; files written by `bwa index`)
struct IndexView {
	const uint32_t *bwt;   // per 128 symbols one 64-byte block: 4 x u64 cumulative A/C/G/T, then 8 x u32 (16 bases each, MSB first)
	const uint64_t *sa;    // sa[i] = SA[i * sa_intv]; sa[0] = -1
	const uint8_t *pac;    // forward strand, 4 bases per byte
	const int64_t *ann_off;
	const int32_t *ann_len;
	const int32_t *ann_alt;
	uint64_t primary, seq_len, L2[5];
	int64_t l_pac;
	int32_t n_seqs, sa_intv;
	// bi-interval of every ktab_k-mer (dev_fm.h: ktab_*), built when the index is opened; null / 0: none
	const uint64_t *ktab; int32_t ktab_k;
	// ... and of every prefix of it down to one base, for the forward extensions of the SMEM pass (levels 1 .. klv_k back to back, level d at
	// entry (4^d - 4) / 3); null / 0: none
	int32_t klv_k; const uint64_t *klv;
	// the WHOLE suffix array and its inverse, 40 bits per entry back to back (dev_fm.h: p40_*), built when the index is opened if the memory is
	// there (GRCh38: 2 x 31 GB): sa40[k] = bwt_sa(k) without the walk; isa40[p] = the row whose suffix starts at text position p.  With both, a
	// forward extension whose interval has shrunk to ONE occurrence goes on by comparing the read with the reference text (dev_fm.h: FwdLane,
	// text mode).  null: none (sampled `sa` only)
	const uint8_t *sa40, *isa40;
};

struct Biv { uint64_t k, l, s, info; };  // bwtintv_t (bwt.h:59): k = x[0], l = x[1], s = x[2], info = beg<<32|end

struct Seed { int64_t rbeg; int32_t qbeg, len; };  // mem_seed_t (bwamem.c:168); score == len on this path

struct Chain {          // mem_chain_t (bwamem.c:174)
	int64_t pos;
	int32_t rid, n, seed_off, w, kept, first, is_alt;
	int32_t head, tail;  // while chaining: linked list through SeedLink::next
	float frac_rep;
};

struct Reg {            // mem_alnreg_t (bwamem.h:66-87) minus hash
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	float frac_rep;
	int32_t pad;
};

struct Aln {            // mem_aln_t (bwamem.h:97-108) minus XA/mapq
	int64_t pos;
	int32_t rid, flag, is_rev, is_alt, NM, n_cigar, cigar_off, score, sub, alt_sc;
};

template <class T> ARX_DEVI T tmin(T a, T b) { return a < b ? a : b; }
template <class T> ARX_DEVI T tmax(T a, T b) { return a > b ? a : b; }
ARX_DEVI int iabs(int x) { return x < 0 ? -x : x; }

// substitution score, bwa_fill_scmat (bwa.c:109-118): match +1, mismatch -4, anything with N -1
ARX_DEVI int sc_mat(int t, int q) { return (t > 3 || q > 3) ? -1 : (t == q ? OPT_A : -OPT_B); }

// ---- reference access (bntseq.c:225 _get_pac, bntseq.h:87 bns_depos, bntseq.c:349 bns_pos2rid, :365 bns_intv2rid)
ARX_DEVI int pac_base(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

// base at doubled coordinate p in [0, 2*l_pac): the reverse half is the complement of the mirrored forward half (bntseq.c:398-419)
ARX_DEVI int ref_base(const IndexView &ix, int64_t p)
{
	return p < ix.l_pac ? pac_base(ix.pac, p) : 3 - pac_base(ix.pac, (ix.l_pac << 1) - 1 - p);
}

ARX_DEVI int64_t depos(const IndexView &ix, int64_t pos, int *is_rev)
{
	*is_rev = pos >= ix.l_pac;
	return *is_rev ? (ix.l_pac << 1) - 1 - pos : pos;
}

ARX_DEVI int pos2rid(const IndexView &ix, int64_t pos_f)
{
	int left = 0, mid = 0, right = ix.n_seqs;
	if (pos_f >= ix.l_pac) return -1;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ix.ann_off[mid]) {
			if (mid == ix.n_seqs - 1) break;
			if (pos_f < ix.ann_off[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

ARX_DEVI int intv2rid(const IndexView &ix, int64_t rb, int64_t re)
{
	int r, rid_b, rid_e;
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	rid_b = pos2rid(ix, depos(ix, rb, &r));
	rid_e = rb < re ? pos2rid(ix, depos(ix, re - 1, &r)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

// clamp [beg,end) to the contig and strand holding mid; the bases themselves are read through ref_base()
// (bntseq.c:421-447 bns_fetch_seq)
ARX_DEVI void fetch_clamp(const IndexView &ix, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int is_rev;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	*rid = pos2rid(ix, depos(ix, mid, &is_rev));
	int64_t far_beg = ix.ann_off[*rid], far_end = far_beg + ix.ann_len[*rid];
	if (is_rev) {
		int64_t t = far_beg;
		far_beg = (ix.l_pac << 1) - far_end;
		far_end = (ix.l_pac << 1) - t;
	}
	if (*beg < far_beg) *beg = far_beg;
	if (*end > far_end) *end = far_end;
}

// cal_max_gap (bwamem.c:621-628) with a=1, o=6, e=1: (int)((double)(qlen-6)/1 + 1.) = qlen-5 for every int qlen
ARX_DEVI int cal_max_gap(int qlen)
{
	int l = qlen - 5;
	l = l > 1 ? l : 1;
	return l < OPT_W << 1 ? l : OPT_W << 1;
}

// ---- klib introsort on an index array, same comparison sequence as ksort.h:176-226 (unstable; tie order is part of parity).
// LT is a functor bool(int a, int b) on element indices.
#ifndef ARX_SORT_ATTR
#define ARX_SORT_ATTR // (toolchain experiments: e.g. __attribute__((noinline, optnone)))
#endif
#ifndef ARX_SORT_ATTR_INS
#define ARX_SORT_ATTR_INS ARX_SORT_ATTR
#endif
#ifndef ARX_SORT_ATTR_COMB
#define ARX_SORT_ATTR_COMB ARX_SORT_ATTR
#endif
// set when ks_introsort's iteration budget runs out (it cannot: see below); one word per translation unit of the device code
#if defined(__HIPCC__)
static __device__ unsigned int g_arx_sort_fail;
#define ARX_SORT_FAIL() atomicOr(&g_arx_sort_fail, 1u)
static __global__ void k_merge_sort_fail(uint32_t *err) { if (g_arx_sort_fail) atomicOr(err, (uint32_t)ERR_INTERNAL); }
#else
static unsigned int g_arx_sort_fail; // host test double
#define ARX_SORT_FAIL() (g_arx_sort_fail = 1u)
#endif
template <class LT> ARX_DEV ARX_SORT_ATTR_INS void ks_insertsort(int *s, int *t, LT lt)
{
	for (int *i = s + 1; i < t; ++i)
		for (int *j = i; j > s && lt(*j, *(j - 1)); --j) { int x = *j; *j = *(j - 1); *(j - 1) = x; }
}

template <class LT> ARX_DEV ARX_SORT_ATTR_COMB void ks_combsort(int n, int *a, LT lt) // ksort.h:154-175
{
	const double shrink = 1.2473309501039786540366528676643;
	int do_swap, gap = n;
	do {
		if (gap > 2) {
			gap = (int)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		do_swap = 0;
		for (int *i = a; i < a + n - gap; ++i) {
			int *j = i + gap;
			if (lt(*j, *i)) { int x = *i; *i = *j; *j = x; do_swap = 1; }
		}
	} while (do_swap || gap > 2);
	if (gap != 1) ks_insertsort(a, a + n, lt);
}

// The body below is klib's ks_introsort (ksort.h:176-226) with its control flow written in structured form and an iteration budget in
// its loops -- same comparisons in the same order, so the (unstable) result is klib's.  Why: hipcc 7.2 (AMD clang 22) at -O2/-O3 emits
// gfx950 code for this function that never terminates inside KDedup / KRescueStep (bisected in round 2: everything else of those
// kernels at -O3 with only this function `optnone` passes every GPU test; `noinline` alone still hangs; the structured form alone still
// hangs; with one more exit edge in the loops -- the budget below, which cannot run out: the loops make fewer than 4 (n + 8)^2 steps
// in total -- the same -O3 build terminates and is bit-identical; stand-alone kernels around the function do not reproduce it,
// tools/repro_introsort_hang.hip).  DESIGN.md "toolchain notes" has the record.
#ifndef ARX_INTROSORT_UNSTRUCTURED
template <class LT> ARX_DEV ARX_SORT_ATTR void ks_introsort(int n, int *a, LT lt)
{
	int *st_l[64], *st_r[64], st_d[64], top = 0;
	int d, rp, x, *s, *t, *i, *j, *k;
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) { x = a[0]; a[0] = a[1]; a[1] = x; } return; }
	for (d = 2; (1 << d) < n; ++d) {}
	s = a; t = a + (n - 1); d <<= 1;
	bool done = false;
	long budget = 4L * (n + 8) * (n + 8); // never reached (see above)
	// ... and if it ever were, nothing stays silent: the flag below ends up in the error word of every batch that finishes afterwards
	// (HipRT::merge_sort_fail -> ERR_INTERNAL), and the range is left fully sorted by the insertion sort (whose order of equal keys
	// need not be klib's: hence the error)
#define ARX_GUARD() if (--budget < 0) { ARX_SORT_FAIL(); ks_insertsort(a, a + n, lt); return; }
	while (!done) {
		ARX_GUARD()
		if (s < t) {
			--d;
			if (d == 0) { ks_combsort((int)(t - s) + 1, s, lt); t = s; }
			else {
				i = s; j = t; k = i + ((j - i) >> 1) + 1;
				if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
				else k = lt(*j, *i) ? i : j;
				rp = *k;
				if (k != t) { x = *k; *k = *t; *t = x; }
				bool more = true;
				while (more) {
					++i; while (lt(*i, rp)) { ++i; ARX_GUARD() }
					--j; while (i <= j && lt(rp, *j)) { --j; ARX_GUARD() }
					ARX_GUARD()
					more = j > i;
					if (more) { x = *i; *i = *j; *j = x; }
				}
				x = *i; *i = *t; *t = x;
				if (i - s > t - i) {
					if (i - s > 16) { st_l[top] = s; st_r[top] = i - 1; st_d[top] = d; ++top; }
					s = t - i > 16 ? i + 1 : t;
				} else {
					if (t - i > 16) { st_l[top] = i + 1; st_r[top] = t; st_d[top] = d; ++top; }
					t = i - s > 16 ? i - 1 : s;
				}
			}
		} else if (top == 0) done = true;
		else { --top; s = st_l[top]; t = st_r[top]; d = st_d[top]; }
	}
	ks_insertsort(a, a + n, lt);
#undef ARX_GUARD
}
#else
template <class LT> ARX_DEV ARX_SORT_ATTR void ks_introsort(int n, int *a, LT lt)
{
	int *st_l[64], *st_r[64], st_d[64], top = 0;
	int d, rp, x, *s, *t, *i, *j, *k;
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) { x = a[0]; a[0] = a[1]; a[1] = x; } return; }
	for (d = 2; (1 << d) < n; ++d) {}
	s = a; t = a + (n - 1); d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { ks_combsort((int)(t - s) + 1, s, lt); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			rp = *k;
			if (k != t) { x = *k; *k = *t; *t = x; }
			for (;;) {
				do ++i; while (lt(*i, rp));
				do --j; while (i <= j && lt(rp, *j));
				if (j <= i) break;
				x = *i; *i = *j; *j = x;
			}
			x = *i; *i = *t; *t = x;
			if (i - s > t - i) {
				if (i - s > 16) { st_l[top] = s; st_r[top] = i - 1; st_d[top] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { st_l[top] = i + 1; st_r[top] = t; st_d[top] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { ks_insertsort(a, a + n, lt); return; }
			--top; s = st_l[top]; t = st_r[top]; d = st_d[top];
		}
	}
}

#endif

} // namespace arx
