// index_build.h -- host-side index builder producing files byte-identical to the reference's `bwa index`
// (bwtindex.c:251-316 bwa_idx_build): <prefix>.pac/.ann/.amb (bntseq.c:227-328 bns_fasta2bntseq, :66-96 bns_dump),
// <prefix>.bwt (bwtindex.c:61-125 bwt_pac2bwt + :151-173 bwt_bwtupdate_core, bwt.c:385-395 bwt_dump_bwt) and
// <prefix>.sa (bwt.c:62-84 bwt_cal_sa, :397-407 bwt_dump_sa).
//
// The BWT of a text is unique, so instead of the reference's two construction algorithms (IS for < 50 Mbp, ropes
// above) any suffix sorter gives its bytes: induced sorting on the host below (32-bit, for builds without a GPU), prefix
// doubling in HBM in hip_index_build.h (what libarachne_amd.so uses when a device is visible; GRCh38-size genomes); the
// sampled SA is read straight off the suffix array instead of being recovered by n LF steps.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace arx {

// ---- drand48 family as POSIX specifies it: X' = (0x5DEECE66D * X + 0xB) mod 2^48; srand48(s): X = s<<16 | 0x330E;
// lrand48 returns the top 31 bits.  The reference replaces every N by lrand48()&3 after srand48(11) (bntseq.c:261,290).
struct Rand48 {
	uint64_t x;
	explicit Rand48(uint32_t seed) : x(((uint64_t)seed << 16) | 0x330E) {}
	uint32_t lrand() { x = (0x5DEECE66DULL * x + 0xB) & ((1ULL << 48) - 1); return (uint32_t)(x >> 17); }
};

// ---- suffix array by induced sorting (SA-IS).  s[0..n) over [0,K), s[n-1] = 0 is the unique smallest symbol.
template <class S, class I> struct SaIs {
	static void get_buckets(const S *s, I *bkt, I n, I K, bool end)
	{
		for (I i = 0; i < K; ++i) bkt[i] = 0;
		for (I i = 0; i < n; ++i) ++bkt[s[i]];
		I sum = 0;
		for (I i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
	}
	static inline bool is_s(const std::vector<uint64_t> &t, I i) { return t[i >> 6] >> (i & 63) & 1; }
	static inline bool is_lms(const std::vector<uint64_t> &t, I i) { return i > 0 && is_s(t, i) && !is_s(t, i - 1); }

	static void induce(const S *s, I *SA, I n, I K, std::vector<I> &bkt, const std::vector<uint64_t> &t)
	{
		get_buckets(s, bkt.data(), n, K, false); // L-type: left to right from bucket starts
		for (I i = 0; i < n; ++i) {
			I j = SA[i] - 1;
			if (SA[i] > 0 && !is_s(t, j)) SA[bkt[s[j]]++] = j;
		}
		get_buckets(s, bkt.data(), n, K, true);  // S-type: right to left from bucket ends
		for (I i = n - 1; i >= 0; --i) {
			I j = SA[i] - 1;
			if (SA[i] > 0 && is_s(t, j)) SA[--bkt[s[j]]] = j;
		}
	}

	static void run(const S *s, I *SA, I n, I K)
	{
		std::vector<uint64_t> t(((size_t)n >> 6) + 1, 0);
		t[(n - 1) >> 6] |= 1ULL << ((n - 1) & 63);
		for (I i = n - 2; i >= 0; --i)
			if (s[i] < s[i + 1] || (s[i] == s[i + 1] && is_s(t, i + 1))) t[i >> 6] |= 1ULL << (i & 63);
		std::vector<I> bkt((size_t)K);
		// stage 1: sort the LMS substrings
		get_buckets(s, bkt.data(), n, K, true);
		for (I i = 0; i < n; ++i) SA[i] = -1;
		for (I i = 1; i < n; ++i) if (is_lms(t, i)) SA[--bkt[s[i]]] = i;
		induce(s, SA, n, K, bkt, t);
		// compact the sorted LMS suffixes into SA[0..n1)
		I n1 = 0;
		for (I i = 0; i < n; ++i) if (is_lms(t, SA[i])) SA[n1++] = SA[i];
		for (I i = n1; i < n; ++i) SA[i] = -1;
		// name the LMS substrings
		I name = 0, prev = -1;
		for (I i = 0; i < n1; ++i) {
			I pos = SA[i];
			bool diff = false;
			for (I d = 0; d < n; ++d) {
				if (prev == -1 || s[pos + d] != s[prev + d] || is_s(t, pos + d) != is_s(t, prev + d)) { diff = true; break; }
				if (d > 0 && (is_lms(t, pos + d) || is_lms(t, prev + d))) break;
			}
			if (diff) { ++name; prev = pos; }
			SA[n1 + (pos >> 1)] = name - 1;
		}
		for (I i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
		// stage 2: solve the reduced problem
		I *SA1 = SA, *s1 = SA + n - n1;
		if (name < n1) SaIs<I, I>::run(s1, SA1, n1, name);
		else for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
		// stage 3: induce the final order from the sorted LMS suffixes
		get_buckets(s, bkt.data(), n, K, true);
		for (I i = 1, j = 0; i < n; ++i) if (is_lms(t, i)) s1[j++] = i;
		for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
		for (I i = n1; i < n; ++i) SA[i] = -1;
		for (I i = n1 - 1; i >= 0; --i) {
			I j = SA[i];
			SA[i] = -1;
			SA[--bkt[s[j]]] = j;
		}
		induce(s, SA, n, K, bkt, t);
	}
};

struct BuildStats { int64_t l_pac = 0; int n_seqs = 0, n_holes = 0; double secs_pack = 0, secs_sa = 0, secs_write = 0; };

static const unsigned char kNt4[256] = { // nst_nt4_table (bntseq.c:47-64): A/C/G/T in either case -> 0..3, '-' -> 5, the rest -> 4
#define R16 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4
	R16, R16, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 5, 4, 4, R16,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
	R16, R16, R16, R16, R16, R16, R16, R16
#undef R16
};

// The step that turns the packed forward strand into <prefix>.bwt and <prefix>.sa: on the host below (induced sorting, 32-bit:
// 2 * l_pac < 2^31), on the device in hip_index_build.h (any size the HBM holds).  Returns "" or an error message.
typedef std::string (*BwtSaFn)(const uint8_t *pac, size_t pac_bytes, int64_t l_pac, const uint64_t cnt_fwd[4], const std::string &prefix);

inline std::string build_bwt_sa_host(const uint8_t *pac, size_t, int64_t l_pac, const uint64_t cnt_fwd[4], const std::string &prefix)
{
	// text = forward + reverse complement (bntseq.c:299-305), symbols shifted by one to make room for the sentinel
	const int64_t n = 2 * l_pac;
	if (n + 1 >= ((int64_t)1 << 31)) return "genome too large for the host suffix sorter (2*l_pac must stay below 2^31): the device builder of libarachne_amd.so has no such limit, but needs an MI355X";
	std::vector<uint8_t> text((size_t)n + 1);
	for (int64_t i = 0; i < l_pac; ++i) { const int c = pac[i >> 2] >> ((~i & 3) << 1) & 3; text[i] = c + 1; text[n - 1 - i] = (3 - c) + 1; }
	text[n] = 0;
	std::vector<int32_t> SA((size_t)n + 1);
	SaIs<uint8_t, int32_t>::run(text.data(), SA.data(), (int32_t)(n + 1), 5);
	// BWT without the $ row, primary = rank of suffix 0 (is.c:208-223 is_bwt)
	uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
	for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + cnt_fwd[c] + cnt_fwd[3 - c];
	std::vector<uint8_t> bw((size_t)n);
	{
		int64_t k = 0;
		for (int64_t i = 0; i <= n; ++i) {
			if (SA[i] == 0) primary = (uint64_t)i;
			else bw[k++] = text[SA[i] - 1] - 1;
		}
	}
	// interleave: every 128 symbols 4 x u64 running counts, then 8 x u32 of packed symbols (bwtindex.c:151-173)
	{
		const uint64_t n_occ = ((uint64_t)n + 127) / 128 + 1;
		const uint64_t bwt_size = (((uint64_t)n + 15) >> 4) + n_occ * 8;
		std::vector<uint32_t> out(bwt_size, 0);
		uint64_t c[4] = {0, 0, 0, 0}, k = 0;
		for (int64_t i = 0; i < n; ++i) {
			if ((i & 127) == 0) { memcpy(&out[k], c, 32); k += 8; }
			if ((i & 15) == 0) ++k;
			out[k - 1] |= (uint32_t)bw[i] << ((15 - (i & 15)) << 1);
			++c[bw[i]];
		}
		memcpy(&out[k], c, 32);
		FILE *o = fopen((prefix + ".bwt").c_str(), "wb");
		if (!o) return "cannot write " + prefix + ".bwt";
		fwrite(&primary, 8, 1, o); fwrite(L2 + 1, 8, 4, o); fwrite(out.data(), 4, out.size(), o);
		fclose(o);
	}
	// sampled SA every 32 rows; sa[0] (= seq_len in the unsampled array) is not stored (bwt.c:397-407)
	{
		const uint64_t intv = 32, n_sa = ((uint64_t)n + intv) / intv, seq_len = (uint64_t)n;
		std::vector<uint64_t> sa(n_sa);
		for (uint64_t i = 0; i < n_sa; ++i) sa[i] = (uint64_t)SA[i * intv];
		FILE *o = fopen((prefix + ".sa").c_str(), "wb");
		if (!o) return "cannot write " + prefix + ".sa";
		fwrite(&primary, 8, 1, o); fwrite(L2 + 1, 8, 4, o); fwrite(&intv, 8, 1, o); fwrite(&seq_len, 8, 1, o);
		fwrite(sa.data() + 1, 8, n_sa - 1, o);
		fclose(o);
	}
	return "";
}

// Build all index files from a plain-text FASTA.  Returns "" or an error message.
inline std::string build_index(const std::string &fasta, const std::string &prefix, BwtSaFn bwt_sa = build_bwt_sa_host, BuildStats *stats = nullptr)
{
	FILE *f = fopen(fasta.c_str(), "rb");
	if (!f) return "cannot open " + fasta;
	struct Ann { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
	struct Amb { int64_t offset; int32_t len; char amb; };
	std::vector<Ann> anns;
	std::vector<Amb> ambs;
	std::vector<uint8_t> pac; // forward strand, 4 bases per byte (first base in the top bits), N already randomised
	int64_t l_pac = 0;
	uint64_t cnt_fwd[4] = {0, 0, 0, 0};
	Rand48 rng(11);
	{
		fseek(f, 0, SEEK_END);
		const long fsz = ftell(f);
		fseek(f, 0, SEEK_SET);
		if (fsz > 0) pac.reserve((size_t)fsz / 4 + 64);
		std::vector<char> buf(1 << 22);
		int lasts = 0;
		bool in_hdr = false;
		std::string hdr;
		uint8_t acc = 0; // the byte being filled
		auto start_seq = [&](const std::string &h) {
			Ann a; size_t i = 0;
			while (i < h.size() && h[i] != ' ' && h[i] != '\t') ++i;
			a.name = h.substr(0, i);
			while (i < h.size() && (h[i] == ' ' || h[i] == '\t')) ++i;
			a.anno = i < h.size() ? h.substr(i) : "(null)";
			a.offset = l_pac; a.len = 0; a.n_ambs = 0;
			anns.push_back(a);
			lasts = 0;
		};
		size_t got;
		while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
			for (size_t k = 0; k < got; ++k) {
				const char ch = buf[k];
				if (in_hdr) {
					if (ch == '\n') { in_hdr = false; while (!hdr.empty() && hdr.back() == '\r') hdr.pop_back(); start_seq(hdr); hdr.clear(); }
					else hdr.push_back(ch);
					continue;
				}
				if (ch == '>') { in_hdr = true; continue; }
				if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
				if (anns.empty()) { fclose(f); return "FASTA does not start with '>'"; }
				Ann &a = anns.back();
				int c = kNt4[(unsigned char)ch];
				if (c >= 4) { // runs of the same ambiguous character form one hole (bntseq.c:245-259)
					if (lasts == ch) ++ambs.back().len;
					else { ambs.push_back(Amb{a.offset + a.len, 1, ch}); ++a.n_ambs; }
					c = rng.lrand() & 3;
				}
				lasts = ch;
				acc |= (uint8_t)(c << ((~l_pac & 3) << 1));
				if ((l_pac & 3) == 3) { pac.push_back(acc); acc = 0; }
				++cnt_fwd[c];
				++l_pac;
				++a.len;
			}
		}
		if (l_pac & 3) pac.push_back(acc);
		fclose(f);
	}
	if (l_pac == 0) return "empty FASTA";
	if (stats) { stats->l_pac = l_pac; stats->n_seqs = (int)anns.size(); stats->n_holes = (int)ambs.size(); }
	// .pac: forward strand only, (l_pac/4 + 1 + 1) bytes; the last byte is l_pac % 4 (bntseq.c:306-319)
	{
		FILE *o = fopen((prefix + ".pac").c_str(), "wb");
		if (!o) return "cannot write " + prefix + ".pac";
		fwrite(pac.data(), 1, pac.size(), o);
		uint8_t ct = 0;
		if (l_pac % 4 == 0) fwrite(&ct, 1, 1, o);
		ct = (uint8_t)(l_pac % 4);
		fwrite(&ct, 1, 1, o);
		fclose(o);
	}
	{
		FILE *o = fopen((prefix + ".ann").c_str(), "w");
		if (!o) return "cannot write " + prefix + ".ann";
		fprintf(o, "%lld %d %u\n", (long long)l_pac, (int)anns.size(), 11u);
		for (auto &a : anns) {
			fprintf(o, "%d %s", 0, a.name.c_str());
			if (!a.anno.empty()) fprintf(o, " %s\n", a.anno.c_str()); else fprintf(o, "\n");
			fprintf(o, "%lld %d %d\n", (long long)a.offset, a.len, a.n_ambs);
		}
		fclose(o);
		o = fopen((prefix + ".amb").c_str(), "w");
		if (!o) return "cannot write " + prefix + ".amb";
		fprintf(o, "%lld %d %u\n", (long long)l_pac, (int)anns.size(), (unsigned)ambs.size());
		for (auto &h : ambs) fprintf(o, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
		fclose(o);
	}
	pac.resize(pac.size() + 16, 0); // readers below may look a few bytes past the end
	return bwt_sa(pac.data(), pac.size() - 16, l_pac, cnt_fwd, prefix);
}

} // namespace arx
