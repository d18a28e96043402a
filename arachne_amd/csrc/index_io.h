// index_io.h -- host-side reader for the index files `bwa index` writes (<prefix>.bwt/.sa/.pac/.ann/.amb/.alt).
// Formats: bwt.c:421-462 (bwt_restore_bwt/sa), bntseq.c:98-206 (bns_restore), bwa.c:262-289 (bwa_idx_load_from_disk).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace arx {

struct HostIndex {
	uint64_t primary = 0, seq_len = 0, L2[5] = {0, 0, 0, 0, 0};
	std::vector<uint32_t> bwt;    // interleaved Occ/BWT blocks, padded to a whole 64-byte block
	std::vector<uint64_t> sa;     // sa[0] = (uint64_t)-1
	int sa_intv = 32;
	int64_t l_pac = 0;
	std::vector<uint8_t> pac;
	std::vector<std::string> names;
	std::vector<int64_t> ann_off;
	std::vector<int32_t> ann_len, ann_alt;
};

inline int64_t file_size(FILE *f) { fseeko(f, 0, SEEK_END); int64_t sz = (int64_t)ftello(f); fseeko(f, 0, SEEK_SET); return sz; }

// returns "" on success, else a message
inline std::string load_index(const std::string &prefix, HostIndex &ix)
{
	{ // .bwt: u64 primary, 4 x u64 L2[1..4], then the interleaved Occ/BWT words (bwt.c:443-462), read straight into place
		FILE *fb = fopen((prefix + ".bwt").c_str(), "rb");
		if (!fb) return "cannot read " + prefix + ".bwt";
		const int64_t sz = file_size(fb);
		uint64_t head[5];
		if (sz < 40 || fread(head, 8, 5, fb) != 5) { fclose(fb); return "cannot read " + prefix + ".bwt"; }
		ix.primary = head[0];
		ix.L2[0] = 0;
		for (int c = 0; c < 4; ++c) ix.L2[c + 1] = head[c + 1];
		ix.seq_len = ix.L2[4];
		const size_t n_words = (size_t)(sz - 40) / 4;
		// a truncated or foreign file would send the device past the end of the table (bwt_size of bwtindex.c:151-173)
		const uint64_t want = ((ix.seq_len + 15) >> 4) + ((ix.seq_len + 127) / 128 + 1) * 8;
		if ((uint64_t)n_words != want || (sz - 40) % 4) { fclose(fb); return prefix + ".bwt does not hold the " + std::to_string(want) + " words its header announces"; }
		ix.bwt.assign(((n_words + 15) / 16 + 1) * 16, 0);
		const bool ok = fread(ix.bwt.data(), 4, n_words, fb) == n_words;
		fclose(fb);
		if (!ok) return "cannot read " + prefix + ".bwt";
	}
	{ // .sa: primary, 4 x u64 skipped, interval, seq_len, then sa[1..] (bwt.c:421-441)
		FILE *fs = fopen((prefix + ".sa").c_str(), "rb");
		if (!fs) return "cannot read " + prefix + ".sa";
		const int64_t sz = file_size(fs);
		uint64_t head[7];
		if (sz < 56 || fread(head, 8, 7, fs) != 7) { fclose(fs); return "cannot read " + prefix + ".sa"; }
		const uint64_t prim = head[0], sintv = head[5], slen = head[6];
		if (prim != ix.primary || slen != ix.seq_len) { fclose(fs); return "SA-BWT inconsistency in " + prefix + ".sa"; }
		if (sintv == 0 || (sintv & (sintv - 1))) { fclose(fs); return "SA interval is not a power of two"; }
		ix.sa_intv = (int)sintv;
		const uint64_t n_sa = (ix.seq_len + sintv) / sintv;
		if ((uint64_t)sz < 56 + (n_sa - 1) * 8) { fclose(fs); return "truncated " + prefix + ".sa"; }
		ix.sa.assign(n_sa, 0);
		ix.sa[0] = (uint64_t)-1;
		const bool ok = n_sa < 2 || fread(ix.sa.data() + 1, 8, n_sa - 1, fs) == n_sa - 1;
		fclose(fs);
		if (!ok) return "cannot read " + prefix + ".sa";
	}
	FILE *f = fopen((prefix + ".ann").c_str(), "r");
	if (!f) return "cannot read " + prefix + ".ann";
	long long lp; int ns; unsigned seed;
	if (fscanf(f, "%lld%d%u", &lp, &ns, &seed) != 3) { fclose(f); return "parse error in .ann"; }
	ix.l_pac = lp;
	for (int i = 0; i < ns; ++i) {
		unsigned gi; char name[8192]; int c; long long off; int len, namb;
		if (fscanf(f, "%u%8191s", &gi, name) != 2) { fclose(f); return "parse error in .ann"; }
		while ((c = fgetc(f)) != '\n' && c != EOF) {}
		if (fscanf(f, "%lld%d%d", &off, &len, &namb) != 3) { fclose(f); return "parse error in .ann"; }
		ix.names.push_back(name); ix.ann_off.push_back(off); ix.ann_len.push_back(len); ix.ann_alt.push_back(0);
	}
	fclose(f);
	if ((uint64_t)ix.l_pac * 2 != ix.seq_len) return "l_pac does not match the BWT length";
	if ((f = fopen((prefix + ".alt").c_str(), "r"))) { // first token of each non-@ line names an ALT contig
		char line[8192];
		while (fgets(line, sizeof line, f)) {
			if (line[0] == '@') continue;
			char *e = line;
			while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			for (size_t i = 0; i < ix.names.size(); ++i) if (ix.names[i] == line) ix.ann_alt[i] = 1;
		}
		fclose(f);
	}
	{
		FILE *fp = fopen((prefix + ".pac").c_str(), "rb");
		if (!fp) return "cannot read " + prefix + ".pac";
		const size_t want = (size_t)(ix.l_pac / 4 + 1);
		ix.pac.assign(want + 16, 0);
		const bool ok = fread(ix.pac.data(), 1, want, fp) == want;
		fclose(fp);
		if (!ok) return "cannot read " + prefix + ".pac";
	}
	return "";
}

} // namespace arx
