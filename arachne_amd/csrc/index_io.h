// index_io.h -- host-side reader for the index files `bwa index` writes (<prefix>.bwt/.sa/.pac/.ann/.amb/.alt).
// Formats: bwt.c:421-462 (bwt_restore_bwt/sa), bntseq.c:98-206 (bns_restore), bwa.c:262-289 (bwa_idx_load_from_disk).
#pragma once
#include <stdint.h>
#include <stdio.h>
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

inline bool read_all(const std::string &fn, std::vector<uint8_t> &buf)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf.resize(sz);
	bool ok = sz == 0 || fread(buf.data(), 1, sz, f) == (size_t)sz;
	fclose(f);
	return ok;
}

// returns "" on success, else a message
inline std::string load_index(const std::string &prefix, HostIndex &ix)
{
	std::vector<uint8_t> raw;
	if (!read_all(prefix + ".bwt", raw) || raw.size() < 40) return "cannot read " + prefix + ".bwt";
	memcpy(&ix.primary, raw.data(), 8);
	memcpy(&ix.L2[1], raw.data() + 8, 32);
	ix.L2[0] = 0;
	ix.seq_len = ix.L2[4];
	size_t n_words = (raw.size() - 40) / 4;
	ix.bwt.assign(((n_words + 15) / 16 + 1) * 16, 0);
	memcpy(ix.bwt.data(), raw.data() + 40, n_words * 4);
	if (!read_all(prefix + ".sa", raw) || raw.size() < 56) return "cannot read " + prefix + ".sa";
	uint64_t prim, sintv, slen;
	memcpy(&prim, raw.data(), 8); memcpy(&sintv, raw.data() + 40, 8); memcpy(&slen, raw.data() + 48, 8);
	if (prim != ix.primary || slen != ix.seq_len) return "SA-BWT inconsistency in " + prefix + ".sa";
	if (sintv == 0 || (sintv & (sintv - 1))) return "SA interval is not a power of two";
	ix.sa_intv = (int)sintv;
	uint64_t n_sa = (ix.seq_len + sintv) / sintv;
	if (raw.size() < 56 + (n_sa - 1) * 8) return "truncated " + prefix + ".sa";
	ix.sa.assign(n_sa, 0);
	ix.sa[0] = (uint64_t)-1;
	memcpy(ix.sa.data() + 1, raw.data() + 56, (n_sa - 1) * 8);
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
	if (!read_all(prefix + ".pac", raw) || (int64_t)raw.size() < ix.l_pac / 4 + 1) return "cannot read " + prefix + ".pac";
	ix.pac.assign(raw.begin(), raw.begin() + ix.l_pac / 4 + 1);
	ix.pac.resize(ix.pac.size() + 16, 0);
	return "";
}

} // namespace arx
