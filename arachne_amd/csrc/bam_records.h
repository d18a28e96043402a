// bam_records.h -- from the path's results to BAM records: the part of DumpToBams / AppendBam (src/aligner/bamwriter.go:283-568, 635-658)
// that turns the placed candidate of every read into a record, for a whole super-batch at a time, on host threads.  Host code of
// libarachne_amd.so (C ABI: arx_recbuf_* in include/arachne_amd.h); the sink that takes the records is bam_sink.h.
//
// One record per read: its ACTIVE candidate (what DoDumpToBam writes as `primary`, bamwriter.go:635-658), with
//   flags      paired 0x1, proper 0x2, unmapped 0x4, mate unmapped 0x8, reverse 0x10, mate reverse 0x20, first / second 0x40 / 0x80,
//              duplicate 0x400, as AppendBam sets them (:286-366); "unmapped" is the reference's rule: not proper and score - 17 < 19
//              (:287-290, aligner.go:140-145) or the placeholder of a read without hits (pos -1)
//   pos / mapq / mate / template length  as :287-346 (the candidate's pos is 0-based already; reverse-strand candidates carry the
//              swapped pos/aend of aligner.go:1577-1582, so TempLen reads the same fields the reference reads)
//   CIGAR      BWA's op codes MIDSH -> BAM's M I D S H (fixCigar's table, :248-276); read and qualities reversed for reverse-strand
//              records (:372-375; reverseComp / reverseQual)
//   aux        RG:Z (the R1 header's last field, reader.go:144-153), AS:i (score), XM:Z:0, AM:Z:0|1, XT:i:0 (:398-458 for a read without
//              mapq_data), and for a unique barcode set whose barcode holds a '-' BX:Z + VX:i:1 (:555-559).
// Left to the caller (documented, not emitted): the split / supplementary records and their SA / XS / XC / AC tags (they come from
// arx_split + the mismatch lists of arx_batch_post_fetch), the debug tags, DM (molecule_difference).
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/arachne_amd.h"

namespace arx {

struct RecBuf {
	std::vector<int64_t> name_off, cigar_off, seq_off, aux_off;
	std::vector<char> names;
	std::vector<int32_t> flag, rid, pos, mate_rid, mate_pos, tlen;
	std::vector<uint8_t> mapq, seq, qual, aux;
	std::vector<uint32_t> cigars;
	std::vector<int32_t> act; // active candidate of every read

	static inline bool unmapped(const arx_cand &a) { return a.pos == -1 || (!a.is_proper && a.score - 17 < 19); }

	// sb: the super-batch the batch was created from; cand_off / cands: arx_batch_rfa_fetch; alns / cigars: arx_batch_fetch (cands[].reg indexes
	// them); post: arx_batch_post_fetch's per-candidate records or NULL (then no duplicate flags)
	bool build(const arx_super_batch &sb, const int32_t *cand_off, const arx_cand *cands, const arx_aln *alns, const uint32_t *cigs, const arx_cand_post *post,
	           int threads, arx_bam_batch *view, std::string &err)
	{
		const int64_t NP = sb.n_pairs, NR = 2 * NP;
		if (threads < 1) threads = 1;
		act.assign((size_t)NR, -1);
		name_off.assign((size_t)NR + 1, 0); cigar_off.assign((size_t)NR + 1, 0); seq_off.assign((size_t)NR + 1, 0); aux_off.assign((size_t)NR + 1, 0);
		flag.resize((size_t)NR); rid.resize((size_t)NR); pos.resize((size_t)NR); mate_rid.resize((size_t)NR); mate_pos.resize((size_t)NR); tlen.resize((size_t)NR); mapq.resize((size_t)NR);
		std::vector<int64_t> base_off((size_t)NR + 1, 0), pair_set((size_t)NP);
		for (int64_t r = 0; r < NR; ++r) base_off[(size_t)r + 1] = base_off[(size_t)r] + sb.lens[r];
		for (int s = 0; s < sb.n_sets; ++s) for (int64_t p = sb.set_pair_off[s]; p < sb.set_pair_off[s + 1]; ++p) pair_set[(size_t)p] = s;
		// which sets get BX / VX: attach_bx = unique_barcode (aligner.go:474, 499) and a '-' in the barcode (bamwriter.go:389, 555)
		std::vector<uint8_t> set_bx((size_t)sb.n_sets, 0);
		for (int s = 0; s < sb.n_sets; ++s) {
			const char *b = sb.barcodes + sb.barcode_off[s]; const int64_t bl = sb.barcode_off[s + 1] - sb.barcode_off[s];
			set_bx[(size_t)s] = sb.unique[s] && memchr(b, '-', (size_t)bl) != nullptr;
		}
		bool bad = false;
		auto par = [&](auto fn) {
			std::vector<std::thread> th;
			for (int t = 0; t < threads; ++t) th.emplace_back([&, t]() { const int64_t lo = NR * t / threads, hi = NR * (t + 1) / threads; for (int64_t r = lo; r < hi; ++r) fn(r); });
			for (auto &x : th) x.join();
		};
		// pass 1: the active candidate and the sizes of every record
		par([&](int64_t r) {
			int a = -1;
			for (int i = cand_off[r]; i < cand_off[r + 1]; ++i) if (cands[i].active) a = i; // exactly one per read
			if (a < 0) { bad = true; a = cand_off[r]; }
			act[(size_t)r] = a;
			const arx_cand &c = cands[a];
			const int64_t p = r >> 1; const int s = (int)pair_set[(size_t)p];
			name_off[(size_t)r + 1] = sb.name_off[p + 1] - sb.name_off[p];
			cigar_off[(size_t)r + 1] = c.reg >= 0 ? alns[c.reg].n_cigar : 0;
			seq_off[(size_t)r + 1] = sb.lens[r];
			const int64_t rgl = sb.rg_off[p + 1] - sb.rg_off[p];
			int64_t ax = (rgl > 0 ? 3 + rgl + 1 : 0) + 7 /* AS:i as int32 */ + 5 /* XM:Z:0 */ + 5 /* AM:Z:x */ + 4 /* XT:C:0 */;
			if (set_bx[(size_t)s]) ax += 3 + (sb.barcode_off[s + 1] - sb.barcode_off[s]) + 1 + 4 /* VX:C:1 */;
			aux_off[(size_t)r + 1] = ax;
		});
		if (bad) { err = "a read without an active candidate: arx_batch_rfa must have run on this batch"; return false; }
		for (int64_t r = 0; r < NR; ++r) { name_off[(size_t)r + 1] += name_off[(size_t)r]; cigar_off[(size_t)r + 1] += cigar_off[(size_t)r]; seq_off[(size_t)r + 1] += seq_off[(size_t)r]; aux_off[(size_t)r + 1] += aux_off[(size_t)r]; }
		names.resize((size_t)name_off[(size_t)NR] + 1); cigars.resize((size_t)cigar_off[(size_t)NR] + 1); seq.resize((size_t)seq_off[(size_t)NR] + 1); qual.resize((size_t)seq_off[(size_t)NR] + 1);
		aux.resize((size_t)aux_off[(size_t)NR] + 1);
		// pass 2: fill
		static const uint32_t op_table[5] = {0, 1, 2, 4, 5}; // fixCigar (bamwriter.go:248-254): BWA's MIDSH -> BAM's M I D S H
		static const char comp[5] = {'T', 'G', 'C', 'A', 'N'}, fwd[5] = {'A', 'C', 'G', 'T', 'N'};
		par([&](int64_t r) {
			const arx_cand &c = cands[act[(size_t)r]], &m = cands[act[(size_t)(r ^ 1)]];
			const int64_t p = r >> 1; const int s = (int)pair_set[(size_t)p];
			const bool un = unmapped(c), mun = unmapped(m);
			int32_t fl = 0x1 | ((r & 1) ? 0x80 : 0x40);
			if (c.is_proper) fl |= 0x2;
			if (mun) fl |= 0x8; else if (m.reversed) fl |= 0x20;
			if (post && post[act[(size_t)r]].duplicate) fl |= 0x400;
			if (un) fl |= 0x4;
			if (c.reversed) fl |= 0x10;
			flag[(size_t)r] = fl;
			rid[(size_t)r] = un ? -1 : c.rid; pos[(size_t)r] = un ? -1 : (int32_t)c.pos; mapq[(size_t)r] = un ? 0 : (uint8_t)(c.mapq < 0 ? 0 : (c.mapq > 255 ? 255 : c.mapq));
			mate_rid[(size_t)r] = mun ? -1 : m.rid; mate_pos[(size_t)r] = mun ? -1 : (int32_t)m.pos;
			int32_t tl = 0;
			if (m.pos != -1 && c.rid == m.rid && (c.is_proper || m.score - 17 >= 19)) tl = c.reversed ? -(int32_t)(c.aend - m.pos) : (int32_t)(m.aend - c.pos); // bamwriter.go:329-343
			tlen[(size_t)r] = tl;
			memcpy(names.data() + name_off[(size_t)r], sb.names + sb.name_off[p], (size_t)(sb.name_off[p + 1] - sb.name_off[p]));
			if (c.reg >= 0) {
				const arx_aln &al = alns[c.reg];
				uint32_t *dst = cigars.data() + cigar_off[(size_t)r];
				for (int k = 0; k < al.n_cigar; ++k) { const uint32_t w = cigs[al.cigar_off + k]; const uint32_t op = w & 15u; dst[k] = (w & ~15u) | (op < 5 ? op_table[op] : op); }
			}
			const int L = sb.lens[r];
			const uint8_t *b = sb.bases + base_off[(size_t)r]; const char *q = sb.quals + base_off[(size_t)r];
			uint8_t *so = seq.data() + seq_off[(size_t)r], *qo = qual.data() + seq_off[(size_t)r];
			if (c.reversed) for (int k = 0; k < L; ++k) { const uint8_t x = b[L - 1 - k]; so[k] = (uint8_t)comp[x > 4 ? 4 : x]; qo[k] = (uint8_t)q[L - 1 - k]; }
			else for (int k = 0; k < L; ++k) { const uint8_t x = b[k]; so[k] = (uint8_t)fwd[x > 4 ? 4 : x]; qo[k] = (uint8_t)q[k]; }
			uint8_t *a = aux.data() + aux_off[(size_t)r];
			const int64_t rgl = sb.rg_off[p + 1] - sb.rg_off[p];
			if (rgl > 0) { *a++ = 'R'; *a++ = 'G'; *a++ = 'Z'; memcpy(a, sb.rgs + sb.rg_off[p], (size_t)rgl); a += rgl; *a++ = 0; }
			*a++ = 'A'; *a++ = 'S'; *a++ = 'i'; { const int32_t v = c.score; memcpy(a, &v, 4); a += 4; }
			*a++ = 'X'; *a++ = 'M'; *a++ = 'Z'; *a++ = '0'; *a++ = 0;
			*a++ = 'A'; *a++ = 'M'; *a++ = 'Z'; *a++ = c.active_molecule ? '1' : '0'; *a++ = 0;
			*a++ = 'X'; *a++ = 'T'; *a++ = 'C'; *a++ = 0;
			if (set_bx[(size_t)s]) {
				const int64_t bl = sb.barcode_off[s + 1] - sb.barcode_off[s];
				*a++ = 'B'; *a++ = 'X'; *a++ = 'Z'; memcpy(a, sb.barcodes + sb.barcode_off[s], (size_t)bl); a += bl; *a++ = 0;
				*a++ = 'V'; *a++ = 'X'; *a++ = 'C'; *a++ = 1;
			}
		});
		view->n_records = NR;
		view->name_off = name_off.data(); view->names = names.data(); view->flag = flag.data(); view->rid = rid.data(); view->pos = pos.data(); view->mapq = mapq.data();
		view->mate_rid = mate_rid.data(); view->mate_pos = mate_pos.data(); view->tlen = tlen.data(); view->cigar_off = cigar_off.data(); view->cigars = cigars.data();
		view->seq_off = seq_off.data(); view->seq = seq.data(); view->qual = qual.data(); view->qual_offset = 33; view->aux_off = aux_off.data(); view->aux = aux.data();
		return true;
	}
};

} // namespace arx
