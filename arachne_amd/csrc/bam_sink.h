// bam_sink.h -- the BAM sink behind the path, host code of libarachne_amd.so.
//
// What it replaces: the reference funnels every record of every barcode through ONE goroutine (BamThread, bamwriter.go:615-627) that
// builds a biogo sam.Record per alignment and writes it twice (AppendBams :279-282: the barcode-sorted BAM and a position bucket), each
// bam.Writer compressing with two BGZF goroutines (bam.NewWriter(file, h, 2), :118).  At a few hundred thousand records per second that
// caps the program far below what the GPU path delivers.  Here a writer takes records in BATCHES as flat arrays (what AppendBam
// computes per alignment -- name, flags, reference, position, MAPQ, CIGAR, mate, template length, bases, qualities, aux bytes -- stays
// the caller's business), encodes them into the BAM byte stream in parallel (record sizes -> prefix sums -> every thread fills its
// slice), cuts the stream into BGZF blocks and deflates the blocks on `threads` host threads; blocks reach the file in order.
//
// Format: SAM/BAM specification v1 (magic BAM\1, header text, references, records with bin from reg2bin, 4-bit bases, BGZF blocks of at
// most 65280 input bytes with the BC extra field, the 28-byte EOF block).  The bytes of the DEFLATE streams depend on the compressor
// (biogo/hts v1.4.5 uses Go's compress/gzip, this uses zlib): "byte-identical to the reference's file" is not a property any BAM writer
// can be held to; what is checked (tests/test_bam_sink.py) is that the decompressed stream is exactly the specified encoding of the
// records handed in.  The reference's library is absent from /root/reference (go.mod:5), so that is the pin: parity unpinned by it.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/arachne_amd.h"

namespace arx {

struct BamSink {
	FILE *f = nullptr;
	int threads = 1, level = 6;
	std::string error;
	std::vector<uint8_t> pending;         // uncompressed bytes not yet cut into a block
	int64_t n_records = 0, n_blocks = 0, bytes_in = 0, bytes_out = 0;
	static constexpr size_t BLOCK_IN = 0xff00; // BGZF: at most 64 KiB per block after compression; 65280 input bytes always fit

	static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
	static void w32(uint8_t *p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); p[2] = (uint8_t)(x >> 16); p[3] = (uint8_t)(x >> 24); }
	static void w16(uint8_t *p, uint32_t x) { p[0] = (uint8_t)x; p[1] = (uint8_t)(x >> 8); }

	bool open(const char *path, int n_contigs, const char *const *names, const int32_t *lens, const char *extra_header, int threads_, int level_)
	{
		f = fopen(path, "wb");
		if (!f) { error = std::string("cannot write ") + path; return false; }
		threads = threads_ > 0 ? threads_ : 1;
		level = level_ >= 0 && level_ <= 9 ? level_ : 6;
		std::string text = "@HD\tVN:1.6\tSO:unknown\n";
		for (int i = 0; i < n_contigs; ++i) text += std::string("@SQ\tSN:") + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
		if (extra_header) text += extra_header;
		std::vector<uint8_t> h;
		h.push_back('B'); h.push_back('A'); h.push_back('M'); h.push_back(1);
		put32(h, (uint32_t)text.size());
		h.insert(h.end(), text.begin(), text.end());
		put32(h, (uint32_t)n_contigs);
		for (int i = 0; i < n_contigs; ++i) {
			const size_t l = strlen(names[i]) + 1;
			put32(h, (uint32_t)l);
			h.insert(h.end(), names[i], names[i] + l);
			put32(h, (uint32_t)lens[i]);
		}
		pending = h;
		return flush(true); // the header ends its own block(s), as htslib and biogo write it
	}

	// reg2bin (SAM specification 5.3): the bin of [beg, end)
	static int reg2bin(int64_t beg, int64_t end)
	{
		--end;
		if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
		if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
		if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
		if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
		if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
		return 0;
	}
	static uint8_t base4(uint8_t c)
	{
		switch (c) { // "=ACMGRSVTWYHKDBN"
		case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3; case 'G': case 'g': return 4;
		case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7; case 'T': case 't': return 8; case 'W': case 'w': return 9;
		case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14;
		default: return 15;
		}
	}

	static size_t record_size(const arx_bam_batch &b, int64_t i)
	{
		const size_t l_name = (size_t)(b.name_off[i + 1] - b.name_off[i]) + 1, n_cig = (size_t)(b.cigar_off[i + 1] - b.cigar_off[i]);
		const size_t l_seq = (size_t)(b.seq_off[i + 1] - b.seq_off[i]), l_aux = (size_t)(b.aux_off[i + 1] - b.aux_off[i]);
		return 4 + 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq + l_aux;
	}
	static void encode(const arx_bam_batch &b, int64_t i, uint8_t *p)
	{
		const size_t sz = record_size(b, i);
		const size_t l_name = (size_t)(b.name_off[i + 1] - b.name_off[i]) + 1, n_cig = (size_t)(b.cigar_off[i + 1] - b.cigar_off[i]);
		const size_t l_seq = (size_t)(b.seq_off[i + 1] - b.seq_off[i]), l_aux = (size_t)(b.aux_off[i + 1] - b.aux_off[i]);
		const uint32_t *cg = b.cigars + b.cigar_off[i];
		int64_t ref_len = 0;
		for (size_t k = 0; k < n_cig; ++k) { const uint32_t op = cg[k] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += cg[k] >> 4; }
		const int64_t pos = b.pos[i];
		const int bin = pos < 0 ? 4680 : reg2bin(pos, pos + (ref_len > 0 ? ref_len : 1)); // unmapped: reg2bin(-1, 0)
		w32(p, (uint32_t)(sz - 4));
		w32(p + 4, (uint32_t)b.rid[i]); w32(p + 8, (uint32_t)pos);
		p[12] = (uint8_t)l_name; p[13] = b.mapq[i]; w16(p + 14, (uint32_t)bin);
		w16(p + 16, (uint32_t)n_cig); w16(p + 18, (uint32_t)b.flag[i]);
		w32(p + 20, (uint32_t)l_seq);
		w32(p + 24, (uint32_t)b.mate_rid[i]); w32(p + 28, (uint32_t)b.mate_pos[i]); w32(p + 32, (uint32_t)b.tlen[i]);
		uint8_t *q = p + 36;
		memcpy(q, b.names + b.name_off[i], l_name - 1); q[l_name - 1] = 0; q += l_name;
		for (size_t k = 0; k < n_cig; ++k) w32(q + 4 * k, cg[k]);
		q += 4 * n_cig;
		const uint8_t *s = b.seq + b.seq_off[i];
		for (size_t k = 0; k + 1 < l_seq; k += 2) *q++ = (uint8_t)(base4(s[k]) << 4 | base4(s[k + 1]));
		if (l_seq & 1) *q++ = (uint8_t)(base4(s[l_seq - 1]) << 4);
		const uint8_t *ql = b.qual + b.seq_off[i];
		if (b.qual_offset == 255) memset(q, 0xff, l_seq); // no qualities
		else for (size_t k = 0; k < l_seq; ++k) q[k] = (uint8_t)(ql[k] - b.qual_offset);
		q += l_seq;
		if (l_aux) memcpy(q, b.aux + b.aux_off[i], l_aux);
	}

	template <class F> void parallel(size_t n, F fn)
	{
		const int T = (int)(n < (size_t)threads ? (n ? n : 1) : threads);
		if (T <= 1) { fn(0, n); return; }
		std::vector<std::thread> th;
		for (int t = 0; t < T; ++t) th.emplace_back([=]() { fn(n * t / T, n * (t + 1) / T); });
		for (auto &x : th) x.join();
	}

	bool write(const arx_bam_batch &b)
	{
		const int64_t n = b.n_records;
		for (int64_t i = 0; i < n; ++i) {
			const int64_t ln = b.name_off[i + 1] - b.name_off[i], nc = b.cigar_off[i + 1] - b.cigar_off[i];
			if (ln < 1 || ln > 254) { error = "read name of record " + std::to_string(i) + " must be 1..254 bytes"; return false; }
			if (nc < 0 || nc > 65535) { error = "record " + std::to_string(i) + " has more than 65535 CIGAR operations"; return false; }
		}
		std::vector<size_t> off((size_t)n + 1, 0);
		for (int64_t i = 0; i < n; ++i) off[i + 1] = off[i] + record_size(b, i);
		const size_t base = pending.size();
		pending.resize(base + off[n]);
		uint8_t *dst = pending.data() + base;
		parallel((size_t)n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) encode(b, (int64_t)i, dst + off[i]); });
		n_records += n;
		return flush(false);
	}

	// compresses every whole block of `pending` (all of it when `all`), writes them in order
	bool flush(bool all)
	{
		const size_t total = pending.size();
		const size_t nb = all ? (total + BLOCK_IN - 1) / BLOCK_IN : total / BLOCK_IN;
		if (nb == 0) return true;
		std::vector<std::vector<uint8_t> > out(nb);
		std::vector<int> ok(nb, 1);
		parallel(nb, [&](size_t lo, size_t hi) {
			for (size_t k = lo; k < hi; ++k) {
				const size_t b0 = k * BLOCK_IN, len = (b0 + BLOCK_IN <= total) ? BLOCK_IN : total - b0;
				ok[k] = deflate_block(pending.data() + b0, len, out[k]) ? 1 : 0;
			}
		});
		for (size_t k = 0; k < nb; ++k) {
			if (!ok[k]) { error = "deflate failed"; return false; }
			if (fwrite(out[k].data(), 1, out[k].size(), f) != out[k].size()) { error = "write failed"; return false; }
			bytes_out += (int64_t)out[k].size(); ++n_blocks;
		}
		const size_t used = all ? total : nb * BLOCK_IN;
		bytes_in += (int64_t)used;
		pending.erase(pending.begin(), pending.begin() + used);
		return true;
	}

	bool deflate_block(const uint8_t *src, size_t len, std::vector<uint8_t> &out)
	{
		out.resize(18 + compressBound((uLong)len) + 8 + 64);
		static const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
		memcpy(out.data(), hdr, 18);
		for (int lv = level;; lv = 0) { // a block that does not shrink is stored (cannot happen with 65280 input bytes and deflate's 5-byte stored overhead, kept for safety)
			z_stream zs;
			memset(&zs, 0, sizeof zs);
			if (deflateInit2(&zs, lv, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
			zs.next_in = (Bytef *)src; zs.avail_in = (uInt)len;
			zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 18 - 8);
			const int rc = deflate(&zs, Z_FINISH);
			const size_t clen = zs.total_out;
			deflateEnd(&zs);
			if (rc != Z_STREAM_END) return false;
			if (18 + clen + 8 <= 65536) {
				w16(out.data() + 16, (uint32_t)(18 + clen + 8 - 1)); // BSIZE = total block size - 1
				w32(out.data() + 18 + clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len));
				w32(out.data() + 18 + clen + 4, (uint32_t)len);
				out.resize(18 + clen + 8);
				return true;
			}
			if (lv == 0) return false;
		}
	}

	bool close()
	{
		bool ok = true;
		if (f) {
			ok = flush(true);
			static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
			if (fwrite(eof, 1, 28, f) != 28) { error = "write failed"; ok = false; }
			if (fclose(f) != 0) { error = "close failed"; ok = false; }
			f = nullptr;
		}
		return ok;
	}
	~BamSink() { if (f) fclose(f); }
};

} // namespace arx
