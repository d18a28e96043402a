// feeder.h -- host side in front of the path: the reference's paired FASTQ reader re-shaped to hand the
// device super-batches of whole barcode sets instead of one read pair per cgo call.
//   OpenFastQ / FastZipReader      /root/reference/src/fastqreader/reader.go:64-86, zipread.go:62 (gunzip pipe there, zlib here)
//   ParseHeader                    reader.go:95-123
//   ReadOneLine                    reader.go:128-190 -- as committed it indexes past its array and takes the '+' line for the
//                                  sequence; implemented with the intended semantics: header, sequence, '+', quality
//   ReadBarcodeSet                 reader.go:209-300 -- kept rule for rule: a set ends before the first record of another barcode,
//                                  after 30000 records (flagged not unique), and while it continues the previous set's barcode after
//                                  201 records (flagged not unique); the tail of such a barcode ends at the barcode change and is
//                                  flagged unique again, as in the reference
//   worthRunningRFA                /root/reference/src/aligner/aligner.go:1018-1030
// Plain C++ (no device code); compiled into libarachne_amd.so and, for the CPU tests, into the host test double.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>
#include "../../include/arachne_amd.h"

namespace arx {

class LineSource { // bufio.Reader.ReadString('\n') over a (possibly gzip-compressed) file
public:
	bool open(const char *path)
	{
		f_ = gzopen(path, "rb");
		if (!f_) return false;
		gzbuffer(f_, 1 << 20);
		buf_.resize(4 << 20);
		return true;
	}
	~LineSource() { if (f_) gzclose(f_); }
	// 1: a line ending in '\n' (returned without it in [b, e)); 0: end of input -- a last line without '\n' counts as end of
	// input, which is what ReadString's (data, io.EOF) amounts to in ReadOneLine; -1: read error
	int next(const char *&b, const char *&e)
	{
		for (;;) {
			const char *nl = (const char *)memchr(buf_.data() + pos_, '\n', end_ - pos_);
			if (nl) { b = buf_.data() + pos_; e = nl; pos_ = (size_t)(nl - buf_.data()) + 1; return 1; }
			if (eof_) return err_ ? -1 : 0;
			if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
			if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
			const int got = gzread(f_, buf_.data() + end_, (unsigned)(buf_.size() - end_));
			if (got < 0) { eof_ = true; err_ = true; }
			else if (got == 0) eof_ = true;
			else end_ += (size_t)got;
		}
	}
private:
	gzFile f_ = nullptr;
	std::vector<char> buf_;
	size_t pos_ = 0, end_ = 0;
	bool eof_ = false, err_ = false;
};

struct FastqRecord { std::string info, rg, barcode, s1, q1, s2, q2; bool valid = false; };

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// leftmost match of `TAG(\S+)\s` in [b, e); the line's own '\n' (at e) counts as the closing whitespace
inline bool find_tag_value(const char *b, const char *e, const char *tag, const char *&vb, const char *&ve)
{
	const size_t tl = strlen(tag);
	for (const char *p = b; p + tl < e + 1; ++p) {
		if (memcmp(p, tag, tl) != 0) continue;
		const char *v = p + tl;
		if (v >= e || is_space(*v)) continue;
		const char *w = v;
		while (w < e && !is_space(*w)) ++w;
		vb = v; ve = w;
		return true;
	}
	return false;
}

// ParseHeader + the ReadGroupId rule of ReadOneLine, on the R1 header without its '@' and '\n'
inline void parse_header(const char *b, const char *e, FastqRecord &r)
{
	r.info.clear(); r.barcode.clear(); r.rg.clear(); r.valid = false;
	// fields: first and last whitespace-separated token
	const char *p = b;
	while (p < e && is_space(*p)) ++p;
	const char *f0 = p;
	while (p < e && !is_space(*p)) ++p;
	const char *f0e = p;
	int n_fields = f0e > f0;
	const char *lb = f0, *le = f0e;
	while (p < e) {
		while (p < e && is_space(*p)) ++p;
		if (p >= e) break;
		lb = p;
		while (p < e && !is_space(*p)) ++p;
		le = p; ++n_fields;
	}
	if (n_fields >= 2) r.rg.assign(lb, le);
	const char *vb, *ve;
	if (!find_tag_value(b, e, "BX:Z:", vb, ve)) return; // no barcode: empty header and barcode, not valid (reader.go:110-112)
	r.barcode.assign(vb, ve);
	if (f0e - f0 >= 2) r.info.assign(f0, f0e - 2); // the id without its trailing "/1" (the reference would panic on a shorter one)
	// VX:i:([01])\s
	for (const char *q = b; q + 7 <= e + 1; ++q) {
		if (memcmp(q, "VX:i:", 5) != 0 || q + 5 >= e) continue;
		const char d = q[5];
		if ((d == '0' || d == '1') && (q + 6 == e || is_space(q[6]))) { r.valid = d == '1'; break; }
	}
}

class Feeder {
public:
	std::string error;
	bool open(const char *r1, const char *r2)
	{
		if (!src1_.open(r1)) { error = std::string("cannot open ") + r1; return false; }
		if (!src2_.open(r2)) { error = std::string("cannot open ") + r2; return false; }
		return true;
	}

	// 1 record, 0 end of input, -1 read error
	int read_one(FastqRecord &r)
	{
		const char *b1, *e1, *b2, *e2;
		for (;;) { // search for the next start-of-record, both files in lockstep
			++line_;
			int a = src1_.next(b1, e1); if (a <= 0) return a;
			a = src2_.next(b2, e2); if (a <= 0) return a;
			if (e1 > b1 && *b1 == '@') { parse_header(b1 + 1, e1, r); break; }
			++bad_lines_;
		}
		for (int i = 0; i < 3; ++i) { // sequence, '+', quality
			int a = src1_.next(b1, e1); if (a <= 0) return a;
			a = src2_.next(b2, e2); if (a <= 0) return a;
			if (i == 0) { r.s1.assign(b1, e1); r.s2.assign(b2, e2); }
			else if (i == 2) { r.q1.assign(b1, e1); r.q2.assign(b2, e2); }
		}
		return 1;
	}

	// ReadBarcodeSet: the set's records are out[0 .. n) (the vector only grows: its strings keep their storage from set to set);
	// returns 1 (a set, `unique` and `n` set), 0 (end of input), -1 (read error)
	int read_barcode_set(std::vector<FastqRecord> &out, size_t &n, bool &unique)
	{
		n = 0;
		unique = false;
		if (deferred_) return deferred_ == 1 ? 0 : -1;
		bool new_barcode = false;
		size_t index = 0;
		auto slot = [&](size_t i) -> FastqRecord & { if (i == out.size()) out.emplace_back(); return out[i]; };
		if (have_pending_) { std::swap(slot(0), pending_); have_pending_ = false; index = 1; }
		n = index;
		for (; index < 30000; ++index) {
			const int rc = read_one(slot(index));
			if (rc <= 0) { // the record the reference leaves half-filled and then cuts off (or, for a read error, hands on) is not counted
				if (index == 0) { deferred_ = rc == 0 ? 1 : 2; return rc == 0 ? 0 : -1; }
				deferred_ = rc == 0 ? 1 : 2;
				break;
			}
			if (out[0].barcode != out[index].barcode) {
				std::swap(pending_, out[index]); have_pending_ = true;
				new_barcode = true;
				break;
			}
			n = index + 1;
			if (have_last_ && out[0].barcode == last_barcode_ && index >= 200) break; // "abnormal break": the 201st record stays in the set
		}
		if (n) { last_barcode_ = out[0].barcode; have_last_ = true; }
		unique = new_barcode || deferred_ == 1;
		return 1;
	}

	// whole sets until at least target_pairs pairs are held (always at least one set); returns the number of sets
	int next(int64_t target_pairs, arx_super_batch *o)
	{
		set_off_.assign(1, 0); unique_.clear(); do_rfa_.clear(); bases_.clear(); quals_.clear(); lens_.clear(); valid_.clear();
		name_off_.assign(1, 0); names_.clear(); rg_off_.assign(1, 0); rgs_.clear(); bc_off_.assign(1, 0); bcs_.clear();
		int64_t pairs = 0;
		int rc = 1;
		while (pairs < target_pairs || set_off_.size() == 1) {
			bool unique;
			size_t n_set;
			rc = read_barcode_set(set_, n_set, unique);
			if (rc <= 0) break;
			for (size_t k = 0; k < n_set; ++k) {
				const FastqRecord &r = set_[k];
				append_read(r.s1, r.q1); append_read(r.s2, r.q2);
				names_ += r.info; name_off_.push_back((int64_t)names_.size());
				rgs_ += r.rg; rg_off_.push_back((int64_t)rgs_.size());
				valid_.push_back(r.valid);
			}
			pairs += (int64_t)n_set;
			set_off_.push_back(pairs);
			unique_.push_back(unique);
			const std::string &bc = set_[0].barcode;
			do_rfa_.push_back(unique && bc.find('-') != std::string::npos && n_set >= 5); // worthRunningRFA
			bcs_ += bc; bc_off_.push_back((int64_t)bcs_.size());
		}
		if (rc < 0 && set_off_.size() == 1) { error = "read error in the FASTQ input"; return -1; }
		memset(o, 0, sizeof *o);
		o->n_sets = (int32_t)set_off_.size() - 1; o->n_pairs = pairs; o->bad_lines = bad_lines_;
		o->set_pair_off = set_off_.data(); o->unique = unique_.data(); o->do_rfa = do_rfa_.data();
		o->bases = bases_.data(); o->quals = quals_.data(); o->lens = lens_.data(); o->valid = valid_.data();
		o->name_off = name_off_.data(); o->names = names_.data(); o->rg_off = rg_off_.data(); o->rgs = rgs_.data();
		o->barcode_off = bc_off_.data(); o->barcodes = bcs_.data();
		return o->n_sets;
	}

private:
	void append_read(const std::string &s, const std::string &q)
	{
		static const Nt4 nt4;
		const size_t at = bases_.size();
		bases_.resize(at + s.size());
		for (size_t i = 0; i < s.size(); ++i) bases_[at + i] = nt4.t[(uint8_t)s[i]];
		// one quality byte per base: the layout of `bases`; a quality line of another length is cut or padded with '!'
		quals_.resize(at + s.size(), '!');
		memcpy(quals_.data() + at, q.data(), q.size() < s.size() ? q.size() : s.size());
		lens_.push_back((int32_t)s.size());
	}
	struct Nt4 { uint8_t t[256]; Nt4() { memset(t, 4, 256); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; } }; // nst_nt4_table (bntseq.c:47)

	LineSource src1_, src2_;
	int64_t line_ = 0, bad_lines_ = 0;
	FastqRecord pending_; bool have_pending_ = false;
	std::string last_barcode_; bool have_last_ = false;
	int deferred_ = 0; // 1: end of input seen, 2: read error seen
	std::vector<FastqRecord> set_;
	std::vector<int64_t> set_off_, name_off_, rg_off_, bc_off_;
	std::vector<uint8_t> unique_, do_rfa_, bases_, valid_;
	std::vector<char> quals_;
	std::vector<int32_t> lens_;
	std::string names_, rgs_, bcs_;
};

} // namespace arx
