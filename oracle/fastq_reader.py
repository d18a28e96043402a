"""oracle/fastq_reader.py -- TEST INFRASTRUCTURE ONLY.

Plain-Python restatement of the reference's paired FASTQ reader, statement by statement:
  ParseHeader      /root/reference/src/fastqreader/reader.go:95-123
  ReadOneLine      reader.go:128-190   (intended 4-line semantics: as committed the record loop indexes past its array and takes the
                                        '+' line for the sequence; header handling is followed as written)
  ReadBarcodeSet   reader.go:209-300
  worthRunningRFA  /root/reference/src/aligner/aligner.go:1018-1030
PARITY UNPINNED BY THE REFERENCE: it holds no tests or vectors for the reader and the Go tree cannot be built here; this file is pinned
by hand-derived cases in tests/test_feeder.py.  Only tests/ may import it.
"""
import io
import re

BX_RE = re.compile(r"BX:Z:(\S+)\s", re.ASCII)
VX_RE = re.compile(r"VX:i:([01])\s", re.ASCII)
EOF = "EOF"


def parse_header(seq_id):
    """reader.go:95-123; seq_id = the R1 header line without '@', with its newline"""
    _id = seq_id.split()[0]
    header = _id[:len(_id) - 2]
    m = BX_RE.search(seq_id)
    if m:
        barcode = m.group(1)
    else:
        return "", "", False
    valid = False
    m = VX_RE.search(seq_id)
    if m:
        valid = m.group(1) != "0"
    return header, barcode, valid


class Reader:
    def __init__(self, r1_text, r2_text):
        self.r1, self.r2 = io.StringIO(r1_text, newline="\n"), io.StringIO(r2_text, newline="\n")
        self.last_barcode = None
        self.deferred = None
        self.pending = None
        self.bad_lines = 0

    @staticmethod
    def _read_string(f):
        """bufio.Reader.ReadString('\\n'): (line, None) or (partial, EOF)"""
        line = f.readline()
        if not line.endswith("\n"):
            return line, EOF
        return line, None

    def read_one(self):
        """reader.go:128-190 -> (record dict, err)"""
        rec = {}
        while True:
            l1, err = self._read_string(self.r1)
            if err:
                return rec, err
            l2, err = self._read_string(self.r2)
            if err:
                return rec, err
            if l1[0] == "@":
                rec["info"], rec["barcode"], rec["valid"] = parse_header(l1[1:])
                fields = l1[1:len(l1) - 1].split()
                rec["rg"] = "" if len(fields) < 2 else fields[-1]
                break
            self.bad_lines += 1
        lines = {}
        for i in range(3):      # intended semantics: sequence, '+', quality from both files
            a, err = self._read_string(self.r1)
            if err:
                return rec, err
            b, err = self._read_string(self.r2)
            if err:
                return rec, err
            lines[i] = (a[:-1], b[:-1])
        rec["s1"], rec["s2"] = lines[0]
        rec["q1"], rec["q2"] = lines[2]
        return rec, None

    def read_barcode_set(self):
        """reader.go:209-300 -> (records, err, unique)"""
        new_barcode = False
        if self.deferred is not None:
            return None, self.deferred, False
        arr = []
        index = 0
        if self.pending is not None:
            arr.append(self.pending)
            self.pending = None
            index += 1
        while index < 30000:
            rec, err = self.read_one()
            arr.append(rec)
            if err is not None:
                if index == 0:
                    return None, err, False
                self.deferred = err
                break
            if arr[0]["barcode"] != arr[index]["barcode"]:
                self.pending = arr[index]
                new_barcode = True
                break
            elif self.last_barcode is not None and arr[0]["barcode"] == self.last_barcode and index >= 200:
                new_barcode = False
                break
            index += 1
        if len(arr) > 0:
            self.last_barcode = arr[0]["barcode"]
        end = len(arr)
        if new_barcode or self.deferred == EOF:
            end -= 1
        elif self.deferred != EOF:
            return arr[:end], None, False
        return arr[:end], None, True


def worth_running_rfa(records, unique):
    """aligner.go:1018-1030"""
    if len(records) == 0 or not unique:
        return False
    if len(records[0]["barcode"].split("-")) < 2:
        return False
    if len(records) < 5:
        return False
    return True


def all_sets(r1_text, r2_text):
    """every set the producer loop of aligner.go:339-358 would hand to a worker"""
    rd = Reader(r1_text, r2_text)
    out = []
    while True:
        recs, err, unique = rd.read_barcode_set()
        if err is not None:
            break
        out.append((recs, unique, worth_running_rfa(recs, unique)))
    return out, rd.bad_lines
