"""BAM -> SAM text, the way `samtools view -h` prints it -- TEST INFRASTRUCTURE ONLY.

The reference does not decode BAM itself: it pipes the file through `samtools view` (xenomapper.py:48-93, samtools
is a third-party tool that is not installed here) and treats the text like SAM input.  This is a plain-Python
restatement of the published BAM layout (SAM/BAM specification v1, section 4: BGZF members, header block, alignment
records, optional-field types) used to check the native decoder (xenomapper_amd/csrc/xm_bam.cpp) on inputs beyond the
reference's two BAM fixtures.  Parity status: PINNED to the reference's own data -- tests/test_oracle_golden.py checks
that it turns the reference's BAM fixtures into exactly the reference's SAM fixtures.  Nothing under xenomapper_amd/
imports this file.
"""
import gzip
import struct

SEQ_CODE = "=ACMGRSVTWYHKDBN"
CIGAR_CODE = "MIDNSHP=XB??????"
_SCALAR = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f", "d": "<d"}


def _real(v):
    return "%g" % v


def _scalar(raw, p, t):
    if t == "A":
        return chr(raw[p]), p + 1
    fmt = _SCALAR[t]
    v, = struct.unpack_from(fmt, raw, p)
    return (_real(v) if t in "fd" else str(v)), p + struct.calcsize(fmt)


def _value_bytes(rec, p, t):
    if t in "ZH":
        return rec.index(b"\0", p) + 1 - p
    if t == "B":
        count, = struct.unpack_from("<I", rec, p + 1)
        return 5 + count * struct.calcsize(_SCALAR[chr(rec[p])])
    return 1 if t == "A" else struct.calcsize(_SCALAR[t])


def _real_cigar_field(rec, aux0, n_cigar, cigar_ops, l_seq, ref_id, pos):
    """Offset of the CG:B:I field holding the real CIGAR, or None (conditions of htslib's bam_tag2cigar)."""
    if n_cigar == 0 or ref_id < 0 or pos < 0 or (cigar_ops[0] & 15) != 4 or (cigar_ops[0] >> 4) != l_seq:
        return None
    p = aux0
    while p + 3 <= len(rec):
        t = chr(rec[p + 2])
        if rec[p:p + 2] == b"CG":
            if t != "B" or chr(rec[p + 3]) != "I":
                return None
            count, = struct.unpack_from("<I", rec, p + 4)
            return p if n_cigar <= count < (1 << 29) else None
        p += 3 + _value_bytes(rec, p + 3, t)
    return None


def record_to_line(rec, ref_names):
    """One alignment record (bytes after its block_size word) -> one SAM line without the newline."""
    ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHIiii", rec, 0)
    p = 32
    name = rec[p:p + l_name].split(b"\0")[0].decode("latin-1")
    p += l_name
    cigar_ops = struct.unpack_from("<%dI" % n_cigar, rec, p)
    p += 4 * n_cigar
    seq = "".join(SEQ_CODE[(rec[p + k // 2] >> (0 if k & 1 else 4)) & 15] for k in range(l_seq)) or "*"
    p += (l_seq + 1) // 2
    qual = "*" if (l_seq == 0 or rec[p] == 0xFF) else "".join(chr(q + 33) for q in rec[p:p + l_seq])
    p += l_seq

    def ref(i):
        return ref_names[i] if 0 <= i < len(ref_names) else "*"
    rnext = "=" if (0 <= next_ref < len(ref_names) and next_ref == ref_id) else ref(next_ref)
    # A CIGAR of more than 65535 operations is stored as the placeholder "<l_seq>S<ref_len>N" plus a CG:B:I field; htslib
    # moves the real operations back when it reads the record (sam.c, bam_tag2cigar), so `samtools view` prints them
    # as the CIGAR and prints no CG field.
    real_at = _real_cigar_field(rec, p, n_cigar, cigar_ops, l_seq, ref_id, pos)
    if real_at is not None:
        count, = struct.unpack_from("<I", rec, real_at + 4)
        cigar_ops = struct.unpack_from("<%dI" % count, rec, real_at + 8)
    cigar = "".join("%d%s" % (v >> 4, CIGAR_CODE[v & 15]) for v in cigar_ops) or "*"
    fields = [name, str(flag), ref(ref_id), str(pos + 1), str(mapq), cigar, rnext, str(next_pos + 1), str(tlen), seq, qual]
    while p < len(rec):
        if real_at is not None and p == real_at:
            p += 8 + 4 * len(cigar_ops)
            continue
        tag, t = rec[p:p + 2].decode("latin-1"), chr(rec[p + 2])
        p += 3
        if t in "cCsSiI":
            v, p = _scalar(rec, p, t)
            fields.append("%s:i:%s" % (tag, v))
        elif t in "Afd":
            v, p = _scalar(rec, p, t)
            fields.append("%s:%s:%s" % (tag, t, v))
        elif t in "ZH":
            end = rec.index(b"\0", p)
            fields.append("%s:%s:%s" % (tag, t, rec[p:end].decode("latin-1")))
            p = end + 1
        elif t == "B":
            sub = chr(rec[p])
            count, = struct.unpack_from("<I", rec, p + 1)
            p += 5
            vals = []
            for _ in range(count):
                v, p = _scalar(rec, p, sub)
                vals.append(v)
            fields.append("%s:B:%s" % (tag, ",".join([sub] + vals)))
        else:
            raise ValueError("unknown optional field type %r" % t)
    return "\t".join(fields)


def bam_to_sam(data):
    """(header text, list of alignment lines) of a BAM file image."""
    raw = gzip.decompress(bytes(data))
    if raw[:4] != b"BAM\1":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack_from("<i", raw, 4)
    header = raw[8:8 + l_text].rstrip(b"\0").decode("latin-1")
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p)
    p += 4
    names = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p)
        names.append(raw[p + 4:p + 4 + l_name].rstrip(b"\0").decode("latin-1"))
        p += 4 + l_name + 4
    lines = []
    while p < len(raw):
        size, = struct.unpack_from("<i", raw, p)
        lines.append(record_to_line(raw[p + 4:p + 4 + size], names))
        p += 4 + size
    return header, lines
