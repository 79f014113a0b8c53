#!/usr/bin/env python3
"""Writes tests/golden/long_cigar_cg.{bam,sam}: alignments whose CIGAR has more than 65535 operations, encoded the way
the SAM/BAM specification prescribes, next to the SAM text they stand for.

SAMv1 section 4.2.2 ("N_CIGAR_OP field"): the BAM record's n_cigar_op is 16 bits wide, so for an alignment with more
than 65535 CIGAR operations the writer stores the real CIGAR in an optional field `CG` of type `B,I` (the operations
packed op_len << 4 | op like the CIGAR field) and sets the CIGAR field itself to the two operations `kSmN`, k = the
length of SEQ and m = the length of the alignment on the reference.  A reader puts the real CIGAR back and drops the CG
field -- so the SAM text of such a record is simply the record with its real CIGAR.  This script is the WRITER side: it
builds the logical records (name, flag, position, real CIGAR, SEQ, QUAL, tags), prints them as SAM from that logical
content, and encodes them as BAM by the rule above.  It shares no code with the decoder (csrc/xm_bam.cpp) or with
oracle/bam_oracle.py; the decoder must turn the .bam into exactly the .sam.

    python tools/make_bam_long_cigar_fixture.py          (rewrites the two fixture files; deterministic)
"""
import os
import struct
import zlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = "MIDNSHP=X"
SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def logical_records():
    """(qname, flag, pos1, mapq, cigar [(len, op)], seq, qual, tags [(tag, type, value)])"""
    recs = []
    # 70 000 operations: 1M1I repeated (the score path: 35 000 insertions), read length 70 000
    cig = [(1, "M"), (1, "I")] * 35000
    recs.append(("long_a", 0, 101, 60, cig, "AC" * 35000, "I" * 70000, [("NM", "i", 35000), ("XS", "i", -7)]))
    # 65 537 operations, just over the 16-bit limit; soft clips at both ends, deletions inside
    cig = [(3, "S")] + [(2, "M"), (1, "D")] * 32767 + [(1, "M"), (4, "S")]
    n_seq = 3 + 2 * 32767 + 1 + 4
    recs.append(("long_b", 16, 5, 3, cig, "G" * n_seq, "#" * n_seq, [("NM", "i", 32767)]))
    # an ordinary record after them (the stream goes on normally), and one with exactly 65 535 operations (no CG needed)
    recs.append(("short_c", 0, 7, 50, [(50, "M")], "T" * 50, "5" * 50, [("NM", "i", 0), ("AS", "i", 100)]))
    cig = [(1, "M"), (1, "I")] * 32767 + [(1, "M")]
    recs.append(("edge_d", 0, 9, 11, cig, "C" * 65535, "+" * 65535, [("NM", "i", 32767)]))
    return recs


def ref_len(cigar):
    return sum(n for n, op in cigar if op in "MDN=X")


def sam_line(rec):
    q, flag, pos, mapq, cigar, seq, qual, tags = rec
    fields = [q, str(flag), "chrL", str(pos), str(mapq), "".join("%d%s" % c for c in cigar), "*", "0", "0", seq, qual]
    fields += ["%s:%s:%d" % t for t in tags]
    return "\t".join(fields)


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bam_record(rec):
    q, flag, pos, mapq, cigar, seq, qual, tags = rec
    packed = [(n << 4) | OPS.index(op) for n, op in cigar]
    aux = b""
    for tag, typ, val in tags:
        assert typ == "i"
        aux += tag.encode() + (b"c" + struct.pack("<b", val) if -128 <= val < 128 else
                               b"s" + struct.pack("<h", val) if -32768 <= val < 32768 else b"i" + struct.pack("<i", val))
    if len(packed) > 65535:                                       # section 4.2.2: real CIGAR -> CG:B,I ; CIGAR := kSmN
        aux += b"CGBI" + struct.pack("<I", len(packed)) + struct.pack("<%dI" % len(packed), *packed)
        packed = [(len(seq) << 4) | OPS.index("S"), (ref_len(cigar) << 4) | OPS.index("N")]
    name = q.encode() + b"\0"
    nib = [SEQ_CODE[c] for c in seq] + [0]
    seq_bytes = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(seq), 2))
    qual_bytes = bytes(ord(c) - 33 for c in qual)
    pos0 = pos - 1
    core = struct.pack("<iiBBHHHIiii", 0, pos0, len(name), mapq, reg2bin(pos0, pos0 + max(ref_len(cigar), 1)), len(packed), flag,
                       len(seq), -1, -1, 0)
    body = core + name + struct.pack("<%dI" % len(packed), *packed) + seq_bytes + qual_bytes + aux
    return struct.pack("<I", len(body)) + body


def bgzf(payload, block=65280):
    out = []
    for at in range(0, len(payload), block):
        part = payload[at:at + block]
        comp = zlib.compressobj(9, zlib.DEFLATED, -15)
        data = comp.compress(part) + comp.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(data) + 25) + data +
                   struct.pack("<II", zlib.crc32(part), len(part)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))      # the empty end-of-file block
    return b"".join(out)


def main():
    header = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chrL\tLN:400000\n"
    recs = logical_records()
    text = header.encode()
    bam = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    bam += struct.pack("<i", 5) + b"chrL\0" + struct.pack("<i", 400000)
    bam += b"".join(bam_record(r) for r in recs)
    out = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(out, "long_cigar_cg.bam"), "wb") as fh:
        fh.write(bgzf(bam))
    with open(os.path.join(out, "long_cigar_cg.sam"), "w") as fh:
        fh.write(header + "".join(sam_line(r) + "\n" for r in recs))
    print("wrote", os.path.join(out, "long_cigar_cg.{bam,sam}"), [len(r[4]) for r in recs], "operations")


if __name__ == "__main__":
    main()
