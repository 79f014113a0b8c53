"""CPU oracle for the xenomappability tool -- TEST INFRASTRUCTURE ONLY.

Plain-Python restatement of /root/reference/xenomapper/mappability.py (v1.0.2).  Pinned by
tests/test_mappability_oracle.py against the digests the reference's own tests hold
(xenomapper/tests/test_mappability.py:40, :101, :109, :118, :135, :143) and against golden vectors recorded from
the imported reference (tools/make_golden.py -> tests/golden/g6_mappability.json).  Nothing under xenomapper_amd/
imports this file.
"""
from collections import Counter
from statistics import mean


def correlate_track(track, mate_density):
    """Paired-end mappability of one chromosome.  Restates mappability.py:94-124: a position that is uniquely
    mappable (== 1) stays 1.0; otherwise the sum over j of track[i+j] * mate_density[j], j running while both
    lists last, accumulated left to right in binary64 (multiply, round, add, round -- no fused multiply-add)."""
    m = len(mate_density)
    out = []
    for i in range(len(track)):
        if track[i] == 1:
            out.append(1.0)
            continue
        acc = 0.0
        j = 0
        while i + j < len(track) and j < m:
            acc += track[i + j] * mate_density[j]
            j += 1
        out.append(acc)
    return out


def read_fasta(handle, token=">"):
    """(name, sequence) records.  Restates mappability.py:127-146."""
    name, seq = None, None
    for line in handle:
        line = line.strip()
        if line.startswith(token):
            if name:
                yield name, seq
            name, seq = line[1:], ""
        elif seq is not None:
            seq += line
    if name:
        yield name, seq


def blocks(seq, size=80):
    """Restates mappability.py:148-157 (note: a final block of exactly `size` characters is produced by the
    else-branch, same text)."""
    return [seq[b:b + size] for b in range(0, len(seq), size)]


def fasta_text(name, seq, size=80):
    """Restates mappability.py:159-166."""
    return ">" + name + "\n" + "\n".join(blocks(seq, size)) + "\n"


def simulated_reads_text(handle, readlength=100):
    """Every window of `readlength` as a FASTA record named chrom_<1-based position>.  Restates :168-173."""
    parts = []
    for name, seq in read_fasta(handle):
        chrom = name.split()[0]
        for x in range(len(seq) - readlength + 1):
            parts.append(fasta_text("%s_%d" % (chrom, x + 1), seq[x:x + readlength]))
    return "".join(parts)


def single_end_track_from_sam(lines):
    """{chromosome: 0/1 list} from a name-sorted SAM of simulated reads (header lines already consumed).
    Restates mappability.py:183-209: a read counts as uniquely mappable when it maps back to its own origin
    (name chrom_pos == RNAME, POS) with MAPQ == '42'."""
    tracks = {}
    cur, pos, vals = None, 0, []
    for line in lines:
        f = line.strip("\n").split()
        name, chrom, at, mapq = f[0], f[2], f[3], f[4]
        true_chrom = "_".join(name.split("_")[:-1])
        if cur != true_chrom:
            if cur and vals:
                tracks[cur] = vals
            cur, pos, vals = chrom, 0, []                     # (sic) the new key is RNAME, as in the reference :189
        pos += 1
        name_pos = int(name.split("_")[-1])
        if name_pos != pos:
            if not name_pos > pos:
                raise ValueError("Name is not sequential")
            vals.extend([0] * (name_pos - pos))
            pos = name_pos
        vals.append(1 if (true_chrom, name_pos) == (chrom, int(at)) and mapq == "42" else 0)
    if cur and vals:
        tracks[cur] = vals
    return tracks


def wiggle_text(tracks, chromosomes=()):
    """Restates Mappability.to_wiggle, mappability.py:46-57."""
    parts = []
    for chrom in sorted(tracks):
        if not chromosomes or chrom in chromosomes:
            parts.append("fixedStep\tchrom=%s\tstart=1\tstep=1\n" % chrom)
            parts.extend(str(v) + "\n" for v in tracks[chrom])
    return "".join(parts)


def tracks_from_wiggle(handle, datatype=float):
    """Restates Mappability.from_wiggle, mappability.py:59-92."""
    tracks, chrom, vals = {}, None, []
    for line in handle:
        if line.startswith("fixedStep"):
            f = line.strip().split("\t")
            if f[1].split("=")[0] != "chrom" or f[2] != "start=1" or f[3] != "step=1":
                raise ValueError("Unsupported wiggle fixed step format")
            if chrom and vals:
                tracks[chrom] = vals
            chrom, vals = f[1].split("=")[1], []
        else:
            vals.append(datatype(line))
    tracks[chrom] = vals
    return tracks


def smoothed(values, width=10):
    """Restates mappability.py:237-238 (statistics.mean: exact, so results may be int or float)."""
    return [mean(values[max(0, x - width - 1):x + width]) for x in range(len(values))]


def mate_density_from_sam(lines, sample_size=10000):
    """Restates mappability.py:255-273: |TLEN| histogram of the first sample_size+1 non-zero inserts -> smoothed ->
    small values removed -> normalised."""
    sizes = []
    for line in lines:
        if not line or line[0] == "@":
            continue
        f = line.strip("\n").split()
        if f[8] != "0":
            sizes.append(abs(int(f[8])))
        if sample_size and len(sizes) > sample_size:
            break
    freq = Counter(sizes)
    hist = [freq[i] if i in freq else 0 for i in range(0, max(sizes))]
    sm = smoothed(hist)
    floor = max(sm) * 0.1
    kept = [x if x > floor else 0 for x in sm]
    total = sum(kept)
    return [x / total for x in kept]


def paired_wiggle_text(wiggle_handle, mate_density, chromosome_sizes):
    """Restates paired_end_mappability, mappability.py:216-235 (chromosomes named in chromosome_sizes start as
    all-zero tracks and are the only ones written)."""
    tracks = {c: [0] * chromosome_sizes[c] for c in chromosome_sizes}
    tracks.update(tracks_from_wiggle(wiggle_handle, float))
    paired = {c: correlate_track(tracks[c], mate_density) for c in tracks}
    return wiggle_text(paired, chromosomes=list(chromosome_sizes.keys()))
