"""Synthetic workloads for the classification path.

Two products, both the build's own (nothing here comes from the reference):

* ``score_columns`` -- structure-of-arrays score columns for the large configurations
  (the form the device ingests); NumPy ``Generator(PCG64(seed))``, generated in chunks.
* ``sam_text_pair`` -- a small SAM *text* twin (two files, primary + secondary species)
  of the same score model, for parity tests against the oracle / golden vectors.

Score model, Bowtie2 ``--local`` on 2x150 bp ("bowtie2" profile): AS in [61, 300];
pair origin primary 0.86 / secondary 0.09 / conserved in both 0.03 / neither 0.02; in the
origin species a mate maps w.p. 0.98 with AS = 300 - 2*(Geometric(0.08)-1) clipped to
[61, 300], XS present w.p. 0.35 (XS = AS w.p. 0.25 else Uniform[61, AS-1]); in the other
species it maps w.p. 0.15 with AS ~ Uniform[61, 220], XS present w.p. 0.4.
"hisat" profile: end-to-end scores AS in [-90, 0] with AS = 0 and ZS = 0 records, the
second-best score lives in a ZS:i field and XS:A:+/- is a strand tag.
"cigar" profile: no AS tag; NM ~ Poisson(1.2) and a CIGAR drawn from
{150M 0.80, soft-clipped one end 0.12, both ends 0.03, one indel 0.04, two indels 0.01}.
"""
from __future__ import annotations

import math

import numpy as np

ABSENT = np.int32(np.iinfo(np.int32).min)      # the integer stand-in for float('-inf')

BAM_OPS = "MIDNSHP=X"
OP_M, OP_I, OP_D, OP_N, OP_S = 0, 1, 2, 3, 4


def _origin(rng, n):
    u = rng.random(n)
    # 0 primary, 1 secondary, 2 both (conserved), 3 neither
    return np.digitize(u, [0.86, 0.95, 0.98]).astype(np.int8)


def _home_scores_bt2(rng, n):
    mapped = rng.random(n) < 0.98
    a = 300 - 2 * (rng.geometric(0.08, n) - 1)
    a = np.clip(a, 61, 300).astype(np.int32)
    has_x = rng.random(n) < 0.35
    same = rng.random(n) < 0.25
    below = 61 + (rng.random(n) * np.maximum(a - 61, 1)).astype(np.int32)
    below = np.minimum(below, np.maximum(a - 1, 61))
    x = np.where(same, a, below).astype(np.int32)
    a = np.where(mapped, a, ABSENT)
    x = np.where(mapped & has_x, x, ABSENT)
    return a, x


def _away_scores_bt2(rng, n):
    mapped = rng.random(n) < 0.15
    a = rng.integers(61, 221, n, dtype=np.int32)
    has_x = rng.random(n) < 0.4
    x = 61 + (rng.random(n) * (a - 60)).astype(np.int32)
    x = np.minimum(x, a)
    a = np.where(mapped, a, ABSENT)
    x = np.where(mapped & has_x, x, ABSENT)
    return a, x


def _home_scores_hisat(rng, n):
    mapped = rng.random(n) < 0.98
    a = -np.minimum(90, 3 * (rng.geometric(0.25, n) - 1)).astype(np.int32)
    has_x = rng.random(n) < 0.35
    same = rng.random(n) < 0.25
    x = np.where(same, a, a - rng.integers(1, 31, n, dtype=np.int32)).astype(np.int32)
    a = np.where(mapped, a, ABSENT)
    x = np.where(mapped & has_x, x, ABSENT)
    return a, x


def _away_scores_hisat(rng, n):
    mapped = rng.random(n) < 0.15
    a = -rng.integers(10, 91, n, dtype=np.int32)
    has_x = rng.random(n) < 0.4
    x = a - rng.integers(0, 31, n, dtype=np.int32)
    a = np.where(mapped, a, ABSENT)
    x = np.where(mapped & has_x, x, ABSENT)
    return a, x


def _species_scores(rng, origin_rec, profile):
    """Scores of every record in both species given the per-record origin
    (0 primary, 1 secondary, 2 conserved in both, 3 neither)."""
    n = origin_rec.shape[0]
    home, away = ((_home_scores_hisat, _away_scores_hisat) if profile == "hisat"
                  else (_home_scores_bt2, _away_scores_bt2))
    ha, hx = home(rng, n)
    oa, ox = away(rng, n)
    _, hx2 = home(rng, n)              # second-best score of the conserved copy
    sec = origin_rec == 1
    both = origin_rec == 2
    none = origin_rec == 3
    as1 = np.where(sec, oa, ha)
    xs1 = np.where(sec, ox, hx)
    as2 = np.where(sec | both, ha, oa)                    # conserved: AS2 == AS1
    cx = np.where((hx2 == ABSENT) | (ha == ABSENT), ABSENT, np.minimum(hx2, ha))
    xs2 = np.where(sec, hx, np.where(both, cx, ox))
    cols = [c.astype(np.int32) for c in (as1, xs1, as2, xs2)]
    for col in cols:
        col[none] = ABSENT
    return tuple(cols)


def interleaved_unit_bits(n_records):
    """Packed unit mask of strictly interleaved mates: record 2p+1 closes pair p."""
    n_words = (n_records + 63) // 64
    bits = np.full(n_words, 0xAAAAAAAAAAAAAAAA, dtype=np.uint64)
    tail = n_records & 63
    if tail:
        bits[-1] &= np.uint64((1 << tail) - 1)
    return bits


def pack_unit_bits(flags):
    """bool/0-1 array (one per record) -> packed little-endian uint64 words."""
    flags = np.asarray(flags, dtype=np.uint8)
    n = flags.shape[0]
    n_words = (n + 63) // 64
    padded = np.zeros(n_words * 64, dtype=np.uint8)
    padded[:n] = flags
    return np.packbits(padded, bitorder="little").view(np.uint64)


def score_columns(n_pairs, seed, profile="bowtie2", chunk_pairs=1 << 22):
    """Score columns of n_pairs interleaved read pairs (2*n_pairs records per species).

    Returns dict(as1, xs1, as2, xs2 : int32[2*n_pairs], unit_bits : uint64[ceil(n/64)]).
    Both mates of a pair share the pair's origin; scores are drawn per record.
    """
    n = 2 * n_pairs
    out = {k: np.empty(n, dtype=np.int32) for k in ("as1", "xs1", "as2", "xs2")}
    rng = np.random.Generator(np.random.PCG64(seed))
    for p0 in range(0, n_pairs, chunk_pairs):
        p1 = min(n_pairs, p0 + chunk_pairs)
        origin = np.repeat(_origin(rng, p1 - p0), 2)
        cols = _species_scores(rng, origin, profile)
        for k, col in zip(("as1", "xs1", "as2", "xs2"), cols):
            out[k][2 * p0:2 * p1] = col
    out["unit_bits"] = interleaved_unit_bits(n)
    return out


# ----------------------------------------------------------------------------------------
# CIGAR workload
# ----------------------------------------------------------------------------------------

def cigar_columns(n_records, seed, read_len=150, mapped_p=0.9):
    """NM + packed-CIGAR CSR columns for one species ("cigar" profile).

    Returns dict(nm int32[n] (ABSENT = no NM field), cig_off uint32[n+1], cig_oplen uint32[]).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    mapped = rng.random(n_records) < mapped_p
    nm = np.where(mapped, rng.poisson(1.2, n_records), 0).astype(np.int32)
    shape = np.digitize(rng.random(n_records), [0.80, 0.92, 0.95, 0.99])   # 0..4
    shape = np.where(mapped, shape, -1)
    n_ops = np.select([shape == -1, shape == 0, shape == 1, shape == 2, shape == 3],
                      [0, 1, 2, 3, 3], default=5).astype(np.uint32)
    off = np.zeros(n_records + 1, dtype=np.uint32)
    np.cumsum(n_ops, out=off[1:])
    ops = np.zeros(int(off[-1]), dtype=np.uint32)
    a = rng.integers(1, 40, n_records).astype(np.uint32)
    b = rng.integers(1, 4, n_records).astype(np.uint32)
    c = rng.integers(1, 30, n_records).astype(np.uint32)
    is_del = rng.random(n_records) < 0.5
    clip_left = rng.random(n_records) < 0.5
    L = np.uint32(read_len)
    base = off[:-1]

    def put(sel, k, length, op):
        opv = np.broadcast_to(np.asarray(op, dtype=np.uint32), (n_records,))
        ops[base[sel] + k] = (length[sel].astype(np.uint32) << np.uint32(4)) | opv[sel]

    s = shape == 0
    ops[base[s]] = (L << np.uint32(4)) | np.uint32(OP_M)
    s = (shape == 1) & clip_left                     # aSbM
    put(s, 0, a, OP_S); put(s, 1, L - a, OP_M)
    s = (shape == 1) & ~clip_left                    # aMbS
    put(s, 0, L - a, OP_M); put(s, 1, a, OP_S)
    s = shape == 2                                   # aSbMcS
    put(s, 0, a, OP_S); put(s, 1, L - a - c, OP_M); put(s, 2, c, OP_S)
    s = shape == 3                                   # aM bI/D cM
    indel = np.where(is_del, OP_D, OP_I).astype(np.uint32)
    put(s, 0, a + 10, OP_M); put(s, 1, b, indel)
    put(s, 2, L - a - 10 - np.where(is_del, 0, b).astype(np.uint32), OP_M)
    s = shape == 4                                   # aM bI cM bD rest M
    put(s, 0, a + 5, OP_M); put(s, 1, b, OP_I)
    put(s, 2, c + 5, OP_M); put(s, 3, b, OP_D)
    put(s, 4, L - a - c - 10 - b, OP_M)
    nm = np.where(mapped, nm, ABSENT).astype(np.int32)
    return {"nm": nm, "cig_off": off, "cig_oplen": ops}


def cigar_string(oplen):
    if len(oplen) == 0:
        return "*"
    return "".join("%d%s" % (int(v) >> 4, BAM_OPS[int(v) & 15]) for v in oplen)


# ----------------------------------------------------------------------------------------
# SAM text twins
# ----------------------------------------------------------------------------------------

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng, length):
    return _BASES[rng.integers(0, 4, length)].tobytes().decode("ascii")


def _rand_qual(rng, length):
    return (rng.integers(35, 75, length).astype(np.uint8)).tobytes().decode("ascii")


def sam_text_pair(n_pairs, seed, profile="bowtie2", paired=True, read_len=150,
                  mixed_ws=0.0, irregular=0.0, header_pg=True):
    """Two SAM texts (primary, secondary) of the same reads in the same order.

    paired      -- two records per QNAME (interleaved mates); else one record per QNAME.
    mixed_ws    -- fraction of lines whose separators are a mix of tabs and spaces.
    irregular   -- fraction of QNAMEs that get 1 or 3 records instead of 2 (paired), or a
                   repeated record (single-end), to exercise the unit mask / skip_repeated.
    Returns (text1, text2, info) where info carries the per-record columns used.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    per_name = np.full(n_pairs, 2 if paired else 1, dtype=np.int64)
    if irregular > 0:
        odd = rng.random(n_pairs) < irregular
        alt = np.where(rng.random(n_pairs) < 0.5, 1, 3) if paired else np.full(n_pairs, 2)
        per_name = np.where(odd, alt, per_name)
    name_of = np.repeat(np.arange(n_pairs), per_name)
    n = int(name_of.shape[0])
    origin = _origin(rng, n_pairs)[name_of]
    prof = "hisat" if profile == "hisat" else "bowtie2"
    as1, xs1, as2, xs2 = _species_scores(rng, origin, prof)
    cig = [None, None]
    if profile == "cigar":
        cig = [cigar_columns(n, seed * 7 + 1 + f, read_len) for f in range(2)]
        # a sprinkling of genuine XS tags so the multi states occur on this path too
        for x in (xs1, xs2):
            x[:] = np.where(rng.random(n) < 0.05, -rng.integers(0, 40, n, dtype=np.int32), ABSENT)

    first_of = np.ones(n, dtype=bool)
    first_of[1:] = name_of[1:] != name_of[:-1]
    seq_of = [(_rand_seq(rng, read_len), _rand_qual(rng, read_len)) for _ in range(n)]
    texts = []
    for f, (a_col, x_col) in enumerate(((as1, xs1), (as2, xs2))):
        species = ("hs", "mm")[f]
        head = ["@HD\tVN:1.0\tSO:unsorted"]
        head += ["@SQ\tSN:%s_chr%d\tLN:%d" % (species, k + 1, 1000000 * (k + 3)) for k in range(3 + f)]
        if header_pg:
            head.append("@PG\tID:%s\tPN:%s\tVN:2.0" % (("hisat2" if profile == "hisat" else "bowtie2",) * 2))
        lines = list(head)
        for i in range(n):
            q = int(name_of[i])
            first = bool(first_of[i])
            a, x = int(a_col[i]), int(x_col[i])
            if profile == "cigar":
                nmv = int(cig[f]["nm"][i])
                mapped = nmv != int(ABSENT)
                ops = cig[f]["cig_oplen"][cig[f]["cig_off"][i]:cig[f]["cig_off"][i + 1]]
                cigar = cigar_string(ops) if mapped else "*"
            else:
                mapped = a != int(ABSENT)
                cigar = ("%dM" % read_len) if mapped else "*"
                nmv = int(rng.integers(0, 4)) if mapped else None
            if paired:
                flag = (99 if first else 147) if mapped else (77 if first else 141)
            else:
                flag = (0 if rng.random() < 0.5 else 16) if mapped else 4
            seq, qual = seq_of[i]
            cols = ["frag%07d" % q, str(flag),
                    ("%s_chr%d" % (species, 1 + q % 3)) if mapped else "*",
                    str(1 + (q * 7919 + i) % 900000) if mapped else "0",
                    str(int(rng.integers(0, 45))) if mapped else "0", cigar,
                    "=" if (mapped and paired) else "*", "0", "0", seq, qual]
            if profile == "cigar":
                if mapped:
                    cols.append("NM:i:%d" % nmv)
                if x != int(ABSENT):
                    cols.append("XS:i:%d" % x)
            else:
                if mapped:
                    cols.append("AS:i:%d" % a)
                    if profile == "hisat":
                        if x != int(ABSENT):
                            cols.append("ZS:i:%d" % x)
                        cols.append("XS:A:%s" % ("+" if (q + i) % 2 else "-"))
                        cols += ["NM:i:%d" % nmv, "NH:i:1"]
                    else:
                        if x != int(ABSENT):
                            cols.append("XS:i:%d" % x)
                        cols += ["XN:i:0", "XM:i:%d" % nmv, "XO:i:0", "XG:i:0", "NM:i:%d" % nmv,
                                 "MD:Z:%d" % read_len]
                        if paired:
                            cols.append("YS:i:%d" % (a - (i % 5)))
                cols.append("YT:Z:%s" % ("CP" if paired else "UU"))
            if mixed_ws > 0 and rng.random() < mixed_ws:
                seps = np.where(rng.random(len(cols) - 1) < 0.5, "\t", " ")
                text = cols[0] + "".join(s + c for s, c in zip(seps, cols[1:]))
            else:
                text = "\t".join(cols)
            lines.append(text)
        texts.append("\n".join(lines) + "\n")
    info = {"n_records": n, "name_of": name_of, "as1": as1, "xs1": xs1, "as2": as2, "xs2": xs2, "cigar": cig}
    return texts[0], texts[1], info


# ----------------------------------------------------------------------------------------
# device-side generator for the large configurations (same score model, torch RNG)
# ----------------------------------------------------------------------------------------

def score_columns_torch(n_pairs, seed, device, profile="bowtie2", chunk_pairs=1 << 24):
    """score_columns() drawn with torch on `device` (so 50 M pairs take milliseconds on the GPU
    instead of a minute of NumPy).  Same distributions, different random stream.  Returns torch
    tensors: as1, xs1, as2, xs2 int32[2*n_pairs], unit_bits int64[ceil(n/64)] (bit pattern of uint64)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    absent = int(ABSENT)
    n = 2 * n_pairs
    out = {k: torch.empty(n, dtype=torch.int32, device=device) for k in ("as1", "xs1", "as2", "xs2")}

    def rand(m):
        return torch.rand(m, generator=gen, device=device)

    def randint(lo, hi, m):
        return torch.randint(lo, hi, (m,), generator=gen, device=device, dtype=torch.int32)

    def geometric(p, m):          # support 1, 2, ...
        u = rand(m).clamp_min(1e-12)
        return (torch.log(u) / math.log1p(-p)).floor().to(torch.int32) + 1

    def where_absent(keep, v):
        return torch.where(keep, v, torch.full_like(v, absent))

    def home(m):
        mapped = rand(m) < 0.98
        if profile == "hisat":
            a = -torch.clamp(3 * (geometric(0.25, m) - 1), max=90)
            has_x = rand(m) < 0.35
            same = rand(m) < 0.25
            x = torch.where(same, a, a - randint(1, 31, m))
        else:
            a = torch.clamp(300 - 2 * (geometric(0.08, m) - 1), 61, 300)
            has_x = rand(m) < 0.35
            same = rand(m) < 0.25
            below = 61 + (rand(m) * torch.clamp(a - 61, min=1)).to(torch.int32)
            below = torch.minimum(below, torch.clamp(a - 1, min=61))
            x = torch.where(same, a, below)
        return where_absent(mapped, a), where_absent(mapped & has_x, x)

    def away(m):
        mapped = rand(m) < 0.15
        has_x = rand(m) < 0.4
        if profile == "hisat":
            a = -randint(10, 91, m)
            x = a - randint(0, 31, m)
        else:
            a = randint(61, 221, m)
            x = torch.minimum(61 + (rand(m) * (a - 60)).to(torch.int32), a)
        return where_absent(mapped, a), where_absent(mapped & has_x, x)

    for p0 in range(0, n_pairs, chunk_pairs):
        p1 = min(n_pairs, p0 + chunk_pairs)
        m = 2 * (p1 - p0)
        u = rand(p1 - p0)
        origin = ((u >= 0.86).to(torch.int8) + (u >= 0.95).to(torch.int8) + (u >= 0.98).to(torch.int8))
        origin = origin.repeat_interleave(2)
        ha, hx = home(m)
        oa, ox = away(m)
        _, hx2 = home(m)
        sec, both, none = origin == 1, origin == 2, origin == 3
        as1 = torch.where(sec, oa, ha)
        xs1 = torch.where(sec, ox, hx)
        as2 = torch.where(sec | both, ha, oa)
        cx = torch.where((hx2 == absent) | (ha == absent), torch.full_like(ha, absent), torch.minimum(hx2, ha))
        xs2 = torch.where(sec, hx, torch.where(both, cx, ox))
        for k, col in zip(("as1", "xs1", "as2", "xs2"), (as1, xs1, as2, xs2)):
            out[k][2 * p0:2 * p1] = torch.where(none, torch.full_like(col, absent), col)
    bits = torch.from_numpy(interleaved_unit_bits(n).view(np.int64)).to(device)
    out["unit_bits"] = bits
    return out


def cigar_columns_torch(n_records, seed, device, read_len=150, mapped_p=0.9):
    """cigar_columns() drawn with torch on `device` (same shape model, different random stream).
    Returns torch tensors nm int32[n], cig_off int32[n+1] (bit pattern of uint32), cig_oplen int32[]."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    n = n_records

    def rand():
        return torch.rand(n, generator=gen, device=device)

    def randint(lo, hi):
        return torch.randint(lo, hi, (n,), generator=gen, device=device, dtype=torch.int32)

    mapped = rand() < mapped_p
    nm = torch.poisson(torch.full((n,), 1.2, device=device), generator=gen).to(torch.int32)
    u = rand()
    shape = (u >= 0.80).to(torch.int32) + (u >= 0.92).to(torch.int32) + (u >= 0.95).to(torch.int32) + (u >= 0.99).to(torch.int32)
    a, b, c = randint(1, 40), randint(1, 4), randint(1, 30)
    is_del = rand() < 0.5
    clip_left = rand() < 0.5
    L = read_len

    def op(length, code):
        return (length << 4) | code

    zero = torch.zeros(n, dtype=torch.int32, device=device)
    indel = torch.where(is_del, torch.full_like(zero, OP_D), torch.full_like(zero, OP_I))
    dense = torch.zeros((n, 5), dtype=torch.int32, device=device)
    count = torch.zeros(n, dtype=torch.int32, device=device)
    rows = {
        0: ([op(torch.full_like(zero, L), OP_M)], shape == 0),
        1: ([op(a, OP_S), op(L - a, OP_M)], (shape == 1) & clip_left),
        2: ([op(L - a, OP_M), op(a, OP_S)], (shape == 1) & ~clip_left),
        3: ([op(a, OP_S), op(L - a - c, OP_M), op(c, OP_S)], shape == 2),
        4: ([op(a + 10, OP_M), (b << 4) | indel, op(L - a - 10 - torch.where(is_del, zero, b), OP_M)], shape == 3),
        5: ([op(a + 5, OP_M), op(b, OP_I), op(c + 5, OP_M), op(b, OP_D), op(L - a - c - 10 - b, OP_M)], shape == 4),
    }
    for ops_list, sel in rows.values():
        sel = sel & mapped
        for k, v in enumerate(ops_list):
            dense[:, k] = torch.where(sel, v, dense[:, k])
        count = torch.where(sel, torch.full_like(count, len(ops_list)), count)
    valid = torch.arange(5, device=device, dtype=torch.int32).unsqueeze(0) < count.unsqueeze(1)
    ops = dense[valid]
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(count.to(torch.int64), 0)
    nm = torch.where(mapped, nm, torch.full_like(nm, int(ABSENT)))
    if ops.numel() == 0:
        ops = torch.zeros(1, dtype=torch.int32, device=device)
    return {"nm": nm.contiguous(), "cig_off": off.to(torch.int32).contiguous(), "cig_oplen": ops.contiguous()}


def cigar_pack_torch(cig):
    """Packed CIGAR columns (include/xenomapper_hip.h) from cigar_columns_torch() output, on the same device.  The
    synthetic records have at most 5 operations, so no record needs a trailer word and the op array is unchanged.
    Returns dict(nm, cig_cnt uint8[n], cig_tile int32[tiles + 1] (bit pattern of uint32), cig_oplen)."""
    import torch
    off = cig["cig_off"].to(torch.int64) & 0xFFFFFFFF
    n = off.numel() - 1
    k = off[1:] - off[:-1]
    assert int(k.max().item()) < 255 if n else True
    n_tiles = (n + 255) // 256
    tile = torch.empty(n_tiles + 1, dtype=torch.int64, device=off.device)
    tile[:n_tiles] = off[0:n:256]
    tile[n_tiles] = off[n]
    return {"nm": cig["nm"], "cig_cnt": k.to(torch.uint8).contiguous(), "cig_tile": tile.to(torch.int32).contiguous(),
            "cig_oplen": cig["cig_oplen"]}
