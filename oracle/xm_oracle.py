"""CPU oracle for the xenomapper classification hot path -- TEST INFRASTRUCTURE ONLY.

This module is a plain-Python restatement of the reference algorithm
(genomematt/xenomapper v1.0.2).  It exists so that the HIP path can be checked
against something that runs everywhere; it is NOT part of the product.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``xenomapper_amd/`` imports it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against (a) the known-answer tables and SHA-224 digests held by the
reference's own test-suite (``xenomapper/tests/test_xenomapper.py:46-51, :93,
:125, :158, :165-183, :191-197, :203-209, :215-232``) and (b) golden vectors
recorded by importing the reference in the build container
(``tools/make_golden.py`` -> ``tests/golden/*.json``).

Every function cites the reference lines it restates.  The encoding of the six
states follows the priority order the reference documents at
``xenomapper.py:364-367``:

    0 primary_specific   1 secondary_specific   2 primary_multi
    3 secondary_multi    4 unresolved           5 unassigned
"""
from __future__ import annotations

import re
from collections import Counter

NEG_INF = float("-inf")

STATE_NAMES = (
    "primary_specific",
    "secondary_specific",
    "primary_multi",
    "secondary_multi",
    "unresolved",
    "unassigned",
)
STATE_INDEX = {name: i for i, name in enumerate(STATE_NAMES)}

PS, SS, PM, SM, UR, UA = range(6)

MODE_SE = 0
MODE_PE_LIBERAL = 1
MODE_PE_CONSERVATIVE = 2

# --------------------------------------------------------------------------
# score extraction (text level)
# --------------------------------------------------------------------------

def tag_score(fields, tag="AS"):
    """Numeric value of an optional SAM field.  Restates xenomapper.py:176-191.

    The match is a *substring* test against every optional field (index 11 on),
    not a prefix test (:186).  No match -> -inf (:187-188); more than one match
    -> ValueError (:189-190); otherwise the text after the last ':' goes
    through float() (:191), which may itself raise ValueError.
    """
    hits = []
    for opt in fields[11:]:
        if tag in opt:
            hits.append(opt)
    if len(hits) == 0:
        return NEG_INF
    if len(hits) > 1:
        raise ValueError(
            "SAM line has multiple values of {0}: {1}".format(tag, fields))
    return float(hits[0].split(":")[-1])


def tag_score_zs(fields, tag="AS"):
    """As tag_score but a request for XS reads ZS.  Restates xenomapper.py:193-206."""
    return tag_score(fields, "ZS" if tag == "XS" else tag)


_CIGAR_OP = re.compile(r"([0-9]+)([MIDNSHPX=])")


def cigar_score(fields, tag="AS"):
    """AS synthesised from CIGAR + NM.  Restates xenomapper.py:228-256.

    Any tag other than AS is read with tag_score (:245-246).  With no field
    containing 'NM' the result is -inf (:247-249).  Otherwise
    -6*NM - 5*(#I + #D) - 3*(sum I + sum D) - 2*sum S  (:250-255), an int.
    """
    if tag != "AS":
        return tag_score(fields, tag)
    nm_fields = [opt for opt in fields[11:] if "NM" in opt]
    if not nm_fields:
        return NEG_INF
    score = -6 * int(nm_fields[0].split(":")[-1])
    for length, op in _CIGAR_OP.findall(fields[5]):
        if op == "I" or op == "D":
            score -= 5 + 3 * int(length)
        elif op == "S":
            score -= 2 * int(length)
    return score


# --------------------------------------------------------------------------
# state function
# --------------------------------------------------------------------------

def mapping_state(as1, xs1, as2, xs2, min_score=NEG_INF):
    """Index (0..5) of the mapping state.  Restates xenomapper.py:258-289.

    `not XS` is true for XS == 0 (:278, :285) -- kept.  Raises RuntimeError for
    inputs that fall through every branch (:289; NaN only).
    """
    low1 = as1 <= min_score
    low2 = as2 <= min_score
    if low1 and low2:                                            # :275-276
        return UA
    if as1 > min_score and (low2 or as1 > as2):                  # :277
        return PS if (not xs1 or as1 > xs1) else PM              # :278-281
    if as1 == as2:                                               # :282-283
        return UR
    if as2 > min_score and (low1 or as2 > as1):                  # :284
        return SS if (not xs2 or as2 > xs2) else SM              # :285-288
    raise RuntimeError(
        "Error in processing logic with values {0} ".format((as1, xs1, as2, xs2)))


def mapping_state_name(as1, xs1, as2, xs2, min_score=NEG_INF):
    return STATE_NAMES[mapping_state(as1, xs1, as2, xs2, min_score)]


# --------------------------------------------------------------------------
# pair combination
# --------------------------------------------------------------------------

def bin_liberal(fwd, rev):
    """Highest-priority state of the two mates.  Restates the if/elif chain at
    xenomapper.py:423-448, which is min() under the encoding above."""
    return fwd if fwd < rev else rev


def bin_conservative(fwd, rev):
    """Restates xenomapper.py:521-550."""
    if fwd == UA or rev == UA:                                   # :521
        return UA
    if fwd == UR or rev == UR:                                   # :525
        return UR
    fwd_primary = fwd in (PS, PM)
    rev_primary = rev in (PS, PM)
    if fwd_primary != rev_primary:                               # :526-529
        return UR
    return fwd if fwd < rev else rev                             # :535-550


def bin_of(mode, fwd, rev):
    if mode == MODE_SE:
        return rev
    if mode == MODE_PE_LIBERAL:
        return bin_liberal(fwd, rev)
    return bin_conservative(fwd, rev)


# --------------------------------------------------------------------------
# lock-step reader, headers, summary (host-side text work)
# --------------------------------------------------------------------------

def read_pairs(sam1, sam2, skip_repeated_reads=False):
    """Yield (fields1, fields2) from two SAM streams in lock-step.
    Restates xenomapper.py:95-118: fields are split on any whitespace (:103),
    the walk stops at the first empty line/EOF of either file (:105), names
    must agree (:106), and with skip_repeated_reads each file independently
    advances past further lines carrying the name just yielded (:110-114)."""
    rec1 = sam1.readline().strip("\n").split()
    rec2 = sam2.readline().strip("\n").split()
    while rec1 and rec2:
        assert rec1[0] == rec2[0]
        yield rec1, rec2
        name1, name2 = rec1[0], rec2[0]
        if skip_repeated_reads:
            while rec1 and rec2 and rec1[0] == name1:
                rec1 = sam1.readline().strip("\n").split()
            while rec1 and rec2 and rec2[0] == name2:
                rec2 = sam2.readline().strip("\n").split()
        else:
            rec1 = sam1.readline().strip("\n").split()
            rec2 = sam2.readline().strip("\n").split()


def read_header(sam):
    """Leading '@' lines; leaves the stream at the first record.
    Restates xenomapper.py:36-46 (IndexError on a header-only file, :40-43)."""
    out = []
    while True:
        here = sam.tell()
        text = sam.readline().strip("\n")
        if text[0] != "@":
            sam.seek(here)
            return out
        out.append(text)


VERSION = "1.0.2"


def header_with_pg(header, comment=None):
    """Restates xenomapper.py:120-131."""
    if any(h[0] != "@" for h in header):
        raise ValueError("Incorrect SAM header format :\n{0}".format("\n".join(header)))
    out = list(header)
    chain = ""
    if out[-1][:3] == "@PG":
        ident = [tok for tok in out[-1].split() if tok[:2] == "ID"][0]
        chain = "PP" + ident[2:] + "\t"
    out.append("@PG\tID:Xenomapper\tPN:Xenomapper\t" + chain + "VN:" + VERSION)
    if comment:
        out.append("@CO\t" + comment)
    return out


_BIN_HEADER = (            # (bin, which input header, @CO text)  xenomapper.py:151-173
    (PS, 0, "species specific reads"),
    (SS, 1, "species specific reads"),
    (PM, 0, "species specific multimapping reads"),
    (SM, 1, "species specific multimapping reads"),
    (UA, 0, "reads that could not be assigned"),
    (UR, 0, "reads that could not be resolved"),
)


def write_headers(sam1, sam2, outs):
    """outs: list of six file-likes (or None) indexed by state.  Restates
    xenomapper.py:133-174 (primary_specific is written unconditionally)."""
    headers = (read_header(sam1), read_header(sam2))
    for b, which, comment in _BIN_HEADER:
        if b == PS or outs[b]:
            print("\n".join(header_with_pg(headers[which], comment)), file=outs[b])


def summary_text(category_counts):
    """Restates xenomapper.py:558-566 (returns the text instead of printing)."""
    rows = ["-" * 80, "Read Count Category Summary\n",
            "|       {0:45s}|     {1:10s}  |".format("Category", "Count"),
            "|:" + "-" * 50 + ":|:" + "-" * 15 + ":|"]
    for key in sorted(category_counts):
        rows.append("|  {0:50s}|{1:15d}  |".format(str(key), category_counts[key]))
    rows.append("")
    return "\n".join(rows) + "\n"


# --------------------------------------------------------------------------
# main loops
# --------------------------------------------------------------------------

class Result(object):
    """What one run of a main loop produced, in index form."""

    def __init__(self):
        self.units = []       # (record index of the unit's last record, fwd, rev, bin)
        self.counts = Counter()
        self.n_records = 0

    def named_counts(self, paired):
        out = Counter()
        for key, val in self.counts.items():
            if paired:
                out[(STATE_NAMES[key[0]], STATE_NAMES[key[1]])] = val
            else:
                out[STATE_NAMES[key]] = val
        return out


def _emit(outs, b, lines1, lines2):
    """Write one unit.  lines1/lines2 are the unit's records from file 1 / 2.
    Restates the emission rules of xenomapper.py:332-350, :423-448, :521-550:
    bins 0,2,5 take file-1 lines, bins 1,3 take file-2 lines, bin 4 takes all
    file-1 lines followed by all file-2 lines; a bin without a sink is only counted.
    """
    sink = outs[b]
    if not sink:
        return
    if b in (PS, PM, UA):
        chosen = lines1
    elif b in (SS, SM):
        chosen = lines2
    else:
        chosen = list(lines1) + list(lines2)
    for rec in chosen:
        print("\t".join(rec), file=sink)


def run_single_end(readpairs, outs, min_score=NEG_INF, scorer=tag_score):
    """Restates xenomapper.py:291-352.  outs is indexed by state (0..5)."""
    res = Result()
    for i, (rec1, rec2) in enumerate(readpairs):
        assert rec1[0] == rec2[0]
        state = mapping_state(scorer(rec1, tag="AS"), scorer(rec1, tag="XS"),
                              scorer(rec2, tag="AS"), scorer(rec2, tag="XS"), min_score)
        res.counts[state] += 1
        res.units.append((i, state, state, state))
        res.n_records = i + 1
        if state == UR:                       # :347-350 one line from each file
            _emit(outs, UR, [rec1], [rec2])
        else:
            _emit(outs, state, [rec1], [rec2])
    return res


def run_paired_end(readpairs, outs, min_score=NEG_INF, scorer=tag_score, conservative=False):
    """Restates xenomapper.py:354-454 (liberal) and :456-556 (conservative).

    A unit is every record whose name equals the previous record's name
    (:402-405); `previous` always advances (:403-404, :451-452) so three equal
    names in a row give two overlapping units."""
    res = Result()
    prev1 = prev2 = None
    for i, (rec1, rec2) in enumerate(readpairs):
        assert rec1[0] == rec2[0]
        res.n_records = i + 1
        if not prev1 or prev1[0] != rec1[0]:
            prev1, prev2 = rec1, rec2
            continue
        # all eight tags are read before either state is evaluated (:408-415, then :417-418), so a malformed
        # tag of the second mate raises before a NaN of the first one can
        p_scores = (scorer(prev1, tag="AS"), scorer(prev1, tag="XS"), scorer(prev2, tag="AS"), scorer(prev2, tag="XS"))
        c_scores = (scorer(rec1, tag="AS"), scorer(rec1, tag="XS"), scorer(rec2, tag="AS"), scorer(rec2, tag="XS"))
        fwd = mapping_state(*p_scores, min_score)
        rev = mapping_state(*c_scores, min_score)
        res.counts[(fwd, rev)] += 1
        b = bin_conservative(fwd, rev) if conservative else bin_liberal(fwd, rev)
        res.units.append((i, fwd, rev, b))
        _emit(outs, b, [prev1, rec1], [prev2, rec2])
        prev1, prev2 = rec1, rec2
    return res


# --------------------------------------------------------------------------
# column form (what the device sees) -- used to check the C oracle and the HIP path
# --------------------------------------------------------------------------

def classify_columns(mode, as1, xs1, as2, xs2, unit_flags, min_score=NEG_INF):
    """Pure-Python loop over score columns.  Returns (code list, Counter).

    code[i] = 0xFF when record i closes no unit, else state (MODE_SE) or
    fwd*8 + rev (paired modes); counts are keyed like the code.
    """
    n = len(as1)
    code = [0xFF] * n
    counts = Counter()
    for i in range(n):
        if not unit_flags[i]:
            continue
        rev = mapping_state(as1[i], xs1[i], as2[i], xs2[i], min_score)
        if mode == MODE_SE:
            code[i] = rev
        else:
            if i == 0:
                continue
            fwd = mapping_state(as1[i - 1], xs1[i - 1], as2[i - 1], xs2[i - 1], min_score)
            code[i] = fwd * 8 + rev
        counts[code[i]] += 1
    return code, counts
