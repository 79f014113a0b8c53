/*
 * CPU oracle for the xenomapper classification hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar C restatement of the reference algorithm (genomematt/xenomapper v1.0.2) on
 * the column (structure-of-arrays) form the device consumes.  It is the checker for
 * the HIP path at sizes a Python loop cannot reach, and the "port" CPU baseline of
 * bench.py.  Nothing under xenomapper_amd/ links, loads or calls this file.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks these functions against
 * the reference's own known-answer rows (xenomapper/tests/test_xenomapper.py:165-183,
 * :215-227) and against golden vectors recorded from the imported reference
 * (tools/make_golden.py -> tests/golden/).
 *
 * State encoding (priority order of xenomapper.py:364-367):
 *   0 primary_specific 1 secondary_specific 2 primary_multi 3 secondary_multi
 *   4 unresolved 5 unassigned ; 6 = "fell through every branch" (xenomapper.py:289, NaN only)
 *
 * Integer columns: INT32_MIN stands for the reference's float('-inf') (tag absent,
 * xenomapper.py:187-188).  min_score is passed as floor(min_score) clamped to int32
 * (-inf -> INT32_MIN, +inf -> INT32_MAX): for integer a, a > m  <=>  a > floor(m).
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#define XMO_NO_UNIT 0xFFu

/* xenomapper.py:258-289, integer columns */
int xmo_state_i32(int32_t as1, int32_t xs1, int32_t as2, int32_t xs2, int32_t m)
{
    if (as1 <= m && as2 <= m) return 5;                               /* :275-276 */
    if (as1 > m && (as2 <= m || as1 > as2))                           /* :277 */
        return (xs1 == 0 || as1 > xs1) ? 0 : 2;                       /* :278-281 */
    if (as1 == as2) return 4;                                         /* :282-283 */
    if (as2 > m && (as1 <= m || as2 > as1))                           /* :284 */
        return (xs2 == 0 || as2 > xs2) ? 1 : 3;                       /* :285-288 */
    return 6;                                                         /* :289 */
}

/* xenomapper.py:258-289, the reference's own arithmetic type (Python float = binary64) */
int xmo_state_f64(double as1, double xs1, double as2, double xs2, double m)
{
    if (as1 <= m && as2 <= m) return 5;
    if (as1 > m && (as2 <= m || as1 > as2))
        return (xs1 == 0.0 || as1 > xs1) ? 0 : 2;
    if (as1 == as2) return 4;
    if (as2 > m && (as1 <= m || as2 > as1))
        return (xs2 == 0.0 || as2 > xs2) ? 1 : 3;
    return 6;
}

/* xenomapper.py:423-448 (liberal = highest priority of the two) and :521-550 (conservative) */
int xmo_bin(int mode, int fwd, int rev)
{
    int lo = fwd < rev ? fwd : rev;
    if (mode == 0) return rev;
    if (fwd > 5 || rev > 5) return 6;
    if (mode == 1) return lo;
    if (fwd == 5 || rev == 5) return 5;                               /* :521 */
    if (fwd == 4 || rev == 4) return 4;                               /* :525 */
    if ((fwd & 1) != (rev & 1)) return 4;                             /* :526-529 species discordant */
    return lo;                                                        /* :535-550 */
}

static int unit_bit(const uint64_t *bits, uint64_t i)
{
    return (int)((bits[i >> 6] >> (i & 63)) & 1u);
}

/*
 * Main-loop core on columns.  mode 0 = main_single_end (:321-330), 1 = main_paired_end
 * (:398-420), 2 = conservative_main_paired_end (:498-520).  unit bit i set means: record i
 * closes a unit (paired: name[i] == name[i-1], :402; single: record i was yielded).
 * code[i] = 0xFF | state | fwd*8+rev ; counts[code]++.
 */
#define XMO_CLASSIFY(NAME, T, STATEFN)                                                     \
void NAME(int mode, uint64_t n, const T *as1, const T *xs1, const T *as2, const T *xs2,    \
          const uint64_t *unit_bits, T m, uint8_t *code, uint64_t counts[64])              \
{                                                                                          \
    for (int k = 0; k < 64; ++k) counts[k] = 0;                                            \
    int prev = 0;                                                                          \
    for (uint64_t i = 0; i < n; ++i) {                                                     \
        int s = STATEFN(as1[i], xs1[i], as2[i], xs2[i], m);                                \
        unsigned c = XMO_NO_UNIT;                                                          \
        if (unit_bit(unit_bits, i)) {                                                      \
            if (mode == 0) c = (unsigned)s;                                                \
            else if (i > 0) c = (unsigned)(prev * 8 + s);                                  \
        }                                                                                  \
        code[i] = (uint8_t)c;                                                              \
        if (c != XMO_NO_UNIT) counts[c]++;                                                 \
        prev = s;                                                                          \
    }                                                                                      \
}

XMO_CLASSIFY(xmo_classify_i32, int32_t, xmo_state_i32)
XMO_CLASSIFY(xmo_classify_f64, double, xmo_state_f64)

/*
 * Stable split of unit indices by bin (the order the reference appends lines to each of
 * its six files, xenomapper.py:332-350 / :423-448 / :521-550).  idx receives, bin after bin,
 * the record index of every unit; bin_offsets[b]..bin_offsets[b+1] delimits bin b
 * (b = 0..5, slot 6 collects state-6 units, bin_offsets[7] = number of units).
 */
void xmo_compact(int mode, uint64_t n, const uint8_t *code, uint32_t *idx, uint64_t bin_offsets[8])
{
    uint64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < n; ++i) {
        if (code[i] == XMO_NO_UNIT) continue;
        cnt[xmo_bin(mode, code[i] >> 3, code[i] & 7)]++;
    }
    uint64_t run[8];
    uint64_t acc = 0;
    for (int b = 0; b < 7; ++b) { bin_offsets[b] = acc; run[b] = acc; acc += cnt[b]; }
    bin_offsets[7] = acc;
    for (uint64_t i = 0; i < n; ++i) {
        if (code[i] == XMO_NO_UNIT) continue;
        idx[run[xmo_bin(mode, code[i] >> 3, code[i] & 7)]++] = (uint32_t)i;
    }
}

/*
 * xenomapper.py:228-256 on packed CIGAR (BAM encoding len<<4|op, ops "MIDNSHP=X" = 0..8):
 * NM absent (INT32_MIN) -> absent (:247-249); else -6*NM - 5*(#I+#D) - 3*(sumI+sumD) - 2*sumS.
 * Returns the number of records whose score does not fit the int32 column.
 */
uint64_t xmo_cigar_scores(uint64_t n, const int32_t *nm, const uint32_t *cig_off,
                          const uint32_t *cig_oplen, int32_t *as_out)
{
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (nm[i] == INT32_MIN) { as_out[i] = INT32_MIN; continue; }
        int64_t s = -6 * (int64_t)nm[i];
        for (uint32_t k = cig_off[i]; k < cig_off[i + 1]; ++k) {
            uint32_t op = cig_oplen[k] & 15u;
            int64_t len = (int64_t)(cig_oplen[k] >> 4);
            if (op == 1 || op == 2) s -= 5 + 3 * len;
            else if (op == 4) s -= 2 * len;
        }
        if (s <= (int64_t)INT32_MIN || s > (int64_t)INT32_MAX) { bad++; s = s < 0 ? (int64_t)INT32_MIN + 1 : INT32_MAX; }
        as_out[i] = (int32_t)s;
    }
    return bad;
}

/*
 * xenomappability: Mappability.single_end_to_paired for one chromosome, mappability.py:94-124.
 * out[i] = 1.0 where track[i] == 1, else sum_j track[i+j] * density[j] for j < min(m, n - i), accumulated left to
 * right with a separately rounded multiply and add (Python's `result += a * b`).  The Makefile compiles with
 * -ffp-contract=off (gcc ignores the STDC FP_CONTRACT pragma), and the volatile temporaries keep the two roundings
 * apart under any flags.
 */
void xmo_mate_correlate(uint64_t n, const double *track, uint64_t m, const double *density, double *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        if (track[i] == 1.0) { out[i] = 1.0; continue; }
        volatile double acc = 0.0;
        for (uint64_t j = 0; j < m && i + j < n; ++j) {
            volatile double prod = track[i + j] * density[j];
            acc = acc + prod;
        }
        out[i] = acc;
    }
}
