/*
 * classify_pairs.c -- the C ABI from plain C: one fused call on a handful of read pairs.
 *
 *   gcc -std=c99 -I include examples/classify_pairs.c -L xenomapper_amd -l:libxenomapper_hip.so \
 *       -Wl,-rpath,$PWD/xenomapper_amd -o classify_pairs && ./classify_pairs
 *
 * Six records = three read pairs (mates adjacent, the second mate closes the unit), scores as Bowtie2 would give them
 * (XM_ABSENT = tag absent = the reference's float('-inf')).  Prints category_counts and the six bins, which a caller
 * maps back to SAM lines as the reference's main_paired_end does (xenomapper.py:423-448).
 */
#include <inttypes.h>
#include <stdio.h>

#include "xenomapper_hip.h"

int main(void)
{
    /* pair 0: both mates better in species 1, unique   -> primary_specific
     * pair 1: both mates better in species 2, repeats  -> secondary_multi
     * pair 2: nothing aligned anywhere                 -> unassigned */
    const int32_t as1[6] = {200, 198, 80, 90, XM_ABSENT, XM_ABSENT};
    const int32_t xs1[6] = {150, XM_ABSENT, XM_ABSENT, 70, XM_ABSENT, XM_ABSENT};
    const int32_t as2[6] = {120, XM_ABSENT, 190, 188, XM_ABSENT, XM_ABSENT};
    const int32_t xs2[6] = {XM_ABSENT, XM_ABSENT, 190, 188, XM_ABSENT, XM_ABSENT};
    const uint64_t unit_bits[1] = {0x2A};                  /* records 1, 3, 5 close a unit */
    static const char *state[6] = {"primary_specific", "secondary_specific", "primary_multi", "secondary_multi",
                                   "unresolved", "unassigned"};
    xm_ctx *ctx = NULL;
    int rc = xm_ctx_create(0, &ctx);
    if (rc != XM_OK) {
        fprintf(stderr, "xm_ctx_create: %s (%s)\n", xm_strerror(rc), xm_last_hip_error(NULL));
        return 2;                                          /* no gfx950 device: there is no CPU fallback */
    }
    uint8_t code[6];
    uint32_t idx[6];
    uint64_t bin_offsets[8], counts[64];
    rc = xm_classify_compact(ctx, XM_MODE_PE_LIBERAL, 6, as1, xs1, as2, xs2, unit_bits, XM_ABSENT /* min_score = -inf */,
                             code, idx, bin_offsets, counts);
    if (rc != XM_OK) {
        fprintf(stderr, "xm_classify_compact: %s (%s)\n", xm_strerror(rc), xm_last_hip_error(ctx));
        return 1;
    }
    for (int c = 0; c < 64; ++c)
        if (counts[c]) printf("count (%s, %s) = %" PRIu64 "\n", state[c >> 3], state[c & 7], counts[c]);
    for (int b = 0; b < 6; ++b)
        for (uint64_t k = bin_offsets[b]; k < bin_offsets[b + 1]; ++k)
            printf("bin %s: records %u and %u\n", state[b], idx[k] - 1, idx[k]);
    for (int i = 0; i < 6; ++i) printf("code[%d] = 0x%02X\n", i, code[i]);
    return xm_ctx_destroy(ctx) == XM_OK ? 0 : 1;
}
