// Host build of xenomapper_amd/csrc/xm_fmtg.h (the device SAM printer's "%g" for binary32) against the C library's own
// snprintf("%g") -- TEST INFRASTRUCTURE (tests/test_fmtg_host.py).  Arguments: <stride> <random count> <seed> <integer step>.
// Checked: every exponent with a strided walk of the mantissas, the edges of every binade, every integer in [10^6 - 10, 2^24]
// and every k + 0.5, k / 8 around the 6-digit boundary (exact ties: round-half-even of the exact value), decimal literals,
// subnormals, zeros, infinities, NaNs of both signs, and random bit patterns.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../xenomapper_amd/csrc/xm_fmtg.h"

static unsigned long long checked = 0, bad = 0;

static void check(uint32_t bits)
{
    float f;
    memcpy(&f, &bits, 4);
    char want[64], got[32];
    snprintf(want, sizeof want, "%g", (double)f);
    const xmfmt::Text16 t = xmfmt::fmt_g_f32(bits);
    uint32_t n = t.n > 16u ? 16u : t.n;
    for (uint32_t k = 0; k < n; ++k) got[k] = (char)t.at(k);
    got[n] = 0;
    ++checked;
    if (t.n > 16u || strcmp(want, got) != 0) {
        if (bad++ < 20) printf("MISMATCH bits %08x: want \"%s\" got \"%s\"\n", bits, want, got);
    }
}

static void check_f(float f) { uint32_t b; memcpy(&b, &f, 4); check(b); check(b ^ 0x80000000u); }

int main(int argc, char **argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 10) : 4099u;
    const unsigned long long n_random = argc > 2 ? strtoull(argv[2], nullptr, 10) : 2000000ull;
    uint64_t rng = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1ull;
    const uint32_t step = argc > 4 ? (uint32_t)strtoul(argv[4], nullptr, 10) : 1u;       // of the walk over the exact integers
    for (uint32_t ex = 0; ex < 256u; ++ex) {
        for (uint32_t fr = 0; fr < (1u << 23); fr += stride) { check((ex << 23) | fr); check(0x80000000u | (ex << 23) | fr); }
        for (uint32_t k = 0; k < 64u; ++k) { check((ex << 23) | k); check((ex << 23) | (0x7FFFFFu - k)); }
    }
    for (uint32_t k = 999990u; k <= (1u << 24); k += (k < 1100000u ? 1u : step)) check_f((float)k);
    for (uint32_t k = 99990u; k <= 1000010u; ++k) { check_f((float)k + 0.5f); check_f((float)k + 0.25f); }
    for (uint32_t k = 0; k <= 8u * 100010u; ++k) check_f((float)k / 8.0f);
    for (uint32_t k = 0; k <= 2000000u; ++k) { check_f((float)k * 1e-4f); check_f((float)k * 1e-7f); check_f((float)k * 1e-10f); check_f((float)k * 1e3f); }
    const float lits[] = {0.0123f, 0.1f, 0.5f, 1.5f, -2.25f, 1e10f, 3.0e-5f, 123456.0f, 1234567.0f, 0.0001f, 0.00001f, 9.99999e-5f, 999999.5f,
                          999999.4f, 0.00099999949f, 3.4028235e38f, 1.17549435e-38f, 1.4e-45f, 1e-40f, 100000.0f, 1e5f, 1e6f, 1e-4f, 1e-5f};
    for (float f : lits) check_f(f);
    for (unsigned long long i = 0; i < n_random; ++i) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        check((uint32_t)(rng >> 32));
    }
    printf("fmt_g_f32: %llu values, %llu mismatches\n", checked, bad);
    return bad ? 1 : 0;
}
