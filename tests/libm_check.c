/* Compares craytracer_amd/csrc/crt_libm.h with the platform libm on every float in [-6.5, 6.5]
 * ([0, 2*pi] and a band beyond for the sampling maps; the negative half for the anisotropic Beckmann
 * azimuth, microfacet.cpp:121-124); prints the number of differing results.  Built and run by tests/test_libm.py. */
#include <math.h>
#include <stdio.h>
#include <stdbool.h>
#include "../craytracer_amd/csrc/crt_libm.h"
int main(void)
{
    const float hi = 6.5f; uint32_t uh; memcpy(&uh, &hi, 4);
    unsigned long long bad = 0, n = 0;
    for (uint32_t u = 0; u <= uh; u++) {
        float f; memcpy(&f, &u, 4);
        for (int neg = 0; neg < 2; neg++) {
            const float g = neg ? -f : f;
            const float a = sinf(g), b = crt_libm_sinf(g), c = cosf(g), d = crt_libm_cosf(g);
            bad += (memcmp(&a, &b, 4) != 0) + (memcmp(&c, &d, 4) != 0);
            n++;
        }
    }
    printf("%llu %llu\n", n, bad);
    return bad != 0;
}
