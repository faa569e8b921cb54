/* Compares craytracer_amd/csrc/crt_libm.h with the platform libm on every float in [0, 2*pi]
 * (and a band beyond); prints the number of differing results.  Built and run by tests/test_libm.py. */
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
        const float a = sinf(f), b = crt_libm_sinf(f), c = cosf(f), d = crt_libm_cosf(f);
        bad += (memcmp(&a, &b, 4) != 0) + (memcmp(&c, &d, 4) != 0);
        n++;
    }
    printf("%llu %llu\n", n, bad);
    return bad != 0;
}
