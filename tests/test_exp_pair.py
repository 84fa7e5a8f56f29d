"""exp_pair.h (e^x and e^-x through one reduction and polynomial, the shortwave solver's exponentials) against long double.

The header is plain C: the same text the device compiles is compiled here with gcc and run over 4e6 arguments of the
solver's domain (|x| <= 700, grtcode_config.h:41) -- uniform, near zero, and next to the reduction's breakpoints
(n + 1/2) ln 2, where |r| is largest."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "grtcode_amd", "csrc", "hip", "exp_pair.h")

PROGRAM = r"""
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "%s"
static double ulps(double got, long double want)
{
    int e;
    frexpl(want, &e);
    return (double)(fabsl((long double)got - want)/ldexpl(1.0L, e - 53));
}
int main(void)
{
    double worst = 0.;
    srand48(7);
    for (long i = 0; i < 4000000; ++i)
    {
        double x;
        switch (i %% 4)
        {
            case 0: x = (drand48()*2 - 1)*700.; break;
            case 1: x = (drand48()*2 - 1)*2.; break;
            case 2: x = (drand48()*2 - 1)*1e-3; break;
            default: x = (floor((drand48()*2 - 1)*1000) + 0.5)*0.6931471805599453 + (drand48() - 0.5)*1e-9; break;
        }
        double p, m;
        grt_exp_pair(x, &p, &m);
        double const a = ulps(p, expl((long double)x)), b = ulps(m, expl(-(long double)x));
        worst = a > worst ? a : worst;
        worst = b > worst ? b : worst;
        double const one = ulps(grt_exp(x), expl((long double)x));         /* the single exponential: the same parts */
        worst = one > worst ? one : worst;
        if (grt_exp(x) != p) worst = 99.;                                   /* ... the pair's e^x, bit for bit */
    }
    double p, m, pn, mn;
    grt_exp_pair(0., &p, &m);
    grt_exp_pair(NAN, &pn, &mn);
    int const limits = grt_exp(800.) == INFINITY && grt_exp(-800.) == 0. && grt_exp(INFINITY) == INFINITY && grt_exp(-INFINITY) == 0.
                       && grt_exp(NAN) != grt_exp(NAN) && grt_exp(0.) == 1.;
    printf("%%.4f %%d %%d %%d\n", worst, p == 1. && m == 1., pn != pn && mn != mn, limits);
    return 0;
}
"""


def test_exp_pair_is_within_about_one_unit_in_the_last_place(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(PROGRAM % HEADER)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    worst, one_at_zero, nan_in_nan_out, limits = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    print("worst error of e^x / e^-x:", worst, "units in the last place")
    assert float(worst) < 1.06
    assert one_at_zero == "1" and nan_in_nan_out == "1"
    assert limits == "1"            # grt_exp beyond its clamp: inf / 0 as exp gives, NaN for NaN
