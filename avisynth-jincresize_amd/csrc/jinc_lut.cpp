// jinc_lut.cpp -- see jinc_lut.h.  Must be compiled with -ffp-contract=off and without -ffast-math:
// the table has to equal the reference's double-precision values exactly, because the plan's
// coefficients (and therefore every output sample) are derived from float(lut[i]).
#include "jinc_lut.h"

#include <cmath>

namespace jinc {
namespace {

// Maclaurin coefficients of jinc in t = x^2 (data from ref :49-82).
constexpr double kSeries[31] = {
    1.0,
    -1.23370055013616982735431137,
    0.507339015802096027273126733,
    -0.104317403816764804365258186,
    0.0128696438477519721233840271,
    -0.00105848577966854543020422691,
    6.21835470803998638484476598e-05,
    -2.73985272294670461142756204e-06,
    9.38932725442064547796003405e-08,
    -2.57413737759717407304931036e-09,
    5.77402672521402031756429343e-11,
    -1.07930605263598241754572977e-12,
    1.70710316782347356046974552e-14,
    -2.31434518382749184406648762e-16,
    2.71924659665997312120515390e-18,
    -2.79561335187943028518083529e-20,
    2.53599244866299622352138464e-22,
    -2.04487273140961494085786452e-24,
    1.47529860450204338866792475e-26,
    -9.57935105257523453155043307e-29,
    5.62764317309979254140393917e-31,
    -3.00555258814860366342363867e-33,
    1.46559362903641161989338221e-35,
    -6.55110024064596600335624426e-38,
    2.69403199029404093412381643e-40,
    -1.02265499954159964097119923e-42,
    3.59444454568084324694180635e-45,
    -1.17313973900539982313119019e-47,
    3.56478606255557746426034301e-50,
    -1.01100655781438313239513538e-52,
    2.68232117541264485328658605e-55,
};

// Zeros of jinc (data from ref :84-102).
constexpr double kZeros[16] = {
    1.2196698912665045,  2.2331305943815286,  3.2383154841662362,  4.2410628637960699,
    5.2427643768701817,  6.2439216898644877,  7.2447598687199570,  8.2453949139520427,
    9.2458926849494673,  10.246293348754916,  11.246622794877883,  12.246898461138105,
    13.247132522181061,  14.247333735806849,  15.247508563037300,  16.247661874700962,
};

// Series branch: the number of terms grows with the argument (ref :203-230).
struct SeriesBranch {
    double below;
    int terms;
};
constexpr SeriesBranch kBranches[4] = {{1.49, 16}, {4.97, 21}, {10.49, 26}, {17.99, 31}};

// P(z)/Q(z), evaluated in z for z <= 1 and in 1/z otherwise (ref :110-140).
double ratio_of_polys(const double (&p)[7], const double (&q)[7], double z) {
    double a, b;
    if (z <= 1.0) {
        a = p[6];
        b = q[6];
        for (int k = 5; k >= 0; --k) {
            a *= z;
            b *= z;
            a += p[k];
            b += q[k];
        }
    } else {
        z = 1.0 / z;
        a = p[0];
        b = q[0];
        for (int k = 1; k < 7; ++k) {
            a *= z;
            b *= z;
            a += p[k];
            b += q[k];
        }
    }
    return a / b;
}

// Hankel-type large-argument form used only for the 8-tap band (ref :148-198).
double jinc_large_arg(double x2) {
    static constexpr double pc[7] = {-4.4357578167941278571e+06, -9.9422465050776411957e+06, -6.6033732483649391093e+06,
                                     -1.5235293511811373833e+06, -1.0982405543459346727e+05, -1.6116166443246101165e+03,
                                     0.0};
    static constexpr double qc[7] = {-4.4357578167941278568e+06, -9.9341243899345856590e+06, -6.5853394797230870728e+06,
                                     -1.5118095066341608816e+06, -1.0726385991103820119e+05, -1.4550094401904961825e+03,
                                     1.0};
    static constexpr double ps[7] = {3.3220913409857223519e+04, 8.5145160675335701966e+04, 6.6178836581270835179e+04,
                                     1.8494262873223866797e+04, 1.7063754290207680021e+03, 3.5265133846636032186e+01,
                                     0.0};
    static constexpr double qs[7] = {7.0871281941028743574e+05, 1.8194580422439972989e+06, 1.4194606696037208929e+06,
                                     4.0029443582266975117e+05, 3.7890229745772202641e+04, 8.6383677696049909675e+02,
                                     1.0};
    const double y2 = M_PI * M_PI * x2;
    const double xp = std::sqrt(y2);
    const double inv = 64.0 / y2;
    // glibc's sin() and sincos() differ by one ulp at some arguments, and GCC at -O2 and above merges the
    // reference's sin(xp)/cos(xp) pair into one sincos() call (clang does not).  The reference's CMake Release build
    // with GCC is the parity target, so sincos() is called explicitly (through a volatile pointer: GCC -O0 lowers the
    // builtin back into sin() + cos()): the table no longer depends on the compiler that builds this file
    // (DESIGN.md section 2).
    void (*volatile glibc_sincos)(double, double*, double*) = ::sincos;
    double s, c;
    glibc_sincos(xp, &s, &c);
    const double amp = std::sqrt(xp / M_PI) * 2.0 / y2;
    const double rc = ratio_of_polys(pc, qc, inv);
    const double rs = ratio_of_polys(ps, qs, inv);
    return amp * (rc * (s - c) + (8.0 / xp) * rs * (s + c));
}

// std::cyl_bessel_j form (ref :231-235, :240-244): libstdc++'s routine, as the reference binds it.
double jinc_bessel(double x2) {
    const double x = M_PI * std::sqrt(x2);
    return 2.0 * std::cyl_bessel_j(1, x) / x;
}

// ref :247-256
double windowed(double x2, double blur2, double radius2) {
    if (blur2 > 0.0) x2 /= blur2;
    return x2 < radius2 ? jinc_of_sqr(x2) : 0.0;
}

}  // namespace

double jinc_radius(int tap) { return kZeros[tap - 1]; }

double jinc_of_sqr(double x2) {
    for (const SeriesBranch& b : kBranches) {
        if (x2 < b.below) {
            double acc = 0.0;
            for (int j = b.terms; j > 0; --j) acc = acc * x2 + kSeries[j - 1];
            return acc;
        }
    }
    if (x2 >= 52.57 && x2 < 68.07) return jinc_large_arg(x2);  // ref :236-239
    return jinc_bessel(x2);
}

void build_lut(JincLut& lut, double radius, double blur) {
    constexpr double kFirstZeroSqr = 1.48759464366204680005356;  // ref :258
    const double radius2 = radius * radius;
    const double blur2 = blur * blur;
    for (int i = 0; i < kLutSamples; ++i) {
        const double t = i / (kLutSamples - 1.0);
        lut.v[i] = windowed(radius2 * t, blur2, radius2) * windowed(kFirstZeroSqr * t, 1.0, radius2);
    }
}

}  // namespace jinc
