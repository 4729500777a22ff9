// jinc_lut.h -- windowed-jinc lookup table of the EWA plan (host, init-time).
// Reproduces Lut::InitLut / Lut::GetFactor of the reference bit for bit
// (/root/reference/src/JincResize.cpp:201-282); see jinc_lut.cpp for the per-branch citations.
#pragma once
#include <array>

namespace jinc {

constexpr int kLutSamples = 1024;  // ref :795

// EWA radius for a tap count 1..16 = the tap-th zero of jinc (ref :84-102, :794).
double jinc_radius(int tap);

// jinc(sqrt(x2)) = 2*J1(pi*sqrt(x2)) / (pi*sqrt(x2))  (ref :201-245)
double jinc_of_sqr(double x2);

struct JincLut {
    std::array<double, kLutSamples> v;
    // ref :277-282
    float factor(int index) const { return index >= kLutSamples ? 0.f : static_cast<float>(v[index]); }
};

// ref :265-275
void build_lut(JincLut& lut, double radius, double blur);

}  // namespace jinc
