// plp_dd.hpp -- double-double arithmetic (an unevaluated sum hi + lo of two doubles, ~106-bit significand) for the
// verifier and the careful LP engine (plp_verify.hpp).  Error-free transformations on v_fma_f64 / v_add_f64: two_sum
// (Knuth), two_prod (one fma).  Built with -ffp-contract=off like the rest of the library: every fma below is explicit and
// no a*b+c is fused behind our back (a fused two_sum would be wrong).  Compiles for the host too (tests/cabi/verify_host.cpp).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PLP_HD __host__ __device__ __forceinline__
#else
#define PLP_HD inline
#endif

namespace plp {

struct dd {
    double hi, lo;
};

PLP_HD dd dd_make(double a) { return dd{a, 0.0}; }
PLP_HD dd two_sum(double a, double b) {
    const double s = a + b;
    const double bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
PLP_HD dd quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return dd{s, b - (s - a)};
}
PLP_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, fma(a, b, -p)};
}
PLP_HD dd dd_add(dd x, dd y) {
    dd s = two_sum(x.hi, y.hi);
    const dd t = two_sum(x.lo, y.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
PLP_HD dd dd_neg(dd x) { return dd{-x.hi, -x.lo}; }
PLP_HD dd dd_sub(dd x, dd y) { return dd_add(x, dd_neg(y)); }
PLP_HD dd dd_mul(dd x, dd y) {
    dd p = two_prod(x.hi, y.hi);
    p.lo += x.hi * y.lo + x.lo * y.hi;
    return quick_two_sum(p.hi, p.lo);
}
PLP_HD dd dd_mul_d(dd x, double y) {
    dd p = two_prod(x.hi, y);
    p.lo += x.lo * y;
    return quick_two_sum(p.hi, p.lo);
}
PLP_HD dd dd_div(dd x, dd y) {
    const double q1 = x.hi / y.hi;
    dd r = dd_sub(x, dd_mul_d(y, q1));
    const double q2 = r.hi / y.hi;
    r = dd_sub(r, dd_mul_d(y, q2));
    const double q3 = r.hi / y.hi;
    const dd q = quick_two_sum(q1, q2);
    return dd_add(q, dd_make(q3));
}
// z - x * y
PLP_HD dd dd_fnma(dd x, dd y, dd z) { return dd_sub(z, dd_mul(x, y)); }
PLP_HD bool dd_lt(dd a, dd b) { return (a.hi < b.hi) | ((a.hi == b.hi) & (a.lo < b.lo)); }
PLP_HD bool dd_gt(dd a, dd b) { return dd_lt(b, a); }
PLP_HD bool dd_eq(dd a, dd b) { return (a.hi == b.hi) & (a.lo == b.lo); }
PLP_HD dd dd_abs(dd a) { return (a.hi < 0.0 || (a.hi == 0.0 && a.lo < 0.0)) ? dd_neg(a) : a; }
PLP_HD bool dd_gt_d(dd a, double b) { return (a.hi > b) | ((a.hi == b) & (a.lo > 0.0)); }
PLP_HD bool dd_lt_d(dd a, double b) { return (a.hi < b) | ((a.hi == b) & (a.lo < 0.0)); }
PLP_HD double dd_to_double(dd a) { return a.hi + a.lo; }

}  // namespace plp
