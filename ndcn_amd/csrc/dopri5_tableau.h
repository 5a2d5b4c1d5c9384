// Dormand-Prince 5(4): dopri5.py:11-36, written as the same rational expressions (evaluated in double, rounded to float where the
// reference multiplies them with a float32 0-d tensor).  Shared by the device-resident solver (solver.hip) and the training tape
// (tape.hip).
#pragma once
#include <math.h>

namespace ndcn {

static const double kAlpha[6] = {1. / 5, 3. / 10, 4. / 5, 8. / 9, 1., 1.};
static const double kBeta[6][6] = {
    {1. / 5},
    {3. / 40, 9. / 40},
    {44. / 45, -56. / 15, 32. / 9},
    {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729},
    {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656},
    {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84},
};
static const double kCErr[7] = {
    35. / 384 - 1951. / 21600, 0, 500. / 1113 - 22642. / 50085, 125. / 192 - 451. / 720,
    -2187. / 6784 - -12231. / 42400, 11. / 84 - 649. / 6300, -1. / 60.,
};
static const double kCMid[7] = {
    6025192743. / 30085553152. / 2, 0, 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
    187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2,
};

static inline double nan_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }
static inline double nan_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }

}  // namespace ndcn
