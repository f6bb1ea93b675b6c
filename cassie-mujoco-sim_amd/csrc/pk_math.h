/*
 * pk_math.h -- small vector / quaternion / spatial algebra
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_MATH_H
#define CASSIE_PK_MATH_H

namespace ck {

/* ------------------------------------------------------------ small math --- */
WV_DEVICE double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
WV_DEVICE void cross3(double *r, const double *a, const double *b) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE double normalize3(double *a) {
    double n = sqrt(dot3(a, a));
    if (n < CM_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
    else { double s = 1.0 / n; a[0] *= s; a[1] *= s; a[2] *= s; }
    return n;
}
WV_DEVICE void normalize4(double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else { double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
/* the same through the hardware reciprocal-square-root estimate and two Newton steps (kinematics of the
 * compile-time-topology kernels: no IEEE square root + division sequence on the stage's chain) */
WV_DEVICE void normalize4_fast(double *q) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 < CM_MINVAL * CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else {
        double y = wv::rsq_estimate(n2);
        y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
        y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
        q[0] *= y; q[1] *= y; q[2] *= y; q[3] *= y;
    }
}
/* normalize3 the same way; returns the norm */
WV_DEVICE double normalize3_fast(double *a) {
    const double n2 = dot3(a, a);
    if (n2 < CM_MINVAL * CM_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return sqrt(n2); }
    double y = wv::rsq_estimate(n2);
    y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
    a[0] *= y; a[1] *= y; a[2] *= y;
    const double n = n2 * y;
    return fma(0.5 * y, fma(-n, n, n2), n); /* one Newton step on the norm itself: n2 * y carries y's rounding */
}
/* sin and cos of a joint's half angle.  |x| < 2^19: three-part Cody-Waite reduction by pi/2 (the first two parts carry
 * 33 bits each, so k * part is exact for |k| < 2^20) and the classic degree-13 / degree-14 minimax polynomials on
 * [-pi/4, pi/4] (the coefficients of fdlibm's __kernel_sin / __kernel_cos, evaluated as two interleaved chains): about 1 ulp,
 * a third of the instructions of the library routine and no branch.  Beyond (a joint that has spun 80 000 turns) the
 * library's sincos, taken by the whole wave. */
WV_DEVICE void sincos_reduced(double x, double &sn, double &cs) { /* |x| < 2^19 */
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double s = fma(r * z, ps, r);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + fma(z * z, pc, (1.0 - w) - hz);
    const int n = (int)k;
    const double a = (n & 1) ? c : s, bq = (n & 1) ? s : c;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -bq : bq;
}
WV_DEVICE void sincos_bounded(double x, double &sn, double &cs) {
    if (wv::ballot(!(fabs(x) < 524288.0)) != 0ull) sincos(x, &sn, &cs);
    else sincos_reduced(x, sn, cs);
}
WV_DEVICE void mulquat(double *r, const double *a, const double *b) {
    double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
WV_DEVICE void quat2mat(double *m, const double *q) {
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03);       m[2] = 2 * (q13 + q02);
    m[3] = 2 * (q12 + q03);       m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
    m[6] = 2 * (q13 - q02);       m[7] = 2 * (q23 + q01);       m[8] = q00 - q11 - q22 + q33;
}
WV_DEVICE void mulmatvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double t1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double t2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE void mulmatTvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    double t1 = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    double t2 = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE void rotvecquat(double *r, const double *v, const double *q) {
    double m[9];
    quat2mat(m, q);
    mulmatvec3(r, m, v);
}
WV_DEVICE double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* spatial algebra, [rotational; translational] */
WV_DEVICE void cross_motion(double *r, const double *vel, const double *v) {
    double a[3], b[3], c[3];
    cross3(a, vel, v); cross3(b, vel, v + 3); cross3(c, vel + 3, v);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
WV_DEVICE void cross_force(double *r, const double *vel, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
WV_DEVICE void mul_inert_vec(double *r, const double *I, const double *v) {
    r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
    r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
    r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
    r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
    r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
    r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}

}  // namespace ck
#endif
