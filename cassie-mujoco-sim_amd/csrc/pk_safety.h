/*
 * pk_safety.h -- cassie_core_sim's safety layer, restated (reference include/cassie_core_sim.h:30-35,
 * called at reference src/cassiemujoco.c:1141 between pd_input_step and cassie_sim_step_ethercat).
 *
 * The block exists only as the closed binary src/libagilitycassie.a(cassie_core_sim.o).  What follows is its LAW: every
 * constant and every piece of it can be recovered from the block's behaviour alone (tools/core_sim_probe.py does so with
 * black-box probes of the live binary: bounds by bisection, gains by fits, the blend width, the coupling, the clamp, the queue;
 * profiles/round6/core_sim_probe.txt), the order of the floating-point operations was read off its disassembly; nothing of the
 * binary is carried here.  Every operation is the individually rounded IEEE operation the binary performs, in its order -- the
 * outputs are BIT FOR BIT the binary's:
 * tests/test_core_safety.py (goldens generated from the real .a by tools/make_golden_core_safety.py through oracle/_ref;
 * 10^7 random + adversarial samples against the live binary where it is present).
 *
 *   inputs    user torques u[10] (cassie_user_in_t.torque: pd_input's output), the MEASURED drive positions q[10] and
 *             velocities w[10] and the drives' torque limits L[10] of cassie_out_t (left leg hip roll, hip yaw, hip pitch,
 *             knee, foot; right leg likewise), radio channel 8 (the STO switch)
 *   limits    22 linear constraints  c_i = A_i . q - b_i <= 0 :  a lower and an upper bound per drive (rows 0-9: -q_k -
 *             LOWER_k, rows 10-19: q_k - UPPER_k) and hip pitch + knee >= -3 pi / 4 per leg (rows 20, 21)
 *   attenuation   every user torque is multiplied by  prod_i (1 - c_i / 0.15)  over the violated constraints (0 once a
 *             violation reaches 0.15 rad): a robot deep in a limit stops obeying its controller altogether
 *   restoring     per violated constraint i and drive k with A_ik != 0:  - A_ik (kp_k p + kq_k p^2) - |A_ik| kd_k w_k s,
 *             p = max(c_i, 0), s = min(c_i / 0.15, 1): a stiffening spring towards the admissible side and a damper
 *             blended in over the first 0.15 rad
 *   clamp     to +- L_k;  STO (radio channel 8 != 1): all torques x 0
 *   messages  diagnostic code 635 once any constraint is violated, 630 once any |torque| reaches its limit, kept in a
 *             4-deep queue sorted by code (cassie_in_t.pelvis.radio.channel[1..4]); sticky until cassie_core_sim_setup
 *
 * Used by the step kernel's drive-level pass in CM_DRIVE_PD_SAFE mode (pk_stages.h: drive_level_io), lane = drive.
 */
#ifndef CASSIE_PK_SAFETY_H
#define CASSIE_PK_SAFETY_H

namespace ck {
namespace safety {

constexpr int NROW = 22, NDRV = 10;
/* b_i of the 22 constraints: -q_k <= LOWER_k (rows 0-9), q_k <= UPPER_k (rows 10-19), -(hip pitch + knee) <= 3 pi / 4 (20, 21) */
WV_DEVICE double bound(int i) {
    /* (in degrees: 6.4056, 13.4056, 41.4056, 147.4056, 131.4056 | 11.4056, 13.4056, 71.4056, -50.5944, -43.5944 | 135: the
     * mechanical ranges of model/cassie.xml:98-149 drawn in by 8.59 degrees).  Selects, not a table: the loops over the
     * constraints are unrolled and the values become literals (a constant table indexed at run time also crashed hipcc 7.2's
     * register allocator in the stand-alone drive kernel) */
    const int leg_lo = i < 10 ? i % 5 : -1, leg_hi = (i >= 10 && i < 20) ? (i - 10) % 5 : -1;
    const bool right = (i >= 5 && i < 10) || (i >= 15 && i < 20);
    if (leg_lo == 0) return right ? 0.1990658503988659 : 0.11179938779914941;
    if (leg_lo == 1) return 0.23397243543875249;
    if (leg_lo == 2) return 0.7226646259971647;
    if (leg_lo == 3) return 2.572713633111154;
    if (leg_lo == 4) return 2.2934609527920613;
    if (leg_hi == 0) return right ? 0.11179938779914941 : 0.1990658503988659;
    if (leg_hi == 1) return 0.23397243543875249;
    if (leg_hi == 2) return 1.2462634015954637;
    if (leg_hi == 3) return -0.8830382858376185;
    if (leg_hi == 4) return -0.7608652381980153;
    return 2.356194490192345;
}
/* spring (linear, quadratic) and damper gains per drive of a leg: hip roll, hip yaw, hip pitch, knee, foot */
WV_DEVICE double gain_p(int k) { const int j = k % 5; return j == 0 ? 1000.0 : j == 1 ? 800.0 : j < 4 ? 1200.0 : 100.0; }
WV_DEVICE double gain_q(int k) { const int j = k % 5; return j == 0 ? 6666.666666666667 : j == 1 ? 5333.333333333334 : j < 4 ? 8000.0 : 666.6666666666667; }
WV_DEVICE double gain_d(int k) { const int j = k % 5; return j < 2 ? 12.0 : j < 4 ? 36.0 : 7.0; }
constexpr double BLEND = 0.15;          /* rad over which attenuation and damping reach their full effect */
/* the drives' torque limits as cassie_out_init writes them into cassie_out_t (reference src/cassiemujoco.c:711-727): the
 * `torqueLimit` the block clamps to is a FIELD of cassie_out_t, which the simulator never changes afterwards */
WV_DEVICE double torque_limit(int k) { const int j = k % 5; return j < 2 ? 140.63 : j < 4 ? 216.16 : 45.14; }
enum { MSG_LIMIT = 1 /* code 635: a joint-limit constraint is violated */, MSG_TORQUE = 2 /* code 630: a torque reached its limit */ };
constexpr int CODE_LIMIT = 635, CODE_TORQUE = 630;

/* coefficient A_ik in {-1, 0, 1} */
WV_DEVICE int coeff(int i, int k) {
    if (i < 10) return k == i ? -1 : 0;
    if (i < 20) return k == i - 10 ? 1 : 0;
    const int leg = i - 20;
    return (k == 5 * leg + 2 || k == 5 * leg + 3) ? -1 : 0;
}
/* c_i = (sum_k A_ik q_k, accumulated from 0 in drive order) - b_i.  (The binary multiplies out the zero coefficients too; for
 * finite q those terms are signed zeros that leave the sum's bits alone.) */
WV_DEVICE double violation(int i, const double *q) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NDRV; ++k) {
        const int a = coeff(i, k);
        if (a != 0) acc = wv::add_rn(acc, a > 0 ? q[k] : -q[k]);
    }
    return wv::sub_rn(acc, bound(i));
}

/* The torque the block sends to drive k, and the message bits the step raises (*msg |= ...; the torque-limit bit concerns
 * drive k only: OR it over the drives).  uk / wk: the drive's user torque and measured velocity; q: the ten measured positions (a
 * drive's torque depends on the others' only through the constraints); L_k: the drive's limit; sto: radio channel 8 != 1. */
WV_DEVICE double drive_torque(int k, double uk, const double *q, double wk, double Lk, bool sto, int *msg) {
    double c[NROW];
    bool any = false;
#pragma unroll
    for (int i = 0; i < NROW; ++i) { c[i] = violation(i, q); any |= c[i] > 0.0; }
    if (any) *msg |= MSG_LIMIT;
    const double kp = gain_p(k), kq = gain_q(k), D = wv::mul_rn(gain_d(k), wk);
    double t;
    if (!any && fabs(D) <= 1.7976931348623157e308) {
        /* No constraint violated (the same verdict in every drive's lane: it depends on q alone) and a finite damper term: the
         * attenuation loop multiplies nothing, and the 44 terms of the restoring loop are signed zeros -- among them (+0) x (-1) =
         * -0 from the drive's own lower-bound row -- whose only trace in the binary's result is that a NEGATIVE zero torque leaves
         * as a positive one.  u + 0 does exactly that, bit for bit, without the loops' 22 divisions. */
        t = wv::add_rn(uk, 0.0);
    } else {
        /* attenuation of the user torque (x_i = c_i / 0.15 is used by both loops: divided once) */
        double x[NROW];
        t = uk;
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            x[i] = wv::div_rn(c[i], BLEND);
            if (x[i] > 0.0) t = wv::mul_rn(t, 1.0 > x[i] ? wv::sub_rn(1.0, x[i]) : 0.0);
        }
        /* restoring spring and damper of every constraint, violated or not (the others contribute signed zeros, which matter
         * for the sign of a zero torque only -- kept, so that the bits are the binary's) */
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const double ci = c[i];
            const double p = ci >= 0.0 ? ci : 0.0;                           /* fmax(c, 0) */
            const double s = x[i] > 0.0 ? (x[i] < 1.0 ? x[i] : 1.0) : 0.0;
            const double g = wv::add_rn(wv::mul_rn(p, kp), wv::mul_rn(wv::mul_rn(p, p), kq));
            const double a = (double)coeff(i, k);
            t = wv::sub_rn(t, wv::mul_rn(g, a));
            t = wv::sub_rn(t, wv::mul_rn(wv::mul_rn(fabs(a), D), s));
        }
    }
    /* message 630: |torque| >= limit (ordered compare), before the clamp */
    if (fabs(t) >= Lk) *msg |= MSG_TORQUE;
    /* clamp to [-L, L] with the binary's branch order (it decides what a NaN or a non-positive limit gives) */
    const double neg = -Lk;
    double r;
    if (t > neg) r = Lk > t ? t : (Lk > neg ? Lk : neg);
    else r = Lk > neg ? neg : (Lk > t ? Lk : t);
    /* STO: the scale factor the block multiplies with is 0 instead of 1 (the product keeps the torque's sign on its zero) */
    return wv::mul_rn(sto ? 0.0 : 1.0, r);
}

}  // namespace safety
}  // namespace ck
#endif
