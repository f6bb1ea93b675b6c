/*
 * pk_wide_solve.h -- the solve of the 127-row instantiation, spread over the two wavefronts of an env
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_WIDE_SOLVE_H
#define CASSIE_PK_WIDE_SOLVE_H

#ifndef CK_WIDE_SWEEP_V2     /* (A/B switch: 0 = round 5's sweep) */
#define CK_WIDE_SWEEP_V2 1
#endif

namespace ck {

/* ---------------- the solve of the 127-row instantiation (MAXR = WIDE_ROWS, two wavefronts): both waves call it behind the barrier
 * J, wave w for the rows 64 w .. 64 w + 63 (row 127 = the qfrc_smooth column, wave 1's lane 63).  The row stages (wave 0, one or two
 * passes) have parked every row's raw Jacobian row in x.Yr and its parameters in x.rowt; here every lane takes its row from there,
 * runs the half solve, puts the staged row back (barrier Y: the staged matrix is complete), forms ITS WAVE'S 64 x 64 block of
 * A = Y Y^T in registers, and solves:
 *   - a substep of at most 64 rows lives on wave 0 alone: the very chain of operations of the 63-row instantiation (so a substep that
 *     fits both gives the same bits in both), wave 1 returns at once;
 *   - with more rows a Gauss-Seidel sweep is wave 0's rows, then wave 1's, in row order -- the oracle's order.  The waves never run
 *     at the same time, so the chain crosses them twice per sweep, not once per row: a wave that has finished its half hands over
 *     v = sum over its rows of (staged row of Y) x (the row's step), the steps' image in joint space (NVP numbers through LDS), and
 *     the other wave's residuals take it in through their own staged rows, res_j += Y_j . v -- which is A_jI step_I summed over the
 *     other wave's rows I without the off-diagonal blocks of A existing anywhere.  The guard (never accept a cost increase) is
 *     local to a half; the sweep's cost change is summed in row order across both halves, as the oracle sums it.
 * Returns this lane's row force; iters / nguarded: the sweeps taken (valid in both waves when the substep has more than 64 rows,
 * in wave 0 otherwise). ---------------- */
template <int NVP, class TOPO, class SH>
WV_DEVICE double wide_solve(SH &S, ModelPtr m, ParamPtr P, const int wid, const int nefc, const int sub, int &iters, int &nguarded) {
    constexpr int H = NROW, MAXR = WIDE_ROWS;
    static_assert(TOPO::is_static, "the 127-row instantiation exists for the compile-time topologies");
    const int lane = wv::fresh_lane();
    const int g = H * wid + lane;                    /* this lane's row (127: the qfrc_smooth column) */
    const bool hasrow = g < nefc, qcol = g == MAXR;
    const int nown = wid == 0 ? (nefc < H ? nefc : H) : (nefc > H ? nefc - H : 0);
    iters = 0; nguarded = 0;
    double rR = 1.0, raref = 0.0, jws = 0.0, code = -1.0;
    {
        const double *rt = S.x.rowt[hasrow ? g : 0];
        const double t0 = rt[0], t1 = rt[1], t2 = rt[2], t3 = rt[3];
        if (hasrow) { rR = t0; raref = t1; jws = t2; code = t3; }
    }
    const bool isrow = hasrow && code >= 0.0, clampf = isrow && code > 0.5;
    double ycol[NVP];
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
        const double raw = S.x.Yr[hasrow ? g : 0][k], qs = S.qfrc_smooth[k < TOPO::nv ? k : 0];
        ycol[k] = hasrow ? raw : ((qcol && k < TOPO::nv) ? qs : 0.0);
    }
    /* ---- half solve in registers: Y = D^-1/2 L^-T [J^T | qfrc_smooth] (env_step's, for a compile-time topology) ---- */
    {
        double ta[NVP], tb[NVP], ra = 0, rb = 0;
        auto fetch = [&](int k, double (&t)[NVP], double &rs) {
#pragma unroll
            for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) t[i] = S.Lp[LPack<TOPO, NVP>::idx(k, i)];
            rs = S.rsd[k];
        };
        fetch(TOPO::nv - 1, ((TOPO::nv - 1) & 1) ? ta : tb, ((TOPO::nv - 1) & 1) ? ra : rb);
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (k >= TOPO::nv) continue;
            if (k > 0) fetch(k - 1, ((k - 1) & 1) ? ta : tb, ((k - 1) & 1) ? ra : rb);
            wv::sched_fence();
            const double (&t)[NVP] = (k & 1) ? ta : tb;
            const double xk = ycol[k];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) ycol[i] -= t[i] * xk;
            ycol[k] = xk * ((k & 1) ? ra : rb);
            wv::sched_fence();
        }
    }
    {   /* (every lane: the lanes that hold no row store zeros, so that all 128 rows of the tile are defined) */
#pragma unroll
        for (int k = 0; k < NVP; ++k) S.x.Yr[g][k] = ycol[k];
    }
    wv::block_barrier(); /* Y: the staged matrix is complete -- both waves' rows and the qfrc_smooth column */
    if (nefc <= H && wid == 1) return 0.0; /* (no rows on this wave) */

    /* ---- this lane's row of its wave's block of A = Y Y^T (one FMA chain per product over the dofs in index order, as everywhere),
     *      b = Y y_q - aref, the diagonal ---- */
    double arow[H];
    const int rbase = H * wid;
#pragma unroll
    for (int r = 0; r < H; r += 2) {
        double acc0 = 0, acc1 = 0;
        if (rbase + r < nefc) {
            double ya[NVP], yb[NVP];
#pragma unroll
            for (int k = 0; k < NVP; ++k) ya[k] = S.x.Yr[rbase + r][k];
#pragma unroll
            for (int k = 0; k < NVP; ++k) yb[k] = S.x.Yr[rbase + r + 1][k];
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < NVP; ++k) { acc0 = fma(ya[k], ycol[k], acc0); acc1 = fma(yb[k], ycol[k], acc1); }
        }
        arow[r] = acc0;
        arow[r + 1] = rbase + r + 1 < nefc ? acc1 : 0.0; /* (row 127 is the qfrc_smooth column, not a row) */
    }
    double rb;
    {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) acc = fma(S.x.Yr[MAXR][k], ycol[k], acc);
        rb = acc - raref;
    }
    double Aii = 1.0;
    if (isrow) {
        double d = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) d = fma(ycol[k], ycol[k], d);
        Aii = d + rR;
    }
    const double invAii = 1.0 / Aii;
    double f = 0, res = isrow ? rb : 0.0;
    const int r_ = lane; /* the row's index within its wave */
    const int nvs = TOPO::nv;
    const double scale = 1.0 / (P->meaninertia * (nvs > 1 ? nvs : 1));
    const double halfAii = 0.5 * Aii;
    const double flo = clampf ? 0.0 : -1e300;
    const double ninvAii = -invAii;
    const int maxiter = m->iterations;
    const double tolerance = m->tolerance;
    const int kk = lane < NVP ? lane : 0;

    if (nefc <= H) {
        /* ======== at most 64 rows: wave 0 alone, the 63-row instantiation's chain of operations ======== */
        if (m->flags & CM_FLAG_WARMSTART) {
            if (isrow) {
                f = -(jws - raref) / rR;
                if (clampf && f < 0) f = 0;
            }
            double af0 = 0, af1 = 0, af2 = 0, af3 = 0;
#pragma unroll
            for (int t = 0; t < H; t += 4) {
                if (t < nefc) {
                    af0 += arow[t] * wv::readlane(f, t);
                    af1 += arow[t + 1] * wv::readlane(f, t + 1);
                    af2 += arow[t + 2] * wv::readlane(f, t + 2);
                    af3 += arow[t + 3] * wv::readlane(f, t + 3);
                }
            }
            const double af = ((af0 + af1) + (af2 + af3)) + (isrow ? rR * f : 0.0);
            double cost = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
            if (cost > 0) f = 0;
            else if (isrow) res = rb + af;
        }
        double sres = res * ninvAii;
#pragma unroll
        for (int t = 0; t < H; ++t) arow[t] *= ninvAii;
        const double cdiag = isrow ? rR * ninvAii : 0.0;
        const bool shortcut_ok = 0.5 * tolerance > (double)MID_ROWS * 1e-10 * scale;
        while (iters < maxiter) {
            const int nrows = wv::opaque(nefc);
            bool converged;
            {
                const double f0 = f, s0 = sres;
                double mys = 0;
                const double lo_f = flo - f;
                pgs_rows_fast<0, H>(arow, nrows, r_, lo_f, sres, mys);
                const double mydelta = wv::max_raw(mys, lo_f);
                const double change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
                const float tol = (float)tolerance;
                const bool one_row_decides = shortcut_ok && wv::ballot(-(float)change * (float)scale > 2.5f * tol) != 0ull;
                const float est = one_row_decides ? 4.0f * tol : -wv::wave_sum_f32((float)change) * (float)scale;
                if (wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded()) {
                    double improvement = 0;
                    f = f0; sres = s0; ++nguarded;
                    pgs_rows<0, H>(arow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
                    sres = fma(cdiag, f - f0, sres);
                    converged = improvement * scale < tolerance;
                } else {
                    if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                    if (est < 0.5f * tol) converged = true;
                    else if (est > 2.0f * tol) converged = false;
                    else {
                        const double tree = -wv::wave_sum(change) * scale, tolv = tolerance;
                        if (fabs(tree - tolv) > 1e-9 * tolv) converged = tree < tolv;
                        else {
                            double improvement = 0;
                            for (int t = 0; t < nrows; ++t) improvement -= wv::readlane(change, t);
                            converged = improvement * scale < tolerance;
                        }
                    }
                }
            }
            ++iters;
            if (converged) break;
        }
        return f;
    }

    /* ======== more than 64 rows: the sweep crosses the waves ======== */
    /* v[wid] = sum over this wave's rows t < nown of (staged row) x val_t, lane = dof.  Round 6: the steps go through LDS and the two
     * halves of the wave take the even and the odd rows (lanes 32 .. 63 used to idle here, and every row cost two lane reads): two
     * partial sums per half, the halves' sums added at the end */
    auto image_of = [&](double val) {
        if constexpr (CK_WIDE_SWEEP_V2) {
            S.x.stepv[wid][lane] = val;     /* (val is 0 in lanes that are not rows; their rows of the tile are zeros) */
            wv::sync();
            const int half = lane >> 5, k = lane & 31;
            double v0 = 0, v1 = 0;
            /* (sixteen rows to a wave-uniform branch: their eight pairs of LDS reads are requested together) */
#pragma unroll
            for (int t0 = 0; t0 < H; t0 += 16) {
                if (t0 < nown) {
                    double y[8], sv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { y[i] = S.x.Yr[rbase + t0 + 2 * i + half][k]; sv[i] = S.x.stepv[wid][t0 + 2 * i + half]; }
                    wv::sched_fence();
#pragma unroll
                    for (int i = 0; i < 8; i += 2) { v0 = fma(y[i], sv[i], v0); v1 = fma(y[i + 1], sv[i + 1], v1); }
                }
            }
            const double mine = v0 + v1, other = wv::from_upper_half(mine);
            if (lane < NVP) S.x.vx[wid][lane] = mine + other;
            return;
        }
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
        for (int t = 0; t < H; t += 4) {
            if (t < nown) {
                v0 = fma(S.x.Yr[rbase + t][kk], wv::readlane(val, t), v0);
                v1 = fma(S.x.Yr[rbase + t + 1][kk], wv::readlane(val, t + 1), v1);
                v2 = fma(S.x.Yr[rbase + t + 2][kk], wv::readlane(val, t + 2), v2);
                v3 = fma(S.x.Yr[rbase + t + 3][kk], wv::readlane(val, t + 3), v3); /* (val is 0 in lanes that are not rows; their rows of the tile are zeros) */
            }
        }
        if (lane < NVP) S.x.vx[wid][lane] = (v0 + v1) + (v2 + v3);
    };
    /* this lane's staged row times the other wave's image: what the other wave's values contribute to this row's A x */
    auto cross = [&]() {
        if constexpr (CK_WIDE_SWEEP_V2) {   /* (four chains of eight instead of one of 32) */
            double vo[NVP], d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            const double *const vother = &S.x.vx[1 - wid][0];
#pragma unroll
            for (int k = 0; k < NVP; ++k) vo[k] = vother[k];       /* (all the broadcast reads requested together) */
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < NVP; k += 4) {
                d0 = fma(ycol[k], vo[k], d0); d1 = fma(ycol[k + 1], vo[k + 1], d1);
                d2 = fma(ycol[k + 2], vo[k + 2], d2); d3 = fma(ycol[k + 3], vo[k + 3], d3);
            }
            return (d0 + d1) + (d2 + d3);
        }
        double d = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) d = fma(ycol[k], S.x.vx[1 - wid][k], d);
        return d;
    };
    if (m->flags & CM_FLAG_WARMSTART) {
        if (isrow) {
            f = -(jws - raref) / rR;
            if (clampf && f < 0) f = 0;
        }
        image_of(f);
        wv::block_barrier();
        double af0 = 0, af1 = 0, af2 = 0, af3 = 0;
#pragma unroll
        for (int t = 0; t < H; t += 4) {
            if (t < nown) {
                af0 += arow[t] * wv::readlane(f, t);
                af1 += arow[t + 1] * wv::readlane(f, t + 1);
                af2 += arow[t + 2] * wv::readlane(f, t + 2);
                af3 += arow[t + 3] * wv::readlane(f, t + 3);
            }
        }
        const double af = (((af0 + af1) + (af2 + af3)) + cross()) + (isrow ? rR * f : 0.0);
        const double part = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
        if (lane == 0) S.x.sums[wid] = part;
        wv::block_barrier();
        const double cost = S.x.sums[0] + S.x.sums[1];
        if (cost > 0) f = 0;
        else if (isrow) res = rb + af;
        wv::block_barrier(); /* (vx and sums are free again) */
    }
    double sres = res * ninvAii;
#pragma unroll
    for (int t = 0; t < H; ++t) arow[t] *= ninvAii;
    const double cdiag = isrow ? rR * ninvAii : 0.0;
    /* the turn word: base + 2 s + 1 = wave 0 has finished its half of sweep s, base + 2 s + 2 = wave 1 has (and has left its verdict) */
    const int base = (sub + 1) << 12;
    const int sweeps_max = maxiter < 2000 ? maxiter : 2000;
    if constexpr (CK_WIDE_SWEEP_V2) {
        /* Round 6.  A sweep's cost change only feeds the convergence test, and summing it in row order was a chain of up to 64 lane
         * reads + additions per half and sweep.  Now, as in the 63-row solve: each half forms a single-precision tree sum; wave 1 adds
         * the two and decides unless the sum lands within a factor two of the tolerance -- only then (or when a half's guard fired,
         * whose guarded re-run sums in order anyway) is the ordered double-precision sum formed, wave 0's part from the per-row
         * changes it left in LDS.  What wave 0 hands over with its turn: sums[2] = its part (exact when sums[4] != 0, else the
         * estimate), chg[] = its rows' changes. */
        const float tolf = (float)tolerance, scalef = (float)scale;
#ifdef CK_WIDE_PROFILE
        long long pt_wait = 0, pt_cross = 0, pt_rows = 0, pt_post = 0, pt_image = 0, pt_pub = 0, pt_t;
#define CK_WP(acc) do { const long long now_ = wv::clock(); acc += now_ - pt_t; pt_t = now_; } while (0)
#else
#define CK_WP(acc) do {} while (0)
#endif
        for (int sweep = 0;; ++sweep) {
            double w0_part = 0.0;
            bool w0_exact = false;
#ifdef CK_WIDE_PROFILE
            pt_t = wv::clock();
#endif
            if (wid == 0) {
                if (sweep > 0) {
                    wv::wait_for_spin(&S.x.turn[1], base + 2 * sweep);
                    CK_WP(pt_wait);
                    if (wv::opaque(S.x.turn[2])) break;
                    sres = fma(ninvAii, cross(), sres);
                    CK_WP(pt_cross);
                }
            } else {
                wv::wait_for_spin(&S.x.turn[1], base + 2 * sweep + 1);
                CK_WP(pt_wait);
                w0_part = S.x.sums[2]; w0_exact = S.x.sums[4] != 0.0;     /* (wave 0's part of the cost change, for the verdict below: requested here) */
                sres = fma(ninvAii, cross(), sres);
                CK_WP(pt_cross);
            }
            const int nrows = wv::opaque(nown);
            /* wave 0's part of the sweep's cost change in row order, from what it left in LDS (wave 1 only, when it is needed) */
            auto ordered_part_of_wave0 = [&]() {
                if (w0_exact) return w0_part;
                const double c0 = S.x.chg[lane];
                double imp = 0.0;
                for (int t = 0; t < H; ++t) imp -= wv::readlane(c0, t);   /* (a sweep that crosses the waves has all 64 rows of wave 0) */
                return imp;
            };
            double dstep, exact = 0.0, change = 0.0;
            float est = 0.0f;
            bool guarded;
            {
                const double f0 = f, s0 = sres;
                double mys = 0;
                const double lo_f = flo - f;
                pgs_rows_fast<0, H>(arow, nrows, r_, lo_f, sres, mys);
                CK_WP(pt_rows);
                const double mydelta = wv::max_raw(mys, lo_f);
                change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
                guarded = wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded();
                if (guarded) {  /* some row of this half would have raised the cost: redo it guarded, the cost change summed in row order */
                    f = f0; sres = s0;
                    ++nguarded;
                    exact = wid == 0 ? 0.0 : ordered_part_of_wave0();
                    pgs_rows<0, H>(arow, nrows, r_, Aii, halfAii, flo, f, sres, exact);
                    sres = fma(cdiag, f - f0, sres);
                    dstep = f - f0;
                } else {
                    dstep = (r_ < nrows) ? mydelta : 0.0;
                    if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                    est = -wv::wave_sum_f32((float)change);
                }
            }
            CK_WP(pt_post);
            image_of(dstep);
            CK_WP(pt_image);
            if (wid == 0) {
                if (!guarded) S.x.chg[lane] = change;
                if (lane == 0) { S.x.sums[2] = guarded ? exact : (double)est; S.x.sums[4] = guarded ? 1.0 : 0.0; }
                wv::publish(&S.x.turn[1], base + 2 * sweep + 1);
                CK_WP(pt_pub);
            } else {
                iters = sweep + 1;
                bool converged;
                if (guarded) converged = exact * scale < tolerance;
                else {
                    /* (wave 0's exact part is a double; in single precision it is as good an estimate as a tree sum) */
                    const float total = ((float)w0_part + est) * scalef;
                    if (total < 0.5f * tolf) converged = true;
                    else if (total > 2.0f * tolf) converged = false;
                    else {
                        double imp = ordered_part_of_wave0();
                        for (int t = 0; t < nrows; ++t) imp -= wv::readlane(change, t);
                        converged = imp * scale < tolerance;
                    }
                }
                const bool stop = converged || iters >= sweeps_max;
                if (lane == 0) { S.x.turn[2] = stop ? 1 : 0; S.x.turn[3] = iters; S.x.sums[3] = (double)nguarded; }
                wv::publish(&S.x.turn[1], base + 2 * sweep + 2);
                CK_WP(pt_pub);
                if (stop) break;
            }
        }
#ifdef CK_WIDE_PROFILE
        /* (profiling aid: this wave's clocks by part of the sweep, summed over the sweeps, for the caller to put into the stamp array) */
        if (lane == 0) { double *o = &S.x.stepv[wid][0]; o[0] = (double)pt_wait; o[1] = (double)pt_cross; o[2] = (double)pt_rows; o[3] = (double)pt_post; o[4] = (double)pt_image; o[5] = (double)pt_pub; }
#endif
    } else
    for (int sweep = 0;; ++sweep) {
        double carried = 0.0; /* the sweep's cost change summed in row order up to this wave's first row */
        if (wid == 0) {
            if (sweep > 0) {
                wv::wait_for(&S.x.turn[1], base + 2 * sweep);
                if (wv::opaque(S.x.turn[2])) break; /* (wave 1's verdict on the sweep before: converged, or out of sweeps) */
                sres = fma(ninvAii, cross(), sres);
            }
        } else {
            wv::wait_for(&S.x.turn[1], base + 2 * sweep + 1);
            sres = fma(ninvAii, cross(), sres);
            carried = S.x.sums[2];
        }
        const int nrows = wv::opaque(nown);
        double improvement = carried, dstep;
        {
            const double f0 = f, s0 = sres;
            double mys = 0;
            const double lo_f = flo - f;
            pgs_rows_fast<0, H>(arow, nrows, r_, lo_f, sres, mys);
            const double mydelta = wv::max_raw(mys, lo_f);
            const double change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
            if (wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded()) { /* some row of this half would have raised the cost: redo it guarded */
                f = f0; sres = s0;
                ++nguarded;
                pgs_rows<0, H>(arow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
                sres = fma(cdiag, f - f0, sres);
                dstep = f - f0;
            } else {
                dstep = (r_ < nrows) ? mydelta : 0.0;
                if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                for (int t = 0; t < nrows; ++t) improvement -= wv::readlane(change, t);
            }
        }
        image_of(dstep);
        if (wid == 0) {
            if (lane == 0) S.x.sums[2] = improvement;
            wv::publish(&S.x.turn[1], base + 2 * sweep + 1);
        } else {
            iters = sweep + 1;
            const bool stop = improvement * scale < tolerance || iters >= sweeps_max;
            if (lane == 0) { S.x.turn[2] = stop ? 1 : 0; S.x.turn[3] = iters; S.x.sums[3] = (double)nguarded; }
            wv::publish(&S.x.turn[1], base + 2 * sweep + 2);
            if (stop) break;
        }
    }
    if (wid == 0) { iters = wv::opaque(S.x.turn[3]); nguarded += (int)S.x.sums[3]; }
    return f;
}

}  // namespace ck
#endif
