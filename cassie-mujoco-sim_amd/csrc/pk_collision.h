/*
 * pk_collision.h -- narrow phase: plane / sphere / capsule / box primitives, box-box by separating axes, height field (closest feature, and CM_FLAG_HFPRISM's one contact per penetrated grid triangle), the contact list
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_COLLISION_H
#define CASSIE_PK_COLLISION_H

namespace ck {

/* ------------------------------------------------------ narrow phase ------ */
struct RawContact { double dist, pos[3], normal[3], tangent[3]; };

WV_DEVICE int plane_sphere(RawContact &c, const double *ppos, const double *pmat, const double *spos, double r,
                           double margin) {
    double n[3] = {pmat[2], pmat[5], pmat[8]};
    double dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
    double dist = dot3(dif, n) - r;
    if (dist > margin) return 0;
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = n[i]; c.pos[i] = spos[i] - n[i] * (r + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE int sphere_sphere(RawContact &c, const double *p1, double r1, const double *p2, double r2, double margin) {
    double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double cd = sqrt(dot3(dif, dif));
    double dist = cd - r1 - r2;
    if (dist > margin) return 0;
    double n[3];
    if (cd < CM_MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; }
    else { n[0] = dif[0] / cd; n[1] = dif[1] / cd; n[2] = dif[2] / cd; }
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = n[i]; c.pos[i] = p1[i] + n[i] * (r1 + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE void segment_closest(const double *p1, const double *a1, double l1, const double *p2, const double *a2,
                               double l2, double &x1, double &x2) {
    double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double mb = -dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
    double det = 1.0 - mb * mb;
    double t1, t2;
    if (fabs(det) >= 1e-12) {
        t1 = (u - mb * v) / det;
        t2 = (v - mb * u) / det;
        if (t1 > l1) { t1 = l1; t2 = v - mb * t1; }
        else if (t1 < -l1) { t1 = -l1; t2 = v - mb * t1; }
        if (t2 > l2) { t2 = l2; t1 = clampd(u - mb * t2, -l1, l1); }
        else if (t2 < -l2) { t2 = -l2; t1 = clampd(u - mb * t2, -l1, l1); }
    } else {
        double s = -mb, c2 = v;
        double lo = fmax(-l2, c2 - l1), hi = fmin(l2, c2 + l1);
        if (lo <= hi) t2 = 0.5 * (lo + hi);
        else t2 = clampd(c2, -l2, l2);
        t1 = clampd((t2 - c2) * (s >= 0 ? 1.0 : -1.0), -l1, l1);
    }
    x1 = t1; x2 = t2;
}
WV_DEVICE void make_frame(double *frame) {
    normalize3(frame);
    if (sqrt(dot3(frame + 3, frame + 3)) < 0.5) {
        frame[3] = frame[4] = frame[5] = 0;
        if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1; else frame[5] = 1;
    }
    double t = dot3(frame, frame + 3);
    for (int i = 0; i < 3; ++i) frame[3 + i] -= t * frame[i];
    normalize3(frame + 3);
    cross3(frame + 6, frame, frame + 3);
}


/* ---- boxes (same definitions as oracle/cassie_oracle.c) ---- */
WV_DEVICE double point_box(const double *q, const double *pb, const double *mb, const double *sb, double *nworld) {
    double d[3] = {q[0] - pb[0], q[1] - pb[1], q[2] - pb[2]}, loc[3], cl[3];
    mulmatTvec3(loc, mb, d);
    bool inside = true;
    for (int k = 0; k < 3; ++k) { cl[k] = clampd(loc[k], -sb[k], sb[k]); if (cl[k] != loc[k]) inside = false; }
    double nl[3] = {0, 0, 0}, dist;
    if (!inside) {
        double dif[3] = {loc[0] - cl[0], loc[1] - cl[1], loc[2] - cl[2]};
        dist = sqrt(dot3(dif, dif));
        for (int k = 0; k < 3; ++k) nl[k] = dif[k] / dist;
    } else {
        const double d0 = sb[0] - fabs(loc[0]), d1 = sb[1] - fabs(loc[1]), d2 = sb[2] - fabs(loc[2]);
        double best = d0;
        int kb = 0;
        if (d1 < best) { best = d1; kb = 1; }
        if (d2 < best) { best = d2; kb = 2; }
        const double sg = (kb == 0 ? loc[0] : (kb == 1 ? loc[1] : loc[2])) >= 0 ? 1.0 : -1.0;
        nl[0] = kb == 0 ? sg : 0.0; nl[1] = kb == 1 ? sg : 0.0; nl[2] = kb == 2 ? sg : 0.0;
        dist = -best;
    }
    mulmatvec3(nworld, mb, nl);
    return dist;
}
WV_DEVICE int sphere_box(RawContact &c, const double *ps, double r, const double *pb, const double *mb, const double *sb, double margin) {
    double nw[3];
    const double dist = point_box(ps, pb, mb, sb, nw) - r;
    if (dist > margin) return 0;
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = -nw[i]; c.pos[i] = ps[i] - nw[i] * (r + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE int capsule_box(RawContact &c0, RawContact &c1, const double *pc, const double *mc, double rad, double h, const double *pb,
                          const double *mb, const double *sb, double margin) {
    const double ax[3] = {mc[2], mc[5], mc[8]}, gr = 0.6180339887498949;
    double lo = -h, hi = h, nw[3];
    double t1 = hi - gr * (hi - lo), t2 = lo + gr * (hi - lo);
    double q1[3] = {pc[0] + ax[0] * t1, pc[1] + ax[1] * t1, pc[2] + ax[2] * t1}, q2[3] = {pc[0] + ax[0] * t2, pc[1] + ax[1] * t2, pc[2] + ax[2] * t2};
    double f1 = point_box(q1, pb, mb, sb, nw), f2 = point_box(q2, pb, mb, sb, nw);
    for (int it = 0; it < 32; ++it) {
        if (f1 <= f2) { hi = t2; t2 = t1; f2 = f1; t1 = hi - gr * (hi - lo); for (int i = 0; i < 3; ++i) q1[i] = pc[i] + ax[i] * t1; f1 = point_box(q1, pb, mb, sb, nw); }
        else { lo = t1; t1 = t2; f1 = f2; t2 = lo + gr * (hi - lo); for (int i = 0; i < 3; ++i) q2[i] = pc[i] + ax[i] * t2; f2 = point_box(q2, pb, mb, sb, nw); }
    }
    double ts = 0.5 * (lo + hi);
    if (ts > h - 1e-9 * (1 + h)) ts = h;
    if (ts < -h + 1e-9 * (1 + h)) ts = -h;
    const double tf = ts >= 0 ? -h : h;
    int n = 0;
    double qa[3] = {pc[0] + ax[0] * ts, pc[1] + ax[1] * ts, pc[2] + ax[2] * ts};
    if (sphere_box(c0, qa, rad, pb, mb, sb, margin)) { for (int i = 0; i < 3; ++i) c0.tangent[i] = ax[i]; n = 1; }
    if (!(fabs(tf - ts) < 1e-6 + 1e-3 * h)) {
        double qb[3] = {pc[0] + ax[0] * tf, pc[1] + ax[1] * tf, pc[2] + ax[2] * tf};
        RawContact t;
        if (sphere_box(t, qb, rad, pb, mb, sb, margin)) { for (int i = 0; i < 3; ++i) t.tangent[i] = ax[i]; if (n == 0) c0 = t; else c1 = t; ++n; }
    }
    return n;
}


/* ---- box vs box (same definition, same arithmetic order and the same tie rules as oracle/cassie_oracle.c box_box):
 * separating-axis test over 15 axes, then either the incident face clipped against the reference face -- lane =
 * candidate vertex of the clipped polygon: 0-3 incident vertices, 4-7 rectangle corners, 8-23 edge crossings; at most
 * the 4 deepest are kept, in candidate order -- or one edge-edge contact (lane 0).  Everything up to the candidates is
 * wave-uniform and computed redundantly by every lane.  Returns whether this lane holds a contact. ---- */
/* run-time picks out of three / four register values by compare-and-select: an array indexed by a run-time value would be
 * placed in scratch memory (box_box_lane used to keep ~200 bytes of such arrays there: 13 scratch stores and 15 loads per
 * box pair, 1.9 GB of HBM writes per 4096-env launch of the tray model) */
WV_DEVICE double sel3(int i, double a0, double a1, double a2) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }
WV_DEVICE double sel4(int i, double a0, double a1, double a2, double a3) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
WV_DEVICE void row3(double (&r)[3], const double (&M)[3][3], int i) {
    for (int x = 0; x < 3; ++x) r[x] = sel3(i, M[0][x], M[1][x], M[2][x]);
}

WV_DEVICE bool box_box_lane(RawContact &rc, int lane, const double *p1, const double *m1, const double *s1, const double *p2, const double *m2,
                            const double *s2, double margin, int keepmax) {
    const double BB_TIE = 1e-10;
    double A[3][3], B[3][3], d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, ta[3], tb[3], C[3][3], Q[3][3];
    const double sa[3] = {s1[0], s1[1], s1[2]}, sb[3] = {s2[0], s2[1], s2[2]};
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { A[i][k] = m1[3 * k + i]; B[i][k] = m2[3 * k + i]; }
    for (int i = 0; i < 3; ++i) { ta[i] = dot3(d, A[i]); tb[i] = dot3(d, B[i]); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
    int best = -1;
    double bestscore = 0;
    bool separated = false;
#pragma unroll
    for (int k = 0; k < 15; ++k) { /* fully unrolled: every index below is a compile-time constant */
        double sep, sc;
        if (k < 3) {
            sep = fabs(ta[k]) - (sa[k] + (sb[0] * Q[k][0] + sb[1] * Q[k][1] + sb[2] * Q[k][2]));
            sc = sep;
        } else if (k < 6) {
            const int j = k - 3;
            sep = fabs(tb[j]) - (sb[j] + (sa[0] * Q[0][j] + sa[1] * Q[1][j] + sa[2] * Q[2][j]));
            sc = sep - BB_TIE;
        } else {
            const int i = (k - 6) / 3, j = (k - 6) % 3, i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double len2 = 1.0 - C[i][j] * C[i][j];
            if (len2 < 1e-6) continue;
            const double proj = ta[i2] * C[i1][j] - ta[i1] * C[i2][j];
            const double ra = sa[i1] * Q[i2][j] + sa[i2] * Q[i1][j], rb = sb[j1] * Q[i][j2] + sb[j2] * Q[i][j1];
            sep = (fabs(proj) - (ra + rb)) / sqrt(len2);
            sc = (sep < 0 ? 1.05 * sep : sep) - 2 * BB_TIE;
        }
        if (sep > margin) separated = true;
        if (best < 0 || sc > bestscore) { best = k; bestscore = sc; }
    }
    if (separated) return false;

    if (best >= 6) {
        const int i = (best - 6) / 3, j = (best - 6) % 3;
        double Ai[3], Bj[3];
        row3(Ai, A, i); row3(Bj, B, j);
        double n[3], pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
        cross3(n, Ai, Bj);
        normalize3(n);
        if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k != i) { const double sg = dot3(n, A[k]) > 0 ? 1.0 : -1.0; for (int x = 0; x < 3; ++x) pa[x] += sg * sa[k] * A[k][x]; }
            if (k != j) { const double sg = dot3(n, B[k]) > 0 ? -1.0 : 1.0; for (int x = 0; x < 3; ++x) pb[x] += sg * sb[k] * B[k][x]; }
        }
        const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        double Ci[3];
        row3(Ci, C, i);
        const double uaub = sel3(j, Ci[0], Ci[1], Ci[2]), q1 = dot3(Ai, ab), q2 = -dot3(Bj, ab), den = 1.0 - uaub * uaub;
        const double s1i = sel3(i, sa[0], sa[1], sa[2]), s2j = sel3(j, sb[0], sb[1], sb[2]);
        const double al = clampd((q1 + uaub * q2) / den, -s1i, s1i), be = clampd((uaub * q1 + q2) / den, -s2j, s2j);
        double ca[3], cb[3];
        for (int x = 0; x < 3; ++x) { ca[x] = pa[x] + al * Ai[x]; cb[x] = pb[x] + be * Bj[x]; }
        const double cd[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
        rc.dist = dot3(cd, n);
        for (int x = 0; x < 3; ++x) { rc.normal[x] = n[x]; rc.tangent[x] = 0; rc.pos[x] = 0.5 * (ca[x] + cb[x]); }
        return lane == 0 && !(rc.dist > margin);
    }

    const bool refA = best < 3;
    const int a = best % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    double R[3][3], I[3][3];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { R[i][k] = refA ? A[i][k] : B[i][k]; I[i][k] = refA ? B[i][k] : A[i][k]; }
    double pr[3], pi[3], sr[3], si[3];
    for (int x = 0; x < 3; ++x) { pr[x] = refA ? p1[x] : p2[x]; pi[x] = refA ? p2[x] : p1[x]; sr[x] = refA ? sa[x] : sb[x]; si[x] = refA ? sb[x] : sa[x]; }
    /* the reference face's normal axis and its two in-plane axes, picked once (Ra, Ra1, Ra2) */
    double Ra[3], Ra1[3], Ra2[3];
    row3(Ra, R, a); row3(Ra1, R, a1); row3(Ra2, R, a2);
    const double sra = sel3(a, sr[0], sr[1], sr[2]);
    const double dri[3] = {pi[0] - pr[0], pi[1] - pr[1], pi[2] - pr[2]};
    const double sgn = dot3(dri, Ra) >= 0 ? 1.0 : -1.0;
    double n[3] = {sgn * Ra[0], sgn * Ra[1], sgn * Ra[2]};
    int kf = 0;
    double kbest = fabs(dot3(n, I[0]));
#pragma unroll
    for (int k = 1; k < 3; ++k) { const double v = fabs(dot3(n, I[k])); if (v > kbest + 1e-9) { kbest = v; kf = k; } }
    const int k1 = (kf + 1) % 3, k2 = (kf + 2) % 3;
    double If[3], I1[3], I2[3];
    row3(If, I, kf); row3(I1, I, k1); row3(I2, I, k2);
    const double sif = sel3(kf, si[0], si[1], si[2]), si1 = sel3(k1, si[0], si[1], si[2]), si2 = sel3(k2, si[0], si[1], si[2]);
    const double isg = dot3(n, If) > 0 ? -1.0 : 1.0;
    double cr[3], ci[3];
    for (int x = 0; x < 3; ++x) { cr[x] = pr[x] + sgn * sra * Ra[x]; ci[x] = pi[x] + isg * sif * If[x]; }
    const double h1 = sel3(a1, sr[0], sr[1], sr[2]), h2 = sel3(a2, sr[0], sr[1], sr[2]);
    const double su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
    double pu[4], pv[4], pw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double x[3];
        for (int t = 0; t < 3; ++t) x[t] = ci[t] + su[q] * si1 * I1[t] + sv[q] * si2 * I2[t] - cr[t];
        pu[q] = dot3(x, Ra1); pv[q] = dot3(x, Ra2); pw[q] = dot3(x, n);
    }
    const double tol = 1e-12;
    /* lane = candidate */
    double cu = 0, cv = 0, cw = 0;
    bool valid = false;
    if (lane < 4) {
        const int q = lane;
        cu = sel4(q, pu[0], pu[1], pu[2], pu[3]); cv = sel4(q, pv[0], pv[1], pv[2], pv[3]); cw = sel4(q, pw[0], pw[1], pw[2], pw[3]);
        valid = fabs(cu) <= h1 + tol && fabs(cv) <= h2 + tol;
    } else if (lane < 8) {
        const int q = lane - 4;
        const double e1u = pu[1] - pu[0], e1v = pv[1] - pv[0], e1w = pw[1] - pw[0], e2u = pu[3] - pu[0], e2v = pv[3] - pv[0], e2w = pw[3] - pw[0];
        const double det = e1u * e2v - e1v * e2u;
        const double gu = (e1w * e2v - e1v * e2w) / det, gv = (e1u * e2w - e1w * e2u) / det;
        const double u = sel4(q, 1.0, -1.0, -1.0, 1.0) * h1, v = sel4(q, 1.0, 1.0, -1.0, -1.0) * h2;
        int pos = 0, neg = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = (e + 1) & 3;
            const double cr2 = (pu[f] - pu[e]) * (v - pv[e]) - (pv[f] - pv[e]) * (u - pu[e]);
            if (cr2 > tol) ++pos; else if (cr2 < -tol) ++neg;
        }
        cu = u; cv = v; cw = pw[0] + gu * (u - pu[0]) + gv * (v - pv[0]);
        valid = !(pos && neg);
    } else if (lane < 24) {
        const int e = (lane - 8) >> 2, l = (lane - 8) & 3, f = (e + 1) & 3;
        const bool along_u = l < 2;
        const double pue = sel4(e, pu[0], pu[1], pu[2], pu[3]), puf = sel4(f, pu[0], pu[1], pu[2], pu[3]);
        const double pve = sel4(e, pv[0], pv[1], pv[2], pv[3]), pvf = sel4(f, pv[0], pv[1], pv[2], pv[3]);
        const double pwe = sel4(e, pw[0], pw[1], pw[2], pw[3]), pwf = sel4(f, pw[0], pw[1], pw[2], pw[3]);
        const double lim = (l & 1) ? -(along_u ? h1 : h2) : (along_u ? h1 : h2);
        const double x0 = along_u ? pue : pve, x1 = along_u ? puf : pvf;
        const double y0 = along_u ? pve : pue, y1 = along_u ? pvf : puf, hy = along_u ? h2 : h1;
        const double dx = x1 - x0;
        if (!(fabs(dx) < 1e-14)) {
            const double sp = (lim - x0) / dx;
            if (sp > 0 && sp < 1) {
                const double y = y0 + sp * (y1 - y0);
                if (!(fabs(y) > hy)) {
                    cu = along_u ? lim : y; cv = along_u ? y : lim; cw = pwe + sp * (pwf - pwe);
                    valid = true;
                }
            }
        }
    }
    if (valid && cw > margin) valid = false;
    const unsigned long long vmask = wv::ballot(valid);
    bool keep = valid;
    if (wv::popc64(vmask) > keepmax) {     /* (keepmax: 4, or 8 with CM_FLAG_BOX8) */
        int rank = 0;
        for (int r = 0; r < 24; ++r) {
            const double wr = wv::readlane(cw, r);
            if (r == lane || !((vmask >> r) & 1ull)) continue;
            if (wr < cw - 1e-9 || (fabs(wr - cw) <= 1e-9 && r < lane)) ++rank;
        }
        if (rank >= keepmax) keep = false;
    }
    if (keep) {
        rc.dist = cw;
        for (int x = 0; x < 3; ++x) {
            const double px = cr[x] + cu * Ra1[x] + cv * Ra2[x] + cw * n[x];
            rc.pos[x] = px - 0.5 * cw * n[x];
            rc.normal[x] = refA ? n[x] : -n[x];
            rc.tangent[x] = 0;
        }
    }
    return keep;
}

/* ---- height field: closest feature of the terrain surface over every grid triangle under the sample sphere's footprint
 *      (same definition, same candidate order as oracle/cassie_oracle.c) ---- */
WV_DEVICE void closest_on_triangle(const double *p, const double *a, const double *b, const double *c, double *q) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    /* the seven Voronoi regions, first match wins as in the oracle; the regions that divide (the three edges, the face) hand their
     * numerator and denominator to ONE division -- a wave whose lanes fall into different regions would otherwise run all four */
    int region;
    double num = 0.0, den = 1.0;
    if (d1 <= 0 && d2 <= 0) region = 0;                                  /* vertex a */
    else if (d3 >= 0 && d4 <= d3) region = 1;                            /* vertex b */
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { region = 2; num = d1; den = d1 - d3; }                                    /* edge ab */
    else if (d6 >= 0 && d5 <= d6) region = 3;                            /* vertex c */
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { region = 4; num = d2; den = d2 - d6; }                                    /* edge ac */
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { region = 5; num = d4 - d3; den = (d4 - d3) + (d5 - d6); }   /* edge bc */
    else { region = 6; num = 1.0; den = va + vb + vc; }                  /* the face */
    const double t = num / den;
    double v = 0, w = 0;
    if (region == 1) v = 1;
    else if (region == 2) v = t;
    else if (region == 3) w = 1;
    else if (region == 4) w = t;
    else if (region == 5) { w = t; v = 1 - w; }
    else if (region == 6) { v = vb * t; w = vc * t; }
    for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i] + w * ac[i];
}
/* One grid triangle against the sample centre p (height-field frame): a closer feature replaces (best, bv, bdiv).  The contact
 * normal of the winner is bdiv ? bv / best : bv -- the division of the closest point's offset by its length is left to whoever
 * ends up with the winning candidate (three divisions per triangle otherwise, for candidates that mostly lose) -- and the
 * triangle's unit normal is formed only where it is needed: under the triangle's plane, within rounding of it (the oracle decides
 * above / below by the sign of the UNIT normal's product with the offset; the raw normal's product has the same sign wherever it is
 * 1e-12 of its terms' magnitude away from zero), or for a centre on the surface itself.  Values are the oracle's bit for bit. */
WV_DEVICE void hfield_triangle(const double *p, const double *a, const double *b, const double *c, double &best, double *bv, bool &bdiv) {
    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, n[3];
    cross3(n, ab, ac);
    if (n[2] < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    const double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]};
    const double sraw = dot3(n, ap);
    const double bound = (fabs(n[0]) + fabs(n[1]) + fabs(n[2])) * (fabs(ap[0]) + fabs(ap[1]) + fabs(ap[2]));
    bool unit = false;
    double s = sraw;
    auto normalise = [&]() {
        const double inv = 1.0 / sqrt(dot3(n, n));
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
        unit = true;
    };
    if (!(sraw > 1e-12 * bound)) { normalise(); s = dot3(n, ap); }
    if (s >= 0) {
        double q[3];
        closest_on_triangle(p, a, b, c, q);
        const double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]}, len = sqrt(dot3(d, d));
        if (len < best) {
            best = len;
            if (len > 1e-12) { bv[0] = d[0]; bv[1] = d[1]; bv[2] = d[2]; bdiv = true; }
            else { if (!unit) normalise(); bv[0] = n[0]; bv[1] = n[1]; bv[2] = n[2]; bdiv = false; }
        }
    } else {
        const double e0 = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]);
        const double e1 = (c[0] - b[0]) * (p[1] - b[1]) - (c[1] - b[1]) * (p[0] - b[0]);
        const double e2 = (a[0] - c[0]) * (p[1] - c[1]) - (a[1] - c[1]) * (p[0] - c[0]);
        const bool inside = (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
        if (inside && s < best) { best = s; bv[0] = n[0]; bv[1] = n[1]; bv[2] = n[2]; bdiv = false; }
    }
}
WV_DEVICE int hfield_sphere(RawContact &c, ModelPtr m, const float *data, const double *ph, const double *mh, const double *ps,
                            double r, double margin) {
    if (!data || m->hfield_nrow < 2 || m->hfield_ncol < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    double d[3] = {ps[0] - ph[0], ps[1] - ph[1], ps[2] - ph[2]}, p[3];
    mulmatTvec3(p, mh, d);
    const double reach = r + (margin > 0 ? margin : 0);
    if (fabs(p[0]) > sx + reach || fabs(p[1]) > sy + reach || p[2] - r > sz + margin) return 0;
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
    int j0 = (int)floor((p[0] - reach + sx) / dx), j1 = (int)floor((p[0] + reach + sx) / dx);
    int i0 = (int)floor((p[1] - reach + sy) / dy), i1 = (int)floor((p[1] + reach + sy) / dy);
    if (j0 < 0) j0 = 0;
    if (i0 < 0) i0 = 0;
    if (j1 > nc - 2) j1 = nc - 2;
    if (i1 > nr - 2) i1 = nr - 2;
    double best = 1e300, bv[3] = {0, 0, 1};
    bool bdiv = false;
    /* touch the first and the last sample of every grid row of the footprint before any of them is needed: all the
     * footprint's cache lines are then in flight together (one memory latency instead of one per cell) */
    float touch = 0.0f;
    for (int i = i0; i <= i1 + 1; ++i) touch += data[i * nc + j0] + data[i * nc + j1 + 1];
    if (touch == -1.2345e30f) best = 0;      /* never true for elevations in [0, 1]: keeps the loads alive */
    const double reach2 = reach * reach;
    for (int i = i0; i <= i1; ++i) {
        const double y0 = -sy + i * dy;
        const double ey = p[1] < y0 ? y0 - p[1] : (p[1] > y0 + dy ? p[1] - (y0 + dy) : 0.0);
        for (int j = j0; j <= j1; ++j) {
            const double x0 = -sx + j * dx;
            /* exact culls: a cell whose rectangle is further than the reach in plan, or whose highest corner is more than
             * the reach below the sphere, cannot hold a point within contact distance */
            const double ex = p[0] < x0 ? x0 - p[0] : (p[0] > x0 + dx ? p[0] - (x0 + dx) : 0.0);
            if (ex * ex + ey * ey > reach2) continue;
            const double z00 = sz * data[i * nc + j], z10 = sz * data[i * nc + j + 1], z01 = sz * data[(i + 1) * nc + j], z11 = sz * data[(i + 1) * nc + j + 1];
            if (p[2] - reach > fmax(fmax(z00, z10), fmax(z01, z11))) continue;
            const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
            hfield_triangle(p, v00, v10, v01, best, bv, bdiv);
            hfield_triangle(p, v11, v01, v10, best, bv, bdiv);
        }
    }
    if (best > 1e299) return 0;
    const double dist = best - r;
    if (dist > margin) return 0;
    double bn[3] = {bv[0], bv[1], bv[2]};
    if (bdiv) { bn[0] = bv[0] / best; bn[1] = bv[1] / best; bn[2] = bv[2] / best; }
    double nw[3];
    mulmatvec3(nw, mh, bn);
    c.dist = dist;
    for (int k = 0; k < 3; ++k) { c.normal[k] = nw[k]; c.pos[k] = ps[k] - nw[k] * (r + 0.5 * dist); c.tangent[k] = 0; }
    return 1;
}

/* The same for up to 64 sample spheres at once, one per lane (`mine`: this lane has one), with the WHOLE WAVE sharing the grid
 * cells under all of them: a sphere of a foot capsule covers 15 .. 25 cells and the pelvis sphere none, and with one lane
 * walking each sphere's cells the wave took as long as its slowest lane (26 k clocks, a quarter of the height-field model's
 * substep).  Here the cells of all spheres form one task list (sphere by sphere, a sphere's cells in its scan order), lane t
 * of round q takes task 64 q + t -- looks its sphere up in a table the spheres wrote their lane numbers into, fetches the
 * sphere from that lane, tests the cell's two triangles -- and a segmented minimum scan over the lanes of one sphere hands the round's
 * closest feature to the sphere's record in LDS.  Ties go to the earlier task, and a sphere's earlier rounds win over later
 * ones, which is the strict `<` of the sequential scan: results are those of hfield_sphere bit for bit.
 * work: HF_WINDOW bytes (the sphere of every task of a window) + HF_RECW doubles per lane (closest distance, the winning candidate's
 * vector, whether the normal is that vector divided by the distance: hfield_triangle). */
#ifdef CK_EMULATED
constexpr int HF_WINDOW = 128;  /* (the CPU emulator's tests go through several windows per pass; results do not depend on the size) */
#else
constexpr int HF_WINDOW = 1024;
#endif
constexpr int HF_RECW = 5;
constexpr int HF_WORK_BYTES = HF_WINDOW + (int)sizeof(double) * HF_RECW * WV_WAVE;
WV_DEVICE int hfield_spheres_wave(RawContact &c, ModelPtr m, const float *data, const double *ph, const double *mh, bool mine, const double *ps,
                                  double r, double margin, int lane, double *work) {
    const bool grid_ok = data && m->hfield_nrow >= 2 && m->hfield_ncol >= 2;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc > 1 ? nc - 1 : 1), dy = 2 * sy / (nr > 1 ? nr - 1 : 1);
    double pl[3] = {0, 0, 0};
    const double reach = r + (margin > 0 ? margin : 0);
    int i0 = 0, j0 = 0, wj = 1, ncell = 0;
    if (mine && grid_ok) {
        double d[3] = {ps[0] - ph[0], ps[1] - ph[1], ps[2] - ph[2]};
        mulmatTvec3(pl, mh, d);
        if (!(fabs(pl[0]) > sx + reach || fabs(pl[1]) > sy + reach || pl[2] - r > sz + margin)) {
            int j1 = (int)floor((pl[0] + reach + sx) / dx), i1 = (int)floor((pl[1] + reach + sy) / dy);
            j0 = (int)floor((pl[0] - reach + sx) / dx); i0 = (int)floor((pl[1] - reach + sy) / dy);
            if (j0 < 0) j0 = 0;
            if (i0 < 0) i0 = 0;
            if (j1 > nc - 2) j1 = nc - 2;
            if (i1 > nr - 2) i1 = nr - 2;
            if (j1 >= j0 && i1 >= i0) { wj = j1 - j0 + 1; ncell = wj * (i1 - i0 + 1); }
        }
    }
    /* running cell counts (inclusive), lane by lane */
    int endx = ncell;
#pragma unroll
    for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) { const int t = wv::shfl_i(endx, (lane - dlt) & 63); if (lane >= dlt) endx += t; }
    const int total = wv::shfl_i(endx, WV_WAVE - 1);
    int maxcell = ncell;
#pragma unroll
    for (int msk = WV_WAVE / 2; msk >= 1; msk /= 2) { const int o = wv::shfl_i(maxcell, lane ^ msk); maxcell = o > maxcell ? o : maxcell; }
    unsigned char *const owner = (unsigned char *)work;      /* the sphere (lane) of every task of a window of HF_WINDOW tasks */
    double *const rec = work + HF_WINDOW / 8 + HF_RECW * lane;
    rec[0] = 1e300; rec[1] = 0; rec[2] = 0; rec[3] = 1; rec[4] = 0;
    const int start = endx - ncell;
    for (int win = 0; win < total; win += HF_WINDOW) {
        /* every sphere writes its lane over its tasks of this window */
        for (int cc = 0; cc < maxcell; ++cc) {
            const int at = start + cc - win;
            if (cc < ncell && at >= 0 && at < HF_WINDOW) owner[at] = (unsigned char)lane;
        }
        wv::sync();
        const int wend = total - win < HF_WINDOW ? total - win : HF_WINDOW;
        /* a round's tasks: sphere, cell, the cell's four heights -- requested one round ahead of their use, so that the trip to
         * memory runs under the previous round's triangles */
        struct Task { bool act; int own; double q0, q1, q2, qreach, x0, y0; float h00, h10, h01, h11; bool cull; };
        auto request = [&](int base, Task &t) {
            const int task = base + lane;
            t.act = task < wend;
            t.own = t.act ? (int)owner[task] : lane;
            t.q0 = wv::shfl(pl[0], t.own); t.q1 = wv::shfl(pl[1], t.own); t.q2 = wv::shfl(pl[2], t.own); t.qreach = wv::shfl(reach, t.own);
            const int qi0 = wv::shfl_i(i0, t.own), qj0 = wv::shfl_i(j0, t.own), qwj = wv::shfl_i(wj, t.own), qstart = wv::shfl_i(start, t.own);
            const int cidx = t.act ? win + task - qstart : 0;
            const int ci = (int)(((float)cidx + 0.5f) * (1.0f / (float)qwj)); /* cidx / qwj: the quotient's distance from an integer is at least 0.5 / qwj */
            const int i = qi0 + ci, j = qj0 + (cidx - ci * qwj);
            t.y0 = -sy + i * dy; t.x0 = -sx + j * dx;
            const double ey = t.q1 < t.y0 ? t.y0 - t.q1 : (t.q1 > t.y0 + dy ? t.q1 - (t.y0 + dy) : 0.0);
            const double ex = t.q0 < t.x0 ? t.x0 - t.q0 : (t.q0 > t.x0 + dx ? t.q0 - (t.x0 + dx) : 0.0);
            t.cull = !t.act || ex * ex + ey * ey > t.qreach * t.qreach;
            t.h00 = t.h10 = t.h01 = t.h11 = 0.0f;
            if (!t.cull) { t.h00 = data[i * nc + j]; t.h10 = data[i * nc + j + 1]; t.h01 = data[(i + 1) * nc + j]; t.h11 = data[(i + 1) * nc + j + 1]; }
        };
        Task cur, nxt;
        request(0, cur);
        for (int base = 0; base < wend; base += WV_WAVE) {
            if (base + WV_WAVE < wend) request(base + WV_WAVE, nxt); /* (wave-uniform) */
            else { nxt.act = false; nxt.cull = true; nxt.own = lane; }
            double best = 1e300, bn[3] = {0, 0, 1};
            bool bdiv = false;
            if (!cur.cull) {
                const double z00 = sz * cur.h00, z10 = sz * cur.h10, z01 = sz * cur.h01, z11 = sz * cur.h11;
                if (!(cur.q2 - cur.qreach > fmax(fmax(z00, z10), fmax(z01, z11)))) {
                    const double q[3] = {cur.q0, cur.q1, cur.q2}, x0 = cur.x0, y0 = cur.y0;
                    const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
                    hfield_triangle(q, v00, v10, v01, best, bn, bdiv);
                    hfield_triangle(q, v11, v01, v10, best, bn, bdiv);
                }
            }
            /* the closest feature among the lanes of one sphere, earlier tasks first: segmented inclusive scan of (distance, lane) */
            double sv = best;
            int ssrc = lane;
            const int seg = cur.act ? cur.own : -1 - lane;
#pragma unroll
            for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) {
                const int from = (lane - dlt) & 63;
                const double ov = wv::shfl(sv, from);
                const int osrc = wv::shfl_i(ssrc, from), oseg = wv::shfl_i(seg, from);
                if (lane >= dlt && oseg == seg && !(sv < ov)) { sv = ov; ssrc = osrc; }
            }
            const int nseg = wv::shfl_i(seg, (lane + 1) & 63);
            const double w0 = wv::shfl(bn[0], ssrc), w1 = wv::shfl(bn[1], ssrc), w2 = wv::shfl(bn[2], ssrc);
            const int wdiv = wv::shfl_i(bdiv ? 1 : 0, ssrc);
            if (cur.act && (lane == WV_WAVE - 1 || nseg != seg)) {
                double *const o = work + HF_WINDOW / 8 + HF_RECW * cur.own;
                if (sv < o[0]) { o[0] = sv; o[1] = w0; o[2] = w1; o[3] = w2; o[4] = (double)wdiv; }
            }
            wv::sync();
            cur = nxt;
        }
    }
    wv::sync();
    const double best = rec[0];
    if (!mine || best > 1e299) return 0;
    const double dist = best - r;
    if (dist > margin) return 0;
    double bn[3] = {rec[1], rec[2], rec[3]};
    if (rec[4] != 0.0) { bn[0] = rec[1] / best; bn[1] = rec[2] / best; bn[2] = rec[3] / best; } /* (the winner's division: hfield_triangle) */
    double nw[3];
    mulmatvec3(nw, mh, bn);
    c.dist = dist;
    for (int k = 0; k < 3; ++k) { c.normal[k] = nw[k]; c.pos[k] = ps[k] - nw[k] * (r + 0.5 * dist); c.tangent[k] = 0; }
    return 1;
}

/* parks one detected contact (geometry only) in the contact list; finish_contacts completes the entries */
template <class SH>
WV_DEVICE void write_raw_contact(SH &S, int slot, int pair, const RawContact &r) {
    S.c_dist[slot] = r.dist;
    S.c_pair[slot] = pair;
    for (int i = 0; i < 3; ++i) { S.c_pos[slot][i] = r.pos[i]; S.c_frame[slot][i] = r.normal[i]; S.c_frame[slot][3 + i] = r.tangent[i]; }
}

/* CM_FLAG_HFPRISM (same definition, same candidate order as oracle/cassie_oracle.c hfield_prism_contacts): ONE CONTACT PER
 * PENETRATED GRID TRIANGLE under every sphere / capsule that has a height-field pair.  Lane = pair first (its capsule in the
 * height field's frame, the cells under its bounding rectangle, its sample count -> a record in `hp`), then lane = (pair, cell,
 * triangle) KEY in the oracle's order -- pair order, cells row-major, the triangle (v00, v10, v01) of a cell before (v11, v01, v10) --
 * 64 keys to a round (round 6: the keys of the cells that survive a cull pass over (pair, cell), see below): a key's lane walks the
 * capsule's sample spheres (no further apart than the radius; exact culls by plan distance and by the cell's highest corner skip
 * most), keeps the deepest, and a key whose deepest sample is within the margin is a contact.  Ballots put the contacts into the list in key order, which is the oracle's.  Returns the number of contacts FOUND;
 * those past the list's `room` slots are not written (the caller caps or hands the substep over).
 * hp: scratch, HP_REC doubles per height-field pair the model format allows + 2 KB for the task table (the idle velocity tiles). */
constexpr int HP_REC = 16;
template <class SH>
WV_DEVICE int hfield_prism_wave(SH &S, ModelPtr m, const float *data, int lane, double *hp, int room) {
    const int nhf = m->nhfpair, nc = m->hfield_ncol, nr = m->hfield_nrow;
    if (!data || nr < 2 || nc < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
    /* ---- lane = pair ---- */
    int nkeys = 0;
    if (lane < nhf) {
        const int p = m->hfpair[lane];
        const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], t2 = m->pair_type[p] >> 8;
        const double margin = m->pair_margin[p], r = m->pair_size[p][3], h = t2 == CM_GEOM_CAPSULE ? m->pair_size[p][4] : 0.0;
        const double *ph = S.x.s.geom_xpos[g1], *mh = S.x.s.geom_xmat[g1], *pc = S.x.s.geom_xpos[g2], *mc = S.x.s.geom_xmat[g2];
        const double axw[3] = {mc[2], mc[5], mc[8]}, d[3] = {pc[0] - ph[0], pc[1] - ph[1], pc[2] - ph[2]};
        double p0[3], ax[3];
        mulmatTvec3(p0, mh, d);
        mulmatTvec3(ax, mh, axw);
        const double reach = r + (margin > 0 ? margin : 0);
        int i0 = 0, j0 = 0, wj = 1, ncell = 0;
        if (!(p0[2] - h * fabs(ax[2]) - r > sz + margin)) {
            const double xa = p0[0] - h * fabs(ax[0]) - reach, xb = p0[0] + h * fabs(ax[0]) + reach;
            const double ya = p0[1] - h * fabs(ax[1]) - reach, yb = p0[1] + h * fabs(ax[1]) + reach;
            int j1 = (int)floor((xb + sx) / dx), i1 = (int)floor((yb + sy) / dy);
            j0 = (int)floor((xa + sx) / dx); i0 = (int)floor((ya + sy) / dy);
            if (j0 < 0) j0 = 0;
            if (i0 < 0) i0 = 0;
            if (j1 > nc - 2) j1 = nc - 2;
            if (i1 > nr - 2) i1 = nr - 2;
            if (j1 >= j0 && i1 >= i0) { wj = j1 - j0 + 1; ncell = wj * (i1 - i0 + 1); }
        }
        int ns = h > 0 ? 1 + (int)ceil(2 * h / r) : 1;
        if (ns > CM_HP_MAXS) ns = CM_HP_MAXS;
        double *rec = hp + HP_REC * lane;
        for (int i = 0; i < 3; ++i) { rec[i] = p0[i]; rec[3 + i] = ax[i]; }
        rec[6] = r; rec[7] = h; rec[8] = margin; rec[9] = (double)ns; rec[10] = (double)i0; rec[11] = (double)j0; rec[12] = (double)wj;
        rec[13] = (double)p; rec[14] = (double)g2;
        nkeys = ncell;      /* (round 6: the scan below is over CELLS; a cell that survives the cull pass is two keys) */
    }
    int endx = nkeys;
#pragma unroll
    for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) { const int t = wv::shfl_i(endx, (lane - dlt) & 63); if (lane >= dlt) endx += t; }
    const int total = wv::shfl_i(endx, WV_WAVE - 1);
    const int start = endx - nkeys;
    wv::sync();
    int ncon = 0;
    const int g1h = m->hfield_geom;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    /* Round 6: THREE STEPS per batch of 64 cells.  A key's lane used to walk the capsule's samples itself, and a round of keys cost
     * what its busiest sample index cost: under a foot every index has some key in reach, so a round was ns (9 .. 12) point-triangle
     * tests long whatever a single key needed, and most cells under a capsule's bounding rectangle are out of every sample's reach.
     *   1. lane = (pair, cell): WHICH samples come within the culls' distance of the cell -- pass 3's culls with a 1e-9 margin and
     *      cheap sample positions -- as a bit mask; a cell with an empty mask is done.
     *   2. the TASKS (cell, triangle, sample of the mask) of the surviving cells are numbered in the oracle's order -- cell, triangle
     *      0 before 1, samples ascending -- by a running count over the lanes; a byte table maps task -> cell lane.
     *   3. lane = task, 64 to a round: ONE exact cull + point-triangle test (the very expressions of the one-pass version); the cell's
     *      lane then walks ITS tasks of the round in order and keeps the first strict minimum per triangle -- the comparisons the key's
     *      lane used to make, in their order -- and picks the winner's normal up from the task's lane.
     * The contacts and their order are what the one-pass version gave, bit for bit. */
    unsigned char *const tabb = (unsigned char *)(hp + HP_REC * CM_MAXHFPAIR);   /* 64 cells x at most 2 x 16 tasks */
    for (int cbase = 0; cbase < total; cbase += WV_WAVE) {
        const int c = cbase + lane;
        const bool actc = c < total;
        int h = 0, hstart = 0;
        for (int hh = 0; hh < nhf; ++hh) {
            const int st = wv::shfl_i(start, hh), en = wv::shfl_i(endx, hh);
            if (c >= st && c < en) { h = hh; hstart = st; }
        }
        const int q = actc ? c - hstart : 0;
        const double *recc = hp + HP_REC * h;
        int smask = 0;
        {
            const double p0[3] = {recc[0], recc[1], recc[2]}, ax[3] = {recc[3], recc[4], recc[5]}, r = recc[6], hl = recc[7], margin = recc[8];
            const int ns = actc ? (int)recc[9] : 0, i0 = (int)recc[10], j0 = (int)recc[11], wj = (int)recc[12];
            const int ci = (int)(((float)q + 0.5f) * (1.0f / (float)wj));
            const int i = i0 + ci, j = j0 + (q - ci * wj);
            const double x0 = -sx + j * dx, y0 = -sy + i * dy, reach = r + (margin > 0 ? margin : 0);
            double zmax = 0;
            if (actc) {
                const double z00 = sz * data[i * nc + j], z10 = sz * data[i * nc + j + 1], z01 = sz * data[(i + 1) * nc + j], z11 = sz * data[(i + 1) * nc + j + 1];
                zmax = fmax(fmax(z00, z10), fmax(z01, z11));
            }
            const double step = ns > 1 ? 2.0 * hl / (ns - 1) : 0.0, reach2 = reach * reach * (1.0 + 1e-9) + 1e-18;
            const double ztop = zmax + reach + 1e-9 * (1.0 + fabs(zmax) + fabs(p0[2]) + hl);
            for (int k = 0; k < CM_HP_MAXS; ++k) {
                if (wv::ballot(k < ns) == 0ull) break; /* (wave-uniform: no lane has a k-th sample) */
                if (k >= ns) continue;
                const double tk = ns > 1 ? hl - k * step : 0.0;
                const double px = p0[0] + tk * ax[0], py = p0[1] + tk * ax[1], pz = p0[2] + tk * ax[2];
                const double ex = px < x0 ? x0 - px : (px > x0 + dx ? px - (x0 + dx) : 0.0);
                const double ey = py < y0 ? y0 - py : (py > y0 + dy ? py - (y0 + dy) : 0.0);
                if (!(ex * ex + ey * ey > reach2 || pz > ztop)) smask |= 1 << k;
            }
        }
        const int mynp = wv::popc64((unsigned long long)(unsigned)smask), cnt = 2 * mynp;
        int tend = cnt;
#pragma unroll
        for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) { const int t = wv::shfl_i(tend, (lane - dlt) & 63); if (lane >= dlt) tend += t; }
        const int ttot = wv::shfl_i(tend, WV_WAVE - 1), tstart = tend - cnt;
        if (ttot == 0) continue; /* (wave-uniform) */
        for (int i = 0; i < 2 * CM_HP_MAXS; ++i) {
            if (wv::ballot(i < cnt) == 0ull) break;
            if (i < cnt) tabb[tstart + i] = (unsigned char)lane;
        }
        wv::sync();
        double best0 = 1e300, best1 = 1e300, bn0[3] = {0, 0, 1}, bn1[3] = {0, 0, 1}, bt0 = 0, bt1 = 0;
        bool bdiv0 = false, bdiv1 = false;
        for (int tbase = 0; tbase < ttot; tbase += WV_WAVE) {
            const int t = tbase + lane;
            const bool act = t < ttot;
            const int src = act ? (int)tabb[t] : 0;
            const int h2 = wv::shfl_i(h, src), cell = wv::shfl_i(q, src), sm2 = wv::shfl_i(smask, src), ts2 = wv::shfl_i(tstart, src);
            const int np2 = wv::popc64((unsigned long long)(unsigned)sm2), li = act ? t - ts2 : 0, tri = li >= np2 ? 1 : 0, kidx = tri ? li - np2 : li;
            int k = 0;
            {
                int seen = 0;
#pragma unroll
                for (int bpos = 0; bpos < CM_HP_MAXS; ++bpos) { if ((sm2 >> bpos) & 1) { if (seen == kidx) k = bpos; ++seen; } }
            }
            const double *rec = hp + HP_REC * h2;
            const double p0[3] = {rec[0], rec[1], rec[2]}, ax[3] = {rec[3], rec[4], rec[5]}, r = rec[6], hl = rec[7], margin = rec[8];
            const int ns = (int)rec[9], i0 = (int)rec[10], j0 = (int)rec[11], wj = (int)rec[12];
            const int ci = (int)(((float)cell + 0.5f) * (1.0f / (float)wj)); /* cell / wj: the quotient's distance from an integer is at least 0.5 / wj */
            const int i = i0 + ci, j = j0 + (cell - ci * wj);
            const double x0 = -sx + j * dx, y0 = -sy + i * dy, reach = r + (margin > 0 ? margin : 0);
            double cur = 1e300, cn[3] = {0, 0, 1}, tk = 0;
            bool cdiv = false;
            if (act) {
                const double z00 = sz * data[i * nc + j], z10 = sz * data[i * nc + j + 1], z01 = sz * data[(i + 1) * nc + j], z11 = sz * data[(i + 1) * nc + j + 1];
                const double zmax = fmax(fmax(z00, z10), fmax(z01, z11));
                const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
                tk = ns > 1 ? hl * (1.0 - 2.0 * k / (ns - 1)) : 0.0;
                const double p[3] = {p0[0] + tk * ax[0], p0[1] + tk * ax[1], p0[2] + tk * ax[2]};
                /* exact culls: a sample further from the cell's rectangle than the reach in plan, or more than the reach above the
                 * cell's highest corner, is not within contact distance of either of its triangles */
                const double ex = p[0] < x0 ? x0 - p[0] : (p[0] > x0 + dx ? p[0] - (x0 + dx) : 0.0);
                const double ey = p[1] < y0 ? y0 - p[1] : (p[1] > y0 + dy ? p[1] - (y0 + dy) : 0.0);
                if (!(ex * ex + ey * ey > reach * reach || p[2] - reach > zmax)) {
                    if (tri == 0) hfield_triangle(p, v00, v10, v01, cur, cn, cdiv); else hfield_triangle(p, v11, v01, v10, cur, cn, cdiv);
                }
            }
            /* lane = cell again: its tasks of this round, in order */
            const int lo = tstart > tbase ? tstart : tbase, hi = tstart + cnt < tbase + WV_WAVE ? tstart + cnt : tbase + WV_WAVE;
            int win0 = -1, win1 = -1;
            for (int i2 = 0; i2 < 2 * CM_HP_MAXS; ++i2) {
                const bool mine = lo + i2 < hi;
                if (wv::ballot(mine) == 0ull) break;
                const int sl = mine ? lo + i2 - tbase : lane;
                const double v = wv::shfl(cur, sl);
                if (mine) {
                    if (lo + i2 - tstart < mynp) { if (v < best0) { best0 = v; win0 = sl; } }
                    else if (v < best1) { best1 = v; win1 = sl; }
                }
            }
            if (wv::ballot(win0 >= 0 || win1 >= 0) != 0ull) {
                const int s0 = win0 >= 0 ? win0 : lane, s1 = win1 >= 0 ? win1 : lane;
                const double a0 = wv::shfl(cn[0], s0), a1 = wv::shfl(cn[1], s0), a2 = wv::shfl(cn[2], s0), at = wv::shfl(tk, s0);
                const int ad = wv::shfl_i(cdiv ? 1 : 0, s0);
                const double c0 = wv::shfl(cn[0], s1), c1 = wv::shfl(cn[1], s1), c2 = wv::shfl(cn[2], s1), ct = wv::shfl(tk, s1);
                const int cd = wv::shfl_i(cdiv ? 1 : 0, s1);
                if (win0 >= 0) { bn0[0] = a0; bn0[1] = a1; bn0[2] = a2; bt0 = at; bdiv0 = ad != 0; }
                if (win1 >= 0) { bn1[0] = c0; bn1[1] = c1; bn1[2] = c2; bt1 = ct; bdiv1 = cd != 0; }
            }
        }
        /* lane = cell: its two keys' verdicts, into the list in key order */
        {
            const double r = recc[6], hl = recc[7], margin = recc[8];
            const int pidx = (int)recc[13], g2 = (int)recc[14];
            if (bdiv0) { bn0[0] = bn0[0] / best0; bn0[1] = bn0[1] / best0; bn0[2] = bn0[2] / best0; } /* (the deepest sample's division: hfield_triangle) */
            if (bdiv1) { bn1[0] = bn1[0] / best1; bn1[1] = bn1[1] / best1; bn1[2] = bn1[2] / best1; }
            const double dist0 = best0 - r, dist1 = best1 - r;
            const bool hit0 = cnt > 0 && best0 < 1e299 && !(dist0 > margin), hit1 = cnt > 0 && best1 < 1e299 && !(dist1 > margin);
            const unsigned long long hb0 = wv::ballot(hit0), hb1 = wv::ballot(hit1);
            const int slot0 = ncon + wv::popc64(hb0 & below) + wv::popc64(hb1 & below), slot1 = slot0 + (hit0 ? 1 : 0);
            if (hit0 || hit1) {
                const double *mh = S.x.s.geom_xmat[g1h], *pc = S.x.s.geom_xpos[g2], *mc = S.x.s.geom_xmat[g2];
                const double axw[3] = {mc[2], mc[5], mc[8]};
                for (int which = 0; which < 2; ++which) {
                    const bool hit = which ? hit1 : hit0;
                    const int slot = which ? slot1 : slot0;
                    if (!hit || slot >= room) continue;
                    const double *bn = which ? bn1 : bn0;
                    const double bt = which ? bt1 : bt0, dist = which ? dist1 : dist0;
                    double nw[3];
                    mulmatvec3(nw, mh, bn);
                    RawContact rc;
                    rc.dist = dist;
                    for (int x = 0; x < 3; ++x) {
                        const double psw = pc[x] + bt * axw[x];
                        rc.normal[x] = nw[x]; rc.pos[x] = psw - nw[x] * (r + 0.5 * dist); rc.tangent[x] = hl > 0 ? axw[x] : 0.0;
                    }
                    write_raw_contact(S, slot, pidx, rc);
                }
            }
            ncon += wv::popc64(hb0) + wv::popc64(hb1);
        }
        wv::sync(); /* (the table is free again) */
    }
    wv::sync();
    return ncon;
}

/* lane = contact: contact frame from (normal, tangent hint) and the pair's pre-mixed parameters (model compile
 * time, cm_model_t::pair_*), once per contact and outside the divergent pair loops */
template <class SH>
WV_DEVICE void finish_contacts(SH &S, ModelPtr m, ParamPtr P, int lane, int ncon) {
    if (lane < ncon) {
        const int p = S.c_pair[lane];
        double fr[9];
        for (int i = 0; i < 6; ++i) fr[i] = S.c_frame[lane][i];
        fr[6] = fr[7] = fr[8] = 0;
        make_frame(fr);
        for (int i = 0; i < 9; ++i) S.c_frame[lane][i] = fr[i];
        S.c_g1[lane] = m->pair_geom1[p]; S.c_g2[lane] = m->pair_geom2[p];
        S.c_margin[lane] = m->pair_includemargin[p];
        S.c_dim[lane] = m->pair_condim[p];
        for (int i = 0; i < 2; ++i) S.c_solref[lane][i] = m->pair_solref[p][i];
        for (int i = 0; i < 5; ++i) S.c_solimp[lane][i] = m->pair_solimp[p][i];
        for (int i = 0; i < 3; ++i) S.c_fri[lane][i] = P->pair_friction[p][i];
        for (int k = 0; k < 2; ++k) { S.c_root[lane][k] = m->pair_root[p][k]; S.c_dofmask[lane][k] = m->pair_dofmask[p][k]; }
        S.c_tran[lane] = P->pair_invweight[p];
    }
}

WV_DEVICE double impedance(const double *solimp, double pos, double margin) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    if (dmin == dmax || width <= CM_MINVAL) return 0.5 * (dmin + dmax);
    double x = fabs((pos - margin) / width);
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
    else y = x; /* other exponents are rejected when the model is compiled (mjcf_loader.cpp) */
    return dmin + y * (dmax - dmin);
}

}  // namespace ck
#endif
