/*
 * pk_factor_solve.h -- dof-tree sparsity (LPack), the L^T D L factorisations, the PGS row chain, triangular solves
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_FACTOR_SOLVE_H
#define CASSIE_PK_FACTOR_SOLVE_H

namespace ck {

/* dof-tree sparsity: compile-time tables for the in-scope models (topo_static.h), or the model's own
 * masks for anything else */
struct TopoRuntime { static constexpr bool is_static = false; static constexpr bool packed = false; static constexpr int nv = 0; };

/* Where entry (k, i < k) of a factor lives in EnvShared::Lp / LHp.
 *   dense  : the full lower triangle by rows, (k, i) at k(k+1)/2 + i; the diagonal slot of a row is never read and takes
 *            the row stores of the lanes at or past the diagonal.  A lane's row or column index is base + immediate.
 *   packed : (TOPO::packed) a dof's ancestors are trunk dofs or dofs of its own block (TOPO::bstart), so row k keeps
 *            only [its trunk entries | the entries of its block below k]: 392 slots instead of 820 for the 40-dof
 *            tray model, 13.7 KB less LDS for the two factors, which is what lets four of its workgroups share a CU.
 *            Costs a few integer ops per staged entry where a lane addresses its own row / column. */
template <class TOPO, int NVP>
struct LPack {
    static constexpr bool packed = TOPO::packed;
    static constexpr int trunk() { if constexpr (TOPO::packed) return TOPO::trunk; else return 0; }
    static constexpr int bs(int k) { /* first dof of k's block */
        if constexpr (TOPO::packed) { int s = 0; for (int b = 0; b < TOPO::nblock; ++b) if (TOPO::bstart[b] <= k) s = TOPO::bstart[b]; return s; }
        else return 0;
    }
    static constexpr int len(int k) { if constexpr (TOPO::packed) return k < trunk() ? k : trunk() + (k - bs(k)); else return k + 1; }
    static constexpr int base(int k) { int s = 0; for (int j = 0; j < k; ++j) s += len(j); return s; }
    static constexpr int count = base(NVP) + (TOPO::packed ? 1 : 0);
    static constexpr int dump = count - 1; /* packed: the slot that takes the stores of lanes outside the row */
    static constexpr bool has(int k, int i) { return i < k && (!TOPO::packed || i < trunk() || i >= bs(k)); }
    static constexpr int idx(int k, int i) { /* compile-time (k, i), has(k, i) */
        if constexpr (TOPO::packed) return base(k) + (i < trunk() ? i : trunk() + i - bs(k)); else return CK_TRI(k, i);
    }
    static constexpr bool covers() { /* every ancestor pair of the topology has a slot */
        if constexpr (TOPO::packed) {
            for (int k = 0; k < NVP; ++k) for (int i = 0; i < k; ++i) if (((TOPO::table[k] >> i) & 1ull) && !has(k, i)) return false;
        }
        return true;
    }
    static constexpr bool distinct() { /* the slots of all (k, i) pairs a row keeps are 0 .. dump - 1, each used once, in order */
        if constexpr (TOPO::packed) {
            int next = 0;
            for (int k = 0; k < NVP; ++k) for (int i = 0; i < k; ++i) if (has(k, i)) { if (idx(k, i) != next) return false; ++next; }
            return next == dump;
        }
        return true;
    }
    /* slot row k (compile time) offers lane `lane`: its entry (k, lane), else a slot nobody reads */
    static WV_DEVICE int row_slot(int k, int lane) {
        if constexpr (TOPO::packed) {
            const int b = base(k), s = bs(k), T = trunk();
            if (k < T) return lane < k ? b + lane : dump;
            return lane < T ? b + lane : (lane >= s && lane < k) ? b + T - s + lane : dump;
        } else return CK_TRI(k, 0) + (lane < k ? lane : k);
    }
    /* a lane's own row: where it starts and which block it belongs to */
    struct Row { int base, bs; };
    static WV_DEVICE Row row_of(int k_) {
        Row r = {0, 0};
        if constexpr (TOPO::packed) {
            int B = 0;
#pragma unroll
            for (int b = 1; b < TOPO::nblock; ++b) if (k_ >= TOPO::bstart[b]) { r.bs = TOPO::bstart[b]; B = base(TOPO::bstart[b]); }
            const int d = k_ - r.bs;
            r.base = B + (r.bs > 0 ? d * trunk() : 0) + d * (d - 1) / 2;
        } else r.base = CK_TRI(k_, 0);
        return r;
    }
    /* entry (k_, i) of the lane's own row, i compile time: slot, and whether the row has it (i < k_ is the caller's) */
    static WV_DEVICE bool row_has(const Row &r, int i) { if constexpr (TOPO::packed) return i < trunk() || i >= r.bs; else return true; }
    static WV_DEVICE int row_idx(const Row &r, int i) { if constexpr (TOPO::packed) return r.base + (i < trunk() ? i : trunk() - r.bs + i); else return r.base + i; }
    /* entry (k, k_) of the lane's own column, k compile time (k_ < k is the caller's) */
    static WV_DEVICE bool col_has(int k, int k_) { if constexpr (TOPO::packed) return k_ < trunk() || k_ >= bs(k); else return true; }
    static WV_DEVICE int col_idx(int k, int k_) { if constexpr (TOPO::packed) return (k_ < trunk() ? base(k) : base(k) + trunk() - bs(k)) + k_; else return CK_TRI(k, k_); }
};

template <class TOPO>
WV_DEVICE unsigned long long anc_mask(ModelPtr m, int k) {
    if constexpr (TOPO::is_static) return TOPO::table[k];
    else return m->dof_ancmask[k];
}

/* bit `c` of a per-lane mask as 0.0 / 1.0: predicates of the dense tree loops are applied by multiplication (two VALU
 * ops, no compare -> scalar mask -> select round trip, which costs ~30 clocks per use on this hardware) */
WV_DEVICE double bitf(unsigned long long mask, int c) { return (double)(unsigned)((mask >> c) & 1ull); }

/* reciprocal to full fp64 accuracy without the IEEE division sequence: hardware estimate + two Newton steps
 * (the pivots are positive and far from the denormal / overflow ranges) */
WV_DEVICE double fast_rcp(double x) {
    double r = wv::rcp_estimate(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

/* L^T D L factorisation of two tree-sparse matrices (M and M + hB) held one column per lane in registers
 * (lane j owns col[i] = A[i][j], i >= j).  Pivot-row entries travel by readlane; the (k, i) loop nest is
 * fully unrolled over the ancestor pattern and the two factorisations are interleaved so that each one's
 * dependent chain hides behind the other's.  No lane predication is needed: entries above the diagonal
 * (col[i] in lanes j > i) are never read, so they may absorb harmless updates.  On exit col[k] holds L[k][j]
 * for k > j; the pivots are returned through dinv / rsd / dinvH (wave-uniform, written by lane 0). */
template <int NVP, class TOPO>
WV_DEVICE void factor_pair_in_registers(ModelPtr m, ParamPtr P, double h, double (&col)[NVP], double (&colh)[NVP], int lane, int nv,
                                        double *dinv, double *rsd, double *dinvH) {
#pragma unroll
    for (int k = NVP - 1; k >= 0; --k) {
        if (TOPO::is_static ? k >= TOPO::nv : k >= nv) continue;
        const unsigned long long anc = anc_mask<TOPO>(m, k);
        const double arm = m->dof_armature[k]; /* diagonal terms, see the mass-matrix stage */
        const double inv = fast_rcp(wv::readlane(col[k], k) + arm), invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * P->dof_damping[k]));
        if (lane == 0) { dinv[k] = inv; rsd[k] = sqrt(inv); dinvH[k] = invh; }
        if (anc == 0ull) continue;
#pragma unroll
        for (int i = k - 1; i >= 0; --i) {
            if (!((anc >> i) & 1ull)) continue;
            const double t = wv::readlane(col[k], i) * inv, th = wv::readlane(colh[k], i) * invh; /* A[k][i] / D_k */
            col[i] -= t * col[k];
            colh[i] -= th * colh[k];
        }
        col[k] *= inv;
        colh[k] *= invh;
    }
}

/* Compile-time-topology variant: the same two factorisations, eliminated height by height.  All dofs of one
 * elimination height (TOPO::height) are mutually unrelated, so a round scales their pivot rows (one multiply per
 * matrix gives L[k][:] in every lane at once), parks them in the packed LDS factors -- where the solves want them
 * anyway -- and then applies the rank-one updates with L[k][i] fetched back as LDS broadcast reads: two FMAs and two
 * reads per ancestor pair, no scalar registers, one LDS round trip per height instead of one per dof. */
/* WHICH: 2 = both factorisations, interleaved (their chains hide each other's latency); 0 = that of M alone, 1 = that of M + hB alone
 * (the two-wave form runs the second one behind the barrier J, while wave 0 solves: only the Euler step reads it) */
template <int NVP, class TOPO, int WHICH = 2, class SH>
WV_DEVICE void factor_pair_by_height(ModelPtr m, ParamPtr P, double h, SH &S, double (&col)[NVP], double (&colh)[NVP], int lane) {
    /* The trunk dofs (the floating base: each one's ancestors are all the lower ones) come last and one to a height: a
     * round through LDS for a single dof is all latency.  They are eliminated in registers instead (below), with
     * v_readlane multipliers whose round trips overlap, so the height rounds stop where the trunk begins. */
    constexpr int first_trunk_height = TOPO::height[TOPO::trunk - 1];
#pragma unroll
    for (int s = 0; s < TOPO::nheight; ++s) {
        if (s >= first_trunk_height) continue;
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (TOPO::height[k] != s) continue;
            const double arm = m->dof_armature[k]; /* diagonal terms, see the mass-matrix stage */
            /* every lane holds the same 1/D: an unpredicated same-address store; lanes at or past the diagonal all land on one
             * unused slot of the row: an unpredicated store too */
            const int at = LPack<TOPO, NVP>::row_slot(k, lane);
            if constexpr (WHICH != 1) { const double inv = fast_rcp(wv::readlane(col[k], k) + arm); S.dinv[k] = inv; S.Lp[at] = col[k] * inv; }
            if constexpr (WHICH != 0) { const double invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * P->dof_damping[k])); S.dinvH[k] = invh; S.LHp[at] = colh[k] * invh; }
        }
        wv::sync();
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (TOPO::height[k] != s) continue;
            /* all of this dof's multipliers are fetched before the first update (the fences keep the scheduler from
             * pairing every LDS read with its own wait): one LDS latency per dof instead of one per ancestor pair */
            double t[NVP], th[NVP];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) {
                if (!((TOPO::table[k] >> i) & 1ull)) continue;
                if constexpr (WHICH != 1) t[i] = S.Lp[LPack<TOPO, NVP>::idx(k, i)];
                if constexpr (WHICH != 0) th[i] = S.LHp[LPack<TOPO, NVP>::idx(k, i)];
            }
            wv::sched_fence();
#pragma unroll
            for (int i = k - 1; i >= 0; --i) {
                if (!((TOPO::table[k] >> i) & 1ull)) continue;
                if constexpr (WHICH != 1) col[i] -= t[i] * col[k];
                if constexpr (WHICH != 0) colh[i] -= th[i] * colh[k];
            }
            wv::sched_fence();
        }
    }
    /* trunk: same arithmetic (multiplier = entry * 1/D, rounded once; update = one FMA), multipliers by v_readlane */
#pragma unroll
    for (int k = TOPO::trunk - 1; k >= 0; --k) {
        const double arm = m->dof_armature[k];
        const int at = LPack<TOPO, NVP>::row_slot(k, lane);
        if constexpr (WHICH != 1) {
            const double inv = fast_rcp(wv::readlane(col[k], k) + arm);
            S.dinv[k] = inv;
            S.Lp[at] = col[k] * inv;
            double t[TOPO::trunk];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) t[i] = wv::readlane(col[k], i) * inv;
#pragma unroll
            for (int i = k - 1; i >= 0; --i) col[i] -= t[i] * col[k];
        }
        if constexpr (WHICH != 0) {
            const double invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * P->dof_damping[k]));
            S.dinvH[k] = invh;
            S.LHp[at] = colh[k] * invh;
            double th[TOPO::trunk];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) th[i] = wv::readlane(colh[k], i) * invh;
#pragma unroll
            for (int i = k - 1; i >= 0; --i) colh[i] -= th[i] * colh[k];
        }
    }
    wv::sync();
    if constexpr (WHICH != 1) if (lane < TOPO::nv) S.rsd[lane] = sqrt(S.dinv[lane]);
}

/* Projected Gauss-Seidel sweeps, one constraint row per lane.  The per-row state is the SCALED residual
 * s_j = -res_j / A_jj, so a row's unclamped step is s itself and the serial chain per row is max, readlane, FMA:
 *     delta_I = max(s_I, lo_I);   s_j += B_jI * delta_I  for every j,   B_jI = -A_jI / A_jj  (brow, per lane).
 * Every lane evaluates its own candidate each row; only lane I's is consumed, through readlane.
 *
 * pgs_rows: the guarded sweep (MuJoCo's rule `never accept a cost increase`, evaluated row by row).  Nested so that
 * the first row index >= nrows ends the sweep with one wave-uniform branch. */
template <int I, int N>
WV_DEVICE void pgs_rows(const double (&brow)[N], int nrows, int r_, double Aii, double halfAii, double flo, double &f,
                        double &sres, double &improvement) {
    if constexpr (I < N) {
        if (I < nrows) {
            double delta = fmax(sres, flo - f); /* = max(f - res / Aii, flo) - f */
            double change = delta * (halfAii * delta - Aii * sres);
            if (change > 1e-10) { delta = 0; change = 0; } /* never accept a cost increase */
            const double dlt = wv::readlane(delta, I), chg = wv::readlane(change, I);
            if (r_ == I) f += dlt;
            improvement -= chg;
            sres += brow[I] * dlt;
            pgs_rows<I + 1, N>(brow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
        }
    }
}

/* The same sweep with the guard off the dependent chain: the row's own lane keeps the residual it started from
 * (its step follows from it), so every row's cost change -- hence the guard and the sweep's improvement -- can be
 * evaluated once, after the sweep.  The caller re-runs the sweep through pgs_rows when a guard would have fired. */
template <int I, int N>
WV_DEVICE void pgs_row_fast(const double (&brow)[N], int r_, double lo_f, double &sres, double &mys) {
    if constexpr (I < N) {
        const double delta = wv::max_raw(sres, lo_f); /* one v_max_f64: fmax() adds a canonicalising self-max to the row chain after every branch */
        if (r_ == I) mys = sres; /* the residual this row started from: its step is recomputed from it after the sweep */
        sres += brow[I] * wv::readlane(delta, I);
    }
}
/* rows go four to a (wave-uniform) branch: rows past the last one are inert -- their column of A is zero in every
 * lane and their own lane's step is finite -- so running up to three of them costs less than three more branches */
template <int I, int N>
WV_DEVICE void pgs_rows_fast(const double (&brow)[N], int nrows, int r_, double lo_f, double &sres, double &mys) {
    if constexpr (I < N) {
        if (I < nrows) {
            pgs_row_fast<I, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 1, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 2, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 3, N>(brow, r_, lo_f, sres, mys);
            pgs_rows_fast<I + 4, N>(brow, nrows, r_, lo_f, sres, mys);
        }
    }
}

/* x := L^-1 x (forward) and x := L^-T x (backward) by substitution, lane = dof: lrow / lcol hold the lane's row /
 * column of the unit-triangular factor (zeros outside its ancestors / descendants) and every hop is a v_readlane round
 * trip (~40 clocks).  With a compile-time topology the hops go LEVEL BY LEVEL of the dof tree (a dof's level = the number
 * of its ancestors): dofs of one level are mutually unrelated, so their broadcasts are all read from the same state of
 * the vector and their terms are summed before they touch it -- the dependent chain is as long as the tree is deep (13
 * for Cassie: floating base, hip, knee, shin, tarsus, crank), not as long as the dof list (32), and the forward pass
 * skips the dofs nobody descends from (their column of L is empty). */
template <class TOPO>
struct DofLevels {
    static constexpr int level(int k) { int n = 0; for (int i = 0; i < TOPO::nv; ++i) n += (int)((TOPO::table[k] >> i) & 1ull); return n; }
    static constexpr bool has_descendants(int j) { for (int k = 0; k < TOPO::nv; ++k) if ((TOPO::table[k] >> j) & 1ull) return true; return false; }
    static constexpr int depth() { int d = 0; for (int k = 0; k < TOPO::nv; ++k) if (level(k) > d) d = level(k); return d; }
};
template <int NVP, class TOPO>
WV_DEVICE double solve_forward(double z, const double (&lrow)[NVP], int lane, int nv) {
    if constexpr (TOPO::is_static) {
        typedef DofLevels<TOPO> LV;
#pragma unroll
        for (int d = 0; d < LV::depth(); ++d) {
            double t0 = 0, t1 = 0;
            int n = 0;
#pragma unroll
            for (int j = 0; j < TOPO::nv; ++j) {
                if (LV::level(j) != d || !LV::has_descendants(j)) continue;
                const double bj = wv::readlane(z, j);
                if ((n++ & 1) == 0) t0 = fma(lrow[j], bj, t0); else t1 = fma(lrow[j], bj, t1);
            }
            z -= t0 + t1;
        }
        return z;
    } else {
#pragma unroll
        for (int i = 0; i < NVP - 1; ++i) {
            if (i >= nv - 1) continue;
            z -= lrow[i] * wv::readlane(z, i);
        }
        return z;
    }
}
template <int NVP, class TOPO>
WV_DEVICE double solve_backward(double w, const double (&lcol)[NVP], int lane, int nv) {
    if constexpr (TOPO::is_static) {
        typedef DofLevels<TOPO> LV;
#pragma unroll
        for (int d = LV::depth(); d >= 1; --d) {
            double t0 = 0, t1 = 0;
            int n = 0;
#pragma unroll
            for (int j = 0; j < TOPO::nv; ++j) {
                if (LV::level(j) != d) continue;
                const double bj = wv::readlane(w, j);
                if ((n++ & 1) == 0) t0 = fma(lcol[j], bj, t0); else t1 = fma(lcol[j], bj, t1);
            }
            w -= t0 + t1;
        }
        return w;
    } else {
#pragma unroll
        for (int k = NVP - 1; k >= 1; --k) {
            if (k >= nv) continue;
            w -= lcol[k] * wv::readlane(w, k);
        }
        return w;
    }
}

}  // namespace ck
#endif
