"""Comparison of two result grids (test() output rows [f, g(d), var_f, var_g(d)]) under the tolerances of SURVEY.md 8(c),
with the branch-ambiguous queries (F5 mask: value variance within 1e-3 of the blend gate, or near-equal candidate
variances) COUNTED and reported separately -- never silently dropped.  Two RMSE figures are reported: over ALL queries
(sdf_rmse / grad_rmse) and over the unmasked ones (sdf_rmse_unmasked / grad_rmse_unmasked: SURVEY 8(c) states its bars
"excluding F5-masked points"); the max bars are on the unmasked queries.  A masked query sits on one of the reference's
own discontinuities: which two clusters are blended depends on the ORDER of near-equal variances, so any two fp32
summation orders can pick different branches there (data/2D: 40 % of the demo grid lies far from all data, where every
candidate's variance is the prior)."""
import numpy as np

# SURVEY.md 8(c) "Tolerances to state"
TOL = dict(sdf_rmse=1e-5, sdf_max=1e-4, grad_rmse=1e-4, grad_max=2e-3, var_f_abs=1e-4, var_g_rel=1e-4)


def compare(a, b, flags, dim, scale):
    """a, b: [n, 2(1+dim)] float32; flags: uint8 per query (bits 2|4 = on a reference discontinuity) or None."""
    nc = 1 + dim
    tos = 3.0 / scale ** 2                       # prior of the gradient variances (1875 in 3-D)
    amb = np.zeros(a.shape[0], dtype=bool) if flags is None else (flags & (2 | 4)) != 0
    ef = (a[:, 0] - b[:, 0]).astype(np.float64)
    eg = (a[:, 1:nc] - b[:, 1:nc]).astype(np.float64)
    ev = np.abs(a[:, nc] - b[:, nc]).astype(np.float64)
    evg = np.abs(a[:, nc + 1:] - b[:, nc + 1:]).astype(np.float64) / tos
    ok = ~amb
    r = dict(n=int(a.shape[0]), masked=int(amb.sum()),
             identical_rows=float(np.mean(np.all(a == b, axis=1))),
             sdf_rmse=float(np.sqrt(np.mean(ef ** 2))), sdf_max=float(np.abs(ef[ok]).max()), sdf_max_all=float(np.abs(ef).max()),
             sdf_rmse_unmasked=float(np.sqrt(np.mean(ef[ok] ** 2))), grad_rmse_unmasked=float(np.sqrt(np.mean(eg[ok] ** 2))),
             grad_rmse=float(np.sqrt(np.mean(eg ** 2))), grad_max=float(np.abs(eg[ok]).max()),
             var_f_abs=float(ev[ok].max()), var_g_rel=float(evg[ok].max()),
             var_f_over=int((ev[ok] >= TOL["var_f_abs"]).sum()))
    return r


def within(r, keys=("sdf_rmse", "sdf_max", "grad_rmse", "grad_max", "var_f_abs", "var_g_rel")):
    return all(r[k] < TOL[k] for k in keys)


def within_same_map(r, grad_max=None):
    """The SURVEY 8(c) bars as stated (RMSE and max over the unmasked queries, every bar at its survey value) -- what two
    arithmetic orders must meet when they are compared on the SAME map and the same training sets.  grad_max: the one
    bar a caller may widen (and must say so): the survey derived 2e-3 from the first 12 bundled frames (clusters of at
    most ~1250 rows); the synthetic scene's clusters reach 2344 rows and one query of 8192 shows 2.2e-3 in one order."""
    return (r["sdf_rmse_unmasked"] < TOL["sdf_rmse"] and r["sdf_max"] < TOL["sdf_max"] and r["grad_rmse_unmasked"] < TOL["grad_rmse"]
            and r["grad_max"] < (grad_max or TOL["grad_max"]) and r["var_f_abs"] < TOL["var_f_abs"] and r["var_g_rel"] < TOL["var_g_rel"])


def fmt(tag, r):
    return ("%-34s n %6d masked %5d identical %.5f | SDF rmse %.2e (unmasked %.2e) max %.2e (all %.2e) | grad rmse %.2e (unmasked %.2e) max %.2e | var_f %.2e | var_g rel %.2e"
            % (tag, r["n"], r["masked"], r["identical_rows"], r["sdf_rmse"], r["sdf_rmse_unmasked"], r["sdf_max"], r["sdf_max_all"], r["grad_rmse"],
               r["grad_rmse_unmasked"], r["grad_max"], r["var_f_abs"], r["var_g_rel"]))
