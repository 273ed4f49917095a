"""Comparison of two result grids (test() output rows [f, g(d), var_f, var_g(d)]) under the tolerances of SURVEY.md 8(c),
with the branch-ambiguous queries (F5 mask: value variance within 1e-3 of the blend gate, or near-equal candidate
distances) COUNTED and reported separately -- never silently dropped: the RMSE bars are asserted on all queries, the
max bars on the unmasked ones."""
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
             grad_rmse=float(np.sqrt(np.mean(eg ** 2))), grad_max=float(np.abs(eg[ok]).max()),
             var_f_abs=float(ev[ok].max()), var_g_rel=float(evg[ok].max()),
             var_f_over=int((ev[ok] >= TOL["var_f_abs"]).sum()))
    return r


def within(r, keys=("sdf_rmse", "sdf_max", "grad_rmse", "grad_max", "var_f_abs", "var_g_rel")):
    return all(r[k] < TOL[k] for k in keys)


def fmt(tag, r):
    return ("%-34s n %6d masked %4d identical %.5f | SDF rmse %.2e max %.2e (all %.2e) | grad rmse %.2e max %.2e | var_f %.2e | var_g rel %.2e"
            % (tag, r["n"], r["masked"], r["identical_rows"], r["sdf_rmse"], r["sdf_max"], r["sdf_max_all"], r["grad_rmse"], r["grad_max"],
               r["var_f_abs"], r["var_g_rel"]))
