"""Independent FP64 ray–torus solver — TEST INFRASTRUCTURE.

A different algorithm from ``oracle/trt_solve.inc`` (companion-matrix eigenvalues of the
quartic + Newton polish in float64), used to check that the oracle's build-defined torus
arithmetic is *right*, not merely self-consistent, and to tag rays whose hit/miss status
is sensitive to perturbation ("marginal" rays, excluded from accuracy — never from
oracle-vs-GPU bit-exactness — checks).

The quartic is SURVEY.md §8a T1 (verified there against ``numpy.roots``): with o ← O−C,
K = o·o + R² − r², n = o·d, a = dx²+dz², b = ox·dx+oz·dz, c = ox²+oz², dd = d·d:

    dd²·t⁴ + 4·dd·n·t³ + (4n² + 2·dd·K − 4R²a)·t² + (4nK − 8R²b)·t + (K² − 4R²c) = 0
"""
import numpy as np


def _quartic(o, d, C, R, r):
    o = np.asarray(o, np.float64) - np.asarray(C, np.float64)
    d = np.asarray(d, np.float64)
    dd = np.einsum("...i,...i", d, d)
    n = np.einsum("...i,...i", o, d)
    K = np.einsum("...i,...i", o, o) + R * R - r * r
    a = d[..., 0] ** 2 + d[..., 2] ** 2
    b = o[..., 0] * d[..., 0] + o[..., 2] * d[..., 2]
    c = o[..., 0] ** 2 + o[..., 2] ** 2
    return np.stack([dd * dd, 4 * dd * n, 4 * n * n + 2 * dd * K - 4 * R * R * a,
                     4 * n * K - 8 * R * R * b, K * K - 4 * R * R * c], -1)


def real_roots(o, d, C, R, r, imag_tol=1e-7):
    """All real roots t (ascending, NaN padded to 4) for rays o,d of shape (n,3)."""
    o = np.atleast_2d(np.asarray(o, np.float64))
    d = np.atleast_2d(np.asarray(d, np.float64))
    C = np.asarray(C, np.float64)
    # shift to the point of closest approach for conditioning (an exact reparametrisation)
    dd = np.einsum("ni,ni->n", d, d)
    tc = -np.einsum("ni,ni->n", o - C, d) / dd
    os_ = o + tc[:, None] * d
    k = _quartic(os_, d, C, R, r)
    k = k / k[:, :1]
    n = len(k)
    comp = np.zeros((n, 4, 4))
    comp[:, 0, :] = -k[:, 1:]
    comp[:, 1, 0] = comp[:, 2, 1] = comp[:, 3, 2] = 1.0
    ev = np.linalg.eigvals(comp)
    scale = np.maximum(1.0, np.abs(ev))
    is_real = np.abs(ev.imag) < imag_tol ** 0.5 * scale  # loose: double roots split by ~sqrt(eps)
    u = np.where(is_real, ev.real, np.nan)
    # Newton polish on the monic quartic; reject candidates that do not converge to a root
    for _ in range(6):
        f = (((u + k[:, 1:2]) * u + k[:, 2:3]) * u + k[:, 3:4]) * u + k[:, 4:5]
        df = ((4 * u + 3 * k[:, 1:2]) * u + 2 * k[:, 2:3]) * u + k[:, 3:4]
        with np.errstate(all="ignore"):
            step = np.where(df != 0, f / df, 0.0)
        u = u - step
    f = (((u + k[:, 1:2]) * u + k[:, 2:3]) * u + k[:, 3:4]) * u + k[:, 4:5]
    fscale = np.abs(k[:, 4:5]) + np.abs(k[:, 2:3]) * u * u + u ** 4 + 1e-300
    u = np.where(np.abs(f) < 1e-9 * fscale, u, np.nan)
    t = np.sort(u + tc[:, None], axis=1)  # NaN sorts last
    return t


def first_hit(o, d, tori, tmin=0.001, tmax=10000.0):
    """Closest hit over tori [(C,R,r), …].  Returns (t (NaN = miss), id (-1 = miss))."""
    o = np.atleast_2d(np.asarray(o, np.float64))
    best = np.full(len(o), np.inf)
    bid = np.full(len(o), -1)
    for i, (C, R, r) in enumerate(tori):
        t = real_roots(o, d, C, float(R), float(r))
        t = np.where((t > tmin) & (t < tmax), t, np.inf)
        ti = np.min(t, axis=1)
        upd = ti < best
        best = np.where(upd, ti, best)
        bid = np.where(upd, i, bid)
    return np.where(np.isfinite(best), best, np.nan), bid


def classify_margin(o, d, tori, tmin=0.001, tmax=10000.0, delta=1e-4, t_tol=1e-3):
    """True where the first hit is robust: same hit/miss, same torus and |Δt| < t_tol when
    every tube radius r is scaled by (1 ± delta) and the interval bounds by (1 ± delta)."""
    t0, id0 = first_hit(o, d, tori, tmin, tmax)
    ok = np.ones(len(t0), bool)
    for s in (1 - delta, 1 + delta):
        ts, ids = first_hit(o, d, [(C, R, r * s) for C, R, r in tori], tmin * s, tmax / s)
        same = (np.isnan(ts) == np.isnan(t0)) & (ids == id0)
        with np.errstate(invalid="ignore"):
            close = np.isnan(t0) | (np.abs(ts - t0) < t_tol)
        ok &= same & close
    return ok


def normal(P, C, R):
    """Outward unit normal at points P (n,3) of the torus (C,R,·), float64."""
    p = np.asarray(P, np.float64) - np.asarray(C, np.float64)
    rho = np.hypot(p[:, 0], p[:, 2])
    q = np.stack([p[:, 0] * R / rho, np.zeros_like(rho), p[:, 2] * R / rho], 1)
    v = p - q
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# ---------------------------------------------------------------------------------------------
# Closed-form families: rays whose intersection with the torus reduces to circles
# ---------------------------------------------------------------------------------------------
# Three families of rays meet a torus (centre C, axis +y, radii R, r) in sections that are circles, so the hit
# parameter is a square root away — no quartic, no iteration, nothing shared with any solver in this repository:
#   "equatorial"  origin and direction in the plane y = Cy: the section is the annulus between the circles of radius
#                 R - r and R + r around C — every crossing of either circle is a surface crossing;
#   "meridional"  origin and direction in the plane z = Cz (a plane through the axis): the section is the two
#                 tube circles of radius r around (Cx ± R, Cy);
#   "axial"       direction ±y: the ray stays at distance rho from the axis and meets the tube circle
#                 (rho - R)² + y² = r².
# closed_form_family() draws rays of a family and returns, per ray, the smallest crossing parameter beyond tmin
# (inf = miss), the outward normal there, and a margin flag: False where a discriminant is so close to zero (or a
# crossing so close to tmin) that FP32 arithmetic may legitimately classify the ray the other way.
def _circle_roots(ox, oy, dx, dy, cx, cy, rad):
    """parameters t of (o + t d) on the circle |p - c| = rad in a plane; d need not be unit; nan where none"""
    ex, ey = ox - cx, oy - cy
    a = dx * dx + dy * dy
    b = ex * dx + ey * dy
    c = ex * ex + ey * ey - rad * rad
    disc = b * b - a * c
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(disc)
    return np.stack([(-b - sq) / a, (-b + sq) / a], -1), disc / (a * rad * rad)   # disc relative to (a rad²)


def closed_form_family(family, n, seed, C=(0.0, 0.0, 0.0), R=1.0, r=0.25, tmin=0.001, box=3.0):
    rng = np.random.default_rng(seed)
    C = np.asarray(C, np.float64)
    o = np.tile(C, (n, 1))
    d = np.zeros((n, 3))
    # the rays are drawn, rounded to FP32 (what every FP32 consumer sees) and the closed form is evaluated on the
    # ROUNDED rays: zero components stay exactly zero, so a ray stays exactly in its plane / parallel to the axis
    # (C must be exactly representable in FP32)
    def f32(a):
        return a.astype(np.float32).astype(np.float64)
    if family == "equatorial":
        o[:, [0, 2]] += rng.uniform(-box, box, (n, 2)) * (R + r)
        ang = rng.uniform(0, 2 * np.pi, n)
        d[:, 0], d[:, 2] = np.cos(ang), np.sin(ang)
        o, d = f32(o), f32(d)
        roots, margins = [], []
        for rad in (R + r, R - r):
            t, m = _circle_roots(o[:, 0], o[:, 2], d[:, 0], d[:, 2], C[0], C[2], rad)
            roots.append(t)
            margins.append(m)
    elif family == "meridional":
        o[:, [0, 1]] += rng.uniform(-box, box, (n, 2)) * (R + r)
        ang = rng.uniform(0, 2 * np.pi, n)
        d[:, 0], d[:, 1] = np.cos(ang), np.sin(ang)
        # 70 % of the rays are aimed at a point within 1.5 r of one of the two tube circles (thin tubes are small targets)
        aim = rng.uniform(size=n) < 0.7
        tgt = np.stack([C[0] + np.where(rng.uniform(size=n) < 0.5, R, -R), np.full(n, C[1])], 1) + rng.uniform(-1.5 * r, 1.5 * r, (n, 2))
        v = tgt - o[:, :2]
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        d[aim, 0], d[aim, 1] = v[aim, 0], v[aim, 1]
        o, d = f32(o), f32(d)
        roots, margins = [], []
        for sx in (+1.0, -1.0):
            t, m = _circle_roots(o[:, 0], o[:, 1], d[:, 0], d[:, 1], C[0] + sx * R, C[1], r)
            roots.append(t)
            margins.append(m)
    elif family == "axial":
        o[:, [0, 2]] += rng.uniform(-(R + 2 * r), R + 2 * r, (n, 2))
        o[:, 1] += rng.uniform(-box, box, n)
        d[:, 1] = np.where(rng.uniform(size=n) < 0.5, 1.0, -1.0)
        o, d = f32(o), f32(d)
        rho = np.hypot(o[:, 0] - C[0], o[:, 2] - C[2])
        disc = r * r - (rho - R) ** 2
        with np.errstate(invalid="ignore"):
            h = np.sqrt(disc)
        ys = np.stack([C[1] - h, C[1] + h], -1)
        roots = [(ys - o[:, 1:2]) / d[:, 1:2]]
        margins = [disc / (r * r)]
    else:
        raise ValueError(family)
    t_all = np.concatenate(roots, 1)
    with np.errstate(invalid="ignore"):
        t_all = np.where(t_all > tmin, t_all, np.inf)
    t_all = np.where(np.isnan(t_all), np.inf, t_all)
    t = t_all.min(1)
    # margin: every discriminant away from 0 (tangency), every crossing away from tmin, o not on the surface
    ok = np.ones(n, bool)
    for m in margins:
        ok &= np.abs(m) > 1e-3
    every = np.concatenate(roots, 1)
    with np.errstate(invalid="ignore"):
        ok &= ~np.any(np.abs(every - tmin) < 1e-3, axis=1)
    # outward normal N = (P - q)/r, q the nearest point of the centre circle
    P = o + np.where(np.isfinite(t), t, 0.0)[:, None] * d
    pl = P - C
    rho = np.hypot(pl[:, 0], pl[:, 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        q = np.stack([pl[:, 0] * R / rho, np.zeros(n), pl[:, 2] * R / rho], 1)
    N = (pl - q) / r
    return o.astype(np.float32), d.astype(np.float32), t, N, ok
