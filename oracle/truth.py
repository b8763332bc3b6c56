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


# ---------------------------------------------------------------------------------------------
# Independent FP64 restatement of the shading chain (raygen → bounce loop → closest-hit shader)
# ---------------------------------------------------------------------------------------------
# Written from the GLSL, not from oracle/trt_oracle.c: numpy float64, vectorised over the pixels that are still
# in the loop, first hits from the companion-matrix solver above, the shadow query as a first hit below the light
# distance.  It shares no arithmetic with the C oracle or the kernels (which are FP32 with explicit fma and a
# different root finder), so agreement on colours pins the control flow and the Phong chain, not only t and N.
# Paths: vk_raytracing_tutorial_KHR/ray_tracing_reflections/shaders (REFL) and ray_tracing__before/shaders (BEF).
def _normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _reflect(i, n):
    """GLSL reflect(I, N) = I - 2 dot(N, I) N"""
    return i - 2.0 * np.einsum("ni,ni->n", n, i)[:, None] * n


def primary_rays(view_inverse, proj_inverse, center, rho, W, H, camera):
    """(origin, direction) per pixel, row-major y*W+x.  view_inverse / proj_inverse: 4x4, math convention M[row, col].
    camera 0: REFL/raytrace.rgen:42-48.  camera 1: BEF/raytrace.rgen:22-57 (angles in degrees, as there)."""
    vi, pi_ = np.asarray(view_inverse, np.float64), np.asarray(proj_inverse, np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    xs, ys = xs.reshape(-1).astype(np.float64), ys.reshape(-1).astype(np.float64)
    n = W * H
    if camera == 0:
        dx, dy = (xs + 0.5) / W * 2.0 - 1.0, (ys + 0.5) / H * 2.0 - 1.0                       # rgen:42-44
        origin = np.tile((vi @ np.array([0.0, 0.0, 0.0, 1.0]))[:3], (n, 1))                   # rgen:46
        target = (pi_ @ np.stack([dx, dy, np.ones(n), np.ones(n)]))[:3].T                     # rgen:47
        direction = (vi[:3, :3] @ _normalize(target).T).T                                     # rgen:48 (w = 0)
        return origin, direction
    eye = (vi @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]                                           # BEF rgen:36
    c = np.asarray(center, np.float64)
    tmp = c - eye                                                                             # :38
    dir2 = np.array([tmp[0], tmp[2]]) / np.hypot(tmp[0], tmp[2])                              # :39
    omega = np.degrees(np.arccos(dir2[0]))                                                    # :40
    if tmp[2] < 0:
        omega = 360.0 - omega                                                                 # :41-43
    theta = 0.0
    if eye[1] != c[1]:                                                                        # :45
        first = np.array([eye[0] + rho * np.cos(np.radians(omega)), eye[1], eye[2] + rho * np.sin(np.radians(omega))])
        tmp = c - first                                                                       # :46-47
        theta = np.degrees(np.arccos(tmp[0] / np.hypot(tmp[0], tmp[1])))                      # :48-49
        if tmp[1] < 0:
            theta = 360.0 - theta                                                             # :50-52
    alfa, beta = 360.0 / W * xs, 360.0 / H * ys                                               # :25-28
    aw, bt = np.radians(alfa + omega), np.radians(beta + theta)
    origin = np.stack([eye[0] + rho * np.cos(aw), np.full(n, eye[1]), eye[2] + rho * np.sin(aw)], 1)   # :56
    direction = np.stack([np.cos(aw) * np.cos(bt), np.sin(bt), np.sin(aw) * np.cos(bt)], 1)           # :57
    return origin, direction


def shade_frame(tori, materials, view_inverse, proj_inverse, center, push, W, H, camera=0, tmin=0.001, tmax=10000.0):
    """tori: [(C, R, r, matId)]; materials: [dict(ambient, diffuse, specular, shininess, illum)];
    push: dict(clearColor(4), lightPosition(3), lightIntensity, lightType, maxDepth, rho).
    Returns (rgba (H, W, 4) float64, robust (H, W) bool, queries dict): `robust` is False for a pixel any of whose queries
    is sensitive to a 1e-4 perturbation (grazing hits, self-shadow terminators) or whose N·L is within 1e-3 of zero;
    queries = closest-hit queries at depth 0 / at depth > 0 / shadow queries (rays, not ray–torus tests)."""
    geo = [(np.asarray(C, np.float64), float(R), float(r)) for C, R, r, _ in tori]
    mat_of = np.array([m for _, _, _, m in tori])
    M = {k: np.array([np.asarray(m[k], np.float64) for m in materials]) for k in ("ambient", "diffuse", "specular")}
    shin = np.array([float(m["shininess"]) for m in materials])
    illum = np.array([int(m["illum"]) for m in materials])
    clear = np.asarray(push["clearColor"], np.float64)[:3]
    lp = np.asarray(push["lightPosition"], np.float64)
    n = W * H
    o, d = primary_rays(view_inverse, proj_inverse, center, float(push.get("rho", 0.0)), W, H, camera)
    hit_value = np.zeros((n, 3))
    attenuation = np.ones((n, 3))                                                             # rgen:56
    robust = np.ones(n, bool)
    live = np.arange(n)                                                                       # pixels still in the loop
    queries = {"primary": 0, "bounce": 0, "shadow": 0}
    for depth in range(int(push["maxDepth"])):                                                # rgen:62, :79
        if len(live) == 0:
            break
        queries["primary" if depth == 0 else "bounce"] += len(live)
        t, tid = first_hit(o, d, geo, tmin, tmax)                                             # rgen:64-75
        robust[live] &= classify_margin(o, d, geo, tmin, tmax)
        hit = ~np.isnan(t)
        prd = np.tile(clear * 0.8, (len(live), 1))                                            # rmiss:37
        done = np.ones(len(live), bool)                                                       # rgen:57 / :84
        nxt_o, nxt_d = o.copy(), d.copy()
        if hit.any():
            ho, hd, ht, hid = o[hit], d[hit], t[hit], tid[hit]
            P = ho + ht[:, None] * hd                                                         # BEF rchit:134
            N = np.zeros_like(P)
            for i, (C, R, r) in enumerate(geo):
                sel = hid == i
                if sel.any():
                    N[sel] = normal(P[sel], C, R)                                             # role of rchit:74-75
            if int(push["lightType"]) == 0:                                                   # rchit:82-88
                ldir = lp - P
                ldist = np.linalg.norm(ldir, axis=1)
                lint = float(push["lightIntensity"]) / (ldist * ldist)
                L = ldir / ldist[:, None]
            else:                                                                             # rchit:89-92
                L = np.tile(lp / np.linalg.norm(lp), (len(P), 1))
                ldist = np.full(len(P), 100000.0)
                lint = np.full(len(P), float(push["lightIntensity"]))
            mi = mat_of[hid]                                                                  # rchit:95-96
            ndl = np.einsum("ni,ni->n", N, L)
            diffuse = M["diffuse"][mi] * np.maximum(ndl, 0.0)[:, None]                        # wavefront.glsl:25-26
            diffuse = diffuse + np.where((illum[mi] >= 1)[:, None], M["ambient"][mi], 0.0)    # :27-28
            specular = np.zeros_like(P)
            att1 = np.ones(len(P))
            want = ndl > 0                                                                    # rchit:112
            hit_px = live[hit]
            robust[hit_px] &= np.abs(ndl) > 1e-3
            if want.any():
                queries["shadow"] += int(want.sum())
                so, sd = P[want], L[want]                                                     # rchit:116-117
                shadowed = np.zeros(len(so), bool)
                stable = np.ones(len(so), bool)
                # one query per light distance would be exact; the distances differ per pixel, so: per-pixel tmax
                ts, _ = first_hit(so, sd, geo, tmin, np.inf)
                with np.errstate(invalid="ignore"):
                    shadowed = ~np.isnan(ts) & (ts < ldist[want])                             # rchit:114-131, any hit below tMax
                    # sensitivity: a first hit that moves across tmin or across the light distance, or appears / vanishes
                    stable = classify_margin(so, sd, geo, tmin, np.inf) & ~(np.abs(ts - ldist[want]) < 1e-3)
                robust[hit_px[want]] &= stable
                att1[want] = np.where(shadowed, 0.3, 1.0)                                     # rchit:133-136
                lit = np.flatnonzero(want)[~shadowed]
                if len(lit):                                                                  # rchit:140 → wavefront.glsl:32-48
                    ml = mi[lit]
                    ks = np.maximum(shin[ml], 4.0)
                    energy = (2.0 + ks) / (2.0 * 3.14159265)
                    V = _normalize(-hd[lit])
                    Rr = _reflect(-L[lit], N[lit])
                    s = energy * np.power(np.maximum(np.einsum("ni,ni->n", V, Rr), 0.0), ks)
                    specular[lit] = np.where((illum[ml] >= 2)[:, None], M["specular"][ml] * s[:, None], 0.0)
            mirror = illum[mi] == 3                                                           # rchit:145
            a_hit = attenuation[hit_px]
            a_hit[mirror] *= M["specular"][mi][mirror]                                        # rchit:149
            attenuation[hit_px] = a_hit
            hdone = np.ones(len(P), bool)
            hdone[mirror] = False                                                             # rchit:150
            done[hit] = hdone
            no, nd = ho.copy(), hd.copy()
            no[mirror] = P[mirror]                                                            # rchit:147,151
            nd[mirror] = _reflect(hd[mirror], N[mirror])                                      # rchit:148,152 (not re-normalised)
            nxt_o[hit], nxt_d[hit] = no, nd
            prd[hit] = (att1 * lint)[:, None] * (diffuse + specular)                          # rchit:155
        hit_value[live] += prd * attenuation[live]                                            # rgen:76
        keep = ~done                                                                          # rgen:79
        live, o, d = live[keep], nxt_o[keep], nxt_d[keep]                                     # rgen:82-84
    rgba = np.concatenate([hit_value, np.ones((n, 1))], 1).reshape(H, W, 4)                   # rgen:87
    return rgba, robust.reshape(H, W), queries
