/*
 * trt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the ray-tracing path of raffaelecicellini/toroidal_ray_tracing
 * (paths relative to vk_raytracing_tutorial_KHR/; REFL = ray_tracing_reflections,
 * BEF = ray_tracing__before).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; libtrt.so never does.
 *
 * PARITY STATUS: "parity unpinned".  The reference cannot be built or run here (Vulkan RT
 * + Windows + absent nvpro_core), it has no CPU path, no tests and no
 * golden vectors, and it contains no ray–torus arithmetic at all.  What this file pins is
 *   - the control flow / payload / shading semantics of the reference's GLSL, restated
 *     line by line below with file:line citations, and
 *   - build-defined torus arithmetic (trt_solve.inc), checked against analytic
 *     known-answer vectors and an independent FP64 solver (oracle/truth.py).
 *
 * Arithmetic contract shared with the HIP kernels (so that they can agree bit for bit):
 *   FP32 throughout; dot3(a,b) = fma(az,bz, fma(ay,by, ax*bx)); normalize(v) =
 *   v * (1/sqrt(dot3(v,v))); mat4·vec4 accumulates columns 0,1,2,3 with fma; no implicit
 *   contraction (-ffp-contract=off).  Only pow() (specular lobe) and the toroidal camera's
 *   per-frame/per-column/per-row cos/sin/acos come from libm.
 */
#include "../include/trt.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* type-generic torus solver, instantiated for float and double                          */
/* ------------------------------------------------------------------------------------ */
#define REAL float
#define SUF(x) x##_f32
#define FMA fmaf
#define SQRT sqrtf
#define FMAX fmaxf
#define FMIN fminf
#define FABS fabsf
#define TRT_DK_TOL 0.0009765625f /* 2^-10 */
#define TRT_USTOP 0.000244140625f /* 2^-12 */
#include "trt_solve.inc"
#undef TRT_DK_TOL
#undef TRT_USTOP
#undef REAL
#undef SUF
#undef FMA
#undef SQRT
#undef FMAX
#undef FMIN
#undef FABS

#define REAL double
#define SUF(x) x##_f64
#define FMA fma
#define SQRT sqrt
#define FMAX fmax
#define FMIN fmin
#define FABS fabs
#define TRT_DK_TOL 2.384185791015625e-07 /* 2^-22 */
#define TRT_USTOP 5.9604644775390625e-08 /* 2^-24 */
#include "trt_solve.inc"
#undef TRT_DK_TOL
#undef TRT_USTOP
#undef REAL
#undef SUF
#undef FMA
#undef SQRT
#undef FMAX
#undef FMIN
#undef FABS

/* ------------------------------------------------------------------------------------ */
/* small FP32 vector helpers (GLSL built-ins restated)                                   */
/* ------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 scale3(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline v3 neg3(v3 a) { v3 r = {-a.x, -a.y, -a.z}; return r; }
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
/* GLSL reflect(I,N) = I - 2·dot(N,I)·N */
static inline v3 reflect3(v3 i, v3 n)
{
  const float k = 2.0f * dot3(n, i);
  v3 r = {fmaf(-k, n.x, i.x), fmaf(-k, n.y, i.y), fmaf(-k, n.z, i.z)};
  return r;
}
/* column-major mat4 · (x,y,z,w), rows 0..2 */
static inline v3 mat4_mul(const float* m, float x, float y, float z, float w)
{
  v3 r;
  r.x = fmaf(m[12], w, fmaf(m[8], z, fmaf(m[4], y, m[0] * x)));
  r.y = fmaf(m[13], w, fmaf(m[9], z, fmaf(m[5], y, m[1] * x)));
  r.z = fmaf(m[14], w, fmaf(m[10], z, fmaf(m[6], y, m[2] * x)));
  return r;
}

/* ------------------------------------------------------------------------------------ */
/* scene                                                                                 */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  int          n;
  trt_torus    tori[TRT_MAX_TORI];
  torus_k_f32  k32[TRT_MAX_TORI];
  torus_k_f64  k64[TRT_MAX_TORI];
  int          nmat;
  trt_material mat[TRT_MAX_MATERIALS];
  int          f64;
  int          dk;   /* 0: Fourier–Newton walk; alternative root solvers: 1 Durand–Kerner, 2 Ferrari */
  int          order[TRT_MAX_TORI]; /* test order: descending bounding radius R + r, ties by index */
  uint32_t     inside[TRT_MAX_TORI]; /* enclosure masks (T3, below): bit k = the torus at test-order position k */
} scene_t;

/* Enclosure cull (build-defined, like the rest of T3).  A ray whose origin lies OUTSIDE the tube of torus j
 * must cross j's surface before it can reach a tube that lies strictly inside j's: the closest hit is never the inner
 * torus, and an inner torus shadows nothing that j does not.  The queries therefore skip the tori named by a mask:
 *   - inside[j]: the tori whose tube lies strictly inside tube j (same axis line; centre circles everywhere
 *     sqrt(dR² + dy²) apart; that distance + r_k < r_j with 2^-10 of r_j to spare);
 *   - primary rays: the union of inside[j] over the tori j whose tube every ray origin of the frame lies outside of
 *     (the eye, or the circle of radius rho around it for the toroidal camera), with the same margin;
 *   - a hit on torus h that the path leaves OUTWARDS adds inside[h]: always for the shadow ray (it is cast only when
 *     N·L > 0), for the reflected ray when the incoming ray met the surface from outside (N·D < 0); the mask of a path
 *     only grows (a segment that ends on a surface has crossed none, so what the origin was outside of, the hit point is).
 * A skipped torus still counts as a test of the query.  g_enclosure_cull = 0 (oracle_set_enclosure_cull, tests only)
 * tests every torus in every query: the images, records and query counts must come out the same — and do, on every
 * fixture and fuzz scene (tests/test_oracle.py::test_enclosure_cull_changes_nothing). */
static int g_enclosure_cull = 1;
void oracle_set_enclosure_cull(int on) { g_enclosure_cull = on; }

static int scene_prepare(const trt_scene* s, int precision, scene_t* out)
{
  if(!s || !s->tori || !s->materials || s->n_tori < 1 || s->n_tori > TRT_MAX_TORI
     || s->n_materials < 1 || s->n_materials > TRT_MAX_MATERIALS)
    return TRT_E_SCENE;
  out->n    = (int)s->n_tori;
  out->nmat = (int)s->n_materials;
  out->f64  = precision == TRT_SOLVE_F64 || precision == TRT_SOLVE_DK_F64 || precision == TRT_SOLVE_FERRARI_F64;
  out->dk   = (precision == TRT_SOLVE_DK_F32 || precision == TRT_SOLVE_DK_F64) ? 1
              : (precision == TRT_SOLVE_FERRARI_F32 || precision == TRT_SOLVE_FERRARI_F64) ? 2 : 0;
  for(int i = 0; i < out->n; ++i)
  {
    const trt_torus* t = &s->tori[i];
    if(!(t->r > 0.0f && t->R > t->r) || t->matId < 0 || t->matId >= out->nmat)
      return TRT_E_SCENE;
    out->tori[i] = *t;
    torus_prepare_f32(t, &out->k32[i]);
    torus_prepare_f64(t, &out->k64[i]);
  }
  memcpy(out->mat, s->materials, sizeof(trt_material) * (size_t)out->nmat);
  /* Test order (build-defined, like the torus arithmetic): largest bounding sphere first, so
   * that an enclosing shell is hit before the shells inside it and the later tests are cut off
   * by the shrinking interval (closest_hit).  Stable insertion sort on the FP32 sum R + r. */
  for(int i = 0; i < out->n; ++i)
  {
    const float key = s->tori[i].R + s->tori[i].r;
    int k = i;
    while(k > 0 && s->tori[out->order[k - 1]].R + s->tori[out->order[k - 1]].r < key)
    {
      out->order[k] = out->order[k - 1];
      --k;
    }
    out->order[k] = i;
  }
  /* enclosure masks: plain double arithmetic on the scene's floats, as trt_api.hip build_scene_uncached */
  for(int j = 0; j < out->n; ++j)
  {
    const trt_torus* J = &s->tori[j];
    out->inside[j] = 0u;
    for(int p = 0; p < out->n && g_enclosure_cull; ++p)
    {
      const int        k = out->order[p];
      const trt_torus* K = &s->tori[k];
      if(k == j || K->center[0] != J->center[0] || K->center[2] != J->center[2]) continue;
      const double dR = (double)K->R - (double)J->R, dy = (double)K->center[1] - (double)J->center[1];
      const double D  = sqrt(dR * dR + dy * dy);
      if(D + (double)K->r < (double)J->r - (double)J->r * 0.0009765625) out->inside[j] |= 1u << p;
    }
  }
  return TRT_OK;
}

/* the camera's share of the enclosure cull: trt_api.hip primary_skip_mask */
static uint32_t primary_skip_mask(const scene_t* S, v3 eye, float reach)
{
  uint32_t mask = 0u;
  for(int j = 0; j < S->n; ++j)
  {
    if(S->inside[j] == 0u) continue;
    const trt_torus* T = &S->tori[j];
    const double ex = (double)eye.x - (double)T->center[0], ey = (double)eye.y - (double)T->center[1], ez = (double)eye.z - (double)T->center[2];
    const double rho = sqrt(ex * ex + ez * ez) - (double)T->R;
    const double d   = sqrt(rho * rho + ey * ey);
    const double rc  = fabs((double)reach);
    if(d > ((double)T->r + (double)T->r * 0.0009765625) + (rc + rc * 0.0009765625)) mask |= S->inside[j];
  }
  return mask;
}

/* Work counters of the calling thread (trt_stats.solved_tests / .evaluations): tests that passed
 * the bounding-volume culls of T1, and the polynomial evaluations of this file's formulation of
 * the walk (the GPU's state machine visits the same points in a different bookkeeping; only
 * solved_tests is comparable across the two). */
static _Thread_local uint64_t tl_solved, tl_evals, tl_traced;

/* One ray against one torus, in the scene's solver precision; result rounded to FP32. */
static inline int torus_hit(const scene_t* S, int i, v3 o, v3 d, float dd, float inv_dd,
                            float tmin, float tmax, float* t)
{
  ++tl_traced;
  int ne = 0, hit;
  if(S->f64)
  {
    const double o64[3] = {o.x, o.y, o.z}, d64[3] = {d.x, d.y, d.z};
    const double dd64   = fma(d64[2], d64[2], fma(d64[1], d64[1], d64[0] * d64[0]));
    double       t64;
    hit = torus_first_hit_f64(o64, d64, dd64, 1.0 / dd64, (double)tmin, (double)tmax, &S->k64[i],
                              S->dk, &t64, &ne);
    if(ne) { ++tl_solved; tl_evals += (uint64_t)ne; }
    if(!hit)
      return 0;
    const float tf = (float)t64;
    if(!(tf > tmin && tf < tmax)) /* rounding to FP32 may land on the open bounds */
      return 0;
    *t = tf;
    return 1;
  }
  const float o32[3] = {o.x, o.y, o.z}, d32[3] = {d.x, d.y, d.z};
  hit = torus_first_hit_f32(o32, d32, dd, inv_dd, tmin, tmax, &S->k32[i], S->dk, t, &ne);
  if(ne) { ++tl_solved; tl_evals += (uint64_t)ne; }
  return hit;
}

/* Closest hit over all tori (role of traceRayEXT + BVH, REFL/shaders/raytrace.rgen:64-75):
 * smallest t, first torus wins ties.  Returns torus index or -1. */
static int closest_hit(const scene_t* S, v3 o, v3 d, float tmin, float tmax, float* t_out,
                       uint64_t* tests, uint32_t skip)
{
  const float dd = dot3(d, d), inv_dd = 1.0f / dd;
  int   id   = -1;
  float best = INFINITY;
  for(int k = 0; k < S->n; ++k)
  {
    /* the interval of every later test ends at the closest hit so far (open interval: a hit
     * returned by torus_hit is already closer; equal t keeps the torus tested first) */
    const int   i  = S->order[k];
    const float tm = fminf(tmax, best);
    float t;
    ++*tests;
    if((skip >> k) & 1u) continue;   /* enclosure cull: counted, not traced */
    if(torus_hit(S, i, o, d, dd, inv_dd, tmin, tm, &t))
    {
      best = t;
      id   = i;
    }
  }
  *t_out = best;
  return id;
}

/* Any hit (shadow query: gl_RayFlagsTerminateOnFirstHitEXT, REFL/shaders/raytrace.rchit:114-131). */
static int any_hit(const scene_t* S, v3 o, v3 d, float tmin, float tmax, uint64_t* tests, uint32_t skip)
{
  const float dd = dot3(d, d), inv_dd = 1.0f / dd;
  for(int k = 0; k < S->n; ++k)
  {
    float t;
    ++*tests;
    if((skip >> k) & 1u) continue;
    if(torus_hit(S, S->order[k], o, d, dd, inv_dd, tmin, tmax, &t))
      return 1;
  }
  return 0;
}

/* T4: outward unit normal at P on torus i — N = normalize(P - q), q the nearest point of
 * the centre circle (role of REFL/shaders/raytrace.rchit:74-75; never flipped). */
static v3 torus_normal(const scene_t* S, int i, v3 P)
{
  const trt_torus* T = &S->tori[i];
  const v3    c   = {T->center[0], T->center[1], T->center[2]};
  const v3    pl  = sub3(P, c);
  const float rho = sqrtf(fmaf(pl.z, pl.z, pl.x * pl.x));
  const float k   = (rho - T->R) / rho;
  const v3    v   = {pl.x * k, pl.y, pl.z * k};
  return normalize3(v);
}

/* ------------------------------------------------------------------------------------ */
/* Phong helpers — REFL/shaders/wavefront.glsl:22-48                                      */
/* ------------------------------------------------------------------------------------ */
static v3 compute_diffuse(const trt_material* m, v3 L, v3 N)
{
  const float dotNL = fmaxf(dot3(N, L), 0.0f);                                 /* :25 */
  v3 c = {m->diffuse[0] * dotNL, m->diffuse[1] * dotNL, m->diffuse[2] * dotNL}; /* :26 */
  if(m->illum >= 1)                                                            /* :27 */
  {
    c.x += m->ambient[0];
    c.y += m->ambient[1];
    c.z += m->ambient[2];
  }
  return c;
}

static v3 compute_specular(const trt_material* m, v3 viewDir, v3 L, v3 N)
{
  v3 z = {0.0f, 0.0f, 0.0f};
  if(m->illum < 2)                                                             /* :34 */
    return z;
  const float kPi        = 3.14159265f;                                        /* :38 */
  const float kShininess = fmaxf(m->shininess, 4.0f);                          /* :39 */
  const float kEnergy    = (2.0f + kShininess) / (2.0f * kPi);                 /* :42 */
  const v3    V          = normalize3(neg3(viewDir));                          /* :43 */
  const v3    R          = reflect3(neg3(L), N);                               /* :44 */
  /* :45 — GLSL defines pow(x,y) = exp2(y*log2(x)) (GLSL 4.60 §8.2); restated that way */
  const float s          = kEnergy * exp2f(kShininess * log2f(fmaxf(dot3(V, R), 0.0f)));
  v3 r = {m->specular[0] * s, m->specular[1] * s, m->specular[2] * s};         /* :47 */
  return r;
}

/* ------------------------------------------------------------------------------------ */
/* ray generation                                                                        */
/* ------------------------------------------------------------------------------------ */
#define TRT_DEG2RAD 0.017453292519943295f /* GLSL radians() */
#define TRT_RAD2DEG 57.29577951308232f    /* GLSL degrees() */

/* Per-frame part of the toroidal camera: BEF/shaders/raytrace.rgen:36-53. */
typedef struct { v3 eye; float omega, theta; } toro_frame;

static toro_frame toroidal_frame(const trt_globals* g, const trt_push* pc)
{
  toro_frame F;
  F.eye         = mat4_mul(g->viewInverse, 0.0f, 0.0f, 0.0f, 1.0f);           /* :36 */
  const v3 ctr  = {g->center[0], g->center[1], g->center[2]};
  v3       temp = sub3(ctr, F.eye);                                            /* :38 */
  float    il   = 1.0f / sqrtf(fmaf(temp.z, temp.z, temp.x * temp.x));         /* :39 */
  float    dirx = temp.x * il;
  F.omega       = acosf(dirx) * TRT_RAD2DEG;                                   /* :40 dot((1,0),dir) = dir.x */
  if(temp.z < 0.0f)                                                            /* :41 */
    F.omega = 360.0f - F.omega;
  F.theta = 0.0f;
  if(F.eye.y != ctr.y)                                                         /* :45 */
  {
    const float w = F.omega * TRT_DEG2RAD;
    v3 p = {fmaf(pc->rho, cosf(w), F.eye.x), F.eye.y, fmaf(pc->rho, sinf(w), F.eye.z)}; /* :46 */
    temp = sub3(ctr, p);                                                       /* :47 */
    il   = 1.0f / sqrtf(fmaf(temp.y, temp.y, temp.x * temp.x));                /* :48 */
    dirx = temp.x * il;
    F.theta = acosf(dirx) * TRT_RAD2DEG;                                       /* :49 */
    if(temp.y < 0.0f)                                                          /* :50 */
      F.theta = 360.0f - F.theta;
  }
  return F;
}

static void raygen(const trt_globals* g, const trt_push* pc, const toro_frame* F, uint32_t W,
                   uint32_t H, int camera, uint32_t x, uint32_t y, v3* origin, v3* dir)
{
  if(camera == TRT_CAMERA_TOROIDAL)
  {
    const float d_alfa = 360.0f / (float)W;                                    /* BEF rgen:25 */
    const float d_beta = 360.0f / (float)H;                                    /* :26 */
    const float alfa   = d_alfa * (float)x;                                    /* :27 */
    const float beta   = d_beta * (float)y;                                    /* :28 */
    const float aw     = (alfa + F->omega) * TRT_DEG2RAD;
    const float bt     = (beta + F->theta) * TRT_DEG2RAD;
    const float ca = cosf(aw), sa = sinf(aw), cb = cosf(bt), sb = sinf(bt);
    origin->x = fmaf(pc->rho, ca, F->eye.x);                                   /* :56 */
    origin->y = F->eye.y;
    origin->z = fmaf(pc->rho, sa, F->eye.z);
    dir->x = ca * cb;                                                          /* :57 */
    dir->y = sb;
    dir->z = sa * cb;
    return;
  }
  /* pinhole: REFL/shaders/raytrace.rgen:42-48 */
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;                      /* :42 */
  const float u = px / (float)W, v = py / (float)H;                            /* :43 */
  const float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;                      /* :44 */
  *origin        = mat4_mul(g->viewInverse, 0.0f, 0.0f, 0.0f, 1.0f);          /* :46 */
  const v3 tgt   = mat4_mul(g->projInverse, dx, dy, 1.0f, 1.0f);              /* :47 */
  const v3 tn    = normalize3(tgt);
  *dir           = mat4_mul(g->viewInverse, tn.x, tn.y, tn.z, 0.0f);          /* :48 */
}

/* ------------------------------------------------------------------------------------ */
/* one pixel: raygen bounce loop + closest-hit shader + miss shaders                     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  v3    color;               /* accumulated hitValue (rgen:61,76)                        */
  float t0; v3 P0, N0; int id0; /* depth-0 hit record; miss: t=inf, P=N=0, id=-1           */
  v3    rayO, rayD;          /* primary ray (BEF rgen:72-73)                             */
} pixel_out;

static void shade_pixel(const scene_t* S, const trt_globals* g, const trt_push* pc,
                        const toro_frame* F, uint32_t W, uint32_t H, int camera, uint32_t x,
                        uint32_t y, uint32_t skip_primary, pixel_out* out, trt_stats* st)
{
  v3 origin, direction;
  raygen(g, pc, F, W, H, camera, x, y, &origin, &direction);
  out->rayO = origin;
  out->rayD = direction;

  const float tMin = 0.001f, tMax = 10000.0f;                                  /* rgen:51-52 */
  int depth = 0, done = 1;                                                     /* rgen:54,57 */
  v3  attenuation = {1.0f, 1.0f, 1.0f};                                        /* rgen:56 */
  v3  hitValue    = {0.0f, 0.0f, 0.0f};                                        /* rgen:61 */
  const v3 zero   = {0.0f, 0.0f, 0.0f};
  const v3 lp     = {pc->lightPosition[0], pc->lightPosition[1], pc->lightPosition[2]};
  out->t0 = INFINITY; out->P0 = zero; out->N0 = zero; out->id0 = -1;
  uint32_t skip = skip_primary;                                                /* enclosure cull (above) */

  for(;;)                                                                      /* rgen:62 */
  {
    v3       prdHit;
    v3       nextO = origin, nextD = direction;
    float    t;
    uint64_t* ctr = depth == 0 ? &st->primary_tests : &st->bounce_tests;
    const int id  = closest_hit(S, origin, direction, tMin, tMax, &t, ctr, skip); /* rgen:64-75 */
    if(id < 0)
    {
      /* miss shader: REFL/shaders/raytrace.rmiss:37 (BEF rmiss:19-21: hitPosition = 0) */
      prdHit.x = pc->clearColor[0] * 0.8f;
      prdHit.y = pc->clearColor[1] * 0.8f;
      prdHit.z = pc->clearColor[2] * 0.8f;
    }
    else
    {
      /* closest-hit shader: REFL/shaders/raytrace.rchit:50-156 */
      const trt_material* mat = &S->mat[S->tori[id].matId];                    /* rchit:95-96 */
      /* hit point O + D·t (BEF rchit:134; shadow origin rchit:116) */
      const v3 P = {fmaf(t, direction.x, origin.x), fmaf(t, direction.y, origin.y),
                    fmaf(t, direction.z, origin.z)};
      const v3 N = torus_normal(S, id, P);
      if(depth == 0) { out->t0 = t; out->P0 = P; out->N0 = N; out->id0 = id; } /* BEF rgen:94-97 */

      v3    L;
      float lightIntensity = pc->lightIntensity;                               /* rchit:79 */
      float lightDistance  = 100000.0f;                                        /* rchit:80 */
      if(pc->lightType == 0)                                                   /* rchit:82 */
      {
        const v3    lDir = sub3(lp, P);                                        /* rchit:84 */
        const float l2   = dot3(lDir, lDir);
        lightDistance    = sqrtf(l2);                                          /* rchit:85 */
        lightIntensity   = pc->lightIntensity / (lightDistance * lightDistance); /* rchit:86 */
        L                = scale3(lDir, 1.0f / lightDistance);                 /* rchit:87 */
      }
      else
        L = normalize3(lp);                                                    /* rchit:91 */

      const v3 diffuse     = compute_diffuse(mat, L, N);                       /* rchit:100 */
      v3       specular    = zero;                                             /* rchit:108 */
      float    attenuation1 = 1.0f;                                            /* rchit:109 */
      if(dot3(N, L) > 0.0f)                                                    /* rchit:112 */
      {
        /* shadow ray: origin O + D·t, tMin .001, tMax lightDistance (rchit:114-131) */
        if(any_hit(S, P, L, 0.001f, lightDistance, &st->shadow_tests, skip | S->inside[id]))
          attenuation1 = 0.3f;                                                 /* rchit:135 */
        else
          specular = compute_specular(mat, direction, L, N);                   /* rchit:140 */
      }
      if(dot3(N, direction) < 0.0f)   /* met from outside: whatever leaves this point by reflection leaves outwards */
        skip |= S->inside[id];
      if(mat->illum == 3)                                                      /* rchit:145 */
      {
        attenuation.x *= mat->specular[0];                                     /* rchit:149 */
        attenuation.y *= mat->specular[1];
        attenuation.z *= mat->specular[2];
        done  = 0;                                                             /* rchit:150 */
        nextO = P;                                                             /* rchit:151 */
        nextD = reflect3(direction, N);                                        /* rchit:148,152 */
      }
      const float k = attenuation1 * lightIntensity;                           /* rchit:155 */
      prdHit.x = k * (diffuse.x + specular.x);
      prdHit.y = k * (diffuse.y + specular.y);
      prdHit.z = k * (diffuse.z + specular.z);
    }
    hitValue.x = fmaf(prdHit.x, attenuation.x, hitValue.x);                    /* rgen:76 */
    hitValue.y = fmaf(prdHit.y, attenuation.y, hitValue.y);
    hitValue.z = fmaf(prdHit.z, attenuation.z, hitValue.z);

    depth++;                                                                   /* rgen:78 */
    if(done == 1 || depth >= pc->maxDepth)                                     /* rgen:79 */
      break;
    origin    = nextO;                                                         /* rgen:82 */
    direction = nextD;                                                         /* rgen:83 */
    done      = 1;                                                             /* rgen:84 */
  }
  out->color = hitValue;
}

/* ------------------------------------------------------------------------------------ */
/* exported C entry points (loaded with ctypes by oracle/oracle.py)                      */
/* ------------------------------------------------------------------------------------ */
int oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Same contract as trt_render_dev (include/trt.h) on host buffers. */
int oracle_render(const trt_globals* g, const trt_push* pc, const trt_scene* scene, uint32_t W,
                  uint32_t H, uint32_t row_begin, uint32_t row_end, int camera, int precision,
                  int nthreads, float* rgba, trt_hits* hits, trt_rendered_data* rendered,
                  trt_stats* stats)
{
  scene_t S;
  int     rc = scene_prepare(scene, precision, &S);
  if(rc) return rc;
  if(!g || !pc || !W || !H || row_end > H || row_begin > row_end) return TRT_E_INVALID;
  const toro_frame F = toroidal_frame(g, pc);
  /* the camera's share of the enclosure cull: ray origins are the eye, or lie |rho| from it (BEF rgen:56) */
  const uint32_t skip_primary = primary_skip_mask(&S, mat4_mul(g->viewInverse, 0.0f, 0.0f, 0.0f, 1.0f),
                                                  camera == TRT_CAMERA_TOROIDAL ? pc->rho : 0.0f);
  uint64_t np = 0, nb = 0, ns = 0, nsol = 0, nev = 0, ntr = 0;
  (void)nthreads;
  /* work items: blocks of 64 pixels of a row (fine enough to keep >100 threads busy on an 8-row band) */
  const int64_t bpr = ((int64_t)W + 63) / 64, nblk = bpr * (int64_t)(row_end - row_begin);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads > 0 ? nthreads : 1) \
    reduction(+ : np, nb, ns, nsol, nev, ntr)
#endif
  for(int64_t blk = 0; blk < nblk; ++blk)
  {
    const uint32_t y  = row_begin + (uint32_t)(blk / bpr);
    const uint32_t xb = (uint32_t)(blk % bpr) * 64, xe = xb + 64 < W ? xb + 64 : W;
    trt_stats st = {0, 0, 0, 0, 0, 0, 0, 0};
    tl_solved = tl_evals = tl_traced = 0;
    for(uint32_t x = xb; x < xe; ++x)
    {
      pixel_out o;
      shade_pixel(&S, g, pc, &F, W, H, camera, x, y, skip_primary, &o, &st);
      const size_t i = (size_t)y * W + x;
      if(rgba)
      {
        rgba[4 * i + 0] = o.color.x;                                           /* rgen:87 */
        rgba[4 * i + 1] = o.color.y;
        rgba[4 * i + 2] = o.color.z;
        rgba[4 * i + 3] = 1.0f;
      }
      if(hits)
      {
        if(hits->t) hits->t[i] = o.t0;
        if(hits->px) hits->px[i] = o.P0.x;
        if(hits->py) hits->py[i] = o.P0.y;
        if(hits->pz) hits->pz[i] = o.P0.z;
        if(hits->nx) hits->nx[i] = o.N0.x;
        if(hits->ny) hits->ny[i] = o.N0.y;
        if(hits->nz) hits->nz[i] = o.N0.z;
        if(hits->id) hits->id[i] = o.id0;
      }
      if(rendered)
      {
        trt_rendered_data* r = &rendered[(size_t)x * H + y];                   /* BEF rgen:72 */
        r->pos[0] = o.P0.x; r->pos[1] = o.P0.y; r->pos[2] = o.P0.z; r->pos[3] = 1.0f;   /* :112 */
        r->color[0] = o.color.x; r->color[1] = o.color.y; r->color[2] = o.color.z;
        r->color[3] = 1.0f;                                                    /* :111 */
        r->rayOrigin[0] = o.rayO.x; r->rayOrigin[1] = o.rayO.y; r->rayOrigin[2] = o.rayO.z;
        r->rayOrigin[3] = 1.0f;                                                /* :56,72 */
        r->rayDir[0] = o.rayD.x; r->rayDir[1] = o.rayD.y; r->rayDir[2] = o.rayD.z;
        r->rayDir[3] = 0.0f;                                                   /* :57,73 */
      }
    }
    np += st.primary_tests;
    nb += st.bounce_tests;
    ns += st.shadow_tests;
    nsol += tl_solved;
    nev += tl_evals;
    ntr += tl_traced;
  }
  if(stats)
  {
    stats->primary_tests = np;
    stats->bounce_tests  = nb;
    stats->shadow_tests  = ns;
    stats->pixels        = (uint64_t)(row_end - row_begin) * W;
    stats->traced_tests  = ntr;            /* the oracle traces every pixel: every test the enclosure cull leaves */
    stats->solved_tests  = nsol;
    stats->evaluations   = nev;
    stats->reserved      = 0;
  }
  return TRT_OK;
}

/* Same contract as trt_trace (include/trt.h). */
int oracle_trace(const trt_rays* in, const trt_scene* scene, float tmin, float tmax,
                 int precision, int nthreads, trt_hits* out, trt_stats* stats)
{
  scene_t S;
  int     rc = scene_prepare(scene, precision, &S);
  if(rc) return rc;
  if(!in || !out) return TRT_E_INVALID;
  uint64_t np = 0, nsol = 0, nev = 0;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : np, nsol, nev)
#endif
  for(int64_t i = 0; i < (int64_t)in->n; ++i)
  {
    const v3 o = {in->ox[i], in->oy[i], in->oz[i]}, d = {in->dx[i], in->dy[i], in->dz[i]};
    float    t;
    uint64_t tests = 0;
    tl_solved = tl_evals = 0;
    const int id = closest_hit(&S, o, d, tmin, tmax, &t, &tests, 0u);   /* arbitrary rays: nothing is known about their origins */
    np += tests;
    nsol += tl_solved;
    nev += tl_evals;
    v3 P = {0.0f, 0.0f, 0.0f}, N = {0.0f, 0.0f, 0.0f};
    if(id >= 0)
    {
      P.x = fmaf(t, d.x, o.x); P.y = fmaf(t, d.y, o.y); P.z = fmaf(t, d.z, o.z);
      N   = torus_normal(&S, id, P);
    }
    if(out->t) out->t[i] = t;
    if(out->px) out->px[i] = P.x;
    if(out->py) out->py[i] = P.y;
    if(out->pz) out->pz[i] = P.z;
    if(out->nx) out->nx[i] = N.x;
    if(out->ny) out->ny[i] = N.y;
    if(out->nz) out->nz[i] = N.z;
    if(out->id) out->id[i] = id;
  }
  if(stats)
  {
    stats->primary_tests = np;
    stats->bounce_tests = stats->shadow_tests = 0;
    stats->pixels = in->n;
    stats->traced_tests = np;
    stats->solved_tests = nsol;
    stats->evaluations  = nev;
    stats->reserved     = 0;
  }
  return TRT_OK;
}

/* Primary ray of pixel (x,y): o[3], d[3]. */
int oracle_raygen(const trt_globals* g, const trt_push* pc, uint32_t W, uint32_t H, int camera,
                  uint32_t x, uint32_t y, float* o, float* d)
{
  if(!g || !pc || !o || !d) return TRT_E_INVALID;
  const toro_frame F = toroidal_frame(g, pc);
  v3 oo, dd;
  raygen(g, pc, &F, W, H, camera, x, y, &oo, &dd);
  o[0] = oo.x; o[1] = oo.y; o[2] = oo.z;
  d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
  return TRT_OK;
}

/* omega, theta (degrees) and eye of the toroidal camera frame. */
int oracle_toroidal_frame(const trt_globals* g, const trt_push* pc, float* omega_theta_eye)
{
  if(!g || !pc || !omega_theta_eye) return TRT_E_INVALID;
  const toro_frame F = toroidal_frame(g, pc);
  omega_theta_eye[0] = F.omega; omega_theta_eye[1] = F.theta;
  omega_theta_eye[2] = F.eye.x; omega_theta_eye[3] = F.eye.y; omega_theta_eye[4] = F.eye.z;
  return TRT_OK;
}

/* GLSL reflect(), exposed for the known-answer test k7. */
void oracle_reflect(const float* i, const float* n, float* r)
{
  const v3 I = {i[0], i[1], i[2]}, N = {n[0], n[1], n[2]};
  const v3 R = reflect3(I, N);
  r[0] = R.x; r[1] = R.y; r[2] = R.z;
}

/* Single ray vs single torus with evaluation count, both precisions (diagnostics). */
int oracle_torus_first_hit(const trt_torus* T, const float* o, const float* d, float tmin,
                           float tmax, int precision, double* t_out, int* evals)
{
  int ne = 0, hit;
  const int dk = (precision == TRT_SOLVE_DK_F32 || precision == TRT_SOLVE_DK_F64) ? 1
                 : (precision == TRT_SOLVE_FERRARI_F32 || precision == TRT_SOLVE_FERRARI_F64) ? 2 : 0;
  if(precision == TRT_SOLVE_F64 || precision == TRT_SOLVE_DK_F64 || precision == TRT_SOLVE_FERRARI_F64)
  {
    torus_k_f64 k;
    torus_prepare_f64(T, &k);
    const double o64[3] = {o[0], o[1], o[2]}, d64[3] = {d[0], d[1], d[2]};
    const double dd = fma(d64[2], d64[2], fma(d64[1], d64[1], d64[0] * d64[0]));
    double t;
    hit = torus_first_hit_f64(o64, d64, dd, 1.0 / dd, tmin, tmax, &k, dk, &t, &ne);
    if(hit) *t_out = t;
  }
  else
  {
    torus_k_f32 k;
    torus_prepare_f32(T, &k);
    const float dd = fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0]));
    float t;
    hit = torus_first_hit_f32(o, d, dd, 1.0f / dd, tmin, tmax, &k, dk, &t, &ne);
    if(hit) *t_out = (double)t;
  }
  if(evals) *evals = ne;
  return hit;
}

/* ------------------------------------------------------------------------------------ */
/* post pass: pow(c, 1/2.2) of REFL/shaders/post.frag:33-37, GLSL pow = exp2(y*log2(x)),   */
/* with the fixed log2/exp2 polynomials of DESIGN.md §4 (same operations as the kernel)    */
/* ------------------------------------------------------------------------------------ */
static inline float log2_poly(float x)
{
  int32_t bits;
  memcpy(&bits, &x, 4);
  int   e = (bits >> 23) - 127;
  int32_t mb = (bits & 0x007fffff) | 0x3f800000;
  float m;
  memcpy(&m, &mb, 4);
  if(m > 1.41421354f) { m *= 0.5f; e += 1; }
  const float s  = (m - 1.0f) / (m + 1.0f);
  const float s2 = s * s;
  float p = fmaf(s2, 0.222222222f, 0.285714286f);
  p = fmaf(s2, p, 0.4f);
  p = fmaf(s2, p, 0.666666667f);
  p = fmaf(s2, p, 2.0f);
  return fmaf(p * s, 1.44269504f, (float)e);
}

static inline float exp2_poly(float y)
{
  y = fminf(fmaxf(y, -126.0f), 127.0f);
  const float n = floorf(y);
  const float z = (y - n) * 0.693147181f;
  float p = fmaf(z, 2.75573192e-6f, 2.48015873e-5f);
  p = fmaf(z, p, 1.98412698e-4f);
  p = fmaf(z, p, 1.38888889e-3f);
  p = fmaf(z, p, 8.33333333e-3f);
  p = fmaf(z, p, 4.16666667e-2f);
  p = fmaf(z, p, 1.66666667e-1f);
  p = fmaf(z, p, 0.5f);
  p = fmaf(z, p, 1.0f);
  p = fmaf(z, p, 1.0f);
  int32_t pb;
  memcpy(&pb, &p, 4);
  pb += (int32_t)n << 23;
  memcpy(&p, &pb, 4);
  return p;
}

static inline float post_gamma(float c)
{
  if(!(c > 0.0f)) return 0.0f;
  if(c > 3.0e38f) return c;
  if(c < 1.17549435e-38f) return 0.0f;
  return exp2_poly(0.454545455f * log2_poly(c));  /* post.frag:35-36, gamma = 1/2.2 */
}

/* Same contract as trt_post_dev on host buffers. */
int oracle_post(const float* rgba_in, uint64_t n_pixels, float* f32_out, uint8_t* unorm8_out)
{
  if(n_pixels && !rgba_in) return TRT_E_INVALID;
  for(uint64_t i = 0; i < 4 * n_pixels; ++i)
  {
    const float o = post_gamma(rgba_in[i]);
    if(f32_out) f32_out[i] = o;
    if(unorm8_out) unorm8_out[i] = (uint8_t)rintf(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
  }
  return TRT_OK;
}

/* ------------------------------------------------------------------------------------ */
/* point-cloud re-projection (SEC = ray_tracing__before_second): sequential rasteriser     */
/* ------------------------------------------------------------------------------------ */
/* Same contract as trt_splat_dev: points in primitive order, depth test LESS on a 24-bit
 * UNORM buffer cleared to 1.0 (SEC/hello_vulkan.cpp:235-241, SEC/main.cpp:221-223), point size
 * 2.5 (SEC/shaders/vert_shader.vert:50), colour of the point (frag_shader.frag:40-45). */
int oracle_splat(const trt_point* pts, uint64_t n_points, const float* vp, uint32_t W, uint32_t H,
                 const float* clear, float point_size, float* rgba)
{
  if(!vp || !clear || !rgba || (n_points && !pts) || !W || !H) return TRT_E_INVALID;
  const size_t npx = (size_t)W * H;
  uint32_t* depth = (uint32_t*)malloc(npx * sizeof(uint32_t));
  if(!depth) return TRT_E_NOMEM;
  for(size_t i = 0; i < npx; ++i)
  {
    depth[i] = 0x00FFFFFFu;                                                  /* clear depth 1.0 */
    memcpy(&rgba[4 * i], clear, 4 * sizeof(float));
  }
  const float half = point_size * 0.5f;
  for(uint64_t i = 0; i < n_points; ++i)
  {
    const float* p = pts[i].pos;
    const float cx = fmaf(vp[12], 1.0f, fmaf(vp[8], p[2], fmaf(vp[4], p[1], vp[0] * p[0])));   /* vert:51 */
    const float cy = fmaf(vp[13], 1.0f, fmaf(vp[9], p[2], fmaf(vp[5], p[1], vp[1] * p[0])));
    const float cz = fmaf(vp[14], 1.0f, fmaf(vp[10], p[2], fmaf(vp[6], p[1], vp[2] * p[0])));
    const float cw = fmaf(vp[15], 1.0f, fmaf(vp[11], p[2], fmaf(vp[7], p[1], vp[3] * p[0])));
    if(!(cw > 0.0f && cx >= -cw && cx <= cw && cy >= -cw && cy <= cw && cz >= 0.0f && cz <= cw))
      continue;
    const float iw = 1.0f / cw;
    const float xf = fmaf(cx * iw, 0.5f, 0.5f) * (float)W;
    const float yf = fmaf(cy * iw, 0.5f, 0.5f) * (float)H;
    const uint32_t z24 = (uint32_t)rintf((cz * iw) * 16777215.0f);
    int x0 = (int)ceilf(xf - half - 0.5f), x1 = (int)ceilf(xf + half - 0.5f);
    int y0 = (int)ceilf(yf - half - 0.5f), y1 = (int)ceilf(yf + half - 0.5f);
    if(x0 < 0) x0 = 0;
    if(y0 < 0) y0 = 0;
    if(x1 > (int)W) x1 = (int)W;
    if(y1 > (int)H) y1 = (int)H;
    for(int y = y0; y < y1; ++y)
      for(int x = x0; x < x1; ++x)
      {
        const size_t k = (size_t)y * W + x;
        if(z24 < depth[k])                                                   /* VK_COMPARE_OP_LESS */
        {
          depth[k] = z24;
          rgba[4 * k + 0] = pts[i].color[0];
          rgba[4 * k + 1] = pts[i].color[1];
          rgba[4 * k + 2] = pts[i].color[2];
          rgba[4 * k + 3] = 1.0f;                                            /* frag:44 */
        }
      }
  }
  free(depth);
  return TRT_OK;
}
