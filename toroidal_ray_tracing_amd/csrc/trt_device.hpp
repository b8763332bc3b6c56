// trt_device.hpp — device-side arithmetic of the toroidal ray tracer (gfx950 / CDNA4).
//
// Everything a lane needs to answer "where does this ray first meet this torus", plus the
// closest-hit / miss / shadow-miss shader bodies of the reference restated for an analytic
// torus.  Reference paths are relative to vk_raytracing_tutorial_KHR/
// (REFL = ray_tracing_reflections, BEF = ray_tracing__before).
//
// Arithmetic contract: only IEEE-754 correctly rounded operations
// (+ - * / sqrt fma), every fused multiply-add spelled out, translation unit compiled with
// -ffp-contract=off.  hipcc lowers `/` and sqrtf() to correctly rounded sequences by
// default (-fhip-fp32-correctly-rounded-divide-sqrt), so the results are reproducible on
// any IEEE machine; tests/ check them bit for bit against a CPU restatement.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trt.h"

namespace trt {

// ------------------------------------------------------------------------------------------
// scalar helpers, float / double
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float  fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  sqrt_(float a) { return sqrtf(a); }
__device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
__device__ __forceinline__ float  max_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double max_(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float  min_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float  abs_(float a) { return fabsf(a); }
__device__ __forceinline__ double abs_(double a) { return fabs(a); }

struct v3 { float x, y, z; };

__device__ __forceinline__ float dot3(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ v3 sub3(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 scale3(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ v3 neg3(v3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrt_(dot3(a, a))); }
// GLSL reflect(I,N) = I - 2·dot(N,I)·N   (REFL/shaders/raytrace.rchit:148)
__device__ __forceinline__ v3 reflect3(v3 i, v3 n)
{
  const float k = 2.0f * dot3(n, i);
  return {fma_(-k, n.x, i.x), fma_(-k, n.y, i.y), fma_(-k, n.z, i.z)};
}
// column-major mat4 · (x,y,z,w), rows 0..2
__device__ __forceinline__ v3 mat4_mul(const float* m, float x, float y, float z, float w)
{
  v3 r;
  r.x = fma_(m[12], w, fma_(m[8], z, fma_(m[4], y, m[0] * x)));
  r.y = fma_(m[13], w, fma_(m[9], z, fma_(m[5], y, m[1] * x)));
  r.z = fma_(m[14], w, fma_(m[10], z, fma_(m[6], y, m[2] * x)));
  return r;
}

// ------------------------------------------------------------------------------------------
// scene as the kernels see it (built on the host by trt_api.hip, staged into LDS per block)
// ------------------------------------------------------------------------------------------
template <class Real>
struct TorusK {   // per-torus constants of the solver
  Real cx, cy, cz;
  Real R;
  Real r2;      // r²
  Real rpol;    // r/32: largest step the geometric polish may take
  Real k0;      // R² - r²
  Real Rb2;     // (R+r)²·(1+2⁻⁹): squared radius of the slightly inflated bounding sphere
  Real rs;      // r·(1+2⁻⁸): half height of the slightly inflated bounding slab |y| <= rs
  Real fourR2;  // 4R²
};

struct TorusShade {  // what the closest-hit stage needs: centre, R, material index
  float cx, cy, cz, R;
  int   matId;
};

struct MaterialK {  // the WaveFrontMaterial fields the Phong model reads (rchit:95-155)
  float ambient[3];
  float diffuse[3];
  float specular[3];
  float shininess;
  int   illum;
};

struct SceneK {
  int            n_tori;
  int            n_mat;
  int            f64;   // 1: FP64 root solve (BASELINE config 4), FP32 I/O
  int            dk;    // 0: Fourier–Newton walk; alternative root solvers: 1 Durand–Kerner, 2 Ferrari
  int            order[TRT_MAX_TORI];  // test order: descending bounding radius R + r, ties by index
  uint32_t       inside[TRT_MAX_TORI]; // inside[i]: the tori whose tube lies strictly inside torus i's tube, as a mask over TEST-ORDER
                                       // positions (bit k = order[k]) — what a ray that leaves i's surface outwards cannot hit first
  TorusK<float>  k32[TRT_MAX_TORI];
  TorusK<double> k64[TRT_MAX_TORI];
  TorusShade     shade[TRT_MAX_TORI];
  MaterialK      mat[TRT_MAX_MATERIALS];
};

// Copy the scene constants from the kernel-argument segment into LDS — only the records in use
// (header + test order + enclosure masks, n_tori solver records of the active precision, n_tori shading records,
// n_mat materials): 46 dwords for one FP32 torus instead of the 388 of the full struct, one load per thread.
// Reads in the hot loops then hit LDS at wave-uniform (broadcast) or material-indexed addresses.
// The dwords of a SceneK that are in use, numbered 0 .. scene_words() - 1: word i of them sits at dword scene_word(i) of the struct.
constexpr uint32_t kSceneHdr = 20, kSceneK32 = kSceneHdr, kSceneK64 = kSceneK32 + 80, kSceneShade = kSceneK64 + 160, kSceneMat = kSceneShade + 40;
__device__ __forceinline__ uint32_t scene_words(const SceneK& arg)
{
  return kSceneHdr + (arg.f64 ? 20u : 10u) * (uint32_t)arg.n_tori + 5u * (uint32_t)arg.n_tori + 11u * (uint32_t)arg.n_mat;
}
__device__ __forceinline__ uint32_t scene_word(const SceneK& arg, uint32_t i)
{
  const uint32_t n = (uint32_t)arg.n_tori;
  const uint32_t c0 = kSceneHdr, c1 = c0 + (arg.f64 ? 0u : 10u * n), c2 = c1 + (arg.f64 ? 20u * n : 0u), c3 = c2 + 5u * n;
  return i < c0 ? i : i < c1 ? kSceneK32 + (i - c0) : i < c2 ? kSceneK64 + (i - c1) : i < c3 ? kSceneShade + (i - c2) : kSceneMat + (i - c3);
}
__device__ __forceinline__ void stage_scene(SceneK* lds, const SceneK& arg)
{
  static_assert(sizeof(SceneK) == 4 * (kSceneMat + 88) && sizeof(TorusK<float>) == 40 && sizeof(TorusShade) == 20
                    && sizeof(MaterialK) == 44 && offsetof(SceneK, k32) == 4 * kSceneK32 && offsetof(SceneK, k64) == 4 * kSceneK64
                    && offsetof(SceneK, shade) == 4 * kSceneShade && offsetof(SceneK, mat) == 4 * kSceneMat, "SceneK layout");
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&arg);
  uint32_t*       dst = reinterpret_cast<uint32_t*>(lds);
  const uint32_t  c4 = scene_words(arg);
  for(uint32_t i = threadIdx.x; i < c4; i += blockDim.x)
  {
    const uint32_t off = scene_word(arg, i);
    dst[off] = src[off];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// T1 + T2: first root of the ray–torus quartic in (tmin, tmax)
// ------------------------------------------------------------------------------------------
// Depressed quartic in u = t - tc (tc = parameter of closest approach to the torus centre):
//   f(u) = A4 u⁴ + P2 u² + Q1 u + S0,  A4 = dd², P2 = 2·dd·κ - 4R²a, Q1 = -8R²b, S0 = κ² - 4R²c
// f'' has constant sign on ≤3 pieces of the window (split at ±w); on each piece Newton's
// iteration started where sign f = sign f'' is monotone (Fourier), so walking the pieces
// left to right finds the smallest root or proves there is none.  The walk is written as
// ONE loop whose every trip evaluates (f, f') at one point per lane — the lanes of a wave
// stay convergent on the evaluation and the division although they sit in different
// pieces/modes.
enum : int { M_FWD = 0, M_BWD = 1, M_PROBE = 2, M_END = 3, M_DONE = 4 };
constexpr int kNewtonCap = 48;
// Newton's early stop, as a fraction of the piece: 2^-12 (FP32), 2^-24 (FP64)
template <class Real> struct StepStop;
template <> struct StepStop<float> { static constexpr float v = 0.000244140625f; };
template <> struct StepStop<double> { static constexpr double v = 5.9604644775390625e-08; };

// One ray-vs-torus test as a resumable state machine: setup() does T1 (returns false when the
// bounding sphere / parameter window culls the ray), every step() evaluates (f, f') once and
// advances the walk (returns false when the test is decided), finish() polishes the root and
// applies the open interval (tmin, tmax).  A lane can park a TorusTest in registers between
// step() calls — the persistent kernel relies on that.
template <class Real>
struct TorusTest {
  // quartic and window
  Real A4, P2, Q1, S0, w, hi, tc;
  Real qx, qy, qz;     // point of closest approach to the torus centre (local frame)
  // walk: piece [A,B] with sign(f'') = sigma; xe = point to evaluate next (END/PROBE: xe == B)
  Real A, B, xe, root;
  int  sigma, sref, it, mode;
  bool split, found;
  static constexpr Real kStepStop = StepStop<Real>::v;

  __device__ __forceinline__ bool setup(Real ox, Real oy, Real oz, Real dx_, Real dy_, Real dz_,
                                        Real dd, Real inv_dd, Real tmin, Real tmax,
                                        const TorusK<Real>& T)
  {
    const Real ex = ox - T.cx, ey = oy - T.cy, ez = oz - T.cz;
    const Real n  = fma_(ez, dz_, fma_(ey, dy_, ex * dx_));
    tc = -n * inv_dd;
    qx = fma_(tc, dx_, ex); qy = fma_(tc, dy_, ey); qz = fma_(tc, dz_, ez);
    const Real m = fma_(qz, qz, fma_(qy, qy, qx * qx));
    mode = M_DONE;
    found = false;
    if(!(m <= T.Rb2))
      return false;
    const Real U  = sqrt_((T.Rb2 - m) * inv_dd);
    Real lo = max_(tmin - tc, -U);
    hi = min_(tmax - tc, U);
    if(!(lo < hi))
      return false;
    const Real a     = fma_(dz_, dz_, dx_ * dx_);
    const Real b     = fma_(qz, dz_, qx * dx_);
    const Real c     = fma_(qz, qz, qx * qx);
    // T1b: clip the window to the bounding slab |y| <= rs (slightly inflated, like the sphere, so that the walk still
    // starts strictly outside the torus).  Sphere ∩ slab holds only ≈1.5·r/(R+r) of the sphere's volume: most rays that
    // pass the sphere but miss the torus end here without a single Newton step, and the others start next to the surface.
    // (Round 1 also clipped to the bounding CYLINDER of the same radius — a square root and a division per test for
    // nothing: with q ⟂ d, sphere(u) − cylinder(u) = (dy·u + qy)² >= 0, so the sphere's interval always lies inside the
    // cylinder's.)
    if(dy_ != Real(0))
    {
      const Real inv_dy = Real(1) / dy_;
      const Real u0 = (-T.rs - qy) * inv_dy, u1 = (T.rs - qy) * inv_dy;
      lo = max_(lo, min_(u0, u1));
      hi = min_(hi, max_(u0, u1));
    }
    else if(!(abs_(qy) <= T.rs))
      return false;
    if(!(lo < hi))
      return false;
    const Real kappa = m + T.k0;
    A4    = dd * dd;
    P2    = fma_(-T.fourR2, a, (Real(2) * dd) * kappa);
    Q1    = (Real(-2) * T.fourR2) * b;
    S0    = fma_(-T.fourR2, c, kappa * kappa);
    split = P2 < Real(0);
    const Real k6 = (inv_dd * inv_dd) * Real(1.0 / 6.0);  // 1/(6·A4) without a division
    w     = split ? sqrt_(-P2 * k6) : Real(0);
    A = lo; B = lo; xe = lo; root = Real(0);
    sigma = 1; sref = 1; it = 0; mode = M_END;
    return true;
  }

  // One evaluation of (f, f') at xe, then the transitions of the walk — written as straight-
  // line predicated code (compares + selects, one division) so that the lanes of a wave, which
  // sit in different pieces and modes, execute ONE instruction stream per trip.
  // Returns true while the test is undecided.
  //
  //   END    xe == B was evaluated after a forward run found no root in [A,B]: next piece.
  //   PROBE  xe == B was evaluated because sign f(A) = -sigma: a root exists in the piece iff
  //          sign f(B) = sigma; then iterate backward from B, else next piece.
  //   FWD    Newton from the left end; monotone while a root lies ahead, otherwise sigma·f'
  //          turns >= 0 or the iterate leaves the piece (-> END).
  //   BWD    Newton from the right end onto the single root of the piece.
  __device__ __forceinline__ bool step()
  {
    const Real u  = xe;
    const Real e1 = fma_(A4 * u, u, P2);
    const Real e2 = fma_(e1, u, Q1);
    const Real fe = fma_(e2, u, S0);
    const Real g1 = fma_((Real(4) * A4) * u, u, Real(2) * P2);
    const Real de = fma_(g1, u, Q1);

    const bool mEnd = mode == M_END, mProbe = mode == M_PROBE, mFwd = mode == M_FWD, mBwd = mode == M_BWD;
    const bool pos = fe > Real(0), neg = fe < Real(0), zer = fe == Real(0);
    const int  sfe = pos ? 1 : -1;
    const int  sb  = pos ? 1 : (neg ? -1 : sigma);

    // leaving the piece at B (END, or PROBE without a sign change) / entering the next one
    const bool probe_ok = mProbe && sb == sigma;
    const bool leave    = mEnd || (mProbe && !probe_ok);
    const bool more     = B < hi;
    const bool enter    = leave && more;
    const bool c1 = split && B < -w, c2 = split && B < w;
    const Real Bn = c1 ? min_(-w, hi) : (c2 ? min_(w, hi) : hi);
    const int  sn = (c2 && !c1) ? -1 : 1;
    const bool e_hit = enter && zer;                 // the piece end is itself a root
    const bool e_fwd = enter && !zer && sfe == sn;
    const bool e_prb = enter && !zer && sfe != sn;

    const Real A2     = enter ? B : A;
    const Real B2     = enter ? Bn : B;
    const int  sigma2 = enter ? sn : sigma;
    const int  sref2  = enter ? sfe : (probe_ok ? sb : sref);
    const int  it2    = (mFwd || mBwd) ? it + 1 : 0;

    // Newton step from u (the current iterate is always the point just evaluated)
    const bool f_flip = mFwd && (zer || sfe != sref);      // rounding carried f across zero
    const bool isF    = (mFwd && !f_flip) || e_fwd;
    const bool isB    = mBwd || probe_ok;
    const bool cap    = (isF || isB) && it2 == kNewtonCap;
    const Real sdx    = sigma2 > 0 ? de : -de;
    const Real st     = fe / de;
    const Real xn     = u - st;
    // Newton stops once a step is below 2^-12 (FP32; 2^-24 FP64) of the piece: the iterate is
    // then within ~step²/r of the root, far inside the capture range of finish()'s geometric
    // step — the one or two evaluations that used to confirm the fixed point are not spent
    const bool small  = abs_(st) <= (B2 - A2) * kStepStop;
    const bool f_no   = isF && !cap && (!(sdx < Real(0)) || !(xn < B2));  // no root in [u,B]
    const bool f_cv   = isF && !cap && !f_no && xn == u;
    const bool f_es   = isF && !cap && !f_no && xn != u && small;
    const bool b_stop = (zer || sfe != sref2) || !(sdx > Real(0));
    const bool b_left = !(xn > A2);
    const bool b_hitx = isB && !cap && (b_stop || (!b_left && xn == u));
    const bool b_hitA = isB && !cap && !b_stop && b_left;
    const bool b_es   = isB && !cap && !b_stop && !b_left && xn != u && small;
    const bool go     = (isF && !cap && !f_no && !f_cv && !f_es) || (isB && !cap && !b_stop && !b_left && xn != u && !b_es);

    const bool hit_x = e_hit || f_flip || cap || f_cv || b_hitx;
    const bool hit_n = f_es || b_es;   // accepted at the new iterate
    // a forward run that proves "no root" in the LAST piece decides the test: nothing is
    // evaluated at hi any more (that evaluation carried no information)
    const bool last_no = f_no && !(B2 < hi);
    const bool done  = (leave && !more) || hit_x || hit_n || b_hitA || last_no;
    found = hit_x || hit_n || b_hitA;
    root  = b_hitA ? A2 : (hit_n ? xn : u);
    A = A2; B = B2; sigma = sigma2; sref = sref2; it = it2;
    xe   = (e_prb || f_no) ? B2 : (go ? xn : u);
    mode = done ? M_DONE : (e_prb ? M_PROBE : (f_no ? M_END : (isF ? M_FWD : M_BWD)));
    return !done;
  }

  // step() for a lane that is iterating (mode FWD or BWD): the same arithmetic and the same
  // decisions, minus every piece-transition term.  The solve loops take this path on the
  // trips in which no running lane of the wave sits at a piece end (wave-uniform choice).
  __device__ __forceinline__ bool step_iter()
  {
    const Real u  = xe;
    const Real e1 = fma_(A4 * u, u, P2);
    const Real e2 = fma_(e1, u, Q1);
    const Real fe = fma_(e2, u, S0);
    const Real g1 = fma_((Real(4) * A4) * u, u, Real(2) * P2);
    const Real de = fma_(g1, u, Q1);
    const bool fwd  = mode == M_FWD;
    const bool flip = fe == Real(0) || ((fe > Real(0)) ? 1 : -1) != sref;
    const int  it2  = it + 1;
    const bool cap  = it2 == kNewtonCap;
    const Real sdx  = sigma > 0 ? de : -de;
    const Real st   = fe / de;
    const Real xn   = u - st;
    const bool small = abs_(st) <= (B - A) * kStepStop;
    // forward: flip -> hit, cap -> hit, slope/range -> END, converged -> hit, small step -> hit(xn)
    // backward: cap -> hit, flip/slope -> hit, left of A -> hit(A), converged -> hit, small step -> hit(xn)
    const bool f_no   = fwd && !flip && !cap && (!(sdx < Real(0)) || !(xn < B));
    const bool b_stop = flip || !(sdx > Real(0));
    const bool b_hitA = !fwd && !cap && !b_stop && !(xn > A);
    const bool hit_x  = fwd ? (flip || cap || (!f_no && xn == u))
                            : (cap || b_stop || (!b_hitA && xn == u));
    const bool hit_n  = !hit_x && !f_no && !b_hitA && small;
    const bool last_no = f_no && !(B < hi);   // "no root" in the last piece: the test is a miss
    const bool done   = hit_x || hit_n || b_hitA || last_no;
    found = hit_x || hit_n || b_hitA;
    root  = b_hitA ? A : (hit_n ? xn : u);
    it    = it2;
    xe    = f_no ? B : (done ? u : xn);
    mode  = done ? M_DONE : (f_no ? M_END : mode);
    return !done;
  }

  // true when step_iter() applies to this lane
  __device__ __forceinline__ bool iterating() const { return mode == M_FWD || mode == M_BWD; }

  // The same walk as step()/step_iter() — same evaluation points, same decisions, hence the same
  // root bit for bit — written as NESTED loops for a lane that owns its test from start to finish
  // (trace, listed and static kernels): an outer loop over the <= 3 pieces of constant sign of f'',
  // an inner Newton loop that serves the forward run (from A, while sign f = sigma) and the backward
  // run (from B, onto the single root of a piece with a sign change) with ONE body: a trip is the
  // evaluation, the division and five compares (~40 issue slots) instead of the ~170 of the general
  // transition table of step(), and the piece bookkeeping runs once per piece instead of once per
  // trip.  Lanes of a wave in different pieces still share the inner loop.  `evals` counts (f, f').
  // Call after a successful setup(); then finish().
  __device__ __forceinline__ void walk(uint32_t& evals)
  {
    const Real A4x4 = Real(4) * A4, P2x2 = Real(2) * P2;
    auto eval = [&](Real u, Real& f, Real& d) {
      const Real e1 = fma_(A4 * u, u, P2);
      const Real e2 = fma_(e1, u, Q1);
      f = fma_(e2, u, S0);
      const Real g1 = fma_(A4x4 * u, u, P2x2);
      d = fma_(g1, u, Q1);
    };
    Real a = A, fa, da;          // setup() leaves the window start in A
    eval(a, fa, da);
    ++evals;
    found = false;
    root  = Real(0);
    mode  = M_DONE;
    for(;;)
    {
      const bool c1 = split && a < -w, c2 = split && a < w;
      const Real b  = c1 ? min_(-w, hi) : (c2 ? min_(w, hi) : hi);
      const Real sg = (c2 && !c1) ? Real(-1) : Real(1);          // sign of f'' on the piece
      if(fa == Real(0)) { found = true; root = a; break; }       // the piece end is itself a root
      const bool bwd = (fa > Real(0)) != (sg > Real(0));         // sign f(A) = -sigma: probe B
      Real x = a, fx = fa, dx = da;
      bool pos = fa > Real(0);                                   // the sign the run must keep
      bool run = true, noroot = false;
      if(bwd)
      {
        eval(b, fx, dx);
        ++evals;
        x = b;
        if(fx == Real(0)) { found = true; root = b; break; }     // (oracle: sb = sigma, first check of the run)
        pos = fx > Real(0);
        if(pos != (sg > Real(0))) { run = false; noroot = true; }  // no sign change: no root in the piece
      }
      if(run)
      {
        const Real ustop = (b - a) * kStepStop;
        const Real sd    = bwd ? -sg : sg;      // forward needs sigma·f' < 0, backward sigma·f' > 0
        const Real lim   = bwd ? -a : b;        // forward: xn < B; backward: xn > A  ⇔  -xn < -A
        for(int it = 0; it < kNewtonCap; ++it)
        {
          if(!(sd * dx < Real(0))) { noroot = !bwd; break; }                 // forward: no root ahead; backward: stop here
          const Real st = fx / dx;
          const Real xn = x - st;
          if(!((bwd ? -xn : xn) < lim)) { noroot = !bwd; if(bwd) x = a; break; }
          if(xn == x) break;
          x = xn;
          if(abs_(st) <= ustop) break;
          eval(x, fx, dx);
          ++evals;
          if(fx == Real(0) || (fx > Real(0)) != pos) break;
        }
        if(!noroot) { found = true; root = x; break; }
      }
      if(!(b < hi)) break;                       // that was the last piece: a miss
      if(!bwd) { eval(b, fx, dx); ++evals; }     // (a failed probe has evaluated B already)
      a = b; fa = fx; da = dx;
    }
  }

  // T2, alternative solver (TRT_SOLVE_DK_*): Durand–Kerner iteration on the monic depressed
  // quartic u⁴ + p·u² + q·u + s — four complex iterates from the spiral (0.4+0.9i)^k scaled by the
  // bounding-sphere radius, updated in place for a FIXED number of sweeps (every lane runs the
  // same trip count), only + - * / fma.  Real candidates (|Im| <= tol·scale) get two guarded
  // Newton steps; the smallest one inside [lo,hi] becomes `root`.  (tests/ compare it bit for bit
  // with a CPU restatement.)  Call after a successful setup(); then finish().
  __device__ __forceinline__ void solve_dk(Real inv_dd, Real Rb2)
  {
    constexpr int  kSweeps = 24;
    constexpr Real kTol = sizeof(Real) == 4 ? Real(0.0009765625) : Real(2.384185791015625e-07);  // 2^-10 / 2^-22
    const Real CR[4] = {Real(1.0), Real(0.4), Real(-0.65), Real(-0.908)};
    const Real CI[4] = {Real(0.0), Real(0.9), Real(0.72), Real(-0.297)};
    const Real iA4 = inv_dd * inv_dd;
    const Real p = P2 * iA4, q = Q1 * iA4, s = S0 * iA4;
    const Real sc  = sqrt_(Rb2 * inv_dd);
    const Real tol = kTol * sc;
    const Real lo  = A;   // setup() leaves the window start in A (= B = xe)
    Real zr[4], zi[4];
#pragma unroll
    for(int k = 0; k < 4; ++k) { zr[k] = sc * CR[k]; zi[k] = sc * CI[k]; }
#pragma unroll 1
    for(int sweep = 0; sweep < kSweeps; ++sweep)
    {
#pragma unroll
      for(int k = 0; k < 4; ++k)
      {
        const Real x = zr[k], y = zi[k];
        const Real ar = fma_(x, x, -(y * y)) + p, ai = (x + x) * y;
        const Real br = fma_(ar, x, -(ai * y)) + q, bi = fma_(ar, y, ai * x);
        const Real fr = fma_(br, x, -(bi * y)) + s, fi = fma_(br, y, bi * x);
        Real dr = Real(1), di = Real(0);
#pragma unroll
        for(int j = 0; j < 4; ++j)
          if(j != k)
          {
            const Real er = x - zr[j], ei = y - zi[j];
            const Real nr = fma_(dr, er, -(di * ei)), ni = fma_(dr, ei, di * er);
            dr = nr; di = ni;
          }
        const Real den = fma_(dr, dr, di * di);
        if(den > Real(0) && den < Real(__builtin_inf()))
        {
          const Real inv = Real(1) / den;
          zr[k] = x - fma_(fr, dr, fi * di) * inv;
          zi[k] = y - fma_(fi, dr, -(fr * di)) * inv;
        }
      }
    }
    bool f = false;
    Real best = Real(0);
    const Real A4x4 = Real(4) * A4, P2x2 = Real(2) * P2;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
      if(!(abs_(zi[k]) <= tol))
        continue;
      Real u = zr[k];
#pragma unroll
      for(int n = 0; n < 2; ++n)
      {
        const Real e1 = fma_(A4 * u, u, P2), e2 = fma_(e1, u, Q1), fu = fma_(e2, u, S0);
        const Real g1 = fma_(A4x4 * u, u, P2x2), du = fma_(g1, u, Q1);
        const Real st = fu / du;
        if(abs_(st) <= tol)
          u = u - st;
      }
      if(u >= lo && u <= hi && (!f || u < best)) { best = u; f = true; }
    }
    found = f;
    root  = best;
    mode  = M_DONE;
  }

  // T2, second alternative solver (TRT_SOLVE_FERRARI_*): Ferrari's factorisation of the monic
  // depressed quartic into (u² - σu + e + h)(u² + σu + e - h), σ = sqrt(2m), e = p/2 + m,
  // h = sign(q)·sqrt(m² + pm + (p² - 4s)/4), m >= 0 a root of the resolvent cubic
  // m³ + pm² + ((p² - 4s)/4)m - q²/8 — bracketed by [0, Cauchy bound], narrowed by a FIXED number
  // of bisections (every lane the same trip count) and two guarded Newton steps, i.e. without the
  // cbrt/acos of the textbook form, only + - * / sqrt fma (bit for bit the CPU restatement of
  // tests/).  Candidates are polished and selected exactly as in solve_dk().
  __device__ __forceinline__ void solve_ferrari(Real inv_dd, Real Rb2)
  {
    constexpr int  kSteps = sizeof(Real) == 4 ? 40 : 72;
    constexpr Real kTol = sizeof(Real) == 4 ? Real(0.0009765625) : Real(2.384185791015625e-07);
    const Real iA4 = inv_dd * inv_dd;
    const Real p = P2 * iA4, q = Q1 * iA4, s = S0 * iA4;
    const Real sc  = sqrt_(Rb2 * inv_dd);
    const Real tol = kTol * sc;
    const Real lo  = A;   // setup() leaves the window start in A
    const Real c1  = fma_(p, p, Real(-4) * s) * Real(0.25);
    const Real c0  = (q * q) * Real(-0.125);
    Real mlo = Real(0), mhi = Real(1) + max_(abs_(p), max_(abs_(c1), abs_(c0)));
#pragma unroll 1
    for(int i = 0; i < kSteps; ++i)
    {
      const Real mid = Real(0.5) * (mlo + mhi);
      const Real r   = fma_(fma_(mid + p, mid, c1), mid, c0);
      if(r > Real(0)) mhi = mid;
      else mlo = mid;
    }
    Real m = Real(0.5) * (mlo + mhi);
#pragma unroll
    for(int i = 0; i < 2; ++i)
    {
      const Real r  = fma_(fma_(m + p, m, c1), m, c0);
      const Real dr = fma_(fma_(Real(3), m, p + p), m, c1);
      const Real mn = m - r / dr;
      if(dr > Real(0) && mn >= mlo && mn <= mhi)
        m = mn;
    }
    const Real sg = sqrt_(m + m);
    const Real h2 = fma_(m + p, m, c1);
    const Real h  = (q < Real(0) ? Real(-1) : Real(1)) * sqrt_(max_(h2, Real(0)));
    const Real e  = fma_(Real(0.5), p, m);
    const Real d1 = fma_(Real(-4), e + h, sg * sg), d2 = fma_(Real(-4), e - h, sg * sg);
    const Real w1 = sqrt_(max_(d1, Real(0))), w2 = sqrt_(max_(d2, Real(0)));
    const Real cand[4] = {Real(0.5) * (sg - w1), Real(0.5) * (sg + w1), Real(0.5) * (-sg - w2), Real(0.5) * (-sg + w2)};
    const bool ok[4]   = {d1 >= Real(0), d1 >= Real(0), d2 >= Real(0), d2 >= Real(0)};
    bool f = false;
    Real best = Real(0);
    const Real A4x4 = Real(4) * A4, P2x2 = Real(2) * P2;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
      if(!ok[k])
        continue;
      Real u = cand[k];
#pragma unroll
      for(int n = 0; n < 2; ++n)
      {
        const Real e1 = fma_(A4 * u, u, P2), e2 = fma_(e1, u, Q1), fu = fma_(e2, u, S0);
        const Real g1 = fma_(A4x4 * u, u, P2x2), du = fma_(g1, u, Q1);
        const Real st = fu / du;
        if(abs_(st) <= tol)
          u = u - st;
      }
      if(u >= lo && u <= hi && (!f || u < best)) { best = u; f = true; }
    }
    found = f;
    root  = best;
    mode  = M_DONE;
  }

  // T2b: one Newton step on g(u) = (ρ-R)² + py² - r², whose rounding error scales with r²
  // instead of R⁴ (a step above r/32 — grazing, g' ≈ 0 — is discarded); then t = u + tc and
  // the open-interval test of the closest-hit query.
  __device__ __forceinline__ bool finish(Real dx_, Real dy_, Real dz_, Real tmin, Real tmax,
                                         const TorusK<Real>& T, Real& t_out) const
  {
    if(!found)
      return false;
    Real r = root;
    const Real px  = fma_(r, dx_, qx), py = fma_(r, dy_, qy), pz = fma_(r, dz_, qz);
    const Real rho = sqrt_(fma_(pz, pz, px * px));
    const Real e   = rho - T.R;
    const Real g   = fma_(e, e, fma_(py, py, -T.r2));
    const Real s   = fma_(pz, dz_, px * dx_);
    const Real gh  = fma_(e, s, (py * dy_) * rho);  // ρ·g'/2
    const Real du  = Real(0.5) * ((g * rho) / gh);
    if(abs_(du) <= T.rpol)
      r = r - du;
    const Real t = r + tc;
    if(!(t > tmin && t < tmax))
      return false;
    t_out = t;
    return true;
  }
};

// Work counters of a lane (dead code in the kernels that do not store them): of the ray–torus
// tests a lane executed, how many passed TorusTest::setup (a quartic was built and walked) and how
// many evaluations of (f, f') the walks took — the numerator of the FLOP/s figure of bench.py.
struct WorkCount {
  uint32_t traced = 0;  // tests that ran TorusTest::setup (pixels of CLEAR tiles never do)
  uint32_t solved = 0;  // … and passed it
  uint32_t evals  = 0;  // evaluations of (f, f') by the default solver's walk
};

// Which form of the walk a kernel runs (both visit the same points and take the same decisions):
//   kWalkNested  TorusTest::walk(): cheapest per trip, but forward runs, backward runs and the pieces of
//                different lanes execute one after the other — the choice for COHERENT rays (render kernels:
//                config 4 −5 %, against the table);
//   kWalkTable   TorusTest::step(), first trip peeled: every lane advances on every trip whatever its piece and
//                mode — the choice for INCOHERENT rays (trt_trace on random aimed rays: 0.087 against 0.113 ms).
enum : int { kWalkNested = 0, kWalkTable = 1 };
#ifdef TRT_RENDER_WALK_TABLE   // timing builds: the table in the render kernels too
constexpr int kRenderWalk = kWalkTable;
#else
constexpr int kRenderWalk = kWalkNested;
#endif

template <class Real, bool DK = false, int WALK = kRenderWalk>
__device__ __forceinline__ bool torus_first_hit(Real ox, Real oy, Real oz, Real dx_, Real dy_,
                                                Real dz_, Real dd, Real inv_dd, Real tmin, Real tmax,
                                                const TorusK<Real>& T, Real& t_out, WorkCount& wc, int alt = 1)
{
  TorusTest<Real> q;
  ++wc.traced;
  if(!q.setup(ox, oy, oz, dx_, dy_, dz_, dd, inv_dd, tmin, tmax, T))
    return false;
  ++wc.solved;
  if(DK)
  {
    if(alt == 2) q.solve_ferrari(inv_dd, T.Rb2);   // wave-uniform: the scene's solver
    else q.solve_dk(inv_dd, T.Rb2);
  }
  else
  {
    if(WALK == kWalkTable)
    {
      // the resumable state machine of the persistent kernel, first trip peeled (right after setup()
      // every field the transitions read is a known constant, so the compiler folds step())
      ++wc.evals;
      bool run = q.step();
      // (Letting the lanes at a piece end WAIT while others iterate — so that more trips are the cheap iterate-only ones —
      // was measured: bit-identical, and slower whatever the threshold: aimed rays 0.085 → 0.107 / 0.126 / 0.147 ms with 8 /
      // 16 / 32 waiting lanes, profiles/r03_trace_wait.txt.  Idle lanes cost more than the cheaper trips bring.)
      while(run)
      {
        ++wc.evals;
        if(__any(!q.iterating()))
          run = q.step();
        else
          run = q.step_iter();
      }
    }
    else
      q.walk(wc.evals);
  }
  return q.finish(dx_, dy_, dz_, tmin, tmax, T, t_out);
}

// Solver constants of torus i in the kernel's precision.
template <class Real> __device__ __forceinline__ const TorusK<Real>& torus_k(const SceneK& S, int i);
template <> __device__ __forceinline__ const TorusK<float>&  torus_k<float>(const SceneK& S, int i) { return S.k32[i]; }
template <> __device__ __forceinline__ const TorusK<double>& torus_k<double>(const SceneK& S, int i) { return S.k64[i]; }

// Per-ray constants of a query in the solver precision (FP32 I/O, FP32 or FP64 solve).
template <class Real>
struct RayK {
  // the ray and its interval stay FP32 (their conversion to the solver precision is exact and
  // costs one v_cvt each where a test uses them); only |d|² and its reciprocal are kept in the
  // solver precision — in the FP64 kernels that is 12 VGPRs less per lane
  float ox, oy, oz, dx, dy, dz, tmin, tmax;
  Real  dd, inv_dd;
  __device__ __forceinline__ void set(v3 o, v3 d, float tmin_, float tmax_)
  {
    ox = o.x; oy = o.y; oz = o.z; dx = d.x; dy = d.y; dz = d.z;
    const Real x = dx, y = dy, z = dz;
    dd     = fma_(z, z, fma_(y, y, x * x));
    inv_dd = Real(1) / dd;
    tmin = tmin_; tmax = tmax_;
  }
};

// Result of a test in the solver precision -> FP32 t; rounding may land on the open bounds.
__device__ __forceinline__ bool round_t(float t, float, float, float& out) { out = t; return true; }
__device__ __forceinline__ bool round_t(double t, float tmin, float tmax, float& out)
{
  const float tf = (float)t;
  if(!(tf > tmin && tf < tmax))
    return false;
  out = tf;
  return true;
}

// One ray against torus i over the open interval (tmin, tmax); t rounded to FP32.
template <class Real, bool DK = false, int WALK = kRenderWalk>
__device__ __forceinline__ bool torus_hit(const SceneK& S, int i, const RayK<Real>& r, float tmin, float tmax, float& t, WorkCount& wc)
{
  Real tt;
  if(!torus_first_hit<Real, DK, WALK>((Real)r.ox, (Real)r.oy, (Real)r.oz, (Real)r.dx, (Real)r.dy, (Real)r.dz, r.dd, r.inv_dd, (Real)r.tmin, (Real)tmax,
                                torus_k<Real>(S, i), tt, wc, S.dk))
    return false;
  return round_t(tt, tmin, tmax, t);
}

// Closest hit over the tori — the role of traceRayEXT + BVH (REFL/shaders/raytrace.rgen:64-75).
// Tori are tested in S.order (largest bounding sphere first) and the interval of every later
// test ends at the closest hit so far: behind an enclosing shell the remaining tests end in
// their window clip without a Newton step.  Equal t keeps the torus tested first.
// `skip` (a mask over test-order positions) names the tori this ray cannot hit first: tubes that lie strictly inside a
// tube the ray's origin is known to be OUTSIDE of (enclosure cull) — they count as tests and cost nothing.
// Returns the torus index or -1; `tests` counts ray–torus tests.
template <class Real, bool DK = false, int WALK = kRenderWalk>
__device__ __forceinline__ int closest_hit(const SceneK& S, v3 o, v3 d, float tmin, float tmax,
                                           float& t_out, uint32_t& tests, WorkCount& wc, uint32_t skip = 0u)
{
  RayK<Real> r;
  r.set(o, d, tmin, tmax);
  int   id   = -1;
  float best = __builtin_inff();
  for(int k = 0; k < S.n_tori; ++k)
  {
    const int i = S.order[k];
    float t;
    ++tests;
    if((skip >> k) & 1u)
      continue;
#ifdef TRT_FULL_WINDOW   // timing experiment only (the tail of config 4): every torus over the FULL interval, minimum afterwards
    if(torus_hit<Real, DK, WALK>(S, i, r, tmin, tmax, t, wc) && t < best)
#else
    if(torus_hit<Real, DK, WALK>(S, i, r, tmin, min_(tmax, best), t, wc))
#endif
    {
      best = t;
      id   = i;
    }
  }
  t_out = best;
  return id;
}

// Any hit — the shadow query with gl_RayFlagsTerminateOnFirstHitEXT (REFL/shaders/raytrace.rchit:114-131).
template <class Real, bool DK = false>
__device__ __forceinline__ bool any_hit(const SceneK& S, v3 o, v3 d, float tmin, float tmax,
                                        uint32_t& tests, WorkCount& wc, uint32_t skip = 0u)
{
  RayK<Real> r;
  r.set(o, d, tmin, tmax);
  for(int k = 0; k < S.n_tori; ++k)
  {
    float t;
    ++tests;
    if((skip >> k) & 1u)
      continue;
    if(torus_hit<Real, DK>(S, S.order[k], r, tmin, tmax, t, wc))
      return true;
  }
  return false;
}

// T4: outward unit normal N = normalize(P - q), q = nearest point of the centre circle
// (role of REFL/shaders/raytrace.rchit:74-75; never flipped towards the ray).
__device__ __forceinline__ v3 torus_normal(const TorusShade& T, v3 P)
{
  const v3    pl  = sub3(P, v3{T.cx, T.cy, T.cz});
  const float rho = sqrt_(fma_(pl.z, pl.z, pl.x * pl.x));
  const float k   = (rho - T.R) / rho;
  return normalize3(v3{pl.x * k, pl.y, pl.z * k});
}

// ------------------------------------------------------------------------------------------
// Phong helpers — REFL/shaders/wavefront.glsl:22-48
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 compute_diffuse(const MaterialK& m, v3 L, v3 N)
{
  const float dotNL = max_(dot3(N, L), 0.0f);
  v3 c = {m.diffuse[0] * dotNL, m.diffuse[1] * dotNL, m.diffuse[2] * dotNL};
  if(m.illum >= 1)
  {
    c.x += m.ambient[0];
    c.y += m.ambient[1];
    c.z += m.ambient[2];
  }
  return c;
}

__device__ __forceinline__ v3 compute_specular(const MaterialK& m, v3 viewDir, v3 L, v3 N)
{
  if(m.illum < 2)
    return {0.0f, 0.0f, 0.0f};
  const float kPi        = 3.14159265f;
  const float kShininess = max_(m.shininess, 4.0f);
  const float kEnergy    = (2.0f + kShininess) / (2.0f * kPi);
  const v3    V          = normalize3(neg3(viewDir));
  const v3    R          = reflect3(neg3(L), N);
  // GLSL pow(x, y) is specified as exp2(y * log2(x)) (GLSL 4.60 §8.2): v_log_f32 / v_exp_f32,
  // two quarter-rate instructions instead of OCML's 230-instruction correctly-rounded powf.
  // For results that matter (x^y > 1e-3) the relative error stays below 1e-6.
  const float s          = kEnergy * __builtin_amdgcn_exp2f(kShininess * __builtin_amdgcn_logf(max_(dot3(V, R), 0.0f)));
  return {m.specular[0] * s, m.specular[1] * s, m.specular[2] * s};
}

// ------------------------------------------------------------------------------------------
// ray generation
// ------------------------------------------------------------------------------------------
// Per-frame constants of the toroidal camera, computed once on the host
// (BEF/shaders/raytrace.rgen:36-53 depend only on the UBO and rho), and per-column /
// per-row trigonometry tables (BEF rgen:25-28,56-57: alfa depends on x only, beta on y only).
struct ToroCam {
  float        eye[3];
  float        rho;
  const float* cos_a;  // [W]  cos(radians(alfa + omega))
  const float* sin_a;  // [W]
  const float* cos_b;  // [H]  cos(radians(beta + theta))
  const float* sin_b;  // [H]
};

__device__ __forceinline__ void raygen(const trt_globals& g, const ToroCam& tc, uint32_t W, uint32_t H,
                                       int camera, uint32_t x, uint32_t y, v3& origin, v3& dir)
{
  if(camera == TRT_CAMERA_TOROIDAL)
  {
    // the table pointers come out of the LDS-staged arguments: tell hipcc they address global memory
    // (global_load instead of flat_load, which would also count on lgkmcnt)
    typedef __attribute__((address_space(1))) const float* gcf;
    const float ca = ((gcf)tc.cos_a)[x], sa = ((gcf)tc.sin_a)[x], cb = ((gcf)tc.cos_b)[y], sb = ((gcf)tc.sin_b)[y];
    origin = {fma_(tc.rho, ca, tc.eye[0]), tc.eye[1], fma_(tc.rho, sa, tc.eye[2])};  // BEF rgen:56
    dir    = {ca * cb, sb, sa * cb};                                                // BEF rgen:57
    return;
  }
  // pinhole, REFL/shaders/raytrace.rgen:42-48
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float u = px / (float)W, v = py / (float)H;
  const float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;
  origin       = mat4_mul(g.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f);
  const v3 tgt = mat4_mul(g.projInverse, dx, dy, 1.0f, 1.0f);
  const v3 tn  = normalize3(tgt);
  dir          = mat4_mul(g.viewInverse, tn.x, tn.y, tn.z, 0.0f);
}

// ------------------------------------------------------------------------------------------
// post pass: pow(c, 1/2.2) with exact-operation log2 / exp2 (REFL/shaders/post.frag:33-37)
// ------------------------------------------------------------------------------------------
// log2(x), x a positive normal float: x = 2^e·m, m ∈ [√½, √2); ln(m) = 2·atanh(s), s = (m-1)/(m+1),
// odd series to s⁹ (|s| <= 0.172: truncation < 5e-10); only + - * / fma and integer bit moves.
__device__ __forceinline__ float log2_poly(float x)
{
  int   bits = __float_as_int(x);
  int   e    = (bits >> 23) - 127;
  float m    = __int_as_float((bits & 0x007fffff) | 0x3f800000);
  if(m > 1.41421354f) { m *= 0.5f; e += 1; }
  const float s  = (m - 1.0f) / (m + 1.0f);
  const float s2 = s * s;
  float p = fma_(s2, 0.222222222f, 0.285714286f);   // 2/9, 2/7
  p = fma_(s2, p, 0.4f);                             // 2/5
  p = fma_(s2, p, 0.666666667f);                     // 2/3
  p = fma_(s2, p, 2.0f);
  return fma_(p * s, 1.44269504f, (float)e);         // ln -> log2, plus the exponent
}
// 2^y for y in [-126, 127]: n = floor(y), 2^f = exp(f·ln2) by its Taylor polynomial to f⁹, then
// the exponent is added to the bit pattern.
__device__ __forceinline__ float exp2_poly(float y)
{
  y = min_(max_(y, -126.0f), 127.0f);
  const float n = floorf(y);
  const float z = (y - n) * 0.693147181f;
  float p = fma_(z, 2.75573192e-6f, 2.48015873e-5f);  // 1/9!, 1/8!
  p = fma_(z, p, 1.98412698e-4f);                     // 1/7!
  p = fma_(z, p, 1.38888889e-3f);                     // 1/6!
  p = fma_(z, p, 8.33333333e-3f);                     // 1/5!
  p = fma_(z, p, 4.16666667e-2f);                     // 1/4!
  p = fma_(z, p, 1.66666667e-1f);                     // 1/3!
  p = fma_(z, p, 0.5f);
  p = fma_(z, p, 1.0f);
  p = fma_(z, p, 1.0f);
  return __int_as_float(__float_as_int(p) + ((int)n << 23));
}
// pow(c, 1/2.2) as the post shader applies it; non-positive and NaN inputs give 0, +inf stays.
__device__ __forceinline__ float post_gamma(float c)
{
  if(!(c > 0.0f)) return 0.0f;
  if(c > 3.0e38f) return c;
  if(c < 1.17549435e-38f) return 0.0f;  // subnormal colours
  return exp2_poly(0.454545455f * log2_poly(c));
}

}  // namespace trt
