// trt_api.hip — implementation of the C ABI declared in include/trt.h.
//
// Host side of the drop-in boundary: validates arguments, derives the per-torus solver
// constants and the toroidal camera frame, owns grow-only device staging buffers for the
// host-pointer entry points, and launches the gfx950 kernels.  Replaces what
// HelloVulkan::raytrace + the descriptor-set / push-constant plumbing do in the reference
// (REFL/hello_vulkan.cpp:913-935, BEF/hello_vulkan.cpp:936-958).  Nothing in here computes
// a ray on the CPU: without a HIP device trt_create fails.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "trt_kernels.hpp"

using namespace trt;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void*  p   = nullptr;
  size_t cap = 0;
};

}  // namespace

struct trt_ctx {
  int           device    = 0;
  int           n_cus     = 256;
  int           precision = TRT_SOLVE_F32;
  RenderVariant variant   = kRenderListed;
  bool          stats_on  = false;
  int           classify  = TRT_CLASSIFY_AUTO;
  std::string   err;
  // Events instead of remembered stream handles (the caller may destroy a stream between calls):
  hipEvent_t    ev_toro  = nullptr;   // the last upload of the toroidal tables has read its host staging
  hipEvent_t    ev_stats = nullptr;   // the last counted launch is done
  bool          ev_toro_set = false, ev_stats_set = false;

  Tuning        tn;                       // launch-shape knobs: defaults, or the environment ONCE in a -DTRT_TUNING build
  unsigned long long* d_stats = nullptr;  // [8]: trt_kernels.hip block_add_stats
  unsigned int*       d_queue = nullptr;  // words 0..2: classification accumulators (zero between frames),
                                          // words 32..33: the published list lengths (trt_kernels.hpp RenderArgs)
  uint64_t            stats_pixels = 0;

  // toroidal camera tables: device copy + pinned host staging + cache key
  DevBuf d_toro;
  float* h_toro     = nullptr;
  size_t h_toro_cap = 0;
  struct { uint32_t W = 0, H = 0; float omega = 0, theta = 0; bool valid = false; } toro_key;

  // staging for the host-pointer entry points (grow-only, freed in trt_destroy)
  DevBuf d_in[6], d_out[8], d_rgba, d_rendered;
  DevBuf d_tiles;  // LIVE + CLEAR tile lists of the persistent kernel
  DevBuf d_cost;   // cost feedback: one word per macro tile (zero = no history)
  DevBuf d_keys;   // depth|index keys of trt_splat_dev (one-pass form)
  DevBuf d_bins;   // … binned form: per-bin count / offset / cursor words (count zero between calls)
  DevBuf d_recs;   // … binned form: point records sorted by bin
  std::vector<void*> retired;   // scratch blocks replaced by larger ones: a hipGraph captured earlier may still use them
  bool               bins_dirty = false;   // a re-projection failed between its count and its resolve: zero d_bins before the next one

  // The scene the caller passed last, validated and turned into kernel constants: a frame loop passes the same few
  // hundred bytes every frame, and a 1/8-part frame is short enough for the host's share of a launch to show.
  struct SceneCache {
    bool         valid = false;
    int          precision = -1;
    uint32_t     n_tori = 0, n_mat = 0;
    trt_torus    tori[TRT_MAX_TORI];
    trt_material mat[TRT_MAX_MATERIALS];
    SceneK       K;
  } scene_cache;
};

namespace {

int fail(trt_ctx* ctx, int code, const char* fmt, ...)
{
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if(ctx) ctx->err = buf;
  else g_create_error = buf;
  return code;
}

#define TRT_HIP(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if(e_ != hipSuccess)                                                                    \
      return fail(ctx, e_ == hipErrorOutOfMemory ? TRT_E_NOMEM : TRT_E_HIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                   \
  } while(0)

bool capturing(hipStream_t st)
{
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if(hipStreamIsCapturing(st, &cap) != hipSuccess)
  {
    (void)hipGetLastError();
    return false;
  }
  return cap == hipStreamCaptureStatusActive;
}


// Grow-only scratch.  Scratch that a *_dev entry point hands to a kernel (tile lists, cost words, toroidal tables,
// re-projection keys / bins / records: `graph_visible`) is RETIRED when a larger block replaces it, not freed, until
// trt_destroy: a hipGraph captured earlier keeps replaying on the old block (its kernel arguments hold the old address
// and capacity), and work still in flight on another stream may be using it; such blocks grow by at least half, so that
// a ctx driven through increasing sizes retains at most ≈3× its peak.  The staging buffers of the host-pointer entry
// points (which synchronise before they return and cannot be captured) are freed at once.  hipMalloc is illegal while
// `st` is being captured — then the call is refused instead (include/trt.h: size the ctx with an eager call first).
int grow(trt_ctx* ctx, DevBuf& b, size_t bytes, hipStream_t st = nullptr, bool graph_visible = true)
{
  if(bytes <= b.cap) return TRT_OK;
  if(capturing(st))
    return fail(ctx, TRT_E_INVALID, "the ctx's scratch would have to grow (%zu -> %zu bytes) while the stream is being "
                "captured into a hipGraph: make one eager call with the same sizes first", b.cap, bytes);
  if(b.p)
  {
    if(graph_visible)
    {
      ctx->retired.push_back(b.p);
      if(bytes < b.cap + b.cap / 2) bytes = b.cap + b.cap / 2;
    }
    else
      TRT_HIP(ctx, hipFree(b.p));   // (synchronises with the device: nothing is using the block any more)
  }
  b.p = nullptr;
  b.cap = 0;
  TRT_HIP(ctx, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return TRT_OK;
}

template <class Real>
void torus_prepare(const trt_torus& t, TorusK<Real>& k)
{
  const Real R = (Real)t.R, r = (Real)t.r;
  const Real R2 = R * R, r2 = r * r, s = R + r, s2 = s * s;
  k.cx = (Real)t.center[0];
  k.cy = (Real)t.center[1];
  k.cz = (Real)t.center[2];
  k.R      = R;
  k.r2     = r2;
  k.rpol   = r * (Real)0.03125;
  k.k0     = R2 - r2;
  k.Rb2    = std::fma(s2, (Real)0.001953125, s2);
  k.rs     = std::fma(r, (Real)0.00390625, r);
  k.fourR2 = (Real)4 * R2;
}

int build_scene_uncached(trt_ctx* ctx, const trt_scene* s, SceneK& out);

// The validated kernel constants of `s` — from the ctx's cache when the caller passes the scene of the previous call
// again (compared byte for byte, solver included), else built and cached.
int build_scene(trt_ctx* ctx, const trt_scene* s, const SceneK*& out)
{
  if(!s || !s->tori || !s->materials)
    return fail(ctx, TRT_E_INVALID, "scene: NULL scene / tori / materials");
  if(s->n_tori < 1 || s->n_tori > TRT_MAX_TORI)
    return fail(ctx, TRT_E_SCENE, "scene: n_tori=%u outside 1..%d", s->n_tori, TRT_MAX_TORI);
  if(s->n_materials < 1 || s->n_materials > TRT_MAX_MATERIALS)
    return fail(ctx, TRT_E_SCENE, "scene: n_materials=%u outside 1..%d", s->n_materials,
                TRT_MAX_MATERIALS);
  auto& c = ctx->scene_cache;
  if(!(c.valid && c.precision == ctx->precision && c.n_tori == s->n_tori && c.n_mat == s->n_materials
       && !std::memcmp(c.tori, s->tori, s->n_tori * sizeof(trt_torus))
       && !std::memcmp(c.mat, s->materials, s->n_materials * sizeof(trt_material))))
  {
    c.valid = false;
    if(int rc = build_scene_uncached(ctx, s, c.K)) return rc;
    c.precision = ctx->precision;
    c.n_tori = s->n_tori;
    c.n_mat  = s->n_materials;
    std::memcpy(c.tori, s->tori, s->n_tori * sizeof(trt_torus));
    std::memcpy(c.mat, s->materials, s->n_materials * sizeof(trt_material));
    c.valid = true;
  }
  out = &c.K;
  return TRT_OK;
}

int build_scene_uncached(trt_ctx* ctx, const trt_scene* s, SceneK& out)
{
  std::memset(&out, 0, sizeof out);
  out.n_tori = (int)s->n_tori;
  out.n_mat  = (int)s->n_materials;
  out.f64    = ctx->precision == TRT_SOLVE_F64 || ctx->precision == TRT_SOLVE_DK_F64 || ctx->precision == TRT_SOLVE_FERRARI_F64;
  out.dk     = (ctx->precision == TRT_SOLVE_DK_F32 || ctx->precision == TRT_SOLVE_DK_F64) ? 1
               : (ctx->precision == TRT_SOLVE_FERRARI_F32 || ctx->precision == TRT_SOLVE_FERRARI_F64) ? 2 : 0;
  for(uint32_t i = 0; i < s->n_tori; ++i)
  {
    const trt_torus& t = s->tori[i];
    if(!(t.r > 0.0f && t.R > t.r))
      return fail(ctx, TRT_E_SCENE, "scene: torus %u is not a ring torus (R=%g, r=%g; need 0<r<R)", i,
                  (double)t.R, (double)t.r);
    if(t.matId < 0 || (uint32_t)t.matId >= s->n_materials)
      return fail(ctx, TRT_E_SCENE, "scene: torus %u has matId=%d outside 0..%u", i, t.matId,
                  s->n_materials - 1);
    torus_prepare<float>(t, out.k32[i]);
    torus_prepare<double>(t, out.k64[i]);
    out.shade[i] = {t.center[0], t.center[1], t.center[2], t.R, t.matId};
  }
  // test order: largest bounding sphere first (stable insertion sort on the FP32 sum R + r)
  for(uint32_t i = 0; i < s->n_tori; ++i)
  {
    const float key = s->tori[i].R + s->tori[i].r;
    uint32_t k = i;
    while(k > 0 && s->tori[out.order[k - 1]].R + s->tori[out.order[k - 1]].r < key)
    {
      out.order[k] = out.order[k - 1];
      --k;
    }
    out.order[k] = (int)i;
  }
  // Enclosure masks: tube k lies strictly inside tube j when both tori turn about the same axis line
  // (centre x and z equal) and every point of k's centre circle is closer to j's than r_j - r_k, with 2^-10 of r_j to
  // spare: the circles are sqrt(dR² + dy²) apart everywhere.  Plain double arithmetic on the scene's floats — the
  // tests' CPU checker takes the same decisions from the same lines.  Bit positions are TEST-ORDER positions.
  for(uint32_t j = 0; j < s->n_tori; ++j)
  {
    out.inside[j] = 0u;
    const trt_torus& J = s->tori[j];
    for(uint32_t p = 0; p < s->n_tori; ++p)
    {
      const uint32_t k = (uint32_t)out.order[p];
      const trt_torus& K = s->tori[k];
      if(k == j || K.center[0] != J.center[0] || K.center[2] != J.center[2]) continue;
      const double dR = (double)K.R - (double)J.R, dy = (double)K.center[1] - (double)J.center[1];
      const double D  = std::sqrt(dR * dR + dy * dy);
      if(D + (double)K.r < (double)J.r - (double)J.r * 0.0009765625 && !ctx->tn.no_enclosure) out.inside[j] |= 1u << p;
    }
  }
  for(uint32_t i = 0; i < s->n_materials; ++i)
  {
    const trt_material& m = s->materials[i];
    if(m.textureId >= 0)
      return fail(ctx, TRT_E_SCENE, "scene: material %u has textureId=%d; tori are untextured", i,
                  m.textureId);
    MaterialK& k = out.mat[i];
    std::memcpy(k.ambient, m.ambient, sizeof k.ambient);
    std::memcpy(k.diffuse, m.diffuse, sizeof k.diffuse);
    std::memcpy(k.specular, m.specular, sizeof k.specular);
    k.shininess = m.shininess;
    k.illum     = m.illum;
  }
  return TRT_OK;
}

// column-major mat4 · (0,0,0,1), rows 0..2 — same accumulation order as the kernels
void mat4_origin(const float* m, float out[3])
{
  for(int r = 0; r < 3; ++r)
    out[r] = std::fma(m[12 + r], 1.0f, std::fma(m[8 + r], 0.0f, std::fma(m[4 + r], 0.0f, m[r] * 0.0f)));
}

// Enclosure cull, the camera's share: the tori no primary ray of the frame can hit first — the tubes inside every tube j
// that all ray origins lie outside of.  The origins are the eye (pinhole) or lie `reach` = |rho| from it (toroidal camera,
// BEF rgen:56); certified when the eye's distance from j's centre circle exceeds r_j + reach with 2^-10 of each to spare.
// Double arithmetic on the same FP32 eye the kernels compute (mat4_origin); the oracle repeats it (shade setup).
uint32_t primary_skip_mask(const trt_scene* s, const SceneK& K, const float eye[3], float reach)
{
  uint32_t mask = 0u;
  for(uint32_t j = 0; j < s->n_tori; ++j)
  {
    if(K.inside[j] == 0u) continue;
    const trt_torus& T = s->tori[j];
    const double ex = (double)eye[0] - (double)T.center[0], ey = (double)eye[1] - (double)T.center[1], ez = (double)eye[2] - (double)T.center[2];
    const double rho = std::sqrt(ex * ex + ez * ez) - (double)T.R;
    const double d   = std::sqrt(rho * rho + ey * ey);
    const double rc  = std::fabs((double)reach);
    if(d > ((double)T.r + (double)T.r * 0.0009765625) + (rc + rc * 0.0009765625)) mask |= K.inside[j];
  }
  return mask;
}

constexpr float kDeg2Rad = 0.017453292519943295f;  // GLSL radians()
constexpr float kRad2Deg = 57.29577951308232f;     // GLSL degrees()

// Per-frame part of the toroidal camera (BEF/shaders/raytrace.rgen:36-53) and the
// per-column / per-row trigonometry of :25-28,56-57, evaluated once on the host.
int build_toro(trt_ctx* ctx, const trt_globals& g, const trt_push& pc, uint32_t W, uint32_t H,
               hipStream_t stream, ToroCam& out, bool must_match = false)
{
  float eye[3];
  mat4_origin(g.viewInverse, eye);                                           // :36
  float tx = g.center[0] - eye[0], ty, tz = g.center[2] - eye[2];            // :38
  float il    = 1.0f / std::sqrt(std::fma(tz, tz, tx * tx));                 // :39
  float omega = std::acos(tx * il) * kRad2Deg;                               // :40
  if(tz < 0.0f) omega = 360.0f - omega;                                      // :41-43
  float theta = 0.0f;
  if(eye[1] != g.center[1])                                                  // :45
  {
    const float w  = omega * kDeg2Rad;
    const float p0 = std::fma(pc.rho, std::cos(w), eye[0]);                  // :46
    tx = g.center[0] - p0;                                                   // :47
    ty = g.center[1] - eye[1];
    il = 1.0f / std::sqrt(std::fma(ty, ty, tx * tx));                        // :48
    theta = std::acos(tx * il) * kRad2Deg;                                   // :49
    if(ty < 0.0f) theta = 360.0f - theta;                                    // :50-52
  }
  const size_t n = 2 * ((size_t)W + H);
  if(int rc = grow(ctx, ctx->d_toro, n * sizeof(float), stream)) return rc;
  auto& key = ctx->toro_key;
  // bit-compare the angles so that a NaN frame (eye above centre) still caches
  const bool same = key.valid && key.W == W && key.H == H && !std::memcmp(&key.omega, &omega, 4)
                    && !std::memcmp(&key.theta, &theta, 4);
  if(!same && must_match)
    return fail(ctx, TRT_E_INVALID, "trt_render_batch: the frames of a batch must share the toroidal camera's trigonometry tables (same eye, centre and — with the eye off the centre's height — rho; the ctx holds "
                "one set of trigonometry tables); render these frames one by one");
  if(!same && capturing(stream))
    return fail(ctx, TRT_E_INVALID, "toroidal camera: the trigonometry tables of this (W, H, centre, rho) frame are not on the "
                "device yet and cannot be uploaded while the stream is being captured into a hipGraph (a replay would "
                "copy whatever the staging buffer holds later): render the frame eagerly once before capturing it");
  if(!same)
  {
    if(ctx->h_toro_cap < n)
    {
      if(ctx->h_toro) TRT_HIP(ctx, hipHostFree(ctx->h_toro));
      ctx->h_toro = nullptr;
      ctx->h_toro_cap = 0;
      TRT_HIP(ctx, hipHostMalloc((void**)&ctx->h_toro, n * sizeof(float), hipHostMallocDefault));
      ctx->h_toro_cap = n;
    }
    else if(ctx->ev_toro_set)
      TRT_HIP(ctx, hipEventSynchronize(ctx->ev_toro));  // an earlier upload may still be reading it
    float* ca = ctx->h_toro, *sa = ca + W, *cb = sa + W, *sb = cb + H;
    const float d_alfa = 360.0f / (float)W, d_beta = 360.0f / (float)H;      // :25-26
    for(uint32_t x = 0; x < W; ++x)
    {
      const float aw = (d_alfa * (float)x + omega) * kDeg2Rad;               // :27,56
      ca[x] = std::cos(aw);
      sa[x] = std::sin(aw);
    }
    for(uint32_t y = 0; y < H; ++y)
    {
      const float bt = (d_beta * (float)y + theta) * kDeg2Rad;               // :28,57
      cb[y] = std::cos(bt);
      sb[y] = std::sin(bt);
    }
    TRT_HIP(ctx, hipMemcpyAsync(ctx->d_toro.p, ctx->h_toro, n * sizeof(float), hipMemcpyHostToDevice,
                                stream));
    TRT_HIP(ctx, hipEventRecord(ctx->ev_toro, stream));
    ctx->ev_toro_set = true;
    key.W = W; key.H = H; key.omega = omega; key.theta = theta; key.valid = true;
  }
  out.eye[0] = eye[0]; out.eye[1] = eye[1]; out.eye[2] = eye[2];
  out.rho   = pc.rho;
  out.cos_a = (const float*)ctx->d_toro.p;
  out.sin_a = out.cos_a + W;
  out.cos_b = out.sin_a + W;
  out.sin_b = out.cos_b + H;
  return TRT_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------
extern "C" int trt_version(void) { return TRT_VERSION_MAJOR * 1000 + TRT_VERSION_MINOR; }

extern "C" const char* trt_last_error(const trt_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int trt_create(int device, trt_ctx** out)
{
  if(!out) return fail(nullptr, TRT_E_INVALID, "trt_create: out is NULL");
  *out = nullptr;
  int        n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess || n <= 0)
    return fail(nullptr, TRT_E_NO_DEVICE,
                "trt_create: no HIP device (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if(device < 0 || device >= n)
    return fail(nullptr, TRT_E_NO_DEVICE, "trt_create: device %d outside 0..%d", device, n - 1);
  trt_ctx* ctx = new(std::nothrow) trt_ctx;
  if(!ctx) return fail(nullptr, TRT_E_NOMEM, "trt_create: out of host memory");
  ctx->device = device;
  hipDeviceProp_t prop;
  if((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess
     || (e = hipMalloc((void**)&ctx->d_stats, 8 * sizeof(unsigned long long))) != hipSuccess
     || (e = hipMalloc((void**)&ctx->d_queue, 64 * sizeof(unsigned int))) != hipSuccess
     || (e = hipMemset(ctx->d_stats, 0, 8 * sizeof(unsigned long long))) != hipSuccess
     || (e = hipMemset(ctx->d_queue, 0, 64 * sizeof(unsigned int))) != hipSuccess
     || (e = hipEventCreateWithFlags(&ctx->ev_toro, hipEventDisableTiming)) != hipSuccess
     || (e = hipEventCreateWithFlags(&ctx->ev_stats, hipEventDisableTiming)) != hipSuccess)
  {
    fail(nullptr, TRT_E_HIP, "trt_create: %s", hipGetErrorString(e));
    trt_destroy(ctx);
    return TRT_E_HIP;
  }
  ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  ctx->tn    = tuning_from_env();
  *out = ctx;
  return TRT_OK;
}

extern "C" void trt_destroy(trt_ctx* ctx)
{
  if(!ctx) return;
  (void)hipSetDevice(ctx->device);
  if(ctx->ev_toro) (void)hipEventDestroy(ctx->ev_toro);
  if(ctx->ev_stats) (void)hipEventDestroy(ctx->ev_stats);
  if(ctx->d_stats) (void)hipFree(ctx->d_stats);
  if(ctx->d_queue) (void)hipFree(ctx->d_queue);
  if(ctx->h_toro) (void)hipHostFree(ctx->h_toro);
  DevBuf* all[] = {&ctx->d_toro, &ctx->d_rgba, &ctx->d_rendered, &ctx->d_tiles, &ctx->d_cost, &ctx->d_keys, &ctx->d_bins, &ctx->d_recs};
  for(DevBuf* b : all)
    if(b->p) (void)hipFree(b->p);
  for(DevBuf& b : ctx->d_in)
    if(b.p) (void)hipFree(b.p);
  for(DevBuf& b : ctx->d_out)
    if(b.p) (void)hipFree(b.p);
  for(void* q : ctx->retired) (void)hipFree(q);
  delete ctx;
}

#ifdef TRT_TUNING
// tools/ only (libtrt_tuning.so): read the TRT_* knobs again, so that one process can time A against B
extern "C" int trt_debug_reload_tuning(trt_ctx* ctx)
{
  if(!ctx) return TRT_E_INVALID;
  ctx->tn = tuning_from_env();
  ctx->scene_cache.valid = false;   // (TRT_NO_ENCLOSURE changes the scene constants)
  return TRT_OK;
}
#endif

#ifdef TRT_TIMELINE
// tools/timeline.py only: a device buffer of [waves of the listed kernel][8] uint64 the kernel stamps (nullptr = off)
namespace trt { hipError_t set_timeline(void* dev_ptr); }
extern "C" int trt_debug_set_timeline(trt_ctx* ctx, void* dev_ptr)
{
  if(!ctx) return TRT_E_INVALID;
  return trt::set_timeline(dev_ptr) == hipSuccess ? TRT_OK : TRT_E_HIP;
}
#endif

extern "C" int trt_set_solver(trt_ctx* ctx, int precision)
{
  if(!ctx) return TRT_E_INVALID;
  if(precision < TRT_SOLVE_F32 || precision > TRT_SOLVE_FERRARI_F64)
    return fail(ctx, TRT_E_INVALID, "trt_set_solver: %d is not one of the TRT_SOLVE_* constants", precision);
  ctx->precision = precision;
  return TRT_OK;
}

extern "C" int trt_set_render_variant(trt_ctx* ctx, const char* name)
{
  if(!ctx || !name) return TRT_E_INVALID;
  if(!std::strcmp(name, "static")) ctx->variant = kRenderStatic;
  else if(!std::strcmp(name, "persistent")) ctx->variant = kRenderPersistent;
  else if(!std::strcmp(name, "listed")) ctx->variant = kRenderListed;
  else return fail(ctx, TRT_E_INVALID, "trt_set_render_variant: unknown variant '%s'", name);
  return TRT_OK;
}

extern "C" int trt_set_classification(trt_ctx* ctx, int level)
{
  if(!ctx) return TRT_E_INVALID;
  if(level < TRT_CLASSIFY_AUTO || level > TRT_CLASSIFY_TILE)
    return fail(ctx, TRT_E_INVALID, "trt_set_classification: %d is not one of the TRT_CLASSIFY_* constants", level);
  ctx->classify = level;
  return TRT_OK;
}

extern "C" const char* trt_get_render_variant(const trt_ctx* ctx)
{
  if(!ctx) return "";
  return ctx->variant == kRenderPersistent ? "persistent" : ctx->variant == kRenderListed ? "listed" : "static";
}

extern "C" int trt_enable_stats(trt_ctx* ctx, int on)
{
  if(!ctx) return TRT_E_INVALID;
  ctx->stats_on = on != 0;
  return TRT_OK;
}

extern "C" int trt_get_stats(trt_ctx* ctx, trt_stats* out)
{
  if(!ctx || !out) return TRT_E_INVALID;
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  if(ctx->ev_stats_set) TRT_HIP(ctx, hipEventSynchronize(ctx->ev_stats));
  unsigned long long h[8];
  TRT_HIP(ctx, hipMemcpy(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost));
  out->primary_tests = h[0];
  out->bounce_tests  = h[1];
  out->shadow_tests  = h[2];
  out->pixels        = ctx->stats_pixels;
  out->traced_tests  = h[4];
  out->solved_tests  = h[5];
  out->evaluations   = h[6];
  out->reserved      = 0;
  return TRT_OK;
}

// ------------------------------------------------------------------------------------------
// trace
// ------------------------------------------------------------------------------------------
extern "C" int trt_trace_dev(trt_ctx* ctx, const trt_rays* in, const trt_scene* scene, float tmin,
                             float tmax, trt_hits* out, void* stream)
{
  if(!ctx) return TRT_E_INVALID;
  if(!in || !out) return fail(ctx, TRT_E_INVALID, "trt_trace: NULL rays or hits");
  if(in->n && (!in->ox || !in->oy || !in->oz || !in->dx || !in->dy || !in->dz))
    return fail(ctx, TRT_E_INVALID, "trt_trace: NULL ray stream");
  const SceneK* Sp = nullptr;
  if(int rc = build_scene(ctx, scene, Sp)) return rc;
  const SceneK& S = *Sp;
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  TraceArgs   a;
  a.rays  = *in;
  a.hits  = *out;
  a.tmin  = tmin;
  a.tmax  = tmax;
  a.stats = nullptr;
  if(ctx->stats_on)
  {
    TRT_HIP(ctx, launch_zero_words((unsigned int*)ctx->d_stats, 16, st));
    a.stats           = ctx->d_stats;
    ctx->stats_pixels = in->n;
  }
  TRT_HIP(ctx, launch_trace(S, a, ctx->tn, st));
  if(ctx->stats_on)
  {
    TRT_HIP(ctx, hipEventRecord(ctx->ev_stats, st));
    ctx->ev_stats_set = true;
  }
  return TRT_OK;
}

extern "C" int trt_trace(trt_ctx* ctx, const trt_rays* in, const trt_scene* scene, float tmin,
                         float tmax, trt_hits* out)
{
  if(!ctx) return TRT_E_INVALID;
  if(!in || !out) return fail(ctx, TRT_E_INVALID, "trt_trace: NULL rays or hits");
  if(in->n && (!in->ox || !in->oy || !in->oz || !in->dx || !in->dy || !in->dz))
    return fail(ctx, TRT_E_INVALID, "trt_trace: NULL ray stream");
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)in->n * sizeof(float);
  const float* src[6] = {in->ox, in->oy, in->oz, in->dx, in->dy, in->dz};
  void*        dst[8] = {out->t, out->px, out->py, out->pz, out->nx, out->ny, out->nz, out->id};
  trt_rays din = *in;
  trt_hits dout;
  const float** dptr_in[6] = {&din.ox, &din.oy, &din.oz, &din.dx, &din.dy, &din.dz};
  for(int k = 0; k < 6 && in->n; ++k)
  {
    if(int rc = grow(ctx, ctx->d_in[k], bytes, nullptr, false)) return rc;
    TRT_HIP(ctx, hipMemcpyAsync(ctx->d_in[k].p, src[k], bytes, hipMemcpyHostToDevice, nullptr));
    *dptr_in[k] = (const float*)ctx->d_in[k].p;
  }
  void** dptr_out[8] = {(void**)&dout.t,  (void**)&dout.px, (void**)&dout.py, (void**)&dout.pz,
                        (void**)&dout.nx, (void**)&dout.ny, (void**)&dout.nz, (void**)&dout.id};
  for(int k = 0; k < 8; ++k)
  {
    *dptr_out[k] = nullptr;
    if(dst[k] && in->n)
    {
      if(int rc = grow(ctx, ctx->d_out[k], bytes, nullptr, false)) return rc;
      *dptr_out[k] = ctx->d_out[k].p;
    }
  }
  if(int rc = trt_trace_dev(ctx, &din, scene, tmin, tmax, &dout, nullptr)) return rc;
  for(int k = 0; k < 8 && in->n; ++k)
    if(dst[k])
      TRT_HIP(ctx, hipMemcpyAsync(dst[k], ctx->d_out[k].p, bytes, hipMemcpyDeviceToHost, nullptr));
  TRT_HIP(ctx, hipStreamSynchronize(nullptr));
  return TRT_OK;
}

// ------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------
namespace {

uint32_t tiling_rows(const trt_tiling& t, uint32_t H)
{
  if(t.n_parts <= 1) return H;
  const uint32_t cycle = t.group_rows * t.n_parts;
  const uint32_t full  = H / cycle, rem = H % cycle, start = t.part * t.group_rows;
  uint32_t extra = 0;
  if(rem > start) extra = rem - start < t.group_rows ? rem - start : t.group_rows;
  return full * t.group_rows + extra;
}

// One frame (the trt_render*_dev entry points) or a batch of them (trt_render_batch_dev): validates, fills one RenderArgs
// per frame — the lists, their counters and capacities are shared by the frames of a batch — and launches.
int render_frames(trt_ctx* ctx, const trt_frame* frames, uint32_t n_frames, const trt_scene* scene,
                  uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, const trt_tiling* tiling,
                  int camera, trt_rendered_data* rendered, void* stream)
{
  if(!ctx) return TRT_E_INVALID;
  if(!frames || n_frames < 1 || n_frames > TRT_MAX_BATCH)
    return fail(ctx, TRT_E_INVALID, "trt_render_batch: %u frames (1..%d)", n_frames, TRT_MAX_BATCH);
  if(W == 0 || H == 0 || row_begin > row_end || row_end > H)
    return fail(ctx, TRT_E_INVALID, "trt_render: bad size/rows W=%u H=%u rows=[%u,%u)", W, H, row_begin, row_end);
  if((uint64_t)W * H > 0x7fffffffull)
    return fail(ctx, TRT_E_INVALID, "trt_render: W*H=%llu exceeds 2^31-1 pixels", (unsigned long long)W * H);
  if(camera != TRT_CAMERA_PINHOLE && camera != TRT_CAMERA_TOROIDAL)
    return fail(ctx, TRT_E_INVALID, "trt_render: unknown camera %d", camera);
  if(tiling && (tiling->group_rows == 0 || tiling->n_parts == 0 || tiling->part >= tiling->n_parts))
    return fail(ctx, TRT_E_INVALID, "trt_render_tiled: bad tiling group_rows=%u n_parts=%u part=%u",
                tiling->group_rows, tiling->n_parts, tiling->part);
  for(uint32_t f = 0; f < n_frames; ++f)
  {
    const trt_frame& fr = frames[f];
    if(!fr.g || !fr.pc) return fail(ctx, TRT_E_INVALID, "trt_render: NULL globals or push constants");
    if(((uintptr_t)fr.rgba_dev | (uintptr_t)rendered) & 15)
      return fail(ctx, TRT_E_INVALID, "trt_render: the rgba image and the RenderedData buffer must be 16-byte aligned (they are written as float4)");
    if(fr.first_hit_dev)
    {
      const trt_hits* h = fr.first_hit_dev;
      const void* hp[8] = {h->t, h->px, h->py, h->pz, h->nx, h->ny, h->nz, h->id};
      for(const void* q : hp)
        if((uintptr_t)q & 3)
          return fail(ctx, TRT_E_INVALID, "trt_render: first-hit streams must be 4-byte aligned");
    }
  }
  const bool batch = n_frames > 1;
  if(batch && (ctx->variant != kRenderListed || ctx->precision > TRT_SOLVE_F64))
    return fail(ctx, TRT_E_INVALID, "trt_render_batch: batches run the listed variant with the default solver (TRT_SOLVE_F32 / _F64) only; "
                                    "render these frames one by one");
  if(batch && W > 8u * 8191u)
    return fail(ctx, TRT_E_INVALID, "trt_render_batch: W <= 65528 (a batch's tile lists pack the tile column in 13 bits)");
  const SceneK* Sp = nullptr;
  if(int rc = build_scene(ctx, scene, Sp)) return rc;
  const SceneK& S = *Sp;
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  if(ctx->variant == kRenderPersistent && S.dk)
    return fail(ctx, TRT_E_INVALID, "trt_render: the persistent variant implements the default solver only; "
                                    "use the listed or static variant with TRT_SOLVE_DK_* / TRT_SOLVE_FERRARI_*");
  RenderBatch B;
  std::memset(&B, 0, sizeof B);
  B.n_frames = n_frames;
  uint32_t n_local_rows = row_end - row_begin;
  if(tiling) n_local_rows = tiling_rows(*tiling, H);
  const size_t n_tiles = (size_t)((W + 7) / 8) * ((n_local_rows + 7) / 8);              // per frame
  const size_t n_macro = (size_t)(((W + 7) / 8 + 3) / 4) * ((n_local_rows + 7) / 8);    // per frame
  const bool   lists   = ctx->variant != kRenderStatic;
  if(lists)
  {
    if(W > 8u * 65535u || n_local_rows > 8u * 32767u)
      return fail(ctx, TRT_E_INVALID, "trt_render: the tile lists pack tile coordinates in 16 + 15 bits (W <= 524280, rows <= 262136)");
    if(n_tiles * n_frames > 0x7fffffffull)
      return fail(ctx, TRT_E_INVALID, "trt_render_batch: %zu tiles in the batch exceed the tile lists", n_tiles * n_frames);
    if(int rc = grow(ctx, ctx->d_tiles, 2 * n_tiles * n_frames * sizeof(uint32_t), st)) return rc;
  }
  // cost feedback of the listed kernel (scheduling only): one word per macro tile (and frame of a batch), zero when the buffer is
  // new.  It pays where the cost of a tile varies much and the frame is bound by the tracing: eight nested tori −11 % (FP64) /
  // −15 % (FP32); a single torus' frame is bound by its stores and LOSES 2–3 % to the bookkeeping — scenes of one torus go without
  const bool cost_fb = lists && ctx->variant == kRenderListed && ctx->tn.heavy_x16 && (uint32_t)S.n_tori >= ctx->tn.heavy_min_tori;
  if(cost_fb && ctx->d_cost.cap < n_macro * n_frames * sizeof(uint32_t))
  {
    if(int rc = grow(ctx, ctx->d_cost, n_macro * n_frames * sizeof(uint32_t), st)) return rc;
    TRT_HIP(ctx, hipMemsetAsync(ctx->d_cost.p, 0, ctx->d_cost.cap, st));   // never inside a capture: grow() refuses there
  }
  uint32_t fine_any = 0;
  for(uint32_t f = 0; f < n_frames; ++f)
  {
    const trt_frame& fr = frames[f];
    RenderArgs& a = B.fr[f];
    a.g = *fr.g;
    a.pc = *fr.pc;
    a.W = W; a.H = H; a.row_begin = row_begin; a.row_end = row_end;
    a.tile_group = 1; a.tile_parts = 1; a.tile_part = 0; a.compact = 0;
    a.n_local_rows = n_local_rows;
    if(tiling)
    {
      a.row_begin = 0; a.row_end = H;
      a.tile_group = tiling->group_rows; a.tile_parts = tiling->n_parts; a.tile_part = tiling->part;
      a.compact = tiling->compact;
    }
    a.camera = camera;
    a.rgba = fr.rgba_dev;
    if(fr.first_hit_dev) a.hits = *fr.first_hit_dev;
    a.rendered = rendered;
    a.counters = ctx->d_queue;
    a.counts   = ctx->d_queue + 32;
    if(camera == TRT_CAMERA_TOROIDAL)
      if(int rc = build_toro(ctx, *fr.g, *fr.pc, W, H, st, a.toro, f > 0)) return rc;
    if(ctx->stats_on) a.stats = ctx->d_stats;
    {
      float eye[3];
      mat4_origin(fr.g->viewInverse, eye);
      a.skip_primary = primary_skip_mask(scene, S, eye, camera == TRT_CAMERA_TOROIDAL ? fr.pc->rho : 0.0f);
    }
    if(!lists) continue;
    a.tiles_live  = (uint32_t*)ctx->d_tiles.p;
    a.tiles_clear = a.tiles_live + n_tiles * n_frames;
    a.cap_live    = (uint32_t)(n_tiles * n_frames);
    a.cap_clear   = (uint32_t)(n_tiles * n_frames);
    if(cost_fb)
    {
      a.tile_cost = (uint32_t*)ctx->d_cost.p + (size_t)f * n_macro;
      a.heavy_x16 = ctx->tn.heavy_x16;
    }
    // tile culling needs tiles that are 8 contiguous image rows; with a RenderedData export the listed
    // kernel still writes the primary ray and the miss record of every pixel of a culled tile (raygen
    // only, no solve), the persistent kernel does not: it then traces every tile
    a.tile_cull = ((rendered == nullptr || ctx->variant == kRenderListed) && (a.tile_parts <= 1 || a.tile_group % 8 == 0)) ? 1u : 0u;
    a.min_batch = ctx->tn.min_batch;
    if(ctx->tn.no_tile_cull) a.tile_cull = 0;
    // The finer, per-tile classification costs 8 µs more at 4096² and pays when most macro tiles
    // touch a bounding volume: the toroidal camera looks in every direction from among the
    // geometry (inside a torus: 0.30 → 0.21 ms); for a pinhole camera outside the scene the
    // macro-level test already culls 85 % of the frame and the extra pass is a net loss (+3…11 %).
    a.fine = camera == TRT_CAMERA_TOROIDAL ? 1u : 0u;
    if(camera == TRT_CAMERA_PINHOLE)
    {
      // a pinhole camera INSIDE the scene (eye within two bounding radii of a torus) sees it the same way
      float eye[3];
      mat4_origin(fr.g->viewInverse, eye);
      for(uint32_t i = 0; i < scene->n_tori; ++i)
      {
        const trt_torus& t = scene->tori[i];
        const float dx = eye[0] - t.center[0], dy = eye[1] - t.center[1], dz = eye[2] - t.center[2];
        const float reach = 2.0f * (t.R + t.r);
        if(dx * dx + dy * dy + dz * dz < reach * reach) a.fine = 1u;
      }
    }
    if(ctx->classify != TRT_CLASSIFY_AUTO) a.fine = (uint32_t)ctx->classify;
    if(ctx->tn.fine >= 0) a.fine = (uint32_t)ctx->tn.fine;
    fine_any |= a.fine;
    a.debug_skip = ctx->tn.debug_skip;   // always 0 in the release build
    uintptr_t bits = 0;
    const void* hp[8] = {a.hits.t, a.hits.px, a.hits.py, a.hits.pz, a.hits.nx, a.hits.ny, a.hits.nz, a.hits.id};
    for(const void* q : hp) bits |= (uintptr_t)q;
    a.vec4_ok = (W % 4 == 0 && (bits & 15) == 0) ? 1u : 0u;
  }
  if(ctx->stats_on)
  {
    TRT_HIP(ctx, launch_zero_words((unsigned int*)ctx->d_stats, 16, st));
    ctx->stats_pixels = (uint64_t)n_local_rows * W * n_frames;
  }
  if(batch)
  {
    for(uint32_t f = 0; f < n_frames; ++f) B.fr[f].fine = fine_any;   // one classification kernel for the whole batch
    const uint64_t lanes = fine_any ? (uint64_t)n_macro * 4 : (uint64_t)n_macro;
    B.per_frame = (uint32_t)((lanes + 63) / 64 * 64);
    TRT_HIP(ctx, launch_render_batch(S, B, ctx->n_cus, ctx->tn, st));
  }
  else
    TRT_HIP(ctx, launch_render(S, B.fr[0], ctx->variant, ctx->n_cus, ctx->tn, st));
  if(ctx->stats_on && !capturing(st))
  {
    TRT_HIP(ctx, hipEventRecord(ctx->ev_stats, st));
    ctx->ev_stats_set = true;
  }
  if(lists && ctx->tn.debug_tiles)
  {
    unsigned int q[4];
    TRT_HIP(ctx, hipStreamSynchronize(st));
    TRT_HIP(ctx, hipMemcpy(q, B.fr[0].counts, sizeof q, hipMemcpyDeviceToHost));
    fprintf(stderr, "[trt] tiles: live=%u (heavy %u, mean cost %u ticks) clear=%u (cull=%u)\n", q[0], q[2], q[3], q[1], B.fr[0].tile_cull);
  }
  return TRT_OK;
}

int render_common(trt_ctx* ctx, const trt_globals* g, const trt_push* pc, const trt_scene* scene,
                  uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, const trt_tiling* tiling,
                  int camera, float* rgba, trt_hits* first_hit, trt_rendered_data* rendered, void* stream)
{
  if(!ctx) return TRT_E_INVALID;
  const trt_frame one = {g, pc, rgba, first_hit};
  return render_frames(ctx, &one, 1, scene, W, H, row_begin, row_end, tiling, camera, rendered, stream);
}

}  // namespace

extern "C" uint32_t trt_tiling_rows(const trt_tiling* tiling, uint32_t H)
{
  if(!tiling || tiling->group_rows == 0 || tiling->n_parts == 0 || tiling->part >= tiling->n_parts) return 0;
  return tiling_rows(*tiling, H);
}

extern "C" int trt_render_dev(trt_ctx* ctx, const trt_globals* g, const trt_push* pc,
                              const trt_scene* scene, uint32_t W, uint32_t H, uint32_t row_begin,
                              uint32_t row_end, int camera, float* rgba, trt_hits* first_hit,
                              trt_rendered_data* rendered, void* stream)
{
  return render_common(ctx, g, pc, scene, W, H, row_begin, row_end, nullptr, camera, rgba, first_hit,
                       rendered, stream);
}

extern "C" int trt_render_tiled_dev(trt_ctx* ctx, const trt_globals* g, const trt_push* pc,
                                    const trt_scene* scene, uint32_t W, uint32_t H, const trt_tiling* tiling,
                                    int camera, float* rgba, trt_hits* first_hit,
                                    trt_rendered_data* rendered, void* stream)
{
  if(ctx && !tiling) return fail(ctx, TRT_E_INVALID, "trt_render_tiled: NULL tiling");
  return render_common(ctx, g, pc, scene, W, H, 0, H, tiling, camera, rgba, first_hit, rendered, stream);
}

extern "C" int trt_render_batch_dev(trt_ctx* ctx, const trt_frame* frames, uint32_t n_frames, const trt_scene* scene,
                                    uint32_t W, uint32_t H, const trt_tiling* tiling, int camera, void* stream)
{
  return render_frames(ctx, frames, n_frames, scene, W, H, 0, H, tiling, camera, nullptr, stream);
}

extern "C" int trt_render(trt_ctx* ctx, const trt_globals* g, const trt_push* pc,
                          const trt_scene* scene, uint32_t W, uint32_t H, int camera, float* rgba_out,
                          trt_hits* first_hit_out)
{
  if(!ctx) return TRT_E_INVALID;
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t npx = (size_t)W * H;
  float*       d_rgba = nullptr;
  if(rgba_out)
  {
    if(int rc = grow(ctx, ctx->d_rgba, npx * 16, nullptr, false)) return rc;
    d_rgba = (float*)ctx->d_rgba.p;
  }
  trt_hits dh;
  std::memset(&dh, 0, sizeof dh);
  void*  dst[8]  = {nullptr};
  void** dptr[8] = {(void**)&dh.t,  (void**)&dh.px, (void**)&dh.py, (void**)&dh.pz,
                    (void**)&dh.nx, (void**)&dh.ny, (void**)&dh.nz, (void**)&dh.id};
  if(first_hit_out)
  {
    void* h[8] = {first_hit_out->t,  first_hit_out->px, first_hit_out->py, first_hit_out->pz,
                  first_hit_out->nx, first_hit_out->ny, first_hit_out->nz, first_hit_out->id};
    for(int k = 0; k < 8; ++k)
      if((dst[k] = h[k]))
      {
        if(int rc = grow(ctx, ctx->d_out[k], npx * 4, nullptr, false)) return rc;
        *dptr[k] = ctx->d_out[k].p;
      }
  }
  if(int rc = trt_render_dev(ctx, g, pc, scene, W, H, 0, H, camera, d_rgba, first_hit_out ? &dh : nullptr,
                             nullptr, nullptr))
    return rc;
  if(rgba_out) TRT_HIP(ctx, hipMemcpyAsync(rgba_out, d_rgba, npx * 16, hipMemcpyDeviceToHost, nullptr));
  for(int k = 0; k < 8; ++k)
    if(dst[k]) TRT_HIP(ctx, hipMemcpyAsync(dst[k], ctx->d_out[k].p, npx * 4, hipMemcpyDeviceToHost, nullptr));
  TRT_HIP(ctx, hipStreamSynchronize(nullptr));
  return TRT_OK;
}

// ------------------------------------------------------------------------------------------
// post pass
// ------------------------------------------------------------------------------------------
extern "C" int trt_post_dev(trt_ctx* ctx, const float* rgba_in, uint64_t n_pixels, float* f32_out,
                            uint8_t* unorm8_out, void* stream)
{
  if(!ctx) return TRT_E_INVALID;
  if(n_pixels && !rgba_in) return fail(ctx, TRT_E_INVALID, "trt_post: NULL input image");
  if(((uintptr_t)rgba_in | (uintptr_t)f32_out) & 15 || ((uintptr_t)unorm8_out & 3))
    return fail(ctx, TRT_E_INVALID, "trt_post: images must be 16-byte (float) / 4-byte (unorm8) aligned");
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  TRT_HIP(ctx, launch_post(rgba_in, n_pixels, f32_out, unorm8_out, ctx->n_cus, ctx->tn, (hipStream_t)stream));
  return TRT_OK;
}

// ------------------------------------------------------------------------------------------
// point-cloud re-projection
// ------------------------------------------------------------------------------------------
extern "C" int trt_splat_dev(trt_ctx* ctx, const trt_point* points, uint64_t n_points, const float* viewProj,
                             uint32_t W, uint32_t H, const float* clearColor, float point_size, float* rgba,
                             void* stream)
{
  if(!ctx) return TRT_E_INVALID;
  if(!viewProj || !clearColor || !rgba || (n_points && !points))
    return fail(ctx, TRT_E_INVALID, "trt_splat: NULL argument");
  if(W == 0 || H == 0 || (uint64_t)W * H > 0x7fffffffull || n_points > 0xffffffffull)
    return fail(ctx, TRT_E_INVALID, "trt_splat: bad sizes W=%u H=%u n_points=%llu", W, H, (unsigned long long)n_points);
  if(!(point_size > 0.0f && point_size <= 64.0f))
    return fail(ctx, TRT_E_INVALID, "trt_splat: point_size %g outside (0, 64]", (double)point_size);
  if(((uintptr_t)points | (uintptr_t)rgba) & 15)
    return fail(ctx, TRT_E_INVALID, "trt_splat: the point buffer and the rgba image must be 16-byte aligned (float4 accesses)");
  TRT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t  st = (hipStream_t)stream;
  SplatScratch sc{};
  // The binned forms reserve room for the worst case, four 12-B records per point (a point on a bin corner): the paged form
  // two pages per bin + a pool for 4n records (8.4 M points, 512 bins: 0.8 GB), the two-pass forms 56 B per point —
  // against 8 B per PIXEL for the one-pass form, which splat_plan picks beyond 8 GiB of records (its scratch does not grow
  // with the cloud).
  sc.plan = splat_plan(W, H, point_size, n_points, ctx->tn);
  if(sc.plan.mode != kSplatOnePass)
  {
    // sized for the largest bin count once: the count, state, pool and ticket words must be zero between calls — the
    // kernels leave them so; a call that failed in between marks them dirty
    if(ctx->d_bins.cap < kSplatBinWords * sizeof(uint32_t))
    {
      if(int rc = grow(ctx, ctx->d_bins, kSplatBinWords * sizeof(uint32_t), st)) return rc;
      ctx->bins_dirty = true;
    }
    if(ctx->bins_dirty)
    {
      TRT_HIP(ctx, launch_zero_words((unsigned int*)ctx->d_bins.p, (uint32_t)kSplatBinWords, st));   // a kernel node when captured
      ctx->bins_dirty = false;
    }
    if(int rc = grow(ctx, ctx->d_recs, sc.plan.total(), st)) return rc;
    sc.bin_words = (uint32_t*)ctx->d_bins.p;
    sc.records   = ctx->d_recs.p;
  }
  else
  {
    if(int rc = grow(ctx, ctx->d_keys, (size_t)W * H * sizeof(unsigned long long), st)) return rc;
    sc.keys = (unsigned long long*)ctx->d_keys.p;
  }
  if(sc.plan.mode != kSplatOnePass) ctx->bins_dirty = true;   // until every kernel of the call is enqueued
  TRT_HIP(ctx, launch_splat(points, n_points, viewProj, W, H, clearColor, point_size, sc, rgba, ctx->n_cus, ctx->tn, st));
  ctx->bins_dirty = false;
  return TRT_OK;
}
