// trt_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the toroidal ray tracer.
//
//   trace_kernel              trace(rays_in → hits_out): SoA rays in, closest hit out.
//   render_static_kernel      one lane per pixel, 8×8 pixel tile per wavefront; each lane runs the
//                             reference's raygen bounce loop (REFL/shaders/raytrace.rgen:62-85).
//   render_persistent_kernel  persistent wavefronts over a global work queue: the bounce
//                             loop is flattened into per-lane queries (closest-hit, shadow,
//                             bounce); a lane whose pixel is finished is refilled at once
//                             (ballot + popcount compaction, one atomic per 256-pixel
//                             chunk per wave), so every trip of the solve loop works on 64
//                             live ray–torus tests.
//
// One lane = one ray.  Scene constants are staged into LDS once per block.  No MFMA: the
// work is scalar FP32/FP64 root finding.  Compiled with -ffp-contract=off (see
// trt_device.hpp for the arithmetic contract).  Every kernel is instantiated for the FP32
// and the FP64 root solve (BASELINE config 4); I/O is FP32 in both.
#include "trt_kernels.hpp"

namespace trt {

// ------------------------------------------------------------------------------------------
// closest-hit shader body, split at the shadow query (REFL/shaders/raytrace.rchit:50-156)
// ------------------------------------------------------------------------------------------
struct HitState {
  v3    P, N, L;
  v3    diffuse;
  float lightIntensity, lightDistance;
  int   matId;
  bool  wantShadow;  // dot(N,L) > 0  (rchit:112)
};

__device__ __forceinline__ void hit_begin(const SceneK& S, const trt_push& pc, int id, float t, v3 o,
                                          v3 d, HitState& h)
{
  h.matId = S.shade[id].matId;                                               // rchit:95-96
  h.P     = {fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)};       // BEF rchit:134
  h.N     = torus_normal(S.shade[id], h.P);
  const v3 lp = {pc.lightPosition[0], pc.lightPosition[1], pc.lightPosition[2]};
  h.lightIntensity = pc.lightIntensity;                                      // rchit:79
  h.lightDistance  = 100000.0f;                                              // rchit:80
  if(pc.lightType == 0)                                                      // rchit:82
  {
    const v3 lDir    = sub3(lp, h.P);
    h.lightDistance  = sqrt_(dot3(lDir, lDir));
    h.lightIntensity = pc.lightIntensity / (h.lightDistance * h.lightDistance);
    h.L              = scale3(lDir, 1.0f / h.lightDistance);
  }
  else
    h.L = normalize3(lp);                                                    // rchit:91
  h.diffuse    = compute_diffuse(S.mat[h.matId], h.L, h.N);                  // rchit:100
  h.wantShadow = dot3(h.N, h.L) > 0.0f;                                      // rchit:112
}

// Finishes the closest-hit shader once the shadow query is answered; returns prd.hitValue and
// updates the payload (attenuation, done, next ray) exactly as rchit:133-155.
__device__ __forceinline__ v3 hit_end(const SceneK& S, const HitState& h, v3 d, bool shadowed,
                                      v3& attenuation, int& done, v3& nextO, v3& nextD)
{
  const MaterialK& mat = S.mat[h.matId];
  v3    specular     = {0.0f, 0.0f, 0.0f};
  float attenuation1 = 1.0f;
  if(h.wantShadow)
  {
    if(shadowed) attenuation1 = 0.3f;                                        // rchit:135
    else specular = compute_specular(mat, d, h.L, h.N);                      // rchit:140
  }
  if(mat.illum == 3)                                                         // rchit:145
  {
    attenuation.x *= mat.specular[0];
    attenuation.y *= mat.specular[1];
    attenuation.z *= mat.specular[2];
    done  = 0;
    nextO = h.P;
    nextD = reflect3(d, h.N);
  }
  const float k = attenuation1 * h.lightIntensity;                           // rchit:155
  return {k * (h.diffuse.x + specular.x), k * (h.diffuse.y + specular.y),
          k * (h.diffuse.z + specular.z)};
}

// ------------------------------------------------------------------------------------------
// pixel addressing: local rows (row band, or interleaved row groups of a multi-GPU tiling)
// ------------------------------------------------------------------------------------------
// local row ly of this launch → image row y
__device__ __forceinline__ uint32_t image_row(const RenderArgs& a, uint32_t ly)
{
  if(a.tile_parts <= 1)
    return a.row_begin + ly;
  return ((ly / a.tile_group) * a.tile_parts + a.tile_part) * a.tile_group + ly % a.tile_group;
}
// index of pixel (x, row) in the rgba / first-hit streams
__device__ __forceinline__ size_t out_index(const RenderArgs& a, uint32_t x, uint32_t y, uint32_t ly)
{
  return (size_t)(a.compact ? ly : y) * a.W + x;
}

__device__ __forceinline__ void store_first_hit(const RenderArgs& a, size_t i, float t, v3 P, v3 N, int id)
{
  if(a.hits.t) a.hits.t[i] = t;
  if(a.hits.px) a.hits.px[i] = P.x;
  if(a.hits.py) a.hits.py[i] = P.y;
  if(a.hits.pz) a.hits.pz[i] = P.z;
  if(a.hits.nx) a.hits.nx[i] = N.x;
  if(a.hits.ny) a.hits.ny[i] = N.y;
  if(a.hits.nz) a.hits.nz[i] = N.z;
  if(a.hits.id) a.hits.id[i] = id;
}

// wave-level sum of a 32-bit counter, then one atomic per wave
__device__ __forceinline__ void wave_add(unsigned long long* dst, uint32_t v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if((threadIdx.x & 63) == 0 && v)
    atomicAdd(dst, (unsigned long long)v);
}

constexpr float kTMin = 0.001f;    // rgen:51, rchit:114
constexpr float kTMax = 10000.0f;  // rgen:52

// ------------------------------------------------------------------------------------------
// trace(rays_in → hits_out)
// ------------------------------------------------------------------------------------------
template <class Real>
__global__ __launch_bounds__(256) void trace_kernel(const SceneK scene, const TraceArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  uint32_t       tests  = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.rays.n; i += stride)
  {
    const v3 o = {a.rays.ox[i], a.rays.oy[i], a.rays.oz[i]};
    const v3 d = {a.rays.dx[i], a.rays.dy[i], a.rays.dz[i]};
    float     t;
    const int id = closest_hit<Real>(S, o, d, a.tmin, a.tmax, t, tests);
    v3 P = {0.0f, 0.0f, 0.0f}, N = {0.0f, 0.0f, 0.0f};
    if(id >= 0)
    {
      P = {fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)};
      N = torus_normal(S.shade[id], P);
    }
    if(a.hits.t) a.hits.t[i] = t;
    if(a.hits.px) a.hits.px[i] = P.x;
    if(a.hits.py) a.hits.py[i] = P.y;
    if(a.hits.pz) a.hits.pz[i] = P.z;
    if(a.hits.nx) a.hits.nx[i] = N.x;
    if(a.hits.ny) a.hits.ny[i] = N.y;
    if(a.hits.nz) a.hits.nz[i] = N.z;
    if(a.hits.id) a.hits.id[i] = id;
  }
  if(a.stats)
    wave_add(&a.stats[0], tests);
}

// ------------------------------------------------------------------------------------------
// render, static mapping: lane ↔ pixel for the whole bounce loop
// ------------------------------------------------------------------------------------------
template <class Real>
__global__ __launch_bounds__(256) void render_static_kernel(const SceneK scene, const RenderArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  // 8×8 pixel tile per wavefront: neighbouring lanes trace neighbouring rays
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tiles_x = (a.W + 7) >> 3;
  const uint32_t tile    = blockIdx.x * (blockDim.x >> 6) + wave;
  const uint32_t x  = (tile % tiles_x) * 8 + (lane & 7);
  const uint32_t ly = (tile / tiles_x) * 8 + (lane >> 3);
  uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0;

  if(x < a.W && ly < a.n_local_rows)
  {
    const uint32_t y = image_row(a, ly);
    const size_t   oi = out_index(a, x, y, ly);
    v3 origin, direction;
    raygen(a.g, a.toro, a.W, a.H, a.camera, x, y, origin, direction);
    float4* rd = a.rendered ? reinterpret_cast<float4*>(&a.rendered[(size_t)x * a.H + y]) : nullptr;  // BEF rgen:72
    if(rd)
    {
      rd[2] = make_float4(origin.x, origin.y, origin.z, 1.0f);               // BEF rgen:56,72
      rd[3] = make_float4(direction.x, direction.y, direction.z, 0.0f);      // BEF rgen:57,73
    }

    int depth = 0, done = 1;                                                 // rgen:54,57
    v3  attenuation = {1.0f, 1.0f, 1.0f};                                    // rgen:56
    v3  hitValue    = {0.0f, 0.0f, 0.0f};                                    // rgen:61
    for(;;)                                                                  // rgen:62
    {
      v3    prdHit, nextO = origin, nextD = direction;
      float t;
      const int id = closest_hit<Real>(S, origin, direction, kTMin, kTMax, t, depth == 0 ? n_primary : n_bounce);
      if(id < 0)
      {
        prdHit = {a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f};  // rmiss:37
        if(depth == 0)
        {
          store_first_hit(a, oi, t, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, -1);  // BEF rmiss:21
          if(rd) rd[0] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        }
      }
      else
      {
        HitState h;
        hit_begin(S, a.pc, id, t, origin, direction, h);
        if(depth == 0)                                                       // BEF rgen:94-97
        {
          store_first_hit(a, oi, t, h.P, h.N, id);
          if(rd) rd[0] = make_float4(h.P.x, h.P.y, h.P.z, 1.0f);             // BEF rgen:112
        }
        bool shadowed = false;
        if(h.wantShadow)
          shadowed = any_hit<Real>(S, h.P, h.L, kTMin, h.lightDistance, n_shadow);  // rchit:114-131
        prdHit = hit_end(S, h, direction, shadowed, attenuation, done, nextO, nextD);
      }
      hitValue.x = fma_(prdHit.x, attenuation.x, hitValue.x);                // rgen:76
      hitValue.y = fma_(prdHit.y, attenuation.y, hitValue.y);
      hitValue.z = fma_(prdHit.z, attenuation.z, hitValue.z);
      depth++;                                                               // rgen:78
      if(done == 1 || depth >= a.pc.maxDepth)                                // rgen:79
        break;
      origin    = nextO;                                                     // rgen:82
      direction = nextD;                                                     // rgen:83
      done      = 1;                                                         // rgen:84
    }
    const float4 c = make_float4(hitValue.x, hitValue.y, hitValue.z, 1.0f);
    if(a.rgba) reinterpret_cast<float4*>(a.rgba)[oi] = c;                    // rgen:87
    if(rd) rd[1] = c;                                                        // BEF rgen:111
  }
  if(a.stats)
  {
    wave_add(&a.stats[0], n_primary);
    wave_add(&a.stats[1], n_bounce);
    wave_add(&a.stats[2], n_shadow);
  }
}

// ------------------------------------------------------------------------------------------
// render, persistent wavefronts + work queue
// ------------------------------------------------------------------------------------------
// The reference's raygen loop (rgen:62-85) calls traceRayEXT, whose closest-hit shader calls
// traceRayEXT again for the shadow ray (rchit:120-131): per pixel a data-dependent chain of
// 1..2·maxDepth queries, each a loop over the tori.  Here that recursion is flattened: a lane
// owns one *query* at a time (closest-hit or shadow) and inside it one ray–torus *test*
// (a TorusTest state machine).  Each trip of the outer loop
//   (A) advances every lane until it has a live test: finished queries run their shader
//       stage (miss / closest-hit / shadow-miss) and spawn the next query or finish the
//       pixel; idle lanes are compacted with a ballot and refilled from the wave's chunk of
//       the global pixel queue; tests culled by the bounding sphere are skipped at once;
//   (B) runs the solve loop — every lane evaluates (f, f') of ITS test, whatever pixel,
//       depth or query kind it belongs to;
//   (C) folds the finished tests into their queries.
enum : int { K_NONE = 0, K_CLOSEST = 1, K_SHADOW = 2 };
constexpr uint32_t kChunk = 256;  // pixels per queue grab: 4 horizontally adjacent 8×8 tiles

template <class Real>
__global__ __launch_bounds__(256) void render_persistent_kernel(const SceneK scene, const RenderArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  const uint32_t lane    = threadIdx.x & 63;
  const uint32_t tiles_x = (a.W + 7) >> 3;
  const uint32_t tiles_y = (a.n_local_rows + 7) >> 3;
  const uint32_t P       = tiles_x * tiles_y * 64;  // pixel slots incl. the ragged border
  const int      n_tori  = S.n_tori;

  // wave-uniform queue state
  uint32_t chunk_next = 0, chunk_end = 0;
  bool     exhausted  = false;

  // lane state: pixel payload (rgen:54-61)
  uint32_t px = 0, py = 0;       // pixel: x and image row
  size_t   oi = 0;               // index into rgba / first-hit streams
  int      depth = 0, done = 1;
  v3       attenuation = {1.0f, 1.0f, 1.0f}, hitValue = {0.0f, 0.0f, 0.0f};
  v3       dir_in = {0.0f, 0.0f, 0.0f};  // direction of the ray whose closest hit is being shaded
  // lane state: current query
  int   kind = K_NONE, ti = 0, best_id = -1;
  float best_t = 0.0f, q_tmax = 0.0f;
  bool  shadow_hit = false;
  v3    qo = {0.0f, 0.0f, 0.0f}, qd = {0.0f, 0.0f, 0.0f};  // query ray (FP32)
  RayK<Real> rk;                 // the same ray in solver precision, with dd and 1/dd
  // lane state: closest-hit shader between hit_begin and hit_end (the shadow query's origin
  // and direction are h.P and h.L, carried in qo/qd)
  v3    hN = {0.0f, 0.0f, 0.0f}, hDiffuse = {0.0f, 0.0f, 0.0f};
  float hLightI = 0.0f;
  int   hMat = 0;
  // lane state: current test
  TorusTest<Real> tst;
  tst.mode = M_DONE;
  tst.found = false;
  bool inflight = false, unconsumed = false;
  uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0;

  for(;;)
  {
    // ------------------------------ (A) advance -----------------------------------------
    for(;;)
    {
      const bool needs = !inflight && !(kind == K_NONE && exhausted);
      if(!__any(needs))
        break;

      // A1: shader stages of finished queries
      if(needs && kind != K_NONE && (ti >= n_tori || shadow_hit))
      {
        bool have_prd = false, shadowed = false, do_end = false;
        v3   prdHit = {0.0f, 0.0f, 0.0f};
        if(kind == K_CLOSEST)
        {
          float4* rd = a.rendered ? reinterpret_cast<float4*>(&a.rendered[(size_t)px * a.H + py]) : nullptr;
          if(best_id < 0)
          {
            // miss shader (REFL rmiss:37; BEF rmiss:21 hitPosition = 0)
            prdHit   = {a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f};
            have_prd = true;
            if(depth == 0)
            {
              store_first_hit(a, oi, __builtin_inff(), {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, -1);
              if(rd) rd[0] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            }
          }
          else
          {
            HitState h;
            hit_begin(S, a.pc, best_id, best_t, qo, qd, h);
            if(depth == 0)                                                   // BEF rgen:94-97
            {
              store_first_hit(a, oi, best_t, h.P, h.N, best_id);
              if(rd) rd[0] = make_float4(h.P.x, h.P.y, h.P.z, 1.0f);
            }
            dir_in = qd;
            hN = h.N; hDiffuse = h.diffuse; hLightI = h.lightIntensity; hMat = h.matId;
            qo = h.P; qd = h.L; q_tmax = h.lightDistance;
            if(h.wantShadow)
            {
              // shadow query (rchit:114-131): any hit in (0.001, lightDistance)
              kind = K_SHADOW; ti = 0; shadow_hit = false;
              rk.set(qo, qd, kTMin, q_tmax);
            }
            else
              do_end = true;
          }
        }
        else
        {
          do_end   = true;
          shadowed = shadow_hit;
        }
        if(do_end)
        {
          HitState h;
          h.P = qo; h.N = hN; h.L = qd; h.diffuse = hDiffuse;
          h.lightIntensity = hLightI; h.lightDistance = q_tmax; h.matId = hMat;
          h.wantShadow = kind == K_SHADOW;
          v3 nextO = qo, nextD = dir_in;
          prdHit   = hit_end(S, h, dir_in, shadowed, attenuation, done, nextO, nextD);
          have_prd = true;
          qo = nextO; qd = nextD;  // the reflected ray, used only if the loop continues
        }
        if(have_prd)
        {
          hitValue.x = fma_(prdHit.x, attenuation.x, hitValue.x);            // rgen:76
          hitValue.y = fma_(prdHit.y, attenuation.y, hitValue.y);
          hitValue.z = fma_(prdHit.z, attenuation.z, hitValue.z);
          depth++;                                                           // rgen:78
          if(done == 1 || depth >= a.pc.maxDepth)                            // rgen:79
          {
            const float4 c = make_float4(hitValue.x, hitValue.y, hitValue.z, 1.0f);
            if(a.rgba) reinterpret_cast<float4*>(a.rgba)[oi] = c;            // rgen:87
            if(a.rendered) reinterpret_cast<float4*>(&a.rendered[(size_t)px * a.H + py])[1] = c;
            kind = K_NONE;
          }
          else
          {
            done = 1;                                                        // rgen:84
            kind = K_CLOSEST; ti = 0; best_id = -1; best_t = __builtin_inff(); shadow_hit = false;
            q_tmax = kTMax;
            rk.set(qo, qd, kTMin, kTMax);                                    // rgen:82-83
          }
        }
      }

      // A2: compaction — refill idle lanes from the wave's chunk of the global queue
      for(;;)
      {
        const unsigned long long want = __ballot(kind == K_NONE && !exhausted);
        if(want == 0)
          break;
        if(chunk_next == chunk_end)
        {
          uint32_t base = 0;
          if(lane == 0)
            base = atomicAdd(a.queue, kChunk);
          base = __builtin_amdgcn_readfirstlane(base);
          if(base >= P) { exhausted = true; break; }
          chunk_next = base;
          chunk_end  = base + kChunk < P ? base + kChunk : P;
        }
        const uint32_t avail = chunk_end - chunk_next;
        const uint32_t rank  = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
        const uint32_t nwant = (uint32_t)__popcll(want);
        if(kind == K_NONE && rank < avail)
        {
          const uint32_t p = chunk_next + rank, tile = p >> 6, within = p & 63;
          const uint32_t x = (tile % tiles_x) * 8 + (within & 7), ly = (tile / tiles_x) * 8 + (within >> 3);
          if(x < a.W && ly < a.n_local_rows)
          {
            px = x;
            py = image_row(a, ly);
            oi = out_index(a, x, py, ly);
            raygen(a.g, a.toro, a.W, a.H, a.camera, px, py, qo, qd);
            if(a.rendered)
            {
              float4* rd = reinterpret_cast<float4*>(&a.rendered[(size_t)px * a.H + py]);
              rd[2] = make_float4(qo.x, qo.y, qo.z, 1.0f);
              rd[3] = make_float4(qd.x, qd.y, qd.z, 0.0f);
            }
            depth = 0; done = 1;
            attenuation = {1.0f, 1.0f, 1.0f};
            hitValue    = {0.0f, 0.0f, 0.0f};
            kind = K_CLOSEST; ti = 0; best_id = -1; best_t = __builtin_inff(); shadow_hit = false;
            q_tmax = kTMax;
            rk.set(qo, qd, kTMin, kTMax);
          }
        }
        chunk_next += nwant < avail ? nwant : avail;
      }

      // A3: set up the next test of every lane that has a query but no test
      if(!inflight && kind != K_NONE && ti < n_tori && !shadow_hit)
      {
        if(kind == K_SHADOW) ++n_shadow;
        else if(depth == 0) ++n_primary;
        else ++n_bounce;
        if(tst.setup(rk.ox, rk.oy, rk.oz, rk.dx, rk.dy, rk.dz, rk.dd, rk.inv_dd, rk.tmin, rk.tmax,
                     torus_k<Real>(S, ti)))
          inflight = true;
        else
          ++ti;  // culled by the bounding sphere / window: this test is a miss
      }
    }
    if(!__any(inflight))
      break;  // queue drained and every pixel finished

    // ------------------------------ (B) solve ---------------------------------------------
    while(__any(inflight))
    {
      if(inflight)
      {
        inflight   = tst.step();
        unconsumed = !inflight;
      }
    }

    // ------------------------------ (C) consume -------------------------------------------
    if(unconsumed)
    {
      unconsumed = false;
      Real  tt;
      float t;
      if(tst.finish(rk.dx, rk.dy, rk.dz, rk.tmin, rk.tmax, torus_k<Real>(S, ti), tt)
         && round_t(tt, kTMin, q_tmax, t))
      {
        if(kind == K_SHADOW) shadow_hit = true;
        else if(t < best_t) { best_t = t; best_id = ti; }
      }
      ++ti;
    }
  }

  if(a.stats)
  {
    wave_add(&a.stats[0], n_primary);
    wave_add(&a.stats[1], n_bounce);
    wave_add(&a.stats[2], n_shadow);
  }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
hipError_t launch_trace(const SceneK& scene, const TraceArgs& a, hipStream_t stream)
{
  if(a.rays.n == 0)
    return hipSuccess;
  const uint64_t want = (a.rays.n + 255) / 256;
  const uint32_t grid = (uint32_t)(want < 256u * 16u ? want : 256u * 16u);
  if(scene.f64)
    hipLaunchKernelGGL(trace_kernel<double>, dim3(grid), dim3(256), 0, stream, scene, a);
  else
    hipLaunchKernelGGL(trace_kernel<float>, dim3(grid), dim3(256), 0, stream, scene, a);
  return hipGetLastError();
}

hipError_t launch_render(const SceneK& scene, const RenderArgs& a, RenderVariant v, int n_cus,
                         hipStream_t stream)
{
  if(a.n_local_rows == 0 || a.W == 0)
    return hipSuccess;
  const uint64_t tiles = (uint64_t)((a.W + 7) / 8) * ((a.n_local_rows + 7) / 8);
  if(v == kRenderPersistent)
  {
    // resident grid: every CU gets kBlocksPerCU blocks of 4 waves; never more blocks than chunks
    const uint64_t chunks = (tiles * 64 + kChunk - 1) / kChunk;
    const uint64_t cap    = (uint64_t)n_cus * kPersistentBlocksPerCU;
    const uint32_t grid   = (uint32_t)((chunks + 3) / 4 < cap ? (chunks + 3) / 4 : cap);
    if(scene.f64)
      hipLaunchKernelGGL(render_persistent_kernel<double>, dim3(grid), dim3(256), 0, stream, scene, a);
    else
      hipLaunchKernelGGL(render_persistent_kernel<float>, dim3(grid), dim3(256), 0, stream, scene, a);
    return hipGetLastError();
  }
  const uint32_t grid = (uint32_t)((tiles + 3) / 4);
  if(scene.f64)
    hipLaunchKernelGGL(render_static_kernel<double>, dim3(grid), dim3(256), 0, stream, scene, a);
  else
    hipLaunchKernelGGL(render_static_kernel<float>, dim3(grid), dim3(256), 0, stream, scene, a);
  return hipGetLastError();
}

}  // namespace trt
