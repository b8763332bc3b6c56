// trt_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the toroidal ray tracer.
//
//   trace_kernel              trace(rays_in → hits_out): SoA rays in, closest hit out.
//   render_static_kernel      one lane per pixel, 8×8 pixel tile per wavefront; each lane runs the
//                             reference's raygen bounce loop (REFL/shaders/raytrace.rgen:62-85).
//   render_persistent_kernel  persistent wavefronts + global work queue: a lane whose pixel
//                             is finished is refilled at once (ballot + popcount compaction,
//                             one atomic per wave), and the closest-hit / shadow / bounce
//                             queries of different pixels share one convergent solve loop.
//
// One lane = one ray.  Scene constants are staged into LDS once per block.  No MFMA: the
// work is scalar FP32/FP64 root finding.  Compiled with -ffp-contract=off (see
// trt_device.hpp for the arithmetic contract).
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

struct PixelResult {
  v3    color;
  float t0;
  v3    P0, N0;
  int   id0;
  v3    rayO, rayD;
};

__device__ __forceinline__ void store_pixel(const RenderArgs& a, uint32_t x, uint32_t y,
                                            const PixelResult& r)
{
  const size_t i = (size_t)y * a.W + x;
  if(a.rgba)
    reinterpret_cast<float4*>(a.rgba)[i] = make_float4(r.color.x, r.color.y, r.color.z, 1.0f);  // rgen:87
  if(a.hits.t) a.hits.t[i] = r.t0;
  if(a.hits.px) a.hits.px[i] = r.P0.x;
  if(a.hits.py) a.hits.py[i] = r.P0.y;
  if(a.hits.pz) a.hits.pz[i] = r.P0.z;
  if(a.hits.nx) a.hits.nx[i] = r.N0.x;
  if(a.hits.ny) a.hits.ny[i] = r.N0.y;
  if(a.hits.nz) a.hits.nz[i] = r.N0.z;
  if(a.hits.id) a.hits.id[i] = r.id0;
  if(a.rendered)
  {
    float4* rd = reinterpret_cast<float4*>(&a.rendered[(size_t)x * a.H + y]);  // BEF rgen:72
    rd[0] = make_float4(r.P0.x, r.P0.y, r.P0.z, 1.0f);                         // BEF rgen:112
    rd[1] = make_float4(r.color.x, r.color.y, r.color.z, 1.0f);                // BEF rgen:111
    rd[2] = make_float4(r.rayO.x, r.rayO.y, r.rayO.z, 1.0f);                   // BEF rgen:56,72
    rd[3] = make_float4(r.rayD.x, r.rayD.y, r.rayD.z, 0.0f);                   // BEF rgen:57,73
  }
}

// wave-level sum of a 32-bit counter, then one atomic per wave
__device__ __forceinline__ void wave_add(unsigned long long* dst, uint32_t v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if((threadIdx.x & 63) == 0 && v)
    atomicAdd(dst, (unsigned long long)v);
}

// ------------------------------------------------------------------------------------------
// trace(rays_in → hits_out)
// ------------------------------------------------------------------------------------------
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
    const int id = closest_hit(S, o, d, a.tmin, a.tmax, t, tests);
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
constexpr float kTMin = 0.001f;    // rgen:51
constexpr float kTMax = 10000.0f;  // rgen:52

__global__ __launch_bounds__(256) void render_static_kernel(const SceneK scene, const RenderArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  // 8×8 pixel tile per wavefront: neighbouring lanes trace neighbouring rays
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tiles_x = (a.W + 7) >> 3;
  const uint32_t tile    = blockIdx.x * (blockDim.x >> 6) + wave;
  const uint32_t x = (tile % tiles_x) * 8 + (lane & 7);
  const uint32_t y = a.row_begin + (tile / tiles_x) * 8 + (lane >> 3);
  uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0;

  if(x < a.W && y < a.row_end)
  {
    PixelResult r;
    v3 origin, direction;
    raygen(a.g, a.toro, a.W, a.H, a.camera, x, y, origin, direction);
    r.rayO = origin;
    r.rayD = direction;
    r.t0   = __builtin_inff();
    r.P0 = r.N0 = {0.0f, 0.0f, 0.0f};
    r.id0 = -1;

    int depth = 0, done = 1;                                                 // rgen:54,57
    v3  attenuation = {1.0f, 1.0f, 1.0f};                                    // rgen:56
    v3  hitValue    = {0.0f, 0.0f, 0.0f};                                    // rgen:61
    for(;;)                                                                  // rgen:62
    {
      v3    prdHit, nextO = origin, nextD = direction;
      float t;
      const int id = closest_hit(S, origin, direction, kTMin, kTMax, t, depth == 0 ? n_primary : n_bounce);
      if(id < 0)
        prdHit = {a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f};  // rmiss:37
      else
      {
        HitState h;
        hit_begin(S, a.pc, id, t, origin, direction, h);
        if(depth == 0) { r.t0 = t; r.P0 = h.P; r.N0 = h.N; r.id0 = id; }     // BEF rgen:94-97
        bool shadowed = false;
        if(h.wantShadow)
          shadowed = any_hit(S, h.P, h.L, 0.001f, h.lightDistance, n_shadow); // rchit:114-131
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
    r.color = hitValue;
    store_pixel(a, x, y, r);
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
  hipLaunchKernelGGL(trace_kernel, dim3(grid), dim3(256), 0, stream, scene, a);
  return hipGetLastError();
}

hipError_t launch_render(const SceneK& scene, const RenderArgs& a, RenderVariant v, int n_cus,
                         hipStream_t stream)
{
  (void)v;
  (void)n_cus;
  const uint32_t rows = a.row_end - a.row_begin;
  if(rows == 0 || a.W == 0)
    return hipSuccess;
  const uint64_t tiles = (uint64_t)((a.W + 7) / 8) * ((rows + 7) / 8);
  hipLaunchKernelGGL(render_static_kernel, dim3((uint32_t)((tiles + 3) / 4)), dim3(256), 0, stream,
                     scene, a);
  return hipGetLastError();
}

}  // namespace trt
