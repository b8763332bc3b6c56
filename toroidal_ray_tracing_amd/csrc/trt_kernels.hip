// trt_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the toroidal ray tracer.
//
//   trace_kernel              trace(rays_in → hits_out): SoA rays in, closest hit out.
//   render_static_kernel      one lane per pixel, 8×8 pixel tile per wavefront; each lane runs the
//                             reference's raygen bounce loop (REFL/shaders/raytrace.rgen:62-85).
//   tile_classify[_fine]_kernel  which 8×8 tiles can be answered without tracing a ray: the CLEAR list (32×8 macro tiles,
//                             constant fills) and the LIVE list (heavy tiles of the previous frame first: cost feedback).
//   render_listed_kernel      the DEFAULT render kernel: a wave takes its entries of both lists — CLEAR macro tiles as
//                             non-temporal full-line fills, LIVE tiles traced per pixel like the static kernel — so the
//                             store-bound part of the frame drains behind the compute-bound part.
//   render_persistent_kernel  persistent wavefronts over a global work queue: the bounce
//                             loop is flattened into per-lane queries (closest-hit, shadow,
//                             bounce); a lane whose pixel is finished is refilled at once
//                             (ballot + popcount compaction from the wave's round-robin tile
//                             sequence), so every trip of the solve loop works on 64 live
//                             ray–torus tests.
//
// One lane = one ray.  Scene constants are staged into LDS once per block.  No MFMA: the
// work is scalar FP32/FP64 root finding.  post_kernel (tonemap) and the splat_* kernels (point-cloud re-projection) serve the
// callers either side of the path.  Compiled with -ffp-contract=off (see
// trt_device.hpp for the arithmetic contract).  Every kernel is instantiated for the FP32
// and the FP64 root solve (BASELINE config 4); I/O is FP32 in both.
#include "trt_kernels.hpp"

#include <cstdlib>

namespace trt {

// ------------------------------------------------------------------------------------------
// global-memory accessors
// ------------------------------------------------------------------------------------------
// The long kernels read their arguments (and thus their output POINTERS) from LDS, so hipcc no
// longer knows that those pointers address global memory and would emit flat_load/flat_store —
// slower, and counted on lgkmcnt as well, so that every LDS wait would also wait for them.
// These helpers cast to the global address space: global_load / global_store.
template <class T> using gptr = __attribute__((address_space(1))) T*;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int   i4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st1(float* base, size_t i, float v) { ((gptr<float>)base)[i] = v; }
__device__ __forceinline__ void st1(int32_t* base, size_t i, int32_t v) { ((gptr<int32_t>)base)[i] = v; }
__device__ __forceinline__ void st4(float* p, float4 v) { *((gptr<f4v>)p) = f4v{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st4(int32_t* p, int x, int y, int z, int w) { *((gptr<i4v>)p) = i4v{x, y, z, w}; }
// Non-temporal (`nt`) dwordx4 stores for the FULL-LINE streams that nothing reads again inside the frame: the constant
// fills of CLEAR macro tiles (85 % of the baseline frame).  Measured (config 3, one box, alternating processes): frame
// 0.131 → 0.119 ms when the chip is in its fast state and 0.156 → 0.124 ms in its slow one — the fills no longer
// compete for L2 / Infinity-Cache lines with the partial-line stores of the traced tiles, which need them to merge.
// Applied to EVERY store the frame got slower (0.185 ms): the traced tiles' dword stores must stay temporal.
#ifdef TRT_NO_NT_CLEAR   // timing builds
#define st4c st4
#else
__device__ __forceinline__ void st4c(float* p, float4 v) { __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, (gptr<f4v>)p); }
__device__ __forceinline__ void st4c(int32_t* p, int x, int y, int z, int w) { __builtin_nontemporal_store(i4v{x, y, z, w}, (gptr<i4v>)p); }
#endif
// -DTRT_TIMELINE (tools/timeline.py only): every wave of the listed kernel stamps the 100-MHz wall clock at entry, past the staging barrier,
// before its first tile and at exit, plus the hardware slot it ran in and its first LIVE tile, into g_timeline[wave][8].
#ifdef TRT_TIMELINE
__device__ unsigned long long* g_timeline = nullptr;
#define TRT_STAMP(k, v) do { if(g_timeline && (threadIdx.x & 63) == 0) g_timeline[(size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + (k)] = (v); } while(0)
#else
#define TRT_STAMP(k, v) do { } while(0)
#endif

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }   // (min(int, uint32_t) resolves to the double overload)
__device__ __forceinline__ uint32_t ld1(const uint32_t* base, size_t i) { return ((gptr<const uint32_t>)base)[i]; }

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

// The id of a miss, materialised at the store: as a plain constant hipcc hoists (-1,-1,-1,-1) out of the
// tile loop, keeps it live across the whole solve and — in the FP64 kernels at 128 VGPRs — spills it
// (20 B of scratch whose every reload is a vector-memory load that drains the output stores).
__device__ __forceinline__ int miss_id()
{
  int m;
  asm volatile("v_mov_b32 %0, -1" : "=v"(m));
  return m;
}

__device__ __forceinline__ void store_first_hit(const RenderArgs& a, size_t i_, float t, v3 P, v3 N, int id)
{
  // The pixel index passes through an opaque copy so that the eight stream addresses are formed
  // HERE, at the store, and not at the top of the pixel's bounce loop — where they would sit in
  // 16 VGPRs across the whole solve (and get spilled).  W·H < 2³¹ (trt_render checks it).
  uint32_t i32 = (uint32_t)i_;
  asm volatile("" : "+v"(i32));
  const size_t i = i32;
  if(a.hits.t) st1(a.hits.t, i, t);
  if(a.hits.px) st1(a.hits.px, i, P.x);
  if(a.hits.py) st1(a.hits.py, i, P.y);
  if(a.hits.pz) st1(a.hits.pz, i, P.z);
  if(a.hits.nx) st1(a.hits.nx, i, N.x);
  if(a.hits.ny) st1(a.hits.ny, i, N.y);
  if(a.hits.nz) st1(a.hits.nz, i, N.z);
  if(a.hits.id) st1(a.hits.id, i, id);
}

// Query counters of a block → the three global totals: wave sums by shuffles, block sums by LDS
// atomics, then ONE global atomic per counter per block (65,536 waves adding to three words one
// by one made the counted pass of the listed kernel 1.2 ms long).  Every thread of the block must
// call it (it contains barriers); `stats` is kernel-uniform.
// Layout of the totals: trt_stats without `pixels` — [0] primary, [1] bounce, [2] shadow tests, [3] unused,
// [4] traced, [5] solved tests, [6] evaluations.
__device__ __forceinline__ void block_add_stats(unsigned long long* stats, uint32_t v0, uint32_t v1, uint32_t v2, const WorkCount& wc)
{
  __shared__ unsigned int acc[8];
  if(threadIdx.x < 8) acc[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t v[6] = {v0, v1, v2, wc.traced, wc.solved, wc.evals};
  for(int off = 32; off > 0; off >>= 1)
#pragma unroll
    for(int k = 0; k < 6; ++k)
      v[k] += __shfl_down(v[k], off, 64);
  if((threadIdx.x & 63) == 0)
  {
#pragma unroll
    for(int k = 0; k < 6; ++k)
      if(v[k]) atomicAdd(&acc[k < 3 ? k : k + 1], v[k]);
  }
  __syncthreads();
  if(threadIdx.x < 8 && acc[threadIdx.x])
    atomicAdd(&stats[threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}

// Retire every outstanding load of this wave, then hide the given registers from hipcc's
// s_waitcnt bookkeeping.  Without this, a value loaded once per batch and read in a loop (the
// per-lane tile-list caches) gets an `s_waitcnt vmcnt(0)` in front of EVERY read — and since
// stores share the counter, each of those waits drains the wave's whole stream of output
// stores (measured: the clear tiles then serialise with the traced tiles instead of
// draining behind them).
__device__ __forceinline__ void settle_loads(uint32_t& a, uint32_t& b)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(a), "+v"(b));
}

// Stage the launch arguments into LDS next to the scene.  Kept in the kernel-argument segment
// they would be pinned in ~150 SGPRs for the whole persistent loop (hipcc loads kernargs once
// and never rematerialises them), and the spills cost a dozen v_readlane per output store.
__device__ __forceinline__ void stage_args(RenderArgs* lds, const RenderArgs& arg)
{
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&arg);
  uint32_t*       dst = reinterpret_cast<uint32_t*>(lds);
  for(uint32_t i = threadIdx.x; i < sizeof(RenderArgs) / 4; i += blockDim.x)
    dst[i] = src[i];
}

// Both stagings in ONE pass for a block of exactly 256 threads: thread t < sizeof(RenderArgs)/4 copies argument dword t,
// the threads from 128 on copy the scene records in use — one load per thread and one barrier.  (The general loops above
// compile to ≈200 instructions per wave with an unknown block size; a wave of the listed kernel lives for one tile, so
// its prologue was 40 % of all instructions the LIVE part of config 3 issued — tools/timeline.py)
__device__ __forceinline__ void stage_block256(SceneK* S, RenderArgs* A, const SceneK& scene, const RenderArgs& arg)
{
  constexpr uint32_t NA = sizeof(RenderArgs) / 4;
  static_assert(NA <= 128 && sizeof(RenderArgs) % 4 == 0, "RenderArgs must fit the lower half of the block");
  const uint32_t tid = threadIdx.x;
  if(tid < NA)
    reinterpret_cast<uint32_t*>(A)[tid] = reinterpret_cast<const uint32_t*>(&arg)[tid];
  else if(tid >= 128u)
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&scene);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(S);
    const uint32_t  c4 = scene_words(scene);
#pragma unroll 1
    for(uint32_t i = tid - 128u; i < c4; i += 128u)
    {
      const uint32_t off = scene_word(scene, i);
      dst[off] = src[off];
    }
  }
  __syncthreads();
}

constexpr float kTMin = 0.001f;    // rgen:51, rchit:114
constexpr float kTMax = 10000.0f;  // rgen:52

// ------------------------------------------------------------------------------------------
// trace(rays_in → hits_out)
// ------------------------------------------------------------------------------------------
template <class Real, bool DK>
__global__ __launch_bounds__(256) void trace_kernel(const SceneK scene, const TraceArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  uint32_t       tests  = 0;
  WorkCount      wc;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.rays.n; i += stride)
  {
    const v3 o = {a.rays.ox[i], a.rays.oy[i], a.rays.oz[i]};
    const v3 d = {a.rays.dx[i], a.rays.dy[i], a.rays.dz[i]};
    float     t;
    const int id = closest_hit<Real, DK, kWalkTable>(S, o, d, a.tmin, a.tmax, t, tests, wc);   // incoherent rays: trt_device.hpp
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
    block_add_stats(a.stats, tests, 0u, 0u, wc);
}

// ------------------------------------------------------------------------------------------
// render, static mapping: lane ↔ pixel for the whole bounce loop
// ------------------------------------------------------------------------------------------
// One pixel, start to finish, on one lane: the reference's raygen main() with the closest-hit,
// miss and shadow-miss shaders inlined (REFL/shaders/raytrace.rgen:40-88).
// ------------------------------------------------------------------------------------------
// RenderedData export (BEF/shaders/raytrace.rgen:72-73,111-112), staged through LDS
// ------------------------------------------------------------------------------------------
// RenderedData is an array of 64-B records at index x*H + y: the 8 pixels of one tile COLUMN are
// 512 contiguous bytes, but a lane that stores its own record piece by piece (rayOrigin/rayDir at
// ray generation, pos at the first hit, colour at the end) writes 16 B of every 64 — a wave
// instruction then touches 32 cache lines for 1 KB, and the four pieces of a record reach the L2
// far apart in time.  The listed kernel therefore collects the 64 records of its 8×8 tile in a
// per-wave LDS image (4 KB) and writes it out transposed: 4 dwordx4 instructions, each covering two
// whole tile columns = 2 × 512 contiguous bytes (8 full lines per instruction, like every other
// store of the frame).  Record of tile pixel (x, y), piece k (0 pos, 1 colour, 2 rayOrigin,
// 3 rayDir) sits at 16-B unit y·32 + ((x ^ y) & 7)·4 + k: the XOR spreads a column over the banks,
// so neither the record writes (lanes of equal y) nor the column reads (lanes of equal x) conflict.
__device__ __forceinline__ uint32_t rd_unit(uint32_t x, uint32_t y, uint32_t k) { return y * 32u + (((x ^ y) & 7u) << 2) + k; }

// Where a lane puts the pieces of its pixel's record: the wave's LDS image (listed kernel), or — the
// other variants — straight to global memory.
struct RdSink {
  float4* lds;   // the wave's 256-unit image, already offset to this lane's record (or nullptr)
  float*  glob;  // &rendered[x*H + y] (or nullptr)
  __device__ __forceinline__ explicit operator bool() const { return lds != nullptr || glob != nullptr; }
  __device__ __forceinline__ void put(uint32_t k, float4 v) const
  {
    if(lds) lds[k] = v;
    else if(glob) st4(glob + 4 * k, v);
  }
};

// Writes the wave's LDS image of tile (tx, ty) to RenderedData, transposed (see above).
// Non-temporal like the CLEAR fills (whole 512-B runs, read by nobody in the frame): capture 0.186 → 0.178 ms.
__device__ __forceinline__ void rd_flush(const RenderArgs& a, const float4* tile, uint32_t tx, uint32_t ty, uint32_t lane)
{
  __builtin_amdgcn_wave_barrier();   // LDS operations of one wave execute in order: the reads below see the records
#pragma unroll
  for(uint32_t j = 0; j < 4; ++j)
  {
    const uint32_t c = j * 64u + lane, xl = c >> 5, yl = (c >> 2) & 7u, k = c & 3u;
    const uint32_t x = tx * 8u + xl, ly = ty * 8u + yl;
    if(x < a.W && ly < a.n_local_rows)
    {
      const float4 v = tile[rd_unit(xl, yl, k)];
      st4c(reinterpret_cast<float*>(&a.rendered[(size_t)x * a.H + image_row(a, ly)]) + 4 * k, v);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The record of a pixel that misses at depth 0, for a whole tile that the classification proved
// empty: primary ray from raygen(), pos = (0,0,0,1) (BEF rmiss:21 → rgen:112), colour = clear·0.8
// (rmiss:37 → rgen:76 with attenuation 1 → rgen:111) — what trace_pixel() would have produced.
__device__ __forceinline__ void rd_miss_tile(const RenderArgs& a, float4* tile, uint32_t tx, uint32_t ty, uint32_t lane)
{
  const uint32_t xl = lane & 7u, yl = lane >> 3, x = tx * 8u + xl, ly = ty * 8u + yl;
  if(x < a.W && ly < a.n_local_rows)
  {
    v3 o, d;
    raygen(a.g, a.toro, a.W, a.H, a.camera, x, image_row(a, ly), o, d);
    float4* r = tile + rd_unit(xl, yl, 0);
    r[0] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    r[1] = make_float4(a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f, 1.0f);
    r[2] = make_float4(o.x, o.y, o.z, 1.0f);
    r[3] = make_float4(d.x, d.y, d.z, 0.0f);
  }
  rd_flush(a, tile, tx, ty, lane);
}

template <class Real, bool DK>
__device__ __forceinline__ void trace_pixel(const SceneK& S, const RenderArgs& a, uint32_t x, uint32_t y, uint32_t ly, const RdSink rd,
                                            uint32_t& n_primary, uint32_t& n_bounce, uint32_t& n_shadow, WorkCount& wc)
{
  const size_t oi = out_index(a, x, y, ly);
  v3 origin, direction;
  raygen(a.g, a.toro, a.W, a.H, a.camera, x, y, origin, direction);
  if(rd)
  {
    rd.put(2, make_float4(origin.x, origin.y, origin.z, 1.0f));            // BEF rgen:56,72
    rd.put(3, make_float4(direction.x, direction.y, direction.z, 0.0f));   // BEF rgen:57,73
  }

  int depth = 0, done = 1;                                                 // rgen:54,57
  // (the constant 1 is materialised per pixel: hoisted out of the tile loop hipcc keeps — and, at 80 VGPRs, spills — it)
  float one0;
  asm volatile("v_mov_b32 %0, 1.0" : "=v"(one0));
  v3  attenuation = {one0, one0, one0};                                    // rgen:56
  v3  hitValue    = {0.0f, 0.0f, 0.0f};                                    // rgen:61
  // enclosure cull (trt_device.hpp closest_hit): the tori this path's rays cannot hit first — tubes strictly inside a tube
  // the ray origin is outside of.  The camera's share is certified per frame on the host; a hit left OUTWARDS adds the
  // tubes inside the torus hit (outside-ness persists along a path: a segment that ended on a surface crossed none).
  uint32_t skip = a.skip_primary;
  for(;;)                                                                  // rgen:62
  {
    v3    prdHit, nextO = origin, nextD = direction;
    float t;
    const int id = closest_hit<Real, DK>(S, origin, direction, kTMin, kTMax, t, depth == 0 ? n_primary : n_bounce, wc, skip);
    if(id < 0)
    {
      prdHit = {a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f};  // rmiss:37
      if(depth == 0)
      {
        store_first_hit(a, oi, t, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, miss_id());  // BEF rmiss:21
        if(rd)
        {
          // materialised here: hoisted out of the tile loop this constant vector gets spilled
          float z, o1;
          asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 1.0" : "=v"(z), "=v"(o1));
          rd.put(0, make_float4(z, z, z, o1));
        }
      }
    }
    else
    {
      HitState h;
      hit_begin(S, a.pc, id, t, origin, direction, h);
      if(depth == 0)                                                       // BEF rgen:94-97
      {
        store_first_hit(a, oi, t, h.P, h.N, id);
        if(rd) rd.put(0, make_float4(h.P.x, h.P.y, h.P.z, 1.0f));          // BEF rgen:112
      }
      bool shadowed = false;
      const uint32_t inside = S.inside[id];
      if(h.wantShadow)   // (N·L > 0: the shadow ray leaves the surface outwards)
        shadowed = any_hit<Real, DK>(S, h.P, h.L, kTMin, h.lightDistance, n_shadow, wc, skip | inside);  // rchit:114-131
      if(dot3(h.N, direction) < 0.0f)   // hit from outside: reflect(D, N) leaves outwards
        skip |= inside;
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
  // alpha 1, materialised here: as a plain constant it is kept live across the bounce loop and spilled when the
  // FP32 kernel is held to 80 VGPRs
  float one;
  asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
  const float4 c = make_float4(hitValue.x, hitValue.y, hitValue.z, one);
  if(a.rgba) st4(a.rgba + 4 * oi, c);                                      // rgen:87
  if(rd) rd.put(1, c);                                                     // BEF rgen:111
}

template <class Real, int TW, bool DK>
__global__ __launch_bounds__(256) void render_static_kernel(const SceneK scene, const RenderArgs a)
{
  __shared__ SceneK S;
  stage_scene(&S, scene);

  // TW×TH pixel tile per wavefront (TW·TH = 64): neighbouring lanes trace neighbouring rays;
  // a row of the tile is TW·16 B of rgba and TW·4 B of every first-hit stream
  constexpr int TH = 64 / TW;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tiles_x = (a.W + TW - 1) / TW;
  const uint32_t tile    = blockIdx.x * (blockDim.x >> 6) + wave;
  const uint32_t x  = (tile % tiles_x) * TW + (lane % TW);
  const uint32_t ly = (tile / tiles_x) * TH + (lane / TW);
  uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0;
  WorkCount wc;
  if(x < a.W && ly < a.n_local_rows)
  {
    const uint32_t y = image_row(a, ly);
    const RdSink rd{nullptr, a.rendered ? reinterpret_cast<float*>(&a.rendered[(size_t)x * a.H + y]) : nullptr};   // BEF rgen:72
    trace_pixel<Real, DK>(S, a, x, y, ly, rd, n_primary, n_bounce, n_shadow, wc);
  }
  if(a.stats)
  {
    block_add_stats(a.stats, n_primary, n_bounce, n_shadow, wc);
  }
}

// ------------------------------------------------------------------------------------------
// tile classification: which 8×8 tiles can be answered without tracing a single ray
// ------------------------------------------------------------------------------------------
// One lane per tile.  A tile is CLEAR when every primary ray of the tile provably misses the
// (inflated) bounding sphere of every torus: its pixels are then misses at depth 0 —
// rgba = (clearColor·0.8, 1), first hit = (inf, 0, 0, -1) — exactly what the per-pixel path
// would compute (raytrace.rmiss:37, BEF rmiss:21), because TorusTest::setup() culls on the
// same sphere.  The bound is conservative: with the tile's centre ray (oc, dc) and its four
// corner-pixel rays, every ray of the tile starts within Δo of oc and points within θ of dc
// (θ = k · max corner chord; the angle to dc is quasi-convex on the image plane, so its
// maximum over the pixel rectangle sits at a corner; k covers chord→angle and, for the
// toroidal camera, the non-planar patch).  The distance from a torus centre C to the ray's
// line is 1-Lipschitz in the origin and |C-o|-Lipschitz in the direction angle, hence
//     dist >= dl - Δo - (L + Δo)·θ,   dl = dist(C, centre line), L = |C - oc|,
// and the tile is clear when that exceeds the sphere radius by 1.6 % + 1e-5·(L+1) — three
// orders of magnitude above the FP32 rounding of the per-pixel test.  A second test does the
// same for the bounding box (cylinder ∩ slab — it contains the sphere ∩ slab that TorusTest::setup() clips to), which removes the
// caps of the sphere's silhouette and everything behind the camera.  Anything doubtful
// (NaN, wide tiles, origin near the sphere) is LIVE.  Tiles are appended to two compact lists
// (one wave-aggregated atomic per list per wave); order within the lists is irrelevant.
// Approximate reciprocal / square roots for the tile classification only: its margins (≥1.6 %)
// are four orders above their rounding (1 ulp), and nothing in the classification has to agree
// bit for bit with anything (a tile is either provably clear or traced ray by ray).
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ v3 fnormalize(v3 a) { return scale3(a, frsq(dot3(a, a))); }

// raygen() with approximate division / normalisation (classification only)
__device__ __forceinline__ void raygen_fast(const trt_globals& g, const ToroCam& tc, uint32_t W, uint32_t H, int camera,
                                            uint32_t x, uint32_t y, v3& origin, v3& dir)
{
  if(camera == TRT_CAMERA_TOROIDAL)
  {
    const float ca = tc.cos_a[x], sa = tc.sin_a[x], cb = tc.cos_b[y], sb = tc.sin_b[y];
    origin = {fma_(tc.rho, ca, tc.eye[0]), tc.eye[1], fma_(tc.rho, sa, tc.eye[2])};
    dir    = {ca * cb, sb, sa * cb};   // unit
    return;
  }
  const float u = ((float)x + 0.5f) * frcp((float)W), v = ((float)y + 0.5f) * frcp((float)H);
  origin       = mat4_mul(g.viewInverse, 0.0f, 0.0f, 0.0f, 1.0f);
  const v3 tgt = mat4_mul(g.projInverse, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f);
  const v3 tn  = fnormalize(tgt);
  dir          = fnormalize(mat4_mul(g.viewInverse, tn.x, tn.y, tn.z, 0.0f));
}

template <bool MARCH>
__device__ __forceinline__ bool tile_is_clear(const SceneK& S, const RenderArgs& a, uint32_t x0, uint32_t ty, uint32_t width)
{
  const uint32_t x1 = min(x0 + width - 1, a.W - 1);
  const uint32_t l0 = ty * 8, l1 = min(l0 + 7, a.n_local_rows - 1);
  const uint32_t y0 = image_row(a, l0), y1 = image_row(a, l1);
  const uint32_t xs[5] = {(x0 + x1 + 1) >> 1, x0, x1, x0, x1};
  const uint32_t ys[5] = {(y0 + y1 + 1) >> 1, y0, y0, y1, y1};
  v3    oc = {0.0f, 0.0f, 0.0f}, dc = {0.0f, 0.0f, 1.0f};
  float chord2 = 0.0f, shift2 = 0.0f;
#pragma unroll
  for(int i = 0; i < 5; ++i)
  {
    v3 o, d;
    raygen_fast(a.g, a.toro, a.W, a.H, a.camera, xs[i], ys[i], o, d);   // unit direction
    if(i == 0) { oc = o; dc = d; }
    else
    {
      const v3 dd = sub3(d, dc), od = sub3(o, oc);
      chord2 = max_(chord2, dot3(dd, dd));
      shift2 = max_(shift2, dot3(od, od));
    }
  }
  const float theta = (a.camera == TRT_CAMERA_PINHOLE ? 1.6f : 2.0f) * fsqrt(chord2);
  const float dO    = 1.5f * fsqrt(shift2);
  if(!(theta < 0.5f))
    return false;
  for(int i = 0; i < S.n_tori; ++i)
  {
    const v3    v  = sub3(v3{S.shade[i].cx, S.shade[i].cy, S.shade[i].cz}, oc);
    const float L2 = dot3(v, v), s = dot3(v, dc);
    const float L  = fsqrt(L2), dl = fsqrt(max_(L2 - s * s, 0.0f));
    const float rb = fsqrt(S.k32[i].Rb2);
    // (1) every line of the bundle misses the bounding sphere
    if(dl - dO - (L + dO) * theta > rb * 1.015625f + 1e-5f * (L + 1.0f))
      continue;
    // (2) the centre ray misses the bounding box (cylinder ∩ slab ⊇ sphere ∩ slab, the solid TorusTest::setup
    //     clips to) inflated by delta, the largest distance between a point of any ray of the
    //     bundle and the centre ray's point at the same parameter, over the parameters at which
    //     the sphere can be met (t <= L + rb): delta = Δo + (L + rb)·θ
    const float delta = 1.02f * (dO + (L + rb) * theta) + 1e-5f * (L + 1.0f);
    const float Rc = rb * 1.015625f + delta, hs = S.k32[i].rs * 1.015625f + delta;
    const float ex = -v.x, ey = -v.y, ez = -v.z;
    float t_lo = 0.0f, t_hi = L + rb + delta;   // forward half-line only, inside the sphere's reach
    const float ca = fma_(dc.z, dc.z, dc.x * dc.x), cb = fma_(ez, dc.z, ex * dc.x), cc = fma_(ez, ez, ex * ex);
    bool miss = false;
    if(ca > 1e-12f)
    {
      const float disc = fma_(cb, cb, -(ca * (cc - Rc * Rc)));
      if(disc < 0.0f) miss = true;
      else
      {
        const float sq = fsqrt(disc), ia = frcp(ca);
        t_lo = max_(t_lo, (-cb - sq) * ia - delta);
        t_hi = min_(t_hi, (sq - cb) * ia + delta);
      }
    }
    else if(cc > Rc * Rc) miss = true;
    if(!miss)
    {
      if(abs_(dc.y) > 1e-6f)
      {
        const float iy = frcp(dc.y), u0 = (-hs - ey) * iy, u1 = (hs - ey) * iy;
        t_lo = max_(t_lo, min_(u0, u1) - delta);
        t_hi = min_(t_hi, max_(u0, u1) + delta);
      }
      else if(abs_(ey) > hs) miss = true;
    }
    if(miss || t_lo > t_hi)
      continue;
    // (3) the bundle passes through the bounding box: march the centre ray through [t_lo, t_hi]
    //     with the torus' distance function dist(P) = |(ρ - R, y)| - r (1-Lipschitz).  Every point
    //     of every ray of the bundle at arc length s lies within dev(s) = Δo + s·θ of the centre
    //     ray's point, so while slack = dist - dev stays positive no ray touches the torus, and a
    //     step of slack / (1 + θ) keeps it positive.  Tiles in the hole or along the silhouette
    //     run out of slack or of steps and stay LIVE (NaNs too).
    if(MARCH)
    {
      const float R = S.shade[i].R, r = fsqrt(S.k32[i].r2);
      const float kstep = 0.9f * frcp(1.0f + theta), floor_ = 0.02f * r, pad = 1e-5f * (L + 1.0f);
      float sArc = t_lo;
      bool  passed = false;
      for(int it = 0; it < 16; ++it)
      {
        const float px = fma_(sArc, dc.x, ex), py = fma_(sArc, dc.y, ey), pz = fma_(sArc, dc.z, ez);
        const float e  = fsqrt(fma_(pz, pz, px * px)) - R;
        const float dist  = fsqrt(fma_(e, e, py * py)) - r;
        const float slack = dist - (1.02f * (dO + sArc * theta) + pad);
        if(!(slack > floor_))
          break;
        sArc = fma_(slack, kstep, sArc);
        if(sArc > t_hi) { passed = true; break; }
      }
      if(passed)
        continue;
    }
    return false;  // this torus may be hit by some ray of the tile
  }
  return true;
}

// End of a classification block.  The list lengths are accumulated in a.counters[0..1] (zero when the
// kernel starts); every block takes a ticket (sharded: see below) once its two reservations have
// returned, and the block that draws the LAST ticket — every other block's additions are then
// performed — moves the totals to a.counts (what the render kernels read) and leaves all three
// accumulators zero for the next frame.  A frame therefore depends on no other frame: no memset, no
// double buffering, nothing that distinguishes eager launches from hipGraph replays.
// The ticket is drawn right after the barrier that follows the reservations and BEFORE the block's list
// writes (classify_ticket), so that the latency of the returning atomic hides behind those stores; the
// publication itself (classify_publish) comes last.  (The list entries are read by the NEXT kernel: the
// kernel boundary orders them, not the ticket.)
// The tickets are sharded over eight words (a.counters[8 + blockIdx % 8]): 256 returning atomics on ONE word
// take ≈3 µs (≈12 ns each, MI355X_MICROARCH.md "fanin") at the tail of a 9-µs kernel; the last block of a shard
// draws a second-level ticket on a.counters[2], and the last of those publishes.
__device__ __forceinline__ unsigned int classify_ticket(const RenderArgs& a)
{
  return threadIdx.x == 0 ? atomicAdd(&a.counters[8u + (blockIdx.x & 7u)], 1u) : 0u;   // the reservations of threads 0 and 1 have returned
}

__device__ __forceinline__ void classify_publish(const RenderArgs& a, unsigned int ticket)
{
  if(threadIdx.x < 64u)   // the block's first wave: thread 0 holds the ticket, lanes 0…4 fetch the five accumulators at once
  {
    int last = 0;
    if(threadIdx.x == 0)
    {
      const unsigned int shard = blockIdx.x & 7u, in_shard = (gridDim.x - shard + 7u) >> 3, n_shards = gridDim.x < 8u ? gridDim.x : 8u;
      if(ticket == in_shard - 1)
      {
        atomicExch(&a.counters[8u + shard], 0u);
        last = atomicAdd(&a.counters[2], 1u) == n_shards - 1 ? 1 : 0;
      }
    }
    last = __shfl(last, 0, 64);
    if(last)
    {
      // (five exchanges in ONE instruction instead of five dependent round trips at the very end of the kernel)
      const uint32_t word = threadIdx.x < 2u ? threadIdx.x : threadIdx.x + 1u;   // counters 0, 1, 3, 4, 5
      const uint32_t v = threadIdx.x < 5u ? atomicExch(&a.counters[word], 0u) : 0u;
      const unsigned int n_norm = __shfl(v, 0, 64), n_clear = __shfl(v, 1, 64), n_heavy = __shfl(v, 2, 64);
      const unsigned int cost_sum = __shfl(v, 3, 64), cost_cnt = __shfl(v, 4, 64);
      if(threadIdx.x == 0)
      {
        const unsigned int n_live = n_norm + n_heavy < a.cap_live ? n_norm + n_heavy : a.cap_live;   // (their sum never exceeds the tiles)
        a.counts[0] = n_live;
        a.counts[1] = n_clear < a.cap_clear ? n_clear : a.cap_clear;
        a.counts[2] = n_heavy < n_live ? n_heavy : n_live;
        a.counts[3] = cost_cnt ? cost_sum / cost_cnt : 0u;   // the threshold of the NEXT frame's classification
        atomicExch(&a.counters[2], 0u);
      }
    }
  }
}

// Logical LIVE entry L → position in tiles_live (RenderArgs::tile_cost): the heavy tiles come first.
__device__ __forceinline__ size_t live_slot(uint32_t cap_live, uint32_t n_heavy, uint64_t L)
{
  return L < n_heavy ? (size_t)cap_live - 1 - (size_t)L : (size_t)(L - n_heavy);
}

// Block size of the classification kernels: the largest there is.  Every block reserves its stretch of each list with ONE
// returning atomic on the list's counter, and returning atomics on one word serialise at ≈11 ns each (MI355X_MICROARCH.md
// "fanin"): with 256-thread blocks a 4096² frame queued 256 of them (≈3 µs of a 9-µs kernel), an 8192² frame 1,024
// (≈12 µs).  1,024 threads: config 3 −3 %, the 8192² frame 0.457 → 0.420 ms, the toroidal captures −6…7 %.
#ifndef TRT_CLASSIFY_THREADS
#define TRT_CLASSIFY_THREADS 1024
#endif
constexpr int kClassifyThreads = TRT_CLASSIFY_THREADS;
constexpr uint32_t kMacroTiles = 4;  // a macro tile = 4 horizontally adjacent 8×8 tiles = 32×8 pixels

// Tile-list entries.  One frame per launch: tx | ty << 16 [| miss flag] (tx < 2^16, ty < 2^15).  A batch of frames
// (trt_render_batch_dev): tx | ty << 13 | frame << 28 [| miss flag] (tx < 2^13: W <= 65528).
constexpr uint32_t kTileMissFlag = 0x80000000u;
template <bool BATCH> struct TileCode;
template <> struct TileCode<false> {
  static __device__ __forceinline__ uint32_t pack(uint32_t tx, uint32_t ty, uint32_t) { return tx | (ty << 16); }
  static __device__ __forceinline__ uint32_t x(uint32_t p) { return p & 0xffffu; }
  static __device__ __forceinline__ uint32_t y(uint32_t p) { return (p >> 16) & 0x7fffu; }
  static __device__ __forceinline__ uint32_t frame(uint32_t) { return 0u; }
};
template <> struct TileCode<true> {
  static __device__ __forceinline__ uint32_t pack(uint32_t tx, uint32_t ty, uint32_t f) { return tx | (ty << 13) | (f << 28); }
  static __device__ __forceinline__ uint32_t x(uint32_t p) { return p & 0x1fffu; }
  static __device__ __forceinline__ uint32_t y(uint32_t p) { return (p >> 13) & 0x7fffu; }
  static __device__ __forceinline__ uint32_t frame(uint32_t p) { return (p >> 28) & 7u; }
};
__device__ __forceinline__ uint32_t tile_x(uint32_t packed) { return TileCode<false>::x(packed); }
__device__ __forceinline__ uint32_t tile_y(uint32_t packed) { return TileCode<false>::y(packed); }

// The launch arguments of the frame a wave works on: the kernel's own RenderArgs, or frame f of a batch (f wave-uniform).
__device__ __forceinline__ const RenderArgs& frame_args(const RenderArgs& a, uint32_t) { return a; }
__device__ __forceinline__ const RenderArgs& frame_args(const RenderBatch& b, uint32_t f) { return b.fr[f]; }
template <bool BATCH> struct LaunchArgs { typedef RenderArgs type; };
template <> struct LaunchArgs<true> { typedef RenderBatch type; };

// What the two classification kernels share.  Per lane: the LIVE tiles it contributes, as NORMAL ones in the low and as
// HEAVY ones in the high half of ONE word (a wave holds at most 256 of either, a block 4,096: the halves never carry
// into each other), its CLEAR macro tile, and the macro tile's previous cost.  Three wave scans (as many shuffles as two
// lists cost before, plus one), a ballot for the number of macro tiles with a cost.  Rows of wave_cnt: 0 packed LIVE
// totals per wave, 1 CLEAR (turned into its prefix in place), 2 cost sums, 3 cost counts, 4 exclusive prefix of row 0.
// Threads 0, 1, 2 reserve the block's stretch of the NORMAL / CLEAR / HEAVY list (a.counters[0], [1], [3]), threads 3 and
// 4 add the block's cost sum and count (a.counters[4], [5]) — five RETURNING atomics whose results are in LDS before
// the barrier, hence performed before the block's ticket.
constexpr int kClassifyRows = 5;

__device__ __forceinline__ uint32_t classify_take_cost(const RenderArgs& a, bool owner, uint32_t macro)
{
  if(!a.tile_cost || !owner)
    return 0u;
  const uint32_t c = a.tile_cost[macro];
  if(c) a.tile_cost[macro] = 0u;
  return c;
}

__device__ __forceinline__ bool classify_is_heavy(const RenderArgs& a, uint32_t cost)
{
  const uint32_t mean = a.counts[3];   // published by the previous classification
  return a.heavy_x16 != 0u && mean != 0u && (uint64_t)cost * 16u > (uint64_t)mean * a.heavy_x16;
}

// pre = {packed LIVE, CLEAR, cost}: inclusive wave scans; the wave's totals go to rows 0..2, its cost count to row 3.
// FB = false (no cost feedback in this launch): two scans, as before the feedback existed; rows 2 and 3 stay zero.
template <bool FB>
__device__ __forceinline__ void classify_scan(uint32_t (&pre)[3], bool has_cost, uint32_t (*wave_cnt)[kClassifyThreads / 64], uint32_t lane, uint32_t wave)
{
#pragma unroll
  for(int off = 1; off < 64; off <<= 1)
#pragma unroll
    for(int k = 0; k < (FB ? 3 : 2); ++k)
    {
      const uint32_t v = __shfl_up(pre[k], off, 64);
      if(lane >= (uint32_t)off) pre[k] += v;
    }
  const uint32_t n_cost = FB ? (uint32_t)__popcll(__ballot(has_cost)) : 0u;
  if(lane == 63)
  {
    wave_cnt[0][wave] = pre[0];
    wave_cnt[1][wave] = pre[1];
    wave_cnt[2][wave] = FB ? pre[2] : 0u;
    wave_cnt[3][wave] = n_cost;
  }
}

template <bool FB>
__device__ __forceinline__ void classify_reserve(const RenderArgs& a, uint32_t (*wave_cnt)[kClassifyThreads / 64], uint32_t* block_base)
{
  const uint32_t k = threadIdx.x;
  if(k < (FB ? (uint32_t)kClassifyRows : 2u))   // no feedback: the NORMAL and the CLEAR list only
  {
    const uint32_t row = k == 2u ? 0u : (k >= 3u ? k - 1u : k);   // thread 2 reads row 0 (its high halves), threads 3, 4 rows 2, 3
    uint32_t sum = 0;
    for(uint32_t w = 0; w < kClassifyThreads / 64; ++w)
    {
      const uint32_t c = wave_cnt[row][w];
      if(k == 0u) wave_cnt[4][w] = sum;       // exclusive packed prefix over the block's waves
      else if(k == 1u) wave_cnt[1][w] = sum;  // CLEAR: in place
      sum += c;
    }
    if(k == 0u) sum &= 0xffffu;
    else if(k == 2u) sum >>= 16;
    const uint32_t word[kClassifyRows] = {0u, 1u, 3u, 4u, 5u};
    block_base[k] = sum ? atomicAdd(&a.counters[word[k]], sum) : 0u;
  }
}

// One lane per MACRO tile (32×8 pixels: one 128-B line of every first-hit stream per row).
// A clear macro tile becomes ONE entry of the CLEAR list (written later with full-line
// dwordx4 stores); any other macro tile contributes its 8×8 tiles to the LIVE list.
// (Ordering the LIVE list heavy-tiles-first was tried: render +10 %, classify 8 → 26 µs.)
template <bool FB, bool BATCH = false>
__global__ __launch_bounds__(kClassifyThreads) void tile_classify_kernel(const SceneK scene, const typename LaunchArgs<BATCH>::type args)
{
  // per-block counts, per-wave offsets inside the block's reservation: ONE device-scope atomic
  // per list per block of macro tiles (a returning atomic on a shared word costs ≈11 ns under
  // contention — MI355X_MICROARCH.md "dequeue" — so they must be rare).
  __shared__ uint32_t wave_cnt[kClassifyRows][kClassifyThreads / 64];
  __shared__ uint32_t block_base[kClassifyRows];
  // a batch: args.per_frame lanes per frame (a multiple of 64: a wave belongs to ONE frame, so its frame's arguments
  // stay scalar loads); lanes past the last frame take part in the scans and barriers with nothing to add
  const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t f = 0, t = lane_id;
  bool in_batch = true;
  if constexpr(BATCH)
  {
    f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane_id / args.per_frame));
    t = lane_id - f * args.per_frame;
    in_batch = f < args.n_frames;
    if(!in_batch) f = 0;
  }
  const RenderArgs& a = frame_args(args, f);
  const uint32_t tiles_x = (a.W + 7) >> 3, tiles_y = (a.n_local_rows + 7) >> 3;
  const uint32_t macro_x = (tiles_x + kMacroTiles - 1) / kMacroTiles;
  const bool     valid = in_batch && t < macro_x * tiles_y;
  const uint32_t mx = t % macro_x, ty = t / macro_x;
  const uint32_t tx0 = mx * kMacroTiles;
  const uint32_t ntile = valid ? min(kMacroTiles, tiles_x - tx0) : 0u;   // 8×8 tiles inside the image
  const bool     clear = valid && a.tile_cull && tile_is_clear<false>(scene, a, tx0 * 8, ty, kMacroTiles * 8);
  const uint32_t nlive = clear ? 0u : ntile;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // cost feedback: what the previous frame's slowest wave spent on this macro tile (read and reset)
  const uint32_t cost  = FB ? classify_take_cost(a, valid, t) : 0u;
  const bool     heavy = FB && nlive != 0u && classify_is_heavy(a, cost);

  // wave-level exclusive prefixes of the packed LIVE counts and of the CLEAR count; sum of the costs
  const uint32_t mine = heavy ? nlive << 16 : nlive;
  uint32_t pre[3] = {mine, clear ? 1u : 0u, nlive ? cost : 0u};
  classify_scan<FB>(pre, nlive != 0u && cost != 0u, wave_cnt, lane, wave);
  pre[0] -= mine;
  pre[1] -= clear ? 1u : 0u;
  __syncthreads();
  classify_reserve<FB>(a, wave_cnt, block_base);
  __syncthreads();
  const unsigned int ticket = classify_ticket(a);
  const uint32_t ic = block_base[1] + wave_cnt[1][wave] + pre[1];
  if(clear && ic < a.cap_clear)
    a.tiles_clear[ic] = TileCode<BATCH>::pack(tx0, ty, f);
  const uint32_t il = heavy ? block_base[2] + (wave_cnt[4][wave] >> 16) + (pre[0] >> 16) : block_base[0] + (wave_cnt[4][wave] & 0xffffu) + (pre[0] & 0xffffu);
  for(uint32_t j = 0; j < nlive; ++j)
    if(il + j < a.cap_live)
      a.tiles_live[heavy ? a.cap_live - 1u - (il + j) : il + j] = TileCode<BATCH>::pack(tx0 + j, ty, f);
  classify_publish(a, ticket);
}

// Second, finer classification (RenderArgs::fine): one lane per 8×8 tile; four consecutive lanes are one MACRO tile (32×8 pixels: one 128-B line
// of every first-hit stream per row).  A macro tile whose four tiles are all clear becomes ONE
// entry of the CLEAR list (written later with full-line dwordx4 stores); otherwise each of its
// tiles goes to the LIVE list, a clear one with kTileMissFlag set: the listed kernel writes its
// miss records without tracing (the other kernels ignore the flag and trace it — same result).
// (Ordering the LIVE list heavy-tiles-first was tried: render +10 %, classify 8 → 26 µs.)

template <bool FB, bool BATCH = false>
__global__ __launch_bounds__(kClassifyThreads) void tile_classify_fine_kernel(const SceneK scene, const typename LaunchArgs<BATCH>::type args)
{
  // per-block counts, per-wave offsets inside the block's reservation: ONE device-scope atomic
  // per list per block (a returning atomic on a shared word costs ≈11 ns under contention —
  // MI355X_MICROARCH.md "dequeue" — so they must be rare).
  __shared__ uint32_t wave_cnt[kClassifyRows][kClassifyThreads / 64];
  __shared__ uint32_t block_base[kClassifyRows];
  const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t f = 0, t = lane_id;
  bool in_batch = true;
  if constexpr(BATCH)   // (see tile_classify_kernel)
  {
    f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane_id / args.per_frame));
    t = lane_id - f * args.per_frame;
    in_batch = f < args.n_frames;
    if(!in_batch) f = 0;
  }
  const RenderArgs& a = frame_args(args, f);
  const uint32_t tiles_x = (a.W + 7) >> 3, tiles_y = (a.n_local_rows + 7) >> 3;
  const uint32_t macro_x = (tiles_x + kMacroTiles - 1) / kMacroTiles;
  const uint32_t m = t / kMacroTiles, j = t % kMacroTiles;      // macro tile, tile inside it
  const uint32_t mx = m % macro_x, ty = m / macro_x;
  const uint32_t tx = mx * kMacroTiles + j;
  const bool     valid = in_batch && ty < tiles_y && tx < tiles_x;
  const bool     clear = valid && a.tile_cull && tile_is_clear<true>(scene, a, tx * 8, ty, 8);
  // all four tiles of the macro tile clear (tiles outside the image count as clear)
  uint32_t c4 = (clear || !valid) ? 1u : 0u;
  c4 &= (uint32_t)__shfl_xor((int)c4, 1, 64);
  c4 &= (uint32_t)__shfl_xor((int)c4, 2, 64);
  const bool     macro_clear = c4 != 0u;
  const uint32_t nlive  = (valid && !macro_clear) ? 1u : 0u;
  const uint32_t nclear = (valid && macro_clear && j == 0) ? 1u : 0u;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // cost feedback: the macro tile's previous cost, read (and reset) by its first lane, shared by its four lanes
  const uint32_t cost  = FB ? (uint32_t)__shfl((int)classify_take_cost(a, in_batch && j == 0 && ty < tiles_y, m), (int)(lane & ~3u), 64) : 0u;
  const bool     heavy = FB && nlive != 0u && classify_is_heavy(a, cost);
  const bool     first = nlive != 0u && j == 0;   // (tile 0 of a macro tile is always inside the image)

  // wave-level exclusive prefixes of the packed LIVE counts and of the CLEAR count; sum of the costs (once per macro tile)
  const uint32_t mine = heavy ? nlive << 16 : nlive;
  uint32_t pre[3] = {mine, nclear, first ? cost : 0u};
  classify_scan<FB>(pre, first && cost != 0u, wave_cnt, lane, wave);
  pre[0] -= mine;
  pre[1] -= nclear;
  __syncthreads();
  classify_reserve<FB>(a, wave_cnt, block_base);
  __syncthreads();
  const unsigned int ticket = classify_ticket(a);
  const uint32_t ic = block_base[1] + wave_cnt[1][wave] + pre[1];
  const uint32_t il = heavy ? block_base[2] + (wave_cnt[4][wave] >> 16) + (pre[0] >> 16) : block_base[0] + (wave_cnt[4][wave] & 0xffffu) + (pre[0] & 0xffffu);
  if(nclear && ic < a.cap_clear)
    a.tiles_clear[ic] = TileCode<BATCH>::pack(tx, ty, f);
  if(nlive && il < a.cap_live)
    a.tiles_live[heavy ? a.cap_live - 1u - il : il] = TileCode<BATCH>::pack(tx, ty, f) | (clear ? kTileMissFlag : 0u);
  classify_publish(a, ticket);
}

// ------------------------------------------------------------------------------------------
// render, persistent wavefronts + work queue
// ------------------------------------------------------------------------------------------
// The reference's raygen loop (rgen:62-85) calls traceRayEXT, whose closest-hit shader calls
// traceRayEXT again for the shadow ray (rchit:120-131): per pixel a data-dependent chain of
// 1..2·maxDepth queries, each a loop over the tori.  Here that recursion is flattened: a lane
// owns one *query* at a time (closest-hit or shadow) and inside it one ray–torus *test*
// (a TorusTest state machine).  Each trip of the outer loop
//   (0) writes one CLEAR tile (constant miss record, 9 coalesced store instructions): the
//       HBM-bound part of the frame drains in the background of the VALU-bound part;
//   (A) advances the lanes: finished queries run their shader stage (miss / closest-hit /
//       shadow-miss) and spawn the next query or finish the pixel; idle lanes are compacted
//       with a ballot and refilled from the wave's LIVE tiles; tests culled by the bounding
//       sphere are skipped at once.  A round of (A) runs only for >= min_batch lanes (or
//       when nothing is in flight), so the shader/refill code never runs for a few stragglers
//       while the other lanes wait;
//   (B) runs the solve loop — every lane evaluates (f, f') of ITS test, whatever pixel,
//       depth or query kind it belongs to;
//   (C) folds the finished tests into their queries.
// Work distribution: the two tile lists are dealt round-robin to the persistent waves (wave g
// takes entries g, g+G, g+2G, …): no shared counter in the loop (one device-wide atomic word
// saturates at ≈88 dequeues/µs, MI355X_MICROARCH.md "dequeue"), and since the LIVE list is
// compact every wave gets the same number of non-trivial tiles.
enum : int { K_NONE = 0, K_CLOSEST = 1, K_SHADOW = 2 };

// Writes the constant miss record of one CLEAR macro tile (32×8 pixels) and returns the number
// of image pixels this lane wrote.  Lane l = (row r = l >> 3, q = l & 7).  Each first-hit
// stream is stored as ONE dwordx4 per lane (pixels 4q..4q+3 of row r): a wave instruction
// writes 8 full 128-B lines.  rgba takes 4 dwordx4 per lane, instruction j writing pixels
// 8j + q: again 8 full lines per instruction.  (Narrow stores are what bounds a streaming
// writer on this chip: a dword store of 8×32-B row pieces is issue-limited to ≈3 B/clk/CU.)
__device__ __forceinline__ uint32_t clear_macro(const RenderArgs& a, uint32_t tx, uint32_t ty, uint32_t lane)
{
  // The constants of the miss record are (re)materialised HERE on purpose: hoisted out of the
  // caller's tile loop they stay live across the whole solve, get spilled to scratch, and every
  // reload is a vector-memory load whose s_waitcnt drains the stream of output stores.
  // (clearColor·0.8, 1): rmiss:37 → rgen:76 with attenuation 1 and hitValue 0 → rgen:87
  float inf, zero, one;
  asm volatile("v_mov_b32 %0, 0x7f800000\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 1.0" : "=v"(inf), "=v"(zero), "=v"(one));
  const float4 c = make_float4(a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f, one);
  const uint32_t x0 = tx * 8, ly = ty * 8 + (lane >> 3), q = lane & 7;
  if(ly >= a.n_local_rows)
    return 0;
  const uint32_t y   = image_row(a, ly);
  const size_t   row = (size_t)(a.compact ? ly : y) * a.W;
  uint32_t n = 0;
  // rgba: pixel 8j + q
#pragma unroll
  for(uint32_t j = 0; j < 4; ++j)
  {
    const uint32_t x = x0 + 8 * j + q;
    if(x < a.W)
    {
      if(a.rgba) st4c(a.rgba + 4 * (row + x), c);
      ++n;
    }
  }
  // first-hit streams: pixels 4q .. 4q+3
  const uint32_t xs = x0 + 4 * q;
  if(a.vec4_ok && xs + 3 < a.W)
  {
    const float4 tv = make_float4(inf, inf, inf, inf), zv = make_float4(zero, zero, zero, zero);
    const size_t i = row + xs;
    if(a.hits.t) st4c(a.hits.t + i, tv);
    if(a.hits.px) st4c(a.hits.px + i, zv);
    if(a.hits.py) st4c(a.hits.py + i, zv);
    if(a.hits.pz) st4c(a.hits.pz + i, zv);
    if(a.hits.nx) st4c(a.hits.nx + i, zv);
    if(a.hits.ny) st4c(a.hits.ny + i, zv);
    if(a.hits.nz) st4c(a.hits.nz + i, zv);
    if(a.hits.id)
    {
      const int m = miss_id();
      st4c(a.hits.id + i, m, m, m, m);
    }
  }
  else
  {
    for(uint32_t k = 0; k < 4; ++k)
      if(xs + k < a.W)
        store_first_hit(a, row + xs + k, inf, {zero, zero, zero}, {zero, zero, zero}, miss_id());
  }
  return n;
}

template <class Real>
__global__ __launch_bounds__(256, (sizeof(Real) == 4 ? 4 : 2)) void render_persistent_kernel(const SceneK scene, const RenderArgs a_arg)
{
  __shared__ SceneK     S;
  __shared__ RenderArgs A_lds;
  stage_args(&A_lds, a_arg);
  stage_scene(&S, scene);
  const RenderArgs& a = A_lds;

  const uint32_t lane    = threadIdx.x & 63;
  const int      n_tori  = S.n_tori;
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  const uint32_t g_wave  = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  // list lengths as published by the classification, never beyond the lists' capacity
  const uint32_t n_live  = umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a.counts, (size_t)0)), a.cap_live);
  const uint32_t n_clear = umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a.counts, (size_t)1)), a.cap_clear);
  const uint32_t n_heavy = umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a.counts, (size_t)2)), n_live);   // live_slot(): they come first

  // Queue state.  Wave g owns entries g, g+G, g+2G, … of both lists.  Lane k caches the
  // wave's k-th entry of the current batch of 64 (one gather load per 64 tiles) and entries
  // are broadcast with v_readlane: the steady-state loop issues NO global loads, so no
  // s_waitcnt vmcnt ever drains the stream of output stores behind it.
  const uint32_t my_live_n  = n_live > g_wave ? (n_live - g_wave + n_waves - 1) / n_waves : 0;   // entries owned
  const uint32_t my_clear_n = n_clear > g_wave ? (n_clear - g_wave + n_waves - 1) / n_waves : 0;
  uint32_t k_live = 0, k_clear = 0;  // next owned entry (wave-uniform)
  uint32_t live_cache  = lane < my_live_n ? ld1(a.tiles_live, live_slot(a.cap_live, n_heavy, g_wave + (uint64_t)lane * n_waves)) : 0u;
  uint32_t clear_cache = lane < my_clear_n ? ld1(a.tiles_clear, g_wave + (size_t)lane * n_waves) : 0u;
  settle_loads(live_cache, clear_cache);
  bool     exhausted = my_live_n == 0;
  uint32_t cur = __builtin_amdgcn_readlane(live_cache, 0);
  uint32_t next_in_tile = 0;  // pixels of the current tile handed out

  // lane state: pixel payload (rgen:54-61)
  uint32_t px = 0, py = 0;       // pixel: x and image row
  size_t   oi = 0;               // index into rgba / first-hit streams
  int      depth = 0, done = 1;
  v3       attenuation = {1.0f, 1.0f, 1.0f}, hitValue = {0.0f, 0.0f, 0.0f};
  v3       dir_in = {0.0f, 0.0f, 0.0f};  // direction of the ray whose closest hit is being shaded
  // lane state: current query
  int   kind = K_NONE, ti = 0, best_id = -1;
  uint32_t skip_path = 0u, skip_q = 0u;   // enclosure cull (trace_pixel): the path's mask, and the current query's
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
  WorkCount wc;

  for(;;)
  {
    // ------------------------------ (0) clear tiles ----------------------------------------
    // one per trip while there is tracing to do; all of them once the wave has none left
    while(k_clear < my_clear_n)
    {
      if((k_clear & 63u) == 0 && k_clear)
      {
        clear_cache = k_clear + lane < my_clear_n ? ld1(a.tiles_clear, g_wave + (size_t)(k_clear + lane) * n_waves) : 0u;
        settle_loads(live_cache, clear_cache);
      }
      const uint32_t packed = __builtin_amdgcn_readlane(clear_cache, k_clear & 63u);
      n_primary += clear_macro(a, tile_x(packed), tile_y(packed), lane) * (uint32_t)n_tori;
      ++k_clear;
      if(!(exhausted && !__any(inflight || kind != K_NONE)))
        break;
    }

    // ------------------------------ (A) advance -----------------------------------------
    for(;;)
    {
      const bool needs = !inflight && !(kind == K_NONE && exhausted);
      const uint32_t n_needs = (uint32_t)__popcll(__ballot(needs));
      if(n_needs == 0 || (n_needs < a.min_batch && __any(inflight)))
        break;

      // A1: shader stages of finished queries
      const bool stage = needs && kind != K_NONE && (ti >= n_tori || shadow_hit);
      if(__any(stage))
      {
        bool have_prd = false, shadowed = false, do_end = false;
        v3   prdHit = {0.0f, 0.0f, 0.0f};
        if(stage && kind == K_CLOSEST)
        {
          float* rd = a.rendered ? reinterpret_cast<float*>(&a.rendered[(size_t)px * a.H + py]) : nullptr;
          if(best_id < 0)
          {
            // miss shader (REFL rmiss:37; BEF rmiss:21 hitPosition = 0)
            prdHit   = {a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f};
            have_prd = true;
            if(depth == 0)
            {
              store_first_hit(a, oi, __builtin_inff(), {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, miss_id());
              if(rd) st4(rd, make_float4(0.0f, 0.0f, 0.0f, 1.0f));
            }
          }
          else
          {
            HitState h;
            hit_begin(S, a.pc, best_id, best_t, qo, qd, h);
            if(depth == 0)                                                   // BEF rgen:94-97
            {
              store_first_hit(a, oi, best_t, h.P, h.N, best_id);
              if(rd) st4(rd, make_float4(h.P.x, h.P.y, h.P.z, 1.0f));
            }
            dir_in = qd;
            const uint32_t inside = S.inside[best_id];
            skip_q = skip_path | inside;                    // the shadow ray leaves the surface outwards (N·L > 0)
            if(dot3(h.N, qd) < 0.0f) skip_path |= inside;   // hit from outside: the reflected ray leaves outwards
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
        else if(stage)
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
            if(a.rgba) st4(a.rgba + 4 * oi, c);                              // rgen:87
            if(a.rendered) st4(reinterpret_cast<float*>(&a.rendered[(size_t)px * a.H + py]) + 4, c);
            kind = K_NONE;
          }
          else
          {
            done = 1;                                                        // rgen:84
            kind = K_CLOSEST; ti = 0; best_id = -1; best_t = __builtin_inff(); shadow_hit = false;
            skip_q = skip_path;
            q_tmax = kTMax;
            rk.set(qo, qd, kTMin, kTMax);                                    // rgen:82-83
          }
        }
      }

      // A2: compaction — idle lanes (ballot) take the next pixels of the wave's current tile
      // in order (rank among the idle lanes = mbcnt of the ballot); a drained tile is replaced
      // by the wave's next LIVE tile.
      for(;;)
      {
        const unsigned long long want = __ballot(kind == K_NONE && !exhausted);
        if(want == 0)
          break;
        const uint32_t avail = 64u - next_in_tile;
        const uint32_t rank  = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
        const uint32_t nwant = (uint32_t)__popcll(want);
        if(kind == K_NONE && rank < avail)
        {
          const uint32_t within = next_in_tile + rank;
          const uint32_t x = tile_x(cur) * 8 + (within & 7), ly = tile_y(cur) * 8 + (within >> 3);
          if(x < a.W && ly < a.n_local_rows)
          {
            px = x;
            py = image_row(a, ly);
            oi = out_index(a, x, py, ly);
            raygen(a.g, a.toro, a.W, a.H, a.camera, px, py, qo, qd);
            if(a.rendered)
            {
              float* rd = reinterpret_cast<float*>(&a.rendered[(size_t)px * a.H + py]);
              st4(rd + 8, make_float4(qo.x, qo.y, qo.z, 1.0f));
              st4(rd + 12, make_float4(qd.x, qd.y, qd.z, 0.0f));
            }
            depth = 0; done = 1;
            attenuation = {1.0f, 1.0f, 1.0f};
            hitValue    = {0.0f, 0.0f, 0.0f};
            kind = K_CLOSEST; ti = 0; best_id = -1; best_t = __builtin_inff(); shadow_hit = false;
            skip_path = skip_q = a.skip_primary;
            q_tmax = kTMax;
            rk.set(qo, qd, kTMin, kTMax);
          }
        }
        next_in_tile += nwant < avail ? nwant : avail;
        if(next_in_tile == 64u)
        {
          next_in_tile = 0;
          ++k_live;
          exhausted = k_live >= my_live_n;
          if(!exhausted)
          {
            if((k_live & 63u) == 0)
            {
              live_cache = k_live + lane < my_live_n ? ld1(a.tiles_live, live_slot(a.cap_live, n_heavy, g_wave + (uint64_t)(k_live + lane) * n_waves)) : 0u;
              settle_loads(live_cache, clear_cache);
            }
            cur = __builtin_amdgcn_readlane(live_cache, k_live & 63u);
          }
        }
      }

      // A3: set up the next test of every lane that has a query but no test
      if(!inflight && kind != K_NONE && ti < n_tori && !shadow_hit)
      {
        if(kind == K_SHADOW) ++n_shadow;
        else if(depth == 0) ++n_primary;
        else ++n_bounce;
        if((skip_q >> ti) & 1u)
          ++ti;  // a tube this ray cannot hit first (enclosure cull): counted, not traced
        else
        {
        ++wc.traced;
        // closest-hit queries end the interval of every later test at the closest hit so far
        if(tst.setup((Real)rk.ox, (Real)rk.oy, (Real)rk.oz, (Real)rk.dx, (Real)rk.dy, (Real)rk.dz, rk.dd, rk.inv_dd, (Real)rk.tmin,
                     (Real)(kind == K_CLOSEST ? min_(q_tmax, best_t) : q_tmax), torus_k<Real>(S, S.order[ti])))
        {
          inflight = true;
          ++wc.solved;
        }
        else
          ++ti;  // culled by the bounding sphere / window: this test is a miss
        }
      }
    }
    if(!__any(inflight))
    {
      if(k_clear < my_clear_n || __any(kind != K_NONE))
        continue;  // clear tiles, or stragglers waiting for a batch, are left
      break;       // both lists drained and every pixel finished
    }

    // ------------------------------ (B) solve ---------------------------------------------
    while(__any(inflight))
    {
      const bool slow = __any(inflight && !tst.iterating());
      if(inflight)
      {
        ++wc.evals;
        inflight   = slow ? tst.step() : tst.step_iter();
        unconsumed = !inflight;
      }
    }

    // ------------------------------ (C) consume -------------------------------------------
    if(unconsumed)
    {
      unconsumed = false;
      Real  tt;
      float t;
      const float tm = kind == K_CLOSEST ? min_(q_tmax, best_t) : q_tmax;   // the interval setup() used
      if(tst.finish((Real)rk.dx, (Real)rk.dy, (Real)rk.dz, (Real)rk.tmin, (Real)tm, torus_k<Real>(S, S.order[ti]), tt)
         && round_t(tt, kTMin, tm, t))
      {
        if(kind == K_SHADOW) shadow_hit = true;
        else { best_t = t; best_id = S.order[ti]; }
      }
      ++ti;
    }
  }

  if(a.stats)
  {
    block_add_stats(a.stats, n_primary, n_bounce, n_shadow, wc);
  }
}

// ------------------------------------------------------------------------------------------
// render, tile lists + static lane↔pixel mapping ("listed")
// ------------------------------------------------------------------------------------------
// After tile_classify_kernel: every wave walks its share of the LIVE list (entries g, g+G, …),
// tracing each 8×8 tile with the plain per-lane bounce loop, then writes its share of the CLEAR
// macro tiles.  The grid is several times larger than the number of resident blocks, so the
// hardware workgroup dispatcher balances the (very uneven) tile costs; the clear stores of
// blocks that finish early overlap the solve of the others.
// Timing ablations (skip the CLEAR or the LIVE part of a frame) exist in -DTRT_TUNING builds only:
// the release library has no switch that returns an incomplete image with TRT_OK.
#ifdef TRT_TUNING
#define TRT_SKIP(a, bit) ((a).debug_skip & (bit))
#else
#define TRT_SKIP(a, bit) false
#endif
constexpr uint32_t kListedThreads = 256;   // block size of the listed kernel: stage_block256() and the launcher rely on it
#ifndef TRT_LISTED_WAVES
#define TRT_LISTED_WAVES 6
#endif
#ifndef TRT_LISTED_WAVES_F64
#define TRT_LISTED_WAVES_F64 4
#endif
// Waves per SIMD the register allocation aims at: 6 for the plain FP32 kernel (80 VGPRs, no scratch: LIVE part
// −3.6 %, eight nested tori FP32 −4.5 % against 5 waves), 4 for FP64.  The counted (STATS) instantiations carry six
// counters per lane and run only in the untimed counted pass, the RD instantiations stage RenderedData through LDS:
// each gets one wave less instead of scratch.
// RD: the launch exports RenderedData (a.rendered != nullptr); every wave then owns a 4-KB LDS image.
// FB: the launch takes part in the cost feedback (RenderArgs::tile_cost): heavy tiles first, every traced tile timed.
// BATCH: the launch renders up to kMaxBatch frames (RenderBatch, trt_render_batch_dev): every list entry names its frame,
// whose arguments the wave takes from the block's LDS copy of the batch.  The single-frame instantiations are the code they
// were before batches existed.
template <class Real, bool STATS, bool DK, bool RD, bool FB = false, bool BATCH = false>
__global__ __launch_bounds__(256, (DK ? 2 : (sizeof(Real) == 4 ? TRT_LISTED_WAVES : TRT_LISTED_WAVES_F64) - (STATS ? 1 : 0) - (RD ? 1 : 0))) void render_listed_kernel(const SceneK scene, const typename LaunchArgs<BATCH>::type args)
{
  static_assert(!(BATCH && (RD || DK)), "batches: default solver, no RenderedData");
  __shared__ SceneK     S;
  __shared__ RenderArgs A_lds[BATCH ? kMaxBatch : 1];
  __shared__ float4     rd_images[RD ? 4 : 1][RD ? 256 : 1];
  float4* const rd_tile = RD ? rd_images[threadIdx.x >> 6] : nullptr;
  const RenderArgs& a_arg = frame_args(args, 0u);   // lists, counters and capacities are the same in every frame of a batch
  TRT_STAMP(0, wall_clock64());
  TRT_STAMP(3, (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32));   // HW_ID, XCC_ID
  // list lengths as published by the classification, never beyond the lists' capacity; read through the kernel
  // arguments BEFORE anything is staged: a block none of whose waves owns an entry leaves at once (the grid is sized
  // for the worst case, one wave per four tiles; ≈15 % of the baseline frame's blocks own nothing)
  const uint32_t n_live  = umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a_arg.counts, (size_t)0)), a_arg.cap_live);
  const uint32_t n_clear = umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a_arg.counts, (size_t)1)), a_arg.cap_clear);
  const uint32_t n_heavy = FB ? umin((uint32_t)__builtin_amdgcn_readfirstlane(ld1(a_arg.counts, (size_t)2)), n_live) : 0u;   // they come first
  if(blockIdx.x * (kListedThreads / 64u) >= (n_live > n_clear ? n_live : n_clear))
  {
    TRT_STAMP(2, wall_clock64());
    return;
  }
  if constexpr(BATCH)
  {
    // the frames' arguments in use (124 dwords each) with every thread, the scene with the upper half of the block
    constexpr uint32_t NA = sizeof(RenderArgs) / 4;
    const uint32_t  n_words = args.n_frames * NA;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&args.fr[0]);
    uint32_t*       dst = reinterpret_cast<uint32_t*>(&A_lds[0]);
#pragma unroll 1
    for(uint32_t i = threadIdx.x; i < n_words; i += kListedThreads)
      dst[i] = src[i];
    if(threadIdx.x >= 128u)
    {
      const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(&scene);
      uint32_t*       sdst = reinterpret_cast<uint32_t*>(&S);
      const uint32_t  c4 = scene_words(scene);
#pragma unroll 1
      for(uint32_t i = threadIdx.x - 128u; i < c4; i += 128u)
      {
        const uint32_t off = scene_word(scene, i);
        sdst[off] = ssrc[off];
      }
    }
    __syncthreads();
  }
  else
    stage_block256(&S, &A_lds[0], scene, args);
  TRT_STAMP(5, wall_clock64());
  const RenderArgs& a0 = A_lds[0];
  const uint32_t lane    = threadIdx.x & 63;
  const uint32_t n_waves = gridDim.x * (kListedThreads / 64u);
  const uint32_t g_wave  = __builtin_amdgcn_readfirstlane(blockIdx.x * (kListedThreads / 64u) + (threadIdx.x >> 6));
  uint32_t n_primary = 0, n_bounce = 0, n_shadow = 0;
  WorkCount wc;
  typedef TileCode<BATCH> TC;

  // Wave g owns entries g, g+G, g+2G, … of both lists.  Lane k prefetches the wave's k-th
  // entry of the current batch of 64 (one gather load per list per batch) and entries are
  // broadcast with v_readlane, so no load — and hence no s_waitcnt vmcnt that would drain the
  // output stores — sits between the tiles.  Clear macro tiles (pure stores) are interleaved
  // with the traced tiles: the HBM-bound half of the frame drains behind the VALU-bound half.
  // Measured alternatives (4096², one process, interleaved rounds): dealing the CLEAR entries
  // only over the waves that own a LIVE tile (no store-only tail of the grid) +21 %; tracing
  // first and clearing afterwards +5 %, on odd waves only +4 %, on odd blocks only +2 %.
  // (entry i of this wave is list entry g_wave + i·n_waves: owned while that index is below the list's length —
  // compared, not divided: the two integer divisions for the entry counts were 70 instructions of every wave)
  uint32_t live_cache = 0, clear_cache = 0;
  for(uint32_t i = 0;; ++i)
  {
    const uint32_t entry = g_wave + i * n_waves;   // < 2^32: the lists hold fewer than 2^31 entries and n_waves <= their capacity
    const bool own_live = entry < n_live, own_clear = entry < n_clear;
    if(!own_live && !own_clear)
      break;
    if((i & 63u) == 0)
    {
      const uint64_t e = entry + (uint64_t)lane * n_waves;
      live_cache  = e < n_live ? ld1(a0.tiles_live, FB ? live_slot(a0.cap_live, n_heavy, e) : (size_t)e) : 0u;
      clear_cache = e < n_clear ? ld1(a0.tiles_clear, (size_t)e) : 0u;
      settle_loads(live_cache, clear_cache);
      if(i == 0) TRT_STAMP(1, wall_clock64());
    }
    // lane-derived values (lane & 7, lane >> 3, …) are recomputed per trip from an opaque copy:
    // hoisted out of the loop they would be spilled, and a spill reload is a vector-memory load
    uint32_t ln = lane;
    asm volatile("" : "+v"(ln));
    if(own_clear && !TRT_SKIP(a0, 1u))
    {
      const uint32_t cpacked = __builtin_amdgcn_readlane(clear_cache, i & 63u);
      const RenderArgs& a = A_lds[TC::frame(cpacked)];
      n_primary += clear_macro(a, TC::x(cpacked), TC::y(cpacked), ln) * (uint32_t)S.n_tori;
      if(RD)   // the four 8×8 tiles of the macro tile: primary rays + miss record, no solve
        for(uint32_t j = 0; j < kMacroTiles; ++j)
          rd_miss_tile(a, rd_tile, TC::x(cpacked) + j, TC::y(cpacked), ln);
    }
    if(own_live && !TRT_SKIP(a0, 2u))
    {
      const uint32_t packed = __builtin_amdgcn_readlane(live_cache, i & 63u);
      const RenderArgs& a = A_lds[TC::frame(packed)];
      const unsigned long long tile_t0 = FB ? wall_clock64() : 0ull;
      if(i == 0) TRT_STAMP(4, 0x100000000ull | packed);
      const uint32_t x = TC::x(packed) * 8 + (ln & 7), ly = TC::y(packed) * 8 + (ln >> 3);
      if(RD && (packed & kTileMissFlag))
        rd_miss_tile(a, rd_tile, TC::x(packed), TC::y(packed), ln);
      if(x < a.W && ly < a.n_local_rows)
      {
        if(packed & kTileMissFlag)
        {
          // classified "every ray of this tile misses": the miss record of trace_pixel, no tracing
          // (rmiss:37 → rgen:76 with attenuation 1 → rgen:87; BEF rmiss:21)
          const size_t oi = out_index(a, x, image_row(a, ly), ly);
          store_first_hit(a, oi, __builtin_inff(), {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, miss_id());
          if(a.rgba)
            st4(a.rgba + 4 * oi, make_float4(a.pc.clearColor[0] * 0.8f, a.pc.clearColor[1] * 0.8f, a.pc.clearColor[2] * 0.8f, 1.0f));
          n_primary += (uint32_t)S.n_tori;
        }
        else
          trace_pixel<Real, DK>(S, a, x, image_row(a, ly), ly, RdSink{RD ? rd_tile + rd_unit(ln & 7, ln >> 3, 0) : nullptr, nullptr},
                                n_primary, n_bounce, n_shadow, wc);
      }
      // cost feedback (RenderArgs::tile_cost): what this wave spent on the tile, kept per macro tile as the maximum over its tiles
      if(FB && !(packed & kTileMissFlag))
      {
        const uint32_t ticks = (uint32_t)(wall_clock64() - tile_t0);
        if(ln == 0u)
          atomicMax(&a.tile_cost[TC::y(packed) * ((((a.W + 7u) >> 3) + kMacroTiles - 1u) / kMacroTiles) + TC::x(packed) / kMacroTiles], ticks ? ticks : 1u);
      }
      if(RD && !(packed & kTileMissFlag))
        rd_flush(a, rd_tile, TC::x(packed), TC::y(packed), ln);
    }
  }
  TRT_STAMP(2, wall_clock64());
  if(STATS && a0.stats)   // STATS = false: the counters are dead code (their VGPRs and increments vanish)
  {
    block_add_stats(a0.stats, n_primary, n_bounce, n_shadow, wc);
  }
}

// ------------------------------------------------------------------------------------------
// post pass (tonemap): streaming, one pixel per lane, 16 B in, 16 B and/or 4 B out
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void post_pixel(float4 c, uint64_t i, float4* __restrict__ f32_out, uint32_t* __restrict__ u8_out)
{
  // the framebuffer's alpha is 1 (rgen:87) and post_gamma(1) is exactly 1 (log2_poly(1) = 0,
  // exp2_poly(0) = 1): when the whole wave sees alpha 1 the fourth pow is skipped
  const float ow = __all(c.w == 1.0f) ? 1.0f : post_gamma(c.w);
  // a wave whose 64 pixels are all grey (r = g = b: the clear colour of the baseline frame, 85 % of its pixels) computes ONE
  // pow per pixel instead of three — the same bits, post_gamma being a function of its argument alone
  const bool  grey = __all(c.x == c.y && c.y == c.z);
  const float ox = post_gamma(c.x);
  float oy = ox, oz = ox;
  if(!grey)
  {
    oy = post_gamma(c.y);
    oz = post_gamma(c.z);
  }
  const float4 o = make_float4(ox, oy, oz, ow);
  if(f32_out) f32_out[i] = o;
  if(u8_out)
  {
    // UNORM8: round-to-nearest-even of clamp(o, 0, 1)·255 (v_rndne via rintf), R in the low byte
    const uint32_t r = (uint32_t)rintf(min_(max_(o.x, 0.0f), 1.0f) * 255.0f);
    const uint32_t g = (uint32_t)rintf(min_(max_(o.y, 0.0f), 1.0f) * 255.0f);
    const uint32_t b = (uint32_t)rintf(min_(max_(o.z, 0.0f), 1.0f) * 255.0f);
    const uint32_t a = (uint32_t)rintf(min_(max_(o.w, 0.0f), 1.0f) * 255.0f);
    u8_out[i] = r | (g << 8) | (b << 16) | (a << 24);
  }
}

// (Non-temporal loads and stores here: within noise, 0.088 against 0.089 ms — the → rgba8 pass is bound by the ≈200
// VALU instructions per pixel of the exact-operation pow, → f32 takes the same time with 1.6× the bytes.)
// Four pixels per lane and trip, their loads issued together: with one 16-B load in flight per
// lane the pass was bound by memory latency (32 KB in flight per CU ≈ 2.8 TB/s of reads), not by
// the ≈55 VALU instructions per channel.
__global__ __launch_bounds__(256) void post_kernel(const float4* __restrict__ in, uint64_t n, float4* __restrict__ f32_out,
                                                   uint32_t* __restrict__ u8_out)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for(; i + 3 * stride < n; i += 4 * stride)
  {
    const float4 c0 = in[i], c1 = in[i + stride], c2 = in[i + 2 * stride], c3 = in[i + 3 * stride];
    post_pixel(c0, i, f32_out, u8_out);
    post_pixel(c1, i + stride, f32_out, u8_out);
    post_pixel(c2, i + 2 * stride, f32_out, u8_out);
    post_pixel(c3, i + 3 * stride, f32_out, u8_out);
  }
  for(; i < n; i += stride)
    post_pixel(in[i], i, f32_out, u8_out);
}

hipError_t launch_post(const float* in, uint64_t n, float* f32_out, uint8_t* u8_out, int n_cus, const Tuning& tn,
                       hipStream_t stream)
{
  if(n == 0) return hipSuccess;
  // many short blocks (one pixel per lane up to 16.8 M pixels): measured 0.074 / 0.087 ms for
  // → rgba8 / → f32 at 4096² against 0.097 / 0.112 ms with 4–16 long-running blocks per CU
  uint64_t want = (n + 255) / 256, cap = (uint64_t)n_cus * 256;
  if(tn.post_blocks_per_cu) cap = (uint64_t)n_cus * tn.post_blocks_per_cu;
  if(cap == 0) cap = 1;
  hipLaunchKernelGGL(post_kernel, dim3((uint32_t)(want < cap ? want : cap)), dim3(256), 0, stream,
                     reinterpret_cast<const float4*>(in), n, reinterpret_cast<float4*>(f32_out),
                     reinterpret_cast<uint32_t*>(u8_out));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// point-cloud re-projection (SEC/): z-buffered point splatting with 64-bit atomic keys
// ------------------------------------------------------------------------------------------
// key = depth24 << 32 | point index.  atomicMin over the keys of a pixel yields the smallest
// depth and, among equal depths, the smallest index — exactly what in-order rasterisation
// with depth test LESS produces.  A cleared pixel holds 0x00FFFFFF'00000000, which no fragment
// can beat unless its depth is < 1.0 (LESS).  HBM-bound integer work: 32 B read per point,
// 8-B atomics on its ≈6 pixels, one 8-B read + 16-B write per pixel in the resolve.
constexpr unsigned long long kSplatClear = 0x00FFFFFFull << 32;

__global__ __launch_bounds__(256) void splat_clear_kernel(unsigned long long* keys, uint64_t n)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    keys[i] = kSplatClear;
}

struct SplatArgs {
  float    vp[16];
  uint32_t W, H;
  float    half;   // point_size / 2
};

// Vertex stage + point rasterisation set-up of one point: false when the vertex is clipped or the point
// covers no pixel centre; else its 24-bit depth and the pixel rectangle [x0,x1) × [y0,y1) it covers.
__device__ __forceinline__ bool splat_project(float4 p, const SplatArgs& a, uint32_t& z24, int& x0, int& x1, int& y0, int& y1)
{
  // gl_Position = uni.viewProj * vec4(position, 1.0)   (SEC vert_shader.vert:51)
  const float cx = fma_(a.vp[12], 1.0f, fma_(a.vp[8], p.z, fma_(a.vp[4], p.y, a.vp[0] * p.x)));
  const float cy = fma_(a.vp[13], 1.0f, fma_(a.vp[9], p.z, fma_(a.vp[5], p.y, a.vp[1] * p.x)));
  const float cz = fma_(a.vp[14], 1.0f, fma_(a.vp[10], p.z, fma_(a.vp[6], p.y, a.vp[2] * p.x)));
  const float cw = fma_(a.vp[15], 1.0f, fma_(a.vp[11], p.z, fma_(a.vp[7], p.y, a.vp[3] * p.x)));
  if(!(cw > 0.0f && cx >= -cw && cx <= cw && cy >= -cw && cy <= cw && cz >= 0.0f && cz <= cw))
    return false;  // point clipping: the vertex is outside the view volume (NaN lands here too)
  const float iw = 1.0f / cw;
  const float xf = fma_(cx * iw, 0.5f, 0.5f) * (float)a.W;
  const float yf = fma_(cy * iw, 0.5f, 0.5f) * (float)a.H;
  z24 = (uint32_t)rintf((cz * iw) * 16777215.0f);
  // pixel centres c = j + 0.5 with lo <= c < hi  ⇔  j in [ceil(lo - 0.5), ceil(hi - 0.5))
  x0 = max((int)ceilf(xf - a.half - 0.5f), 0); x1 = min((int)ceilf(xf + a.half - 0.5f), (int)a.W);
  y0 = max((int)ceilf(yf - a.half - 0.5f), 0); y1 = min((int)ceilf(yf + a.half - 0.5f), (int)a.H);
  return x0 < x1 && y0 < y1;
}

__global__ __launch_bounds__(256) void splat_points_kernel(const trt_point* __restrict__ pts, uint64_t n, const SplatArgs a,
                                                           unsigned long long* keys)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(!splat_project(reinterpret_cast<const float4*>(pts)[2 * i], a, z24, x0, x1, y0, y1))
      continue;
    const unsigned long long key = ((unsigned long long)z24 << 32) | (unsigned long long)(uint32_t)i;
    for(int y = y0; y < y1; ++y)
      for(int x = x0; x < x1; ++x)
        atomicMin(&keys[(size_t)y * a.W + x], key);
  }
}

// ---- binned re-projection: the depth test in LDS -------------------------------------------------
// The one-pass form above sends ≈6 64-bit atomicMin per point to the memory side (≈26 G atomics/s chip-wide:
// 2 ms for 8.4 M random points).  Here the screen is cut into bins of 128 × 64 pixels — 8,192 keys = 64 KB, one
// LDS image — and the points are first sorted by bin:
//   count    every block projects its chunk of points ONCE — the 8-B result {depth24, pixel rectangle} goes to a
//            side array — and histograms it over the bins in LDS, one global add per non-empty bin; the block that
//            finishes LAST (a ticket) turns the bin counts into their exclusive prefix (no scan launch);
//   scatter  the chunks again, from the side array, ONE sweep: a block histograms its 4,096 points in LDS, reserves
//            one range per bin (one global add), SORTS its records by bin in an LDS staging area and writes them out
//            in sorted order — a wave's store instruction covers 768 contiguous bytes of a few bin runs instead of
//            64 records at 64 unrelated addresses (round 2: 16-B records written one by one, each 32-B sector of
//            HBM written twice — 273 MB of writes for 134 MB of records);
//   resolve  ONE block per bin: atomicMin of the keys in LDS (ds_min_u64), then every pixel of the bin is written
//            once — colour of the winning point, or the clear colour.
// Records are 12 B: {point index, depth24, rectangle inside the bin (4 × 7 bits)}.
// The keys, and so the image, are those of the one-pass form bit for bit (a minimum does not depend on the
// order of its operands).  A point wider than a bin edge would need more than 4 records: such sizes, and images
// with more than kSplatMaxBins bins, take the one-pass form.
constexpr uint32_t kBinW = 128, kBinH = 64, kSplatMaxBins = 8192, kSplatChunk = 8192, kSplatMaxDim = 16383;
constexpr uint32_t kSplatTicketWord = 5 * kSplatMaxBins;   // bin_words[…]: blocks of `count` that have finished (zero between calls)

struct SplatRec { uint32_t idx, z, rect; };   // rect = rx0 | rx1 << 7 | ry0 << 15 | ry1 << 21  (bin-relative, half-open)

struct SplatBins {
  uint32_t  bins_x, bins_y, n_bins;
  uint32_t* count;    // [n_bins]   points per bin: accumulated by `count`, zeroed again by `resolve`
  uint32_t* offset;   // [n_bins]   first record of the bin (written by the last block of `count`)
  uint32_t* cursor;   // [n_bins]   records handed out so far (zeroed with the offsets)
  uint32_t* ticket;   // blocks of `count` that have finished
  uint32_t* table;    // [chunks of 4,096 points][n_bins] (sorted scatter only): where in its bin's range a chunk's records start —
                      // what `count`'s returning add on the bin's count word returned; nullptr: `scatter` draws from `cursor`
  SplatRec* records;  // [<= 4 · n_points]
  uint2*    proj;     // [n_points] the projected points (count → scatter)
  uint32_t  debug_skip;   // -DTRT_TUNING builds only (timing ablations of `resolve`: 16 no colour gather, 32 no depth test, 64 no record loads)
  // paged scatter (splat_bin_kernel): `records` is cut into pages of 1 << page_shift records; page k < n_bins is bin k's
  // first page, the pages behind them are handed out from a pool as bins fill up
  unsigned long long* state;  // [n_bins] fill (24 bits) | epoch (12) | handle of the current page (14) | handle of the next one (14): the bin's two-page window (splat_bin_kernel); handle 0 = the bin's own page, h > 0 = pool page h - 1; zero between calls
  uint32_t* pool;             // pool pages handed out (zero between calls)
  uint32_t* page_bin;         // [pool_pages] the bin a pool page belongs to
  uint32_t* rticket;          // blocks of `resolve` that have finished (zero between calls)
  uint32_t* page_seq;         // [pool_pages] … and which of that bin's pages it is (abstract page number, 12 bits)
  uint32_t  page_shift, pool_pages;   // a page holds 1 << page_shift records; pages 0 .. 2·n_bins-1 are the bins' own two
};

// projected point in 64 bits: x0 (14) | y0 (14) | width low 4 bits ; depth24 | width high 2 bits | height (6).
// W, H <= 16383 and point_size <= 32 (width, height <= 33) on this path; width 0 = the point draws nothing.
__device__ __forceinline__ uint2 splat_pack(uint32_t z24, int x0, int x1, int y0, int y1)
{
  const uint32_t w = (uint32_t)(x1 - x0), h = (uint32_t)(y1 - y0);
  return make_uint2((uint32_t)x0 | ((uint32_t)y0 << 14) | ((w & 15u) << 28), z24 | ((w >> 4) << 24) | (h << 26));
}
__device__ __forceinline__ bool splat_unpack(uint2 p, uint32_t& z24, int& x0, int& x1, int& y0, int& y1)
{
  const uint32_t w = (p.x >> 28) | (((p.y >> 24) & 3u) << 4), h = p.y >> 26;
  x0 = (int)(p.x & 16383u); y0 = (int)((p.x >> 14) & 16383u); x1 = x0 + (int)w; y1 = y0 + (int)h;
  z24 = p.y & 0xffffffu;
  return w != 0u;
}
__device__ __forceinline__ uint32_t splat_rect(int rx0, int rx1, int ry0, int ry1)
{
  return (uint32_t)rx0 | ((uint32_t)rx1 << 7) | ((uint32_t)ry0 << 15) | ((uint32_t)ry1 << 21);
}

// calls f(bin, x0r, x1r, y0r, y1r) for every bin the rectangle touches, rectangle clipped to the bin, bin-relative
template <class F>
__device__ __forceinline__ void splat_for_bins(const SplatBins& b, int x0, int x1, int y0, int y1, F f)
{
  const int bx0 = x0 / (int)kBinW, bx1 = (x1 - 1) / (int)kBinW, by0 = y0 / (int)kBinH, by1 = (y1 - 1) / (int)kBinH;
  for(int by = by0; by <= by1; ++by)
    for(int bx = bx0; bx <= bx1; ++bx)
    {
      const int ox = bx * (int)kBinW, oy = by * (int)kBinH;
      f((uint32_t)by * b.bins_x + (uint32_t)bx, max(x0 - ox, 0), min(x1 - ox, (int)kBinW), max(y0 - oy, 0), min(y1 - oy, (int)kBinH));
    }
}

// M sub-chunks of SUB points per block, 256 threads each (blockDim.x = M·256).  The direct scatter (M = 1, SUB = kSplatChunk)
// only needs the bins' totals.  The sorted scatter (TABLE: SUB = its chunk of 4,096 points, M = 4) also needs to know where in
// a bin's range each of ITS chunks starts: the block keeps one histogram per sub-chunk, adds their sum to the bin's count word
// with ONE returning atomic — what comes back is the number of records earlier blocks have claimed — and writes the sub-chunks'
// starts to table[chunk][bin]; the scatter then draws from no cursor at all (one million returning atomics less per call).
template <uint32_t SUB, uint32_t M, bool TABLE>
__global__ __launch_bounds__(M * 256) void splat_count_kernel(const trt_point* __restrict__ pts, uint64_t n, const SplatArgs a, const SplatBins b)
{
  extern __shared__ uint32_t hist_all[];   // [M][n_bins]
  __shared__ uint32_t part[256];
  __shared__ int      is_last;
  for(uint32_t k = threadIdx.x; k < M * b.n_bins; k += M * 256u) hist_all[k] = 0u;
  __syncthreads();
  const uint32_t m = threadIdx.x >> 8, t = threadIdx.x & 255u;
  uint32_t* hist = hist_all + m * b.n_bins;
  const uint64_t i0 = ((uint64_t)blockIdx.x * M + m) * SUB, i1 = min(n, i0 + SUB);
  // four points per lane and trip, their loads issued together (one 16-B load in flight per lane was latency-bound)
  constexpr uint32_t U = 4;
  for(uint64_t ib = i0 + t; ib < i1; ib += U * 256u)
  {
    float4 p[U];
#pragma unroll
    for(uint32_t u = 0; u < U; ++u)
    {
      const uint64_t i = ib + u * 256u;
      p[u] = i < i1 ? reinterpret_cast<const float4*>(pts)[2 * i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for(uint32_t u = 0; u < U; ++u)
    {
      const uint64_t i = ib + u * 256u;
      if(i >= i1) continue;
      uint32_t z24;
      int x0, x1, y0, y1;
      uint2 pk = make_uint2(0u, 0u);
      if(splat_project(p[u], a, z24, x0, x1, y0, y1))
      {
        pk = splat_pack(z24, x0, x1, y0, y1);
        splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int, int, int, int) { atomicAdd(&hist[bin], 1u); });
      }
      b.proj[i] = pk;
    }
  }
  __syncthreads();
  // RETURNING adds: their results are back — the additions performed, at device scope — before the barrier in front of the
  // block's ticket.  (No __threadfence(): an agent-scope release writes back the XCD's whole L2, the chunk's freshly written
  // side array included — measured: the call 0.32 → 0.43 ms.)
  uint32_t sink = 0;
  for(uint32_t k = threadIdx.x; k < b.n_bins; k += M * 256u)
  {
    uint32_t c[M], total = 0;
#pragma unroll
    for(uint32_t j = 0; j < M; ++j) { c[j] = hist_all[j * b.n_bins + k]; total += c[j]; }
    uint32_t run = total ? atomicAdd(&b.count[k], total) : 0u;
    sink |= run;
    if(TABLE)
    {
#pragma unroll
      for(uint32_t j = 0; j < M; ++j)
      {
        b.table[((size_t)blockIdx.x * M + j) * b.n_bins + k] = run;
        run += c[j];
      }
    }
  }
  if(sink == 0xffffffffu) part[threadIdx.x & 255u] = sink;   // (never true: a bin holds fewer than 2^32 records) keeps the results live
  // the block that finishes last — every other block's additions are then performed — turns the counts into offsets
  __syncthreads();
  if(threadIdx.x == 0)
    is_last = atomicAdd(b.ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if(!is_last || threadIdx.x >= 256u)
    return;
  const uint32_t per = (b.n_bins + 255u) / 256u, k0 = threadIdx.x * per, k1 = min(b.n_bins, k0 + per);
  uint32_t sum = 0;
  for(uint32_t k = k0; k < k1; ++k) sum += __hip_atomic_load(&b.count[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  part[threadIdx.x] = sum;
  __syncthreads();   // (only the block's first 256 threads are left: four whole waves)
  if(threadIdx.x == 0)
  {
    uint32_t run = 0;
    for(uint32_t q = 0; q < 256u; ++q) { const uint32_t c = part[q]; part[q] = run; run += c; }
    *b.ticket = 0u;   // the next call counts its blocks from zero
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for(uint32_t k = k0; k < k1; ++k)
  {
    b.offset[k] = run;
    b.cursor[k] = 0u;
    run += __hip_atomic_load(&b.count[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// scatter, sorted: up to kSortBins bins (2048: images up to 4096² and beyond), 4,096 points per block of 512 threads,
// 12-B records staged in LDS in bin order.  LDS: 3 words per bin + 12 B per staged record = 78 KB: two blocks per CU.
constexpr uint32_t kSortBins = 2048, kSortChunk = 4096, kSortThreads = 512, kSortPer = kSortChunk / kSortThreads, kSortStage = 4608;
template <uint32_t NB>   // bins the block's LDS arrays hold: 512 (images up to 2048²: 61 KB of LDS) or kSortBins (79 KB)
__global__ __launch_bounds__(kSortThreads) void splat_scatter_sorted_kernel(uint64_t n, const SplatBins b)
{
  __shared__ uint32_t hist[NB];    // points of this block per bin, then the rank counter
  __shared__ uint32_t lbase[NB];   // first staged record of the bin
  __shared__ uint32_t gbase[NB];   // first global record of this block's range in the bin
  __shared__ SplatRec stage[kSortStage];  // records in bin order; the bin rides in the spare bits (z: 31..24, rect: 30..28)
  __shared__ uint32_t wsum[kSortThreads / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  for(uint32_t k = tid; k < b.n_bins; k += kSortThreads) hist[k] = 0u;
  __syncthreads();
  const uint64_t i0 = (uint64_t)blockIdx.x * kSortChunk, i1 = min(n, i0 + kSortChunk);
  uint2 pk[kSortPer];
#pragma unroll
  for(uint32_t u = 0; u < kSortPer; ++u)
  {
    const uint64_t i = i0 + u * kSortThreads + tid;
    pk[u] = i < i1 ? b.proj[i] : make_uint2(0u, 0u);
  }
#pragma unroll
  for(uint32_t u = 0; u < kSortPer; ++u)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(splat_unpack(pk[u], z24, x0, x1, y0, y1))
      splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int, int, int, int) { atomicAdd(&hist[bin], 1u); });
  }
  __syncthreads();
  // exclusive prefix of the block's bin counts (the staging order) + one global reservation per non-empty bin
  constexpr uint32_t kPerT = NB / kSortThreads;   // 1 or 4 consecutive bins per thread
  uint32_t c[kPerT], mine = 0;
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
  {
    const uint32_t k = tid * kPerT + j;
    c[j] = k < b.n_bins ? hist[k] : 0u;
    mine += c[j];
  }
  uint32_t inc = mine;
#pragma unroll
  for(int off = 1; off < 64; off <<= 1)
  {
    const uint32_t v = __shfl_up(inc, off, 64);
    if(lane >= (uint32_t)off) inc += v;
  }
  if(lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = inc - mine;
  for(uint32_t w = 0; w < wave; ++w) before += wsum[w];
  uint32_t total = 0;
  for(uint32_t w = 0; w < kSortThreads / 64; ++w) total += wsum[w];
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
  {
    const uint32_t k = tid * kPerT + j;
    if(k < b.n_bins)
    {
      lbase[k] = before;
      gbase[k] = c[j] ? b.offset[k] + b.table[(size_t)blockIdx.x * b.n_bins + k] : 0u;   // (this chunk's start in the bin: `count`)
      hist[k]  = 0u;
      before += c[j];
    }
  }
  __syncthreads();
  // place: rank inside the bin from the LDS counter; records beyond the staging area (a chunk whose points straddle
  // many bins) go straight to their place in global memory
#pragma unroll
  for(uint32_t u = 0; u < kSortPer; ++u)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(splat_unpack(pk[u], z24, x0, x1, y0, y1))
    {
      const uint32_t idx = (uint32_t)(i0 + u * kSortThreads + tid);
      splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int rx0, int rx1, int ry0, int ry1) {
        const uint32_t rank = atomicAdd(&hist[bin], 1u), slot = lbase[bin] + rank, rect = splat_rect(rx0, rx1, ry0, ry1);
        if(slot < kSortStage)
          stage[slot] = SplatRec{idx, z24 | (bin << 24), rect | ((bin >> 8) << 28)};
        else
          b.records[gbase[bin] + rank] = SplatRec{idx, z24, rect};
      });
    }
  }
  __syncthreads();
  const uint32_t n_staged = umin(total, kSortStage);
  for(uint32_t j = tid; j < n_staged; j += kSortThreads)
  {
    const SplatRec r = stage[j];
    const uint32_t bin = (r.z >> 24) | ((r.rect >> 28) << 8);
    b.records[gbase[bin] + (j - lbase[bin])] = SplatRec{r.idx, r.z & 0xffffffu, r.rect & 0x0fffffffu};
  }
}

// ---- paged scatter: count + scatter in ONE sweep over the points (default for up to kSortBins bins) -----------------
// The two-pass form reads the cloud to count (268 MB for 8.4 M points), writes the 8-B projections, reads them back and
// only then knows where a bin's records go.  Here nobody needs to know: the record area is cut into PAGES of S records,
// every bin owns two of them and takes more from a pool as it fills up, and a 64-bit word per bin says where the next
// record goes (the two-page window, in the kernel).  A block projects its 4,096 points, sorts their records by bin in LDS
// (as the sorted scatter does) and reserves a bin's run with ONE returning 64-bit add of its length.  Every page of a bin
// except the last two is completely filled and a pool page carries a note {bin, which page of the bin}, so `resolve`
// needs no page table: it picks its bin's pages out of the notes (a few thousand words) and reads the fill of the last
// two from the bin's word.  One launch less, no side array, no table: 8.4 M random points 0.215 -> 0.19 ms, a captured
// cloud in capture order 0.134 -> 0.095 ms, 873 -> 728 MB per call.
constexpr uint32_t kPageSpins = 1u << 22;
constexpr unsigned long long kPagePoison = (0x3fffull << 50) | (0x3fffull << 36) | (0xfffull << 24);   // both handles all ones: adds to `fill` leave it recognisable
template <uint32_t NB>   // bins the block's LDS arrays hold: 512 (images up to 2048²) or kSortBins
__global__ __launch_bounds__(kSortThreads, 4) void splat_bin_kernel(const trt_point* __restrict__ pts, uint64_t n, const SplatArgs a, const SplatBins b)
{
  constexpr uint32_t kStage = NB <= 512u ? 4608u : 3968u;   // 65 KB / 79 KB of LDS: two blocks per CU either way
  __shared__ uint32_t hist[NB];     // points of this block per bin, then the rank counter
  __shared__ uint32_t lfirst[NB];   // first staged record of the bin (bits 0..15) | records of its run in the first page (16..28)
  __shared__ uint32_t g0[NB];       // where the run starts (record index); ~0: the run is dropped
  __shared__ uint32_t g1[NB];       // where it goes on behind the page end
  __shared__ SplatRec stage[kStage];
  __shared__ uint32_t wsum[kSortThreads / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // The block walks the chunks blockIdx, blockIdx + gridDim, …; the points of its NEXT chunk are loaded (8 x 16 B per
  // lane) before the current one is sorted, so that the reads run behind the LDS phases — two blocks per CU leave too few
  // waves to hide them otherwise (one chunk per block: 110 µs for the 8.4 M-point cloud).
  const uint64_t n_chunks = (n + kSortChunk - 1) / kSortChunk;
  float4 pn[kSortPer];
  auto fetch = [&](uint64_t chunk) {
    const uint64_t f0 = chunk * kSortChunk, f1 = min(n, f0 + kSortChunk);
#pragma unroll
    for(uint32_t u = 0; u < kSortPer; ++u)
    {
      const uint64_t i = f0 + u * kSortThreads + tid;
      pn[u] = chunk < n_chunks && i < f1 ? reinterpret_cast<const float4*>(pts)[2 * i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  constexpr bool kPrefetch = NB <= 512u;   // (the wider instantiation has four bins per thread to keep: prefetching would spill)
  if(kPrefetch) fetch(blockIdx.x);
  for(uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x)
  {
  if(!kPrefetch) fetch(chunk);
  for(uint32_t k = tid; k < b.n_bins; k += kSortThreads) hist[k] = 0u;
  __syncthreads();   // (also: the previous chunk's write-out has read the staging area and the bins' places)
  const uint64_t i0 = chunk * kSortChunk, i1 = min(n, i0 + kSortChunk);
  uint2 pk[kSortPer];
  {
    float4 p[kSortPer];
#pragma unroll
    for(uint32_t u = 0; u < kSortPer; ++u) p[u] = pn[u];
    if(kPrefetch) fetch(chunk + gridDim.x);
#pragma unroll
    for(uint32_t u = 0; u < kSortPer; ++u)
    {
      uint32_t z24;
      int x0, x1, y0, y1;
      pk[u] = make_uint2(0u, 0u);
      if(i0 + u * kSortThreads + tid < i1 && splat_project(p[u], a, z24, x0, x1, y0, y1))
      {
        pk[u] = splat_pack(z24, x0, x1, y0, y1);
        splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int, int, int, int) { atomicAdd(&hist[bin], 1u); });
      }
    }
  }
  __syncthreads();
  // exclusive prefix of the block's bin counts (the staging order)
  constexpr uint32_t kPerT = NB / kSortThreads;   // 1 or 4 consecutive bins per thread
  uint32_t c[kPerT], mine = 0;
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
  {
    const uint32_t k = tid * kPerT + j;
    c[j] = k < b.n_bins ? hist[k] : 0u;
    mine += c[j];
  }
  // one reservation per non-empty bin, ISSUED here and looked at after the prefix: the round trip of the returning add
  // runs behind it
  unsigned long long got[kPerT];
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
    got[j] = c[j] ? atomicAdd(&b.state[tid * kPerT + j], (unsigned long long)c[j]) : 0ull;
  uint32_t inc = mine;
#pragma unroll
  for(int off = 1; off < 64; off <<= 1)
  {
    const uint32_t v = __shfl_up(inc, off, 64);
    if(lane >= (uint32_t)off) inc += v;
  }
  if(lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = inc - mine;
  for(uint32_t w = 0; w < wave; ++w) before += wsum[w];
  uint32_t total = 0;
  for(uint32_t w = 0; w < kSortThreads / 64; ++w) total += wsum[w];
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
  {
    const uint32_t k = tid * kPerT + j;
    if(k >= b.n_bins) continue;
    lfirst[k] = before;
    hist[k]   = 0u;
    before += c[j];
  }
  __syncthreads();
  // Where a bin's run goes — the two-page WINDOW.  A bin's records fill pages of S records one after the other: abstract
  // page 0 and 1 are the bin's own (static), the following ones come from the pool.  The bin's 64-bit word holds
  // {fill, epoch e, handle of the CURRENT page (abstract page e), handle of the NEXT one (e + 1)}; `fill` counts from the
  // start of the current page, so the word describes a window of 2·S slots whose pages are both known.  An add of c returns
  // where the run starts (f) and in which pages: nobody waits for a page to be handed out while its run fits the window.
  // The adder whose run contains the next page's FIRST slot (f <= S < f + c) ROTATES the window once its add is back: it
  // takes a page from the pool, notes {bin, abstract page} for `resolve`, and adds ONE delta to the word that makes
  // {fill - S, e + 1, next, new page} of it — an add, so the adds of others commute with it and what they got back still
  // means the same slots.  Only a run that ends beyond the window (more than S records arrived at this bin within one
  // rotation: a crowded bin) waits — re-reading the word once per loop trip, so that lanes of the same wave that rotate for
  // other bins are never held up — until the epoch has advanced far enough; the rotation it waits for is made by an adder
  // inside the window, who waits for nothing.  Should the window have moved PAST a waiter's slots before it looks again, it
  // finds its pages by their abstract number in the pool's notes (slow, and never seen).  A bound on the re-reads
  // (kPageSpins) ends a wait that would not end: the bin is poisoned and stays empty, the grid drains.
  const uint32_t S = 1u << b.page_shift;
  auto page_at = [&](uint32_t k, uint32_t handle, uint32_t P) -> uint32_t {   // first record of abstract page P, handle as in the word
    return handle ? (2u * b.n_bins + (handle - 1u)) << b.page_shift : (((P & 1u) ? b.n_bins : 0u) + k) << b.page_shift;   // (handle 0: P is 0 or 1, the bin's own)
  };
  auto find_page = [&](uint32_t k, uint32_t P) -> uint32_t {   // abstract page P of bin k (the slow way)
    if(P < 2u) return ((P ? b.n_bins : 0u) + k) << b.page_shift;
    // the page exists (the window has been there); its notes were stored before the rotation's add but may become
    // visible after it: look again until they are (bounded)
    for(uint32_t tries = 0; tries < 4096u; ++tries)
    {
      const uint32_t used = umin(__hip_atomic_load(b.pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b.pool_pages);
      for(uint32_t q = 0; q < used; ++q)
        if(__hip_atomic_load(&b.page_bin[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k
           && __hip_atomic_load(&b.page_seq[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (P & 0xfffu))
          return (2u * b.n_bins + q) << b.page_shift;
      __builtin_amdgcn_s_sleep(64);
    }
    return ~0u;
  };
  auto resolve = [&]() {
#pragma unroll
  for(uint32_t j = 0; j < kPerT; ++j)
  {
    const uint32_t k = tid * kPerT + j, cj = c[j];
    if(k >= b.n_bins) continue;
    uint32_t first = cj, at0 = 0u, at1 = 0u, spins = 0u;
    bool done = cj == 0u;
    const unsigned long long mine = got[j];
    const uint32_t f0 = (uint32_t)mine & 0xffffffu, e0 = (uint32_t)(mine >> 24) & 0xfffu;
    unsigned long long w = mine;   // the word as last seen: the add's result first, re-read while the run lies beyond the window
    while(!done)
    {
      const uint32_t e = (uint32_t)(w >> 24) & 0xfffu, cur = (uint32_t)(w >> 36) & 0x3fffu, nxt = (uint32_t)(w >> 50);
      const uint32_t de = (e - e0) & 0xfffu;                       // rotations since the add
      const long long f = (long long)f0 - (long long)de * (long long)S;   // the run's start, counted from the current page
      if(cur == 0x3fffu && nxt == 0x3fffu) { at0 = ~0u; done = true; }   // poisoned
      else if(f < 0 && f + (long long)cj <= 0)
      {
        // the window has moved past the whole run (never seen): its page by its abstract number
        const uint32_t P = e0 + f0 / S, r0 = f0 % S;
        const uint32_t a0 = find_page(k, P), a1 = r0 + cj > S ? find_page(k, P + 1u) : 0u;
        first = umin(cj, S - r0);
        at0 = a0 == ~0u || a1 == ~0u ? ~0u : a0 + r0;
        at1 = a1;
        done = true;
      }
      else if(f + (long long)cj <= 2ll * S)
      {
        // inside the window (or straddling its start: never seen either)
        const long long g = f < 0 ? 0 : f;   // (f < 0: the run began in the page before the current one)
        if(f < 0)
        {
          const uint32_t a0 = find_page(k, e0 + f0 / S);
          first = (uint32_t)(-f);
          at0 = a0 == ~0u ? ~0u : a0 + f0 % S;
          at1 = page_at(k, cur, e);
        }
        else if(g >= (long long)S) { at0 = page_at(k, nxt, e + 1u) + (uint32_t)(g - S); }
        else
        {
          first = umin(cj, S - (uint32_t)g);
          at0 = page_at(k, cur, e) + (uint32_t)g;
          at1 = page_at(k, nxt, e + 1u);
        }
        if(f <= (long long)S && f + (long long)cj > (long long)S)
        {
          // this run holds the NEXT page's first slot (the current page is reserved to its end): rotate the window.  (Not
          // "the current page's last slot": a run that ends exactly there would have to rotate later, when it no longer looks.)
          const uint32_t q = atomicAdd(b.pool, 1u);
          if(q < b.pool_pages)
          {
            __hip_atomic_store(&b.page_bin[q], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b.page_seq[q], ((e + 2u) & 0xfffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (modular 64-bit arithmetic: the word + delta = {fill - S, e + 1, nxt, q + 1} whatever others have added to fill)
            const unsigned long long delta = (1ull << 24) + (((unsigned long long)nxt - (unsigned long long)cur) << 36)
                                             + (((unsigned long long)(q + 1u) - (unsigned long long)nxt) << 50) - (unsigned long long)S;
            atomicAdd(&b.state[k], delta);
          }
          else   // (cannot happen: the pool holds more pages than the bins can fill)
            atomicExch(&b.state[k], kPagePoison);
        }
        done = true;
      }
      else
      {
        __builtin_amdgcn_s_sleep(8);
        if(TRT_SKIP(b, 128u))   // tuning build: waiters that look again only after ~50 µs — the window moves past them (find_page)
          for(int z = 0; z < 64; ++z) __builtin_amdgcn_s_sleep(127);
        w = __hip_atomic_load(&b.state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(++spins > kPageSpins)
        {
          at0 = ~0u;
          atomicExch(&b.state[k], kPagePoison);
          done = true;
        }
      }
    }
    // (belt and braces: a word that adds have pushed out of its poison pattern could name pages that do not exist —
    // nothing is ever written outside the record area)
    const uint32_t limit = (2u * b.n_bins + b.pool_pages) << b.page_shift;
    if(at0 != ~0u && (at0 > limit - first || (first < cj && at1 > limit - (cj - first)))) at0 = ~0u;
    lfirst[k] |= first << 16;
    g0[k]      = at0;
    g1[k]      = at1;
  }
  };
  // The reservations are looked at BEFORE the records are sorted — the add's round trip has run behind the prefix — because a
  // run that has to rotate its bin's window should do so at once: others may be waiting for it (measured: resolving after
  // the sort gains nothing on spread clouds and costs 20 % on a cloud that fills a third of the view).
  resolve();
  __syncthreads();
  // place: rank inside the bin from the LDS counter; records beyond the staging area (a chunk whose points straddle
  // many bins) go straight to their place in global memory
  auto place = [&](uint32_t bin, uint32_t rank) -> uint32_t {
    const uint32_t fr = lfirst[bin] >> 16;
    return rank < fr ? g0[bin] + rank : g1[bin] + (rank - fr);
  };
#pragma unroll
  for(uint32_t u = 0; u < kSortPer; ++u)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(splat_unpack(pk[u], z24, x0, x1, y0, y1))
    {
      const uint32_t idx = (uint32_t)(i0 + u * kSortThreads + tid);
      splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int rx0, int rx1, int ry0, int ry1) {
        const uint32_t rank = atomicAdd(&hist[bin], 1u), slot = (lfirst[bin] & 0xffffu) + rank, rect = splat_rect(rx0, rx1, ry0, ry1);
        if(slot < kStage)
          stage[slot] = SplatRec{idx, z24 | (bin << 24), rect | ((bin >> 8) << 28)};
        else if(g0[bin] != ~0u)
          b.records[place(bin, rank)] = SplatRec{idx, z24, rect};
      });
    }
  }
  __syncthreads();
  const uint32_t n_staged = umin(total, kStage);
  for(uint32_t j = tid; j < n_staged; j += kSortThreads)
  {
    const SplatRec r = stage[j];
    const uint32_t bin = (r.z >> 24) | ((r.rect >> 28) << 8);
    if(g0[bin] != ~0u)
      b.records[place(bin, j - (lfirst[bin] & 0xffffu))] = SplatRec{r.idx, r.z & 0xffffffu, r.rect & 0x0fffffffu};
  }
  }   // chunks
}

// scatter, direct (images of more than kSortBins bins): every record written at its place.  ONE sweep over the side array:
// the block's 8,192 projected points stay in registers (16 per lane of 512, loaded together) between the counting and the
// placing pass.
constexpr uint32_t kDirectThreads = 512, kDirectPer = kSplatChunk / kDirectThreads;
__global__ __launch_bounds__(kDirectThreads) void splat_scatter_kernel(uint64_t n, const SplatBins b)
{
  extern __shared__ uint32_t lds[];
  uint32_t* hist = lds;             // [n_bins] points of this block per bin, then the rank counter
  uint32_t* base = lds + b.n_bins;  // [n_bins] first record of this block's range in the bin
  for(uint32_t k = threadIdx.x; k < b.n_bins; k += kDirectThreads) hist[k] = 0u;
  __syncthreads();
  const uint64_t i0 = (uint64_t)blockIdx.x * kSplatChunk, i1 = min(n, i0 + kSplatChunk);
  uint2 pk[kDirectPer];
#pragma unroll
  for(uint32_t u = 0; u < kDirectPer; ++u)
  {
    const uint64_t i = i0 + u * kDirectThreads + threadIdx.x;
    pk[u] = i < i1 ? b.proj[i] : make_uint2(0u, 0u);
  }
#pragma unroll
  for(uint32_t u = 0; u < kDirectPer; ++u)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(splat_unpack(pk[u], z24, x0, x1, y0, y1))
      splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int, int, int, int) { atomicAdd(&hist[bin], 1u); });
  }
  __syncthreads();
  for(uint32_t k = threadIdx.x; k < b.n_bins; k += kDirectThreads)
  {
    base[k] = hist[k] ? b.offset[k] + atomicAdd(&b.cursor[k], hist[k]) : 0u;
    hist[k] = 0u;
  }
  __syncthreads();
#pragma unroll
  for(uint32_t u = 0; u < kDirectPer; ++u)
  {
    uint32_t z24;
    int x0, x1, y0, y1;
    if(splat_unpack(pk[u], z24, x0, x1, y0, y1))
      splat_for_bins(b, x0, x1, y0, y1, [&](uint32_t bin, int rx0, int rx1, int ry0, int ry1) {
        const uint32_t rank = atomicAdd(&hist[bin], 1u);
        b.records[base[bin] + rank] = SplatRec{(uint32_t)(i0 + u * kDirectThreads + threadIdx.x), z24, splat_rect(rx0, rx1, ry0, ry1)};
      });
  }
}

constexpr int kSplatResolveThreads = 1024;   // 2 blocks of 64 KB LDS per CU: 32 waves, all the CU holds
constexpr uint32_t kPageList = 2056;         // pages a bin can hold (PAGED; see the kernel)
// the depth test of a batch of records: ds_min_u64 on the pixels of their rectangles.  (A plain LDS read in front of the
// atomic, to keep the three quarters of the fragments that lose away from the atomic unit, makes it SLOWER — 104.6 against
// 81.6 µs: the read is a round trip, the atomic is not.)
template <uint32_t kR>
__device__ __forceinline__ void splat_depth_test(unsigned long long* keys, const SplatRec (&rec)[kR])
{
#pragma unroll
  for(uint32_t u = 0; u < kR; ++u)
  {
    const unsigned long long key = ((unsigned long long)rec[u].z << 32) | (unsigned long long)rec[u].idx;
    const uint32_t x0 = rec[u].rect & 127u, x1 = (rec[u].rect >> 7) & 255u, y0 = (rec[u].rect >> 15) & 63u, y1 = (rec[u].rect >> 21) & 127u;
    for(uint32_t y = y0; y < y1; ++y)
      for(uint32_t x = x0; x < x1; ++x)
        atomicMin(&keys[y * kBinW + x], key);
  }
}

template <bool PAGED>
__global__ __launch_bounds__(kSplatResolveThreads) void splat_resolve_bins_kernel(const trt_point* __restrict__ pts, const SplatArgs a, const SplatBins b,
                                                                                  float4 clear, float4* rgba)
{
  __shared__ unsigned long long keys[kBinW * kBinH];
  __shared__ uint32_t plist[PAGED ? kPageList : 1], n_list;
  for(uint32_t k = threadIdx.x; k < kBinW * kBinH; k += blockDim.x) keys[k] = kSplatClear;
  const uint32_t bin = blockIdx.x;
  constexpr uint32_t kR = 4;   // four records per lane and trip, their loads issued together (eight: +2 %)
  if(PAGED)
  {
    // the bin's pages: its own two (abstract pages 0 and 1) and the pool pages whose notes name it; how many records a
    // page holds follows from its abstract number P and the bin's word {fill, epoch e, …}: every page before the current
    // one (P < e) is full, the current one holds min(fill, S), the next one what is left of fill.  The page size
    // (splat_plan) makes n_points / S < 2,049, and a bin holds at most one record per point: the list cannot overflow.
    const unsigned long long st = b.state[bin];
    const uint32_t S = 1u << b.page_shift, fill = (uint32_t)st & 0xffffffu, e = (uint32_t)(st >> 24) & 0xfffu;
    const bool     poisoned = ((uint32_t)(st >> 36) & 0x3fffu) == 0x3fffu && (uint32_t)(st >> 50) == 0x3fffu;   // (a reservation gave up, splat_bin_kernel: the bin stays empty)
    const uint32_t used = umin(__hip_atomic_load(b.pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b.pool_pages);
    if(threadIdx.x == 0)
    {
      n_list   = 2u;
      plist[0] = bin;                      // page index | abstract number << 16
      plist[1] = (b.n_bins + bin) | (1u << 16);
    }
    __syncthreads();   // (also: the keys are initialised)
    for(uint32_t q = threadIdx.x; q < used && !poisoned; q += blockDim.x)
      if(b.page_bin[q] == bin)
      {
        const uint32_t en = atomicAdd(&n_list, 1u);
        if(en < kPageList) plist[en] = (2u * b.n_bins + q) | (b.page_seq[q] << 16);
      }
    __syncthreads();
    const uint32_t slots = poisoned ? 0u : umin(n_list, kPageList) << b.page_shift;
    for(uint32_t rb = threadIdx.x; rb < slots; rb += kR * blockDim.x)
    {
      SplatRec rec[kR];
#pragma unroll
      for(uint32_t u = 0; u < kR; ++u)
      {
        const uint32_t v = rb + u * blockDim.x, r = v & (S - 1u);
        const uint32_t en = v < slots ? plist[v >> b.page_shift] : 0u, page = en & 0xffffu, P = en >> 16;
        const uint32_t have = P < e ? S : (P == e ? umin(fill, S) : (P == e + 1u && fill > S ? umin(fill - S, S) : 0u));
        const bool     in = v < slots && r < have && !TRT_SKIP(b, 64u);
        rec[u] = in ? b.records[((size_t)page << b.page_shift) + r] : SplatRec{0u, 0u, 0u};   // an empty rectangle
      }
      if(TRT_SKIP(b, 32u)) continue;
      splat_depth_test<kR>(keys, rec);
    }
    __syncthreads();
    if(threadIdx.x == 0)
    {
      // the next call starts from zero: this bin's state word, and — once every block has read it — the pool counter
      b.state[bin] = 0ull;
      if(atomicAdd(b.rticket, 1u) == gridDim.x - 1u)
      {
        __hip_atomic_store(b.pool, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(b.rticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  else
  {
    __syncthreads();
    const uint32_t cnt = b.count[bin], off = b.offset[bin];
    for(uint32_t rb = threadIdx.x; rb < cnt; rb += kR * blockDim.x)
    {
      SplatRec rec[kR];
#pragma unroll
      for(uint32_t u = 0; u < kR; ++u)
      {
        const uint32_t r = rb + u * blockDim.x;
        rec[u] = r < cnt && !TRT_SKIP(b, 64u) ? b.records[off + r] : SplatRec{0u, 0u, 0u};   // an empty rectangle
      }
      if(TRT_SKIP(b, 32u)) continue;
      splat_depth_test<kR>(keys, rec);
    }
    __syncthreads();
    if(threadIdx.x == 0) b.count[bin] = 0u;   // the next call counts from zero
  }
  const uint32_t ox = (bin % b.bins_x) * kBinW, oy = (bin / b.bins_x) * kBinH;
  // the colour gather is a dependent random 16-B read per covered pixel: four of them in flight per lane
  constexpr uint32_t kU = 4;
  for(uint32_t k0 = threadIdx.x; k0 < kBinW * kBinH; k0 += kU * blockDim.x)
  {
    unsigned long long key[kU];
    float4 c[kU];
#pragma unroll
    for(uint32_t u = 0; u < kU; ++u)
    {
      const uint32_t k = k0 + u * blockDim.x;
      key[u] = k < kBinW * kBinH ? keys[k] : kSplatClear;
    }
#pragma unroll
    for(uint32_t u = 0; u < kU; ++u)
    {
      c[u] = clear;
      if(key[u] != kSplatClear && !TRT_SKIP(b, 16u))
      {
        const float4 pc = reinterpret_cast<const float4*>(pts)[2 * (size_t)(uint32_t)key[u] + 1];
        c[u] = make_float4(pc.x, pc.y, pc.z, 1.0f);   // o_color = vec4(current.color.xyz, 1.0)  (frag_shader.frag:44)
      }
    }
#pragma unroll
    for(uint32_t u = 0; u < kU; ++u)
    {
      const uint32_t k = k0 + u * blockDim.x;
      const uint32_t x = ox + (k % kBinW), y = oy + (k / kBinW);
      if(k < kBinW * kBinH && x < a.W && y < a.H)
        rgba[(size_t)y * a.W + x] = c[u];
    }
  }
}

__global__ __launch_bounds__(256) void splat_resolve_kernel(const unsigned long long* __restrict__ keys, uint64_t n,
                                                            const trt_point* __restrict__ pts, float4 clear, float4* rgba)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
  {
    const unsigned long long k = keys[i];
    float4 c = clear;
    if(k != kSplatClear)
    {
      const float4 pc = reinterpret_cast<const float4*>(pts)[2 * (size_t)(uint32_t)k + 1];
      c = make_float4(pc.x, pc.y, pc.z, 1.0f);   // o_color = vec4(current.color.xyz, 1.0)  (frag_shader.frag:44)
    }
    rgba[i] = c;
  }
}

SplatPlan splat_plan(uint32_t W, uint32_t H, float point_size, uint64_t n_points, const Tuning& tn)
{
  SplatPlan pl{};
  const uint64_t nb = (uint64_t)((W + kBinW - 1) / kBinW) * ((H + kBinH - 1) / kBinH), np = n_points ? n_points : 1;
  auto al = [](uint64_t v) { return (size_t)((v + 15) & ~(uint64_t)15); };
  // the one-pass form: images or points beyond what a record can say, or more than 8 GiB of records
  if(tn.splat_variant == kSplatOnePass || nb > kSplatMaxBins || W > kSplatMaxDim || H > kSplatMaxDim || !(point_size <= 32.0f) ||
     4 * n_points > 0xffffffffull || np * (4 * kSplatRecordSize + 8) > ((uint64_t)8 << 30))
    return pl;
  pl.n_bins = (uint32_t)nb;
  pl.mode   = nb <= kSortBins ? kSplatPaged : kSplatDirect;
  if(tn.splat_variant == kSplatSorted && nb <= kSortBins) pl.mode = kSplatSorted;
  if(tn.splat_variant == kSplatDirect) pl.mode = kSplatDirect;
  if(pl.mode == kSplatPaged)
  {
    // Page size S: at least 4,096 records (a chunk's run for a bin is at most 4,096 long: it spans at most two pages, the
    // window of splat_bin_kernel) and large enough that the worst case (4 records per point) fills at most 2,048 pool pages:
    // an evenly spread cloud then leaves a bin in its own two pages or one rotation beyond (8.4 M points, 512 bins: S =
    // 16,384 against 16 k records per bin), and `resolve` picks a bin's pages out of a few thousand notes.  Every bin owns
    // two pages; a bin leaves less than one pool page unfilled and holds one more handed out ahead, so the pool never runs
    // dry with 4n/S + 2·n_bins + 2 pages.  (S = 4,096 for that cloud: the same on spread clouds, 4-5x slower on one that
    // fills a third of the view — a page end every 16 µs per bin is more than the rotations keep up with.)
    uint32_t shift = 12;
    while(shift < 19 && ((4 * np) >> shift) > 2048u) ++shift;
    // … and a window (2·S) that takes what one round of resident blocks (512 x 4,096 points) brings a bin four times
    // more crowded than the average in one burst: below that the adders of a moderately crowded bin wait for rotations
    // (2 M points over a third of the view, S = 4,096: 0.138 against 0.082 ms)
    const uint64_t burst = np < ((uint64_t)1 << 21) ? np : ((uint64_t)1 << 21);
    while(shift < 19 && ((uint64_t)1 << shift) * nb < 4 * burst) ++shift;
#ifdef TRT_TUNING
    if(const char* e = getenv("TRT_SPLAT_PAGE_SHIFT")) shift = (uint32_t)atoi(e);
#endif
    const uint64_t pool = ((4 * np) >> shift) + 2 * nb + 2, slots = (2 * nb + pool) << shift;
    if(slots <= 0xffffffffull && ((4 * np) >> shift) <= 2048u && 2 * nb + pool < 65536u && slots * kSplatRecordSize <= ((uint64_t)8 << 30))
    {
      pl.page_shift = shift; pl.pool_pages = (uint32_t)pool;
      pl.rec_bytes = al(slots * kSplatRecordSize); pl.pagebin_bytes = al(pool * sizeof(uint32_t)) * 2;   // page_bin + page_seq
      return pl;
    }
    pl.mode = kSplatSorted;   // (clouds beyond what the page scheme addresses, or whose pages would take more than 8 GiB: the two-pass form)
  }
  pl.rec_bytes  = al(np * 4 * kSplatRecordSize);
  pl.proj_bytes = al(np * 8);
  if(pl.mode == kSplatSorted) pl.table_bytes = al((np / kSortChunk + 4) * nb * sizeof(uint32_t));
  return pl;
}

hipError_t launch_splat(const trt_point* pts, uint64_t n_points, const float* vp, uint32_t W, uint32_t H,
                        const float* clear, float point_size, const SplatScratch& sc, float* rgba, int n_cus,
                        const Tuning& tn, hipStream_t stream)
{
  SplatArgs a;
  for(int i = 0; i < 16; ++i) a.vp[i] = vp[i];
  a.W = W; a.H = H; a.half = point_size * 0.5f;
  const SplatPlan& pl = sc.plan;
  if(pl.mode != kSplatOnePass)
  {
    SplatBins b{};
    b.bins_x = (W + kBinW - 1) / kBinW; b.bins_y = (H + kBinH - 1) / kBinH; b.n_bins = pl.n_bins;
    // fixed layout whatever n_bins is: the count words of one call never alias another call's offsets
    b.count = sc.bin_words; b.offset = sc.bin_words + kSplatMaxBins; b.cursor = sc.bin_words + 2 * (size_t)kSplatMaxBins;
    b.state   = reinterpret_cast<unsigned long long*>(sc.bin_words + 3 * (size_t)kSplatMaxBins);
    b.ticket  = sc.bin_words + kSplatTicketWord;
    b.pool    = sc.bin_words + kSplatTicketWord + 1;
    b.rticket = sc.bin_words + kSplatTicketWord + 2;
    char* base = static_cast<char*>(sc.records);
    b.records  = reinterpret_cast<SplatRec*>(base);
    b.proj     = reinterpret_cast<uint2*>(base + pl.rec_bytes);
    b.table    = pl.mode == kSplatSorted ? reinterpret_cast<uint32_t*>(base + pl.rec_bytes + pl.proj_bytes) : nullptr;
    b.page_bin = reinterpret_cast<uint32_t*>(base + pl.rec_bytes + pl.proj_bytes + pl.table_bytes);
    b.page_seq = b.page_bin + pl.pagebin_bytes / 2 / sizeof(uint32_t);
    b.page_shift = pl.page_shift; b.pool_pages = pl.pool_pages;
    b.debug_skip = tn.debug_skip;   // always 0 in the release build
    const dim3 rgrid(pl.n_bins), rblock(kSplatResolveThreads);
    const float4 cl = make_float4(clear[0], clear[1], clear[2], clear[3]);
    if(pl.mode == kSplatPaged)
    {
      if(n_points)
      {
        const uint32_t schunks = (uint32_t)((n_points + kSortChunk - 1) / kSortChunk);
        // two blocks per CU are resident (64-80 KB of LDS each): that many blocks, each walking its share of the chunks
        uint32_t bgrid = (uint32_t)n_cus * 2u;
        if(tn.splat_blocks_per_cu) bgrid = (uint32_t)(n_cus * tn.splat_blocks_per_cu);
        if(bgrid == 0u || bgrid > schunks) bgrid = schunks;
        if(pl.n_bins <= 512u) hipLaunchKernelGGL(splat_bin_kernel<512>, dim3(bgrid), dim3(kSortThreads), 0, stream, pts, n_points, a, b);
        else hipLaunchKernelGGL(splat_bin_kernel<kSortBins>, dim3(bgrid), dim3(kSortThreads), 0, stream, pts, n_points, a, b);
      }
      hipLaunchKernelGGL(splat_resolve_bins_kernel<true>, rgrid, rblock, 0, stream, pts, a, b, cl, reinterpret_cast<float4*>(rgba));
      return hipGetLastError();
    }
    if(n_points)
    {
      const uint32_t chunks = (uint32_t)((n_points + kSplatChunk - 1) / kSplatChunk), schunks = (uint32_t)((n_points + kSortChunk - 1) / kSortChunk);
      if(b.table)   // sorted scatter, two passes (-DTRT_TUNING builds: TRT_SPLAT_VARIANT=1)
      {
        hipLaunchKernelGGL((splat_count_kernel<kSortChunk, 4, true>), dim3((schunks + 3) / 4), dim3(1024), 4 * pl.n_bins * sizeof(uint32_t), stream, pts, n_points, a, b);
        if(pl.n_bins <= 512u)
          hipLaunchKernelGGL(splat_scatter_sorted_kernel<512>, dim3(schunks), dim3(kSortThreads), 0, stream, n_points, b);
        else
          hipLaunchKernelGGL(splat_scatter_sorted_kernel<kSortBins>, dim3(schunks), dim3(kSortThreads), 0, stream, n_points, b);
      }
      else
      {
        hipLaunchKernelGGL((splat_count_kernel<kSplatChunk, 1, false>), dim3(chunks), dim3(256), pl.n_bins * sizeof(uint32_t), stream, pts, n_points, a, b);
        hipLaunchKernelGGL(splat_scatter_kernel, dim3(chunks), dim3(kDirectThreads), 2 * pl.n_bins * sizeof(uint32_t), stream, n_points, b);
      }
    }
    hipLaunchKernelGGL(splat_resolve_bins_kernel<false>, rgrid, rblock, 0, stream, pts, a, b, cl, reinterpret_cast<float4*>(rgba));
    return hipGetLastError();
  }
  unsigned long long* keys = sc.keys;
  uint64_t npx = (uint64_t)W * H, cap = (uint64_t)n_cus * 16;
  if(tn.splat_blocks_per_cu) cap = (uint64_t)n_cus * tn.splat_blocks_per_cu;
  if(cap == 0) cap = 1;
  auto grid = [&](uint64_t n) { const uint64_t w = (n + 255) / 256; return dim3((uint32_t)(w < cap ? (w ? w : 1) : cap)); };
  hipLaunchKernelGGL(splat_clear_kernel, grid(npx), dim3(256), 0, stream, keys, npx);
  if(n_points)
    hipLaunchKernelGGL(splat_points_kernel, grid(n_points), dim3(256), 0, stream, pts, n_points, a, keys);
  hipLaunchKernelGGL(splat_resolve_kernel, grid(npx), dim3(256), 0, stream, keys, npx, pts,
                     make_float4(clear[0], clear[1], clear[2], clear[3]), reinterpret_cast<float4*>(rgba));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
hipError_t launch_trace(const SceneK& scene, const TraceArgs& a, const Tuning& tn, hipStream_t stream)
{
  if(a.rays.n == 0)
    return hipSuccess;
  const uint64_t want = (a.rays.n + 255) / 256;
  uint64_t cap = 256u * 16u;
  if(tn.trace_blocks) cap = tn.trace_blocks;
  if(cap == 0) cap = 1;
  const uint32_t grid = (uint32_t)(want < cap ? want : cap);
  if(scene.f64 && scene.dk) hipLaunchKernelGGL((trace_kernel<double, true>), dim3(grid), dim3(256), 0, stream, scene, a);
  else if(scene.f64) hipLaunchKernelGGL((trace_kernel<double, false>), dim3(grid), dim3(256), 0, stream, scene, a);
  else if(scene.dk) hipLaunchKernelGGL((trace_kernel<float, true>), dim3(grid), dim3(256), 0, stream, scene, a);
  else hipLaunchKernelGGL((trace_kernel<float, false>), dim3(grid), dim3(256), 0, stream, scene, a);
  return hipGetLastError();
}

// Zeroes up to 64 words (the query counters of a counted launch) with a one-wave kernel: a kernel
// node when the stream is being captured — the *_dev entry points put no memset node into a graph
// (32 memset nodes between 64 kernel nodes faulted on replay under ROCm 7.2).
__global__ void zero_words_kernel(unsigned int* q, uint32_t n)
{
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) q[i] = 0u;
}

hipError_t launch_zero_words(unsigned int* words, uint32_t n, hipStream_t stream)
{
  if(n <= 64u) hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, stream, words, n);
  else hipLaunchKernelGGL(zero_words_kernel, dim3((n + 1023u) / 1024u < 64u ? (n + 1023u) / 1024u : 64u), dim3(1024), 0, stream, words, n);
  return hipGetLastError();
}

#ifdef TRT_TIMELINE
hipError_t set_timeline(void* dev_ptr)
{
  unsigned long long* p = static_cast<unsigned long long*>(dev_ptr);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &p, sizeof(p));
}
#endif

Tuning tuning_from_env()
{
  Tuning t;
#ifdef TRT_TUNING
  auto u64 = [](const char* name, uint64_t& v) { if(const char* e = getenv(name)) v = (uint64_t)atoll(e); };
  auto i32 = [](const char* name, int& v) { if(const char* e = getenv(name)) v = atoi(e); };
  auto u32 = [](const char* name, uint32_t& v) { if(const char* e = getenv(name)) v = (uint32_t)atoi(e); };
  u32("TRT_MIN_BATCH", t.min_batch);
  i32("TRT_FINE_CLASSIFY", t.fine);
  if(getenv("TRT_NO_TILE_CULL")) t.no_tile_cull = 1;
  if(getenv("TRT_NO_ENCLOSURE")) t.no_enclosure = 1;
  u32("TRT_DEBUG_SKIP", t.debug_skip);
  u32("TRT_HEAVY_X16", t.heavy_x16);
  u32("TRT_HEAVY_MIN_TORI", t.heavy_min_tori);
  if(getenv("TRT_DEBUG_TILES")) t.debug_tiles = 1;
  u64("TRT_PERSIST_BLOCKS", t.persist_blocks);
  u64("TRT_LISTED_BLOCKS", t.listed_blocks);
  i32("TRT_TILE", t.static_tile);
  u64("TRT_TRACE_BLOCKS", t.trace_blocks);
  u64("TRT_POST_BLOCKS_PER_CU", t.post_blocks_per_cu);
  u64("TRT_SPLAT_BLOCKS_PER_CU", t.splat_blocks_per_cu);
  i32("TRT_TRACE_VARIANT", t.trace_variant);
  i32("TRT_SPLAT_VARIANT", t.splat_variant);
#endif
  return t;
}

hipError_t launch_render(const SceneK& scene, const RenderArgs& a, RenderVariant v, int n_cus, const Tuning& tn,
                         hipStream_t stream)
{
  if(a.n_local_rows == 0 || a.W == 0)
    return hipSuccess;
  const uint64_t tiles = (uint64_t)((a.W + 7) / 8) * ((a.n_local_rows + 7) / 8);
  if(v == kRenderPersistent || v == kRenderListed)
  {
    // 1. classify the tiles into the LIVE and CLEAR lists (one lane per macro tile, or per 8×8 tile when a.fine)
    const uint64_t macros = (uint64_t)(((a.W + 7) / 8 + kMacroTiles - 1) / kMacroTiles) * ((a.n_local_rows + 7) / 8);
    const uint64_t lanes  = a.fine ? macros * kMacroTiles : macros;
    // cost feedback: the plain listed kernels only (the counted and the alternative-solver instantiations go without)
    const bool fb = v == kRenderListed && a.tile_cost != nullptr && a.heavy_x16 != 0u && a.stats == nullptr && !scene.dk;
    const dim3 cgrid((uint32_t)((lanes + kClassifyThreads - 1) / kClassifyThreads));
    if(a.fine && fb) hipLaunchKernelGGL(tile_classify_fine_kernel<true>, cgrid, dim3(kClassifyThreads), 0, stream, scene, a);
    else if(a.fine) hipLaunchKernelGGL(tile_classify_fine_kernel<false>, cgrid, dim3(kClassifyThreads), 0, stream, scene, a);
    else if(fb) hipLaunchKernelGGL(tile_classify_kernel<true>, cgrid, dim3(kClassifyThreads), 0, stream, scene, a);
    else hipLaunchKernelGGL(tile_classify_kernel<false>, cgrid, dim3(kClassifyThreads), 0, stream, scene, a);
    // 2. resident grid: kPersistentBlocksPerCU blocks of 4 waves per CU, never more waves than tiles
    uint64_t cap = (uint64_t)n_cus * kPersistentBlocksPerCU;
    if(tn.persist_blocks) cap = tn.persist_blocks;
    if(v == kRenderListed)
    {
      // one wave per 16 tiles (4096²: 16,384 blocks = 64 per CU): with ≈16 % of the tiles LIVE a
      // wave traces at most one tile and writes ≈1 clear macro tile, so the dispatcher balances
      // single tiles and compute/store phases of different blocks interleave on every CU
      // (measured optimum at 2048², 4096² and 8192²; 8,192 blocks cost +25 % at 4096²)
      uint64_t lcap = tiles / 16 > (uint64_t)n_cus * 4 ? tiles / 16 : (uint64_t)n_cus * 4;
      if(tn.listed_blocks) lcap = tn.listed_blocks;
      constexpr uint32_t bthreads = kListedThreads, wpb = bthreads / 64;   // the kernel's staging assumes 256-thread blocks (64 / 128: measured slower)
      const uint32_t lgrid = (uint32_t)((tiles + wpb - 1) / wpb < lcap ? (tiles + wpb - 1) / wpb : lcap);
#define TRT_LAUNCH_LISTED(REAL, DK_)                                                                                   \
  do {                                                                                                                 \
    if(fb && a.rendered) hipLaunchKernelGGL((render_listed_kernel<REAL, false, false, true, true>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);  \
    else if(fb) hipLaunchKernelGGL((render_listed_kernel<REAL, false, false, false, true>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);      \
    else if(a.rendered && a.stats) hipLaunchKernelGGL((render_listed_kernel<REAL, true, DK_, true>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);  \
    else if(a.rendered) hipLaunchKernelGGL((render_listed_kernel<REAL, false, DK_, true>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);       \
    else if(a.stats) hipLaunchKernelGGL((render_listed_kernel<REAL, true, DK_, false>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);          \
    else hipLaunchKernelGGL((render_listed_kernel<REAL, false, DK_, false>), dim3(lgrid), dim3(bthreads), 0, stream, scene, a);                     \
  } while(0)
      if(scene.f64 && scene.dk) TRT_LAUNCH_LISTED(double, true);
      else if(scene.f64) TRT_LAUNCH_LISTED(double, false);
      else if(scene.dk) TRT_LAUNCH_LISTED(float, true);
      else TRT_LAUNCH_LISTED(float, false);
#undef TRT_LAUNCH_LISTED
      return hipGetLastError();
    }
    if(scene.dk)
      return hipErrorInvalidValue;  // the persistent variant implements the default solver only (trt_api.hip checks)
    const uint32_t grid = (uint32_t)((tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap);
    if(scene.f64)
      hipLaunchKernelGGL(render_persistent_kernel<double>, dim3(grid), dim3(256), 0, stream, scene, a);
    else
      hipLaunchKernelGGL(render_persistent_kernel<float>, dim3(grid), dim3(256), 0, stream, scene, a);
    return hipGetLastError();
  }
  // wave tile shape of the static kernel: TRT_TILE = 8x8 (default) | 16x4 | 32x2 | 64x1
  int tw = tn.static_tile;
  if(tw != 8 && tw != 16 && tw != 32 && tw != 64) tw = 8;
  const uint64_t stiles = (uint64_t)((a.W + tw - 1) / tw) * ((a.n_local_rows + 64 / tw - 1) / (64 / tw));
  const uint32_t grid = (uint32_t)((stiles + 3) / 4);
#define TRT_LAUNCH_STATIC(REAL, TW_, DK_) \
  hipLaunchKernelGGL((render_static_kernel<REAL, TW_, DK_>), dim3(grid), dim3(256), 0, stream, scene, a)
  if(scene.f64 && scene.dk) TRT_LAUNCH_STATIC(double, 8, true);
  else if(scene.f64) TRT_LAUNCH_STATIC(double, 8, false);
  else if(scene.dk) TRT_LAUNCH_STATIC(float, 8, true);
  else if(tw == 16) TRT_LAUNCH_STATIC(float, 16, false);
  else if(tw == 32) TRT_LAUNCH_STATIC(float, 32, false);
  else if(tw == 64) TRT_LAUNCH_STATIC(float, 64, false);
  else TRT_LAUNCH_STATIC(float, 8, false);
#undef TRT_LAUNCH_STATIC
  return hipGetLastError();
}

// A batch of frames with the listed kernel (trt_render_batch_dev): one classification over every frame's tiles, one
// render kernel over the joint lists — the launch shape of one frame with as many tiles as all of them together.
hipError_t launch_render_batch(const SceneK& scene, const RenderBatch& b, int n_cus, const Tuning& tn, hipStream_t stream)
{
  const RenderArgs& a = b.fr[0];
  if(b.n_frames == 0 || a.n_local_rows == 0 || a.W == 0)
    return hipSuccess;
  if(scene.dk || a.rendered)
    return hipErrorInvalidValue;   // (trt_api.hip refuses these before)
  const uint64_t tiles = (uint64_t)((a.W + 7) / 8) * ((a.n_local_rows + 7) / 8) * b.n_frames;
  const uint64_t lanes = (uint64_t)b.per_frame * b.n_frames;
  const bool fb = a.tile_cost != nullptr && a.heavy_x16 != 0u && a.stats == nullptr;
  const dim3 cgrid((uint32_t)((lanes + kClassifyThreads - 1) / kClassifyThreads));
  if(a.fine && fb) hipLaunchKernelGGL((tile_classify_fine_kernel<true, true>), cgrid, dim3(kClassifyThreads), 0, stream, scene, b);
  else if(a.fine) hipLaunchKernelGGL((tile_classify_fine_kernel<false, true>), cgrid, dim3(kClassifyThreads), 0, stream, scene, b);
  else if(fb) hipLaunchKernelGGL((tile_classify_kernel<true, true>), cgrid, dim3(kClassifyThreads), 0, stream, scene, b);
  else hipLaunchKernelGGL((tile_classify_kernel<false, true>), cgrid, dim3(kClassifyThreads), 0, stream, scene, b);
  uint64_t lcap = tiles / 16 > (uint64_t)n_cus * 4 ? tiles / 16 : (uint64_t)n_cus * 4;   // as launch_render: one wave per 16 tiles
  if(tn.listed_blocks) lcap = tn.listed_blocks;
  constexpr uint32_t wpb = kListedThreads / 64;
  const uint32_t lgrid = (uint32_t)((tiles + wpb - 1) / wpb < lcap ? (tiles + wpb - 1) / wpb : lcap);
#define TRT_LAUNCH_BATCH(REAL)                                                                                                        \
  do {                                                                                                                                \
    if(fb) hipLaunchKernelGGL((render_listed_kernel<REAL, false, false, false, true, true>), dim3(lgrid), dim3(kListedThreads), 0, stream, scene, b);      \
    else if(a.stats) hipLaunchKernelGGL((render_listed_kernel<REAL, true, false, false, false, true>), dim3(lgrid), dim3(kListedThreads), 0, stream, scene, b);  \
    else hipLaunchKernelGGL((render_listed_kernel<REAL, false, false, false, false, true>), dim3(lgrid), dim3(kListedThreads), 0, stream, scene, b);       \
  } while(0)
  if(scene.f64) TRT_LAUNCH_BATCH(double);
  else TRT_LAUNCH_BATCH(float);
#undef TRT_LAUNCH_BATCH
  return hipGetLastError();
}

}  // namespace trt
