// trt_kernels.hpp — launch interface between the C ABI (trt_api.hip) and the gfx950 kernels
// (trt_kernels.hip).  Host-side only types; no HIP runtime types leak past this header
// except hipStream_t / hipError_t.
#pragma once

#include <hip/hip_runtime.h>

#include "trt_device.hpp"

namespace trt {

struct RenderArgs {
  trt_globals g;       // GlobalUniforms, by value in the kernel-argument segment
  trt_push    pc;      // PushConstantRay
  ToroCam     toro;    // toroidal camera frame + device trig tables
  uint32_t    W, H;
  uint32_t    row_begin, row_end;  // contiguous band (tile_parts <= 1)
  // interleaved row groups (multi-GPU tiling): this launch owns the rows y with
  // (y / tile_group) % tile_parts == tile_part; n_local_rows of them.  compact != 0: the rgba
  // and first-hit streams are indexed by LOCAL row (buffers hold only this part's rows).
  uint32_t    tile_group, tile_parts, tile_part, compact;
  uint32_t    n_local_rows;
  int         camera;
  float*             rgba;      // [H][W][4]                      (rgen:87)
  trt_hits           hits;      // SoA depth-0 hit record, y*W+x  (optional streams)
  trt_rendered_data* rendered;  // AoS, x*H+y                     (BEF rgen:72-73,111-112)
  unsigned long long* stats;    // [4]: primary, bounce, shadow tests, pixels (optional)
  // persistent kernel: tile lists built by tile_classify_kernel (packed tx | ty << 16)
  unsigned int*       queue;        // [0] = #LIVE tiles, [1] = #CLEAR macro tiles of this frame (zero on entry)
  unsigned int*       queue_next;   // the counter set of the next frame, zeroed by this one
  uint32_t*           tiles_live;   // tiles that need ray tracing
  uint32_t*           tiles_clear;  // tiles whose every pixel misses every bounding sphere
  uint32_t            min_batch;    // persistent kernel: lanes needed to run a shader/refill round (default 24)
  uint32_t            tile_cull;    // 0: classify every tile as LIVE
  uint32_t            fine;         // 1: classify per 8×8 tile with the distance-function march (toroidal camera)
  uint32_t            debug_skip;   // diagnostics (TRT_DEBUG_SKIP): 1 = skip clear tiles, 2 = skip traced tiles
  uint32_t            vec4_ok;      // W % 4 == 0 and all first-hit streams 16-B aligned: dwordx4 clears
};

struct TraceArgs {
  trt_rays rays;
  trt_hits hits;
  float    tmin, tmax;
  unsigned long long* stats;
};

enum RenderVariant { kRenderStatic = 0, kRenderPersistent = 1, kRenderListed = 2 };
constexpr int kPersistentBlocksPerCU = 16;  // 4× the resident 4 blocks/CU: the dispatcher evens out the tile costs

hipError_t launch_post(const float* in, uint64_t n, float* f32_out, uint8_t* u8_out, int n_cus, hipStream_t stream);
hipError_t launch_splat(const trt_point* pts, uint64_t n_points, const float* vp, uint32_t W, uint32_t H,
                        const float* clear, float point_size, unsigned long long* keys, float* rgba, int n_cus,
                        hipStream_t stream);
hipError_t launch_trace(const SceneK& scene, const TraceArgs& a, hipStream_t stream);
hipError_t launch_zero_counters(unsigned int* queue, hipStream_t stream);
hipError_t launch_render(const SceneK& scene, const RenderArgs& a, RenderVariant v, int n_cus,
                         hipStream_t stream);

}  // namespace trt
