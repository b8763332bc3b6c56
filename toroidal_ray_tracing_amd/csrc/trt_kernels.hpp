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
  // tile lists built by tile_classify*_kernel (packed tx | ty << 16 [| miss flag])
  unsigned int*       counters;     // accumulators of the classification: [0] LIVE, [1] CLEAR, [2] shards done, [8..15] blocks done per shard.
                                    // Zero between frames: the LAST classification block of a frame publishes the
                                    // totals to `counts` and resets them (no memset, no double buffering: a frame
                                    // depends on no other frame, eager or replayed from a hipGraph)
  unsigned int*       counts;       // [0] = #LIVE tiles, [1] = #CLEAR macro tiles of this frame (written, never added to);
                                    // [2] = how many of the LIVE tiles are HEAVY, [3] = mean cost of the previous frame's traced macro tiles
  uint32_t*           tiles_live;   // tiles that need ray tracing
  uint32_t*           tiles_clear;  // macro tiles whose every pixel provably misses
  // Cost feedback: the listed kernel leaves the time (100-MHz ticks) its slowest wave spent on each traced macro tile; the
  // next classification reads and resets it, calls a macro tile HEAVY when that exceeds heavy_x16/16 × the previous mean,
  // and stores the HEAVY tiles downwards from the END of tiles_live: logical entry L < #HEAVY is tiles_live[cap_live-1-L],
  // any other one tiles_live[L - #HEAVY] — the render kernels start with the heavy tiles.  Scheduling only: no result
  // depends on the order of the list, and a frame without history (or heavy_x16 == 0) has no heavy tiles.
  uint32_t*           tile_cost;    // [macro tiles of the launch] or nullptr
  uint32_t            heavy_x16;
  uint32_t            cap_live;     // capacity of tiles_live / tiles_clear in entries: nothing indexes past them
  uint32_t            cap_clear;
  uint32_t            min_batch;    // persistent kernel: lanes needed to run a shader/refill round (default 24)
  uint32_t            tile_cull;    // 0: classify every tile as LIVE
  uint32_t            fine;         // 1: classify per 8×8 tile with the distance-function march (toroidal camera)
  uint32_t            debug_skip;   // -DTRT_TUNING builds only (TRT_DEBUG_SKIP): 1 = skip clear tiles, 2 = skip traced tiles
  uint32_t            vec4_ok;      // W % 4 == 0 and all first-hit streams 16-B aligned: dwordx4 clears
  uint32_t            skip_primary; // enclosure cull: test-order mask of the tori no primary ray of this frame can hit first (tubes strictly
                                    // inside a tube every ray origin of the frame lies outside of; certified on the host, trt_api.hip)
};

// A batch of frames rendered by ONE pair of launches (trt_render_batch_dev): the frames share the scene, the size, the
// tiling, the camera model, the tile lists and their counters; everything else — uniforms, push constants, toroidal
// frame, output pointers, cost words — is per frame.  Tile-list entries of a batch carry the frame in three bits
// (TileCode<true> in trt_kernels.hip).  per_frame: classification lanes per frame, a multiple of 64, so that a wave of
// the classification kernels belongs to one frame.
constexpr uint32_t kMaxBatch = TRT_MAX_BATCH;
struct RenderBatch {
  uint32_t   n_frames;
  uint32_t   per_frame;
  RenderArgs fr[kMaxBatch];
};

// Launch-shape knobs.  The release library uses the defaults below; a -DTRT_TUNING build
// (libtrt_tuning.so, tools/ only) reads each of them ONCE from the environment in trt_create.
struct Tuning {
  uint32_t min_batch          = 24;   // TRT_MIN_BATCH
  int      fine               = -1;   // TRT_FINE_CLASSIFY: -1 = chosen per frame (camera model / eye position)
  int      no_tile_cull       = 0;    // TRT_NO_TILE_CULL
  uint32_t debug_skip         = 0;    // TRT_DEBUG_SKIP (timing ablations: the frame is then INCOMPLETE)
  int      no_enclosure       = 0;    // TRT_NO_ENCLOSURE: no enclosure cull (every torus tested by every query; same images, A/B timing)
  int      debug_tiles        = 0;    // TRT_DEBUG_TILES: print the list lengths after every frame (synchronises)
  uint32_t heavy_x16          = 24;   // TRT_HEAVY_X16: a macro tile is HEAVY above heavy_x16/16 (1.5) x the mean cost; 0 = no cost feedback
  uint32_t heavy_min_tori     = 2;    // TRT_HEAVY_MIN_TORI: cost feedback only for scenes with at least this many tori (below)
  uint64_t persist_blocks     = 0;    // TRT_PERSIST_BLOCKS   (0 = default)
  uint64_t listed_blocks      = 0;    // TRT_LISTED_BLOCKS
  int      static_tile        = 8;    // TRT_TILE
  uint64_t trace_blocks       = 0;    // TRT_TRACE_BLOCKS
  uint64_t post_blocks_per_cu = 0;    // TRT_POST_BLOCKS_PER_CU
  uint64_t splat_blocks_per_cu = 0;   // TRT_SPLAT_BLOCKS_PER_CU
  int      trace_variant      = -1;   // TRT_TRACE_VARIANT
  int      splat_variant      = -1;   // TRT_SPLAT_VARIANT
};
Tuning tuning_from_env();   // defaults in the release build

struct TraceArgs {
  trt_rays rays;
  trt_hits hits;
  float    tmin, tmax;
  unsigned long long* stats;
};

enum RenderVariant { kRenderStatic = 0, kRenderPersistent = 1, kRenderListed = 2 };
constexpr int kPersistentBlocksPerCU = 16;  // 4× the resident 4 blocks/CU: the dispatcher evens out the tile costs

hipError_t launch_post(const float* in, uint64_t n, float* f32_out, uint8_t* u8_out, int n_cus, const Tuning& tn,
                       hipStream_t stream);
// Scratch of the re-projection: the binned form (mode != 0: kSplatBinWords zero-initialised words + the 12-B records of
// the mode, splat_plan) or the one-pass form (keys: W·H 64-bit words).
constexpr size_t kSplatBinWords   = 5 * 8192 + 64;   // count / offset / cursor / 64-bit page state for the largest bin count, + the ticket, pool and error words
constexpr size_t kSplatRecordSize = 12;
constexpr size_t kSplatSortBins = 2048, kSplatSortChunk = 4096;   // = kSortBins, kSortChunk of trt_kernels.hip
enum SplatMode { kSplatOnePass = 0, kSplatSorted = 1, kSplatDirect = 2, kSplatPaged = 3 };
// What a call needs (splat_plan): the form it takes and the bytes of ctx scratch behind `records`, laid out as
// [records | proj | table | page_bin], every part 16-byte aligned.
struct SplatPlan {
  int      mode;        // SplatMode
  uint32_t n_bins;
  uint32_t page_shift;  // paged: a page holds 1 << page_shift records
  uint32_t pool_pages;  // paged: pages behind the bins' first pages
  size_t   rec_bytes, proj_bytes, table_bytes, pagebin_bytes;
  size_t   total() const { return rec_bytes + proj_bytes + table_bytes + pagebin_bytes; }
};
struct SplatScratch {
  SplatPlan           plan;
  uint32_t*           bin_words;
  void*               records;   // plan.total() bytes
  unsigned long long* keys;
};
SplatPlan  splat_plan(uint32_t W, uint32_t H, float point_size, uint64_t n_points, const Tuning& tn);
hipError_t launch_splat(const trt_point* pts, uint64_t n_points, const float* vp, uint32_t W, uint32_t H,
                        const float* clear, float point_size, const SplatScratch& sc, float* rgba, int n_cus,
                        const Tuning& tn, hipStream_t stream);
hipError_t launch_trace(const SceneK& scene, const TraceArgs& a, const Tuning& tn, hipStream_t stream);
hipError_t launch_zero_words(unsigned int* words, uint32_t n, hipStream_t stream);
hipError_t launch_render(const SceneK& scene, const RenderArgs& a, RenderVariant v, int n_cus, const Tuning& tn,
                         hipStream_t stream);
// listed variant, default solver, no RenderedData: b.fr[0 .. n_frames) filled like the RenderArgs of launch_render, with
// the SAME lists / counters / capacities in every frame (capacities = tiles of all frames together)
hipError_t launch_render_batch(const SceneK& scene, const RenderBatch& b, int n_cus, const Tuning& tn, hipStream_t stream);

}  // namespace trt
