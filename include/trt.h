/*
 * trt.h — C ABI of the MI355X-native toroidal ray tracer (libtrt.so).
 *
 * This is the drop-in boundary for ONE path of raffaelecicellini/toroidal_ray_tracing:
 * the ray-tracing dispatch `HelloVulkan::raytrace(cmdBuf, clearColor)` and the shader
 * binding contract behind it.  Paths below are relative to
 * vk_raytracing_tutorial_KHR/ in the reference (REFL = ray_tracing_reflections,
 * BEF = ray_tracing__before).
 *
 *   reference interface                                  replaced by
 *   ---------------------------------------------------  ----------------------------
 *   HelloVulkan::raytrace          REFL/hello_vulkan.cpp:913-935,
 *                                  BEF/hello_vulkan.cpp:936-958          trt_render*
 *   raygen binding contract        REFL/shaders/raytrace.rgen:30-35,
 *                                  BEF/shaders/raytrace.rgen:10-17       trt_globals/trt_push/outputs
 *   traceRayEXT closest/any hit    REFL/shaders/raytrace.rgen:64-75,
 *                                  REFL/shaders/raytrace.rchit:120-131   trt_trace*
 *   TLAS + ObjDesc + materials     REFL/hello_vulkan.cpp:645-683,264-273 trt_scene
 *   RenderedData SSBO              BEF/shaders/host_device.h:101-107     trt_rendered_data
 *
 * Plain C types only: pointers, sizes, PODs.  No exceptions cross this boundary; every
 * entry point returns 0 (TRT_OK) or a negative TRT_E_* code and trt_last_error() holds
 * the message.  There is NO CPU fallback inside the library: without a usable HIP
 * device trt_create() fails with TRT_E_NO_DEVICE.
 */
#ifndef TRT_H_
#define TRT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRT_VERSION_MAJOR 0
#define TRT_VERSION_MINOR 3

/* ---- status codes ------------------------------------------------------------------ */
enum {
  TRT_OK            = 0,
  TRT_E_INVALID     = -1, /* NULL / out-of-range argument                               */
  TRT_E_NO_DEVICE   = -2, /* no HIP device / device index out of range                  */
  TRT_E_HIP         = -3, /* a HIP runtime call failed (message has hipGetErrorString)  */
  TRT_E_SCENE       = -4, /* scene rejected: >TRT_MAX_TORI, bad matId, not a ring torus */
  TRT_E_NOMEM       = -5  /* device or host allocation failed                           */
};

#define TRT_MAX_TORI      8 /* BASELINE config 4: "8 nested tori (tokamak shells)"       */
#define TRT_MAX_MATERIALS 8
#define TRT_MAX_BATCH     8 /* frames per trt_render_batch_dev call                        */

/* ---- camera models ----------------------------------------------------------------- */
enum {
  TRT_CAMERA_PINHOLE  = 0, /* REFL/shaders/raytrace.rgen:42-48                          */
  TRT_CAMERA_TOROIDAL = 1  /* BEF/shaders/raytrace.rgen:21-57                           */
};

/* ---- root solver and its precision ---------------------------- */
/* Default: the Fourier–Newton walk on the depressed quartic, FP32; _F64 = BASELINE config 4
 * ("FP64 root solve", FP32 I/O).  _DK_* and _FERRARI_* select the two solvers north_star names —
 * the Durand–Kerner iteration (fixed sweep count, complex arithmetic) and Ferrari's factorisation
 * (resolvent cubic by a fixed-count Newton iteration instead of cbrt/acos): same interface and
 * outputs, several times the cost and a tolerance- / discriminant-based real-root test — kept as
 * measured alternatives.  The persistent render variant implements the default
 * solver only. */
enum { TRT_SOLVE_F32 = 0, TRT_SOLVE_F64 = 1, TRT_SOLVE_DK_F32 = 2, TRT_SOLVE_DK_F64 = 3,
       TRT_SOLVE_FERRARI_F32 = 4, TRT_SOLVE_FERRARI_F64 = 5 };

/* ---- uniform / push-constant blocks, byte-for-byte the reference's host structs ---- */

/* GlobalUniforms, BEF/shaders/host_device.h:69-75 (REFL/…:67-72 is the same without
 * `center`).  mat4 is column-major (nvmath::mat4f / GLSL): element (row r, col c) is
 * m[c*4 + r]. */
typedef struct trt_globals {
  float viewProj[16];
  float viewInverse[16];
  float projInverse[16];
  float center[3]; /* camera look-at point; used by the toroidal camera only          */
} trt_globals;      /* 204 bytes                                                        */

/* PushConstantRay, BEF/shaders/host_device.h:90-98 (REFL/…:86-93 lacks `rho`). */
typedef struct trt_push {
  float   clearColor[4];
  float   lightPosition[3];
  float   lightIntensity;
  int32_t lightType; /* 0 = point, otherwise directional (REFL rchit:81-91)             */
  int32_t maxDepth;  /* bounce-loop bound, REFL/shaders/raytrace.rgen:79                */
  float   rho;       /* toroidal camera: radius of the ray-origin circle                */
} trt_push;          /* 44 bytes                                                        */

/* WaveFrontMaterial, REFL/shaders/host_device.h:103-115 (scalar block layout). */
typedef struct trt_material {
  float   ambient[3];
  float   diffuse[3];
  float   specular[3];
  float   transmittance[3];
  float   emission[3];
  float   shininess;
  float   ior;
  float   dissolve;
  int32_t illum;
  int32_t textureId; /* must be -1: tori are untextured (REFL rchit:100)                */
} trt_material;      /* 80 bytes                                                        */

/* One analytic torus: centre C, symmetry axis +y (the reference's world-up,
 * REFL/main.cpp:95), major radius R, tube radius r, 0 < r < R.  Replaces one TLAS
 * instance + its ObjDesc (REFL/shaders/host_device.h:57-64). */
typedef struct trt_torus {
  float   center[3];
  float   R;
  float   r;
  int32_t matId;
} trt_torus; /* 24 bytes */

/* Tori may intersect and may be nested (BASELINE config 4: eight shells of one tube).  trt_render* skips, for a ray that
 * starts outside a tube, the tori whose tube lies strictly inside it — they cannot be the closest hit;
 * results and query counts are those of testing every torus.  trt_trace tests every torus for every ray. */
typedef struct trt_scene {
  const trt_torus*    tori;
  uint32_t            n_tori;      /* 1..TRT_MAX_TORI      */
  const trt_material* materials;
  uint32_t            n_materials; /* 1..TRT_MAX_MATERIALS */
} trt_scene;

/* RenderedData, BEF/shaders/host_device.h:101-107; written at index x*H + y
 * (BEF/shaders/raytrace.rgen:72-73,111-112). */
typedef struct trt_rendered_data {
  float pos[4];
  float color[4];
  float rayOrigin[4];
  float rayDir[4];
} trt_rendered_data; /* 64 bytes */

/* ---- ray / hit streams, structure-of-arrays ---------------------------------------- */
typedef struct trt_rays {
  const float* ox; const float* oy; const float* oz; /* origins                        */
  const float* dx; const float* dy; const float* dz; /* directions, any non-zero length */
  uint64_t     n;
} trt_rays;

/* Miss: t = +INFINITY, P = N = 0 (BEF/shaders/raytrace.rmiss:21), id = -1.
 * Any pointer may be NULL to skip that stream. */
typedef struct trt_hits {
  float* t;
  float* px; float* py; float* pz; /* hit point  O + t·D  (BEF rchit:134)               */
  float* nx; float* ny; float* nz; /* outward unit normal, never flipped (REFL rchit:74-75) */
  int32_t* id;                     /* index of the torus hit                            */
} trt_hits;

/* Per-frame query counters (one "test" = one ray against one torus).  primary_tests counts every
 * pixel x torus, also for the pixels a tile classification answers without tracing (their miss
 * record is written by a constant fill); traced_tests counts the tests a lane actually executed
 * (primary on traced pixels + bounce + shadow), solved_tests those that passed the bounding-volume
 * culls so that a quartic was built and walked (rows T1/T2 of SURVEY.md 8a), evaluations the
 * (f, f') evaluations of the default solver's walk. */
typedef struct trt_stats {
  uint64_t primary_tests;
  uint64_t bounce_tests;
  uint64_t shadow_tests;
  uint64_t pixels;
  uint64_t traced_tests;
  uint64_t solved_tests;
  uint64_t evaluations;
  uint64_t reserved;
} trt_stats;

typedef struct trt_ctx trt_ctx;

/* ---- lifetime ---------------------------------------------------------------------- */
int         trt_version(void);                       /* MAJOR*1000 + MINOR               */
int         trt_create(int device, trt_ctx** out);   /* one ctx per device, not re-entrant */
void        trt_destroy(trt_ctx* ctx);
const char* trt_last_error(const trt_ctx* ctx);      /* ctx may be NULL: create errors   */
int         trt_set_solver(trt_ctx* ctx, int solver);    /* one of TRT_SOLVE_*                 */

/* ---- trace(rays_in -> hits_out): closest hit of every ray against the scene -------- */
/* Host buffers: copies in, launches, copies out, synchronises. */
int trt_trace(trt_ctx* ctx, const trt_rays* in, const trt_scene* scene,
              float tmin, float tmax, trt_hits* out);
/* Device-resident buffers, asynchronous on `stream` (a hipStream_t; NULL = default). */
int trt_trace_dev(trt_ctx* ctx, const trt_rays* in_dev, const trt_scene* scene,
                  float tmin, float tmax, trt_hits* out_dev, void* stream);

/* ---- render: the faithful equivalent of HelloVulkan::raytrace ---------------------- */
/* rgba_out: W*H*4 floats, row-major, image[y][x] = (hitValue, 1)  (rgen:87); 16-byte aligned.
 * first_hit_out: optional SoA record of the depth-0 hit per pixel, row-major y*W+x. */
int trt_render(trt_ctx* ctx, const trt_globals* g, const trt_push* pc, const trt_scene* scene,
               uint32_t W, uint32_t H, int camera, float* rgba_out, trt_hits* first_hit_out);
/* The *_dev entry points allocate nothing and synchronise nothing once the ctx's scratch has been
 * sized by a first call with the same sizes, and they put only kernel nodes on the stream: a frame loop
 * on them can be captured into a hipGraph, replayed, and mixed with eager frames in any order (every
 * frame leaves the ctx's tile-list counters as it found them).  A call that would have to grow the
 * scratch, or to upload the tables of a new toroidal-camera frame, while `stream` is being captured
 * returns TRT_E_INVALID instead of allocating inside the capture.  Scratch that a later, larger call
 * replaces stays allocated until trt_destroy, so graphs captured before keep replaying correctly.
 * The only thing a frame leaves behind in the ctx is a scheduling hint: for scenes of two or more tori the
 * time its slowest wave spent on each traced tile, which the next frame uses to start the heavy tiles
 * first.  No output bit depends on it (first frame, replay, camera cut: same images, same counts).
 *
 * Rows [row_begin,row_end) only; outputs are indexed relative to the FULL image, so a
 * rank that owns a row band passes pointers to the full-frame buffers (or to buffers
 * offset by -row_begin*W elements).  rendered_dev is optional (BEF RenderedData, x*H+y). */
int trt_render_dev(trt_ctx* ctx, const trt_globals* g, const trt_push* pc, const trt_scene* scene,
                   uint32_t W, uint32_t H, uint32_t row_begin, uint32_t row_end, int camera,
                   float* rgba_dev, trt_hits* first_hit_dev, trt_rendered_data* rendered_dev,
                   void* stream);

/* Multi-GPU image tiling.  The frame is cut into groups of `group_rows`
 * consecutive rows dealt round-robin to `n_parts` owners: part p owns every row y with
 * (y / group_rows) % n_parts == p (interleaving balances the load: hit pixels cluster).
 * compact != 0: rgba / first-hit buffers hold ONLY the owned rows, packed in order (local
 * row ly = (y / (group_rows*n_parts)) * group_rows + y % group_rows) — the layout an
 * all-gather wants as its send buffer.  compact == 0: full-frame buffers, row y at y.
 * RenderedData (x*H + y) is always full-frame. */
typedef struct trt_tiling {
  uint32_t group_rows;
  uint32_t n_parts;
  uint32_t part;
  uint32_t compact;
} trt_tiling;

/* Number of rows part `part` owns in an H-row frame under `tiling`. */
uint32_t trt_tiling_rows(const trt_tiling* tiling, uint32_t H);

/* trt_render_dev restricted to the rows `tiling` assigns to tiling->part. */
int trt_render_tiled_dev(trt_ctx* ctx, const trt_globals* g, const trt_push* pc, const trt_scene* scene,
                         uint32_t W, uint32_t H, const trt_tiling* tiling, int camera,
                         float* rgba_dev, trt_hits* first_hit_dev, trt_rendered_data* rendered_dev,
                         void* stream);

/* A batch of consecutive frames of a frame loop in ONE pair of launches.  The reference records one command
 * buffer per frame and keeps several of them in flight (REFL/main.cpp:249-257: prepareFrame / the swapchain's
 * frames in flight); consecutive frames are independent of one another.  Here the frames of a batch share the
 * scene, the image size, the tiling and the camera model; each has its own uniforms, push constants and output
 * buffers (which must not overlap).  Why: a 1/8 part of a 4096² frame — what one rank of an 8-GPU job renders —
 * does not fill an MI355X, and back-to-back small launches on one stream wait for each frame's slowest tile; eight
 * such parts in one launch are the work of one full frame and run like one.  Results are those of
 * n_frames calls of trt_render_tiled_dev / trt_render_dev, bit for bit.
 * Restrictions (TRT_E_INVALID otherwise; render such frames one by one): the listed render variant and the default
 * root solver (TRT_SOLVE_F32 / _F64); no RenderedData export; W <= 65528; 1 <= n_frames <= TRT_MAX_BATCH; with the
 * toroidal camera all frames must have the same trigonometry tables, because the ctx holds one set: the same eye and
 * centre — and, when the eye is above or below the centre (BEF rgen:45-53: theta then depends on rho), the same rho; with
 * the eye at the centre's height the frames may differ in rho (the rho sweep of BEF/main.cpp:236-258).
 * tiling may be NULL (whole frames). */
typedef struct trt_frame {
  const trt_globals* g;
  const trt_push*    pc;
  float*             rgba_dev;      /* rows of this part only when tiling->compact, else the full frame */
  const trt_hits*    first_hit_dev; /* optional (NULL), device streams like trt_render_dev              */
} trt_frame;

int trt_render_batch_dev(trt_ctx* ctx, const trt_frame* frames, uint32_t n_frames, const trt_scene* scene,
                         uint32_t W, uint32_t H, const trt_tiling* tiling, int camera, void* stream);

/* ---- post pass: the tonemap of REFL/shaders/post.frag:33-37 ------------------------------ */
/* out = pow(in, 1/2.2) on all four channels (GLSL pow(x,y) = exp2(y*log2(x)); x <= 0 or NaN -> 0).
 * f32_out (n_pixels*4 floats) and/or unorm8_out (n_pixels*4 bytes, R,G,B,A, round-to-nearest of
 * clamp(out,0,1)*255 — the 8-bit image a swapchain presents; 4x smaller to all-gather) may be
 * NULL.  exp2/log2 are evaluated with a fixed fma polynomial, so the bytes are
 * reproducible bit for bit on any IEEE machine. */
int trt_post_dev(trt_ctx* ctx, const float* rgba_in_dev, uint64_t n_pixels, float* f32_out_dev,
                 uint8_t* unorm8_out_dev, void* stream);

/* ---- point-cloud re-projection: the consumer of the captures (SEC = ray_tracing__before_second) --- */
/* Point, SEC/shaders/host_device.h:113-117: what loadPoints()/createCloudDataBuffer()
 * (SEC/hello_vulkan.cpp:496-660) build from renderedPosition*.txt / renderedColor*.txt. */
typedef struct trt_point {
  float pos[4];
  float color[4];
} trt_point; /* 32 bytes */

/* Rasterises the points as the reference's POINT_LIST pipeline does (SEC/hello_vulkan.cpp:143-270,
 * 313-330; SEC/shaders/vert_shader.vert:43-52, frag_shader.frag:40-45):
 *   clip = viewProj * (pos.xyz, 1); a point whose vertex is outside the clip volume
 *   (-w <= x,y <= w, 0 <= z <= w, w > 0) is discarded; window position
 *   xf = (x/w*0.5+0.5)*W, yf = (y/w*0.5+0.5)*H, depth z/w quantised to 24-bit UNORM (the
 *   offscreen depth format X8_D24, SEC/hello_vulkan.h) with round-to-nearest;
 *   the point covers every pixel whose centre (i+0.5, j+0.5) satisfies
 *   xf - size/2 <= i+0.5 < xf + size/2 (same in y), size = gl_PointSize = 2.5;
 *   depth test LESS against a buffer cleared to 1.0, depth write on; equal depths keep the
 *   EARLIER point (primitive order); colour (color.xyz, 1); untouched pixels = clearColor.
 * rgba_dev: W*H*4 floats, row-major. */
int trt_splat_dev(trt_ctx* ctx, const trt_point* points_dev, uint64_t n_points, const float* viewProj,
                  uint32_t W, uint32_t H, const float* clearColor, float point_size, float* rgba_dev,
                  void* stream);

/* Counters of the last render or trace call made with counting enabled. */
int trt_enable_stats(trt_ctx* ctx, int on);
int trt_get_stats(trt_ctx* ctx, trt_stats* out); /* waits for the last counted launch (a graph replay: synchronise it yourself) */

/* Name of the kernel variant used by trt_render* ("listed" (default) | "persistent" | "static"). */
int         trt_set_render_variant(trt_ctx* ctx, const char* name);
const char* trt_get_render_variant(const trt_ctx* ctx);

/* Level of the tile classification of the listed / persistent variants: TRT_CLASSIFY_AUTO (default:
 * per-tile for the toroidal camera and for a pinhole camera within two bounding radii of a torus,
 * per-macro-tile otherwise), TRT_CLASSIFY_MACRO (32x8 pixels per test), TRT_CLASSIFY_TILE (8x8
 * pixels per test + distance-function march).  Never changes an output bit, only the time. */
enum { TRT_CLASSIFY_AUTO = -1, TRT_CLASSIFY_MACRO = 0, TRT_CLASSIFY_TILE = 1 };
int trt_set_classification(trt_ctx* ctx, int level);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* TRT_H_ */
