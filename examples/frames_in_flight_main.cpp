// frames_in_flight_main.cpp — a frame loop with K frames in flight, written against the plain C ABI (include/trt.h).
//
// The reference renders 60 frames per camera radius and reads the image back after the last one
// (ray_tracing__before/main.cpp:337-402); consecutive frames are independent.  A trt_ctx is not re-entrant, but contexts
// are independent of one another: K contexts on K HIP streams, each with its own output set, let the tail of one frame
// (a few heavy tiles on an otherwise idle chip — DESIGN.md §5) run beside the body of the next.  Eight nested tori,
// 4096², FP64 solve: 0.31 ms per frame with one context, 0.26 ms with two.
//
// The second half is the other way to keep frames in flight, the one a rank of a multi-GPU job uses: trt_render_batch_dev renders
// `parts` consecutive frames' 1/parts part of the image (trt_tiling) with ONE pair of launches on ONE stream — as much work as a
// full frame, where a single part does not fill the chip (a 1/8 part of the 4096² single-torus frame: 41 µs per frame one by
// one, 14.5 µs eight per launch) — and checks the batch against the same frames rendered one by one, bit for bit.
//
// Usage: frames_in_flight [K=2] [frames=64] [size=4096] [nested=1] [f64=1] [parts=8]
// Prints the time per frame for 1 … K frames in flight and checks that every context produced the same image.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/trt.h"

#define CK(x)                                                                                        \
  do {                                                                                               \
    if((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } \
  } while(0)
#define TK(c, x)                                                                                     \
  do {                                                                                               \
    if((x) != TRT_OK) { std::fprintf(stderr, "trt error: %s\n", trt_last_error(c)); return 1; }     \
  } while(0)

// column-major 4×4 helpers, enough for a look-at camera with a Vulkan-style perspective (nvmath::perspectiveVK)
static void mul(const float* a, const float* b, float* o)
{
  for(int c = 0; c < 4; ++c)
    for(int r = 0; r < 4; ++r)
    {
      float s = 0;
      for(int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
      o[c * 4 + r] = s;
    }
}

int main(int argc, char** argv)
{
  const int      K = argc > 1 ? atoi(argv[1]) : 2, frames = argc > 2 ? atoi(argv[2]) : 64;
  const uint32_t W = argc > 3 ? atoi(argv[3]) : 4096, H = W;
  const bool     nested = argc > 4 ? atoi(argv[4]) != 0 : true, f64 = argc > 5 ? atoi(argv[5]) != 0 : true;
  const uint32_t parts = argc > 6 ? (uint32_t)atoi(argv[6]) : 8u;
  if(K < 1 || K > 8 || frames < 1) { std::fprintf(stderr, "K in 1..8, frames >= 1\n"); return 1; }

  // scene: BASELINE config 4 (eight nested tori, the outer shells mirrors) or config 3 (one mirror torus)
  std::vector<trt_torus>    tori;
  std::vector<trt_material> mats(2);
  std::memset(mats.data(), 0, mats.size() * sizeof(trt_material));
  mats[0].specular[0] = mats[0].specular[1] = mats[0].specular[2] = 0.95f;   // mirror (REFL/README.md:30-38)
  mats[0].shininess = 32.f; mats[0].ior = 1.f; mats[0].dissolve = 1.f; mats[0].illum = 3; mats[0].textureId = -1;
  mats[1] = mats[0];
  mats[1].diffuse[0] = 0.8f; mats[1].diffuse[1] = 0.3f; mats[1].diffuse[2] = 0.2f; mats[1].illum = 2;   // plastic
  for(int i = 0; i < (nested ? 8 : 1); ++i)
  {
    trt_torus t{};
    t.R = 1.0f;
    t.r = nested ? 0.05f * float(i + 1) : 0.25f;
    t.matId = (!nested || i >= 4) ? 0 : 1;
    tori.push_back(t);
  }
  const trt_scene scene{tori.data(), (uint32_t)tori.size(), mats.data(), (uint32_t)mats.size()};

  // camera: eye (0, 1.5, -4) looking at the origin, fov 60°, near / far 0.1 / 1000 (REFL/hello_vulkan.cpp:63)
  trt_globals g{};
  {
    const float eye[3] = {0.f, 1.5f, -4.f};
    float f[3] = {-eye[0], -eye[1], -eye[2]};
    const float fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for(float& v : f) v /= fl;
    const float up[3] = {0, 1, 0};
    float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    const float sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for(float& v : s) v /= sl;
    const float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    // view^-1 (camera to world): columns s, u, -f, eye
    const float vi[16] = {s[0], s[1], s[2], 0, u[0], u[1], u[2], 0, -f[0], -f[1], -f[2], 0, eye[0], eye[1], eye[2], 1};
    float view[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0, 0, 0, 0, 1};
    view[12] = -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]);
    view[13] = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
    view[14] = (f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]);
    const float th = std::tan(60.f * 3.14159265f / 360.f), n = 0.1f, fa = 1000.f, asp = float(W) / float(H);
    const float proj[16] = {1.f / (asp * th), 0, 0, 0, 0, -1.f / th, 0, 0, 0, 0, fa / (n - fa), -1, 0, 0, fa * n / (n - fa), 0};
    // proj^-1 of that matrix
    const float pi[16] = {asp * th, 0, 0, 0, 0, -th, 0, 0, 0, 0, 0, (n - fa) / (fa * n), 0, 0, -1, 1.f / n};
    mul(proj, view, g.viewProj);
    std::memcpy(g.viewInverse, vi, sizeof vi);
    std::memcpy(g.projInverse, pi, sizeof pi);
  }
  trt_push pc{};
  pc.clearColor[0] = pc.clearColor[1] = pc.clearColor[2] = pc.clearColor[3] = 1.f;
  pc.lightPosition[0] = 10.f; pc.lightPosition[1] = 15.f; pc.lightPosition[2] = 8.f;
  pc.lightIntensity = 100.f;
  pc.lightType = 0;
  pc.maxDepth = 5;

  std::vector<trt_ctx*>     ctx(K, nullptr);
  std::vector<hipStream_t>  stream(K);
  std::vector<float*>       image(K, nullptr);
  const size_t              bytes = (size_t)W * H * 4 * sizeof(float);
  for(int k = 0; k < K; ++k)
  {
    TK(nullptr, trt_create(0, &ctx[k]));
    if(f64) TK(ctx[k], trt_set_solver(ctx[k], TRT_SOLVE_F64));
    CK(hipStreamCreate(&stream[k]));
    CK(hipMalloc((void**)&image[k], bytes));
  }
  auto frame = [&](int k) {
    return trt_render_dev(ctx[k], &g, &pc, &scene, W, H, 0, H, TRT_CAMERA_PINHOLE, image[k], nullptr, nullptr, stream[k]);
  };
  for(int k = 0; k < K; ++k)   // sizes every context's scratch, gives the cost feedback its first history
    for(int i = 0; i < 3; ++i) TK(ctx[k], frame(k));
  CK(hipDeviceSynchronize());

  for(int inflight = 1; inflight <= K; ++inflight)
  {
    const auto t0 = std::chrono::steady_clock::now();
    for(int f = 0; f < frames; ++f) TK(ctx[f % inflight], frame(f % inflight));
    CK(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d frame(s) in flight: %.4f ms per frame (%d frames of %ux%u, %zu tori, %s solve)\n", inflight, ms / frames, frames, W, H,
                tori.size(), f64 ? "FP64" : "FP32");
  }

  // every context rendered the same frame: the images are bit-identical
  std::vector<float> a((size_t)W * H * 4), b(a.size());
  CK(hipMemcpy(a.data(), image[0], bytes, hipMemcpyDeviceToHost));
  int rc = 0;
  for(int k = 1; k < K; ++k)
  {
    CK(hipMemcpy(b.data(), image[k], bytes, hipMemcpyDeviceToHost));
    if(std::memcmp(a.data(), b.data(), bytes) != 0) { std::printf("context %d: image DIFFERS\n", k); rc = 1; }
  }
  std::printf("images of the %d contexts %s; centre pixel = %g %g %g %g\n", K, rc ? "DIFFER" : "identical",
              a[((size_t)(H / 2) * W + W / 2) * 4], a[((size_t)(H / 2) * W + W / 2) * 4 + 1], a[((size_t)(H / 2) * W + W / 2) * 4 + 2],
              a[((size_t)(H / 2) * W + W / 2) * 4 + 3]);
  // ---- a batch of frames per launch: what one rank of a `parts`-GPU job renders (part 0 of the interleaved row tiling) ----
  if(parts >= 2 && parts <= TRT_MAX_BATCH && H % (parts * 8) == 0)
  {
    uint32_t group = H / (parts * 16);              // 16 interleaved groups per rank, whole 8-row tile bands
    if(group < 8 || group % 8) group = 8;
    const trt_tiling til{group, parts, 0, 1};
    const uint32_t rows = trt_tiling_rows(&til, H);
    const size_t   pbytes = (size_t)rows * W * 4 * sizeof(float);
    std::vector<float*>    out(2 * parts, nullptr);   // [0, parts): the batch's output sets; [parts, 2 parts): one by one
    std::vector<trt_push>  pcs(parts, pc);
    std::vector<trt_frame> fr(parts);
    for(uint32_t i = 0; i < 2 * parts; ++i) CK(hipMalloc((void**)&out[i], pbytes));
    for(uint32_t i = 0; i < parts; ++i)
    {
      pcs[i].maxDepth = 1 + int(i % 5);             // the frames differ (a real loop moves the camera: g is per frame too)
      fr[i] = trt_frame{&g, &pcs[i], out[i], nullptr};
    }
    for(int r = 0; r < 3; ++r) TK(ctx[0], trt_render_batch_dev(ctx[0], fr.data(), parts, &scene, W, H, &til, TRT_CAMERA_PINHOLE, stream[0]));
    for(uint32_t i = 0; i < parts; ++i)
      TK(ctx[0], trt_render_tiled_dev(ctx[0], &g, &pcs[i], &scene, W, H, &til, TRT_CAMERA_PINHOLE, out[parts + i], nullptr, nullptr, stream[0]));
    CK(hipDeviceSynchronize());
    std::vector<float> x((size_t)rows * W * 4), y(x.size());
    for(uint32_t i = 0; i < parts; ++i)
    {
      CK(hipMemcpy(x.data(), out[i], pbytes, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y.data(), out[parts + i], pbytes, hipMemcpyDeviceToHost));
      if(std::memcmp(x.data(), y.data(), pbytes) != 0) { std::printf("batch frame %u DIFFERS from the frame rendered alone\n", i); rc = 1; }
    }
    const int reps = frames / (int)parts > 0 ? frames / (int)parts : 1;
    auto t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < reps; ++r)
      for(uint32_t i = 0; i < parts; ++i)
        TK(ctx[0], trt_render_tiled_dev(ctx[0], &g, &pcs[i], &scene, W, H, &til, TRT_CAMERA_PINHOLE, out[parts + i], nullptr, nullptr, stream[0]));
    CK(hipDeviceSynchronize());
    const double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * parts);
    t0 = std::chrono::steady_clock::now();
    for(int r = 0; r < reps; ++r) TK(ctx[0], trt_render_batch_dev(ctx[0], fr.data(), parts, &scene, W, H, &til, TRT_CAMERA_PINHOLE, stream[0]));
    CK(hipDeviceSynchronize());
    const double bat = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * parts);
    std::printf("1/%u part of the frame (%u rows in groups of %u): %.1f us per frame one by one, %.1f us per frame %u per launch; batch %s\n", parts, rows,
                group, one, bat, parts, rc ? "DIFFERS" : "identical to the frames rendered alone");
    for(float* q : out) (void)hipFree(q);
  }
  for(int k = 0; k < K; ++k)
  {
    trt_destroy(ctx[k]);
    (void)hipFree(image[k]);
    (void)hipStreamDestroy(stream[k]);
  }
  return rc;
}
