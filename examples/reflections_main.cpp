// reflections_main.cpp — the frame loop of ray_tracing_reflections/main.cpp:82-344 without the
// window: set-up, then per frame updateUniformBuffer → raytrace (→ copyColorImage →
// writeColorImage when saving).  Usage: reflections [width height frames maxDepth outdir/]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../toroidal_ray_tracing_amd/host/hello_hip.hpp"

int main(int argc, char** argv)
{
  const uint32_t W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;  // main.cpp:73-74
  const int frames = argc > 3 ? atoi(argv[3]) : 60, depth = argc > 4 ? atoi(argv[4]) : 10;
  const char* outdir = argc > 5 ? argv[5] : nullptr;
  try
  {
    HelloHip helloVk;
    helloVk.setup(0);
    helloVk.createOffscreenRender(W, H);
    trt_material mirror{};  // ray_tracing_reflections/README.md:30-38: illum 3, Ks 0.95
    mirror.specular[0] = mirror.specular[1] = mirror.specular[2] = 0.95f;
    mirror.shininess = 32.f; mirror.ior = 1.f; mirror.dissolve = 1.f; mirror.illum = 3; mirror.textureId = -1;
    const float c[3] = {0, 0, 0};
    helloVk.addTorus(c, 1.0f, 0.25f, helloVk.addMaterial(mirror));
    helloVk.setLookat({0.f, 1.5f, -4.f}, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.f});
    helloVk.m_pcRay.maxDepth = depth;
    const std::array<float, 4> clearColor{1, 1, 1, 1};  // main.cpp:212
    const auto t0 = std::chrono::steady_clock::now();
    for(int f = 0; f < frames; ++f)
    {
      helloVk.updateUniformBuffer();         // main.cpp:260
      helloVk.raytrace(nullptr, clearColor); // main.cpp:287
    }
    helloVk.copyColorImage(nullptr);         // main.cpp:315-318 (synchronises)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%ux%u, maxDepth %d: %d frames in %.2f ms (%.3f ms/frame incl. readback of the last)\n", W, H, depth,
                frames, ms, ms / frames);
    if(outdir)
    {
      helloVk.writeColorImage(outdir);  // main.cpp:326-330
      // the ground-truth dump of ray_tracing_reflections/hello_vulkan.cpp:1065-1110 ("data/<scene>gTruth.txt") and
      // the presented image (post.frag) as a PPM
      helloVk.writeColorImageAs(std::string(outdir) + "data/torusgTruth.txt");
      helloVk.drawPost(nullptr);
      helloVk.copyPostImage(nullptr);
      helloVk.writePostImagePPM(std::string(outdir) + "data/torusgTruth.ppm");
    }
    const float* px = &helloVk.colorImage()[((size_t)(H / 2) * W + W / 2) * 4];
    std::printf("centre pixel = %g %g %g %g\n", px[0], px[1], px[2], px[3]);
  }
  catch(const std::exception& e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
