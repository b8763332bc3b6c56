// toroidal_sweep_main.cpp — the capture loop of ray_tracing__before/main.cpp: toroidal camera,
// rho swept 4.5 → 10.0 in steps of 0.5 (main.cpp:236-258,337-341), each capture dumped as
// renderedPosition<rho>.txt / renderedColor<rho>.txt plus origins.txt / directions.txt
// (main.cpp:376-395), exit at rho == 10 (main.cpp:399-402).
// Usage: toroidal_sweep outdir/ [width height]   (outdir/ must contain a data/ directory)
#include <cstdio>
#include <cstdlib>

#include "../toroidal_ray_tracing_amd/host/hello_hip.hpp"

int main(int argc, char** argv)
{
  if(argc < 2) { std::fprintf(stderr, "usage: %s outdir/ [width height]\n", argv[0]); return 2; }
  const uint32_t W = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 256;
  try
  {
    HelloHip helloVk;
    helloVk.setup(0);
    helloVk.createOffscreenRender(W, H);
    helloVk.m_camera = TRT_CAMERA_TOROIDAL;
    trt_material plastic{};
    plastic.ambient[0] = plastic.ambient[1] = plastic.ambient[2] = 0.05f;
    plastic.diffuse[0] = 0.7f; plastic.diffuse[1] = 0.2f; plastic.diffuse[2] = 0.2f;
    plastic.specular[0] = plastic.specular[1] = plastic.specular[2] = 0.5f;
    plastic.shininess = 24.f; plastic.ior = 1.f; plastic.dissolve = 1.f; plastic.illum = 2; plastic.textureId = -1;
    const float c[3] = {0, 0, 0};
    helloVk.addTorus(c, 14.0f, 3.0f, helloVk.addMaterial(plastic));  // the camera circle sits in the hole
    helloVk.setLookat({0.f, 0.f, 0.f}, {10.f, 0.f, 0.f}, {0.f, 1.f, 0.f});  // main.cpp:124
    const std::array<float, 4> clearColor{1, 1, 1, 1};
    for(float rho = 4.5f; rho <= 10.0f; rho += 0.5f)
    {
      helloVk.m_pcRay.rho = rho;
      helloVk.updateUniformBuffer();
      helloVk.raytrace(nullptr, clearColor);
      helloVk.copyRenderedPosition(nullptr);
      helloVk.copyColorImage(nullptr);
      helloVk.writeRenderedPosition(argv[1]);
      helloVk.writeColorImage(argv[1]);
      if(rho == 4.5f) helloVk.writeRenderedRays(argv[1]);
      std::printf("rho = %g written\n", rho);
    }
  }
  catch(const std::exception& e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
