// reproject_main.cpp — ray_tracing__before_second/main.cpp without the window: load a capture
// (renderedPosition<rho>.txt + renderedColor<rho>.txt as written by toroidal_sweep), rasterise it
// as a point cloud from a pinhole viewpoint, tonemap, dump the image as text — the reference's
// data/<scene>ptCloudImage_10.txt (ray_tracing__before_second/hello_vulkan.cpp:781-825: "r g b" per
// pixel, row-major) — and as a PPM of the presented 8-bit image.
// Usage: reproject dir/ rho [width height]      e.g.  reproject /tmp/cap/ 4.500000 512 512
#include <cstdio>
#include <cstdlib>

#include "../toroidal_ray_tracing_amd/host/hello_hip.hpp"

int main(int argc, char** argv)
{
  if(argc < 3) { std::fprintf(stderr, "usage: %s dir/ rho [width height]\n", argv[0]); return 2; }
  const std::string dir = argv[1], rho = argv[2];
  const uint32_t W = argc > 3 ? atoi(argv[3]) : 512, H = argc > 4 ? atoi(argv[4]) : 512;
  try
  {
    HelloHip helloVk;
    helloVk.setup(0);
    helloVk.createOffscreenRender(W, H);
    helloVk.loadPoints(dir + "data/renderedPosition" + rho + ".txt", dir + "data/renderedColor" + rho + ".txt");
    // NOTE the capture stores positions in x*H+y order and colours row-major;
    // like the reference's loadPoints this driver pairs them line by line, so the capture must
    // be square or written with matching orders.
    helloVk.createCloudDataBuffer();
    helloVk.setLookat({0.f, 0.f, 0.f}, {10.f, 0.f, 0.f}, {0.f, 1.f, 0.f});   // SEC main.cpp camera
    helloVk.updateUniformBuffer();
    helloVk.rasterize(nullptr, {0.8f, 0.8f, 0.8f, 1.0f});                    // SEC main.cpp:178
    helloVk.drawPost(nullptr);
    helloVk.copyColorImage(nullptr);
    helloVk.copyPostImage(nullptr);
    helloVk.writeColorImageAs(dir + "data/torusptCloudImage_10.txt");   // SEC writeColorImage (:797-806 naming)
    helloVk.writePostImagePPM(dir + "data/torusptCloudImage_10.ppm");
    size_t drawn = 0;
    for(size_t i = 0; i < (size_t)W * H; ++i)
      drawn += helloVk.colorImage()[4 * i] != 0.8f || helloVk.colorImage()[4 * i + 1] != 0.8f;
    std::printf("%zu points -> %ux%u, %zu pixels covered, centre byte %u\n", helloVk.numPoints(), W, H, drawn,
                (unsigned)helloVk.postImage()[((size_t)(H / 2) * W + W / 2) * 4]);
  }
  catch(const std::exception& e)
  {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
