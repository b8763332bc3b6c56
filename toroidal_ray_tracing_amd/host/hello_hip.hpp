// hello_hip.hpp — C++ host-side mirror of the reference's sample class for the ray-tracing path.
//
// The reference drives its ray tracer through `class HelloVulkan`
// (vk_raytracing_tutorial_KHR/ray_tracing_reflections/hello_vulkan.h:37-163,
//  ray_tracing__before/hello_vulkan.h): main() calls updateUniformBuffer(), raytrace(cmdBuf,
// clearColor) and — when saving — copyRenderedPosition()/copyColorImage() followed by
// writeRenderedPosition()/writeRenderedRays()/writeColorImage().  `HelloHip` keeps those
// names, argument meanings and member names (m_pcRaster, m_pcRay, m_size) for exactly that
// subset, with the Vulkan objects replaced by a trt_ctx and HIP buffers; a command buffer
// becomes a HIP stream.  Everything else of HelloVulkan (swapchain, raster pipeline, OBJ
// loading, BLAS/TLAS, SBT, post pass, ImGui) is out of scope.
//
// Scene: the TLAS of triangle instances is replaced by analytic tori (`addTorus`), materials
// keep the WaveFrontMaterial layout (`addMaterial`).
#pragma once

#include <array>
#include <string>
#include <vector>

#include "../../include/trt.h"

struct PushConstantRaster  // ray_tracing__before/shaders/host_device.h:78-86 (light state only)
{
  float lightPosition[3]{10.f, 15.f, 8.f};  // hello_vulkan.h:74-80
  float lightIntensity{100.f};
  int   lightType{0};
};

class HelloHip
{
public:
  struct Extent { uint32_t width{0}, height{0}; };

  // --- set-up (replaces setup()/createOffscreenRender()/createRtDescriptorSet()) ----------
  void setup(int device = 0);                          // creates the trt_ctx; throws std::runtime_error
  void createOffscreenRender(uint32_t w, uint32_t h);  // rgba32f image + RenderedData buffer on the device
  int  addMaterial(const trt_material& m);             // returns the material index
  void addTorus(const float center[3], float R, float r, int matId);
  void destroyResources();
  ~HelloHip() { destroyResources(); }

  // --- camera (CameraManip.setLookat + updateUniformBuffer, hello_vulkan.cpp:57-98) --------
  void setLookat(const std::array<float, 3>& eye, const std::array<float, 3>& center,
                 const std::array<float, 3>& up, float fovDegrees = 60.f);
  void updateUniformBuffer();  // viewProj / viewInverse / projInverse / center → m_globals

  // --- the path ------------------------------------------------------------------------------
  // HelloVulkan::raytrace(cmdBuf, clearColor), hello_vulkan.cpp:913-935 / 936-958: refresh the
  // push constants from m_pcRaster and clearColor, then launch width × height rays.
  void raytrace(void* stream, const std::array<float, 4>& clearColor);

  // --- readback + text dumps (ray_tracing__before/hello_vulkan.cpp:991-1259) ----------------
  void copyRenderedPosition(void* stream);  // RenderedData device → host
  void copyColorImage(void* stream);        // rgba32f image device → host
  void writeRenderedPosition(const char* dir);  // dir + "data/renderedPosition<rho>.txt": "x y z\n", index x*H+y
  void writeRenderedRays(const char* dir);      // dir + "data/origins.txt", "data/directions.txt"
  void writeColorImage(const char* dir);        // dir + "data/renderedColor<rho>.txt": row-major "r g b\n"
  // the same dump under an explicit name — ray_tracing__before_second/hello_vulkan.cpp:781-825 writes the
  // re-projected image to dir + "data/<scene>ptCloudImage_10.txt", ray_tracing_reflections/…:1065-1110 the
  // ground truth to dir + "data/<scene>gTruth.txt": one "r g b\n" line per pixel, row-major
  void writeColorImageAs(const std::string& path);

  // --- post pass (drawPost, ray_tracing_reflections/hello_vulkan.cpp:560-579 + post.frag) ------
  void drawPost(void* stream);              // tonemap m_dColor -> 8-bit image (pow(c, 1/2.2))
  void copyPostImage(void* stream);         // device -> host
  const std::vector<uint8_t>& postImage() const { return m_hostPost; }
  // The presented image as a file: binary PPM (P6, 8-bit RGB) of the tonemapped frame — what the swapchain
  // shows after post.frag (REFL/shaders/post.frag:33-37); call drawPost() + copyPostImage() first.
  void writePostImagePPM(const std::string& path) const;

  // --- point-cloud re-projection (ray_tracing__before_second) -------------------------------
  // loadPoints (SEC/hello_vulkan.cpp:496-628): "x y z" per line, "-nan" -> numeric_limits<float>::lowest()
  void loadPoints(const std::string& positionFile, const std::string& colorFile);
  void createCloudDataBuffer();             // :633-660 — throws if the two files disagree in length
  void rasterize(void* stream, const std::array<float, 4>& clearColor);  // :313-330, POINT_LIST draw into m_dColor
  size_t numPoints() const { return m_cloudData.size(); }

  // --- state, named as in the reference -----------------------------------------------------
  PushConstantRaster m_pcRaster;
  trt_push           m_pcRay{{0, 0, 0, 0}, {0, 0, 0}, 0.f, 0, 10, 0.f};  // maxDepth 10 (hello_vulkan.h:157)
  Extent             m_size;
  int                m_camera{TRT_CAMERA_PINHOLE};  // REFL: pinhole; BEF: TRT_CAMERA_TOROIDAL
  trt_globals        m_globals{};

  const std::vector<float>&             colorImage() const { return m_hostColor; }
  const std::vector<trt_rendered_data>& renderedData() const { return m_hostRendered; }
  trt_ctx*                              ctx() const { return m_ctx; }

private:
  void check(int rc, const char* what) const;

  trt_ctx*                   m_ctx{nullptr};
  int                        m_device{0};
  std::vector<trt_torus>     m_tori;
  std::vector<trt_material>  m_materials;
  std::array<float, 3>       m_eye{0, 0, 0}, m_center{10, 0, 0}, m_up{0, 1, 0};  // main.cpp:95
  float                      m_fov{60.f};
  float*                     m_dColor{nullptr};
  trt_rendered_data*         m_dRendered{nullptr};
  std::vector<float>             m_hostColor;
  std::vector<trt_rendered_data> m_hostRendered;
  uint8_t*                       m_dPost{nullptr};
  std::vector<uint8_t>           m_hostPost;
  std::vector<std::array<float, 3>> m_positions, m_colors;   // SEC: m_positions / m_colors
  std::vector<trt_point>         m_cloudData;                // SEC: m_cloudData
  trt_point*                     m_dCloud{nullptr};
};
