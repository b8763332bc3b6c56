// hello_hip.cpp — see hello_hip.hpp.  Host code only: every ray is traced by libtrt.so.
#include "hello_hip.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace {

using mat4 = std::array<float, 16>;  // column-major, element (row r, col c) = m[c*4 + r] (nvmath::mat4f)

mat4 mul(const mat4& a, const mat4& b)
{
  mat4 r{};
  for(int c = 0; c < 4; ++c)
    for(int row = 0; row < 4; ++row)
    {
      float s = 0.f;
      for(int k = 0; k < 4; ++k) s += a[k * 4 + row] * b[c * 4 + k];
      r[c * 4 + row] = s;
    }
  return r;
}

// general 4x4 inverse by cofactors (the role of nvmath::invert, hello_vulkan.cpp:67-68)
mat4 invert(const mat4& m)
{
  mat4 inv{};
  inv[0]  = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8]  = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5]  = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2]  = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7]  = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const float id  = 1.f / det;
  for(float& v : inv) v *= id;
  return inv;
}

// right-handed look-at (camera looks down -z), the view matrix CameraManip.getMatrix() returns
mat4 lookAt(const std::array<float, 3>& eye, const std::array<float, 3>& center, const std::array<float, 3>& up)
{
  auto sub = [](auto a, auto b) { return std::array<float, 3>{a[0] - b[0], a[1] - b[1], a[2] - b[2]}; };
  auto dot = [](auto a, auto b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto cross = [](auto a, auto b) { return std::array<float, 3>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; };
  auto norm = [&](auto a) { const float l = std::sqrt(dot(a, a)); return std::array<float, 3>{a[0] / l, a[1] / l, a[2] / l}; };
  const auto f = norm(sub(center, eye)), s = norm(cross(f, up)), u = cross(s, f);
  mat4 m{};
  m[0] = s[0]; m[4] = s[1]; m[8]  = s[2];  m[12] = -dot(s, eye);
  m[1] = u[0]; m[5] = u[1]; m[9]  = u[2];  m[13] = -dot(u, eye);
  m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2]; m[14] = dot(f, eye);
  m[15] = 1.f;
  return m;
}

// Vulkan-convention perspective (y down, depth 0..1): the shape of nvmath::perspectiveVK(fov, aspect, n, f)
mat4 perspectiveVK(float fovDeg, float aspect, float n, float f)
{
  const float t = std::tan(fovDeg * 0.017453292519943295f * 0.5f);
  mat4 m{};
  m[0]  = 1.f / (aspect * t);
  m[5]  = -1.f / t;
  m[10] = f / (n - f);
  m[14] = (f * n) / (n - f);
  m[11] = -1.f;
  return m;
}

void hipCheck(hipError_t e, const char* what)
{
  if(e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

void HelloHip::check(int rc, const char* what) const
{
  if(rc != TRT_OK) throw std::runtime_error(std::string(what) + ": " + trt_last_error(m_ctx));
}

void HelloHip::setup(int device)
{
  m_device = device;
  const int rc = trt_create(device, &m_ctx);
  if(rc != TRT_OK) throw std::runtime_error(std::string("trt_create: ") + trt_last_error(nullptr));
}

void HelloHip::createOffscreenRender(uint32_t w, uint32_t h)
{
  hipCheck(hipSetDevice(m_device), "hipSetDevice");
  if(m_dColor) hipCheck(hipFree(m_dColor), "hipFree");
  if(m_dRendered) hipCheck(hipFree(m_dRendered), "hipFree");
  m_dColor = nullptr; m_dRendered = nullptr;
  m_size = {w, h};
  const size_t n = (size_t)w * h;
  hipCheck(hipMalloc((void**)&m_dColor, n * 4 * sizeof(float)), "hipMalloc(color)");            // rgba32f, hello_vulkan.h:123
  hipCheck(hipMalloc((void**)&m_dRendered, n * sizeof(trt_rendered_data)), "hipMalloc(rData)"); // BEF createRtDescriptorSet
  m_hostColor.assign(n * 4, 0.f);
  m_hostRendered.assign(n, trt_rendered_data{});
}

int HelloHip::addMaterial(const trt_material& m)
{
  m_materials.push_back(m);
  return (int)m_materials.size() - 1;
}

void HelloHip::addTorus(const float center[3], float R, float r, int matId)
{
  m_tori.push_back(trt_torus{{center[0], center[1], center[2]}, R, r, matId});
}

void HelloHip::destroyResources()
{
  if(m_dColor) (void)hipFree(m_dColor);
  if(m_dRendered) (void)hipFree(m_dRendered);
  if(m_dPost) (void)hipFree(m_dPost);
  if(m_dCloud) (void)hipFree(m_dCloud);
  m_dColor = nullptr; m_dRendered = nullptr; m_dPost = nullptr; m_dCloud = nullptr;
  if(m_ctx) trt_destroy(m_ctx);
  m_ctx = nullptr;
}

void HelloHip::setLookat(const std::array<float, 3>& eye, const std::array<float, 3>& center,
                         const std::array<float, 3>& up, float fovDegrees)
{
  m_eye = eye; m_center = center; m_up = up; m_fov = fovDegrees;
}

void HelloHip::updateUniformBuffer()
{
  const float aspect = m_size.width / static_cast<float>(m_size.height);        // hello_vulkan.cpp:60
  const mat4  view   = lookAt(m_eye, m_center, m_up);
  const mat4  proj   = perspectiveVK(m_fov, aspect, 0.1f, 1000.0f);             // :63
  const mat4  vp = mul(proj, view), vi = invert(view), pi = invert(proj);      // :66-68
  std::memcpy(m_globals.viewProj, vp.data(), sizeof vp);
  std::memcpy(m_globals.viewInverse, vi.data(), sizeof vi);
  std::memcpy(m_globals.projInverse, pi.data(), sizeof pi);
  std::memcpy(m_globals.center, m_center.data(), sizeof m_globals.center);      // BEF :70
}

void HelloHip::raytrace(void* stream, const std::array<float, 4>& clearColor)
{
  // Initializing push constant values (hello_vulkan.cpp:917-920)
  std::memcpy(m_pcRay.clearColor, clearColor.data(), sizeof m_pcRay.clearColor);
  std::memcpy(m_pcRay.lightPosition, m_pcRaster.lightPosition, sizeof m_pcRay.lightPosition);
  m_pcRay.lightIntensity = m_pcRaster.lightIntensity;
  m_pcRay.lightType      = m_pcRaster.lightType;
  const trt_scene scene{m_tori.data(), (uint32_t)m_tori.size(), m_materials.data(), (uint32_t)m_materials.size()};
  // vkCmdTraceRaysKHR(..., width, height, 1) (:931)
  check(trt_render_dev(m_ctx, &m_globals, &m_pcRay, &scene, m_size.width, m_size.height, 0, m_size.height,
                       m_camera, m_dColor, nullptr, m_dRendered, stream),
        "HelloHip::raytrace");
}

void HelloHip::copyRenderedPosition(void* stream)
{
  hipCheck(hipMemcpyAsync(m_hostRendered.data(), m_dRendered, m_hostRendered.size() * sizeof(trt_rendered_data),
                          hipMemcpyDeviceToHost, (hipStream_t)stream), "copyRenderedPosition");
  hipCheck(hipStreamSynchronize((hipStream_t)stream), "copyRenderedPosition");
}

void HelloHip::copyColorImage(void* stream)
{
  hipCheck(hipMemcpyAsync(m_hostColor.data(), m_dColor, m_hostColor.size() * sizeof(float), hipMemcpyDeviceToHost,
                          (hipStream_t)stream), "copyColorImage");
  hipCheck(hipStreamSynchronize((hipStream_t)stream), "copyColorImage");
}

// The three writers keep the reference's file names, record order and number formatting
// (default ostream precision, "x y z\n"; std::to_string(rho) in the name).
void HelloHip::writeRenderedPosition(const char* dir)
{
  std::ofstream outfile(std::string(dir) + "data/renderedPosition" + std::to_string(m_pcRay.rho) + ".txt");
  for(const trt_rendered_data& r : m_hostRendered)                               // index x*H + y
    outfile << r.pos[0] << " " << r.pos[1] << " " << r.pos[2] << "\n";
}

void HelloHip::writeRenderedRays(const char* dir)
{
  std::ofstream origins(std::string(dir) + "data/origins.txt");
  for(const trt_rendered_data& r : m_hostRendered)
    origins << r.rayOrigin[0] << " " << r.rayOrigin[1] << " " << r.rayOrigin[2] << "\n";
  std::ofstream directions(std::string(dir) + "data/directions.txt");
  for(const trt_rendered_data& r : m_hostRendered)
    directions << r.rayDir[0] << " " << r.rayDir[1] << " " << r.rayDir[2] << "\n";
}

void HelloHip::writeColorImage(const char* dir)
{
  writeColorImageAs(std::string(dir) + "data/renderedColor" + std::to_string(m_pcRay.rho) + ".txt");
}

void HelloHip::writeColorImageAs(const std::string& path)
{
  std::ofstream outfile(path);
  if(!outfile) throw std::runtime_error("cannot write " + path);
  for(uint32_t y = 0; y < m_size.height; ++y)
    for(uint32_t x = 0; x < m_size.width; ++x)
    {
      const float* p = &m_hostColor[((size_t)y * m_size.width + x) * 4];
      outfile << p[0] << " " << p[1] << " " << p[2] << std::endl;
    }
}

// ---------------------------------------------------------------------------------------------
// post pass
// ---------------------------------------------------------------------------------------------
void HelloHip::drawPost(void* stream)
{
  const size_t n = (size_t)m_size.width * m_size.height;
  if(!m_dPost) hipCheck(hipMalloc((void**)&m_dPost, n * 4), "hipMalloc(post)");
  check(trt_post_dev(m_ctx, m_dColor, n, nullptr, m_dPost, stream), "HelloHip::drawPost");
}

void HelloHip::writePostImagePPM(const std::string& path) const
{
  const size_t n = (size_t)m_size.width * m_size.height;
  if(m_hostPost.size() != n * 4) throw std::runtime_error("writePostImagePPM: call drawPost() and copyPostImage() first");
  std::ofstream out(path, std::ios::binary);
  if(!out) throw std::runtime_error("cannot write " + path);
  out << "P6\n" << m_size.width << " " << m_size.height << "\n255\n";
  std::vector<uint8_t> rgb(n * 3);
  for(size_t i = 0; i < n; ++i)
  {
    rgb[3 * i] = m_hostPost[4 * i];
    rgb[3 * i + 1] = m_hostPost[4 * i + 1];
    rgb[3 * i + 2] = m_hostPost[4 * i + 2];
  }
  out.write(reinterpret_cast<const char*>(rgb.data()), (std::streamsize)rgb.size());
}

void HelloHip::copyPostImage(void* stream)
{
  m_hostPost.resize((size_t)m_size.width * m_size.height * 4);
  hipCheck(hipMemcpyAsync(m_hostPost.data(), m_dPost, m_hostPost.size(), hipMemcpyDeviceToHost, (hipStream_t)stream), "copyPostImage");
  hipCheck(hipStreamSynchronize((hipStream_t)stream), "copyPostImage");
}

// ---------------------------------------------------------------------------------------------
// point-cloud re-projection (ray_tracing__before_second)
// ---------------------------------------------------------------------------------------------
namespace {
// one "x y z" text file -> vec3 list; "-nan" fields and unreadable lines become lowest()
void loadVec3File(const std::string& filename, std::vector<std::array<float, 3>>& out)
{
  std::ifstream in(filename);
  if(!in) throw std::runtime_error("loadPoints: cannot open " + filename);
  const float low = std::numeric_limits<float>::lowest();
  std::string line;
  while(std::getline(in, line))
  {
    std::array<float, 3> v{low, low, low};
    std::istringstream   iss(line);
    std::string          f[3];
    if(iss >> f[0] >> f[1] >> f[2])
      for(int k = 0; k < 3; ++k)
        v[k] = f[k].find("-nan") != std::string::npos ? low : std::stof(f[k]);
    out.push_back(v);
  }
}
}  // namespace

void HelloHip::loadPoints(const std::string& positionFile, const std::string& colorFile)
{
  m_positions.clear();
  m_colors.clear();
  loadVec3File(positionFile, m_positions);
  loadVec3File(colorFile, m_colors);
}

void HelloHip::createCloudDataBuffer()
{
  if(m_positions.size() != m_colors.size())
    throw std::runtime_error("Number of positions and colors don't match!");   // SEC :636-639
  m_cloudData.resize(m_positions.size());
  for(size_t i = 0; i < m_positions.size(); ++i)
    m_cloudData[i] = trt_point{{m_positions[i][0], m_positions[i][1], m_positions[i][2], 0.f},
                               {m_colors[i][0], m_colors[i][1], m_colors[i][2], 0.f}};   // vec4(…, 0), :646-647
  if(m_dCloud) hipCheck(hipFree(m_dCloud), "hipFree");
  m_dCloud = nullptr;
  if(!m_cloudData.empty())
  {
    hipCheck(hipMalloc((void**)&m_dCloud, m_cloudData.size() * sizeof(trt_point)), "hipMalloc(cloud)");
    hipCheck(hipMemcpy(m_dCloud, m_cloudData.data(), m_cloudData.size() * sizeof(trt_point), hipMemcpyHostToDevice), "upload cloud");
  }
}

void HelloHip::rasterize(void* stream, const std::array<float, 4>& clearColor)
{
  // vkCmdDraw(numPoints, 1, 0, 0) on the POINT_LIST pipeline, gl_PointSize = 2.5 (SEC :313-330)
  check(trt_splat_dev(m_ctx, m_dCloud, m_cloudData.size(), m_globals.viewProj, m_size.width, m_size.height,
                      clearColor.data(), 2.5f, m_dColor, stream),
        "HelloHip::rasterize");
}
