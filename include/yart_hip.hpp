// yart_hip.hpp — C++ host-side mirror of the reference's renderer interface over the C ABI.
//
// The reference's seam is `yart::Renderer` (reference src/core/renderer.hpp:17-104) as
// implemented by `yart::cpu::TileRenderer<Sampler, Integrator>`
// (src/cpu/tile-renderer.hpp:22-310): public knobs, `render()` (async), `abort()`,
// `wait()`, `renderSync()` returning `RenderData{buffer, samplesTaken, totalSamples,
// totalRays, totalTime}`. `yart::hip::HipTileRenderer` keeps those names, argument
// meanings and the "null scene -> empty render" behaviour (src/cpu/integrator.cpp:6), and
// adds what a device boundary needs: error codes become exceptions (`yart::hip::Error`).
//
// Header-only; link against libyart_hip.so. INTEGRATION.md shows the adapter that plugs
// this into the reference's `main()` / frontend in place of TileRenderer.
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "yart_hip.h"

namespace yart::hip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* msg) : std::runtime_error(std::string("yart_hip: ") + msg), code(c) {}
};
inline void check(int rc) { if (rc != YART_OK) throw Error(rc, yart_hip_last_error()); }

// Owns a device-resident scene (flattened geometry, BVHs, materials, lights).
class DeviceScene {
 public:
  explicit DeviceScene(const YartSceneDesc& desc, int device = -1) { check(yart_hip_scene_create(&desc, device, &h_)); }
  explicit DeviceScene(const std::string& yscnPath, int device = -1) { check(yart_hip_scene_load(yscnPath.c_str(), device, &h_)); }
  // a glTF 2.0 / GLB asset, as gltf::load + the frontend's environment light (src/gltf/gltf.cpp:319-358, src/main.cpp:78-86)
  static DeviceScene fromGltf(const std::string& path, const std::string& envHdrPath = "", float envRadius = 100.0f, int device = -1) {
    YartImportOptions o{};
    o.env_hdr_path = envHdrPath.empty() ? nullptr : envHdrPath.c_str();
    o.env_radius = envRadius;
    YartScene* h = nullptr;
    check(yart_hip_scene_load_gltf(path.c_str(), &o, device, &h));
    return DeviceScene(h);
  }
  DeviceScene(DeviceScene&& other) noexcept : h_(other.h_) { other.h_ = nullptr; }
  DeviceScene(const DeviceScene&) = delete;
  DeviceScene& operator=(const DeviceScene&) = delete;
  ~DeviceScene() { yart_hip_scene_destroy(h_); }
  YartScene* handle() const { return h_; }

 private:
  explicit DeviceScene(YartScene* h) : h_(h) {}
  YartScene* h_ = nullptr;
};

// RGBA32F framebuffer, row-major, alpha = 1 (reference src/core/buffer.hpp:12-47)
class Buffer {
 public:
  Buffer(uint32_t w, uint32_t h) : w_(w), h_(h), data_(size_t(w) * h * 4, 0.0f) {}
  uint32_t width() const { return w_; }
  uint32_t height() const { return h_; }
  const float* operator()(size_t x, size_t y) const { return &data_[(y * w_ + x) * 4]; }
  float* data() { return data_.data(); }
  const float* data() const { return data_.data(); }

 private:
  uint32_t w_, h_;
  std::vector<float> data_;
};

class HipTileRenderer {
 public:
  struct RenderData {                       // renderer.hpp:22-28
    const Buffer& buffer;
    size_t samplesTaken, totalSamples;
    uint64_t totalRays;
    std::chrono::milliseconds totalTime;
  };
  struct WaveData {                         // renderer.hpp:33-38
    size_t wave, waveSamples;
    uint64_t rays;
    std::chrono::milliseconds time;
  };
  template <class... Ts> using RenderCallback = std::optional<std::function<void(Ts...)>>;

  // knobs of TileRenderer (tile-renderer.hpp:27-32) and Renderer (renderer.hpp:52-58)
  uint32_t samples = 64, firstWaveSamples = 64, maxWaveSamples = 128, tileSize = 64;
  uint32_t maxDepth = 30;                   // RayIntegrator::m_maxDepth (ray-integrator.hpp:14)
  float backgroundColor[3] = {0, 0, 0};
  uint32_t estimator = YART_ESTIMATOR_GMON; // core/estimator.hpp class (integrator.cpp:17-18 fixes it at compile time)
  int tonemapLook = -1;                     // TileRenderer::tonemapper: -1 none (linear HDR), 0 AgX none, 1 golden, 2 punchy
  const DeviceScene* scene = nullptr;
  RenderCallback<RenderData> onRenderComplete, onRenderAborted;
  RenderCallback<RenderData, WaveData> onRenderWaveComplete;

  HipTileRenderer(Buffer&& buffer, const YartCameraDesc& camera) : camera_(camera), buffer_(std::move(buffer)) {
    camera_.width = buffer_.width(); camera_.height = buffer_.height();
  }
  ~HipTileRenderer() { wait(); }

  void render() {                           // async, notifies through the callbacks
    wait();
    aborted_ = false;
    worker_ = std::thread([this] {
      RenderData d = renderSync();
      auto& cb = aborted_ ? onRenderAborted : onRenderComplete;
      if (cb) (*cb)(d);
    });
  }
  void abort() { aborted_ = true; }         // takes effect after the wave in flight
  void wait() { if (worker_.joinable()) worker_.join(); }

  RenderData renderSync() {
    auto t0 = std::chrono::high_resolution_clock::now();
    YartStats st{};
    if (scene) {                            // integrator.cpp:6: "if (!scene) return;"
      YartRenderParams p{};
      p.samples = samples; p.first_wave_samples = firstWaveSamples < samples ? firstWaveSamples : samples;
      p.max_wave_samples = maxWaveSamples; p.tile_size = tileSize; p.max_depth = maxDepth;
      for (int i = 0; i < 3; i++) p.background[i] = backgroundColor[i];
      p.rank = 0; p.world_size = 1;
      p.estimator = estimator;
      taken_ = 0; t0_ = t0;
      const int rc = yart_hip_render_waves(scene->handle(), &camera_, &p, buffer_.data(), &st, &HipTileRenderer::onWave, this);
      if (rc != YART_ABORTED) check(rc);
      if (tonemapLook >= 0)                   // tile-renderer.hpp:234-239, on the whole frame
        check(yart_hip_tonemap_host(buffer_.data(), buffer_.width(), buffer_.height(), tonemapLook, buffer_.data(), nullptr));
    }
    auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0);
    stats_ = st;
    return {buffer_, scene ? taken_ : 0, samples, st.rays, ms};
  }
  const YartStats& stats() const { return stats_; }

 private:
  // per wave: the linear frame blended so far is in buffer_ (renderer.hpp:33-38); abort() stops after this wave
  static int onWave(void* user, const YartStats* st, uint32_t wave, uint32_t waveSamples, uint32_t taken, uint32_t total) {
    HipTileRenderer& r = *static_cast<HipTileRenderer*>(user);
    r.taken_ = taken;
    if (r.onRenderWaveComplete) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - r.t0_);
      (*r.onRenderWaveComplete)(RenderData{r.buffer_, taken, total, st->rays, ms},
                                WaveData{wave, waveSamples, st->rays, std::chrono::milliseconds(int64_t(st->ms_device))});
    }
    return r.aborted_ ? 1 : 0;
  }
  size_t taken_ = 0;
  std::chrono::high_resolution_clock::time_point t0_;
  YartCameraDesc camera_;
  Buffer buffer_;
  std::thread worker_;
  volatile bool aborted_ = false;
  YartStats stats_{};
};

}  // namespace yart::hip
