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
#include <atomic>
#include <chrono>
#include <cstdint>
#include <exception>
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
// the loaded library must speak the ABI these wrappers were compiled against (struct layouts, flag values)
inline void requireAbi() {
  if (yart_hip_abi_version() != YART_HIP_ABI_VERSION) throw Error(YART_E_INVALID, "libyart_hip.so and yart_hip.h disagree on YART_HIP_ABI_VERSION");
}

// Owns a device-resident scene (flattened geometry, BVHs, materials, lights).
class DeviceScene {
 public:
  explicit DeviceScene(const YartSceneDesc& desc, int device = -1) { requireAbi(); check(yart_hip_scene_create(&desc, device, &h_)); }
  explicit DeviceScene(const std::string& yscnPath, int device = -1) { requireAbi(); check(yart_hip_scene_load(yscnPath.c_str(), device, &h_)); }
  // a glTF 2.0 / GLB asset, as gltf::load + the frontend's environment light (src/gltf/gltf.cpp:319-358, src/main.cpp:78-86)
  static DeviceScene fromGltf(const std::string& path, const std::string& envHdrPath = "", float envRadius = 100.0f, int device = -1) {
    YartImportOptions o{};
    o.env_hdr_path = envHdrPath.empty() ? nullptr : envHdrPath.c_str();
    o.env_radius = envRadius;
    YartScene* h = nullptr;
    requireAbi();
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

// The scene replicated on several GPUs of this node (yart_hip_multi_*): TileRenderer's worker pool with a GPU per worker.
class MultiDeviceScene {
 public:
  MultiDeviceScene(const std::string& path, const std::vector<int>& devices, const std::string& envHdrPath = "", float envRadius = 100.0f) {
    YartImportOptions o{};
    o.env_hdr_path = envHdrPath.empty() ? nullptr : envHdrPath.c_str();
    o.env_radius = envRadius;
    check(yart_hip_multi_load(path.c_str(), &o, devices.data(), uint32_t(devices.size()), &h_));
  }
  MultiDeviceScene(const YartSceneDesc& desc, const std::vector<int>& devices) {
    check(yart_hip_multi_create(&desc, devices.data(), uint32_t(devices.size()), &h_));
  }
  MultiDeviceScene(const MultiDeviceScene&) = delete;
  MultiDeviceScene& operator=(const MultiDeviceScene&) = delete;
  ~MultiDeviceScene() { yart_hip_multi_destroy(h_); }
  YartMulti* handle() const { return h_; }
  int deviceCount() const { return yart_hip_multi_device_count(h_); }
  // replicas a device failure took out of service (their pixel blocks are rendered on the first device from then on); empty normally
  std::vector<int> failedReplicas() const {
    std::vector<int> out(64);
    const int n = yart_hip_multi_failed_devices(h_, out.data(), uint32_t(out.size()));
    out.resize(size_t(n < 64 ? n : 64));
    return out;
  }

 private:
  YartMulti* h_ = nullptr;
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
  struct TileData {                         // renderer.hpp:40-50 (rays: 0, the device counts rays per wave)
    uint32_t offset[2], size[2];
    size_t index, total;
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
  uint32_t maxBatchPaths = 0;               // YartRenderParams.max_batch_paths: how many tiles finish together (0: a whole wave)
  const DeviceScene* scene = nullptr;
  const MultiDeviceScene* multiScene = nullptr;   // if set: all its devices render every wave of the frame (same callbacks)
  RenderCallback<RenderData> onRenderComplete, onRenderAborted;
  RenderCallback<RenderData, WaveData> onRenderWaveComplete;
  RenderCallback<RenderData, TileData> onRenderTileComplete;

  HipTileRenderer(Buffer&& buffer, const YartCameraDesc& camera) : camera_(camera), buffer_(std::move(buffer)) {
    camera_.width = buffer_.width(); camera_.height = buffer_.height();
  }
  ~HipTileRenderer() { if (worker_.joinable()) worker_.join(); }

  void render() {                           // async, notifies through the callbacks
    wait();
    aborted_ = false;
    failure_ = nullptr;
    worker_ = std::thread([this] {
      // an exception must not leave the thread function (std::terminate): it is kept and rethrown by wait(), and the
      // caller is told through onRenderAborted
      try {
        RenderData d = renderSync();
        auto& cb = aborted_ ? onRenderAborted : onRenderComplete;
        if (cb) (*cb)(d);
      } catch (...) {
        failure_ = std::current_exception();
        if (onRenderAborted) (*onRenderAborted)(RenderData{buffer_, taken_, samples, rays_, elapsed()});
      }
    });
  }
  void abort() { aborted_ = true; }         // takes effect after the batch in flight (a wave, or max_batch_paths of it)
  void wait() {                             // joins; rethrows what the render thread failed with
    if (worker_.joinable()) worker_.join();
    if (failure_) { auto e = failure_; failure_ = nullptr; std::rethrow_exception(e); }
  }

  RenderData renderSync() {
    t0_ = std::chrono::high_resolution_clock::now();
    taken_ = 0; rays_ = 0;
    YartStats st{};
    if (scene || multiScene) {              // integrator.cpp:6: "if (!scene) return;"
      YartRenderParams p{};
      p.samples = samples; p.first_wave_samples = firstWaveSamples < samples ? firstWaveSamples : samples;
      p.max_wave_samples = maxWaveSamples; p.tile_size = tileSize; p.max_depth = maxDepth;
      for (int i = 0; i < 3; i++) p.background[i] = backgroundColor[i];
      p.rank = 0; p.world_size = 1;
      p.estimator = estimator;
      p.max_batch_paths = maxBatchPaths;
      int rc;
      if (multiScene) {
        rc = yart_hip_multi_render_tiles(multiScene->handle(), &camera_, &p, buffer_.data(), &st, &HipTileRenderer::onWave,
                                         onRenderTileComplete ? &HipTileRenderer::onTile : nullptr, this);
      } else if (onRenderTileComplete) {
        rc = yart_hip_render_tiles(scene->handle(), &camera_, &p, buffer_.data(), &st, &HipTileRenderer::onWave, &HipTileRenderer::onTile, this);
      } else {
        rc = yart_hip_render_waves(scene->handle(), &camera_, &p, buffer_.data(), &st, &HipTileRenderer::onWave, this);
      }
      if (rc != YART_ABORTED) check(rc);
      if (tonemapLook >= 0)                   // tile-renderer.hpp:234-239, on the whole frame
        check(yart_hip_tonemap_host(buffer_.data(), buffer_.width(), buffer_.height(), tonemapLook, buffer_.data(), nullptr));
    }
    stats_ = st;
    return {buffer_, (scene || multiScene) ? taken_ : 0, samples, rays_, elapsed()};
  }
  const YartStats& stats() const { return stats_; }

 private:
  // per wave: the linear frame blended so far is in buffer_ (renderer.hpp:33-38); abort() stops after this wave.
  // RenderData.totalRays is cumulative, as the reference's m_totalRays (tile-renderer.hpp:213-214)
  static int onWave(void* user, const YartStats* st, uint32_t wave, uint32_t waveSamples, uint32_t taken, uint32_t total) {
    HipTileRenderer& r = *static_cast<HipTileRenderer*>(user);
    r.taken_ = taken; r.rays_ += st->rays;
    if (r.onRenderWaveComplete)
      (*r.onRenderWaveComplete)(RenderData{r.buffer_, taken, total, r.rays_, r.elapsed()},
                                WaveData{wave, waveSamples, st->rays, std::chrono::milliseconds(int64_t(st->ms_device))});
    return r.aborted_ ? 1 : 0;
  }
  static int onTile(void* user, const YartTileInfo* t) {
    HipTileRenderer& r = *static_cast<HipTileRenderer*>(user);
    if (r.onRenderTileComplete)
      (*r.onRenderTileComplete)(RenderData{r.buffer_, size_t(t->samples_taken - t->wave_samples), t->total_samples, r.rays_, r.elapsed()},
                                TileData{{t->x, t->y}, {t->width, t->height}, t->index, t->total, t->rays,
                                         std::chrono::milliseconds(int64_t(t->ms))});
    return r.aborted_ ? 1 : 0;
  }
  std::chrono::milliseconds elapsed() const {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0_);
  }
  size_t taken_ = 0;
  uint64_t rays_ = 0;
  std::chrono::high_resolution_clock::time_point t0_;
  YartCameraDesc camera_;
  Buffer buffer_;
  std::thread worker_;
  std::atomic<bool> aborted_{false};
  std::exception_ptr failure_;
  YartStats stats_{};
};

}  // namespace yart::hip
