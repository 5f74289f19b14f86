// hip-renderer.hpp — the adapter a yart maintainer drops into the reference tree (e.g. as src/hip/hip-renderer.hpp):
// a yart::Renderer (src/core/renderer.hpp:17-104) that renders on libyart_hip.so instead of the CPU tile threads of
// yart::cpu::TileRenderer (src/cpu/tile-renderer.hpp:22-310). Same public knobs and callbacks; the tonemapper stays
// the reference's own host object, applied where tile-renderer.hpp:234-239 applies it.
//
// Two things the reference keeps private decide the adapter's shape:
//  * yart::Camera has no accessors (core/camera.hpp: only getRay is public), so the adapter is given the camera
//    parameters as a YartCameraDesc — the values main.cpp:32-76 passes to the Camera constructor / moveAndLookAt;
//  * ParametricBSDF has no accessors either (bsdf/parametric.hpp:51-73), so a loaded yart::Scene cannot be
//    flattened from outside: the device scene is imported from the asset by the library itself
//    (yart::hip::DeviceScene::fromGltf — the same mapping as src/gltf/gltf.cpp), or described by the loader.
// `Renderer::scene` is therefore not read; `deviceScene` is. A null deviceScene renders nothing, as a null scene
// does in the reference (cpu/integrator.cpp:6). (INTEGRATION.md "Limits" gives the two accessor lines in the reference
// that would let the adapter flatten `scene` itself.)
// Callbacks: onRenderWaveComplete per wave; onRenderTileComplete — if set — per finished tile (finishTile,
// tile-renderer.hpp:243-262), fed by yart_hip_render_tiles; TileData.rays is the block's own ray count of the wave.
//
// Compiled against the reference's headers by `make -C oracle ref_hip` (oracle/adapter_main.cpp) and run on the GPU
// by tests/test_adapter.py. Needs: -I<reference>/src -I<this repo>/include -L<this repo>/yart_amd -lyart_hip.
#pragma once
#include <algorithm>
#include <atomic>
#include <exception>
#include <chrono>
#include <thread>
#include <vector>

#include <core/core.hpp>
#include <core/renderer.hpp>
#include <core/tonemapping.hpp>

#include <yart_hip.hpp>

namespace yart::hip_backend {

class HipRenderer final : public yart::Renderer {
public:
  // TileRenderer's knobs (tile-renderer.hpp:27-32); threadCount has no meaning here
  uint32_t samples = 64, firstWaveSamples = 64, maxWaveSamples = 128, tileSize = 64;
  uint32_t maxDepth = 30;                              // RayIntegrator::m_maxDepth (cpu/ray-integrator.hpp:14)
  const tonemap::Tonemap* tonemapper = nullptr;
  const yart::hip::DeviceScene* deviceScene = nullptr;
  const yart::hip::MultiDeviceScene* multiScene = nullptr;   // if set: every GPU it names renders the frame (yart_hip_multi_render)
  uint32_t maxBatchPaths = 0;                          // how many tiles finish together (0: a whole wave; YartRenderParams.max_batch_paths)
  YartCameraDesc cameraDesc{};                         // width / height are taken from the buffer

  HipRenderer(Buffer&& buffer, const Camera& camera) noexcept
    : Renderer(std::move(buffer), camera), m_hdr(size_t(m_buffer.width()) * m_buffer.height() * 4, 0.0f) {}
  ~HipRenderer() { if (m_worker.joinable()) m_worker.join(); }

  void render() override {
    wait();
    m_aborted = false;
    m_failure = nullptr;
    m_worker = std::thread([this] {
      // a device error must not leave the thread function (std::terminate would take the host application down):
      // it is kept for wait() and the caller is told through onRenderAborted
      try {
        const RenderData d = renderSync();
        const auto& cb = m_aborted ? onRenderAborted : onRenderComplete;
        if (cb) (*cb)(d);
      } catch (...) {
        m_failure = std::current_exception();
        if (onRenderAborted) (*onRenderAborted)(RenderData{m_buffer, m_taken, samples, m_rays, elapsed()});
      }
    });
  }
  void abort() override { m_aborted = true; }          // takes effect after the batch in flight
  void wait() override {
    if (m_worker.joinable()) m_worker.join();
    if (m_failure) { auto e = m_failure; m_failure = nullptr; std::rethrow_exception(e); }
  }

  RenderData renderSync() override {
    m_t0 = std::chrono::high_resolution_clock::now();
    m_taken = 0; m_rays = 0;
    if (deviceScene || multiScene) {
      YartCameraDesc cam = cameraDesc;
      cam.width = m_buffer.width(); cam.height = m_buffer.height();
      YartRenderParams p{};
      p.samples = samples; p.first_wave_samples = std::min(firstWaveSamples, samples); p.max_wave_samples = maxWaveSamples;
      p.tile_size = tileSize; p.max_depth = maxDepth;
      p.background[0] = backgroundColor[0]; p.background[1] = backgroundColor[1]; p.background[2] = backgroundColor[2];
      p.rank = 0; p.world_size = 1;
      p.max_batch_paths = maxBatchPaths;
      YartStats st{};
      int rc;
      if (multiScene) {
        // all GPUs of the node: every wave rendered by all of them, merged and reported — the same callbacks as with one device
        rc = yart_hip_multi_render_tiles(multiScene->handle(), &cam, &p, m_hdr.data(), &st, &HipRenderer::onWave,
                                         onRenderTileComplete ? &HipRenderer::onTile : nullptr, this);
      } else if (onRenderTileComplete) {
        // the wave schedule of tile-renderer.hpp:264-289 with a callback per finished tile (finishTile, :243-262)
        rc = yart_hip_render_tiles(deviceScene->handle(), &cam, &p, m_hdr.data(), &st, &HipRenderer::onWave, &HipRenderer::onTile, this);
      } else {
        // one library call; it walks the wave schedule and reports every wave with the frame blended so far in m_hdr
        // (the reference's m_hdrBuffer); a non-zero return from the callback is abort()
        rc = yart_hip_render_waves(deviceScene->handle(), &cam, &p, m_hdr.data(), &st, &HipRenderer::onWave, this);
      }
      if (rc != YART_ABORTED) yart::hip::check(rc);
    }
    return {m_buffer, m_taken, samples, m_rays, elapsed()};
  }

private:
  // m_buffer := tonemapped (or plain) copy of a rectangle of the linear frame, as finishTile does per tile (tile-renderer.hpp:234-239)
  void expose(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
    const uint32_t W = m_buffer.width();
    std::unique_lock lock(m_bufferMutex);
    for (uint32_t y = y0; y < y0 + h; y++)
      for (uint32_t x = x0; x < x0 + w; x++) {
        const float* px = &m_hdr[(size_t(y) * W + x) * 4];
        const float4 hdr(px[0], px[1], px[2], px[3]);
        m_buffer(x, y) = tonemapper ? float4((*tonemapper)(float3(hdr)), 1.0f) : hdr;
      }
  }
  static int onWave(void* user, const YartStats* st, uint32_t wave, uint32_t waveSamples, uint32_t taken, uint32_t total) {
    HipRenderer& r = *static_cast<HipRenderer*>(user);
    if (!r.onRenderTileComplete) r.expose(0, 0, r.m_buffer.width(), r.m_buffer.height());   // (tiles have been exposed one by one)
    r.m_taken = taken; r.m_rays += st->rays;
    if (r.onRenderWaveComplete)
      (*r.onRenderWaveComplete)(RenderData{r.m_buffer, taken, total, r.m_rays, r.elapsed()},
                                WaveData{wave, waveSamples, st->rays, std::chrono::milliseconds(int64_t(st->ms_device))});
    return r.m_aborted ? 1 : 0;
  }
  static int onTile(void* user, const YartTileInfo* t) {
    HipRenderer& r = *static_cast<HipRenderer*>(user);
    r.expose(t->x, t->y, t->width, t->height);
    if (r.onRenderTileComplete)
      (*r.onRenderTileComplete)(RenderData{r.m_buffer, size_t(t->samples_taken - t->wave_samples), t->total_samples, r.m_rays, r.elapsed()},
                                TileData{uint2(t->x, t->y), uint2(t->width, t->height), t->index, t->total, t->rays,
                                         std::chrono::milliseconds(int64_t(t->ms))});
    return r.m_aborted ? 1 : 0;
  }
  std::chrono::milliseconds elapsed() const {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - m_t0);
  }
  std::chrono::high_resolution_clock::time_point m_t0;
  size_t m_taken = 0;
  uint64_t m_rays = 0;
  std::vector<float> m_hdr;
  std::thread m_worker;
  std::atomic<bool> m_aborted{false};
  std::exception_ptr m_failure;
};

}  // namespace yart::hip_backend
