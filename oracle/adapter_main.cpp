// adapter_main.cpp — TEST INFRASTRUCTURE: a reference-side program that renders through the adapter a maintainer
// would add (integration/hip-renderer.hpp: a yart::Renderer on libyart_hip.so), built against the reference's own
// headers where they lie (`make -C oracle ref_hip` -> oracle/_ref/yart_ref_hip). It stands where src/main.cpp:19-107
// stands with `TileRenderer` swapped for `HipRenderer`:
//   yart_ref_hip <asset.glb> <env_oct.hdr|-> <params.txt> <out.f32> [look: -|none|golden|punchy] [tiles=<max batch paths>] [devices=0,1,...]
// writes the renderer's buffer (RGBA32F) and prints the wave (and, with tiles=, the tile) callbacks it received;
// devices= renders on all the listed GPUs through yart::hip::MultiDeviceScene.
#include <core/core.hpp>
#include <core/tonemapping.hpp>

#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../integration/hip-renderer.hpp"
#include "params.hpp"

using namespace yart;
using namespace yart::math;

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: yart_ref_hip asset.glb env.hdr|- params.txt out.f32 [look]\n"); return 1; }
  try {
    const auto p = params::load(argv[3]);
    const std::string env = std::string(argv[2]) == "-" ? "" : argv[2];
    const std::string look = argc > 5 ? argv[5] : "-";
    uint32_t tilesBatch = 0; bool tiles = false;
    std::vector<int> devices;
    for (int a = 6; a < argc; a++) {
      const std::string opt = argv[a];
      if (opt.rfind("tiles=", 0) == 0) { tiles = true; tilesBatch = uint32_t(std::stoul(opt.substr(6))); }
      else if (opt.rfind("devices=", 0) == 0) {
        std::string rest = opt.substr(8);
        for (size_t pos = 0; pos <= rest.size();) {
          const size_t c = rest.find(',', pos);
          devices.push_back(std::stoi(rest.substr(pos, c == std::string::npos ? std::string::npos : c - pos)));
          if (c == std::string::npos) break;
          pos = c + 1;
        }
      }
    }

    Buffer buffer(p.width, p.height);                                             // main.cpp:20-30
    Camera camera({buffer.width(), buffer.height()}, p.focal, p.fnumber);         // main.cpp:32
    camera.exposure = p.exposure;
    camera.apertureSides = p.apertureSides;
    camera.moveAndLookAt(float3(p.eye[0], p.eye[1], p.eye[2]), float3(p.target[0], p.target[1], p.target[2]),
                         float3(p.up[0], p.up[1], p.up[2]));

    yart::hip::DeviceScene scene = yart::hip::DeviceScene::fromGltf(argv[1], env, 100.0f);   // main.cpp:78-83
    std::unique_ptr<yart::hip::MultiDeviceScene> multi;
    if (!devices.empty()) multi = std::make_unique<yart::hip::MultiDeviceScene>(argv[1], devices, env, 100.0f);

    tonemap::AgX agx;                                                             // main.cpp:88-89
    agx.look = look == "golden" ? tonemap::AgX::golden : look == "punchy" ? tonemap::AgX::punchy : tonemap::AgX::none;

    yart::hip_backend::HipRenderer renderer(std::move(buffer), camera);           // main.cpp:91-94
    renderer.deviceScene = &scene;
    renderer.multiScene = multi.get();
    renderer.maxBatchPaths = tilesBatch;
    renderer.samples = p.spp;                                                     // main.cpp:96-99
    renderer.maxWaveSamples = p.maxWave;
    renderer.firstWaveSamples = p.firstWave;
    renderer.tileSize = p.tile;
    renderer.maxDepth = p.depth;
    renderer.backgroundColor = float3(p.background[0], p.background[1], p.background[2]);
    if (look != "-") renderer.tonemapper = &agx;
    // the camera values once more, in the form the device boundary takes them (Camera keeps them private)
    YartCameraDesc& c = renderer.cameraDesc;
    c.focal_length = p.focal; c.f_number = p.fnumber; c.sensor[0] = 36.0f; c.sensor[1] = 24.0f;
    for (int i = 0; i < 3; i++) { c.position[i] = p.eye[i]; c.target[i] = p.target[i]; c.up[i] = p.up[i]; }
    c.exposure = p.exposure; c.aperture_sides = p.apertureSides;

    size_t waves = 0, lastTaken = 0;
    renderer.onRenderWaveComplete = [&](Renderer::RenderData d, Renderer::WaveData wv) {
      std::printf("wave %zu: %zu samples, %zu / %zu taken\n", wv.wave, wv.waveSamples, d.samplesTaken, d.totalSamples);
      waves++; lastTaken = d.samplesTaken;
    };
    if (tiles)
      renderer.onRenderTileComplete = [&](Renderer::RenderData d, Renderer::TileData t) {
        std::printf("tile %zu/%zu at %u,%u size %ux%u, %zu taken before this wave\n", t.index, t.total, t.offset.x(), t.offset.y(),
                    t.size.x(), t.size.y(), d.samplesTaken);
      };
    const Renderer::RenderData done = renderer.renderSync();
    if (done.samplesTaken != size_t(p.spp) || (multi ? false : (lastTaken != size_t(p.spp) || waves == 0))) { std::fprintf(stderr, "incomplete render\n"); return 4; }

    std::ofstream o(argv[4], std::ios::binary);
    for (unsigned y = 0; y < done.buffer.height(); y++)
      for (unsigned x = 0; x < done.buffer.width(); x++) {
        const float4& v = done.buffer(x, y);
        const float q[4] = {v[0], v[1], v[2], v[3]};
        o.write(reinterpret_cast<const char*>(q), 16);
      }
    std::printf("rays %llu\n", (unsigned long long) done.totalRays);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "yart_ref_hip: %s\n", e.what());
    return 3;
  }
  return 0;
}
