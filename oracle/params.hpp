// params.hpp — "key v0 v1 ..." text parameter files shared by the oracle tools.
// TEST INFRASTRUCTURE. The same keys are written by yart_amd/params.py.
//
// Keys mirror the reference's public knobs:
//   size W H                  Buffer(W,H)                       (src/main.cpp:27)
//   spp N / first_wave N / max_wave N / tile N / threads N      (src/cpu/tile-renderer.hpp:27-31)
//   depth N                   RayIntegrator::m_maxDepth         (src/cpu/ray-integrator.hpp:14)
//   focal F / fnumber F / sensor SX SY                          (src/core/camera.hpp:76-83)
//   eye X Y Z / target X Y Z / up X Y Z     Camera::moveAndLookAt (camera.hpp:123-130)
//   exposure EV / aperture_sides N                              (camera.hpp:62-63)
//   background R G B          Renderer::backgroundColor         (src/core/renderer.hpp:52)
//   probe_pixels x y x y ...  pixels whose per-sample radiance / primary hits are dumped
//   shard_rank R / shard_world N / shard_tile T   (oracle restatement only) render only the pixel blocks the
//                             library deals to rank R of N (include/yart_hip.h YartRenderParams.rank / world_size /
//                             shard_tile: blocks in Morton order, round-robin); the other pixels stay 0
#pragma once
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace params {

struct Params {
  uint32_t width = 256, height = 256;
  uint32_t spp = 16, firstWave = 0, maxWave = 0, tile = 64, threads = 0, depth = 30;
  float focal = 35.0f, fnumber = 0.0f, sensor[2] = {36.0f, 24.0f};
  float eye[3] = {0, 0, 0}, target[3] = {0, 0, -1}, up[3] = {0, 1, 0};
  float exposure = 0.0f;
  uint32_t apertureSides = 0;
  float background[3] = {0, 0, 0};
  std::vector<uint32_t> probePixels;  // x,y pairs
  uint32_t shardRank = 0, shardWorld = 1, shardTile = 0;
};

inline Params load(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("params: cannot open " + path);
  Params p;
  std::string line;
  bool fw = false, mw = false;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    auto f3 = [&](float* v) { ss >> v[0] >> v[1] >> v[2]; };
    if (key == "size") ss >> p.width >> p.height;
    else if (key == "spp") ss >> p.spp;
    else if (key == "first_wave") { ss >> p.firstWave; fw = true; }
    else if (key == "max_wave") { ss >> p.maxWave; mw = true; }
    else if (key == "tile") ss >> p.tile;
    else if (key == "threads") ss >> p.threads;
    else if (key == "depth") ss >> p.depth;
    else if (key == "focal") ss >> p.focal;
    else if (key == "fnumber") ss >> p.fnumber;
    else if (key == "sensor") ss >> p.sensor[0] >> p.sensor[1];
    else if (key == "eye") f3(p.eye);
    else if (key == "target") f3(p.target);
    else if (key == "up") f3(p.up);
    else if (key == "exposure") ss >> p.exposure;
    else if (key == "aperture_sides") ss >> p.apertureSides;
    else if (key == "background") f3(p.background);
    else if (key == "shard_rank") ss >> p.shardRank;
    else if (key == "shard_world") ss >> p.shardWorld;
    else if (key == "shard_tile") ss >> p.shardTile;
    else if (key == "probe_pixels") { uint32_t v; while (ss >> v) p.probePixels.push_back(v); }
    else throw std::runtime_error("params: unknown key " + key);
  }
  if (!fw) p.firstWave = p.spp;   // single wave, as src/main.cpp:97-99 configures it
  if (!mw) p.maxWave = p.spp;
  return p;
}

}  // namespace params
