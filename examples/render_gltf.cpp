// render_gltf.cpp — what the reference's src/main.cpp does, on libyart_hip.so:
//   load a glTF / GLB asset (main.cpp:78), add the octahedral .hdr environment (:80-83), set the camera (:32-76),
//   render with the TileRenderer knobs (:92-99), AgX-tonemap (:88-89, tile-renderer.hpp:234-239) and write out.ppm
//   (frontend main.cpp:263-275 -> output/ppm.cpp:7-21). No window: the frontend is out of scope.
//
//   g++ -std=c++17 -Iinclude examples/render_gltf.cpp -Lyart_amd -lyart_hip -Wl,-rpath,$PWD/yart_amd -lpthread -o render_gltf
//   ./render_gltf asset.glb sky_oct.hdr out.ppm [width height spp depth  eye(3) target(3)  focal fnumber exposure look]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "yart_hip.hpp"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s asset.glb env_oct.hdr|- out.ppm [w h spp depth ex ey ez tx ty tz focal fnumber exposure look]\n", argv[0]);
    return 1;
  }
  auto num = [&](int i, double dflt) { return argc > i ? std::atof(argv[i]) : dflt; };
  const uint32_t w = uint32_t(num(4, 1920)), h = uint32_t(num(5, 1200));          // main.cpp:27
  try {
    const std::string env = std::string(argv[2]) == "-" ? "" : argv[2];
    yart::hip::DeviceScene scene = yart::hip::DeviceScene::fromGltf(argv[1], env, 100.0f);

    YartCameraDesc cam{};                                                             // Camera({w, h}, 35.0f, 4.0f), main.cpp:32
    cam.width = w; cam.height = h;
    cam.focal_length = float(num(14, 35.0)); cam.f_number = float(num(15, 4.0));
    cam.sensor[0] = 36.0f; cam.sensor[1] = 24.0f;                                     // camera.hpp default sensor
    cam.position[0] = float(num(8, 8.5)); cam.position[1] = float(num(9, 1.8)); cam.position[2] = float(num(10, 0.0));
    cam.target[0] = float(num(11, 0.0)); cam.target[1] = float(num(12, 3.2)); cam.target[2] = float(num(13, 0.0));
    cam.up[0] = 0.0f; cam.up[1] = 1.0f; cam.up[2] = 0.0f;
    cam.exposure = float(num(16, 5.0));                                               // main.cpp:34
    cam.aperture_sides = 0;

    yart::hip::HipTileRenderer renderer(yart::hip::Buffer(w, h), cam);
    renderer.scene = &scene;
    renderer.samples = uint32_t(num(6, 2048));                                        // main.cpp:97-99
    renderer.maxWaveSamples = renderer.samples;
    renderer.firstWaveSamples = renderer.samples;
    renderer.maxDepth = uint32_t(num(7, 30));
    renderer.tonemapLook = int(num(17, 0));                                           // AgX::none, main.cpp:88-89; -1: no tonemapper
    const auto done = renderer.renderSync();

    std::vector<uint8_t> rgb8(size_t(w) * h * 3);                                    // output::writePPM without a tonemap pass
    yart::hip::check(yart_hip_tonemap_host(done.buffer.data(), w, h, -1, nullptr, rgb8.data()));
    FILE* f = std::fopen(argv[3], "wb");
    if (!f) { std::fprintf(stderr, "cannot create %s\n", argv[3]); return 2; }
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    std::fwrite(rgb8.data(), 1, rgb8.size(), f);
    std::fclose(f);
    const auto& st = renderer.stats();
    std::printf("{\"samples\": %zu, \"rays\": %llu, \"ms\": %lld, \"ms_device\": %.1f, \"msamples_per_s\": %.1f}\n", done.totalSamples,
                (unsigned long long) done.totalRays, (long long) done.totalTime.count(), st.ms_device,
                double(w) * h * renderer.samples / (st.ms_device * 1e3));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
