/* yart_hip.h — C ABI of the MI355X-native path-tracing integrator (libyart_hip.so).
 *
 * The reference (teofum/yart) has no plugin / FFI layer: its seam is the abstract
 * C++ class yart::Renderer (reference src/core/renderer.hpp:17-104) implemented by
 * yart::cpu::TileRenderer (src/cpu/tile-renderer.hpp:22-310). This header is what
 * a binding for that seam calls: plain C, POD structs, caller-owned buffers, no
 * C++/torch types. INTEGRATION.md shows the adapter class a maintainer of the
 * reference would add on top of it.
 *
 * Every entry point returns 0 on success or a negative YART_E_* code;
 * yart_hip_last_error() gives the message. There is NO CPU fallback: without a
 * HIP device every compute entry point fails with YART_E_NO_DEVICE.
 */
#ifndef YART_HIP_H
#define YART_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YART_HIP_ABI_VERSION 3

enum {
  YART_OK = 0,
  YART_E_INVALID = -1,    /* bad descriptor (null pointer, index out of range, ...) */
  YART_E_NO_DEVICE = -2,  /* no usable HIP device */
  YART_E_HIP = -3,        /* a HIP runtime call failed */
  YART_E_IO = -4,         /* scene file could not be read */
  YART_E_RCCL = -5,       /* an RCCL call of the multi-device merge failed */
  YART_ABORTED = 1        /* yart_hip_render_waves: the wave callback asked to stop (the frame holds the waves done) */
};

/* Texture as the reference holds it after load (src/core/texture.hpp:21-49):
 * u8 with 1-4 channels or float RGB; sRGB-typed data is gamma-2 encoded
 * (texture.hpp:78-84). type: 0 LinearRGB, 1 sRGB, 2 NonColor. */
typedef struct YartTextureDesc {
  uint32_t width, height, channels, is_float, type;
  const void* data;
} YartTextureDesc;

/* Constructor arguments of ParametricBSDF (src/bsdf/parametric.hpp:16-37);
 * tex_* are indices into YartSceneDesc.textures or -1. */
typedef struct YartMaterialDesc {
  float base[3];
  float emission[3];
  float metallic, roughness, transmission, ior;
  float anisotropic, aniso_rotation, clearcoat, clearcoat_roughness;
  float normal_scale;
  uint32_t thin_transmission;
  float volume_color[3];
  float volume_density;
  int32_t tex_base, tex_mr, tex_transmission, tex_normal, tex_clearcoat, tex_emission;
} YartMaterialDesc;

/* Arguments of Mesh(vertices, vertexData, faces) (src/core/mesh.hpp:54-61) plus the
 * per-triangle light index the loader assigns (src/gltf/gltf.cpp:299-309). */
typedef struct YartMeshDesc {
  uint32_t n_vertices, n_faces;
  const float* positions;   /* 3 per vertex */
  const float* normals;     /* 3 per vertex */
  const float* tangents;    /* 4 per vertex (xyz + handedness) */
  const float* uvs;         /* 2 per vertex */
  const uint32_t* faces;    /* 4 per face: i0, i1, i2, material */
  const int32_t* face_light;/* 1 per face: light index or -1 */
} YartMeshDesc;

/* Scene-graph node (src/core/scene.hpp:11-64), pre-order, node 0 = root.
 * fwd / inv: row-major 4x4 Transform matrices (src/math/transform.hpp). */
typedef struct YartNodeDesc {
  int32_t parent, mesh;
  float fwd[16], inv[16];
} YartNodeDesc;

/* Light (src/core/light.hpp): type 0 AreaLight(tri of mesh, emission, transform),
 * 1 UniformInfiniteLight(radius, emission), 2 ImageInfiniteLight(radius, float RGB
 * octahedral texture) with its public `transform`. */
typedef struct YartLightDesc {
  uint32_t type;
  int32_t mesh;
  uint32_t tri, two_sided;
  int32_t texture;
  float radius;
  float emission[3];
  float fwd[16], inv[16];
} YartLightDesc;

typedef struct YartSceneDesc {
  uint32_t n_textures, n_materials, n_meshes, n_nodes, n_lights;
  const YartTextureDesc* textures;
  const YartMaterialDesc* materials;
  const YartMeshDesc* meshes;
  const YartNodeDesc* nodes;
  const YartLightDesc* lights;
} YartSceneDesc;

/* Camera(imageSize, focalLength, fNumber, sensorSize) + moveAndLookAt + exposure /
 * apertureSides (src/core/camera.hpp:62-130). */
typedef struct YartCameraDesc {
  uint32_t width, height;
  float focal_length, f_number;
  float sensor[2];
  float position[3], target[3], up[3];
  float exposure;
  uint32_t aperture_sides;
} YartCameraDesc;

/* TileRenderer knobs (src/cpu/tile-renderer.hpp:27-32), Renderer::backgroundColor
 * (src/core/renderer.hpp:52), RayIntegrator::m_maxDepth (src/cpu/ray-integrator.hpp:14).
 * rank/world_size: this process renders the pixel blocks b with b % world_size == rank (blocks of shard_tile,
 * by default tile_size, pixels numbered in Morton order) and leaves the other pixels 0. */
typedef struct YartRenderParams {
  uint32_t samples, first_wave_samples, max_wave_samples, tile_size, max_depth;
  float background[3];
  uint32_t rank, world_size;
  uint32_t flags;            /* YART_FLAG_* */
  /* Resumable accumulation (the reference keeps its blended m_hdrBuffer between waves, tile-renderer.hpp:93,
   * 220-232): render only the waves that cover samples [start_sample, stop_sample) of the schedule that
   * `samples / first_wave_samples / max_wave_samples` define; both must lie on wave boundaries of that
   * schedule (stop_sample 0 = samples). With start_sample > 0 the output buffer must hold the frame
   * accumulated so far and is blended into, exactly as an uninterrupted render would continue. */
  uint32_t start_sample, stop_sample;
  /* Per-pixel estimator (core/estimator.hpp): the reference's Integrator::render picks one at compile time
   * (cpu/integrator.cpp:17-18: GMoNEstimator(samples, 15) as shipped, MeanEstimator in the commented line);
   * MoN and GMoNb take the same (samples, 15). 0 keeps the shipped behaviour. */
  uint32_t estimator;        /* YART_ESTIMATOR_* */
  /* Edge of the square pixel blocks dealt to the ranks (Morton order, round-robin); 0 = tile_size, the
   * reference's unit of parallel work. Which process renders a pixel does not change it (the sampler only
   * knows tile_size), so a smaller block only evens out the load between GPUs. */
  uint32_t shard_tile;
  /* Upper bound on the (pixel, sample) paths per batch; 0 = 2^28 (a fixed number: the memory a render holds does not depend on
   * what happens to be free on the device; only a device that cannot hold the batch renders smaller ones). A wave is rendered
   * batch by batch over this rank's pixels in tile order, every batch through all bounces and the estimator: 251 bytes per path
   * of the batch (path state, queues, per-sample radiance, the compacted state of the late bounces). A smaller batch means
   * finished tiles arrive earlier (yart_hip_render_tiles) and less memory is held, at ~10 ms per batch on an MI355X (the C3 frame
   * of 531 M paths: +1.4 % in 2 batches, +4.5 % in 4, +19 % in 16). The frame does not depend on it. */
  uint32_t max_batch_paths;
  /* ABI 3. With YART_FLAG_PATH_POOL: the path slots of the pool (0 = 2^25: 5.6 GB). The paths of a batch are started in these
   * slots, and whenever a path ends its slot takes the batch's next (pixel, sample) — path regeneration, as a worker of the
   * reference takes the next tile the moment it has finished one (tile-renderer.hpp:161-167) — so the path state is bounded
   * by the pool (168 bytes per slot) and only 16 bytes per path of the batch remain. The frame does not depend on it. */
  uint32_t pool_paths;
} YartRenderParams;
#define YART_ESTIMATOR_GMON 0u      /* core/estimator.hpp:148-198 */
#define YART_ESTIMATOR_MEAN 1u      /* :29-46 */
#define YART_ESTIMATOR_MON 2u       /* :53-92 */
#define YART_ESTIMATOR_GMONB 3u     /* :94-146 */

#define YART_FLAG_MEGAKERNEL 1u     /* single-kernel integrator instead of the wavefront pipeline */
#define YART_FLAG_NO_REFILL 16u     /* one-ray-per-lane lean kernels instead of the ones with in-wave ray
                                       replacement (trace_lean.hpp) */
#define YART_FLAG_SHADE_SORT 2u     /* bucket each wave's 256 shade-queue entries by lobe class before shading them: the default
                                       since round 2 (measured: shade stage -8.4 % on the McLaren-class scene, -0.5 % on the
                                       Sponza-class one; round 1 bucketed by material index and lost 5 % there) */
#define YART_FLAG_NO_SHADE_SORT 64u /* shade the queue entries in queue order */
#define YART_FLAG_DIRECT_SAMPLER 8u /* evaluate every ZSobol index digit per draw (no per-render sampler tables) */
#define YART_FLAG_NO_COMPACTION 32u  /* keep every bounce on the batch-sized path state (no copy of the survivors into a dense one) */
#define YART_FLAG_NO_RESUME 128u    /* rays the lean traversal kernels hand to the general ones are traced again from the root instead of
                                       being taken up where the lean kernel stood (the default; same frame either way) */
#define YART_FLAG_GENERAL_TRACE 4u  /* general traversal kernels for every ray instead of lean kernels + retry */
/* value 256 (YART_FLAG_WIDE_TREES of rounds 4-5: 8-wide trees of the lean kernels' own, walked one ray per lane, then by eight lanes per
   ray) is retired and ignored: both forms were bit-identical and slower than the walk of the reference's tree
   (profiles/r4_ab_lean_tree.txt, profiles/r5_ab_coop_tree.txt) */
#define YART_FLAG_PATH_POOL 512u    /* ABI 3: run a batch through a pool of pool_paths path slots with path regeneration (a slot whose path has ended
                                       takes the batch's next path) instead of one slot per path of the batch. Same frame either way */
/* ABI history. 3: YART_FLAG_PATH_POOL (and the since-retired value 256); YartStats grew (wide_* fields at the end, now always 0); value 128 has meant NO_RESUME since the end of
 * ABI 2 (it selected a since-removed 4-wide re-layout before: an old client passing it gets the same frame, a little slower);
 * YartRenderParams grew (pool_paths); max_batch_paths = 0 now means a fixed 2^28 paths, no longer a share of the free device memory; value 1024 is retired and ignored; YartTileInfo.rays is a real count; yart_hip_multi_render_tiles was added. */

/* Renderer::RenderData counters (src/core/renderer.hpp:22-28) + per-stage device time. */
typedef struct YartStats {
  uint64_t samples;          /* pixel samples taken by this rank */
  uint64_t rays;             /* path segments + unoccluded shadow rays (mis-integrator.cpp:22,126) */
  double ms_total;           /* wall time of the call */
  double ms_device;          /* HIP-event time of all kernels */
  double ms_traverse;        /* HIP-event time of the traversal kernels (wavefront: extend + connect;
                                megakernel: the whole path kernel) */
  uint64_t traversals;       /* rays traced (closest-hit + shadow) */
  uint64_t box_tests, tri_tests;   /* exact counts when collected (instrumented build), else 0 */
  uint32_t waves;            /* progressive waves rendered */
  uint32_t launches_traverse;      /* launches summed into ms_traverse */
  uint64_t shaded_hits;      /* instrumented build only */
  double ms_extend, ms_shade, ms_connect, ms_gmon;   /* per-stage HIP-event time (wavefront pipeline) */
  uint32_t launches_extend, launches_connect;
  /* the lean closest-hit kernel (k_wf_extend_fast) alone: HIP-event time, launches, and its share
     of the exact test counters (instrumented build) — the roofline kernel of bench.py */
  double ms_extend_lean;
  uint64_t lean_traversals, lean_box_tests, lean_tri_tests;
  uint32_t launches_extend_lean, reserved0;
  /* the shade kernel (k_wf_shade) and the lean any-hit kernel (k_wf_shadow_lean) alone, as above */
  double ms_shade_kernel, ms_shadow_lean;
  uint32_t launches_shade_kernel, launches_shadow_lean;
  uint64_t shadow_lean_traversals, shadow_lean_box_tests, shadow_lean_tri_tests;
  uint64_t shade_entries;      /* instrumented build: queue entries the shade kernel processed (hits + misses) */
  uint64_t texture_tap_bytes;  /* instrumented build: 4 taps x channels x texel bytes summed over every texture lookup */
  uint32_t pipeline_flags;     /* the YART_FLAG_* set this render ran with, after the per-scene defaults */
  uint32_t reserved1;
  /* instrumented build: rays the lean kernels abandoned at an alpha-tested / transparent candidate and the general
     kernels traced again from the root (they are counted in lean_traversals / shadow_lean_traversals as well) */
  uint64_t retry_extend_traversals, retry_shadow_traversals;
  /* ABI 3: counters of the retired 8-wide walk (value 256 above); kept for the layout, always 0 */
  uint64_t wide_extend_nodes, wide_extend_tris, wide_shadow_nodes, wide_shadow_tris;
  uint64_t wide_extend_handed[4], wide_shadow_handed[4];
  /* ABI 3, batch-synchronous wavefront pipeline: paths of this rank that entered bounce b (b < 16; [0] = every path), summed over
     the batches: how the work of a rank decays with the depth (an imbalance between ranks shows here first) */
  uint64_t paths_at_bounce[16];
} YartStats;

typedef struct YartScene YartScene;

/* Build the device scene (BVH build per mesh — on the device, see yart_hip_scene_create_flags —, flattening, upload). device < 0: current. */
int yart_hip_scene_create(const YartSceneDesc* desc, int device, YartScene** out);
/* ... with options. The BVH of every mesh is built on the device (yart_hip_bvh_build_device: the same node array and index
 * permutation as the host build, so the same frames; a mesh the device build refuses is built on the host) unless
 * YART_SCENE_HOST_BVH — or the environment variable YART_HOST_BVH — asks for the host builder. YART_SCENE_DEVICE_BVH names the default. */
#define YART_SCENE_DEVICE_BVH 1u
#define YART_SCENE_HOST_BVH 2u
int yart_hip_scene_create_flags(const YartSceneDesc* desc, int device, uint32_t scene_flags, YartScene** out);
/* Same, from a .yscn container (yart_amd/yscn.py). */
int yart_hip_scene_load(const char* path, int device, YartScene** out);
void yart_hip_scene_destroy(YartScene* scene);

/* glTF 2.0 / GLB import — what `gltf::load(path)` (src/gltf/gltf.cpp:319-358) followed
 * by the frontend's environment set-up (src/main.cpp:78-86) gives the renderer: materials with the KHR
 * transmission / ior / anisotropy / clearcoat / volume / emissive_strength extensions (gltf.cpp:62-176),
 * gamma-2 re-encoded textures (core/texture.hpp:62-92), the primitives of each mesh merged (gltf.cpp:178-270),
 * the T*R*S node tree and one AreaLight per emissive triangle with per-node light indices (gltf.cpp:272-317).
 * Embedded PNG and JPEG (baseline / progressive Huffman) images are decoded to the bytes the reference's
 * stb_image call yields; arithmetic-coded / CMYK JPEG images and sparse accessors are refused (YART_E_IO, see yart_hip_last_error). opts may be NULL (asset only). env_hdr_path: octahedral-mapped Radiance .hdr wrapped
 * in ImageInfiniteLight(env_radius, texture) (core/texture.cpp:5-20); uniform_env != 0 adds
 * UniformInfiniteLight(env_radius, uniform_emission). env_radius <= 0 means 100 (main.cpp:82). */
typedef struct YartImportOptions {
  const char* env_hdr_path;
  float env_radius;
  uint32_t uniform_env;
  float uniform_emission[3];
  uint32_t reserved[4];
} YartImportOptions;
int yart_hip_scene_load_gltf(const char* path, const YartImportOptions* opts, int device, YartScene** out);
/* Host only (no device needed): the imported scene written as a .yscn container — the same bytes
 * yart_hip_scene_load reads, and what oracle/ takes to render the asset with the reference. */
int yart_hip_gltf_to_yscn(const char* gltf_path, const YartImportOptions* opts, const char* yscn_path);

/* Blocking render (Renderer::renderSync). out_rgba: caller-owned host buffer of
 * width*height*4 floats, linear HDR with exposure applied, alpha = 1 — the
 * reference's m_hdrBuffer (tile-renderer.hpp:93, tonemapper == nullptr). */
int yart_hip_render(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                    float* out_rgba, YartStats* stats);
/* Same, one wave of the schedule at a time (tile-renderer.hpp:264-289): after every wave out_rgba holds the frame
 * blended so far and on_wave is called — what Renderer::onRenderWaveComplete reports (renderer.hpp:33-38, 56) —
 * with that wave's statistics; a non-zero return stops the render after this wave, as Renderer::abort() does
 * between tiles, and the call returns YART_ABORTED. on_wave may be NULL. */
typedef int (*YartWaveCallback)(void* user, const YartStats* wave_stats, uint32_t wave, uint32_t wave_samples,
                                uint32_t samples_taken, uint32_t total_samples);
int yart_hip_render_waves(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                          float* out_rgba, YartStats* stats, YartWaveCallback on_wave, void* user);
/* Same, with tile granularity (Renderer::onRenderTileComplete, renderer.hpp:40-50, 58; fired by finishTile,
 * tile-renderer.hpp:243-262 — the frontend uploads each finished tile from it, frontend main.cpp:206). A wave is
 * rendered batch by batch over this rank's pixel blocks in Morton order (YartRenderParams.max_batch_paths bounds a
 * batch; blocks are tile_size, or shard_tile, pixels wide); when a batch ends, the blocks it completed are copied into
 * out_rgba — which then holds, for those pixels, the frame blended up to this wave — and on_tile is called once per
 * block. A non-zero return stops the render after the current batch (YART_ABORTED; out_rgba then holds whatever had
 * been blended). YartTileInfo.rays is the reference's TileData.rays: the rays (path segments + unoccluded NEE rays,
 * mis-integrator.cpp:22, 126) of the block's pixels in this wave — every finished path carries its own count, the blend
 * kernel sums them per pixel, a small kernel per block; the blocks' counts of a wave sum to that wave's YartStats.rays. */
typedef struct YartTileInfo {
  uint32_t x, y, width, height;     /* TileData.offset / size */
  uint32_t index, total;            /* TileData.index (1-based count of finished blocks of this wave) / total */
  uint32_t wave, wave_samples, samples_taken, total_samples;   /* samples_taken: after this wave */
  uint64_t rays;                    /* TileData.rays: this block's rays of this wave */
  double ms;                        /* since the wave started */
} YartTileInfo;
typedef int (*YartTileCallback)(void* user, const YartTileInfo* tile);
int yart_hip_render_tiles(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                          float* out_rgba, YartStats* stats, YartWaveCallback on_wave, YartTileCallback on_tile, void* user);
/* Same as yart_hip_render, writing a DEVICE buffer (e.g. a torch tensor's data_ptr) on `stream`
 * (hipStream_t, may be NULL); returns after the work has been enqueued and
 * completed on that stream. */
int yart_hip_render_device(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                           float* d_out_rgba, void* stream, YartStats* stats);

/* Several GPUs of one node behind one handle — what the reference's worker pool is to CPU threads
 * (TileRenderer::renderImpl starts threadCount workers that pull tiles, tile-renderer.hpp:150-197; finishTile merges
 * each finished tile into the one m_hdrBuffer, :225-241). The scene is replicated on every listed device; device i of
 * N renders the pixel blocks b with b % N == i (blocks of shard_tile / tile_size pixels in Morton order — combined
 * with params->rank / world_size as device i of N of process rank r of R); one host thread per device drives its
 * launches; the merge sends each device's OWN pixels (a packed slab, 1/N of the frame) to devices[0] with RCCL
 * point-to-point calls over xGMI, where they are scattered into the frame, which is then copied to out_rgba.
 * RCCL failures return YART_E_RCCL. A device listed more than once (rehearsal on a one-GPU box) is served by a
 * device-to-device copy instead of RCCL, everything else being the same; bound max_batch_paths then, the replicas
 * share that device's memory. Blocking; the frame equals the single-device render bit for bit. */
typedef struct YartMulti YartMulti;
int yart_hip_multi_create(const YartSceneDesc* desc, const int* devices, uint32_t n_devices, YartMulti** out);
/* from a .yscn container or a .glb / .gltf asset (opts as for yart_hip_scene_load_gltf; ignored for .yscn) */
int yart_hip_multi_load(const char* path, const YartImportOptions* opts, const int* devices, uint32_t n_devices, YartMulti** out);
void yart_hip_multi_destroy(YartMulti* multi);
int yart_hip_multi_device_count(const YartMulti* multi);
/* Failure of a device (SURVEY §5 "failure detection"; the reference has none: a worker thread that dies takes the process along). A
 * HIP error on the host thread of a replica other than devices[0] takes THAT replica out of service for the rest of the handle's life:
 * the call still succeeds — the replica's pixel blocks are rendered on devices[0] after the merge (blocks are idempotent: which
 * device renders a pixel does not change it), in this and in every later render. This entry point reports the replicas out of service
 * (indices into the `devices` list, at most `capacity` written; returns their number) and leaves the first failure's message in
 * yart_hip_last_error(). A failure of devices[0] itself, where the frame is merged, is the call's error (YART_E_HIP). */
int yart_hip_multi_failed_devices(const YartMulti* multi, int* replicas_out, uint32_t capacity);
/* Diagnostic: the RCCL calls of the merge (ncclCommInitAll, one group of ncclSend + ncclRecv on a stream, ncclCommDestroy) on a
 * one-rank communicator of `device` — the rank sends a slab of n_floats to itself and compares. Shows on a one-GPU box that
 * RCCL is linked, initialises and moves a slab; YART_E_RCCL otherwise. */
int yart_hip_multi_rccl_selftest(int device, uint32_t n_floats);
/* stats: samples / rays / test counters summed over the devices, ms_* the slowest device's, ms_total the call's wall time */
int yart_hip_multi_render(YartMulti* multi, const YartCameraDesc* cam, const YartRenderParams* params, float* out_rgba,
                          YartStats* stats);
/* The progressive form of yart_hip_multi_render (what yart_hip_render_waves / _tiles are to yart_hip_render): one wave of the
 * schedule at a time on all devices, merged, copied to out_rgba and reported through on_wave (Renderer::onRenderWaveComplete
 * fires whatever the number of workers, tile-renderer.hpp:243-282); with on_tile every block of the frame is reported once per
 * wave after that wave's merge, in Morton order, with its own ray count. Either callback may be NULL; a non-zero return stops
 * the render after the current wave (YART_ABORTED). */
int yart_hip_multi_render_tiles(YartMulti* multi, const YartCameraDesc* cam, const YartRenderParams* params, float* out_rgba,
                                YartStats* stats, YartWaveCallback on_wave, YartTileCallback on_tile, void* user);

/* Diagnostics (device code paths, used by the parity tests):
 * per-sample radiance (before exposure) of n (x, y, sample) triples -> 3 floats each */
int yart_hip_probe_samples(YartScene* scene, const YartCameraDesc* cam, const YartRenderParams* params,
                           uint32_t n, const uint32_t* xys, float* out_rgb, uint64_t* out_rays);
/* Diagnostic: the ZSobol / FastOwen sampler alone on the device (reference core/sampler.hpp:84-173, scrambler.hpp:57-65): for each
 * of n cases (pixel x, pixel y, sample index: 3 words) startPixelSample, then n_draws draws following `pattern` (1 = get1D, 2 = get2D);
 * out receives sum(pattern) floats per case. use_tables != 0: through the per-render sampler tables the wavefront kernels read. */
int yart_hip_probe_sampler(YartScene* scene, uint32_t spp, uint32_t tile, uint32_t n, const uint32_t* cases, uint32_t n_draws,
                           const uint8_t* pattern, int use_tables, float* out);
/* closest hit of n world rays (ox,oy,oz,dx,dy,dz) -> 16 floats each:
 * hit, t, u, v, px,py,pz, nx,ny,nz, tx,ty,tz, triIdx, lightIdx, backSide */
int yart_hip_probe_hits(YartScene* scene, uint32_t n, const float* rays, float* out);
/* the BVH the kernels traverse: nodes (8 x u32 each: bounds, left|first, span) and
 * the index permutation of mesh `mesh` */
/* the 32 device counter words of the last render on this scene ([0] rays, [1..4] exact
 * traversal tallies in the instrumented build, [8..] kernel-phase statistics in debug builds) */
int yart_hip_debug_counters(YartScene* scene, uint64_t* out32);
/* measurement builds (-DYART_SHADE_REGIONS=1, tools/shade_regions.py) only: wave cycles [0..15], visits [16..31] and
 * active lanes [32..47] per code region of the shade kernel since the last call; YART_E_INVALID in the product build */
int yart_hip_debug_shade_regions(uint64_t* out48);
int yart_hip_bvh_info(YartScene* scene, uint32_t mesh, uint32_t* n_nodes, uint32_t* n_tris);
/* The reference's binned-SAH build (core/bvh.hpp:41-184, 273-347) of one mesh, outside a scene: on the device
 * (csrc/bvh_build_device.inc: level by level, one workgroup per node, the sequential partition in closed form) and on
 * the host (csrc/bvh_build_host = csrc/bvh_build.hpp, `threads` workers, 0 = all). Both write the reference's node array
 * (8 words per node: bounds, left|first, span; room for 2 * n_faces nodes) and index permutation, byte for byte the
 * same. faces: n_faces x face_stride words, the first three of each the vertex indices. The device build refuses input
 * with NaN coordinates or a tree deeper than 192 levels (YART_E_INVALID): build those on the host. */
int yart_hip_bvh_build_device(int device, const float* positions, uint32_t n_verts, const uint32_t* faces, uint32_t face_stride,
                              uint32_t n_faces, uint32_t* nodes_out, uint32_t* indices_out, uint32_t* n_nodes, double* ms_device);
int yart_hip_bvh_build_host(const float* positions, uint32_t n_verts, const uint32_t* faces, uint32_t face_stride, uint32_t n_faces,
                            uint32_t threads, uint32_t* nodes_out, uint32_t* indices_out, uint32_t* n_nodes, double* ms_host);
int yart_hip_bvh_copy(YartScene* scene, uint32_t mesh, uint32_t* nodes_out, uint32_t* indices_out);

/* The step after the path: AgX tonemap of the RGBA32F frame as
 * cpu/tile-renderer.hpp:234-237 applies it per tile (core/tonemapping.hpp:14-92; look 0 none, 1 golden,
 * 2 punchy; alpha = 1), and the 8-bit encoding of output/ppm.cpp:7-21 (gamma 1/2.2, * 255.999, truncated;
 * 3 bytes per pixel, row-major, no header). Device buffers; the calls return after completion on `stream`. */
int yart_hip_tonemap_agx(const float* d_hdr_rgba, uint32_t width, uint32_t height, int look, float* d_ldr_rgba,
                         void* stream);
int yart_hip_encode_rgb8(const float* d_rgba, uint32_t width, uint32_t height, uint8_t* d_rgb8, void* stream);
/* Host-buffer convenience: tonemap (look -1: none, as with a null tonemapper) + encode through the device;
 * ldr_rgba / rgb8 may be NULL. */
int yart_hip_tonemap_host(const float* hdr_rgba, uint32_t width, uint32_t height, int look, float* ldr_rgba,
                          uint8_t* rgb8);

const char* yart_hip_last_error(void);
int yart_hip_abi_version(void);
int yart_hip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* YART_HIP_H */
