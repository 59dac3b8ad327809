/* C ABI of the MI355X path-tracing core (librt_hip.so).
 *
 * This is the seam at which the reference application's Vulkan back end is replaced: each entry
 * point names the reference interface it stands in for (paths relative to the reference checkout).
 * The reference keeps its renderer behind one facade type, `VulkanRenderer` (src/vulkan/mod.rs:1-15,
 * src/vulkan/renderer.rs:62-507), fed by `Scene` (src/scene/mod.rs:67-205); a Rust `HipRenderer`
 * binding these symbols is a drop-in for that facade's headless subset.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative rt_status; nothing throws or aborts across the ABI;
 *   - rt_last_error(ctx) returns a NUL-terminated UTF-8 message owned by the context
 *     (rt_last_error(NULL) returns the message of the last failed context-less call on this thread);
 *   - the caller owns every host pointer; the library copies during the call;
 *   - a context is bound to one GPU and must be used from one thread at a time
 *     (the reference confines its renderer to one "render-thread", src/app/render_controller.rs:24-34);
 *   - plain pointers and sizes only: no C++, HIP or PyTorch types.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK               = 0,
    RT_ERR_INVALID      = -1,   /* bad argument (NULL, zero size, inconsistent counts)              */
    RT_ERR_DEVICE       = -2,   /* HIP runtime error; message carries hipGetErrorString             */
    RT_ERR_NO_SCENE     = -3,   /* render/trace called before rt_upload_scene                        */
    RT_ERR_UNSUPPORTED  = -4,   /* input honours the wire format but exceeds a documented limit      */
    RT_ERR_OOM          = -5
} rt_status;

/* Behaviour switches. The defaults (flags = 0) give the *intended* semantics of the reference
 * shader with an fp32 HDR accumulator; the RT_REF_* switches reproduce upstream defects bit-for-bit
 * where a caller wants the legacy picture. */
enum {
    RT_REF_RGBA8_ACCUM   = 1u << 0, /* running mean stored in 8-bit UNORM per frame (main.comp:64,262-266)        */
    RT_REF_CDF_PAD       = 1u << 1, /* light CDF length = 16-byte padded length (disney.comp:272, mod.rs:145-148) */
    RT_FIX_ABSORPTION    = 1u << 2, /* Beer-Lambert with the physical sign instead of main.comp:182-184           */
    RT_COUNT_TRAVERSAL   = 1u << 3, /* count node pops / triangle tests (mesh_tests, intersects.comp:136,140)     */
    RT_KERNEL_TIMING     = 1u << 4, /* bracket every kernel launch with events (gpu_timer.rs:18-78 analogue)      */
    RT_EXACT_ORDER       = 1u << 6, /* closest-hit search exactly as hit_meshes does it (intersects.comp:124-175): the
                                       binary tree, the shader's box arithmetic, visiting order and tie rules, so triangle
                                       ids equal the shader's bit for bit. Without it (the default) rays walk a compressed
                                       4-wide tree with conservative boxes, about 1.25x faster in that kernel: distances and ids still
                                       equal the shader's except where two triangles lie within rounding distance of
                                       each other along the ray (either may be reported), or where the shader's own box
                                       test loses a hit to rounding (the hit is then reported). Shadow rays follow the
                                       same rule: blocked whenever the shader says blocked, up to blockers within
                                       rounding distance of the light.                                            */
    RT_EXACT_ARITH       = 1u << 7, /* shade stage (main.comp:137-209, disney.comp) with correctly rounded division and
                                       square root and no fused multiply-add, so that everything outside sin/cos/log/exp/pow
                                       equals an operation-by-operation IEEE evaluation of the shader bit for bit. Without
                                       it (the default) the same formulas use the hardware reciprocal, reciprocal square
                                       root and square root (1 ulp each), FMA contraction within an expression and base-2 log/exp: about 1.3x
                                       faster in that kernel, every result within 1e-5 relative of the exact mode wherever
                                       the formula is well conditioned. Implied by RT_EXACT_ORDER. Ray generation,
                                       traversal, triangle tests, texture probes and accumulation are exact in both modes. */
    RT_SINGLE_STREAM     = 1u << 5  /* every kernel of every pass on one stream, one after another. By default the
                                       shadow rays of a bounce run on a side stream beside the next bounce's closest-hit
                                       search (while no other pass is in flight), and short passes (up to 24 Mi paths, 12
                                       frames of 1080p) alternate between three sets of path buffers so that consecutive
                                       passes overlap; RT_HIP_LANES=1..8 (environment, read at rt_create) fixes the number
                                       of sets for all passes.
                                       Accumulation order, and therefore every output bit, is the same either way.
                                       Implied by RT_KERNEL_TIMING so that per-kernel times are those of kernels
                                       running alone.                                                            */
};

typedef struct rt_config {
    int32_t  device;          /* HIP device ordinal                                                  */
    uint32_t width, height;   /* render resolution (VulkanRenderer::render_resolution)               */
    uint32_t max_bounces;     /* BOUNCES in main.comp:4 (24 upstream); 0 selects 24                  */
    uint32_t flags;           /* RT_* switches above                                                 */
    uint32_t frames_per_pass; /* samples traced concurrently by one wavefront pass; 0 = automatic.
                                 Forced to 1 when RT_REF_RGBA8_ACCUM is set. width * height *
                                 frames_per_pass may not exceed 2^28 - 2^14 path slots
                                 (RT_ERR_UNSUPPORTED otherwise).                                     */
} rt_config;

typedef struct rt_ctx rt_ctx;

/* Per-kernel device time of the last rt_render / rt_trace_* call, valid when RT_KERNEL_TIMING is set
 * (the context then runs its passes one after another on a single stream).
 * Stands in for VulkanRenderer::last_timer_perf (renderer.rs:239-248). */
typedef struct rt_timings {
    float    total_ms;        /* whole call, first launch to last                      */
    float    raygen_ms, extend_ms, shade_ms, connect_ms, accumulate_ms;
    uint32_t raygen_launches, extend_launches, shade_launches, connect_launches, accumulate_launches;
} rt_timings;

/* Work counters since the last rt_reset_counters. node_pops / tri_tests stay 0 without RT_COUNT_TRAVERSAL. */
typedef struct rt_counters {
    uint64_t camera_rays;       /* paths started                                         */
    uint64_t extend_rays;       /* closest-hit rays traced (camera + bounce rays)        */
    uint64_t shadow_rays;       /* NEE visibility rays traced                            */
    uint64_t extend_node_pops;  /* interior nodes popped by extend                       */
    uint64_t extend_tri_tests;  /* triangles tested by extend                            */
    uint64_t shadow_node_pops;
    uint64_t shadow_tri_tests;
    uint64_t extend_leaf_pops;
    uint64_t shadow_leaf_pops;
    uint64_t tail_paths;        /* paths whose last bounces ran in the per-path kernel (their rays are in the totals above) */
} rt_counters;

/* What rt_tune_resident_set measured and did. Index 0: the binary tree (walked under RT_EXACT_ORDER); 1: the compressed
 * 4-wide tree (closest hits by default); 2: the uncompressed 4-wide tree (shadow rays that may stop at any hit), which
 * shares its record order with 1. */
typedef struct rt_tune_report {
    uint64_t sampled_visits[3];         /* node visits counted (every 16th ray of one sample per pixel)                  */
    uint32_t resident_records[3];       /* records of that tree a traversal workgroup keeps in LDS                       */
    float    resident_share_before[3];  /* fraction of the counted visits whose record was among them: old order ...     */
    float    resident_share_after[3];   /* ... and new order                                                             */
    uint32_t records_moved[3];          /* records whose slot changed ([2] repeats [1])                                  */
} rt_tune_report;

/* 16-byte closest-hit record written by the extend stage. prim = 0xFFFFFFFF on a miss. */
typedef struct rt_hit {
    float    t, u, v;
    uint32_t prim;
} rt_hit;

/* One path on arrival at a vertex, as the extend stage leaves it: the ray of this bounce, its closest hit, and what the
 * path carries (main.comp:122-134,188-209). Input record of rt_shade_vertices. 64 bytes. */
typedef struct rt_shade_vertex {
    float    ray_o[3];
    float    prev_pdf;        /* pdf of the BSDF sample that produced the ray (bsdfPdf, main.comp:197); 0 at bounce 0 */
    float    ray_d[3];
    uint32_t seed;            /* RNG state (random.comp:6-12) before this vertex's draws                            */
    float    throughput[3];
    uint32_t reserved;
    rt_hit   hit;             /* prim = 0xFFFFFFFF: the ray left the scene                                          */
} rt_shade_vertex;

/* What the shade stage does with one vertex (main.comp:137-209). 144 bytes. */
typedef struct rt_shade_result {
    float    emitted[3];      /* radiance added at the vertex itself: sky on a miss, MIS-weighted emission on a hit  */
    uint32_t seed;            /* RNG state after this vertex's draws                                                */
    float    throughput[3];   /* after absorption, the BSDF sample and Russian roulette                            */
    float    pdf;             /* pdf of the BSDF sample (carried to the next vertex)                                */
    float    next_o[3];
    uint32_t alive;           /* 1: the path continues along next_o / next_d                                        */
    float    next_d[3];
    uint32_t has_shadow;      /* 1: a light sample was queued for the connect stage                                 */
    float    shadow_o[3];
    float    shadow_tmax;     /* distance to the sampled point minus 0.001 (disney.comp:699)                        */
    float    shadow_d[3];
    int32_t  light_index;     /* int(rand * numOfLights), disney.comp:673; -1 when the scene has no light           */
    float    shadow_contrib[3]; /* weight * f * Le / pdf * throughput: what connect adds if nothing opaque intervenes */
    int32_t  lobe;            /* lobe DisneySample chose: 0 diffuse, 1 reflection, 2 refraction, 3 clearcoat; -1 none */
    int32_t  light_tri;       /* entry of the emissive-triangle table SampleMeshCDF chose; -1 for point lights / none */
    float    light_pdf;       /* pdf of the light sample (selection x area measure converted to solid angle)        */
    float    bsdf_pdf_light;  /* DisneyEval's pdf for the light direction (the other side of the MIS weight)        */
    float    roughness;       /* roughness the BSDF used: after the metallic-roughness texture and the 0.001 clamp   */
    float    medium_throughput[3]; /* throughput after the absorption step (main.comp:182-184), before the BSDF sample */
    uint32_t reserved;
} rt_shade_result;

typedef struct rt_bvh_params {
    float    traversal_cost;  /* 1.0 */
    float    intersect_cost;  /* 1.0 */
    uint32_t min_leaf;        /* 2   */
    uint32_t max_leaf;        /* 8   */
    uint32_t threads;         /* 0 = all hardware threads */
} rt_bvh_params;

typedef struct rt_bvh_stats {
    float    sah_cost;
    uint32_t node_count, leaf_count, max_depth, max_leaf_size;
    float    avg_leaf_size;
    float    build_ms;
} rt_bvh_stats;

/* ------------------------------------------------------------------------------------------------
 * Host-side scene preparation (no GPU needed).
 * ---------------------------------------------------------------------------------------------- */

const char* rt_version(void);
const char* rt_last_error(const rt_ctx* ctx);

/* Binned-SAH object-split BVH over `n_tri` triangles, honouring the traversal contract of
 * src/accelerators/bvh_based/bin_sah.rs:39-44 (what Scene::build_bvh, src/scene/mod.rs:84-115, produces
 * with SplitPolicy::Object). `tris` is permuted in place into leaf order; `nodes` must have room for
 * max(1, 2*n_tri-1) records. NULL params selects BinSah::default() (bin_sah.rs:115-125). */
int rt_bvh_build(rt_triangle* tris, uint32_t n_tri, const rt_bvh_params* params,
                 rt_bvh_node* nodes, uint32_t* n_nodes, rt_bvh_stats* stats);

/* The same contract built with spatial splits (SBVH), the policy Scene::build_bvh actually selects
 * (BinSah::spatial(), src/accelerators/bvh_based/bin_sah.rs:94-100,127-134,862-1450): alpha gates the spatial
 * candidate on child overlap / root area (reference preset 1e-5), budget_fraction caps duplicated references as a
 * fraction of n_tri (preset 0.3). Triangles straddling a chosen plane are duplicated, so the leaf-ordered triangles
 * go to `tris_out` (capacity >= n_tri + floor(n_tri*budget_fraction), must not alias `tris`) and `*n_tris_out` may
 * exceed n_tri; `nodes` needs 2 * that capacity. The tree does not depend on the thread count. */
int rt_bvh_build_spatial(const rt_triangle* tris, uint32_t n_tri, const rt_bvh_params* params, float alpha,
                         float budget_fraction, rt_triangle* tris_out, uint32_t tri_capacity, uint32_t* n_tris_out,
                         rt_bvh_node* nodes, uint32_t node_capacity, uint32_t* n_nodes, rt_bvh_stats* stats);

/* The reference's first builder, `Bvh` (src/accelerators/bvh_based/bvh.rs:183-331; `impl Accelerator for Bvh`): eight
 * candidate planes per axis, centre-of-extent assignment, a node stays a leaf when its best split does not beat the cost
 * its parent was split at. Single-threaded like the original; leaves are not capped (the renderer accepts up to 64
 * triangles per leaf). Same contract and argument meaning as rt_bvh_build; benches/bvh.rs:20-27 times this builder. */
int rt_bvh_build_sweep(rt_triangle* tris, uint32_t n_tri, rt_bvh_node* nodes, uint32_t* n_nodes, rt_bvh_stats* stats);

/* Emissive-triangle table and un-normalised power CDF, as Scene::build_lights (src/scene/mod.rs:207-270).
 * `emissive` and `cdf` need room for n_tri entries. Call after rt_bvh_build (tri_index refers to leaf order). */
int rt_build_lights(const rt_triangle* tris, uint32_t n_tri, const rt_material* mats, uint32_t n_mat,
                    rt_emissive_tri* emissive, float* cdf, uint32_t* n_emissive, float* total_power);

/* Section sizes (each rounded up to 16 bytes) of the packed scene blob
 * [bvh | materials | triangles | lights | emissive | cdf] — Scene::get_buffer_sizes (src/scene/mod.rs:117-157). */
int rt_scene_buffer_sizes(uint32_t n_bvh, uint32_t n_mat, uint32_t n_tri, uint32_t n_light,
                          uint32_t n_emissive, uint64_t sizes[6]);

/* Packs the six sections into `blob` (capacity >= sum of sizes) — Scene::write_buffers (src/scene/mod.rs:166-195). */
int rt_scene_write_buffers(const rt_bvh_node* nodes, uint32_t n_bvh, const rt_material* mats, uint32_t n_mat,
                           const rt_triangle* tris, uint32_t n_tri, const rt_light* lights, uint32_t n_light,
                           const rt_emissive_tri* emissive, const float* cdf, uint32_t n_emissive,
                           void* blob, uint64_t blob_capacity);

/* Fills a camera block from a look-at pose exactly like the reference's YAML camera
 * (src/types/camera_buffer_obj.rs:26-33,150-170): rotation = transpose(look_at_rh(location, look_at, +Y)). */
int rt_camera_look_at(rt_camera* cam, uint32_t width, uint32_t height, const float location[3],
                      const float look_at[3], float focal_length, float focus_distance, float aperture_radius);

/* ------------------------------------------------------------------------------------------------
 * Renderer (one context per GPU).
 * ---------------------------------------------------------------------------------------------- */

/* VulkanRenderer::create (renderer.rs:62-120) + PathTracer::new (path_tracer/mod.rs:56-137), headless. */
int  rt_create(const rt_config* cfg, rt_ctx** out);
/* VulkanRenderer::destroy (renderer.rs:493-507). */
void rt_destroy(rt_ctx* ctx);

/* VulkanRenderer::upload_scene / reload_scene (renderer.rs:122-131,193-231) -> PathTracer::upload_scene
 * (path_tracer/mod.rs:275-288). `blob`/`sizes` are what write_buffers / get_buffer_sizes produce; the counts are
 * the true element counts (the shader derives them from padded byte ranges, which is how the upstream CDF-length
 * defect arises). Re-lays the data out for the GPU and resets accumulation. */
int rt_upload_scene(rt_ctx* ctx, const void* blob, const uint64_t sizes[6],
                    uint32_t n_bvh, uint32_t n_mat, uint32_t n_tri, uint32_t n_light, uint32_t n_emissive);

/* Texture array + cube map upload (path_tracer/mod.rs:99-137). All images are RGBA8 holding sRGB-encoded values,
 * sampled bilinearly with one mip level; 2-D textures REPEAT, the cube clamps to the face edge
 * (src/vulkan/core/image.rs:307-318, sampler.rs:10-34). skybox == NULL installs the 1x1 white cube, replacing any cube map uploaded earlier
 * (src/scene/mod.rs:49-65). */
int rt_upload_textures(rt_ctx* ctx, const rt_texture* textures, uint32_t n_textures, const rt_texture* skybox);

/* A cube map of linear radiance in fp32 ("HDR skybox"). No counterpart in the reference, whose cube map is RGBA8 holding
 * sRGB values like every other image (src/scene/mod.rs:36-65,272-299; src/vulkan/core/image.rs:307-318): an 8-bit sky cannot
 * hold a sun. rgba: 6 faces in the order +x -x +y -y +z -z, each height x width texels of {r, g, b, a} floats. Sampled exactly
 * like the 8-bit cube (main.comp:82-84: face by major axis, bilinear, clamped at the face edge) without the sRGB decode; a
 * sample's luminance is still clamped at 5 before accumulation (main.comp:114-120). Replaces the cube map in use until the
 * next rt_upload_textures call; resets the accumulation. */
int rt_upload_skybox_f32(rt_ctx* ctx, uint32_t width, uint32_t height, const float* rgba);

/* Orders the node records by measured use, for the scene and camera now installed. The traversal kernels keep the first
 * records of each tree in LDS; rt_upload_scene fills that prefix by surface area, which ranks nodes by how often a
 * random ray meets them. This call traces one sample per pixel with per-record visit counters (every 16th ray counts),
 * moves the most visited records to the front and uploads the re-ordered trees. Rendered values do not change (the trees
 * and the order their children are visited in are the same); accumulation is neither used nor restarted; the work
 * counters are left as they were. Waits for queued work first. A later rt_upload_scene goes back to the area order.
 * `report` may be NULL. No counterpart upstream: the shader reads every node from its storage buffer
 * (intersects.comp:65-66). */
int rt_tune_resident_set(rt_ctx* ctx, rt_tune_report* report);

/* update_uniform_buffer (renderer.rs:465-475) + set_camera_pose (renderer.rs:165-168): installs the camera block and
 * restarts accumulation. `resolution`, `view_port_uv` and `frame` in the block are overwritten by the context. */
int rt_set_camera(rt_ctx* ctx, const rt_camera* cam);
/* handle_resize (renderer.rs:133-162): new render resolution, restarts accumulation. */
int rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height);
/* reset_accumulator (renderer.rs:170-177): frame counter back to 0, accumulator cleared. */
int rt_reset_accumulation(rt_ctx* ctx);

/* Multi-GPU sample slicing: this context renders global frame indices first, first+stride, first+2*stride, ...
 * The per-pixel seed uses the global index (main.comp:251), so the union over all slices equals a 1-GPU run.
 * Restarts accumulation. Default (0,1). No counterpart upstream (single GPU). */
int rt_set_frame_slice(rt_ctx* ctx, uint32_t first, uint32_t stride);
/* Multi-GPU tile slicing, the other partition of the same job: the image is cut into 8x8-pixel tiles (the footprint of one
 * workgroup of the reference's dispatch, main.comp:13; path_tracer/mod.rs:332-337 launches ceil(w/8) x ceil(h/8) of them),
 * numbered row by row, and this context renders tiles first, first+stride, ... of every frame; the other pixels of its
 * accumulator stay zero (sum and sample count). Every pixel is then rendered by exactly one context with the seeds and in the
 * order of a 1-GPU run, so the sum over all slices EQUALS the 1-GPU accumulator bit for bit. first < stride; default (0,1) =
 * the whole image. Restarts accumulation. Independent of rt_set_frame_slice. No counterpart upstream (single GPU). */
int rt_set_tile_slice(rt_ctx* ctx, uint32_t first, uint32_t stride);

/* n_samples x dispatch_compute (renderer.rs:269-321): each sample is one full path per pixel with the frame index
 * of its dispatch. Asynchronous: returns once the work is queued on the context's stream. */
int rt_render(rt_ctx* ctx, uint32_t n_samples);
/* Blocks until all queued work of the context has finished. */
int rt_sync(rt_ctx* ctx);
/* frame_complete (renderer.rs:251-267) asks a fence without waiting, so that the render loop can keep several dispatches in
 * flight (render_controller.rs:233-244): *done = 1 when everything queued so far has finished, else 0. Never blocks and
 * does not keep the next rt_render from overlapping the work still running. */
int rt_poll(rt_ctx* ctx, int* done);
/* sample_count (renderer.rs:180-182): samples accumulated by this context since the last reset. */
uint32_t rt_sample_count(const rt_ctx* ctx);

/* HDR mean (sum / count) as width*height RGBA floats, alpha = 1. New capability (upstream accumulates in 8 bits). */
int rt_read_accum_f32(rt_ctx* ctx, float* rgba, size_t n_floats);
/* Raw fp32 sum buffer (RGBA, alpha = sample count) for external reduction / checkpointing. */
int rt_read_sum_f32(rt_ctx* ctx, float* rgba, size_t n_floats);
/* Adds `rgba_sum` (same layout as rt_read_sum_f32) into the context's sum buffer: the receive side of a host-staged
 * multi-GPU reduce, and resume-from-checkpoint. */
int rt_add_sum_f32(rt_ctx* ctx, const float* rgba_sum, size_t n_floats);
/* Device address and byte size of the fp32 RGBA sum buffer, for in-place RCCL reduction by the caller. */
int rt_accum_device_buffer(rt_ctx* ctx, void** device_ptr, size_t* n_bytes);
/* Tone-mapped 8-bit image, RGBA order: the contents SaveFrame writes (main.comp:215-244,267;
 * src/vulkan/utils/save_frame.rs:27-65). */
int rt_read_framebuffer_rgba8(rt_ctx* ctx, uint8_t* rgba, size_t n_bytes);

int rt_last_timings(const rt_ctx* ctx, rt_timings* out);
int rt_get_counters(rt_ctx* ctx, rt_counters* out);
int rt_reset_counters(rt_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Stage-level entry points: the same kernels rt_render launches, on caller-supplied inputs.
 * They exist so each stage can be checked in isolation against a CPU evaluation of the reference shader.
 * ---------------------------------------------------------------------------------------------- */

/* extend stage = hit_meshes(ray, false) (intersects.comp:124-175). origins/directions: n x 3 floats. */
int rt_trace_closest(rt_ctx* ctx, uint32_t n, const float* origins, const float* directions, rt_hit* hits);
/* connect stage = the shadow test of DirectLight (disney.comp:697-701): visible[i] = 0 when the closest hit along the
 * ray is nearer than tmax[i] and its material has spec_trans <= 0.5, else 1. */
int rt_trace_shadow(rt_ctx* ctx, uint32_t n, const float* origins, const float* directions, const float* tmax,
                    uint8_t* visible);
/* Device-resident variants for throughput measurement: rays already in HBM as float4 {o.xyz,_} / {d.xyz,_};
 * repeats the launch `repeat` times and reports the mean kernel time in milliseconds. */
int rt_bench_trace(rt_ctx* ctx, uint32_t n, const float* origins, const float* directions, int shadow,
                   uint32_t repeat, float* ms_per_launch);

/* rt_trace_closest for callers that may not allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc, processes in which another stream is being captured into a HIP graph. No counterpart in the reference. Same
 * inputs, same outputs (bit for bit), but the call allocates no device memory: rays, hits and the work cursor are staged in
 * one fixed device array that is part of the library image (about 10 MiB per GPU, held until the process ends), which
 * limits a call to RT_STATIC_MAX_RAYS rays; more is refused with RT_ERR_UNSUPPORTED. The call is synchronous (it copies
 * from and to the caller's host buffers on the context's own stream and waits for it), so it cannot itself be part of a
 * captured graph. All contexts on one GPU share the array: two such calls that overlap in time on the same GPU race, so
 * serialise them. Per-kernel timing (RT_KERNEL_TIMING) does not cover this call. */
#define RT_STATIC_MAX_RAYS (1u << 18)
int rt_trace_closest_static(rt_ctx* ctx, uint32_t n, const float* origins, const float* directions, rt_hit* hits);

/* raygen stage = seed rule + generate_ray (main.comp:251; generate_ray.comp:2-29) for every pixel of frame `frame`.
 * origins/directions: width*height x 3 floats; seeds: RNG state after the 8 camera draws. */
int rt_generate_rays(rt_ctx* ctx, uint32_t frame, float* origins, float* directions, uint32_t* seeds);

/* DisneyEval (disney.comp:560-619) on n independent inputs. V, N, L: n x 3 floats (world space, N = face-forward normal);
 * eta: n floats; out_f: n x 3; out_pdf: n. Material roughness is used as given (the caller applies the 0.001 clamp). */
int rt_eval_bsdf(rt_ctx* ctx, uint32_t n, const rt_material* mats, const float* eta, const float* V, const float* N,
                 const float* L, float* out_f, float* out_pdf);
/* DisneySample (disney.comp:476-558) with explicit random numbers r1r2 (n x 2). out_L: n x 3 world-space. */
int rt_sample_bsdf(rt_ctx* ctx, uint32_t n, const rt_material* mats, const float* eta, const float* V, const float* N,
                   const float* r1r2, float* out_L, float* out_f, float* out_pdf);
/* Bilinear sRGB fetch of texture `tex` (or the cube map when tex == 0xFFFFFFFF, with uvw = direction). uvw: n x 3. */
int rt_sample_texture(rt_ctx* ctx, uint32_t tex, uint32_t n, const float* uvw, float* rgba);

/* shade stage = everything ray_color does at one path vertex between the two traversals (main.comp:137-209;
 * DirectLight disney.comp:623-720 up to, not including, its shadow test; DisneySample disney.comp:476-558): the kernel
 * rt_render launches, run on n caller-supplied vertices as vertex number `bounce` of their paths. n may not exceed
 * width * height * frames_per_pass of the context, and bounce must be below max_bounces. At the last bounce
 * (bounce + 1 == max_bounces) the stage draws no BSDF sample: nothing it could produce is ever looked at. */
int rt_shade_vertices(rt_ctx* ctx, uint32_t n, uint32_t bounce, const rt_shade_vertex* vertices, rt_shade_result* results);

/* Per-pixel radiance of ONE sample (frame index `frame`), before firefly clamp and accumulation:
 * ray_color (main.comp:122-213). Does not touch the accumulator. rgb: width*height x 3 floats. */
int rt_render_radiance(rt_ctx* ctx, uint32_t frame, float* rgb);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
