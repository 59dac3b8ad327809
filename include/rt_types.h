/* Wire-format records shared by the host application and the HIP render core.
 *
 * Every struct here is byte-identical to the reference's `repr(C)` Rust type and to the std430
 * GLSL struct the reference's compute shader reads, so a scene blob produced by the reference's
 * `Scene::write_buffers` (src/scene/mod.rs:166-195) can be handed to rt_upload_scene unchanged.
 *
 * Plain C: usable from C, C++, HIP and (via ctypes / bindgen) from Python and Rust.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 32-byte BVH node. Reference: src/accelerators/bvh_based/bvh.rs:19-98, src/shaders/main.comp:24-29.
 * amt != 0 : leaf over triangles [idx, idx+amt) of the (reordered) triangle buffer.
 * amt == 0 : interior, children are nodes idx and idx+1.  Root is node 0. */
typedef struct rt_bvh_node {
    float    min[3];
    uint32_t idx;
    float    max[3];
    uint32_t amt;
} rt_bvh_node;

/* 144-byte triangle. Reference: src/types/triangle.rs:7-16, src/shaders/imports/global.comp:39-45.
 * vec3 members occupy 16-byte slots (std430 array stride). */
typedef struct rt_triangle {
    uint32_t material_idx;
    uint32_t is_sphere;          /* ignored by the render path (spheres are dead code upstream) */
    uint32_t _pad0[2];
    float    v[3][4];            /* vertices, xyz + pad   @16,32,48  */
    float    n[3][4];            /* normals,  xyz + pad   @64,80,96  */
    float    uv[3][2];           /* texture coordinates   @112       */
    uint32_t _pad1[2];           /* -> 144                           */
} rt_triangle;

/* 96-byte Disney material. Reference: src/types/material.rs:6-30, global.comp:16-37. */
typedef struct rt_material {
    float    base_colour[4];     /* rgb, w unused */
    float    emission[4];        /* rgb, w unused */
    float    metallic;
    float    roughness;
    float    subsurface;
    float    anisotropic;
    float    specular_tint;
    float    sheen;
    float    sheen_tint;
    float    clearcoat;
    float    clearcoat_roughness;
    float    spec_trans;
    float    ior;
    uint32_t shade_smooth;
    uint32_t base_color_tex;          /* 0xFFFFFFFF = no texture */
    uint32_t metallic_roughness_tex;
    uint32_t normal_tex;
    uint32_t emission_tex;
} rt_material;

#define RT_NO_TEXTURE 0xFFFFFFFFu

/* 48-byte analytic light. Reference: src/types/light.rs:25-32, global.comp:65-70. type 0 = point. */
typedef struct rt_light {
    float    position[4];
    float    emission[4];
    uint32_t type;
    uint32_t _pad[3];
} rt_light;

/* 16-byte emissive-triangle record. Reference: src/types/light.rs:8-15, global.comp:72-78. */
typedef struct rt_emissive_tri {
    uint32_t tri_index;
    uint32_t mesh_index;         /* = material index upstream */
    float    area;
    float    power;
} rt_emissive_tri;

/* 128-byte camera block (112 used). Reference: src/types/camera_buffer_obj.rs:11-23, main.comp:41-50.
 * `rotation` is column-major and maps camera space to world space; the camera looks down -Z. */
typedef struct rt_camera {
    uint32_t resolution[2];
    float    view_port_uv[2];
    float    focal_length;
    float    focus_distance;
    float    aperture_radius;
    uint32_t frame;              /* "time" upstream: the running frame index, written per dispatch */
    float    location[4];
    float    rotation[16];
    uint32_t _pad[4];
} rt_camera;

/* One RGBA8 (sRGB-encoded) image. For the skybox, `pixels` holds 6 consecutive faces
 * (+X,-X,+Y,-Y,+Z,-Z) of width x height each. Reference: src/scene/mod.rs:41-65. */
typedef struct rt_texture {
    uint32_t       width;
    uint32_t       height;
    const uint8_t* pixels;
} rt_texture;

#ifdef __cplusplus
}
static_assert(sizeof(rt_bvh_node) == 32, "BvhNode wire size");
static_assert(sizeof(rt_triangle) == 144, "Triangle wire size");
static_assert(sizeof(rt_material) == 96, "Material wire size");
static_assert(sizeof(rt_light) == 48, "Light wire size");
static_assert(sizeof(rt_emissive_tri) == 16, "EmissiveTri wire size");
static_assert(sizeof(rt_camera) == 128, "Camera block wire size");
#endif

#endif /* RT_TYPES_H */
