"""Scene ingestion for the harness: the reference's live scene format (YAML + Wavefront OBJ), read into the wire-layout
`Scene` the renderer consumes, plus the frame saver.

Mirrors `Scene::from_new` / `load_scene_yaml` (src/scene/loader.rs:11-232):

    skybox:    [px, nx, py, ny, pz, nz]          six image paths relative to the YAML file (loader.rs:36-46)
    materials: {name: {field: value, ...}}       fields of `Material` (src/types/material.rs:9-29), anything missing keeps
                                                 Material::default(); unknown keys are ignored; colours are integer
                                                 triples divided by 255 (src/types/aligned.rs:44-56)
    surfaces:  - {file: mesh.obj, material: name}                     OBJ, triangulated, one index per corner
               - {type: mesh, material: name, vertices: [[x,y,z]..], triangles: [[i,j,k]..]}   inline, flat normals
    camera:    {resolution: [w,h], location: [..], look_at: [..], focal_length, focus_distance, aperture_radius}
                                                 (src/types/camera_buffer_obj.rs:59-170)

The OBJ reader implements what the reference asks of `tobj` (`single_index: true, triangulate: true`, loader.rs:75-82):
`v`/`vn`/`vt`/`f`, 1-based and negative indices, polygons split into a fan around their first corner, one model per
`o`/`g` statement; a model whose faces carry no normals (or texture coordinates) gets zeros there, as loader.rs:109-139.
Parity for this file is pinned by the format description above and by hand-written fixtures only: the reference ships no
loader tests and its parser dependencies cannot be run here.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import yaml

from . import api
from .scene import Scene, default_material, look_at_camera

_F32_FIELDS = ("metallic", "roughness", "subsurface", "anisotropic", "specular_tint", "sheen", "sheen_tint", "clearcoat",
               "clearcoat_roughness", "spec_trans", "ior")
_U32_FIELDS = ("shade_smooth", "base_color_tex", "metallic_roughness_tex", "normal_tex", "emission_tex")
_VEC_FIELDS = ("base_colour", "emission")


class SceneFormatError(ValueError):
    """The file breaks the format in a way the reference answers with a panic."""


def material_from_mapping(name: str, data) -> np.ndarray:
    """One `Material` from its YAML mapping (serde `#[serde(default)]` semantics)."""
    m = default_material(1)
    if data is None:
        return m
    if not isinstance(data, dict):
        raise SceneFormatError(f"Failed to deserialize material '{name}': expected a mapping")
    for key, val in data.items():
        if key in _VEC_FIELDS:
            # AVec3 deserialises a list of unsigned integers and divides by 255
            if (not isinstance(val, (list, tuple)) or len(val) < 3
                    or any(isinstance(v, bool) or not isinstance(v, int) or v < 0 for v in val[:3])):
                raise SceneFormatError(f"Failed to deserialize material '{name}': {key} must be three non-negative integers")
            m[key][0, :3] = np.array([np.float32(v) / np.float32(255.0) for v in val[:3]], dtype=np.float32)
        elif key in _F32_FIELDS:
            if isinstance(val, bool) or not isinstance(val, (int, float)):
                raise SceneFormatError(f"Failed to deserialize material '{name}': {key} must be a number")
            m[key] = np.float32(val)
        elif key in _U32_FIELDS:
            if isinstance(val, bool) or not isinstance(val, int) or not 0 <= val <= 0xFFFFFFFF:
                raise SceneFormatError(f"Failed to deserialize material '{name}': {key} must be a u32")
            m[key] = val
        # anything else (relative_ior, thin, ...) is not a field of Material and is skipped
    return m


def load_obj(path) -> list[dict]:
    """Models of an OBJ file as dicts {name, v (n,3,3), n (n,3,3) or None, uv (n,3,2) or None}, faces in file order."""
    pos, nrm, tex = [], [], []
    models, cur = [], None

    def start(name):
        nonlocal cur
        cur = {"name": name, "corners": []}
        models.append(cur)

    def resolve(idx: str, count: int) -> int:
        i = int(idx)
        if i == 0:
            raise SceneFormatError(f"{path}: index 0 in a face")
        i = i - 1 if i > 0 else count + i
        if not 0 <= i < count:
            raise SceneFormatError(f"{path}: face index {idx} out of range")
        return i

    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].split()
            if not line:
                continue
            tag, args = line[0], line[1:]
            if tag == "v":
                pos.append([float(a) for a in args[:3]])
            elif tag == "vn":
                nrm.append([float(a) for a in args[:3]])
            elif tag == "vt":
                uv = [float(a) for a in args[:2]]
                tex.append(uv + [0.0] * (2 - len(uv)))
            elif tag in ("o", "g"):
                start(" ".join(args) if args else "unnamed_object")
            elif tag == "f":
                if cur is None:
                    start("unnamed_object")
                corners = []
                for a in args:
                    parts = a.split("/")
                    vi = resolve(parts[0], len(pos))
                    ti = resolve(parts[1], len(tex)) if len(parts) > 1 and parts[1] else -1
                    ni = resolve(parts[2], len(nrm)) if len(parts) > 2 and parts[2] else -1
                    corners.append((vi, ti, ni))
                if len(corners) < 3:
                    continue
                for k in range(1, len(corners) - 1):           # fan around the first corner
                    cur["corners"].append((corners[0], corners[k], corners[k + 1]))
    P = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    N = np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
    T = np.asarray(tex, dtype=np.float32).reshape(-1, 2)
    out = []
    for m in models:
        if not m["corners"]:
            continue
        c = np.asarray(m["corners"], dtype=np.int64)            # (faces, 3 corners, 3 indices)
        v = P[c[:, :, 0]]
        has_n = bool((c[:, :, 2] >= 0).any())
        has_t = bool((c[:, :, 1] >= 0).any())
        n = np.where((c[:, :, 2] >= 0)[..., None], N[np.maximum(c[:, :, 2], 0)], 0).astype(np.float32) if has_n and len(N) else None
        uv = np.where((c[:, :, 1] >= 0)[..., None], T[np.maximum(c[:, :, 1], 0)], 0).astype(np.float32) if has_t and len(T) else None
        out.append({"name": m["name"], "v": v, "n": n, "uv": uv})
    return out


def _triangles(v, n, uv, material_index) -> np.ndarray:
    t = np.zeros(len(v), dtype=api.TRIANGLE)
    t["v"][:, :, :3] = v
    if n is not None:
        t["n"][:, :, :3] = n
    if uv is not None:
        t["uv"] = uv
    t["material_idx"] = material_index
    return t


def _flat_normals(v: np.ndarray) -> np.ndarray:
    """(v1 - v0) x (v2 - v0), normalised as glam does (v * (1 / sqrt(v.v))), all in f32 (loader.rs:199)."""
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    c = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                  e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(np.float32)
    len2 = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]).astype(np.float32) + c[:, 2] * c[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1.0) / np.sqrt(len2, dtype=np.float32)).astype(np.float32)
    return (c * inv[:, None]).astype(np.float32)


def load_image_rgba8(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGBA"), dtype=np.uint8).copy()


def load_cubemap(paths) -> np.ndarray:
    """Six faces, +X -X +Y -Y +Z -Z, as one (6, h, w, 4) array (src/scene/mod.rs:272-299)."""
    faces = [load_image_rgba8(p) for p in paths]
    if any(f.shape != faces[0].shape for f in faces):
        raise SceneFormatError("All cubemap faces must have the same dimensions")
    return np.stack(faces)


def load_scene_yaml(path, log=None) -> Scene:
    """The components of a scene file, before the BVH and light table are built (loader.rs:23-232)."""
    path = Path(path)
    base = path.parent
    with open(path, "r") as fh:
        doc = yaml.safe_load(fh)
    if not isinstance(doc, dict):
        raise SceneFormatError("Failed to parse YAML: top level is not a mapping")

    skybox = None
    if isinstance(doc.get("skybox"), list):
        if len(doc["skybox"]) != 6:
            raise SceneFormatError("skybox needs exactly six faces")
        skybox = load_cubemap([base / str(p) for p in doc["skybox"]])

    mats, index_of = [], {}
    if isinstance(doc.get("materials"), dict):
        for name, data in doc["materials"].items():
            index_of[str(name)] = len(mats)
            mats.append(material_from_mapping(str(name), data))

    tris = []
    if isinstance(doc.get("surfaces"), list):
        for surf in doc["surfaces"]:
            if not isinstance(surf, dict):
                continue
            if isinstance(surf.get("file"), str):
                mname = surf.get("material")
                if not isinstance(mname, str):
                    raise SceneFormatError("Material not specified")
                if mname not in index_of:
                    raise SceneFormatError(f"surface references unknown material '{mname}'")
                obj_path = base / surf["file"]
                try:
                    models = load_obj(obj_path)
                except (OSError, ValueError) as e:      # the reference logs and carries on (loader.rs:157)
                    if log:
                        log(f"Failed to load OBJ file {surf['file']}: {e}")
                    continue
                for m in models:
                    tris.append(_triangles(m["v"], m["n"], m["uv"], index_of[mname]))
            elif surf.get("type") == "mesh":
                mname = surf.get("material")
                if not isinstance(mname, str):
                    raise SceneFormatError("Material not specified")
                if mname not in index_of:
                    raise SceneFormatError(f"surface references unknown material '{mname}'")
                if not isinstance(surf.get("vertices"), list):
                    raise SceneFormatError("Vertices not found or invalid")
                if not isinstance(surf.get("triangles"), list):
                    raise SceneFormatError("Triangles not found or invalid")
                verts = np.asarray(surf["vertices"], dtype=np.float64).astype(np.float32).reshape(-1, 3)
                idx = np.asarray(surf["triangles"], dtype=np.int64).reshape(-1, 3)
                if len(idx) and (idx.min() < 0 or idx.max() >= len(verts)):
                    raise SceneFormatError("triangle index out of range")
                v = verts[idx]
                fn = _flat_normals(v)
                tris.append(_triangles(v, np.repeat(fn[:, None, :], 3, axis=1), None, index_of[mname]))

    camera = np.zeros(1, dtype=api.CAMERA)
    if isinstance(doc.get("camera"), dict):
        cam = doc["camera"]
        for key in ("resolution", "location", "look_at"):
            if key not in cam:
                raise SceneFormatError(f"Failed to deserialize camera: missing field `{key}`")
        unknown = set(cam) - {"resolution", "focal_length", "focus_distance", "aperture_radius", "location", "look_at"}
        if unknown:
            raise SceneFormatError(f"Failed to deserialize camera: unknown field `{sorted(unknown)[0]}`")
        w, h = (int(x) for x in cam["resolution"][:2])
        camera = look_at_camera(w, h, cam["location"], cam["look_at"], float(cam.get("focal_length", 1.0)),
                                float(cam.get("focus_distance", 1.0)), float(cam.get("aperture_radius", 0.0)))

    triangles = np.concatenate(tris) if tris else np.zeros(0, dtype=api.TRIANGLE)
    materials = np.concatenate(mats) if mats else np.zeros(0, dtype=api.MATERIAL)
    return Scene(triangles=triangles, materials=materials, camera=camera, skybox=skybox, name=path.stem)


def from_new(path, spatial: bool = True, threads: int = 0, log=print) -> Scene:
    """`Scene::from_new` (loader.rs:11-21): load, build the BVH with the reference's policy (spatial splits), build lights."""
    sc = load_scene_yaml(path, log=log)
    sc.build_bvh(threads=threads, spatial=spatial)
    sc.build_lights()
    if log:
        log(f"Triangles: {len(sc.triangles)}")
    return sc


def save_frame(renderer, path) -> Path:
    """Writes the tone-mapped 8-bit frame as an RGBA image, the file `SaveImage::save_frame` produces
    (src/vulkan/utils/save_frame.rs:27-65; the swapchain's BGRA order is already undone on the device)."""
    from PIL import Image
    path = Path(path)
    if not path.is_absolute():
        path = Path(os.getcwd()) / path
    Image.fromarray(renderer.read_framebuffer(), "RGBA").save(path)
    return path
