"""Scene ingestion for the harness: the reference's live scene format (YAML + Wavefront OBJ), read into the wire-layout
`Scene` the renderer consumes, plus the frame saver.

Mirrors `Scene::from_new` / `load_scene_yaml` (src/scene/loader.rs:11-232):

    skybox:    [px, nx, py, ny, pz, nz]          six image paths relative to the YAML file (loader.rs:36-46)
    skybox_hdr: sky.hdr                          (extension, not in the reference) a Radiance RGBE panorama, resampled to an
                                                 fp32 cube map of skybox_hdr_size (512) texels per edge; replaces `skybox`
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


def load_hdr(path) -> np.ndarray:
    """A Radiance RGBE picture (.hdr, '-Y h +X w' orientation, flat or new-style run-length encoded scanlines) as an
    (h, w, 3) float32 array of linear radiance. Extension of the build: the reference reads 8-bit images only."""
    with open(path, "rb") as fh:
        data = fh.read()
    if not (data.startswith(b"#?RADIANCE") or data.startswith(b"#?RGBE")):
        raise SceneFormatError(f"{path}: not a Radiance HDR file")
    pos = data.find(b"\n\n")
    if pos < 0:
        raise SceneFormatError(f"{path}: header not terminated")
    if b"FORMAT=32-bit_rle_rgbe" not in data[:pos]:
        raise SceneFormatError(f"{path}: only FORMAT=32-bit_rle_rgbe is supported")
    end = data.find(b"\n", pos + 2)
    dims = data[pos + 2:end].split()
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise SceneFormatError(f"{path}: unsupported orientation {data[pos + 2:end]!r}")
    h, w = int(dims[1]), int(dims[3])
    buf = np.frombuffer(data, dtype=np.uint8, offset=end + 1)
    rgbe = np.empty((h, w, 4), dtype=np.uint8)
    p = 0
    for y in range(h):
        if 8 <= w < 32768 and p + 4 <= len(buf) and buf[p] == 2 and buf[p + 1] == 2 and (int(buf[p + 2]) << 8 | int(buf[p + 3])) == w:
            p += 4
            for ch in range(4):                      # each channel of the scanline run-length encoded on its own
                x = 0
                while x < w:
                    n = int(buf[p]); p += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, ch] = buf[p]; p += 1
                    else:
                        rgbe[y, x:x + n, ch] = buf[p:p + n]; p += n
                    x += n
                if x != w:
                    raise SceneFormatError(f"{path}: scanline {y} overruns")
        else:
            rgbe[y] = buf[p:p + 4 * w].reshape(w, 4); p += 4 * w
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def save_hdr(path, img) -> None:
    """Writes an (h, w, 3) float array as a flat (not run-length encoded) Radiance RGBE picture."""
    img = np.asarray(img, dtype=np.float32)
    m = img.max(axis=-1)
    mant, ex = np.frexp(m)
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-38))
    rgbe = np.zeros(img.shape[:2] + (4,), dtype=np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m < 1e-32, 0, ex + 128).astype(np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (img.shape[0], img.shape[1]))
        fh.write(rgbe.tobytes())


def equirect_to_cube(img, size: int) -> np.ndarray:
    """A latitude-longitude panorama (h, w, 3), +Y up, as the (6, size, size, 4) float32 cube map rt_upload_skybox_f32 takes:
    every cube texel takes the bilinearly filtered panorama value in the direction of its centre, with the face and
    (s, t) conventions of the cube fetch (main.comp:82-84; +x -x +y -y +z -z)."""
    img = np.asarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    c = (np.arange(size, dtype=np.float64) + 0.5) / size * 2.0 - 1.0
    sc, tc = np.meshgrid(c, c)                                   # sc along x (columns), tc along y (rows)
    one = np.ones_like(sc)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    out = np.ones((6, size, size, 4), dtype=np.float32)
    for f, (dx, dy, dz) in enumerate(dirs):
        n = np.sqrt(dx * dx + dy * dy + dz * dz)
        lon = np.arctan2(dx, -dz)                                # -z is the centre of the panorama
        lat = np.arcsin(dy / n)
        u = (lon / (2 * np.pi) + 0.5) * w - 0.5
        v = (0.5 - lat / np.pi) * h - 0.5
        u0, v0 = np.floor(u).astype(int), np.floor(v).astype(int)
        fu, fv = (u - u0)[..., None], (v - v0)[..., None]
        x0, x1 = u0 % w, (u0 + 1) % w
        y0, y1 = np.clip(v0, 0, h - 1), np.clip(v0 + 1, 0, h - 1)
        top = img[y0, x0, :3] * (1 - fu) + img[y0, x1, :3] * fu
        bot = img[y1, x0, :3] * (1 - fu) + img[y1, x1, :3] * fu
        out[f, ..., :3] = top * (1 - fv) + bot * fv
    return out


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
    skybox_hdr = None
    if isinstance(doc.get("skybox_hdr"), str):            # the build's extension: one Radiance .hdr panorama
        skybox_hdr = equirect_to_cube(load_hdr(base / doc["skybox_hdr"]), int(doc.get("skybox_hdr_size", 512)))

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
    return Scene(triangles=triangles, materials=materials, camera=camera, skybox=skybox, skybox_hdr=skybox_hdr, name=path.stem)


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


# ---------------------------------------------------------------------------------------------------------
# glTF 2.0 (.gltf + external/embedded buffers, or .glb), as `Scene::from_gltf` reads it (src/scene/gltf.rs:130-433)
# ---------------------------------------------------------------------------------------------------------

_GLTF_DTYPES = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


def gltf_default_camera() -> np.ndarray:
    """`camera_sponza()`, the camera from_gltf installs before looking for one in the file (gltf.rs:59-80,139)."""
    return look_at_camera(1920, 1080, (3.0, 2.0, 0.0), (-4.0, 2.0, 0.0), 1.0, 55.0, 0.0)


class _Gltf:
    def __init__(self, path: Path):
        import base64
        import json
        import struct
        self.base = path.parent
        raw = path.read_bytes()
        self.glb_bin = None
        if raw[:4] == b"glTF":
            _, _, total = struct.unpack_from("<III", raw, 0)
            off, doc = 12, None
            while off + 8 <= min(total, len(raw)):
                n, kind = struct.unpack_from("<II", raw, off)
                chunk = raw[off + 8: off + 8 + n]
                if kind == 0x4E4F534A:
                    doc = json.loads(chunk.decode("utf-8"))
                elif kind == 0x004E4942 and self.glb_bin is None:
                    self.glb_bin = chunk
                off += 8 + n + (-n % 4)
            if doc is None:
                raise SceneFormatError("GLB without a JSON chunk")
            self.doc = doc
        else:
            self.doc = json.loads(raw.decode("utf-8"))
        self._b64 = base64
        self._buffers = {}

    def _uri_bytes(self, uri: str) -> bytes:
        from urllib.parse import unquote
        if uri.startswith("data:"):
            return self._b64.b64decode(uri.split(",", 1)[1])
        return (self.base / unquote(uri)).read_bytes()

    def buffer(self, index: int) -> bytes:
        if index not in self._buffers:
            b = self.doc["buffers"][index]
            if "uri" in b:
                self._buffers[index] = self._uri_bytes(b["uri"])
            elif self.glb_bin is not None:
                self._buffers[index] = self.glb_bin
            else:
                raise SceneFormatError("buffer without uri outside a GLB container")
        return self._buffers[index]

    def view_bytes(self, index: int) -> bytes:
        v = self.doc["bufferViews"][index]
        off = v.get("byteOffset", 0)
        return self.buffer(v["buffer"])[off: off + v["byteLength"]]

    def accessor(self, index: int) -> np.ndarray:
        a = self.doc["accessors"][index]
        if "sparse" in a:
            raise SceneFormatError("sparse accessors are not supported")
        dt = np.dtype(_GLTF_DTYPES[a["componentType"]]).newbyteorder("<")
        width, count = _GLTF_WIDTH[a["type"]], a["count"]
        if "bufferView" not in a:
            return np.zeros((count, width), dtype=dt)
        v = self.doc["bufferViews"][a["bufferView"]]
        start = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        elem = dt.itemsize * width
        stride = v.get("byteStride", 0) or elem
        buf = self.buffer(v["buffer"])
        if start + stride * (count - 1) + elem > len(buf) and count:
            raise SceneFormatError("accessor reaches beyond its buffer")
        out = np.ndarray((count, width), dtype=dt, buffer=buf, offset=start, strides=(stride, dt.itemsize))
        return np.array(out)

    def accessor_f32(self, index: int) -> np.ndarray:
        """Values as f32; normalised integer encodings are expanded the way glTF prescribes (`into_f32`)."""
        a = self.doc["accessors"][index]
        raw = self.accessor(index)
        ct = a["componentType"]
        if ct == 5126:
            return raw.astype(np.float32)
        scale = {5121: 255.0, 5123: 65535.0, 5120: 127.0, 5122: 32767.0}.get(ct)
        if scale is None:
            return raw.astype(np.float32)
        out = raw.astype(np.float32) / np.float32(scale)
        return np.maximum(out, np.float32(-1.0)) if ct in (5120, 5122) else out

    def image(self, index: int) -> np.ndarray:
        import io
        from PIL import Image
        im = self.doc["images"][index]
        data = self._uri_bytes(im["uri"]) if "uri" in im else self.view_bytes(im["bufferView"])
        with Image.open(io.BytesIO(data)) as pil:
            return np.asarray(pil.convert("RGBA"), dtype=np.uint8).copy()

    def node_matrix(self, node) -> np.ndarray:
        """Column-major 4x4 as a (4,4) array M with M @ [x,y,z,1] the transformed point (`node.transform().matrix()`)."""
        if "matrix" in node:
            return np.asarray(node["matrix"], dtype=np.float32).reshape(4, 4).T.copy()
        t = np.asarray(node.get("translation", [0, 0, 0]), dtype=np.float32)
        x, y, z, w = (np.float32(v) for v in node.get("rotation", [0, 0, 0, 1]))
        s = np.asarray(node.get("scale", [1, 1, 1]), dtype=np.float32)
        one, two = np.float32(1), np.float32(2)
        r = np.array([[one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
                      [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
                      [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], dtype=np.float32)
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = r * s[None, :]
        m[:3, 3] = t
        return m


def _gltf_texture_source(g: _Gltf, info) -> int:
    """Texture reference -> *image* index, the id the reference stores (gltf.rs:299-317)."""
    if not info:
        return api.NO_TEXTURE
    tex = g.doc["textures"][info["index"]]
    return int(tex["source"]) if "source" in tex else api.NO_TEXTURE


def gltf_material(g: _Gltf, index) -> np.ndarray:
    """`process_material` (gltf.rs:292-413). `index` None selects glTF's default material."""
    md = g.doc["materials"][index] if index is not None else {}
    pbr = md.get("pbrMetallicRoughness", {})
    ext = md.get("extensions", {})
    m = default_material(1)
    m["base_colour"][0, :3] = np.asarray(pbr.get("baseColorFactor", [1, 1, 1, 1]), dtype=np.float32)[:3]
    m["metallic"] = np.float32(pbr.get("metallicFactor", 1.0))
    m["roughness"] = np.float32(pbr.get("roughnessFactor", 1.0))
    m["ior"] = np.float32(ext.get("KHR_materials_ior", {}).get("ior", 1.5)) if "KHR_materials_ior" in ext else np.float32(1.5)
    m["spec_trans"] = np.float32(ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0))
    emissive = np.asarray(md.get("emissiveFactor", [0, 0, 0]), dtype=np.float32)
    strength = np.float32(ext["KHR_materials_emissive_strength"].get("emissiveStrength", 1.0)) if "KHR_materials_emissive_strength" in ext else np.float32(0.0)
    m["emission"][0, :3] = emissive * strength if strength != 0 else emissive
    # the reference looks for the clearcoat block among the extensions of pbrMetallicRoughness (gltf.rs:350-363)
    cc = pbr.get("extensions", {}).get("KHR_materials_clearcoat")
    cc = cc if isinstance(cc, dict) else {}
    m["clearcoat"] = np.float32(cc.get("clearcoatFactor", 0.0))
    m["clearcoat_roughness"] = np.float32(cc.get("clearcoatRoughnessFactor", 0.0))
    m["shade_smooth"] = 0
    m["base_color_tex"] = _gltf_texture_source(g, pbr.get("baseColorTexture"))
    m["metallic_roughness_tex"] = _gltf_texture_source(g, pbr.get("metallicRoughnessTexture"))
    m["normal_tex"] = _gltf_texture_source(g, md.get("normalTexture"))
    m["emission_tex"] = _gltf_texture_source(g, md.get("emissiveTexture"))
    return m


def load_gltf(path) -> Scene:
    """Components of a glTF file: every image as an RGBA8 texture, one material per primitive, triangles of the default
    scene's node hierarchy in world space, camera from a top-level camera node if there is one (gltf.rs:131-290,415-432).
    Like the reference, only indexed primitives produce triangles and normals are transformed by the node matrix itself."""
    path = Path(path)
    g = _Gltf(path)
    doc = g.doc
    textures = [g.image(i) for i in range(len(doc.get("images", [])))]
    scenes = doc.get("scenes", [])
    if not scenes:
        raise SceneFormatError("No scenes found in GLTF file")
    root = scenes[doc["scene"]] if "scene" in doc else scenes[0]
    camera = gltf_default_camera()

    for ni in root.get("nodes", []):
        node = doc["nodes"][ni]
        if "camera" in node:
            cam = doc["cameras"][node["camera"]]
            if cam.get("type") != "perspective":
                raise NotImplementedError("ORTHOGRAPHIC CAMERA NOT IMPLEMENTED")
            m = g.node_matrix(node)
            camera["focal_length"] = np.float32(1.0) / np.tan(np.float32(cam["perspective"]["yfov"]) * np.float32(0.5), dtype=np.float32)
            camera["location"][0, :3] = m[:3, 3]
            camera["rotation"][0] = m.T.reshape(-1)      # stored column-major, not transposed: as upstream (gltf.rs:425)

    mats, tris = [], []

    def visit(ni, parent):
        node = doc["nodes"][ni]
        m = (parent @ g.node_matrix(node)).astype(np.float32)
        if "mesh" in node:
            for prim in doc["meshes"][node["mesh"]].get("primitives", []):
                mat_index = len(mats)
                mats.append(gltf_material(g, prim.get("material")))
                attrs = prim.get("attributes", {})
                if "POSITION" not in attrs:
                    continue
                pos = g.accessor_f32(attrs["POSITION"])
                nrm = g.accessor_f32(attrs["NORMAL"]) if "NORMAL" in attrs else np.tile(np.float32([0, 0, 1]), (len(pos), 1))
                uv = g.accessor_f32(attrs["TEXCOORD_0"]) if "TEXCOORD_0" in attrs else np.zeros((len(pos), 2), np.float32)
                if "indices" not in prim:
                    continue
                idx = g.accessor(prim["indices"]).astype(np.int64).reshape(-1)
                idx = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                if len(idx) and idx.max() >= len(pos):
                    raise SceneFormatError("primitive index out of range")
                a, t = m[:3, :3], m[:3, 3]
                # transform_point3: x_axis*p.x + y_axis*p.y + z_axis*p.z + w_axis, in f32
                wp = (pos[:, 0:1] * a[:, 0][None] + pos[:, 1:2] * a[:, 1][None] + pos[:, 2:3] * a[:, 2][None] + t[None]).astype(np.float32)
                wn = (nrm[:, 0:1] * a[:, 0][None] + nrm[:, 1:2] * a[:, 1][None] + nrm[:, 2:3] * a[:, 2][None]).astype(np.float32)
                len2 = (wn[:, 0] * wn[:, 0] + wn[:, 1] * wn[:, 1]).astype(np.float32) + wn[:, 2] * wn[:, 2]
                with np.errstate(divide="ignore", invalid="ignore"):
                    wn = (wn * (np.float32(1.0) / np.sqrt(len2, dtype=np.float32))[:, None]).astype(np.float32)
                tris.append(_triangles(wp[idx], wn[idx], uv[idx, :2], mat_index))
        for ci in node.get("children", []):
            visit(ci, m)

    for ni in root.get("nodes", []):
        visit(ni, np.eye(4, dtype=np.float32))

    triangles = np.concatenate(tris) if tris else np.zeros(0, dtype=api.TRIANGLE)
    materials = np.concatenate(mats) if mats else np.zeros(0, dtype=api.MATERIAL)
    return Scene(triangles=triangles, materials=materials, camera=camera, textures=textures, name=path.stem)


def from_gltf(path, spatial: bool = True, threads: int = 0, build_lights: bool = False, log=print) -> Scene:
    """`Scene::from_gltf` (gltf.rs:131-156). Upstream builds the BVH only, so emissive meshes are never sampled directly;
    `build_lights=True` adds the light table the YAML path builds."""
    sc = load_gltf(path)
    if log:
        log(f"Triangles: {len(sc.triangles)}")
    sc.build_bvh(threads=threads, spatial=spatial)
    if build_lights:
        sc.build_lights()
    return sc
