"""ctypes front-end of oracle/wavefront_oracle.c -- TEST INFRASTRUCTURE ONLY (see the C file's header).

A scene is a plain dict (what forge3d_amd.wavefront.WavefrontScene.as_dict() returns):
    spheres          list of dict(center, radius, albedo, metallic, roughness, ior, emissive, ax, ay)
    meshes           list of (vertices (N,3) f32, indices (M,3) u32)          one BLAS each
    instances        list of dict(object_to_world (16,), world_to_object (16,), blas_index, material_id)
    dir_lights       list of dict(direction, intensity, color, importance)
    area_lights      list of dict(position, radius, normal, intensity, color, importance)
    object_importance list of float
    env_ground, env_sky, miss_ground, miss_sky     rgb(a)
    cam_origin, cam_right, cam_up, cam_forward, cam_fov_y (radians), cam_exposure, seed_hi, seed_lo
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SRC, _LIB = _HERE / "wavefront_oracle.c", _HERE / "libwavefront_oracle.so"
_TERRAIN_SRC = _HERE / "f3d_oracle.c"  # the terrain tracer's oracle: terrain_trace for the heightfield primitive


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("albedo", C.c_float * 3), ("metallic", C.c_float),
                ("roughness", C.c_float), ("ior", C.c_float), ("emissive", C.c_float * 3), ("ax", C.c_float), ("ay", C.c_float)]


class DirLight(C.Structure):
    _fields_ = [("direction", C.c_float * 3), ("intensity", C.c_float), ("color", C.c_float * 3), ("importance", C.c_float)]


class AreaLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("radius", C.c_float), ("normal", C.c_float * 3), ("intensity", C.c_float),
                ("color", C.c_float * 3), ("importance", C.c_float)]


class Instance(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16), ("blas_index", C.c_uint32),
                ("material_id", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32), ("indices", C.c_void_p), ("triangle_count", C.c_uint32)]


class Hair(C.Structure):
    _fields_ = [("p0", C.c_float * 3), ("r0", C.c_float), ("p1", C.c_float * 3), ("r1", C.c_float), ("material_id", C.c_uint32), ("pad", C.c_uint32 * 3)]


class Scene(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("sphere_count", C.c_uint32),
                ("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_uint32),
                ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint32),
                ("dir_lights", C.POINTER(DirLight)), ("dir_light_count", C.c_uint32),
                ("area_lights", C.POINTER(AreaLight)), ("area_light_count", C.c_uint32),
                ("object_importance", C.POINTER(C.c_float)), ("importance_count", C.c_uint32),
                ("env_ground", C.c_float * 4), ("env_sky", C.c_float * 4), ("miss_ground", C.c_float * 4), ("miss_sky", C.c_float * 4),
                ("cam_origin", C.c_float * 3), ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3), ("cam_forward", C.c_float * 3),
                ("cam_fov_y", C.c_float), ("cam_exposure", C.c_float), ("seed_hi", C.c_uint32), ("seed_lo", C.c_uint32),
                ("terrain", C.c_void_p), ("terrain_material", C.c_uint32), ("hair", C.POINTER(Hair)), ("hair_count", C.c_uint32),
                ("medium_g", C.c_float), ("medium_sigma_t", C.c_float), ("medium_density", C.c_float), ("medium_enabled", C.c_float)]


def build(force: bool = False) -> Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < max(_SRC.stat().st_mtime, _TERRAIN_SRC.stat().st_mtime):
        import os

        tmp = _LIB.with_suffix(f".{os.getpid()}.tmp")
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", str(_SRC), str(_TERRAIN_SRC), "-o", str(tmp), "-lm"],
                       check=True, capture_output=True)
        os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.wfo_render.restype = C.c_int
        _lib.f3do_terrain_open.restype = C.c_void_p
        _lib.f3do_terrain_open.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]
        _lib.f3do_terrain_close.argtypes = [C.c_void_p]
    return _lib


def _vec(ctype_array, values, pad=0.0):
    vals = [float(v) for v in values]
    n = len(ctype_array)
    vals = (vals + [pad] * n)[:n]
    for i, v in enumerate(vals):
        ctype_array[i] = v


def _fill(struct, d, vectors=()):
    for name, _t in struct._fields_:
        if name not in d:
            continue
        if name in vectors:
            _vec(getattr(struct, name), d[name])
        else:
            setattr(struct, name, d[name])
    return struct


def scene_struct(scene: dict):
    """-> (Scene, keepalive list)."""
    keep = []
    s = Scene()

    def array(cls, items, vectors):
        arr = (cls * max(1, len(items)))()
        for dst, src in zip(arr, items):
            _fill(dst, src, vectors)
        keep.append(arr)
        return arr

    spheres = array(Sphere, scene.get("spheres", []), ("center", "albedo", "emissive"))
    s.spheres, s.sphere_count = spheres, len(scene.get("spheres", []))
    meshes = (Mesh * max(1, len(scene.get("meshes", []))))()
    for dst, (v, i) in zip(meshes, scene.get("meshes", [])):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(i, np.uint32).reshape(-1, 3)
        keep += [v, i]
        dst.vertices, dst.vertex_count, dst.indices, dst.triangle_count = v.ctypes.data, v.shape[0], i.ctypes.data, i.shape[0]
    keep.append(meshes)
    s.meshes, s.mesh_count = meshes, len(scene.get("meshes", []))
    inst = array(Instance, scene.get("instances", []), ("object_to_world", "world_to_object"))
    s.instances, s.instance_count = inst, len(scene.get("instances", []))
    dl = array(DirLight, scene.get("dir_lights", []), ("direction", "color"))
    s.dir_lights, s.dir_light_count = dl, len(scene.get("dir_lights", []))
    al = array(AreaLight, scene.get("area_lights", []), ("position", "normal", "color"))
    s.area_lights, s.area_light_count = al, len(scene.get("area_lights", []))
    imp = scene.get("object_importance", [])
    imp_arr = (C.c_float * max(1, len(imp)))(*[float(x) for x in imp])
    keep.append(imp_arr)
    s.object_importance, s.importance_count = imp_arr, len(imp)
    for name in ("env_ground", "env_sky", "miss_ground", "miss_sky", "cam_origin", "cam_right", "cam_up", "cam_forward"):
        _vec(getattr(s, name), scene[name])
    s.cam_fov_y, s.cam_exposure = float(scene["cam_fov_y"]), float(scene.get("cam_exposure", 1.0))
    s.seed_hi, s.seed_lo = int(scene["seed_hi"]) & 0xFFFFFFFF, int(scene["seed_lo"]) & 0xFFFFFFFF
    if scene.get("terrain") is not None:
        t = scene["terrain"]
        dem = np.ascontiguousarray(t["heights"], np.float32)
        handle = lib().f3do_terrain_open(dem.ctypes.data, dem.shape[1], dem.shape[0], float(t["spacing"][0]), float(t["spacing"][1]),
                                         float(t["exaggeration"]))
        if not handle:
            raise ValueError("wavefront oracle: the terrain could not be opened")
        keep.append(_TerrainHandle(handle))
        s.terrain, s.terrain_material = handle, int(t["material_id"])
    hair = array(Hair, scene.get("hair") or [], ("p0", "p1"))
    s.hair, s.hair_count = hair, len(scene.get("hair") or [])
    if scene.get("medium") is not None:
        m = scene["medium"]
        s.medium_g, s.medium_sigma_t, s.medium_density, s.medium_enabled = float(m["g"]), float(m["sigma_t"]), float(m["density"]), float(m["enabled"])
    return s, keep


class _TerrainHandle:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            lib().f3do_terrain_close(self.handle)
        except Exception:  # noqa: BLE001
            pass


def render(scene: dict, width: int, height: int, frames: int, first_frame: int = 0, accum=None):
    """Adds `frames` frames to `accum` (zeros when None) and returns dict(accum, hdr, rgba)."""
    s, keep = scene_struct(scene)
    if accum is None:
        accum = np.zeros((height, width, 4), np.float32)
    accum = np.ascontiguousarray(accum, np.float32)
    rc = lib().wfo_render(C.byref(s), C.c_uint32(width), C.c_uint32(height), C.c_uint32(first_frame), C.c_uint32(frames),
                          accum.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"wavefront oracle: status {rc}")
    del keep
    total = first_frame + frames
    hdr, rgba = resolve(accum, total, float(scene.get("cam_exposure", 1.0)))
    return {"accum": accum, "hdr": hdr, "rgba": rgba, "frames": total}


def resolve(accum, frames: int, exposure: float = 1.0):
    accum = np.ascontiguousarray(accum, np.float32)
    h, w = accum.shape[:2]
    hdr = np.empty((h, w, 4), np.float32)
    rgba = np.empty((h, w, 4), np.uint8)
    lib().wfo_resolve(accum.ctypes.data_as(C.c_void_p), C.c_uint64(h * w), C.c_uint32(frames), C.c_float(exposure),
                      hdr.ctypes.data_as(C.c_void_p), rgba.ctypes.data_as(C.c_void_p))
    return hdr, rgba
