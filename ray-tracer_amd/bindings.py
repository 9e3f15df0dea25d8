"""Thin ctypes layer over include/rt_native.h and host/rth_api.h (no computation happens in Python)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# numpy views of the buffer ABI (include/rt_types.h)
DTYPES = {
    "BV": np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("indexA", "<i4"), ("indexB", "<i4")]),
    "Triangle": np.dtype([("v", "<i4", 3), ("localNormal", "<f4", 3), ("area", "<f4")]),
    "Vertex": np.dtype([("pos", "<f4", 3), ("uv", "<f4", 2)]),
    "SceneObject": np.dtype([("worldToLocal", "<f4", 16), ("localToWorld", "<f4", 16), ("materialIndex", "<i4"),
                             ("bvStartIndex", "<i4"), ("maxDepthBVH", "<i4"), ("areaApprox", "<f4"),
                             ("min", "<f4", 3), ("max", "<f4", 3)]),
    "LightObject": np.dtype([("localToWorld", "<f4", 16), ("areaApprox", "<f4"), ("brightness", "<f4"),
                             ("triStart", "<i4"), ("totTris", "<i4")]),
    "Material": np.dtype([("brightness", "<f4"), ("col", "<f4", 3), ("colTexLoc", "<i4", 2), ("colTexDims", "<i4", 2),
                          ("bump", "<f4"), ("bumpTexLoc", "<i4", 2), ("bumpTexDims", "<i4", 2),
                          ("roughness", "<f4"), ("roughnessTexLoc", "<i4", 2), ("roughnessTexDims", "<i4", 2),
                          ("metallicity", "<f4"), ("metallicityTexLoc", "<i4", 2), ("metallicityTexDims", "<i4", 2),
                          ("normalsTexLoc", "<i4", 2), ("normalsTexDims", "<i4", 2)]),
    "HitInfo": np.dtype([("dst", "<f4"), ("hitPoint", "<f4", 3), ("uv", "<f4", 2), ("normal", "<f4", 3),
                         ("incomingDir", "<f4", 3), ("materialIndex", "<i4")]),
    "Reservoir": np.dtype([("dir", "<f4", 3), ("hitPoint", "<f4", 3), ("normal", "<f4", 3), ("brdf", "<f4"),
                           ("chosenWeight", "<f4"), ("totWeights", "<f4"), ("totCandidates", "<f4")]),
    "TraceRecord": np.dtype([("triIndex", "<i4"), ("objIndex", "<i4"), ("visitHash", "<u4"), ("lightObj", "<i4"),
                             ("lightTri", "<i4"), ("shadowVisitHash", "<u4"), ("shadowTri", "<i4"),
                             ("bounceVisitHash", "<u4")]),
    "Int2": np.dtype([("x", "<i4"), ("y", "<i4")]),
    # RIS project (BVH_RIS_DL)
    "RisTri": np.dtype([("vA", "<f4", 3), ("vB", "<f4", 3), ("vC", "<f4", 3), ("uvA", "<f4", 2), ("uvB", "<f4", 2),
                        ("uvC", "<f4", 2), ("worldNormal", "<f4", 3), ("parentIndex", "<i4")]),
    "RisMaterial": np.dtype([("color", "<f4", 3), ("specularColor", "<f4", 3), ("brightness", "<f4"), ("smoothness", "<f4")]),
    "RisHitInfo": np.dtype([("dst", "<f4"), ("hitPoint", "<f4", 3), ("uv", "<f4", 2), ("normal", "<f4", 3),
                            ("materialIndex", "<i4")]),
    "Float4": np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")]),
}
assert [DTYPES[k].itemsize for k in ("RisTri", "RisMaterial", "RisHitInfo", "Float4")] == [76, 32, 40, 16]
assert [DTYPES[k].itemsize for k in ("BV", "Triangle", "Vertex", "SceneObject", "LightObject", "Material", "HitInfo",
                                     "Reservoir", "TraceRecord")] == [32, 28, 20, 168, 80, 108, 52, 52, 32]

RT_FORMAT_R32F, RT_FORMAT_RGBA32F, RT_FORMAT_RGBA8_UNORM = 1, 2, 3


class NativeError(RuntimeError):
    pass


_native = None
_host = None


def native_lib():
    """librtnative.so.  Raises if it has not been built: there is no fallback implementation."""
    global _native
    if _native is None:
        path = os.environ.get("RT_NATIVE_LIB") or os.path.join(_HERE, "librtnative.so")  # override: kernel experiments
        if not os.path.exists(path):
            raise NativeError(f"{path} is missing: build it with `make native` (no CPU fallback exists)")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, ci, cs, cf = C.c_void_p, C.c_int, C.c_char_p, C.c_float
        sig = {
            "rt_create": (ci, [ci, C.POINTER(vp)]), "rt_destroy": (None, [vp]), "rt_sync": (ci, [vp]),
            "rt_last_error": (cs, [vp]), "rt_stream": (vp, [vp]), "rt_transfer_stream": (vp, [vp]),
            "rt_flush": (ci, [vp]), "rt_set_fusion": (ci, [vp, ci]), "rt_set_overlap": (ci, [vp, ci]), "rt_set_tuning": (ci, [vp, cs, ci]),
            "rt_stream_order": (ci, [vp, ci]), "rt_mark": (ci, [vp, ci, ci]), "rt_wait_mark": (ci, [vp, ci, ci]),
            "rt_debug_note_rows": (ci, [vp, ci, vp, vp, ci, ci, ci]),
            "rt_debug_trace": (ci, [vp, ci]), "rt_debug_trace_dump": (C.c_int64, [vp, C.c_char_p, C.c_int64]), "rt_mark_elapsed": (ci, [vp, ci, ci, C.POINTER(C.c_double)]), "rt_set_row_band": (ci, [vp, ci, ci]),
            "rt_shader_open": (ci, [vp, cs, C.POINTER(vp)]), "rt_shader_close": (None, [vp]),
            "rt_find_kernel": (ci, [vp, cs]), "rt_dispatch": (ci, [vp, ci, ci, ci, ci]),
            "rt_dispatch_no_alloc": (ci, [vp, ci, ci, ci, ci]),
            "rt_enable_keyword": (ci, [vp, cs]), "rt_disable_keyword": (ci, [vp, cs]),
            "rt_set_int": (ci, [vp, cs, ci]), "rt_set_float": (ci, [vp, cs, cf]), "rt_set_bool": (ci, [vp, cs, ci]),
            "rt_set_ints": (ci, [vp, cs, C.POINTER(ci), ci]), "rt_set_vector": (ci, [vp, cs, C.POINTER(cf)]),
            "rt_set_matrix": (ci, [vp, cs, C.POINTER(cf)]),
            "rt_set_buffer": (ci, [vp, ci, cs, vp]), "rt_set_texture": (ci, [vp, ci, cs, vp]),
            "rt_buffer_create": (ci, [vp, C.c_int64, ci, C.POINTER(vp)]),
            "rt_buffer_set_data": (ci, [vp, vp, C.c_size_t]), "rt_buffer_get_data": (ci, [vp, vp, C.c_size_t]),
            "rt_buffer_get_range": (ci, [vp, vp, C.c_int64, C.c_int64]), "rt_texture_get_rows": (ci, [vp, vp, ci, ci]),
            "rt_buffer_release": (None, [vp]), "rt_buffer_count": (C.c_int64, [vp]), "rt_buffer_stride": (ci, [vp]),
            "rt_texture_create": (ci, [vp, ci, ci, ci, C.POINTER(vp)]),
            "rt_texture_set_data": (ci, [vp, vp, C.c_size_t]), "rt_texture_get_data": (ci, [vp, vp, C.c_size_t]),
            "rt_texture_release": (None, [vp]),
            "rt_buffer_map_rows": (ci, [vp, ci, ci, ci, vp, C.POINTER(ci)]), "rt_texture_map_rows": (ci, [vp, ci, ci, vp]),
            "rt_buffer_rows_written": (ci, [vp, ci, ci, ci]),
            "rt_enable_stats": (ci, [vp, ci]), "rt_get_stats": (ci, [vp, ci, vp]),
            "rt_profile_enable": (ci, [vp, ci]), "rt_profile_select": (ci, [vp, cs]), "rt_profile_reset": (ci, [vp]), "rt_profile_get": (ci, [vp, cs, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _native = lib
    return _native


class MainConfig(C.Structure):
    """rth_main_config (host/rth_api.h): the inspector fields of Main (Main.cs:8-53) that reach the compute path."""
    _fields_ = [("Resolution", C.c_int32 * 2), ("fieldOfView", C.c_float), ("MaxBounceCount", C.c_int32),
                ("ScatterProbability", C.c_float), ("DefocusStrength", C.c_float), ("FocalPlaneFactor", C.c_float),
                ("DebugMaxTriChecks", C.c_int32), ("DebugMaxBVChecks", C.c_int32),
                ("SceneObjectReservoirTestsNum", C.c_int32), ("TriReservoirTestsNum", C.c_int32),
                ("CandidateReservoirTestsNum", C.c_int32), ("SpatialReuseIterations", C.c_int32),
                ("TemporalCandidatesNum", C.c_int32), ("SpatialReuseBlur", C.c_int32),
                ("TemporalPrecisionThreshold", C.c_float), ("TemporalReuseWeight", C.c_float),
                ("PixelMovementThreshold", C.c_float), ("SpatialHitPointDiffThreshold", C.c_float),
                ("SpatialNormalsAngleThreshold", C.c_float), ("SpatialBRDFThreshold", C.c_float),
                ("VisibilityReuseThreshold", C.c_float), ("DoVisibilityReuse", C.c_int32),
                ("DoWeightRecalc", C.c_int32), ("DoBRDF", C.c_int32), ("cameraPosition", C.c_float * 3),
                ("cameraRotation", C.c_float * 4), ("frameRandBase", C.c_uint32), ("frameRandStep", C.c_uint32),
                ("Unorm8RenderTargets", C.c_int32)]


class RisConfig(C.Structure):
    """rth_ris_config (host/rth_api.h): the inspector fields of the RIS project's Main that reach the compute path."""
    _fields_ = [("Resolution", C.c_int32 * 2), ("fieldOfView", C.c_float), ("MaxBounceCount", C.c_int32),
                ("RaysPerPixel", C.c_int32), ("ScatterProbability", C.c_float), ("DefocusStrength", C.c_float),
                ("FocalPlaneFactor", C.c_float), ("DebugMaxTriChecks", C.c_int32), ("DebugMaxBVChecks", C.c_int32),
                ("SceneObjectCandidatesNum", C.c_int32), ("TriCandidatesNum", C.c_int32),
                ("SceneObjectReservoirTestsNum", C.c_int32), ("TriReservoirTestsNum", C.c_int32),
                ("CandidateReservoirTestsNum", C.c_int32), ("cameraPosition", C.c_float * 3),
                ("cameraEulerDeg", C.c_float * 3), ("smoothness", C.c_float), ("frameRandBase", C.c_uint32),
                ("frameRandStep", C.c_uint32), ("Unorm8RenderTargets", C.c_int32)]


class _RisSceneData(C.Structure):
    _fields_ = [("Tris", C.c_void_p), ("nTris", C.c_int64), ("Materials", C.c_void_p), ("nMaterials", C.c_int64)]


class _MaterialInput(C.Structure):
    _fields_ = [("brightness", C.c_float), ("col", C.c_float * 3), ("roughness", C.c_float),
                ("metallicity", C.c_float), ("bump", C.c_float)]


class _SceneData(C.Structure):
    _fields_ = [("BVs", C.c_void_p), ("nBVs", C.c_int64), ("Vertices", C.c_void_p), ("nVertices", C.c_int64),
                ("Triangles", C.c_void_p), ("nTriangles", C.c_int64), ("SceneObjects", C.c_void_p),
                ("nSceneObjects", C.c_int64), ("LightObjects", C.c_void_p), ("nLightObjects", C.c_int64),
                ("Materials", C.c_void_p), ("nMaterials", C.c_int64), ("ComponentDatas", C.c_void_p),
                ("nComponentDatas", C.c_int64), ("maxBVHDepth", C.c_int32), ("emittingObjectsNum", C.c_int32),
                ("sceneBVHStartIndex", C.c_int32), ("totArea", C.c_float), ("atlas", C.c_void_p),
                ("atlasWidth", C.c_int32), ("atlasHeight", C.c_int32)]


class _TextureInput(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixels", C.c_void_p)]


TEXTURE_SLOTS = ("col", "bump", "roughness", "metallicity", "normals")   # RTH_TEX_* order


class DispatchStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "pops", "bvChecks", "objVisits", "triIters", "triChecks",
                                          "validNeighbours", "activePixels")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Xfer(C.Structure):
    _fields_ = [("send", C.c_int32), ("peer", C.c_int32), ("ptr", C.c_void_p), ("bytes", C.c_size_t)]


class Segment(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_size_t)]


class KernelProfile(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("milliseconds", C.c_double), ("stats", DispatchStats)]


KERNEL_NAMES = ("primary", "candidates", "temporal", "spatial", "transfer", "trace_rays", "accumulate", "calc_tri_normals",
                "ris_primary", "ris_candidates", "ris_trace_rays", "ris_calc_tri_normals")


def host_lib():
    """librthost.so (scene builder + frame driver).  Loading it also loads librtnative.so."""
    global _host
    if _host is None:
        native_lib()
        path = os.path.join(_HERE, "librthost.so")
        if not os.path.exists(path):
            raise NativeError(f"{path} is missing: build it with `make host`")
        lib = C.CDLL(path)
        vp, ci, cs = C.c_void_p, C.c_int, C.c_char_p
        fp = C.POINTER(C.c_float)
        sig = {
            "rth_mesh_create": (vp, [vp, ci, vp, ci]), "rth_mesh_load": (vp, [cs]), "rth_mesh_subdivide9": (vp, [vp]),
            "rth_mesh_jitter": (vp, [vp, C.c_uint32, C.c_float]), "rth_mesh_disc": (vp, [ci]), "rth_mesh_ring_disc": (vp, [ci]),
            "rth_mesh_free": (None, [vp]), "rth_mesh_counts": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
            "rth_mesh_vertices": (vp, [vp]), "rth_mesh_triangles": (vp, [vp]),
            "rth_scene_create": (vp, [ci, ci]), "rth_scene_free": (None, [vp]),
            "rth_scene_add_material": (ci, [vp, C.POINTER(_MaterialInput)]), "rth_scene_add_mesh": (ci, [vp, vp]),
            "rth_scene_add_material_textured": (ci, [vp, C.POINTER(_MaterialInput), C.POINTER(_TextureInput)]),
            "rth_scene_add_object": (ci, [vp, ci, fp, fp, fp, ci, ci]), "rth_scene_build": (ci, [vp, ci]),
            "rth_scene_set_object_transform": (ci, [vp, ci, fp, fp, fp]), "rth_scene_rebuild": (ci, [vp, ci]),
            "rth_main_reload": (ci, [vp, C.POINTER(MainConfig)]), "rth_ris_main_reload": (ci, [vp]),
            "rth_scene_get": (ci, [vp, C.POINTER(_SceneData)]), "rth_scene_save_json": (ci, [vp, cs]),
            "rth_scene_load_json": (ci, [vp, cs, cs, C.c_char_p, ci]), "rth_world_to_local": (None, [fp, fp, fp, fp, fp]),
            "rth_view_space_dims": (None, [C.c_float, ci, ci, fp]), "rth_camera_matrices": (None, [fp, fp, fp, fp]),
            "rth_main_default_config": (None, [C.POINTER(MainConfig)]),
            "rth_main_create": (vp, [vp, C.POINTER(MainConfig), vp, ci, ci]), "rth_main_destroy": (None, [vp]),
            "rth_main_last_error": (cs, [vp]), "rth_main_render_frame": (ci, [vp]),
            "rth_main_update_per_frame": (ci, [vp]), "rth_main_run_restir_shader": (ci, [vp]),
            "rth_main_run_post_processing": (ci, [vp]), "rth_main_run_stage": (ci, [vp, ci]),
            "rth_main_run_spatial_pass": (ci, [vp, ci]), "rth_main_run_transfer": (ci, [vp]),
            "rth_main_frame_count": (ci, [vp]), "rth_main_set_camera": (ci, [vp, fp, fp]), "rth_main_buffer": (vp, [vp, cs]), "rth_main_texture": (vp, [vp, cs]),
            "rth_main_shader": (vp, [vp, cs]),
            "rth_mg_band_bounds": (ci, [ci, ci, C.POINTER(ci)]),
            "rth_mg_balanced_bounds": (ci, [ci, ci, C.POINTER(C.c_double), ci, C.POINTER(ci)]),
            "rth_mg_set_bounds": (ci, [vp, C.POINTER(ci)]), "rth_mg_row_costs": (ci, [vp, C.POINTER(C.c_double)]),
            "rth_mg_balance": (ci, [vp]), "rth_mg_enable_timing": (ci, [vp, ci]),
            "rth_mg_times": (ci, [vp, C.POINTER(C.c_double)]), "rth_mg_halo_plan": (ci, [ci, ci, C.POINTER(ci), C.POINTER(ci)]),
            "rth_hub_create": (vp, [ci]), "rth_hub_destroy": (None, [vp]), "rth_mg_use_hub": (ci, [vp, vp]),
            "rth_mg_unique_id": (ci, [C.c_char_p]), "rth_mg_create": (vp, [vp, ci, ci, C.POINTER(ci), C.c_char_p, C.c_char_p, ci]),
            "rth_mg_destroy": (None, [vp]), "rth_mg_last_error": (cs, [vp]), "rth_mg_bounds": (ci, [vp, C.POINTER(ci)]),
            "rth_mg_render_frames": (ci, [vp, ci]), "rth_mg_loopback": (ci, [vp, ci, ci, ci]),
            "rth_mg_phase": (ci, [vp, ci]), "rth_mg_plan": (ci, [vp, ci, vp, ci]),
            "rth_mg_create_ris": (vp, [vp, ci, ci, C.POINTER(ci), C.c_char_p, C.c_char_p, ci]),
            "rth_ris_main_ctx": (vp, [vp]), "rth_ris_main_resolution": (ci, [vp, C.POINTER(ci)]),
            "rth_ris_default_config": (None, [C.POINTER(RisConfig)]), "rth_ris_camera_transform": (None, [fp, fp]),
            "rth_ris_main_create": (vp, [vp, C.POINTER(RisConfig), vp, ci, ci]), "rth_ris_main_destroy": (None, [vp]),
            "rth_ris_main_last_error": (cs, [vp]), "rth_ris_main_render_frame": (ci, [vp]),
            "rth_ris_main_update_per_frame": (ci, [vp]), "rth_ris_main_run_stage": (ci, [vp, ci]),
            "rth_ris_main_run_post_processing": (ci, [vp]), "rth_ris_main_frame_count": (ci, [vp]),
            "rth_ris_main_scene": (ci, [vp, C.POINTER(_RisSceneData)]), "rth_ris_main_buffer": (vp, [vp, cs]),
            "rth_ris_main_texture": (vp, [vp, cs]), "rth_ris_main_shader": (vp, [vp, cs]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _host = lib
    return _host


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def world_to_local(position, euler_deg, scale):
    w2l, l2w = (C.c_float * 16)(), (C.c_float * 16)()
    host_lib().rth_world_to_local(_f3(position), _f3(euler_deg), _f3(scale), w2l, l2w)
    return np.array(w2l, dtype=np.float32), np.array(l2w, dtype=np.float32)


def default_config():
    cfg = MainConfig()
    host_lib().rth_main_default_config(C.byref(cfg))
    return cfg


def ris_default_config():
    cfg = RisConfig()
    host_lib().rth_ris_default_config(C.byref(cfg))
    return cfg


class Mesh:
    def __init__(self, handle):
        if not handle:
            raise NativeError("mesh creation failed")
        self.h = handle

    @classmethod
    def load(cls, path):
        return cls(host_lib().rth_mesh_load(path.encode()))

    @classmethod
    def disc(cls, ntris, rings=False):
        """Unit disc facing +z: a fan of ntris triangles, or (rings=True) sectors x rings of similar-sized ones."""
        return cls(host_lib().rth_mesh_ring_disc(ntris) if rings else host_lib().rth_mesh_disc(ntris))

    @classmethod
    def from_arrays(cls, verts, tris):
        verts = np.ascontiguousarray(verts, dtype=DTYPES["Vertex"])
        tris = np.ascontiguousarray(tris, dtype=np.int32)
        return cls(host_lib().rth_mesh_create(verts.ctypes.data, len(verts), tris.ctypes.data, len(tris)))

    def subdivide9(self):
        return Mesh(host_lib().rth_mesh_subdivide9(self.h))

    def jitter(self, seed, amplitude):
        return Mesh(host_lib().rth_mesh_jitter(self.h, seed, amplitude))

    def counts(self):
        nv, nt = C.c_int(), C.c_int()
        host_lib().rth_mesh_counts(self.h, C.byref(nv), C.byref(nt))
        return nv.value, nt.value

    def arrays(self):
        nv, nt = self.counts()
        v = np.ctypeslib.as_array(C.cast(host_lib().rth_mesh_vertices(self.h), C.POINTER(C.c_uint8)), (nv * 20,))
        t = np.ctypeslib.as_array(C.cast(host_lib().rth_mesh_triangles(self.h), C.POINTER(C.c_int32)), (nt, 3))
        return v.view(DTYPES["Vertex"]).copy(), t.copy()

    def __del__(self):
        if getattr(self, "h", None) and _host is not None:
            _host.rth_mesh_free(self.h)
            self.h = None


class Scene:
    """rth_scene: the registered objects of ObjectHelper plus ConstructScene()."""

    def __init__(self, split_resolution=5, max_depth_scene_bvh=10):
        self.h = host_lib().rth_scene_create(split_resolution, max_depth_scene_bvh)
        if not self.h:
            raise NativeError("scene creation failed")
        self._data = None

    def add_material(self, brightness, col, roughness=0.0, metallicity=0.0, bump=0.0, textures=None):
        """textures: optional {"col" | "bump" | "roughness" | "metallicity" | "normals": (H, W, 4) uint8 array}."""
        m = _MaterialInput(brightness, (C.c_float * 3)(*col), roughness, metallicity, bump)
        if not textures:
            return host_lib().rth_scene_add_material(self.h, C.byref(m))
        tex = (_TextureInput * 5)()
        keep = []
        for k, slot in enumerate(TEXTURE_SLOTS):
            if slot in textures:
                a = np.ascontiguousarray(textures[slot], dtype=np.uint8)
                assert a.ndim == 3 and a.shape[2] == 4, "textures are (H, W, 4) RGBA8 arrays"
                keep.append(a)
                tex[k] = _TextureInput(a.shape[1], a.shape[0], a.ctypes.data)
        r = host_lib().rth_scene_add_material_textured(self.h, C.byref(m), tex)
        if r < 0:
            raise NativeError("rth_scene_add_material_textured failed")
        return r

    def add_mesh(self, mesh):
        return host_lib().rth_scene_add_mesh(self.h, mesh.h)

    def add_object(self, mesh_id, position, euler_deg, scale, material_index, max_depth_bvh):
        r = host_lib().rth_scene_add_object(self.h, mesh_id, _f3(position), _f3(euler_deg), _f3(scale), material_index,
                                            max_depth_bvh)
        if r < 0:
            raise NativeError("rth_scene_add_object failed")
        return r

    def build(self, threads=0):
        if host_lib().rth_scene_build(self.h, threads) != 0:
            raise NativeError("rth_scene_build failed")
        d = _SceneData()
        if host_lib().rth_scene_get(self.h, C.byref(d)) != 0:
            raise NativeError("rth_scene_get failed")
        self._data = d
        return self

    def set_object_transform(self, object_index, position, euler_deg, scale):
        """A GameObject moved (object_index as returned by add_object); takes effect with the next rebuild()."""
        if host_lib().rth_scene_set_object_transform(self.h, object_index, _f3(position), _f3(euler_deg), _f3(scale)) != 0:
            raise NativeError("rth_scene_set_object_transform failed")

    def rebuild(self, keep_old_scene_bvh=True):
        """ConstructScene again on the built scene (a reload): mesh BVHs are kept, the per-object data, the scene BVH
        and the light list are made anew.  keep_old_scene_bvh: the ReSTIR project's behaviour (the old scene BVH stays
        in BVs, unreferenced); False: the RIS project's."""
        if host_lib().rth_scene_rebuild(self.h, 1 if keep_old_scene_bvh else 0) != 0:
            raise NativeError("rth_scene_rebuild failed")
        d = _SceneData()
        if host_lib().rth_scene_get(self.h, C.byref(d)) != 0:
            raise NativeError("rth_scene_get failed")
        self._data = d
        return self

    def save_json(self, folder):
        """Writes the reference's JSON scene cache (FileLoader.cs:26-44) into an existing folder."""
        if host_lib().rth_scene_save_json(self.h, folder.encode()) != 0:
            raise NativeError(f"rth_scene_save_json({folder}) failed")

    def load_json(self, folder, prefix=None):
        """Fills the scene from a JSON scene cache instead of building it (materials must be registered first)."""
        err = C.create_string_buffer(512)
        if host_lib().rth_scene_load_json(self.h, folder.encode(), prefix.encode() if prefix else None, err, 512) != 0:
            raise NativeError("rth_scene_load_json: " + err.value.decode())
        d = _SceneData()
        if host_lib().rth_scene_get(self.h, C.byref(d)) != 0:
            raise NativeError("rth_scene_get failed")
        self._data = d
        return self

    def _arr(self, ptr, n, key):
        dt = DTYPES[key]
        raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (int(n) * dt.itemsize,))
        return raw.view(dt).copy()

    def arrays(self):
        """Copies of the flat buffers ConstructScene returns (ObjectHelper.cs:737) plus the scalar constants."""
        d = self._data
        return {
            "BVs": self._arr(d.BVs, d.nBVs, "BV"), "Vertices": self._arr(d.Vertices, d.nVertices, "Vertex"),
            "Triangles": self._arr(d.Triangles, d.nTriangles, "Triangle"),
            "SceneObjects": self._arr(d.SceneObjects, d.nSceneObjects, "SceneObject"),
            "LightObjects": self._arr(d.LightObjects, d.nLightObjects, "LightObject"),
            "Materials": self._arr(d.Materials, d.nMaterials, "Material"),
            "ComponentDatas": self._arr(d.ComponentDatas, d.nComponentDatas, "Int2"),
            "maxBVHDepth": int(d.maxBVHDepth), "emittingObjectsNum": int(d.emittingObjectsNum),
            "sceneBVHStartIndex": int(d.sceneBVHStartIndex), "totArea": float(d.totArea),
            "atlas": np.ctypeslib.as_array(C.cast(d.atlas, C.POINTER(C.c_uint8)),
                                           (int(d.atlasHeight), int(d.atlasWidth), 4)).copy(),
        }

    def __del__(self):
        if getattr(self, "h", None) and _host is not None:
            _host.rth_scene_free(self.h)
            self.h = None


class RtContext:
    """rt_ctx plus Unity-shaped helpers used by the per-kernel parity tests (all calls go through the C-ABI)."""

    def __init__(self, device_id=0):
        self.lib = native_lib()
        h = C.c_void_p()
        rc = self.lib.rt_create(device_id, C.byref(h))
        if rc != 0:
            raise NativeError(f"rt_create({device_id}) failed with status {rc}: no usable gfx950 device")
        self.h = h

    def check(self, rc, what=""):
        if rc < 0:
            raise NativeError(f"{what}: status {rc}: {self.lib.rt_last_error(self.h).decode()}")
        return rc

    def sync(self):
        self.check(self.lib.rt_sync(self.h), "rt_sync")

    def stream(self):
        return self.lib.rt_stream(self.h)

    def flush(self):
        """Launches a dispatch that is held back for merging, without waiting."""
        self.check(self.lib.rt_flush(self.h), "rt_flush")

    def set_fusion(self, enabled):
        self.check(self.lib.rt_set_fusion(self.h, 1 if enabled else 0), "rt_set_fusion")

    def set_tuning(self, name, value):
        """Speed-only implementation choices (see rt_set_tuning in include/rt_native.h)."""
        self.check(self.lib.rt_set_tuning(self.h, name.encode(), int(value)), f"rt_set_tuning({name})")

    def trace(self, on):
        """Starts (a new) or stops the log of launches, event records and waits (rt_debug_trace)."""
        self.check(self.lib.rt_debug_trace(self.h, 1 if on else 0), "rt_debug_trace")

    def trace_text(self):
        need = self.lib.rt_debug_trace_dump(self.h, None, 0)
        buf = C.create_string_buffer(int(need))
        self.lib.rt_debug_trace_dump(self.h, buf, need)
        return buf.value.decode()

    def set_overlap(self, mode):
        """0: off; 1: the next frame's primary rays run beside the previous frame's last kernels (whole-image
        dispatches); 2: also on row bands."""
        self.check(self.lib.rt_set_overlap(self.h, mode), "rt_set_overlap")

    def set_row_band(self, row0, row1):
        self.check(self.lib.rt_set_row_band(self.h, row0, row1), "rt_set_row_band")

    def enable_stats(self, level):
        self.check(self.lib.rt_enable_stats(self.h, level), "rt_enable_stats")

    def open_shader(self, name):
        h = C.c_void_p()
        self.check(self.lib.rt_shader_open(self.h, name.encode(), C.byref(h)), f"rt_shader_open({name})")
        return h

    def create_buffer(self, count, stride, data=None):
        h = C.c_void_p()
        self.check(self.lib.rt_buffer_create(self.h, count, stride, C.byref(h)), "rt_buffer_create")
        if data is not None:
            self.set_buffer_data(h, data)
        return h

    def set_buffer_data(self, buf, data):
        data = np.ascontiguousarray(data)
        self.check(self.lib.rt_buffer_set_data(buf, data.ctypes.data, data.nbytes), "rt_buffer_set_data")

    def get_buffer(self, buf, dtype):
        n = self.lib.rt_buffer_count(buf)
        out = np.empty(n, dtype=dtype)
        self.check(self.lib.rt_buffer_get_data(buf, out.ctypes.data, out.nbytes), "rt_buffer_get_data")
        return out

    def get_buffer_range(self, buf, dtype, first, count):
        out = np.empty(count, dtype=dtype)
        self.check(self.lib.rt_buffer_get_range(buf, out.ctypes.data, first, count), "rt_buffer_get_range")
        return out

    def get_texture_rows(self, tex, w, row0, row1, channels):
        n = row1 - row0
        out = np.empty((n, w, channels) if channels > 1 else (n, w), dtype=np.float32)
        self.check(self.lib.rt_texture_get_rows(tex, out.ctypes.data, row0, row1), "rt_texture_get_rows")
        return out

    def create_texture(self, w, h, fmt):
        t = C.c_void_p()
        self.check(self.lib.rt_texture_create(self.h, w, h, fmt, C.byref(t)), "rt_texture_create")
        return t

    def get_texture(self, tex, w, h, channels):
        out = np.empty((h, w, channels) if channels > 1 else (h, w), dtype=np.float32)
        self.check(self.lib.rt_texture_get_data(tex, out.ctypes.data, out.nbytes), "rt_texture_get_data")
        return out

    def get_stats(self, shader, kernel):
        s = DispatchStats()
        self.check(self.lib.rt_get_stats(shader, kernel, C.byref(s)), "rt_get_stats")
        return s.as_dict()

    def buffer_rows(self, buf, width, row0, row1):
        """[(device pointer, bytes), ...] of image rows [row0, row1) of a per-pixel buffer (rt_buffer_map_rows)."""
        segs, n = (Segment * 4)(), C.c_int()
        self.check(self.lib.rt_buffer_map_rows(buf, width, row0, row1, segs, C.byref(n)), "rt_buffer_map_rows")
        return [(segs[i].ptr, segs[i].bytes) for i in range(n.value)]

    def texture_rows(self, tex, row0, row1):
        seg = Segment()
        self.check(self.lib.rt_texture_map_rows(tex, row0, row1, C.byref(seg)), "rt_texture_map_rows")
        return [(seg.ptr, seg.bytes)]

    def profile_enable(self, timing, only=None):
        """Brackets launches with HIP events; only: name of the single kernel to time (None = all of them)."""
        self.check(self.lib.rt_profile_select(self.h, only.encode() if only else None), "rt_profile_select")
        self.check(self.lib.rt_profile_enable(self.h, 1 if timing else 0), "rt_profile_enable")

    def profile_reset(self):
        self.check(self.lib.rt_profile_reset(self.h), "rt_profile_reset")

    def profile(self):
        """{kernel name: {"launches", "ms", counters...}} for every internal kernel launched since the last reset."""
        out = {}
        for name in KERNEL_NAMES:
            p = KernelProfile()
            self.check(self.lib.rt_profile_get(self.h, name.encode(), C.byref(p)), "rt_profile_get")
            if p.launches:
                out[name] = {"launches": int(p.launches), "ms": float(p.milliseconds), **p.stats.as_dict()}
        return out

    def close(self):
        if self.h:
            self.lib.rt_destroy(self.h)
            self.h = None


class Driver:
    """rth_main: the C++ restatement of Main.cs driving a context through the C-ABI."""

    def __init__(self, ctx, cfg, scene, row0=0, row1=0):
        self.ctx, self.cfg, self.scene = ctx, cfg, scene
        self.lib = host_lib()
        self.h = self.lib.rth_main_create(ctx.h, C.byref(cfg), scene.h, row0, row1)
        if not self.h:
            raise NativeError("rth_main_create failed: " + ctx.lib.rt_last_error(ctx.h).decode())
        self.W, self.H = cfg.Resolution[0], cfg.Resolution[1]

    def _check(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what}: {self.lib.rth_main_last_error(self.h).decode()}")

    def render_frame(self):
        self._check(self.lib.rth_main_render_frame(self.h), "render_frame")

    def update_per_frame(self):
        """Main.Update: UpdatePerFrame, then a requested reload."""
        self._check(self.lib.rth_main_update_per_frame(self.h), "update_per_frame")

    def reload(self, cfg=None):
        """Requests DoUpdateSettings + DoReloadData (an inspector change or a moved scene object): the next frame's
        Update re-runs UpdateSettings(true).  Call scene.rebuild() first if objects moved."""
        self._check(self.lib.rth_main_reload(self.h, C.byref(cfg) if cfg is not None else None), "reload")
        if cfg is not None:
            self.cfg = cfg

    def run_stage(self, stage):
        self._check(self.lib.rth_main_run_stage(self.h, stage), f"run_stage({stage})")

    def run_spatial_pass(self, i):
        self._check(self.lib.rth_main_run_spatial_pass(self.h, i), f"run_spatial_pass({i})")

    def run_transfer(self):
        self._check(self.lib.rth_main_run_transfer(self.h), "run_transfer")

    def run_post_processing(self):
        self._check(self.lib.rth_main_run_post_processing(self.h), "run_post_processing")

    def set_camera(self, position, rotation):
        """transform.position / transform.rotation (quaternion x, y, z, w) from the next frame on."""
        self._check(self.lib.rth_main_set_camera(self.h, _f3(position), (C.c_float * 4)(*[float(x) for x in rotation])),
                    "set_camera")

    def shader(self, name="rt"):
        return self.lib.rth_main_shader(self.h, name.encode())

    def buffer_handle(self, name):
        h = self.lib.rth_main_buffer(self.h, name.encode())
        if not h:
            raise NativeError(f"no driver buffer named {name}")
        return h

    def texture_handle(self, name):
        h = self.lib.rth_main_texture(self.h, name.encode())
        if not h:
            raise NativeError(f"no driver texture named {name}")
        return h

    _BUF_DTYPE = {"Candidates": "Reservoir", "CandidatesB": "Reservoir", "TemporalFrameBuffer": "Reservoir",
                  "HitInfos": "HitInfo", "Triangles": "Triangle", "BVs": "BV", "Vertices": "Vertex"}

    def buffer(self, name):
        return self.ctx.get_buffer(self.buffer_handle(name), DTYPES[self._BUF_DTYPE[name]])

    def texture(self, name):
        ch = 1 if name == "DepthBuffer" else 4
        return self.ctx.get_texture(self.texture_handle(name), self.W, self.H, ch)

    def buffer_rows(self, name, row0, row1):
        """Image rows [row0, row1) of a per-pixel buffer."""
        return self.ctx.get_buffer_range(self.buffer_handle(name), DTYPES[self._BUF_DTYPE[name]], row0 * self.W,
                                         (row1 - row0) * self.W)

    def texture_rows(self, name, row0, row1):
        return self.ctx.get_texture_rows(self.texture_handle(name), self.W, row0, row1, 1 if name == "DepthBuffer" else 4)

    def close(self):
        if self.h:
            self.lib.rth_main_destroy(self.h)
            self.h = None


class RisDriver:
    """rth_ris_main: the C++ restatement of the RIS project's Main.cs driving a context through the C-ABI."""

    def __init__(self, ctx, cfg, scene, row0=0, row1=0):
        self.ctx, self.cfg, self.scene = ctx, cfg, scene
        self.lib = host_lib()
        self.h = self.lib.rth_ris_main_create(ctx.h, C.byref(cfg), scene.h, row0, row1)
        if not self.h:
            raise NativeError("rth_ris_main_create failed: " + ctx.lib.rt_last_error(ctx.h).decode())
        self.W, self.H = cfg.Resolution[0], cfg.Resolution[1]

    def _check(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what}: {self.lib.rth_ris_main_last_error(self.h).decode()}")

    def render_frame(self):
        self._check(self.lib.rth_ris_main_render_frame(self.h), "render_frame")

    def update_per_frame(self):
        self._check(self.lib.rth_ris_main_update_per_frame(self.h), "update_per_frame")

    def reload(self):
        """DoUpdateSettings: the next frame's Update re-runs UpdateSettings -> SetData."""
        self._check(self.lib.rth_ris_main_reload(self.h), "reload")

    def run_stage(self, stage):
        self._check(self.lib.rth_ris_main_run_stage(self.h, stage), f"run_stage({stage})")

    def run_post_processing(self):
        self._check(self.lib.rth_ris_main_run_post_processing(self.h), "run_post_processing")

    def shader(self, name="rt"):
        return self.lib.rth_ris_main_shader(self.h, name.encode())

    def ris_arrays(self):
        """The RIS project's triangle and material arrays as uploaded (worldNormal still zero)."""
        d = _RisSceneData()
        self._check(self.lib.rth_ris_main_scene(self.h, C.byref(d)), "rth_ris_main_scene")
        out = {}
        for key, ptr, n, dt in (("Tris", d.Tris, d.nTris, "RisTri"), ("Materials", d.Materials, d.nMaterials, "RisMaterial")):
            raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (int(n) * DTYPES[dt].itemsize,))
            out[key] = raw.view(DTYPES[dt]).copy()
        return out

    def buffer_handle(self, name):
        h = self.lib.rth_ris_main_buffer(self.h, name.encode())
        if not h:
            raise NativeError(f"no driver buffer named {name}")
        return h

    def texture_handle(self, name):
        h = self.lib.rth_ris_main_texture(self.h, name.encode())
        if not h:
            raise NativeError(f"no driver texture named {name}")
        return h

    _BUF_DTYPE = {"Candidates": "Float4", "HitInfos": "RisHitInfo", "Tris": "RisTri", "BVs": "BV"}

    def buffer(self, name):
        return self.ctx.get_buffer(self.buffer_handle(name), DTYPES[self._BUF_DTYPE[name]])

    def texture(self, name):
        return self.ctx.get_texture(self.texture_handle(name), self.W, self.H, 4)

    def close(self):
        if self.h:
            self.lib.rth_ris_main_destroy(self.h)
            self.h = None
