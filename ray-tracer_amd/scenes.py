"""Scene descriptions used by the tests and the benchmark (the values of the reference's committed Unity scene).

Only describes scenes; the building is done by the native scene builder (host/scene_builder.cpp).
"""
import os

from .bindings import Mesh, Scene

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def _materials(scene, light_brightness):
    # materialInputs of BVH_ReSTIR_DL/Assets/Scenes/Ray Tracer.unity:689-728 (the dragon's normal map is not in
    # the snapshot, so material 1 carries none)
    scene.add_material(0.0, (1, 1, 0), roughness=0.0, metallicity=0.0)
    scene.add_material(0.0, (0, 1, 0), roughness=0.8, metallicity=1.0)
    scene.add_material(light_brightness, (1, 1, 1), roughness=1.0, metallicity=1.0)


def build_scene_s0(subdivide=True, dragon_depth=None, light_brightness=40000.0, threads=0):
    """Scene S0: Suzanne light + dragon, transforms of Ray Tracer.unity:468-506,785-822.

    subdivide=True gives the "80kScene*" stand-in: the 8,712-triangle dragon with every triangle split in 9
    (78,408 triangles), because the reference's 80k-triangle dragon OBJ is not in the snapshot.
    """
    dragon = Mesh.load(os.path.join(ASSET_DIR, "dragon8k.rtmesh"))
    if subdivide:
        dragon = dragon.subdivide9()
    suzanne = Mesh.load(os.path.join(ASSET_DIR, "suzanne.rtmesh"))
    s = Scene(split_resolution=5, max_depth_scene_bvh=10)
    _materials(s, light_brightness)
    md, ms = s.add_mesh(dragon), s.add_mesh(suzanne)
    if dragon_depth is None:
        dragon_depth = 24 if subdivide else 18
    # sceneObjects order of the Unity scene: Suzanne first (its BVH occupies nodes 0..1894 in the committed dump)
    s.add_object(ms, (0.5, 4.4, 14.6), (0, 190, 0), (2, 2, 2), 2, 14)
    s.add_object(md, (-4, -2, 13), (0, 0, 0), (12, 12, 12), 1, dragon_depth)
    return s.build(threads)


def build_scene_instanced(grid=(16, 8), emissive_objects=10, emissive_tris=1000, unique=True, subdivide=True,
                          max_depth_scene_bvh=16, threads=0, emissive_fans=False):
    """The deep-BVH scene: a grid of dragons (each vertex-jittered so it is distinct geometry when unique=True)
    lit by discs of emissive triangles (sectors x rings; emissive_fans=True: plain triangle fans, whose slivers
    through one vertex make every ray near a disc test dozens of triangles)."""
    base = Mesh.load(os.path.join(ASSET_DIR, "dragon8k.rtmesh"))
    if subdivide:
        base = base.subdivide9()
    s = Scene(split_resolution=5, max_depth_scene_bvh=max_depth_scene_bvh)
    _materials(s, 40000.0)
    gx, gy = grid
    shared = s.add_mesh(base) if not unique else None
    k = 0
    for j in range(gy):
        for i in range(gx):
            mid = shared if not unique else s.add_mesh(base.jitter(1000 + k, 1e-4))
            s.add_object(mid, (-4 + 9.0 * (i - gx / 2), -2 + 9.0 * (j - gy / 2) * 0.0 + 0.0, 13 + 9.0 * j), (0, 0, 0),
                         (12, 12, 12), 1, 24 if subdivide else 18)
            k += 1
    disc = Mesh.disc(emissive_tris, rings=not emissive_fans)
    md = s.add_mesh(disc)
    for e in range(emissive_objects):
        s.add_object(md, (-4 + 9.0 * gx * (e / max(1, emissive_objects - 1) - 0.5), 12.0, 13 + 4.5 * gy), (90, 0, 0),
                     (6, 6, 6), 2, 12)
    return s.build(threads)


def procedural_textures(seed=7):
    """Small RGBA8 textures for the textured-material tests: a colour checker, a bump map, a roughness map and a
    normal map (tangent-space style: mostly +z)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:32, 0:32]
    col = np.zeros((32, 32, 4), np.uint8)
    col[..., 0] = 40 + 200 * ((xx // 4 + yy // 4) % 2)
    col[..., 1] = (xx * 8) % 256
    col[..., 2] = (yy * 8) % 256
    col[..., 3] = 255
    bump = np.zeros((8, 4, 4), np.uint8)
    bump[..., 0] = rng.integers(0, 4, (8, 4))          # bump offsets of up to 3/255 along the incoming direction
    rough = np.zeros((8, 8, 4), np.uint8)
    rough[..., 0] = rng.integers(30, 255, (8, 8))
    normals = np.zeros((16, 16, 4), np.uint8)
    normals[..., 0] = 128 + rng.integers(-40, 40, (16, 16))
    normals[..., 1] = 128 + rng.integers(-40, 40, (16, 16))
    normals[..., 2] = 255
    return {"col": col, "bump": bump, "roughness": rough, "normals": normals}


def build_scene_textured(threads=0):
    """Scene S0 (8.7k dragon) whose dragon material takes colour, bump, roughness and normals from textures and whose
    light is tinted by a colour texture: every atlas-sampling path of the kernels except the metallicity texture
    (which the reference's atlas builder cannot produce, ObjectHelper.cs:373-379,430-437)."""
    dragon = Mesh.load(os.path.join(ASSET_DIR, "dragon8k.rtmesh"))
    suzanne = Mesh.load(os.path.join(ASSET_DIR, "suzanne.rtmesh"))
    tex = procedural_textures()
    s = Scene(split_resolution=5, max_depth_scene_bvh=10)
    s.add_material(0.0, (1, 1, 0), roughness=0.0, metallicity=0.0)
    s.add_material(0.0, (0, 1, 0), roughness=0.8, metallicity=1.0, textures=tex)
    light = tex["col"].copy()
    light[..., :3] = 255 - light[..., :3] // 4
    s.add_material(40000.0, (1, 1, 1), roughness=1.0, metallicity=1.0, textures={"col": light[:16, :24]})
    md, ms = s.add_mesh(dragon), s.add_mesh(suzanne)
    s.add_object(ms, (0.5, 4.4, 14.6), (0, 190, 0), (2, 2, 2), 2, 14)
    s.add_object(md, (-4, -2, 13), (0, 0, 0), (12, 12, 12), 1, 18)
    return s.build(threads)
