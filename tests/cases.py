"""Named parity cases shared by the golden generator, the CPU tests and the GPU tests.

Each case -> dict(shapes, tris, mats, rd, sd, frames) where frames is a list of
`time` seeds rendered into ONE canvas without clearing (progressive accumulation,
/root/reference/src/main.cpp:277-290)."""
import numpy as np

from simple_raytracer_amd import records as R, scenes as S


def _case(scene, w, h, spp, nb=10, frames=(12345,), cam=None, fov=1.0, **kw):
    shapes, tris, mats = scene
    cam = S.default_camera() if cam is None else cam
    rd = R.render_data(w, h, spp, nb, fov_scale=fov, camera_to_world=cam, time=frames[0], **kw)
    sd = R.scene_data(len(shapes))
    return dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=sd, frames=list(frames))


def glass_scene():
    """Glass-heavy: nested refractive spheres and a camera INSIDE a glass sphere."""
    mats = np.zeros(4, R.MATERIAL)
    mats[0] = R.material((0.8, 0.8, 0.8))
    mats[1] = R.material((0.95, 1.0, 0.95), smoothness=1.0, transmittance=1.0, refraction_index=1.5)
    mats[2] = R.material((1.0, 0.9, 0.9), smoothness=0.6, transmittance=0.8, refraction_index=1.33, specular=0.1)
    mats[3] = R.material((1, 1, 1), emission=(1, 0.9, 0.7), emission_strength=3.0)
    shapes = np.zeros(6, R.SHAPE)
    shapes[0] = R.plane(0, (0, -1, 0), (0, 1, 0))
    shapes[1] = R.sphere(1, (0, 0.5, 5.0), 1.2)      # camera sits inside this one
    shapes[2] = R.sphere(1, (0.0, 0.2, 1.0), 1.2)
    shapes[3] = R.sphere(2, (0.0, 0.2, 1.0), 0.6)    # nested
    shapes[4] = R.sphere(2, (-2.2, 0.0, 0.5), 1.0)
    shapes[5] = R.sphere(3, (2.5, 2.0, 0.0), 0.5)
    return shapes, R.box_triangles(), mats


def box_instances_scene():
    """Box instances sharing the 12 box triangles (src/shape.cpp:76-89), including
    rotated and non-uniformly scaled copies made like the GUI's gizmo does."""
    mats = np.zeros(3, R.MATERIAL)
    mats[0] = R.material((0.8, 0.8, 0.9))
    mats[1] = R.material((0.9, 0.4, 0.2), smoothness=0.3, specular=0.2)
    mats[2] = R.material((0.3, 0.5, 0.9), smoothness=0.9, metallic=0.8)
    tris = R.box_triangles()
    shapes = np.zeros(5, R.SHAPE)
    shapes[0] = R.plane(0, (0, -1, 0), (0, 1, 0))
    shapes[1] = R.box_model(1, 0, (-2.0, 0.0, -1.0))
    shapes[2] = R.box_model(2, 0, (1.5, 0.0, -2.0))
    shapes[3] = R.model(1, tris, 0, 12, R.mat_mul(R.translate((0.0, -0.3, 1.0)), R.mat_mul(R.euler_yxz(0.5, 0.4, 0.0), R.scale_matrix((0.4, 0.7, 0.4)))))
    shapes[4] = R.model(2, tris, 0, 12, R.mat_mul(R.translate((2.6, 1.2, 0.3)), R.mat_mul(R.euler_yxz(-1.0, 0.2, 0.0), R.scale_matrix((0.3, 1.1, 0.6)))))
    return shapes, tris, mats


def empty_scene():
    """What the app starts with: no shapes, one default material, 12 box triangles
    (src/main.cpp:95-102)."""
    mats = np.zeros(1, R.MATERIAL)
    mats[0] = R.material()
    return np.zeros(0, R.SHAPE), R.box_triangles(), mats


def build_cases():
    cam_tilt = R.camera_matrix((1.0, 1.2, 4.5), 0.25, -0.15)
    return {
        "spheres": _case(S.sphere_scene(), 64, 64, 8),
        "spheres_accum": _case(S.sphere_scene(), 48, 40, 3, frames=(12345, 987654321, 77)),
        "normals": _case(S.mixed_test_scene(), 64, 48, 2, show_normals=True),
        "glass": _case(glass_scene(), 64, 48, 6),
        "boxes": _case(box_instances_scene(), 64, 48, 4, cam=cam_tilt),
        "mixed": _case(S.mixed_test_scene(), 64, 48, 4, cam=cam_tilt, fov=0.8),
        "mesh_smooth": _case(S.mesh_scene(2, 10, 11, smooth=True), 48, 40, 3),
        "mesh_flat": _case(S.mesh_scene(1, 8, 7, smooth=False), 48, 40, 3),
        "empty": _case(empty_scene(), 32, 24, 2),
        "one_bounce": _case(S.sphere_scene(), 40, 30, 4, nb=1),
        "ragged": _case(S.sphere_scene(), 37, 29, 3, nb=4),
        "even_time": _case(S.sphere_scene(), 32, 24, 4, frames=(4096,)),  # time*5304 loses low bits
    }


def render_case(render_fn, case, sky):
    """Accumulate all frames of a case with `render_fn(rd, canvas) -> canvas`."""
    canvas = None
    for tm in case["frames"]:
        rd = case["rd"].copy()
        rd["time"] = np.uint32(tm & 0xFFFFFFFF)
        canvas = render_fn(rd, canvas)
    return canvas
