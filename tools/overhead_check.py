"""Per-ray fixed cost of the per-lane trace kernel: camera rays that miss everything (camera looks away from Box.glb),
with the packet kernel disabled (MI_PT_NO_PACKET=1) vs enabled."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu
from vk_gltf_renderer_amd import pathtracer as ptmod

s0 = pu.Setup(os.path.join(ROOT, "assets", "Box.glb"), 1920, 1080, max_depth=2, hdr_path=os.path.join(ROOT, "assets", "std_env.hdr"))
cam = s0.scene.camera(0)
cam.center[0], cam.center[1], cam.center[2] = cam.eye[0], cam.eye[1], cam.eye[2] + 10.0  # look away from the box
s = pu.Setup(os.path.join(ROOT, "assets", "Box.glb"), 1920, 1080, max_depth=2, hdr_path=os.path.join(ROOT, "assets", "std_env.hdr"), camera=cam)
tr = ptmod.PathTracer(s.scene)
tr.set_environment(s.hdr); tr.resize(1920, 1080); tr.set_frame_info(s.frame_info); tr.set_sky(s.sky)
r = ptmod.HeadlessRenderer(tr, s.params)
r.render(16, in_flight=16); tr.synchronize()
tr.enable_timing(True)
r.render(64, in_flight=16); tr.synchronize()
t = tr.frame_timing()
print("packet" if not os.environ.get("MI_PT_NO_PACKET") else "per-lane", {k: round(v / 64, 4) for k, v in t.items() if k.endswith("Ms")})
tr.close()
