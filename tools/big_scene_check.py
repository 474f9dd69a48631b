"""Build-time / memory / throughput check on a Bistro-class triangle count (SURVEY §8d config 4 stand-in: the atrium generator
at a detail level that yields ~2.8 M triangles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import parity_util as pu
from vk_gltf_renderer_amd import scenegen, pathtracer as ptmod

detail = float(sys.argv[1]) if len(sys.argv) > 1 else 2.6
t0 = time.time()
path = scenegen.scene_atrium_class("/tmp/big.glb", seed=4321, detail=detail, tex_size=256)
t1 = time.time()
s = pu.Setup(path, 1920, 1080, max_depth=6)
t2 = time.time()
print(f"generate {t1-t0:.1f}s  load {t2-t1:.1f}s  triangles {s.scene.num_triangles}")
for bvh in (0,):
    t3 = time.time()
    tr = ptmod.PathTracer(s.scene, bvh=bvh)
    t4 = time.time()
    tr.resize(1920, 1080); tr.set_frame_info(s.frame_info); tr.set_sky(s.sky)
    r = ptmod.HeadlessRenderer(tr, s.params)
    r.render(8, in_flight=8); tr.synchronize()
    t5 = time.time()
    r.render(32, in_flight=8); tr.synchronize()
    t6 = time.time()
    img = tr.read_accum()
    print(f"bvh={bvh} create {t4-t3:.2f}s  first batch {t5-t4:.2f}s  32 frames {t6-t5:.3f}s -> {1920*1080*32/(t6-t5)/1e6:.1f} Msamples/s  finite {np.isfinite(img).all()} mean {img[...,:3].mean():.4f}")
    tr.close()
