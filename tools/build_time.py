"""Scene build time (mi_pt_create: upload + BVH2 + 8-wide collapse + records) of a workload, phases on stderr.  usage: python tools/build_time.py street"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MI_PT_BUILD_TIMING"] = "1"
import bench  # noqa: E402
from vk_gltf_renderer_amd import pathtracer as ptmod  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "street"
scene = ptmod.Scene(bench.scene_path(name, 0))
for k in range(2):  # the second create runs with a warm HIP context
    t0 = time.perf_counter()
    t = ptmod.PathTracer(scene)
    t.synchronize()
    print(f"{name}: mi_pt_create #{k} {1e3 * (time.perf_counter() - t0):.1f} ms, {scene.num_triangles} triangles, collapse = {'host' if os.environ.get('MI_PT_HOST_COLLAPSE') else 'device'}")
    if k == 1:  # what a moving instance pays: mi_pt_update_render_nodes = a rebuild over the resident geometry (same table again)
        d = scene.desc.contents
        for _ in range(8):
            t1 = time.perf_counter()
            t.update_render_nodes(d.renderNodes, d.numRenderNodes, d.renderNodeVisible)
            t.synchronize()
            print(f"{name}: mi_pt_update_render_nodes {1e3 * (time.perf_counter() - t1):.1f} ms (MI_PT_REINSERT_UPDATE={os.environ.get('MI_PT_REINSERT_UPDATE', 'default')})", flush=True)
    t.close()
