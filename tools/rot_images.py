"""usage: python tools/rot_images.py <lib> ...  -- wall time per frame of rotated / perspective image scenes (no bench workload covers
the textured general-quad path), each library in turn"""
import sys, time
sys.path.insert(0, ".")
from webrender_amd import scenes
from webrender_amd.harness import render_direct
for lib in sys.argv[1:]:
    for name, mk in (("rotated_images", lambda: scenes.rotated_images(n=60)),
                     ("perspective_images", lambda: scenes.rotated_images(n=60, perspective="all")),
                     ("perspective_images_quad", lambda: scenes.rotated_images(n=60, perspective="all", encoding="quad"))):
        render_direct(lib, mk(), frames=3)
        t0 = time.perf_counter()
        render_direct(lib, mk(), frames=40)
        t1 = time.perf_counter()
        render_direct(lib, mk(), frames=20)
        t2 = time.perf_counter()
        print(lib.split("/")[-1], name, "ms/frame %.3f" % (1e3 * ((t1 - t0) - (t2 - t1)) / 20))
