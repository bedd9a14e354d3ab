"""Helper of the staging-ring wrap tests (run as a subprocess: the ring size and the A/B switches are read once per process).
Renders a sequence of different frames back to back WITHOUT a Finish in between, several rounds, each frame blitted to a keeper
texture; prints one sha256 per frame.  usage: ring_wrap_driver.py <backend.so> <rounds>"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from webrender_amd import scenes  # noqa: E402
from webrender_amd.harness import render_pipelined  # noqa: E402


def frames(rounds):
    out = []
    for r in range(rounds):
        out += [scenes.cfg2_overlapping_rects(width=512, height=512, n=300, seed=40 + r),
                scenes.masked_rects(width=512, height=512, n=60, seed=5 + r),
                scenes.image_grid(width=512, height=512, n=40, seed=50 + r),
                scenes.cfg5_many_rects(width=512, height=512, n=3000, seed=7 + r),
                scenes.gradient_grid(width=512, height=512, n=20, seed=60 + r)]
    return out


if __name__ == "__main__":
    got = render_pipelined(sys.argv[1], frames(int(sys.argv[2])))
    print(json.dumps([hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest() for g in got]))
