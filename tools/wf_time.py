"""Time the multi-bounce PBR tracer on the adjudication scene: tools/wf_time.py SIZE FRAMES (F3D_WF_FRAMES_PER_LANE overrides)."""
import os
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from forge3d_amd import wavefront as w
size = int(sys.argv[1]); frames = int(sys.argv[2])
best = None
for _ in range(3):
    out = w.render_scene(w.adjudication_scene(), size, size, frames)
    best = out if best is None or out["loop_seconds"] < best["loop_seconds"] else best
print("fpl %s size %d frames %d: kernel %.1f ms, %.2f Gpaths/s, %.2f Gvertices/s" % (os.environ.get("F3D_WF_FRAMES_PER_LANE", "auto"), size, frames, best["loop_seconds"] * 1e3, best["paths"] / best["loop_seconds"] / 1e9, best["path_vertices"] / best["loop_seconds"] / 1e9))
