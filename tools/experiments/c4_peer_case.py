import sys, tempfile, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from forge3d_amd import datasets
from forge3d_amd.session import TerrainSession
import test_gpu_parity as t
dem, cam, kw = datasets.rainier_proxy_scene(2048)
v, i = datasets.proxy_buildings(dem, kw["spacing"][0])
frames = 2
k = dict(kw, spp=2, max_frames=frames, min_frames=frames, variance_threshold=1e30, mesh_vertices=v, mesh_indices=i)
with TerrainSession(dem, 4096, 4096, cam, memory_budget_bytes=16 << 30, mesh_builder=1, **k) as s:
    s.enqueue_frames(0, frames, True)
    sah = s.resolve(frames)
tmp = tempfile.mkdtemp()
for key in ("rgba", "albedo", "normal", "depth"):
    np.save(f"{tmp}/{key}.npy", sah[key])
import torch
bounds = [0, 1500, 2600, 4096]
streams = [torch.cuda.Stream() for _ in range(3)]
sessions = [TerrainSession(dem, 4096, 4096, cam, row_begin=b, row_end=e, memory_budget_bytes=16 << 30, stream=st.cuda_stream, **k)
            for b, e, st in zip(bounds[:-1], bounds[1:], streams)]
exports = [s.halo_export() for s in sessions]
for n, s in enumerate(sessions):
    if n > 0:
        s.halo_connect(0, exports[n - 1])
    if n < 2:
        s.halo_connect(1, exports[n + 1])
for s in sessions:
    s.enqueue_batch_strip(0, frames, True)
torch.cuda.synchronize()
print("timeouts", [s.halo_timeouts() for s in sessions])
parts = [s.resolve(frames) for s in sessions]
for key in ("rgba", "albedo", "normal", "depth"):
    got = np.concatenate([p[key] for p in parts], 0)
    a, b = got.reshape(4096, 4096, -1), sah[key].reshape(4096, 4096, -1)
    bad = np.argwhere(~np.all((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))), axis=-1))
    print(key, "differing pixels", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "first", bad[:5].tolist())
