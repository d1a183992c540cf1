"""Developer probe: the GPU-driven street drive against the oracle-driven fixture, frame by frame (where do they part?)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from eskf_lio_amd import replay  # noqa: E402
import make_drive_fixture as mk  # noqa: E402

ref = np.load(os.path.join(ROOT, "tests", "golden", "drive_c4.npz"))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else int(ref["frames"])
cfg = mk.drive_config()
backend = replay.DeviceBackend(cfg, 0)
odo = replay.Odometry(cfg, backend)
traj = odo.run(mk.lazy_events(frames))
print(odo.report())
it = np.array(backend.iterations)
prev = 0.0
for f, ((s1, T1), T0) in enumerate(zip(traj, ref["poses"])):
    dt = float(np.linalg.norm(T1[:3, 3] - T0[:3, 3]))
    flag = ""
    if f > 0 and it[f - 1] != ref["iterations"][f - 1]:
        flag += f" rounds {it[f - 1]} vs {ref['iterations'][f - 1]}"
    if backend.kept[f] != ref["kept"][f]:
        flag += f" kept {backend.kept[f]} vs {ref['kept'][f]}"
    if dt > 10 * max(prev, 1e-13) or flag or f % 25 == 0:
        print(f"frame {f}: |dt| {dt:.3e}{flag}")
    prev = max(prev, dt)
print("removed", backend.removed, "fixture", ref["removed"])
