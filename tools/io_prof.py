import os, sys, time, tempfile, shutil
from pathlib import Path
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from biahub_amd import io
from biahub_amd.deskew import _fast_deskew_czyx
root = Path(tempfile.mkdtemp(prefix="bh_io_", dir="/dev/shm"))
shape = (1, 1, 256, 1024, 1024)
src = root / "in.zarr"
io.create_empty_plate(src, [("A", "1", "0")], ["c0"], shape, dtype=np.uint16)
vol = (np.random.default_rng(0).random(shape[2:]) * 400 + 100).astype(np.uint16)
p = io.open_ome_zarr(src / "A/1/0"); p.data[0, 0] = vol
kw = dict(ls_angle_deg=36.17, px_to_scan_ratio=0.371, keep_overhang=True, average_n_slices=3, overhang_fill="mean")
out = root / "out.zarr"
io.create_empty_plate(out, [("A", "1", "0")], ["c0"], (1, 1, 342, 1024, 1517), dtype=np.float32)
q = io.open_ome_zarr(out / "A/1/0")
for rep in range(3):
    t0 = time.perf_counter(); v = p.data.read_volume(0, 0); t1 = time.perf_counter()
    r = _fast_deskew_czyx(v[None], device="cuda", **kw); t2 = time.perf_counter()
    q.data.write_volume(0, 0, r[0]); t3 = time.perf_counter()
    print(f"read {t1-t0:.3f} s ({v.nbytes/(t1-t0)/1e9:.1f} GB/s)  op {t2-t1:.3f} s  write {t3-t2:.3f} s ({r.nbytes/(t3-t2)/1e9:.1f} GB/s)", flush=True)
shutil.rmtree(root)
