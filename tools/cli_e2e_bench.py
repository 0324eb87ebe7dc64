"""Zarr-to-zarr throughput of the deskew command on one GPU (I/O + PCIe + kernels), BH_IO_THREADS sweep."""
import os, shutil, subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from biahub_amd import io

root = Path(tempfile.mkdtemp(prefix="bh_e2e_", dir=os.environ.get("BH_E2E_DIR", "/tmp")))
shape = (2, 2, 256, 1024, 1024)
src = root / "in.zarr"
io.create_empty_plate(src, [("A", "1", "0"), ("A", "2", "0")], ["c0", "c1"], shape, scale=(1, 1, 0.313, 0.116, 0.116), dtype=np.uint16)
rng = np.random.default_rng(0)
vol = (rng.random(shape[2:]) * 400 + 100).astype(np.uint16)
t0 = time.perf_counter()
for pos in ("A/1/0", "A/2/0"):
    p = io.open_ome_zarr(src / pos)
    for t in range(shape[0]):
        for c in range(shape[1]):
            p.data[t, c] = vol
print(f"wrote input plate: {8 * vol.nbytes / (time.perf_counter() - t0) / 1e9:.2f} GB/s", flush=True)
(root / "d.yml").write_text("pixel_size_um: 0.116\nls_angle_deg: 36.17\npx_to_scan_ratio: 0.371\nscan_step_um: 0.313\n"
                            "keep_overhang: true\naverage_n_slices: 3\noverhang_fill: mean\n")
V = 8 * int(np.prod(shape[2:]))
for threads in ("1", "8"):
    out = root / f"out{threads}.zarr"
    env = dict(os.environ, BH_IO_THREADS=threads)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "biahub_amd", "deskew", "-i", str(src / "A/1/0"), str(src / "A/2/0"), "-c",
                        str(root / "d.yml"), "-o", str(out), "--cluster", "debug"], env=env, capture_output=True, text=True,
                       cwd=str(Path(__file__).resolve().parent.parent))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    print(f"deskew CLI, BH_IO_THREADS={threads}: {dt:.2f} s for 8 volumes of {shape[2:]} uint16 -> {V / dt / 1e9:.2f} Gvox/s "
          f"(in {V * 2 / 1e9:.1f} GB, out {8 * 342 * 1024 * 1517 * 4 / 1e9:.1f} GB)", flush=True)
shutil.rmtree(root)
