"""deskew CLI, Blosc (zstd) input plate -> Blosc-lz4 output store, one position of 8 volumes, with BH_IO_THREADS = default, 12, 8, 4:
per-stage seconds of the pipeline (BH_PIPE_TIMING) — does the operator thread's encode stage starve for CPU while the reader's
zstd threads hold every core?"""
import os, shutil, subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from biahub_amd import io

root = Path(tempfile.mkdtemp(prefix="bh_e2e_", dir=os.environ.get("BH_E2E_DIR", "/tmp")))
shape = (4, 2, 256, 1024, 1024)
rng = np.random.default_rng(0)
vol = (rng.poisson(6, shape[2:]) + 110 + (60 * np.sin(np.arange(shape[-1]) / 50.0)).astype(np.int64)).astype(np.uint16)
src = root / "in.zarr"
io.create_empty_plate(src, [("A", "1", "0")], ["c0", "c1"], shape, scale=(1, 1, 0.313, 0.116, 0.116), dtype=np.uint16, compressor="blosc")
p = io.open_ome_zarr(src / "A/1/0")
for t in range(shape[0]):
    for c in range(shape[1]):
        p.data[t, c] = vol
(root / "d.yml").write_text("pixel_size_um: 0.116\nls_angle_deg: 36.17\npx_to_scan_ratio: 0.371\nscan_step_um: 0.313\n"
                            "keep_overhang: true\naverage_n_slices: 3\noverhang_fill: mean\n")
print("cores:", len(os.sched_getaffinity(0)), flush=True)
for threads in sys.argv[1:] or ("", "12", "8", "4"):
    out = root / f"out_{threads or 'default'}.zarr"
    env = dict(os.environ, BH_IO_THREADS=threads, BH_ZARR_COMPRESSOR="blosc-lz4", BH_PIPE_TIMING=os.environ.get("BH_PIPE_TIMING", "1"))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "biahub_amd", "deskew", "-i", str(src / "A/1/0"), "-c", str(root / "d.yml"), "-o", str(out),
                        "--cluster", "debug"], env=env, capture_output=True, text=True, cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0, r.stdout + r.stderr
    dt = time.perf_counter() - t0
    print(f"BH_IO_THREADS={threads or 'default'}: {dt:.2f} s for 8 volumes", flush=True)
    for l in r.stderr.splitlines():
        if l.startswith("pipe timing") or "encode_volume_device" in l:
            print("   " + l.replace(str(root), ""), flush=True)
    shutil.rmtree(out)
shutil.rmtree(root)
