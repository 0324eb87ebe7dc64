"""Zarr-to-zarr throughput of the deskew command on one GPU (I/O + PCIe + kernels): uncompressed stores with 1 and the
default number of I/O threads, then iohub-style Blosc (zstd-1, bit shuffle) stores on both sides."""
import os, shutil, subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from biahub_amd import io

root = Path(tempfile.mkdtemp(prefix="bh_e2e_", dir=os.environ.get("BH_E2E_DIR", "/tmp")))
shape = (2, 2, 256, 1024, 1024)
rng = np.random.default_rng(0)
vol = (rng.poisson(6, shape[2:]) + 110 + (60 * np.sin(np.arange(shape[-1]) / 50.0)).astype(np.int64)).astype(np.uint16)
srcs = {}
NT = {None: 2, "blosc": 4}  # time points per position: the compressed plate is twice as long (warm-up of pinned blocks and
# allocator caches is ~3 volumes of every run; raw outputs of that length would not fit the scratch disk)
for comp in (None, "blosc"):
    src = srcs[comp] = root / f"in_{comp}.zarr"
    io.create_empty_plate(src, [("A", "1", "0"), ("A", "2", "0")], ["c0", "c1"], (NT[comp],) + shape[1:], scale=(1, 1, 0.313, 0.116, 0.116),
                          dtype=np.uint16, compressor=comp)
    t0 = time.perf_counter()
    for pos in ("A/1/0", "A/2/0"):
        p = io.open_ome_zarr(src / pos)
        for t in range(NT[comp]):
            for c in range(shape[1]):
                p.data[t, c] = vol
    print(f"wrote input plate ({comp or 'uncompressed'}): {4 * NT[comp] * vol.nbytes / (time.perf_counter() - t0) / 1e9:.2f} GB/s", flush=True)
(root / "d.yml").write_text("pixel_size_um: 0.116\nls_angle_deg: 36.17\npx_to_scan_ratio: 0.371\nscan_step_um: 0.313\n"
                            "keep_overhang: true\naverage_n_slices: 3\noverhang_fill: mean\n")
V = 8 * int(np.prod(shape[2:]))
def du(path):
    return sum(f.stat().st_size for f in Path(path).rglob("*") if f.is_file())


# "blosc-lz4": iohub-style Blosc input, output store with the lz4 inner codec — the result is permuted AND compressed on the GPU
# (csrc/lz4.hip), compressed frames cross PCIe, the host only writes files
for comp, threads in ((None, "1"), (None, ""), ("blosc", ""), ("blosc-lz4", "")):
    out = root / f"out_{comp}_{threads or 'default'}.zarr"
    src = srcs["blosc" if comp == "blosc-lz4" else comp]
    env = dict(os.environ, BH_IO_THREADS=threads, BH_ZARR_COMPRESSOR=comp or "none", BH_PIPE_TIMING="1")
    def run(positions, dest):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "biahub_amd", "deskew", "-i", *[str(src / p) for p in positions], "-c",
                            str(root / "d.yml"), "-o", str(dest), "--cluster", "debug"], env=env, capture_output=True, text=True,
                           cwd=str(Path(__file__).resolve().parent.parent))
        assert r.returncode == 0, r.stdout + r.stderr
        run.timing = [l for l in r.stderr.splitlines() if l.startswith("pipe timing")]
        return time.perf_counter() - t0

    dt = run(["A/1/0", "A/2/0"], out)
    timing8 = run.timing
    # the same command on ONE position (4 volumes): the difference is what 4 more volumes cost once the process is up
    # (interpreter + torch import + library and plate set-up are ~2 s of every CLI call)
    dt1 = run(["A/1/0"], root / "one.zarr")
    shutil.rmtree(root / "one.zarr")
    nv = 4 * NT["blosc" if comp else None]  # volumes of the two-position run
    marg = max(dt - dt1, 1e-9) / (nv // 2)
    print(f"deskew CLI, {comp or 'uncompressed'} stores, BH_IO_THREADS={threads or 'default'}: {dt:.2f} s for {nv} volumes of {shape[2:]} "
          f"uint16 -> {V / 8 * nv / dt / 1e9:.2f} Gvox/s (in {du(src) / 1e9:.1f} GB on disk, out {du(out) / 1e9:.1f} GB on disk, "
          f"{nv * 342 * 1024 * 1517 * 4 / 1e9:.1f} GB raw); {nv // 2} volumes {dt1:.2f} s -> marginal {marg:.3f} s per volume = "
          f"{V / 8 / marg / 1e9:.2f} Gvox/s", flush=True)
    for l in timing8:
        print("   " + l.replace(str(root), ""), flush=True)
    shutil.rmtree(out)
shutil.rmtree(root)
