"""Write the Blosc (zstd) input plate of tools/cli_e2e_bench.py (one position, T time points) and the deskew config under DIR:
   python tools/cli_prof_setup.py DIR [T]   — then profile the CLI itself:
   rocprofv3 --kernel-trace --stats -d OUT -- python3 -m biahub_amd deskew -i DIR/in.zarr/A/1/0 -c DIR/d.yml -o DIR/out.zarr --cluster debug"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from biahub_amd import io

root = Path(sys.argv[1]); root.mkdir(parents=True, exist_ok=True)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shape = (T, 2, 256, 1024, 1024)
rng = np.random.default_rng(0)
vol = (rng.poisson(6, shape[2:]) + 110 + (60 * np.sin(np.arange(shape[-1]) / 50.0)).astype(np.int64)).astype(np.uint16)
io.create_empty_plate(root / "in.zarr", [("A", "1", "0")], ["c0", "c1"], shape, scale=(1, 1, 0.313, 0.116, 0.116), dtype=np.uint16, compressor="blosc")
p = io.open_ome_zarr(root / "in.zarr" / "A/1/0")
for t in range(T):
    for c in range(2):
        p.data[t, c] = vol
(root / "d.yml").write_text("pixel_size_um: 0.116\nls_angle_deg: 36.17\npx_to_scan_ratio: 0.371\nscan_step_um: 0.313\n"
                            "keep_overhang: true\naverage_n_slices: 3\noverhang_fill: mean\n")
