"""Output-path mirroring and sbatch-file parsing — ``biahub/utils/ngff.py:42-98``, ``biahub/cli/parsing.py:198-249``."""

from __future__ import annotations

from pathlib import Path

PROVENANCE_METADATA_KEYS = ("biahub-*", "waveorder", "cytoland")  # ngff.py:24


def get_output_paths(input_paths, output_zarr_path, ensure_unique_positions: bool = None):
    """Mirror the trailing row/col/fov of every input under ``output_zarr_path``; with
    ``ensure_unique_positions`` a repeated position gets ``d<k>`` appended to its column part."""
    seen: dict[str, int] = {}
    out = []
    for path in input_paths:
        parts = list(Path(path).parts[-3:])
        name = "/".join(parts)
        if ensure_unique_positions and name in seen:
            seen[name] += 1
            parts[1] = f"{parts[1]}d{seen[name]}"
        elif ensure_unique_positions:
            seen[name] = 0
        out.append(Path(output_zarr_path, *parts))
    return out


def sbatch_to_submitit(filepath: str) -> dict:
    """``#SBATCH --k=v`` -> ``slurm_k``; ``#LOCAL --k=v`` -> ``k``; ints parsed, dashes -> underscores."""
    params = {}
    with open(filepath) as f:
        for line in f:
            for keyword, prefix in (("SBATCH", "slurm_"), ("LOCAL", "")):
                head = f"#{keyword} --"
                if line.startswith(head):
                    key, value = line[len(head):].strip().split("=", 1)
                    key = key.replace("-", "_").strip()
                    value = value.strip()
                    try:
                        value = int(value)
                    except ValueError:
                        pass
                    params[prefix + key] = value
    return params
