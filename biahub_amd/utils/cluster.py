"""Per-position resource sizing and the RESOURCES: contract — mirror of ``biahub/utils/cluster.py``."""

from __future__ import annotations

import json
import math
import os

import numpy as np


def echo_resources(num_cpus: int, mem_gb: int, time_minutes: int) -> None:
    """Print ``RESOURCES:{json}`` for the Nextflow pipeline (cluster.py:17-43, nextflow/modules/common.nf:6-17)."""
    print("RESOURCES:" + json.dumps({"cpus": int(num_cpus), "mem_gb": int(mem_gb), "time_minutes": int(time_minutes)}))


def get_submitit_cluster(local: bool = False, cluster: str | None = None) -> str:
    """'debug' under CI=true, else the explicit cluster, else legacy --local (cluster.py:46-59)."""
    if os.environ.get("CI") == "true":
        return "debug"
    if cluster is not None:
        return cluster
    return "local" if local else "slurm"


def estimate_resources(shape, dtype=np.float32, ram_multiplier: float = 1.0, time_multiplier: float = 1.0,
                       max_num_cpus: int = 64, min_ram_per_cpu: int = 4, min_time_minutes: int = 30):
    """(time_minutes, num_cpus, gb_ram_per_cpu) for a (T,C,Z,Y,X) dataset (cluster.py:62-140)."""
    if len(shape) != 5:
        raise ValueError("The shape must be a 5-tuple (T, C, Z, Y, X).")
    if ram_multiplier <= 0 or time_multiplier <= 0:
        raise ValueError("ram_multiplier and time_multiplier must be > 0.")
    T, C, Z, Y, X = shape
    num_cpus = 1 if os.environ.get("CI") == "true" else min(T * C, max_num_cpus)
    gb_per_volume = Z * Y * X * np.dtype(dtype).itemsize / 2**30
    gb_ram_per_cpu = math.ceil(max(min_ram_per_cpu, gb_per_volume * ram_multiplier))
    minutes = max(min_time_minutes, T * C * time_multiplier)
    return int(math.ceil(minutes / 10.0) * 10), int(num_cpus), int(gb_ram_per_cpu)
