"""YAML <-> settings model helpers and the resume fingerprint — mirror of ``biahub/utils/config.py``."""

from __future__ import annotations

import hashlib
import json
from pathlib import Path

import yaml


def settings_fingerprint(settings) -> str:
    """sha256[:16] of the sorted JSON dump (biahub/utils/config.py:16-26): the resume token."""
    payload = json.dumps(settings.model_dump(mode="json"), sort_keys=True, default=str)
    return hashlib.sha256(payload.encode()).hexdigest()[:16]


def update_model(model_instance, update_dict):
    """Copy with updates; nested dicts update nested models field-wise (config.py:29-46)."""
    changes = {}
    for key, value in update_dict.items():
        if isinstance(value, dict) and hasattr(model_instance, key):
            changes[key] = getattr(model_instance, key).model_copy(update=value)
        else:
            changes[key] = value
    return model_instance.model_copy(update=changes)


def model_to_yaml(model, yaml_path: Path) -> None:
    """Dump non-None fields in declaration order (config.py:49-91)."""
    if not hasattr(model, "model_dump"):
        raise TypeError("The 'model' object does not have a 'dict()' method.")
    clean = {k: v for k, v in model.model_dump().items() if v is not None}
    with open(Path(yaml_path), "w+") as f:
        yaml.dump(clean, f, default_flow_style=False, sort_keys=False)


def yaml_to_model(yaml_path: Path, model):
    """Instantiate ``model`` from a YAML file (config.py:94-141)."""
    if not callable(getattr(model, "__init__", None)):
        raise TypeError("The provided model must be a class with a callable constructor.")
    try:
        with open(Path(yaml_path)) as f:
            raw = yaml.safe_load(f)
    except FileNotFoundError:
        raise FileNotFoundError(f"The YAML file '{yaml_path}' does not exist.") from None
    return model(**raw)
