"""Crash-safe resume manifest (contract of the reference's connectomics/chunked/manifest.py:18-96):
JSON {"config": {...}, "completed": [...]}, written through <path>.tmp + fsync + os.replace; a resumed run
whose chunk_shape / overlap / output_dtype / output_shape differ is refused."""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Iterable, Mapping

# the geometry a resumed run must agree on with the run that wrote the manifest
_CHECKED = ("chunk_shape", "overlap", "output_dtype", "output_shape")


class ManifestConfigMismatch(ValueError):
    pass


def _atomic_json(path: Path, payload) -> None:
    """Readers see the old file or the new one, never a torn write: temp file in the same directory, fsync, rename."""
    scratch = path.with_name(path.name + ".tmp")
    with open(scratch, "w") as fh:
        fh.write(json.dumps(payload, indent=2))
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(scratch, path)


class ResumeManifest:
    def __init__(self, path, config: Mapping[str, Any]):
        self.path, self.config = Path(path), dict(config)
        self._completed: set[str] = set()

    @classmethod
    def load_or_create(cls, path, config: Mapping[str, Any], *, overwrite: bool = False) -> "ResumeManifest":
        manifest = cls(path, config)
        if overwrite:
            manifest.path.unlink(missing_ok=True)
        try:
            stored = json.loads(manifest.path.read_text())
        except FileNotFoundError:
            manifest._write()
            return manifest
        manifest._completed.update(stored.get("completed", ()))
        before = stored.get("config", {})
        clashes = "; ".join(f"{key}: existing={before[key]} requested={manifest.config[key]}" for key in _CHECKED
                            if key in before and key in manifest.config and before[key] != manifest.config[key])
        if clashes:
            raise ManifestConfigMismatch(f"Resume manifest at {manifest.path} disagrees with requested config: {clashes}"
                                         ". Re-run with overwrite=True or change the requested config.")
        return manifest

    @property
    def completed(self) -> set[str]:
        return set(self._completed)

    def mark_completed(self, chunk_key: str) -> None:
        self.mark_many((chunk_key,))

    def mark_many(self, chunk_keys: Iterable[str]) -> None:
        before = len(self._completed)
        self._completed.update(chunk_keys)
        if len(self._completed) != before:
            self._write()

    def _write(self) -> None:
        _atomic_json(self.path, {"config": self.config, "completed": sorted(self._completed)})


__all__ = ["ResumeManifest", "ManifestConfigMismatch"]
