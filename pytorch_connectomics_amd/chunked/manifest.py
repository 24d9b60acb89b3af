"""Crash-safe resume manifest (contract of the reference's connectomics/chunked/manifest.py:18-96):
JSON {"config": {...}, "completed": [...]}, written through <path>.tmp + fsync + os.replace; a resumed run
whose chunk_shape / overlap / output_dtype / output_shape differ is refused."""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Iterable, Mapping

_CHECKED = ("chunk_shape", "overlap", "output_dtype", "output_shape")


class ManifestConfigMismatch(ValueError):
    pass


class ResumeManifest:
    def __init__(self, path, config: Mapping[str, Any]):
        self.path = Path(path)
        self.config = dict(config)
        self._completed: set[str] = set()

    @classmethod
    def load_or_create(cls, path, config: Mapping[str, Any], *, overwrite: bool = False) -> "ResumeManifest":
        path = Path(path)
        if overwrite and path.exists():
            path.unlink()
        m = cls(path, config)
        if not path.exists():
            m._write()
            return m
        payload = json.loads(path.read_text())
        m._completed = set(payload.get("completed", []))
        old = payload.get("config", {})
        diffs = [f"{k}: existing={old[k]} requested={m.config[k]}" for k in _CHECKED
                 if k in m.config and k in old and old[k] != m.config[k]]
        if diffs:
            raise ManifestConfigMismatch(f"Resume manifest at {path} disagrees with requested config: "
                                         + "; ".join(diffs)
                                         + ". Re-run with overwrite=True or change the requested config.")
        return m

    @property
    def completed(self) -> set[str]:
        return set(self._completed)

    def mark_completed(self, chunk_key: str) -> None:
        self.mark_many([chunk_key])

    def mark_many(self, chunk_keys: Iterable[str]) -> None:
        new = {k for k in chunk_keys if k not in self._completed}
        if new:
            self._completed |= new
            self._write()

    def _write(self) -> None:
        tmp = self.path.with_suffix(self.path.suffix + ".tmp")
        with tmp.open("w") as fh:
            json.dump({"config": self.config, "completed": sorted(self._completed)}, fh, indent=2)
            fh.flush()
            os.fsync(fh.fileno())
        os.replace(tmp, self.path)


__all__ = ["ResumeManifest", "ManifestConfigMismatch"]
