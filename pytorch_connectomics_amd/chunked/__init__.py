"""connectomics.chunked counterpart: chunk grid, halo math and the crash-safe resume manifest."""
from .chunk_grid import ChunkRef, build_chunk_grid
from .halo import resolve_halo_region
from .manifest import ManifestConfigMismatch, ResumeManifest

__all__ = ["ChunkRef", "build_chunk_grid", "resolve_halo_region", "ResumeManifest", "ManifestConfigMismatch"]
