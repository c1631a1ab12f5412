"""fujiyama-renderer_amd: MI355X-native ray-intersection + integrator core for
the Fujiyama renderer's hot path.

The directory name carries a hyphen (repo layout contract), so import it
through the `fujiyama_renderer_amd` shim at the repo root or with importlib.

    host     ctypes face of lib/libfjscene.so  (Si* scene API, scene text parser)
    gpu      ctypes face of lib/libfjgpu.so    (HIP core: scene, render_tiles, trace)
    fujiyama Py3 scene-description emitter (mirror of the reference's python API)
    workloads / synth   BASELINE.json workloads on seeded synthetic assets
"""
from . import ffi, host, gpu  # noqa: F401

__all__ = ["ffi", "host", "gpu"]
