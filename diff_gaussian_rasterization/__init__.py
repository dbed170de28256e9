"""Drop-in for the third-party CUDA module the reference imports at
src/model/decoder/cuda_splatting.py:5-8 (requirements.txt:17).  Putting this
repository on PYTHONPATH makes `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer` resolve to the gfx950 HIP
rasterizer (libgsr_hip.so); see INTEGRATION.md."""
from styl3r_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
