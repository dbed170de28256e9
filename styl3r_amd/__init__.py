"""styl3r_amd -- MI355X-native (gfx950) hot path of Styl3R: the differentiable
Gaussian-splatting rasterizer and the ViT encoder kernels behind the
reference's own operator interfaces.  See DESIGN.md."""

__version__ = "0.1.0"
