"""consistentid_amd: MI355X-native (gfx950) implementation of ConsistentID's UNet-denoise
hot path behind the reference's own plugin surface.  See DESIGN.md.

Importing the package does not touch the GPU; the HIP library is loaded on first use
(``consistentid_amd._lib.load()``) and there is no CPU fallback."""
from .unet_spec import UNetConfig, sd15_config, sdxl_config, tiny_config  # noqa: F401

__all__ = ["UNetConfig", "sd15_config", "sdxl_config", "tiny_config"]
__version__ = "0.1.0"
