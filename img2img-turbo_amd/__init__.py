"""MI355X-native one-step image-to-image generator (Pix2Pix_Turbo / CycleGAN_Turbo forward path).

Python here is tensor plumbing only: weight loading/packing, building the op program, and the
reference-compatible ``.forward()`` API.  Every FLOP of the forward runs in the hand-written gfx950
kernels of ``csrc/`` behind the C ABI declared in ``include/i2i_turbo.h``.
Import as ``img2img_turbo_amd``.
"""
__version__ = "0.1.0"
