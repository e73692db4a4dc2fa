"""MI355X-native hot path of Fast-Image-Editing-with-Generative-Models (LCM img2img + Canny ControlNet).

Layout: ``csrc/`` HIP kernels + C-ABI (``include/fie.h``); Python host code mirroring the reference's
``FastEditor`` / diffusers-pipeline interface.  Import as ``fie_amd`` (see ../fie_amd.py).
"""
__version__ = "0.1.0"
