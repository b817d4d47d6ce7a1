"""mmgt_amd: MI355X-native Stage-2 denoising path of MMGT (UNet3D + MM-HAA, DDIM loop, VAE decode).

The arithmetic lives in hand-written HIP kernels (mmgt_amd/csrc, built into libmmgt_hip.so and reached through the
C ABI declared in include/mmgt_hip.h); the Python here mirrors the reference's operator interface
(src/models/unet_3d.py, src/pipelines/pipeline_pose2vid_long.py) and does plumbing only.
"""
__version__ = "0.1.0"
