"""MI355X-native DDPM reverse-sampling hot path of Point Diffusion-Refinement.

Layout mirrors the reference's importable surfaces for this path:
    point_diffusion_refinement_amd.pointnet2_ops   <-> pointnet2_ops_lib/pointnet2_ops
    point_diffusion_refinement_amd.pointnet2       <-> pointnet2/{util,util_fastdpmv2,emd,
                                                       chamfer_loss_new,models/...}.py
All native work goes through libpdr_hip.so (include/pdr_hip.h).
"""
__version__ = "0.1.0"
