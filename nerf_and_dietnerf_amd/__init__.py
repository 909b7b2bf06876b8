"""nerf_and_dietnerf_amd -- MI355X-native drop-in for the NeRF render hot path of
Sahar-E/NeRF-and-DietNeRF (ray generation, stratified + inverse-CDF sampling, positional encoding,
coarse/fine MLP forward, alpha compositing) as hand-written HIP kernels behind a C ABI
(include/nerf_mi355.h).  See DESIGN.md / INTEGRATION.md.

Importing this package does not need a GPU; creating a Context does (there is no CPU fallback).
"""
from . import _lib
from ._lib import (NERF_MEM_DEVICE, NERF_MEM_HOST, NERF_NET_COARSE, NERF_NET_FINE, NERF_PRECISION_F16X3,
                   NERF_PRECISION_FP32)
from .render import (Context, NeRF, NetHandle, default_context, get_rays_directions, get_size_of_splits,
                     get_z_vals_from_prob_dist_func, get_z_values, model_predict, positional_encoding_for_views,
                     positional_encoding_for_xyz, ray_marching, render_rays, split_to_batches)
from .keras_h5 import load_nerf_checkpoint, read_keras_weights, save_nerf_checkpoint, write_keras_weights
from .sharding import allreduce_mean, dist_world, gather_slabs, ray_slab, render_image_sharded
from .dataset import RayDataset, c2w_to_rays_prepare_ds, fit, prepare_ds
from .dietnerf import DietNeRF
from . import config, scene
from .scene import estimate_point_of_interest_in_scene
from .config import (get_nerf, get_num_of_batches, get_psnr_values, get_train_data, load_config,
                     save_psnr_values)
from .datasets import (get_data_from_blender, get_data_from_colmap, get_train_images_indices, load_llff_data,
                       poses_avg, recenter_poses, spherify_poses)
from .video import (get_c2w_matrices_between_2_c2w, get_c2w_matrices_between_2_c2w_with_stretch, get_path_c2w_matrices,
                    get_l_to_r_c2w_matrices_to_render, get_path_c2w_matrices_to_render,
                    get_sphere_c2w_matrices_to_render,
                    get_rotation_matrix_from_source_to_dest_mats, interpolation_type_slerp_for_c2w,
                    slerp_rotation_matrix, get_l_to_r_c2w_matrices, get_sphere_matrices, get_sphere_matrix, histogram_equalize_depth,
                    render_video)
from .weights import blob_size, glorot_blob, layer_shapes

__all__ = [n for n in dir() if not n.startswith("_")]
