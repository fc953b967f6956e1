from vptq_amd.utils.pack import absorb_perm, absorb_perm_layer, pack_index, unpack_index_tensor

__all__ = ["pack_index", "unpack_index_tensor", "absorb_perm_layer", "absorb_perm"]
