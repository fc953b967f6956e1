from vptq_amd.utils.pack import (absorb_perm, absorb_perm_layer, dtype_convert, pack_index,
                                 pack_layer_tensors, pack_state_dict, unpack_index_tensor)

__all__ = ["pack_index", "unpack_index_tensor", "dtype_convert", "pack_layer_tensors",
           "pack_state_dict", "absorb_perm_layer", "absorb_perm"]
