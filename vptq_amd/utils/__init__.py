from vptq_amd.utils.pack import pack_index, unpack_index_tensor

__all__ = ["pack_index", "unpack_index_tensor"]
