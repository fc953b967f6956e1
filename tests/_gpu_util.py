"""Helpers shared by the GPU tests, smoke() and bench.py: move an oracle
LayerSpec (numpy bit patterns) into a vptq_amd.VQuantLinear on a device."""
import numpy as np
import torch

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def bits_to_tensor(bits: np.ndarray, dtype: str, device) -> torch.Tensor:
    """uint16 bit patterns -> fp16/bf16 tensor with the same bits."""
    a = np.ascontiguousarray(bits).view(np.int16)
    return torch.from_numpy(a.copy()).to(device).view(TORCH_DT[dtype])


def tensor_to_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def spec_to_module(L, device):
    """oracle LayerSpec -> vptq_amd.VQuantLinear holding identical bits."""
    import vptq_amd
    dt = TORCH_DT[L.dtype]
    m = vptq_amd.VQuantLinear(
        L.in_features, L.out_features,
        vector_lens=[L.outlier_vector_len, L.vector_len],
        num_centroids=[L.num_outlier_centroids, L.num_centroids],
        num_res_centroids=[-1, L.num_res_centroids if L.num_res_centroids > 0 else -1],
        group_num=L.num_codebooks, group_size=L.group_size, outlier_size=L.outlier_size,
        indices_as_float=False, enable_norm=L.weight_scale is not None,
        enable_perm=L.perm is not None, is_indice_packed=True, bias=L.bias is not None,
        dtype=dt, device=device, enable_proxy_error=False)

    def put(param, bits, as_float=True):
        if as_float:
            t = bits_to_tensor(bits, L.dtype, device)
        else:
            t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).to(device)
        assert t.numel() == param.numel(), (t.shape, param.shape)
        param.data = t.reshape(param.shape)

    m.indices.data = torch.from_numpy(np.ascontiguousarray(L.indices).view(np.int32).copy()).to(device)
    put(m.centroids.weight, L.centroids)
    if L.num_res_centroids > 0:
        put(m.res_centroids.weight, L.res_centroids)
    if L.enable_outlier:
        put(m.outlier_centroids.weight, L.outlier_centroids)
        put(m.outlier_indices, L.outlier_indices, as_float=False)
    if L.perm is not None:
        put(m.perm, L.perm, as_float=False)
    if L.weight_scale is not None:
        put(m.weight_scale, L.weight_scale)
        put(m.weight_bias, L.weight_bias)
    if L.bias is not None:
        put(m.bias, L.bias)
    return m.eval()


def module_desc(m, need_inv_perm=False, prefetch=None):
    """C-ABI descriptor of a vptq_amd.VQuantLinear (tests call the ABI directly
    to reach flags the Python API does not expose)."""
    from vptq_amd import _backend as B
    return B.make_layer_desc(
        indices=m.indices, centroids=m.centroids.weight,
        res_centroids=m.res_centroids.weight if m.enable_residual else None,
        outlier_indices=m.outlier_indices,
        outlier_centroids=m.outlier_centroids.weight if m.enable_outlier else None,
        perm=m.perm if m.enable_perm else None, weight_scale=m.weight_scale,
        weight_bias=m.weight_bias, bias=m.bias, in_features=m.in_features,
        out_features=m.out_features, vector_len=m.vector_len, num_codebooks=m.num_codebooks,
        num_centroids=m.num_centroids,
        num_res_centroids=m.num_res_centroids if m.enable_residual else 0,
        group_size=m.group_size, outlier_size=m.outlier_size,
        outlier_vector_len=m.outlier_vector_len,
        num_outlier_centroids=m.num_outlier_centroids, need_inv_perm=need_inv_perm,
        prefetch=prefetch)


def gemv_abi(m, x, flags=0, workspace=True, out_f32=False):
    """vptq_quant_gemv through the C ABI with explicit flags; workspace=True hands over the scratch buffer
    vptq_quant_gemv_workspace_bytes asks for (what the kernel-name query assumes), False passes NULL."""
    from vptq_amd import _backend as B
    desc, keep = module_desc(m)
    tokens = x.numel() // x.shape[-1]
    y = torch.empty(x.shape[:-1] + (m.out_features,), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    ws = None
    if workspace:
        nb = B.lib().vptq_quant_gemv_workspace_bytes(desc, tokens, flags)
        if nb:
            ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    B.check(B.lib().vptq_quant_gemv(desc, x.data_ptr(), y.data_ptr(), tokens, flags | (B.GEMV_OUT_F32 if out_f32 else 0),
                                    None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
                                    B.current_stream_ptr(x.device)), "vptq_quant_gemv")
    if ws is not None:
        torch.cuda.current_stream(x.device).synchronize()   # (the scratch buffer dies with this frame)
    return y


def kernel_name(m, tokens=1, flags=0):
    from vptq_amd import _backend as B
    desc, keep = module_desc(m)
    n = B.lib().vptq_quant_gemv_kernel_name(desc, tokens, flags)
    return None if n is None else n.decode()


def module_to_spec(m):
    """vptq_amd.VQuantLinear (any device) -> oracle LayerSpec with the same bits."""
    from oracle import vptq_oracle as vo
    dtype = "f16" if m.centroids.weight.dtype == torch.float16 else "bf16"
    u16 = lambda t: t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    L = vo.LayerSpec(m.in_features, m.out_features, m.vector_len, m.num_centroids,
                     m.num_res_centroids if m.enable_residual else -1, m.num_codebooks,
                     m.group_size, m.outlier_size if m.enable_outlier else 0,
                     m.outlier_vector_len, m.num_outlier_centroids, dtype)
    L.indices = m.indices.detach().cpu().numpy()
    L.centroids = u16(m.centroids.weight).reshape(m.num_codebooks, m.num_centroids, m.vector_len)
    if m.enable_residual:
        L.res_centroids = u16(m.res_centroids.weight).reshape(m.num_codebooks, m.num_res_centroids, m.vector_len)
    if m.enable_outlier:
        L.outlier_indices = u16(m.outlier_indices)
        L.outlier_centroids = u16(m.outlier_centroids.weight).reshape(
            1, m.num_outlier_centroids, m.outlier_vector_len)
    if m.enable_perm:
        L.perm = u16(m.perm)
    if m.enable_norm:
        L.weight_scale, L.weight_bias = u16(m.weight_scale), u16(m.weight_bias)
    if m.bias is not None:
        L.bias = u16(m.bias)
    return L
