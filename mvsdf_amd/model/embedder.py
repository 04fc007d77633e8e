"""NeRF positional encoding [x, sin(2^k x), cos(2^k x)]_k (reference code/model/embedder.py:10-50).

The hot path never calls this: the encoding is fused into the HIP kernels (csrc/tile_engine.h::mv_pe_rows,
csrc/diff_mlp.hip::k_pe_global).  Kept for API parity (plotting / external callers)."""
import torch


def get_embedder(multires, input_dims=3):
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, multires)

    def embed(x):
        out = [x]
        for f in freqs:
            out += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(out, -1)
    return embed, input_dims * (1 + 2 * multires)
