"""ctypes binding of libisg_hip.so (C ABI declared in include/isg.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is not on
an MI355X, the ops raise.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libisg_hip.so")

ISG_OK = 0
ABI_VERSION = 20

# name -> (restype, argtypes); one entry per symbol declared in include/isg.h
SIGNATURES = {
    "isg_abi_version": (c_int, []),
    "isg_status_string": (c_char_p, [c_int]),
    "isg_last_hip_error": (c_char_p, []),
    "isg_graph_ptr": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "isg_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "isg_csr_build": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                              c_void_p]),
    "isg_instr_gate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_node_to_edge_mask": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "isg_gatv2_mp_fwd": (c_int, [c_void_p] * 12 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "isg_gatv2_mp_fwd_f16": (c_int, [c_void_p] * 12 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "isg_graph_plan_build": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64] + [c_void_p] * 9 + [c_size_t, c_void_p]),
    "isg_graph_edge_ptr": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "isg_scatter_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_node_gate": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_topk_gumbel": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_uint64, c_void_p, c_int32,
                                c_float, c_void_p, c_void_p, c_void_p]),
    "isg_topk_threshold": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_float, c_uint64,
                                   c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_scatter_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_graph_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int32, c_void_p,
                               c_int64, c_int32, c_void_p]),
    "isg_instr_attn_graphnorm_residual": (c_int, [c_void_p] * 7 + [c_double, c_void_p, c_void_p, c_int64, c_int32,
                                                                  c_void_p]),
    "isg_simple_topk": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_uint64, c_void_p, c_int32, c_void_p,
                                c_void_p, c_void_p]),
    "isg_instr_attn_graphnorm_residual_bwd": (c_int, [c_void_p] * 7 + [c_double] + [c_void_p] * 7 + [c_int64, c_int32, c_void_p]),
    "isg_global_attn_pool_bwd": (c_int, [c_void_p] * 9 + [c_int64, c_int32, c_void_p]),
    "isg_instr_gate_bwd": (c_int, [c_void_p] * 6 + [c_int64, c_int32, c_void_p]),
    "isg_node_gate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int64, c_int32, c_void_p]),
    "isg_linear_wgrad_splits": (c_int64, [c_int64, c_int32, c_int32]),
    "isg_linear_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int64,
                                 c_void_p]),
    "isg_topk_gumbel_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_uint64, c_void_p, c_int32,
                                    c_float, c_void_p, c_void_p, c_void_p]),
    "isg_gatv2_mp_bwd": (c_int, [c_void_p] * 19 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "isg_node_to_edge_mask_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "isg_split_bf16x3": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "isg_linear_bf16x6": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_void_p]),
    "isg_linear_skinny": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                  c_int32, c_void_p]),
    "isg_linear_bf16x6_f16": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                      c_int32, c_int32, c_int32, c_void_p]),
    "isg_split_bf16x3_frag_elems": (c_int64, [c_int64, c_int32]),
    "isg_split_bf16x3_frag": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "isg_linear_panel": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                 c_int32, c_int32, c_int32, c_void_p]),
    "isg_gather_add": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                               c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_mha_small": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                              c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_add_layernorm": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_float, c_void_p, c_int32,
                                  c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_linear_panel_multi": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                       c_int32, c_int32, c_int32, c_int32, c_int64, c_void_p]),
    "isg_split_f16x2_frag_elems": (c_int64, [c_int64, c_int32]),
    "isg_split_f16x2_frag": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_linear_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                 c_int32, c_int32, c_int32, c_int64, c_void_p]),
    "isg_linear_f16x3_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_int32, c_int64, c_void_p]),
    "isg_gatv2_mp_fwd_rowmax": (c_int, [c_void_p] * 13 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "isg_gatv2_edge_logits": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int32,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_int32, c_int32, c_int32, c_float, c_void_p]),
    "isg_gatv2_edge_logits_f16": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int32,
                                          c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_int32, c_int32, c_int32, c_float, c_void_p]),
    "isg_gatv2_mp_fwd_logits": (c_int, [c_void_p] * 12 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "isg_gatv2_mp_fwd_logits_f16": (c_int, [c_void_p] * 11 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                            c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "isg_gatv2_mp_fwd_logits_planes": (c_int, [c_void_p] * 12 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                               c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "isg_split_f16x2_rows": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_linear_f16x3_tile": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "isg_row_absmax": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "isg_planes32_elems": (c_int64, [c_int64, c_int32]),
    "isg_split_planes32": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_instr_gate_planes32": (c_int, [c_void_p] * 6 + [c_int64, c_int32, c_void_p]),
    "isg_linear_h3p": (c_int, [c_void_p] * 9 + [c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "isg_gatv2_mp_fwd_planes": (c_int, [c_void_p] * 13 + [c_int64, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "isg_linear_h3p_store_policy": (c_int, [c_int32]),
    "isg_tile_plan_capacity": (c_int64, [c_int64, c_int64, c_int64, c_int32, c_int32]),
    "isg_tile_plan": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "isg_mgat_dense_tail": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32] + [c_void_p] * 12 + [c_double] +
                            [c_void_p] * 11 + [c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "isg_instr_gate_planes": (c_int, [c_void_p] * 6 + [c_int64, c_int32, c_void_p]),
    "isg_tile_plan_edge_planes": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_cat_mul_rowmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_node_gate_planes": (c_int, [c_void_p] * 7 + [c_int32, c_void_p, c_int64, c_int32, c_void_p]),
    "isg_gatv2_layer_conv": (c_int, [c_void_p, c_void_p] + [c_void_p] * 15 + [c_int64] + [c_void_p] * 3 +
                             [c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "isg_readout_tile": (c_int, [c_void_p, c_int32] + [c_void_p] * 16 + [c_int64, c_int64, c_int32, c_void_p]),
    "isg_edge_planes": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "isg_gatv2_tile_conv": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p] + [c_void_p] * 10 + [c_int64] +
                            [c_void_p] * 3 + [c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_float,
                                              c_void_p]),
    "isg_global_attn_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                     c_void_p]),
}

_lib = None


class IsgError(RuntimeError):
    pass


def load():
    """Load libisg_hip.so once; raise (never fall back) when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IsgError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    v = lib.isg_abi_version()
    if v != ABI_VERSION:
        raise IsgError(f"libisg_hip.so ABI version {v}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != ISG_OK:
        lib = load()
        msg = lib.isg_status_string(status).decode()
        hip = lib.isg_last_hip_error().decode()
        raise IsgError(f"{what}: {msg} (status {status})" + (f"; HIP: {hip}" if hip else ""))
