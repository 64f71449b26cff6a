"""Bind arena views to the parameter structs the C ABI takes (TnPropNet / TnField)."""
from __future__ import annotations

from .arena import ParamArena
from .config import ThermalNerfactoModelConfig
from .ops import FieldParams, PropNetParams, level_resolutions


def prop_params(arena: ParamArena, prefix: str, i: int, cfg: ThermalNerfactoModelConfig, with_grads: bool = False) -> PropNetParams:
    a = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
    names = {
        "table": f"{prefix}.{i}.mlp_base.0.hash_table",
        "w0": f"{prefix}.{i}.mlp_base.1.layers.0.weight",
        "b0": f"{prefix}.{i}.mlp_base.1.layers.0.bias",
        "w1": f"{prefix}.{i}.mlp_base.1.layers.1.weight",
        "b1": f"{prefix}.{i}.mlp_base.1.layers.1.bias",
    }
    grads = {k: arena.grad_view(n) for k, n in names.items()} if with_grads else None
    return PropNetParams(
        **{k: arena.view(n) for k, n in names.items()},
        num_levels=a["num_levels"],
        log2_hashmap_size=a["log2_hashmap_size"],
        res=level_resolutions(a["num_levels"], a.get("base_res", 16), a["max_res"]),
        grads=grads,
    )


def field_params(arena: ParamArena, prefix: str, cfg: ThermalNerfactoModelConfig, with_grads: bool = False) -> FieldParams:
    names = {
        "table": f"{prefix}.mlp_base.model.0.hash_table",
        "w0": f"{prefix}.mlp_base.model.1.layers.0.weight",
        "b0": f"{prefix}.mlp_base.model.1.layers.0.bias",
        "w1": f"{prefix}.mlp_base.model.1.layers.1.weight",
        "b1": f"{prefix}.mlp_base.model.1.layers.1.bias",
        "hw0": f"{prefix}.mlp_head.layers.0.weight",
        "hb0": f"{prefix}.mlp_head.layers.0.bias",
        "hw1": f"{prefix}.mlp_head.layers.1.weight",
        "hb1": f"{prefix}.mlp_head.layers.1.bias",
        "hw2": f"{prefix}.mlp_head.layers.2.weight",
        "hb2": f"{prefix}.mlp_head.layers.2.bias",
        "emb": f"{prefix}.embedding_appearance.embedding.weight",
    }
    grads = {k: arena.grad_view(n) for k, n in names.items()} if with_grads else None
    channels = arena.layout[names["hw2"]][1][0]
    return FieldParams(
        **{k: arena.view(n) for k, n in names.items()},
        num_levels=cfg.num_levels,
        log2_hashmap_size=cfg.log2_hashmap_size,
        res=level_resolutions(cfg.num_levels, cfg.base_res, cfg.max_res),
        num_channels=channels,
        grads=grads,
    )
