"""Architecture tables for the CORE forward path (product side).

ref: lib/build_model.py:31-49 (SAM ViT sizes), lib/support_branch.py:19-26 (accepted SigLIP names / widths).
SigLIP tower hyper-parameters follow the published open_clip model configs (open_clip_torch 2.31.0; the
package is not vendored by the reference and cannot be checked offline). The GELU flavour is a parameter PER TOWER
(`v_gelu` for the timm vision trunk, `t_gelu` for the open_clip text tower; "erf" | "tanh"): the published SigLIP2
configs set the tanh approximation through `act_kwargs`, and whether that reaches both towers is exactly what cannot be
verified here, so the two are independent knobs (legacy key `gelu` sets both).
"""
from __future__ import annotations

SAM = {
    "sam_base": dict(dim=768, depth=12, heads=12, global_idx=(2, 5, 8, 11)),
    "sam_large": dict(dim=1024, depth=24, heads=16, global_idx=(5, 11, 17, 23)),
    "sam_huge": dict(dim=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31)),
}
SAM_COMMON = dict(window=14, img=1024, patch=16, out=256)

SIGLIP = {
    "ViT-B-16-SigLIP-384": dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, image=384, vocab=32000, ctx=64,
                                t_depth=12, t_heads=12, t_mlp=3072, v_gelu="erf", t_gelu="erf"),
    "ViT-B-16-SigLIP2-384": dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, image=384, vocab=256000, ctx=64,
                                 t_depth=12, t_heads=12, t_mlp=3072, v_gelu="tanh", t_gelu="tanh"),
    "ViT-L-16-SigLIP-384": dict(dim=1024, depth=24, heads=16, mlp=4096, patch=16, image=384, vocab=32000, ctx=64,
                                t_depth=24, t_heads=16, t_mlp=4096, v_gelu="erf", t_gelu="erf"),
    "ViT-L-16-SigLIP2-384": dict(dim=1024, depth=24, heads=16, mlp=4096, patch=16, image=384, vocab=256000, ctx=64,
                                 t_depth=24, t_heads=16, t_mlp=4096, v_gelu="tanh", t_gelu="tanh"),
    "ViT-SO400M-14-SigLIP-384": dict(dim=1152, depth=27, heads=16, mlp=4304, patch=14, image=384, vocab=32000, ctx=64,
                                     t_depth=27, t_heads=16, t_mlp=4304, v_gelu="erf", t_gelu="erf"),
}


def sam_cfg(name: str) -> dict:
    if name not in SAM:
        raise ValueError(f"Invalid SAM model: {name}")
    return dict(SAM[name], **SAM_COMMON)


def siglip_cfg(name: str) -> dict:
    if name not in SIGLIP:
        raise ValueError(f"Invalid SigLIP model: {name}")
    return normalize_siglip_cfg(dict(SIGLIP[name]))


def normalize_siglip_cfg(g: dict) -> dict:
    """Accept the legacy single `gelu` key (sets both towers) next to the per-tower `v_gelu` / `t_gelu`."""
    if "gelu" in g:
        g.setdefault("v_gelu", g["gelu"])
        g.setdefault("t_gelu", g["gelu"])
    for k in ("v_gelu", "t_gelu"):
        if g.get(k, "erf") not in ("erf", "tanh"):
            raise ValueError(f"{k} must be 'erf' or 'tanh', got {g[k]!r}")
        g.setdefault(k, "erf")
    return g
