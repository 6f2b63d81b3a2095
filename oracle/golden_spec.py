"""Golden-vector case list shared by oracle/make_golden.py (reference side) and tests (oracle / HIP side).

Every case is described by plain data: which operator, its sizes, and the *names* from which inputs and weights
are regenerated with oracle/weights.py (nothing random, nothing stored but the reference outputs).
"""
import torch

from . import unet_ref, weights

T = lambda name, shape, kind="normalish": torch.from_numpy(weights.fill_like(name, shape, kind))

# ---- G3: per-operator cases (full channel widths, reduced spatial sizes) ------------------------------------------
# (case name, op kind, params)
OP_CASES = [
    ("gn_silu_320", "gn_silu", dict(C=320, N=2, H=4, W=8)),
    ("gn_silu_960", "gn_silu", dict(C=960, N=2, H=4, W=8)),
    ("gn_silu_1920", "gn_silu", dict(C=1920, N=1, H=4, W=4)),
    ("normalize_640", "normalize", dict(C=640, N=2, H=4, W=8)),
    ("layernorm_640", "layernorm", dict(C=640, N=2, L=32)),
    ("conv3x3_320_640", "conv3x3", dict(Cin=320, Cout=640, N=2, H=6, W=10)),
    ("down_320", "down", dict(C=320, N=2, H=8, W=12)),
    ("conv1x1_960_320", "conv1x1", dict(Cin=960, Cout=320, N=2, H=4, W=8)),
    ("up_640", "up", dict(C=640, N=1, H=4, W=6)),
    ("res_320_640", "res", dict(Cin=320, Cout=640, N=2, H=8, W=8)),
    ("res_640_640", "res", dict(Cin=640, Cout=640, N=2, H=4, W=8)),
    ("attn_self_320", "attn", dict(C=320, heads=5, ctx=None, N=2, L=128, Lc=None)),
    ("attn_self_640", "attn", dict(C=640, heads=10, ctx=None, N=1, L=512, Lc=None)),
    ("attn_cross_640", "attn", dict(C=640, heads=10, ctx=1024, N=2, L=64, Lc=77)),
    ("ff_320", "ff", dict(C=320, N=2, L=64)),
    ("tblock_640", "tblock", dict(C=640, heads=10, ctx=1024, N=2, L=64, Lc=77)),
    ("st_320", "st", dict(C=320, heads=5, ctx=1024, N=2, H=8, W=8, Lc=77)),
]


def op_inputs(name, kind, p):
    """Deterministic inputs for an operator case (fp32 torch tensors)."""
    if kind in ("gn_silu", "normalize"):
        return dict(x=T(name + ".x", (p["N"], p["C"], p["H"], p["W"])))
    if kind == "layernorm":
        return dict(x=T(name + ".x", (p["N"], p["L"], p["C"])))
    if kind in ("conv3x3", "conv1x1"):
        return dict(x=T(name + ".x", (p["N"], p["Cin"], p["H"], p["W"])))
    if kind in ("down", "up"):
        return dict(x=T(name + ".x", (p["N"], p["C"], p["H"], p["W"])))
    if kind == "res":
        return dict(x=T(name + ".x", (p["N"], p["Cin"], p["H"], p["W"])), emb=T(name + ".emb", (p["N"], 1280)))
    if kind in ("attn", "tblock"):
        d = dict(x=T(name + ".x", (p["N"], p["L"], p["C"])))
        if p["ctx"]:
            d["ctx"] = T(name + ".ctx", (p["N"], p["Lc"], p["ctx"]))
        return d
    if kind == "ff":
        return dict(x=T(name + ".x", (p["N"], p["L"], p["C"])))
    if kind == "st":
        return dict(x=T(name + ".x", (p["N"], p["C"], p["H"], p["W"])), ctx=T(name + ".ctx", (p["N"], p["Lc"], p["ctx"])))
    raise ValueError(kind)


def op_shapes(kind, p):
    """state-dict shapes (reference key names relative to the module) for an operator case."""
    s = {}
    if kind in ("gn_silu", "normalize"):
        s = {"weight": (p["C"],), "bias": (p["C"],)}
    elif kind == "layernorm":
        s = {"weight": (p["C"],), "bias": (p["C"],)}
    elif kind == "conv3x3":
        s = {"weight": (p["Cout"], p["Cin"], 3, 3), "bias": (p["Cout"],)}
    elif kind == "conv1x1":
        s = {"weight": (p["Cout"], p["Cin"], 1, 1), "bias": (p["Cout"],)}
    elif kind == "down":
        s = {"op.weight": (p["C"], p["C"], 3, 3), "op.bias": (p["C"],)}
    elif kind == "up":
        s = {"conv.weight": (p["C"], p["C"], 3, 3), "conv.bias": (p["C"],)}
    elif kind == "res":
        unet_ref._res_shapes("m", p["Cin"], p["Cout"], 1280, s)
        s = {k[2:]: v for k, v in s.items()}
    elif kind == "attn":
        c, cd = p["C"], p["ctx"] or p["C"]
        s = {"to_q.weight": (c, c), "to_k.weight": (c, cd), "to_v.weight": (c, cd), "to_out.0.weight": (c, c),
             "to_out.0.bias": (c,)}
    elif kind == "ff":
        c = p["C"]
        s = {"net.0.proj.weight": (8 * c, c), "net.0.proj.bias": (8 * c,), "net.2.weight": (c, 4 * c),
             "net.2.bias": (c,)}
    elif kind in ("tblock", "st"):
        cfg = unet_ref.UNetConfig(context_dim=p["ctx"])
        full = {}
        unet_ref._st_shapes("m", p["C"], cfg, full)
        if kind == "tblock":
            pre = "m.transformer_blocks.0."
            s = {k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}
        else:
            s = {k[2:]: v for k, v in full.items()}
    return s


def op_state(name, kind, p):
    """{relative key: tensor} filled from names `<case>.<key>`."""
    return weights.fill_state_dict(op_shapes(kind, p), prefix=name + ".")


def op_oracle(name, kind, p, mode="fp32"):
    """Run the oracle restatement for an operator case. Returns fp32 tensor (reference layout)."""
    import torch.nn.functional as F

    m = unet_ref._Mode(mode)
    sd = {"m." + k: v for k, v in op_state(name, kind, p).items()}
    i = op_inputs(name, kind, p)
    cfg = unet_ref.UNetConfig(context_dim=p.get("ctx") or 1024)
    if kind == "gn_silu":
        return F.silu(unet_ref.group_norm(i["x"], sd["m.weight"], sd["m.bias"], 1e-5))
    if kind == "normalize":
        return unet_ref.group_norm(i["x"], sd["m.weight"], sd["m.bias"], 1e-6)
    if kind == "layernorm":
        return unet_ref.layer_norm(i["x"], sd["m.weight"], sd["m.bias"])
    if kind == "conv3x3":
        return F.conv2d(i["x"], sd["m.weight"], sd["m.bias"], padding=1)
    if kind == "conv1x1":
        return F.conv2d(i["x"], sd["m.weight"], sd["m.bias"])
    if kind == "down":
        return unet_ref.run_layer(sd, ("down", "m", p["C"]), i["x"], None, None, m, cfg)
    if kind == "up":
        return unet_ref.run_layer(sd, ("up", "m", p["C"]), i["x"], None, None, m, cfg)
    if kind == "res":
        return unet_ref.resblock(sd, "m", i["x"], i["emb"], m)
    if kind == "attn":
        return unet_ref.cross_attention(sd, "m", i["x"], i.get("ctx", i["x"]), p["heads"], m)
    if kind == "ff":
        return unet_ref.feed_forward(sd, "m", i["x"], m)
    if kind == "tblock":
        return unet_ref.transformer_block(sd, "m", i["x"], i["ctx"], p["heads"], m, cfg)
    if kind == "st":
        return unet_ref.spatial_transformer(sd, "m", i["x"], i["ctx"], p["heads"], m, cfg)
    raise ValueError(kind)


# ---- G4: whole-UNet cases ------------------------------------------------------------------------------------
# (case, config name, N, H, W, timesteps)
UNET_CASES = [
    ("unet_small_32x64", "SMALL", 2, 32, 64, [981, 1]),
    ("unet_small_16x32_b4", "SMALL", 4, 16, 32, [481, 481, 21, 21]),
    ("unet_mid_16x32", "MID", 2, 16, 32, [981, 1]),
    ("unet_mid_32x64_b4", "MID", 4, 32, 64, [481, 481, 21, 21]),
    ("unet_full_8x16", "FULL", 2, 8, 16, [981, 1]),
    ("unet_full_16x32", "FULL", 2, 16, 32, [501, 501]),
]
CONFIGS = {"SMALL": unet_ref.SMALL, "MID": unet_ref.MID, "FULL": unet_ref.FULL}


def unet_inputs(case, cfg, N, H, W, ts):
    x = T(case + ".x", (N, cfg.in_channels, H, W))
    ctx = T(case + ".ctx", (N, 77, cfg.context_dim))
    return x, torch.tensor(ts, dtype=torch.long), ctx


def unet_state(cfg_name):
    """Weights depend only on the config (prefix = config name) so several cases share one fill."""
    cfg = CONFIGS[cfg_name]
    return weights.fill_state_dict(unet_ref.param_shapes(cfg), prefix=f"unet.{cfg_name}.")


# ---- G5: multi-view cases --------------------------------------------------------------------------------------
# (case, view_num, concat_target, batch b, per-canvas H, W)   -- UNet batch = b * (view_num-1 if concat_target else view_num)
MV_CASES = [
    ("mv_v5_concat", 5, True, 1, 8, 16),
    ("mv_v2_plain", 2, False, 2, 8, 8),
    ("mv_v4_plain", 4, False, 1, 8, 8),
    ("mv_v2_concat", 2, True, 2, 8, 16),   # one canvas per sample: the 1-rank case of the canvas-sharded path
]


def mv_config(view_num, concat_target):
    return unet_ref.UNetConfig(model_channels=128, num_head_channels=64, context_dim=256, multiview=True,
                               view_num=view_num, concat_target=concat_target)


# ---- G6/G7: sampler cases ----------------------------------------------------------------------------------------
STEP_CASES = [("step_eta0", 50, 0.0, 17), ("step_eta1", 50, 1.0, 49), ("step_eta1_last", 10, 1.0, 0)]  # (case,S,eta,index)
TRAJ_CASES = [("traj_s10", 10, 0.0, 1, 8, 16), ("traj_s50", 50, 0.0, 1, 8, 16), ("traj_s10_eta1_b2", 10, 1.0, 2, 8, 16)]
MULTI_CASES = [("multi_k3_s10", 10, 1.0, 1, 8, 16, 3, 1234)]   # (case, S, eta, B, h, w, K conditionings, random.seed)
TRAIN_CASES = [("train_b2", 2, 16, 32, [501, 21])]   # (case, B, h, w, timesteps): p_losses + backward to the context
CFG_SCALE = 2.5
TRAJ_CONFIG = "MID"


# ---- prompt encoder: constructor kwargs as the shipped YAMLs pass them (configs/ref_inpainting.yaml:60-72,
# novel_view_synthesis.yaml:61-75) + prompts; run through the reference's PromptCLIPEmbedder on oracle/clip_stub.py ----------
_TXT = "The whole image is splited into two parts with the same size, they share the same scene captured with different viewpoints"
TEXT_CASES = [
    ("txt_repeat8_pen", dict(layer="penultimate", special_tokens=["repeat_8_<special-token>"], init_text=[_TXT]),
     ["".join(f"<special-token{i}>" for i in range(8)), "", "a photo of <special-token3> and a cat"]),
    ("txt_lr_last", dict(layer="last", special_tokens=["<left>", "<right>"], init_text=["left part", "right part"]),
     ["<left> is the reference , <right> is the target", "plain prompt without special tokens " * 12]),
    ("txt_tokenwise", dict(layer="penultimate", special_tokens=["repeat_5_<special-token>"], init_text=["alpha beta gamma"],
                           tokenwise_init=True),
     ["<special-token0><special-token1><special-token4> tail"]),
    ("txt_deep", dict(layer="penultimate", special_tokens=["repeat_3_<special-token>"], init_text=["deep prompt init"],
                      deep_prompt=True, cross_attn_layers=4),
     [["".join(f"<special-token{i}-layer{l}>" for i in range(3)), ""] for l in range(4)]),
]

# multi-view prompt encoder (multiview_Refill_modules.PromptCLIPEmbedder: per-view learned tokens, one prompt list per view) and the
# NVS encoder (NVS_modules.NVSCLIPEmbedder: pose token from RelPosModel, optional second pose head / per-view tokens); the prompts
# follow the reference's callers (multiview_ref_inpainting_ldm.py / NVS_ldm.py build them from the token names)
_MV_KW = dict(layer="penultimate", special_tokens=["repeat_4_<special-token>"], init_text=[_TXT], view_num=3, view_token_len=2)
_MV_SP = "".join(f"<special-token{i}>" for i in range(4))
MV_TEXT_CASES = [
    ("txt_mv_views", _MV_KW, [[_MV_SP + "".join(f"<view_direct-{j}-{l}" for l in range(2)), ""] for j in range(3)]),
    ("txt_mv_plain", dict(layer="last", special_tokens=["<left>", "<right>"], init_text=["left part", "right part"], view_prompt=False),
     ["<left> and <right>", "no special token here"]),
]
_NVS_SP = "".join(f"<special-token{i}>" for i in range(6))
NVS_Z_ROWS = [0, 1, 6, 7, 8, 40, 75, 76]      # positions of z kept in the fixture of the 1024-wide cases (7 = the pose slot, 76 = the 2nd head's)


def nvs_pose_state(name, like):
    """RNG-free values for RelPosModel's parameters (the reference draws them from torch's default init)."""
    return {k: torch.from_numpy(weights.fill_like(f"{name}.rel_pos_model.{k}", tuple(v.shape))) for k, v in like.items()}

NVS_TEXT_CASES = [
    # name, kwargs, prompts, rel_pos shape (None: prompts only)
    ("txt_nvs_pose", dict(arch="stub-1024", layer="penultimate", special_tokens=["repeat_6_<special-token>"], init_text=[_TXT]),
     [_NVS_SP, _NVS_SP, ""], (3, 4)),
    ("txt_nvs_pose2", dict(arch="stub-1024", layer="penultimate", special_tokens=["repeat_6_<special-token>"], init_text=[_TXT],
                           pos_strengthen=True),
     [_NVS_SP, _NVS_SP], (2, 4)),
    ("txt_nvs_views", dict(layer="last", special_tokens=["repeat_2_<special-token>"], init_text=[_TXT], view_prompt=True, view_num=2,
                           view_token_len=1), ["<special-token0><special-token1><view_direct-1-0>", "<view_direct-0-0>"], None),
]


# ---- NVS task model (inpainting_ldm/NVS_ldm.py:22-104): NVSUnetModel at the shipped width (the separator tokens are keyed by the
# SD2 channel counts), latent 8x16 ----------------------------------------------------------------------------------------------
# (case, use_sep, c_input shape or None, N, H, W, timesteps)
NVS_UNET_CASES = [
    ("nvs_sep_cinput", True, (2, 320, 8, 17), 2, 8, 16, [981, 21]),      # separator column + c_input over the whole (W + 1) canvas
    ("nvs_cinput_half", False, (2, 320, 8, 8), 2, 8, 16, [501, 501]),   # plain UNet + c_input on the right half
    ("nvs_sep_only", True, None, 2, 8, 16, [701, 301]),
]
NVS_SEP_CHANNELS = (9, 320, 640, 1280, 2560, 1920, 960)


def nvs_unet_state(use_sep):
    sd = dict(unet_state("FULL"))
    if use_sep:
        for ch in NVS_SEP_CHANNELS:
            sd[f"sep_token.{ch}"] = torch.from_numpy(weights.fill_like(f"nvs.sep_token.{ch}", (ch,), "normalish"))
    return sd


def nvs_unet_inputs(case, c_shape, N, H, W, ts):
    x, t, ctx = unet_inputs(case, CONFIGS["FULL"], N, H, W, ts)
    c_input = None if c_shape is None else T(case + ".c_input", c_shape) * 0.5
    return x, t, ctx, c_input
