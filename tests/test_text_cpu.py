"""Prompt encoder glue (CPU): the drop-in PromptCLIPEmbedder against golden vectors produced by the REFERENCE's own
PromptCLIPEmbedder (oracle/make_golden_text.py -> tests/golden/text.npz), both on oracle/clip_stub.py, the stand-in for the
un-vendored `open_clip` package: token ids, special-token expansion, initial special embeddings, the splice, deep prompts and
the transformer (PyTorch path here; the HIP tower is checked against the same goldens in tests/test_gpu_text.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import clip_stub, golden_spec as G


@pytest.fixture()
def embedder_cls():
    import leftrefill_amd.dropin as dropin
    dropin.install()
    old = sys.modules.get("open_clip")
    sys.modules["open_clip"] = clip_stub
    from ldm.modules.encoders.Refill_modules import PromptCLIPEmbedder
    yield PromptCLIPEmbedder
    if old is None:
        sys.modules.pop("open_clip", None)
    else:
        sys.modules["open_clip"] = old


def _gold():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text.npz"))


@pytest.mark.parametrize("name,kw,prompts", G.TEXT_CASES, ids=[c[0] for c in G.TEXT_CASES])
def test_prompt_embedder_matches_reference(embedder_cls, name, kw, prompts):
    from ldm.modules.encoders import Refill_modules as R
    g = _gold()
    emb = embedder_cls(device="cpu", **kw).eval()
    assert torch.equal(emb.special_embeddings.weight.detach(), torch.from_numpy(g[name + ".special_embeddings"]))
    if kw.get("deep_prompt"):
        toks = torch.stack([R.tokenize(emb.tokenizer, p) for p in prompts], dim=1)
    else:
        toks = R.tokenize(emb.tokenizer, prompts)
    assert torch.equal(toks, torch.from_numpy(g[name + ".tokens"]))          # bit-exact token ids
    assert (toks >= emb.vocab_size).any(), "the case must exercise the learned tokens"
    with torch.no_grad():
        z = emb(prompts)
    ref = torch.from_numpy(g[name + ".z"])
    assert z.shape == ref.shape
    assert torch.allclose(z, ref, rtol=1e-5, atol=1e-5), (z - ref).abs().max()
    for p in emb.parameters():
        assert not p.requires_grad        # freeze() covers the special embeddings too (reference 149-152)


def test_tokenize_truncates_and_keeps_end_token(embedder_cls):
    from ldm.modules.encoders import Refill_modules as R
    tok = clip_stub.SimpleTokenizer(special_tokens=["<a>"])
    t = R.tokenize(tok, ["word " * 200, "", "<a>"])
    assert t.shape == (3, 77) and t.dtype == torch.long
    sot, eot = tok.encoder["<start_of_text>"], tok.encoder["<end_of_text>"]
    assert t[0, 0] == sot and t[0, -1] == eot and (t[0, 1:-1] != 0).all()
    assert t[1].tolist() == [sot, eot] + [0] * 75
    assert t[2, 1] == tok.encoder["<a>"] and t[2, 2] == eot


def test_random_init_and_unconditional_prompt(embedder_cls):
    emb = embedder_cls(device="cpu", layer="penultimate", special_tokens=["repeat_4_<special-token>"], init_text=["<random>"])
    assert emb.special_embeddings.weight.shape == (4, clip_stub.WIDTH)
    assert emb.special_tokens == [f"<special-token{i}>" for i in range(4)]
    with torch.no_grad():
        z = emb([""] * 3)          # get_unconditional_conditioning (ref_inpainting_ldm.py:31-35)
    assert z.shape == (3, 77, clip_stub.WIDTH) and torch.isfinite(z).all()


@pytest.fixture()
def stub_open_clip():
    import leftrefill_amd.dropin as dropin
    dropin.install()
    old = sys.modules.get("open_clip")
    sys.modules["open_clip"] = clip_stub
    yield
    if old is None:
        sys.modules.pop("open_clip", None)
    else:
        sys.modules["open_clip"] = old


@pytest.mark.parametrize("name,kw,prompts", G.MV_TEXT_CASES, ids=[c[0] for c in G.MV_TEXT_CASES])
def test_multiview_prompt_embedder_matches_reference(stub_open_clip, name, kw, prompts):
    """multiview_Refill_modules.PromptCLIPEmbedder (per-view learned tokens, one prompt list per view) vs the reference's own class
    on the open_clip stand-in (oracle/make_golden_text.py)."""
    from ldm.modules.encoders.multiview_Refill_modules import PromptCLIPEmbedder
    g = _gold()
    emb = PromptCLIPEmbedder(device="cpu", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}).eval()
    assert len(emb.special_tokens) == int(g[name + ".n_special"])
    assert torch.equal(emb.special_embeddings.weight.detach(), torch.from_numpy(g[name + ".special_embeddings"]))
    with torch.no_grad():
        z = emb(prompts)
    ref = torch.from_numpy(g[name + ".z"])
    assert z.shape == ref.shape                      # view prompts: [B * view, 77, C], one context per canvas
    assert torch.allclose(z, ref, rtol=1e-5, atol=1e-5), (z - ref).abs().max()


@pytest.mark.parametrize("name,kw,prompts,pose_shape", G.NVS_TEXT_CASES, ids=[c[0] for c in G.NVS_TEXT_CASES])
def test_nvs_prompt_embedder_matches_reference(stub_open_clip, name, kw, prompts, pose_shape):
    """NVS_modules.NVSCLIPEmbedder: pose token from RelPosModel spliced over the last learned token, second pose head on the last
    output position (pos_strengthen), per-view tokens -- vs the reference's own class on the stand-in."""
    from ldm.modules.encoders.NVS_modules import NVSCLIPEmbedder
    g = _gold()
    emb = NVSCLIPEmbedder(device="cpu", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}).eval()
    assert torch.equal(emb.special_embeddings.weight.detach(), torch.from_numpy(g[name + ".special_embeddings"]))
    if emb.rel_pos_model is not None:
        emb.rel_pos_model.load_state_dict(G.nvs_pose_state(name, emb.rel_pos_model.state_dict()))
    with torch.no_grad():
        z = emb(prompts) if pose_shape is None else emb([prompts, G.T(name + ".rel_pos", pose_shape)])
    ref = torch.from_numpy(g[name + ".z"])
    if z.shape[-1] > 256:
        z = z[:, G.NVS_Z_ROWS]
    assert z.shape == ref.shape
    assert torch.allclose(z, ref, rtol=1e-5, atol=2e-5), (z - ref).abs().max()
    if pose_shape is not None:      # the pose really lands where the reference puts it
        with torch.no_grad():
            z2 = emb([prompts, G.T(name + ".rel_pos", pose_shape) + 1.0])
        z2 = z2[:, G.NVS_Z_ROWS] if z2.shape[-1] > 256 else z2
        assert not torch.allclose(z2, ref)


def test_nvs_training_cfg_dropout_replaces_whole_samples(stub_open_clip):
    from ldm.modules.encoders.NVS_modules import NVSCLIPEmbedder
    name, kw, prompts, pose_shape = G.NVS_TEXT_CASES[1]
    emb = NVSCLIPEmbedder(device="cpu", cfg_rate=1.0, freeze=False, **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
    emb.train()
    with torch.no_grad():
        z_drop = emb([prompts, G.T(name + ".rel_pos", pose_shape)])
        emb.eval()
        z_null = emb([""] * len(prompts))
    # cfg_rate = 1: every sample is encoded as the empty prompt, and the second pose head is replaced by the encoder's own output
    assert torch.allclose(z_drop, z_null, atol=1e-5)
