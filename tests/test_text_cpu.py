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
