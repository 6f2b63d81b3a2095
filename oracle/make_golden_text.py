"""Golden vectors of the reference's prompt encoder (authoring container only; writes tests/golden/text.npz).

Imports /root/reference/ldm/modules/encoders/Refill_modules.py with `open_clip` bound to oracle/clip_stub.py (the real
package is un-vendored and absent) and runs the reference's own PromptCLIPEmbedder: tokenisation, special-token splice,
init_special_embeddings, repeat_ / deep-prompt expansion, encode_with_transformer.  Only inputs-independent outputs are
stored (token ids, initial special embeddings, encoder outputs); the prompts live in oracle/golden_spec.TEXT_CASES.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import clip_stub, golden_spec as G, ref_import  # noqa: E402


def main():
    sys.modules["open_clip"] = clip_stub
    ref_import.install_stubs()
    if ref_import.REF not in sys.path:
        sys.path.insert(0, ref_import.REF)
    for k in list(sys.modules):
        if k == "ldm" or k.startswith("ldm."):
            del sys.modules[k]
    from ldm.modules.encoders import Refill_modules as R
    out = {}
    for name, kw, prompts in G.TEXT_CASES:
        torch.manual_seed(0)
        emb = R.PromptCLIPEmbedder(device="cpu", **kw)
        emb.eval()
        out[name + ".special_embeddings"] = emb.special_embeddings.weight.detach().numpy().copy()
        if kw.get("deep_prompt"):
            toks = torch.stack([R.tokenize(emb.tokenizer, p) for p in prompts], dim=1)
        else:
            toks = R.tokenize(emb.tokenizer, prompts)
        out[name + ".tokens"] = toks.numpy()
        with torch.no_grad():
            out[name + ".z"] = emb(prompts).numpy()
        print(name, out[name + ".tokens"].shape, out[name + ".z"].shape, len(emb.special_tokens))
    path = os.path.join(ROOT, "tests", "golden", "text.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
