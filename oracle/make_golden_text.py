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
    # the multi-view encoder file imports four names from `transformers` it never uses on this path: resolve them before the
    # torchvision stub exists (transformers probes for torchvision at import time)
    from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5Tokenizer  # noqa: F401
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
    from ldm.modules.encoders import multiview_Refill_modules as MV
    for name, kw, prompts in G.MV_TEXT_CASES:
        torch.manual_seed(0)
        emb = MV.PromptCLIPEmbedder(device="cpu", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        emb.eval()
        out[name + ".special_embeddings"] = emb.special_embeddings.weight.detach().numpy().copy()
        out[name + ".n_special"] = np.asarray(len(emb.special_tokens))
        with torch.no_grad():
            out[name + ".z"] = emb(prompts).numpy()
        print(name, out[name + ".z"].shape, len(emb.special_tokens))
    from ldm.modules.encoders import NVS_modules as NV
    for name, kw, prompts, pose_shape in G.NVS_TEXT_CASES:
        torch.manual_seed(0)
        emb = NV.NVSCLIPEmbedder(device="cpu", **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
        emb.eval()
        out[name + ".special_embeddings"] = emb.special_embeddings.weight.detach().numpy().copy()
        if emb.rel_pos_model is not None:      # the pose MLP is randomly initialised by the reference: name-keyed fills instead
            emb.rel_pos_model.load_state_dict(G.nvs_pose_state(name, emb.rel_pos_model.state_dict()))
        with torch.no_grad():
            if pose_shape is None:
                z = emb(prompts)
            else:
                z = emb([prompts, G.T(name + ".rel_pos", pose_shape)])
        out[name + ".z"] = z[:, G.NVS_Z_ROWS].numpy() if z.shape[-1] > 256 else z.numpy()      # 1024-wide cases: sampled positions
        print(name, out[name + ".z"].shape, len(emb.special_tokens))
    path = os.path.join(ROOT, "tests", "golden", "text.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
