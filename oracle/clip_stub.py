"""Stand-in for the un-vendored third-party package `open_clip` (open_clip_torch==2.16.0, reference requirements.txt:18) --
TEST INFRASTRUCTURE ONLY (imported by oracle/make_golden_text.py and tests/, never by the product).

The reference's prompt encoder (ldm/modules/encoders/Refill_modules.py) uses exactly this surface of open_clip:
  * `create_model_and_transforms(arch, device=, pretrained=)` -> (model, _, _) with `model.visual`, `.token_embedding`,
    `.vocab_size`, `.positional_embedding`, `.transformer.resblocks[i](x, attn_mask=)`, `.transformer.grad_checkpointing`,
    `.ln_final`, `.attn_mask`;
  * `SimpleTokenizer(special_tokens=[...])` with `.encoder` (token -> id, containing '<start_of_text>', '<end_of_text>' and
    the special tokens appended AFTER them, so special ids >= model.vocab_size) and `.encode(text)`.
The package and its ViT-H-14 weights / BPE vocabulary are absent from this image and there is no network.  This module
re-creates that surface with the published architecture of the text tower (pre-LN residual attention blocks,
nn.MultiheadAttention, GELU MLP, causal mask) at a small width and a toy word-level vocabulary, so that the reference's OWN
Python (tokenize, special-token splice, init_special_embeddings, deep prompts, encode_with_transformer) can be executed to
produce golden vectors, and the drop-in can be run against the same stand-in.  The weights are the RNG-free fills of
oracle/weights.py.
"""
import re
import zlib

import torch
import torch.nn as nn

from . import weights

WIDTH, HEADS, LAYERS, CTX, BASE_VOCAB = 256, 4, 3, 77, 510   # vocab_size = BASE_VOCAB + 2 = 512 (sot 510, eot 511)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, 4 * d))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(4 * d, d))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width=WIDTH, heads=HEADS, layers=LAYERS):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])
        self.grad_checkpointing = False


# arch "stub-1024": the real tower's width (the NVS encoder's pose MLP is hard-wired to 1024 channels, NVS_modules.py:167), 2 layers
ARCHS = {"stub-1024": (1024, 16, 2)}


class TextModel(nn.Module):
    def __init__(self, width=WIDTH, heads=HEADS, layers=LAYERS):
        super().__init__()
        WIDTH = width
        self.vocab_size = BASE_VOCAB + 2
        self.token_embedding = nn.Embedding(self.vocab_size, WIDTH)
        self.positional_embedding = nn.Parameter(torch.zeros(CTX, WIDTH))
        self.transformer = _Transformer(width, heads, layers)
        self.ln_final = nn.LayerNorm(WIDTH)
        self.visual = nn.Identity()
        self.register_buffer("attn_mask", torch.full((CTX, CTX), float("-inf")).triu_(1), persistent=False)
        sd = {k: torch.from_numpy(weights.fill_like("clipstub." + k, v.shape)) for k, v in self.state_dict().items()}
        sd["token_embedding.weight"] = torch.from_numpy(weights.fill_like("clipstub.tok", (self.vocab_size, WIDTH), "normalish")) * 0.5
        sd["positional_embedding"] = torch.from_numpy(weights.fill_like("clipstub.pos", (CTX, WIDTH), "normalish")) * 0.1
        self.load_state_dict(sd)


def create_model_and_transforms(arch, device=None, pretrained=None):
    return TextModel(*ARCHS.get(arch, (WIDTH, HEADS, LAYERS))), None, None


class SimpleTokenizer:
    def __init__(self, special_tokens=None):
        specials = ["<start_of_text>", "<end_of_text>"] + list(special_tokens or [])
        self.encoder = {f"w{i}": i for i in range(BASE_VOCAB)}
        for t in specials:
            self.encoder[t] = len(self.encoder)
        pat = "|".join(re.escape(t) for t in specials)
        self.pat = re.compile(pat + r"|[a-z0-9]+|[^\s a-z0-9]", re.IGNORECASE)

    def encode(self, text):
        out = []
        for tok in re.findall(self.pat, text.strip().lower()):
            out.append(self.encoder[tok] if tok in self.encoder else zlib.crc32(tok.encode("utf-8")) % BASE_VOCAB)
        return out
