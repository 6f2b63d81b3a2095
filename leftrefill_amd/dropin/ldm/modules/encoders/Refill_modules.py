"""`ldm.modules.encoders.Refill_modules.PromptCLIPEmbedder` (reference Refill_modules.py:91-204).

Prompt-token glue: OpenCLIP ViT-H/14 text tower (penultimate layer) with the 50 learned `<special-token>` embeddings
spliced into the token embeddings.  It runs once per batch and its output [B, 77, 1024] is an INPUT of the hot path.
Tokenizer, pretrained weights and the module definitions come from the third-party `open_clip` package, which is not in
this image: the class is import-gated.  With it installed, the transformer part (positional embedding, residual attention
blocks under the causal mask, ln_final) runs on the HIP kernels for CUDA inference (leftrefill_amd/text_engine.py: causal
attention + GELU-epilogue GEMMs, checked against a PyTorch module of the published architecture, tests/test_gpu_text.py);
when autograd is recording (training of the special tokens) the PyTorch modules run so that torch.autograd applies.
"""
import torch
import torch.nn as nn


class PromptCLIPEmbedder(nn.Module):
    LAYERS = ["last", "penultimate"]
    use_hip = True

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", special_tokens=None, init_text=None, tokenwise_init=False, deep_prompt=False,
                 cross_attn_layers=16, **kwargs):
        super().__init__()
        try:
            import open_clip
        except ImportError as e:  # pragma: no cover - open_clip is not installed in the build image
            raise ImportError("PromptCLIPEmbedder needs the `open_clip` package (open_clip_torch==2.16.0, "
                              "requirements.txt:18) and its ViT-H-14 weights; pass precomputed [B,77,1024] contexts to "
                              "the sampler instead (see bench.py / tests).") from e
        assert layer in self.LAYERS
        specials = []
        for tok in (special_tokens or []):
            if tok.startswith("repeat_"):
                _, n, name = tok.split("_", 2)
                base = name[:-1] if name.endswith(">") else name
                specials.extend(f"{base}{i}>" for i in range(int(n)))
            else:
                specials.append(tok)
        self.special_tokens = specials
        model, _, _ = open_clip.create_model_and_transforms(arch, device=torch.device("cpu"), pretrained=version)
        del model.visual
        self.model = model
        self.tokenizer = open_clip.SimpleTokenizer(special_tokens=specials) if specials else open_clip.tokenize
        self.vocab_size = model.token_embedding.weight.shape[0]
        self.special_embeddings = nn.Embedding(len(specials), model.token_embedding.weight.shape[1])
        self.device = device
        self.max_length = max_length
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        if freeze:
            self.model.eval()
            for p in self.model.parameters():
                p.requires_grad = False

    def forward(self, text):
        tokens = self.tokenizer(text).to(self.special_embeddings.weight.device)
        is_special = tokens >= self.vocab_size
        x = self.model.token_embedding(tokens.clamp(max=self.vocab_size - 1))
        if is_special.any():
            x = torch.where(is_special[..., None], self.special_embeddings((tokens - self.vocab_size).clamp(min=0)), x)
        if x.is_cuda and self.use_hip and not (torch.is_grad_enabled() and x.requires_grad):
            from leftrefill_amd import text_engine
            sig = tuple((p_.data_ptr(), p_._version) for p_ in self.model.parameters())
            if getattr(self, "_lr_tower", None) is None or self._lr_tower[0] != sig:
                self._lr_tower = (sig, text_engine.PackedTextTower(self.model, self.layer_idx))
            return text_engine.encode_with_transformer(x, self._lr_tower[1])
        x = x + self.model.positional_embedding
        x = x.permute(1, 0, 2)
        blocks = self.model.transformer.resblocks
        for i, r in enumerate(blocks):
            if i == len(blocks) - self.layer_idx:
                break
            x = r(x, attn_mask=self.model.attn_mask)
        return self.model.ln_final(x.permute(1, 0, 2))

    def encode(self, text):
        return self(text)
