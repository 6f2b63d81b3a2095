"""`ldm.modules.encoders.multiview_Refill_modules.PromptCLIPEmbedder` (reference multiview_Refill_modules.py:94-227): the prompt
encoder of the multi-view task model (configs/multiview_ref_inpainting.yaml).

On top of the single-reference encoder (Refill_modules.py) it learns `view_token_len` tokens per view -- named
"<view_direct-{j}-{l}" in the reference (136-141; the closing '>' is missing there and the names are kept as they are, they are
state-dict-visible through the tokenizer) -- initialised from one fixed sentence, and its forward takes one prompt list PER
VIEW: text[view][batch] -> tokens [B, view, 77] -> z [B * view, 77, C] (185-192; unlike deep prompts the result stays flat, one
context per canvas of the '(b v)' UNet batch).  Deep prompts are not implemented by the reference for this variant (120-121).
The transformer part runs on the HIP text tower like the base class.
"""
from ldm.modules.encoders.Refill_modules import (AbstractEncoder, IdentityEncoder, PromptCLIPEmbedder as _Base,  # noqa: F401
                                                 expand_special_tokens, init_special_embeddings, tokenize)

VIEW_INIT_TEXT = ("The whole image is splited into two parts with the same size, they share the same scene/landmark captured with "
                  "different viewpoints and times")


def view_token_names(view_num, view_token_len, closing=""):
    return [f"<view_direct-{j}-{l}{closing}" for j in range(view_num) for l in range(view_token_len)]


class PromptCLIPEmbedder(_Base):
    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 special_tokens=("<left>", "<right>"), init_text=None, tokenwise_init=False, deep_prompt=False, cross_attn_layers=16,
                 view_prompt=True, view_num=4, view_token_len=30, **kwargs):
        AbstractEncoder.__init__(self)
        if deep_prompt:
            raise NotImplementedError("the multi-view prompt encoder has no deep prompts (reference multiview_Refill_modules.py:120-121)")
        names, texts = expand_special_tokens(special_tokens, init_text, False, cross_attn_layers)
        self.view_prompt = view_prompt
        if view_prompt:
            extra = view_token_names(view_num, view_token_len)
            names = names + extra
            texts = (texts or []) + [VIEW_INIT_TEXT] * len(extra)
        self._build(arch, version, device, max_length, layer, names, texts, tokenwise_init, False, cross_attn_layers)
        if freeze:
            self.freeze()

    def forward(self, text):
        tokens, _shape = self._tokens(text, self.view_prompt)     # view prompts: text[view][batch] -> [B * view, 77]
        return self.encode_with_transformer(self._embed(tokens))
